/*
 * slender_hip.h — C ABI of libslender_hip.so, the MI355X (gfx950) kernels of the SlenderObjDet training hot path.
 *
 * Conventions (every entry point):
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless a parameter says "host";
 *   - the caller owns every buffer (outputs, workspaces); nothing is allocated inside;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued on it and the call returns without synchronising;
 *   - return value: 0 = ok, SOD_EARG (-1) bad argument / unsupported shape, SOD_ESIZE (-2) tensor exceeds the
 *     32-bit buffer addressing range (2 GiB per operand), >0 = hipError_t of a failed launch;
 *   - re-entrant per stream.  Process-wide state is limited to (a) one-time kernel attribute set-up and cached device properties
 *     (the library assumes every GPU of the process is the same model; one process per GPU is the supported deployment), (b) the
 *     tile-policy and experiment knobs read from SOD_* environment variables on first use and sod_conv_set_tile256, which select
 *     between kernels with identical contracts, and (c) the profiling aids sod_conv_prof_* (process-wide list) / sod_conv_last_variant (per thread).
 *     Scratch memory is never cached inside the library: every entry point that needs a workspace takes it as an argument.
 *
 * Each declaration cites the reference interface it replaces (paths under wanzysky/SlenderObjDet; "d2" =
 * detectron2 @ 8bc84a2ff8a0b5787ec and "fvcore" are the un-vendored third-party packages the reference calls,
 * see SURVEY.md §0 / §2.3).
 */
#ifndef SLENDER_HIP_H_
#define SLENDER_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

#define SOD_MAX_LEVELS 8
#define SOD_CONV_MAX_LEVELS 6   /* tensors one multi-level conv launch may cover */

/* conv flags */
#define SOD_CONV_RELU 1     /* y = max(y, 0) after bias/residual */
#define SOD_CONV_RES_UP2 2  /* residual is half resolution and nearest-2x upsampled (FPN top-down path) */
#define SOD_CONV_CWIN 4     /* grouped convolution in CHANNEL-WINDOW mode (ResNeXt 3x3: detectron2 BottleneckBlock(num_groups), configs/
                             * ablation_studies/pointset/base_X101.yaml): C == K multiples of 128, groups whose channels divide 128; w is
                             * [K][R][S][128] - row q holds, per tap, its weights towards the 128 input channels of q's own 128-channel
                             * tile (block-diagonal inside the window), so a 128-wide output tile contracts over 128 instead of C channels:
                             * 128 / (C / groups) x the FLOPs of a true grouped kernel instead of groups x for the dense embedding */

/* iou_loss types — slender_det/layers/iou_loss.py:25-32 */
#define SOD_IOU_LOSS_IOU 0
#define SOD_IOU_LOSS_LINEAR 1
#define SOD_IOU_LOSS_GIOU 2

const char* sod_version(void);
/* ---------------------------------------------------------------------------------------------------------
 * Streams.  No reference counterpart: the reference runs its whole step on PyTorch's default CUDA stream (train_net.py:185-195 ->
 * detectron2 SimpleTrainer.run_step); the MI355X step overlaps independent launch chains on side streams, and a side
 * stream may be confined to a subset of the compute units so that its whole-CU workgroups cannot displace the main stream's.
 * mask_words: HOST array of nwords 32-bit words, bit i = logical CU i as hipExtStreamCreateWithCUMask numbers them (on MI355X bit i lies on
 * XCD i % 8, so a contiguous range of 8*k bits is k CUs on every XCD).  *out_stream receives a hipStream_t the caller destroys with
 * sod_stream_destroy (after synchronising it). */
int sod_stream_create_cumask(const unsigned* mask_words, int nwords, void** out_stream);
int sod_stream_destroy(void* stream);
/* Rehearsal aid (no reference counterpart): `wgs` (1..256) workgroups of 256 threads stay resident on `stream` for `usec` (1..100000)
 * microseconds - the CU occupancy of RCCL's channel kernels while a bucket is reduced, emulated on a one-GPU box
 * (bench.py --rccl-rehearsal --rehearsal-occupancy). */
int sod_debug_occupy(int wgs, int usec, void* stream);

/* bytes of the scratch buffer `ws` the reduction-type entry points need */
long long sod_reduce_workspace_bytes(void);

/* ---------------------------------------------------------------------------------------------------------
 * Convolution, NHWC bf16, weights [K][R][S][C] bf16, fp32 accumulation on MFMA.
 * Replaces ATen conv2d forward/backward under d2 ResNet/FPN (slender_det/modeling/backbone/fpn.py:94-115)
 * and FCOSHead (slender_det/modeling/meta_arch/fcos/fcosv2.py:277-381).
 *   y[n,ho,wo,k] = act( sum_{r,s,c} x[n,ho*stride-pad+r*dil, wo*stride-pad+s*dil, c] * w[k,r,s,c] + bias[k] + res )
 * C must be a multiple of 8. *_img_stride = elements between consecutive images (<=0: dense), which lets the
 * per-level head outputs land directly in the concatenated (N, sum(Hi*Wi), K) buffer the losses read
 * (replaces permute_and_concat, slender_det/modeling/meta_arch/fcos/utils.py:32-52).
 * out_f32: write fp32 instead of bf16.
 * --------------------------------------------------------------------------------------------------------- */
int sod_conv2d_fwd(const void* x, const void* w, const float* bias, const void* res, void* y,
                   int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                   long long x_img_stride, long long y_img_stride, long long res_img_stride,
                   int flags, int out_f32, void* stream);
/* dx[n,h,w,c] = sum dy * w  (+ accum) masked by relu_mask>0.  wt is the transposed copy [C][R][S][K] bf16.
 * K (channels of dy) must be a multiple of 8 (pad the gradient rows). */
int sod_conv2d_dgrad(const void* dy, const void* wt, const void* accum, const void* relu_mask, void* dx,
                     int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                     long long dy_img_stride, long long dx_img_stride, void* stream);
/* dw[k,r,s,c] += qscale[k] * sum_{n,ho,wo} dy[n,ho,wo,k] * x[n,hi,wi,c]   (fp32, accumulated: zero it once per step; qscale
 * optional = the folded FrozenBatchNorm2d scale, chain rule through w_eff = w * scale).
 * ws / ws_bytes: caller-owned scratch (16-B aligned; sod_conv2d_wgrad_workspace_bytes() is enough for every shape), used in stream
 * order by this call only.  With it, shapes with K and C multiples of 256 run on the 256x256 8-wave kernel, whose pixel splits meet
 * in fp32 slabs summed in a fixed order; the other shapes add their splits into dw with fp32 atomics.  ws == NULL: atomics only.
 * flags: SOD_WGRAD_DETERMINISTIC = never use atomics (every shape goes through slabs; SOD_EARG if the workspace is too small), the
 * result is then bit-identical from run to run.  splits > 0 forces the 128x128 kernel with that many pixel splits, splits < 0 the
 * 256x256 kernel (SOD_EARG when the shape or the workspace does not allow it); 0 = the library chooses. */
#define SOD_WGRAD_DETERMINISTIC 1
#define SOD_WGRAD_DIAG 2    /* channel-window mode (see SOD_CONV_CWIN): dw is [K][R][S][128] and only the tiles of a q-tile's own channels run */
long long sod_conv2d_wgrad_workspace_bytes(void);
int sod_conv2d_wgrad(const void* dy, const void* x, float* dw, const float* qscale,
                     int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                     long long dy_img_stride, long long x_img_stride, int splits, int flags,
                     void* ws, long long ws_bytes, void* stream);

/* 1-bit ReLU masks for the bottleneck backward (detectron2 BottleneckBlock: out = relu(conv3 + shortcut), whose mask the backward
 * pass of the NEXT block's first conv applies to d(out)): sod_conv2d_fwd_bits is sod_conv2d_fwd with dense bf16 output that also
 * writes bit i of relu_bits (uint8[N*Ho*Wo*K/8], bit e of byte j = element 8j+e) = "stored y[i] > 0"; sod_conv2d_dgrad_bits is
 * sod_conv2d_dgrad whose mask operand is such a bit array of dx's shape (1/16 of the bytes of the bf16 tensor it replaces).
 * accum_even != 0: accum is (N, H/2, W/2, C) and is added at EVEN (h, w) only - the compact data gradient of the next stage's
 * stride-2 1x1 convolutions (STRIDE_IN_1X1), which the un-fused path scatters into a zero tensor of dx's shape first. */
int sod_conv2d_fwd_bits(const void* x, const void* w, const float* bias, const void* res, void* y, void* relu_bits,
                        int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil, int flags, void* stream);
int sod_conv2d_dgrad_bits(const void* dy, const void* wt, const void* accum, int accum_even, const void* relu_bits, void* dx,
                          int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil, void* stream);

/* Multi-level forms: ONE launch applies the same weights to `nlev` tensors (the FPN levels the FCOS towers and
 * prediction convs share, fcosv2.py:358-380 loops over them). x/y/dy/dx are HOST arrays of device pointers, H/W host
 * arrays of the per-level input sizes. y_img_stride / dy_img_stride (elements, <=0: dense per level) is common to all
 * levels, so the per-level outputs can land inside the concatenated (N, sum Hi*Wi, K) buffer. */
int sod_conv2d_fwd_ml(int nlev, const void* const* x, const void* w, const float* bias, void* const* y,
                      int N, const int* H, const int* W, int C, int K, int R, int S, int stride, int pad, int dil,
                      long long y_img_stride, int flags, int out_f32, void* stream);
/* sod_conv2d_fwd_ml with bf16 output that ALSO gathers the GroupNorm statistics of what it stores: the FCOS / RetinaNet-GN tower unit is
 * conv3x3 -> GroupNorm(32, 256) -> ReLU (fcosv2.py:300-336), and the statistics pass of the norm would re-read the tensor the conv
 * epilogue just held in registers.  gn_sums: nlev consecutive [N][G][2] blocks, zeroed by this call, then accumulated (float atomics)
 * with the per-(image, group) sum and sum of squares of the STORED bf16 values; K must equal 8 * G.  sod_groupnorm_apply_ml turns them
 * into (mean, rstd) and applies the norm. */
int sod_conv2d_fwd_ml_gnsum(int nlev, const void* const* x, const void* w, const float* bias, void* const* y,
                            int N, const int* H, const int* W, int C, int K, int R, int S, int stride, int pad, int dil,
                            long long y_img_stride, int flags, float* gn_sums, int G, void* stream);
int sod_conv2d_dgrad_ml(int nlev, const void* const* dy, const void* wt, void* const* dx,
                        int N, const int* H, const int* W, int C, int K, int R, int S, int stride, int pad, int dil,
                        long long dy_img_stride, void* stream);
/* sod_conv2d_dgrad_ml with the ReLU backward of the tensors dx is the gradient of folded into the epilogue: dx[l] = relu_mask[l] > 0 ?
 * (data gradient) : 0, relu_mask[l] = the post-ReLU tensor (bf16, dx[l]'s shape).  Consecutive [conv3x3 -> ReLU] tower units
 * (RetinaNetHead, retina_rotated.py:418-430): the consumer's data gradient applies the producer's mask, one launch per level less. */
int sod_conv2d_dgrad_ml_mask(int nlev, const void* const* dy, const void* wt, const void* const* relu_mask, void* const* dx,
                             int N, const int* H, const int* W, int C, int K, int R, int S, int stride, int pad, int dil,
                             long long dy_img_stride, void* stream);
/* sod_conv2d_dgrad_ml (stride 1) for dY rows of Kpitch channels contracted as Kp >= Kpitch channels per tap (Kp % 64 == 0): wt_pad is
 * [C][R][S][Kp] with zero columns from Kpitch on; nothing past a pixel's Kpitch channels is ever read (zero fill).  The 720 class-score channels of RetinaNetHead.cls_score (retina_rotated.py:432-437) run as a
 * 768-wide contraction on the linear K loops (256x256 kernel) instead of the per-chunk gather path. */
int sod_conv2d_dgrad_ml_kpitch(int nlev, const void* const* dy, const void* wt_pad, void* const* dx,
                               int N, const int* H, const int* W, int C, int Kp, int Kpitch, int R, int S, int pad, int dil,
                               long long dy_img_stride, void* stream);
/* sod_conv2d_dgrad_ml + accum[l] (bf16, dx[l]'s shape) added in the epilogue: the second of two consumers of the same tensors (the two
 * towers of FCOSHead read the same FPN outputs, fcosv2.py:342-361) leaves the SUM of both data gradients - autograd's accumulation pass
 * (read 2, write 1 per level) is not launched.  relu_mask (NULL or one post-ReLU tensor per level): the ReLU backward of the tensor the sum
 * is the gradient of, applied after the addition - StandardRPNHead (d2, the reference's RRPN configs): hidden = relu(conv(x)) feeds the
 * objectness and the anchor-delta convs. */
int sod_conv2d_dgrad_ml_accum(int nlev, const void* const* dy, const void* wt, const void* const* accum, const void* const* relu_mask,
                              void* const* dx, int N, const int* H, const int* W, int C, int K, int R, int S, int stride, int pad, int dil,
                              long long dy_img_stride, void* stream);
int sod_conv2d_wgrad_ml(int nlev, const void* const* dy, const void* const* x, float* dw, const float* qscale,
                        int N, const int* H, const int* W, int C, int K, int R, int S, int stride, int pad, int dil,
                        long long dy_img_stride, int splits, int flags, void* ws, long long ws_bytes, void* stream);
/* Tile policy of sod_conv2d_fwd / _dgrad (process-wide): 1 (default, or env SOD_CONV256) = shapes with >= 1 round of 256x256 output
 * tiles, Nout % 256 == 0 and R*S*C >= 1024 run on the 256x256x64 8-phase kernel (whole rounds; a short remainder goes to the
 * 128x128 kernel); 0 = 128x128 kernel only; 2 = the 256 kernel for every shape it supports (tests); -1 = re-read the env. */
/* Data gradient of a grouped convolution in channel-window mode (SOD_CONV_CWIN): wt_win [C][R][S][128] - row c holds, per tap, the
 * weights towards the 128 output channels of c's own tile; relu_mask as for sod_conv2d_dgrad.  C == K multiples of 128. */
int sod_conv2d_dgrad_cwin(const void* dy, const void* wt_win, const void* relu_mask, void* dx,
                          int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil, void* stream);
int sod_conv_set_tile256(int mode);
/* Process-wide: 1 (default; -1 = re-read SOD_CONV_PW) sends the EXPANDING 1x1 convolutions of the bottleneck blocks and their data gradients
 * (C in {128, 256, 512} -> 4 / 8 / 16 x 128 channels, stride 1, one dense level, >= 16384 pixels, 256-CU device; forward epilogues bias /
 * shortcut / ReLU / bit mask, backward accumulate / bit mask) to the persistent weight-stationary kernel of conv_pw.hip; 0 keeps them on
 * the tiled kernels.  Bit-identical results either way (d2 BottleneckBlock conv3 / conv1 under slender_det/modeling/backbone/fpn.py:94-115). */
int sod_conv_set_pw(int on);
/* Process-wide policy of the persistent weight-stationary 3x3 kernel (conv_ws3.hip: 3x3 / stride 1 / pad 1, 128 -> 128 channels, one dense level,
 * 256-CU device; forward bias / ReLU, backward a bf16 ReLU mask tensor) behind sod_conv2d_fwd / sod_conv2d_dgrad: -1 = re-read SOD_CONV_WS3
 * (default 1), 0 = never, 1 = launches of at least 1 024 tiles of 8 x 14 pixels (conv2 of the res3 bottleneck blocks at batch 16, d2
 * BottleneckBlock under slender_det/modeling/backbone/fpn.py:94-115), 2 = every supported shape (parity tests). */
int sod_conv_set_ws3(int mode);
/* Kernel policy of sod_conv2d_wgrad / _wgrad_ml for the shapes the 256x256 kernel does not take (process-wide): -1 (default) = env
 * SOD_WGRAD_VARIANT or the library's per-shape choice; 0 = conv_wgrad_kernel (two 4-wave workgroups per CU, float atomics);
 * G*1000 + NSTAGE*100 + EPI*10 + FDB = one variant of conv_wgrad_ring_kernel for every shape (G = 1 | 2 groups of four waves per
 * workgroup that split the pixel range and combine through LDS, NSTAGE = 3 | 4 ring slots, EPI 0 = atomics / 1 = slabs + reduce,
 * FDB = fragment reads one K-step ahead): parity tests and A/B measurements.  Same contraction (reference: the weight gradients of
 * slender_det/modeling/backbone/fpn.py:94-115), another summation order. */
int sod_conv_set_wgrad_variant(int variant);
/* Per CALLING THREAD (the forward thread and autograd's worker thread bracket their own launches; a launch issued by another thread
 * in between is not affected): while on, the single-level sod_conv2d_fwd / sod_conv2d_dgrad launches of this thread walk their
 * output tiles last to first, so that a kernel reading a tensor its predecessor has just written starts with the part still in the
 * Infinity Cache.  Results are unaffected. */
int sod_conv_set_reverse(int on);
/* Which kernel the last sod_conv2d_fwd / _dgrad call of this thread dispatched to (profiling aid): 256 = the 256x256x64 kernel
 * (possibly followed by a short 128x128 tail launch), otherwise BQ*100000 + BP*100 + BK (+1 for the generic-channel path). */
int sod_conv_last_variant(void);
/* In-library timing of the conv launches of the process (forward on the caller's thread, backward on autograd's): while enabled, every sod_conv2d_fwd / _dgrad / _wgrad dispatch
 * records one hipEvent pair on ITS launch stream right around its main kernel (for a weight gradient: the kernel plus its slab
 * reduce).  sod_conv_prof_collect synchronises, writes duration (ms), kernel variant (as above; weight gradients: 256 = the
 * 256x256 kernel, 32003 / 32002 / 64002 = pixels per K-step and LDS slots of the 128x128 kernel), the main kernel's share of the
 * dispatch's output pixels (1 unless a 128x128 tail launch followed) and mode (0 fwd / 1 dgrad / 2 wgrad) of up to max dispatches
 * in call order, clears the list and returns the count (capacity 8192). */
int sod_conv_prof_enable(int on);
int sod_conv_prof_collect(float* ms, int* variant, float* frac, int* mode, int max);

/* ---------------------------------------------------------------------------------------------------------
 * GroupNorm (+ fused ReLU), NHWC bf16 — nn.GroupNorm(32, C) + nn.ReLU in FCOSHead (fcosv2.py:315-336).
 * mean_rstd: [N][G][2] fp32 (saved for backward). C/G must be a multiple of 8.
 * det_ws / det_ws_bytes (also sod_bias_grad): optional caller-owned fp32 scratch.  NULL: block partial sums meet in float atomics
 * (run-to-run differences in the last bits).  Non-NULL: deterministic mode - every block stores its partials into the scratch and a
 * second kernel adds them in a fixed order (bit-identical results; SOD_EARG if the scratch is too small:
 * 4 * blocks * N * (C/4 + 3*C) bytes with blocks <= 1024/N + levels; sod_conv2d_wgrad_workspace_bytes() always suffices).
 * --------------------------------------------------------------------------------------------------------- */
int sod_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean_rstd,
                      int N, int HW, int C, int G, long long img_stride, float eps, int relu,
                      float* det_ws, long long det_ws_bytes, void* stream);
int sod_groupnorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean_rstd,
                      void* dx, float* dgamma, float* dbeta, float* dxsum /* optional [C]: += sum over pixels of dx = bias
                      gradient of the convolution that produced x */, float* red_ws /* 2*N*G floats */,
                      int N, int HW, int C, int G, long long img_stride, int relu, float* det_ws, long long det_ws_bytes, void* stream);
/* The same GroupNorm over several FPN levels that share gamma/beta (FCOS / RepPoints towers): one launch per pass instead of one per
 * level. x/y/dy/dx: arrays of nlev device pointers (dense (N,hw[l],C) bf16); mean_rstd / red_ws: nlev consecutive [N][G][2] blocks. */
int sod_groupnorm_fwd_ml(int nlev, const void* const* x, const float* gamma, const float* beta, void* const* y, float* mean_rstd,
                         int N, const int* hw, int C, int G, float eps, int relu, float* det_ws, long long det_ws_bytes, void* stream);
/* Second half of sod_groupnorm_fwd_ml for statistics that sod_conv2d_fwd_ml_gnsum gathered: sums_to_mean_rstd holds nlev [N][G][2]
 * blocks of (sum, sum of squares) on entry and (mean, rstd) - what the backward pass takes - on return; y = [relu](norm(x)). */
int sod_groupnorm_apply_ml(int nlev, const void* const* x, const float* gamma, const float* beta, void* const* y, float* sums_to_mean_rstd,
                           int N, const int* hw, int C, int G, float eps, int relu, void* stream);
int sod_groupnorm_bwd_ml(int nlev, const void* const* dy, const void* const* x, const float* gamma, const float* beta,
                         const float* mean_rstd, void* const* dx, float* dgamma, float* dbeta, float* dxsum, float* red_ws,
                         int N, const int* hw, int C, int G, int relu, float* det_ws, long long det_ws_bytes, void* stream);

/* elementwise helpers on bf16 tensors of n elements (n % 8 == 0) */
int sod_relu_fwd(const void* x, void* y, long long n, void* stream);
int sod_relu_bwd(const void* dy, const void* y, void* dx, long long n, void* stream);
int sod_add_bf16(const void* a, const void* b, void* out, long long n, void* stream);
/* out = a + nearest-2x-upsample(b): a (N,H,W,C), b (N,H/2,W/2,C) — d2 FPN's top-down sum when FPN.NORM != "" (the norm sits
 * between the lateral conv and the sum, so it cannot ride in the conv epilogue); the configs under configs/rep-points use NORM: "GN" */
int sod_add_up2_bf16(const void* a, const void* b, void* out, int N, int H, int W, int C, void* stream);
/* dbias[c] += sum over (n, pixel) of dy — bias gradient of nn.Conv2d(bias=True) */
int sod_bias_grad(const void* dy, float* dbias, int N, int HW, int C, long long img_stride, float* det_ws, long long det_ws_bytes, void* stream);
/* dbias[c] += s * sum over (n, pixel) of dy, s = scale_num[0] / max(scale_den[0] * den_mul, den_min) read on the device (either pointer may
 * be NULL = 1): the bias gradient of a conv whose output gradient is stored un-scaled (sod_sigmoid_focal_loss_fwd_grad). */
int sod_bias_grad_scaled(const void* dy, float* dbias, const float* scale_num, const float* scale_den, float den_mul, float den_min,
                         int N, int HW, int C, long long img_stride, float* det_ws, long long det_ws_bytes, void* stream);
/* d2 BasicStem: F.max_pool2d(x, kernel_size=3, stride=2, padding=1) */
/* sod_bias_grad over several dense (N, hw[l], C) bf16 tensors that share the bias (a conv applied to all FPN levels): one launch,
 * dbias[c] += sum over levels, images and pixels (float atomics). */
int sod_bias_grad_ml(int nlev, const void* const* dy, float* dbias, int N, const int* hw, int C, void* stream);
int sod_maxpool3x3s2(const void* x, void* y, int N, int H, int W, int C, void* stream);
/* backward of F.interpolate(scale_factor=2, mode="nearest") in d2 FPN */
int sod_upsample2x_bwd(const void* g, void* dprev, int N, int Hc, int Wc, int C, void* stream);

/* fp32 master weights [K][RS][C] -> bf16 [K][RS][Cpad] (times scale[k]: folded d2 FrozenBatchNorm2d) and the
 * transposed bf16 [C][RS][K] copy for dgrad */
int sod_weight_prep(const float* w, const float* scale, void* w_krsc, void* w_crsk, int K, int RS, int C, int Cpad, void* stream);
/* sod_weight_prep for all trainable convolutions of a model in one launch: table_dev = n entries
 * {int64 tile0, src_off, krsc_off, crsk_off, scale_off (-1: none); int32 K, RS, C, Cpad} sorted by tile0 = prefix sums of the
 * per-weight tile counts ceil(K/64)*RS*ceil(C/64) (total_elems = their sum); offsets in elements into params (fp32), the two bf16
 * compute arenas and scales (fp32). */
int sod_weight_prep_batched(const float* params, const float* scales, const void* table_dev, int n, long long total_elems,
                            void* krsc_arena, void* crsk_arena, void* stream);
int sod_scale_rows(float* g, const float* scale, int K, long long row, void* stream);

/* torch.optim.SGD(momentum, nesterov, weight_decay) over the flat arena (slender_det/solver/build.py:21-25).
 * segments_dev: device array of {long long begin, end; float lr_mult, wd}. */
int sod_sgd_step(float* params, const float* grads, float* momentum_buf, const void* segments_dev, int nseg,
                 const float* lr_dev, float lr, float momentum, int nesterov, int first_step, float grad_scale, void* stream);

/* torch.optim.Adam / AdamW / Adagrad over the flat arena: SOLVER.OPTIM "ADAM" / "ADAMW" / "ADAGRAD"
 * (slender_det/solver/build.py:26-31), one launch for all parameters, per-segment lr multiplier and weight decay as sod_sgd_step.
 * mode 0 Adam (L2 decay in the gradient), 1 AdamW (decoupled decay), 2 Adagrad (exp_avg unused, exp_avg_sq = running sum of squares;
 * lr = the caller's clr = lr / (1 + (t-1) * lr_decay)).  bias_correction1 = 1 - beta1^t, bias_correction2_sqrt = sqrt(1 - beta2^t). */
int sod_adaptive_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const void* segments_dev, int nseg,
                      int mode, float lr, float beta1, float beta2, float eps, float bias_correction1, float bias_correction2_sqrt,
                      float grad_scale, void* stream);

/* FCOSV2.preprocess_image (fcosv2.py:268-275) + ImageList.from_tensors: (x-mean)/std, zero pad, CHW -> NHWC(8) bf16.
 * mean3/std3 are HOST pointers. */
int sod_preprocess_image(const void* img, int is_uint8, int C, int H, int W, void* out, int Hp, int Wp, int Cpad,
                         const float* mean3, const float* std3, void* stream);
/* The same for a whole batch in one launch: imgs / H / W are HOST arrays of n (<= 64) device pointers and sizes (all images of
 * one dtype); out is the (n, Hp, Wp, 8) batch buffer. */
int sod_preprocess_batch(int n, const void* const* imgs, int is_uint8, int C, const int* H, const int* W, void* out, int Hp, int Wp,
                         int Cpad, const float* mean3, const float* std3, void* stream);
/* preprocess_image + the frozen ResNet stem in ONE kernel: uint8 image -> (v - mean) / std -> conv 7x7 stride 2 pad 3 with the
 * folded FrozenBatchNorm2d -> ReLU -> max-pool 3x3 stride 2 pad 1 (fcosv2.py:268-275 + detectron2 BasicStem).
 * imgs / H / W: HOST arrays of n (<= 64) device pointers to (3, H, W) uint8 planes and their sizes; w_packed: [64][24][8] bf16 =
 * w_eff[k][c*7 + r][s] (zero for s = 7 and chunks 21..23); bias: [64] folded BN shift; out: (n, Hp/4, Wp/4, 64) bf16 for the
 * batch padded to (Hp, Wp) (multiples of 4; images are zero-padded bottom / right as ImageList.from_tensors does). */
int sod_stem_fused(int n, const void* const* imgs, const int* H, const int* W, const void* w_packed, const float* bias,
                   void* out, int Hp, int Wp, const float* mean3, const float* std3, void* stream);
/* A FROZEN bottleneck block of the ResNet body in ONE kernel (detectron2 BottleneckBlock with FrozenBatchNorm2d folded, as
 * build_resnet_backbone builds res2 under MODEL.BACKBONE.FREEZE_AT >= 2; reached from slender_det/modeling/backbone/fpn.py:103):
 *   out = relu(conv3(relu(conv2_3x3(relu(conv1(x))))) + shortcut(x)),  64 bottleneck channels, 256 output channels, stride 1.
 * x: (N, H, W, Cin) bf16 NHWC; w1: [64][Cin], w2: [64][3][3][64], w3: [256][64] bf16 with the BN scale folded; b1 / b2: [64],
 * b3: [256] fp32 folded BN shifts (b3 INCLUDES the projection shortcut's shift when wsc is given); wsc: [256][Cin] bf16
 * projection-shortcut weights with Cin = 64 (first block of res2), or NULL for the identity shortcut with Cin = 256 (SOD_EARG for any
 * other combination); out: (N, H, W, 256) bf16.  The two 64-channel intermediates stay in LDS (a frozen block has no backward pass). */
int sod_bottleneck_frozen_fwd(const void* x, int N, int H, int W, int Cin, const void* w1, const float* b1, const void* w2,
                              const float* b2, const void* w3, const float* b3, const void* wsc, void* out, void* stream);
/* The augmentation + batching stage in front of the model on the device: ResizeShortestEdge / ResizeLongestEdge and
 * RandomFlip of build_augmentation (slender_det/data/utils.py:29-50; detectron2 ResizeTransform = PIL BILINEAR resize of the uint8
 * image, HFlipTransform) fused with sod_preprocess_batch's normalise + zero-pad + NHWC(8) bf16, ONE launch for n (<= 64) images.
 * imgs[i]: device (H[i], W[i], 3) uint8, channel-interleaved.  xbounds[i] / ybounds[i]: device int32 [new][2] = {first source
 * index, number of taps}; xcoef[i] / ycoef[i]: device int32 [new][kx[i]] / [new][ky[i]] = Pillow's fixed-point (22-bit) filter
 * coefficients (slenderobjdet_amd/data/transforms.py:pil_bilinear_coeffs restates Pillow's precompute_coeffs).  flip[i] != 0 mirrors
 * the resized image horizontally.  All array arguments are HOST arrays of n entries.  Resized pixels equal PIL's bit for bit. */
int sod_resize_flip_preprocess_batch(int n, const void* const* imgs, const int* H, const int* W, const int* newH, const int* newW,
                                     const int* const* xbounds, const int* const* xcoef, const int* kx,
                                     const int* const* ybounds, const int* const* ycoef, const int* ky, const int* flip,
                                     void* out, int Hp, int Wp, int Cpad, const float* mean3, const float* std3, void* stream);
int sod_nchw_f32_to_nhwc_bf16(const float* x, void* y, int N, int C, int HW, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Losses and targets (fp32)
 * --------------------------------------------------------------------------------------------------------- */
/* fvcore.nn.sigmoid_focal_loss_jit(inputs, targets, alpha, gamma, reduction) — call site fcosv2.py:124.
 * Targets are either class indices `labels[M]` (value k in [0,K) marks column k; anything else = background) or a
 * dense one-hot/soft `dense_targets[M,K]`. logits rows have pitch `ld`. elem_out (optional) [M,K] = reduction "none". */
int sod_sigmoid_focal_loss_fwd(const float* logits, const int* labels, const float* dense_targets,
                               long long M, int K, int ld, float alpha, float gamma, float* elem_out,
                               float* sum_out, float* ws, void* stream);
/* dlogits = dloss/dlogits * scale_num[0] / max(scale_den[0]*den_mul, den_min); rows of pitch ld_out, columns >= K zeroed */
/* Forward sum and the UN-scaled gradient d(sum)/d(logits) (bf16 rows of ld_out >= K elements, padding columns zero) in one pass over the
 * logits; class-index labels only (label < 0 = ignored row), K, ld, ld_out multiples of 4.  The training step's backward differs from this
 * gradient by one scalar (upstream gradient / loss normaliser): the consumers of the gradient apply it - the data gradient through scaled
 * weights, the weight gradient through its per-output-channel factor, the bias gradient through sod_bias_grad_scaled. */
int sod_sigmoid_focal_loss_fwd_grad(const float* logits, const int* labels, long long M, int K, int ld, float alpha, float gamma,
                                    float* sum_out, float* ws, void* dlogits_bf16, int ld_out, void* stream);
int sod_sigmoid_focal_loss_bwd(const float* logits, const int* labels, const float* dense_targets,
                               long long M, int K, int ld, float alpha, float gamma,
                               const float* scale_num, const float* scale_den, float den_mul, float den_min,
                               void* dlogits, int ld_out, int out_bf16, void* stream);
/* slender_det.layers.iou_loss(pred, target, weight, loss_type) — layers/iou_loss.py:4-37. pred/target [P,4] LTRB.
 * mask (optional): rows with mask<0 or mask==mask_bg contribute 0. */
int sod_iou_loss_fwd(const float* pred, const float* target, const float* weight, const int* mask, int mask_bg,
                     long long P, int loss_type, float* elem_out, float* sum_out, float* ws, void* stream);
int sod_iou_loss_bwd(const float* pred, const float* target, const float* weight, const int* mask, int mask_bg,
                     long long P, int loss_type, const float* grad_scale, float* dpred, void* stream);
/* compute_targets_for_locations + get_sample_region + compute_centerness_targets
 * (slender_det/modeling/meta_arch/fcos/utils.py:108-212, 295-300) for a whole batch in one launch.
 * boxes [sumG,4] XYXY, classes [sumG], box_offsets [N+1] (device); level tables are HOST arrays.
 * Outputs: labels [N,L] (background = num_classes), reg_targets [N,L,4], ctr_targets [N,L] (0 on background),
 * stats[2] = {number of positives, sum of centerness targets} over the batch. */
int sod_fcos_assign(const float* boxes, const int* classes, const int* box_offsets, int N,
                    int nlevels, const int* lvl_h, const int* lvl_w, const int* lvl_stride,
                    const float* lvl_lo, const float* lvl_hi, float radius, int num_classes,
                    int* labels, float* reg_targets, float* ctr_targets, float* stats, float* ws, void* stream);
/* FCOSV2.losses, regression + centerness part (fcosv2.py:128-145) fused with Scale/exp of FCOSHead.forward
 * (fcosv2.py:372-378): sums[0] = sum_pos iou_loss(pred, tgt)*ctr, sums[1] = sum_pos BCEWithLogits(ctr_logit, ctr). */
int sod_fcos_regctr_loss_fwd(const float* box_raw, int ld_box, const float* ctr_logit, int ld_ctr,
                             const int* labels, const float* reg_targets, const float* ctr_targets,
                             const float* scales, int N, int nlevels, const int* lvl_h, const int* lvl_w,
                             const int* lvl_stride, int num_classes, int loss_type, int norm_reg_targets,
                             float* sums, float* ws, void* stream);
int sod_fcos_regctr_loss_bwd(const float* box_raw, int ld_box, const float* ctr_logit, int ld_ctr,
                             const int* labels, const float* reg_targets, const float* ctr_targets,
                             const float* scales, int N, int nlevels, const int* lvl_h, const int* lvl_w,
                             const int* lvl_stride, int num_classes, int loss_type, int norm_reg_targets,
                             const float* grad_reg, const float* grad_ctr, const float* norm, float inv_world,
                             void* dbox, int ld_out, int ctr_col, void* dctr, int ld_dctr, int dctr_col,
                             float* dscales, float* ws, void* stream);
/* out3 = {cls_loss, reg_loss, centerness_loss} with the all-reduced normalisers (fcosv2.py:115-145) */
int sod_fcos_finalize_losses(const float* focal_sum, const float* regctr_sums, const float* stats,
                             float inv_world, float* out3, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Detection operators (detectron2 / torchvision / fvcore)
 * --------------------------------------------------------------------------------------------------------- */
/* torchvision.ops.nms behind detectron2.layers.batched_nms (fcosv2.py:241): `order` = indices sorted by descending score
 * (stable), greedy suppression of IoU > iou_threshold, IoU = inter/(a+b-inter). keep[0..*num_keep) = kept ORIGINAL indices
 * in score order. mask_ws: sod_nms_workspace_bytes(n) bytes. n <= 65536. */
long long sod_nms_workspace_bytes(int n);
int sod_nms(const float* boxes, const long long* order, int n, float iou_threshold, long long* keep, int* num_keep,
            void* mask_ws, void* stream);
/* Dense-detector post-processing on the device for the whole batch.
 * sod_fcos_decode = the per-level part of FCOS(V2).inference_single_image (fcosv2.py:194-238, fcos.py:385-436), one launch:
 *   keep = sigmoid(cls) > pre_nms_thresh;  score = sigmoid(cls) * sigmoid(centerness);  per (image, level) the pre_nms_top_n best
 *   scores (all of them when fewer; ties at the cut resolved towards the lower (location, class) index);  boxes = location -+ the
 *   regression decoded as in FCOSHead (exp(scale * x), or relu(scale * x) * stride with NORM_REG_TARGETS);  score = sqrt(score).
 * cls_logits (N, L, ld_cls), box_raw (N, L, ld_box) fp32 = the head's prediction buffers (L = sum H[l]*W[l], level-major); the
 * centerness logit is column ctr_col_box of box_raw (CENTERNESS_ON_REG) or column ctr_col_cls of cls_logits - exactly one >= 0.
 * Outputs have pre_nms_top_n slots per (image, level), filled in torch.nonzero() order; unused slots carry score -inf / class -1;
 * out_counts (N, nlev) = candidates per level.  No host synchronisation. */
int sod_fcos_decode(const float* cls_logits, int ld_cls, const float* box_raw, int ld_box, const float* scales,
                    int N, int nlev, const int* H, const int* W, const int* strides, int num_classes,
                    int ctr_col_box, int ctr_col_cls, int norm_reg_targets, float pre_nms_thresh, int pre_nms_top_n,
                    float* out_boxes, float* out_scores, int* out_classes, int* out_counts, void* stream);
/* The same selection for heads without a centerness branch: rows of K class logits (RetinaNet: anchors, retina_rotated.py:296-340;
 * RepPoints: points, rpd.py:717-765), logits (N, R, ld) with R = sum rows[l] (level-major).  by_row_max = 0: candidates are the
 * (row, class) pairs with sigmoid(logit) > score_thresh, the top_n best per (image, level) are kept (= sort, take top_n, threshold);
 * by_row_max = 1: candidates are rows scored by their best class (scores, classes = logits.sigmoid().max(1)).  out_rows = the row
 * index inside its level of every kept slot (the caller gathers / decodes the boxes); scores without sqrt; rest as sod_fcos_decode. */
int sod_dense_topk_select(const float* logits, int ld, int N, int nlev, const int* rows, int num_classes, int by_row_max,
                          float score_thresh, int top_n, int* out_rows, float* out_scores, int* out_classes, int* out_counts, void* stream);
/* detectron2.layers.batched_nms / batched_nms_rotated (fcosv2.py:241, proposal_utils.py:104) + keep[: max_keep] for B images of M
 * candidate slots each (score -inf = empty slot), box_dim 4 (XYXY) or 5 (cx, cy, w, h, angle_deg):
 * prepare() writes the class-shifted boxes and the per-image candidate count into ws; the caller then sorts the scores of every
 * image in stable descending order (order: (B, M) int64 local indices, empty slots last) and run() performs the greedy suppression
 * (IoU > iou_threshold) per image, stopping after max_keep survivors: keep (B, max_keep) local indices in score order, num_keep (B).
 * The candidate counts never leave the device. */
long long sod_batched_nms_workspace_bytes(int B, int M, int box_dim);
int sod_batched_nms_prepare(const float* boxes, const float* scores, const int* classes, int B, int M, int box_dim, void* ws, void* stream);
int sod_batched_nms_run(const long long* order, int B, int M, int box_dim, float iou_threshold, int max_keep, long long* keep, int* num_keep,
                        void* ws, void* stream);
/* The per-image part of find_top_rpn_proposals (slender_det/modeling/proposal_generator/proposal_utils.py:45-120, = detectron2's)
 * for the whole batch, in place: entries with a non-finite box or score are counted in *bad_count (zero it first) and emptied,
 * boxes are clipped to their image (image_hw: device (B, 2) floats; rotated boxes as RotatedBoxes.clip: angles normalised, only
 * |angle| <= 1 degree clipped), boxes with a side <= min_size are emptied (score = -inf). */
int sod_rpn_clip_filter(float* boxes, float* scores, const float* image_hw, int B, int M, int box_dim, float min_size, int* bad_count,
                        void* stream);
/* detectron2 nms_rotated / box_iou_rotated (csrc/nms_rotated, csrc/box_iou_rotated; reached through RRPN / RROIHeads selected by
 * configs/rotated/Base-RRCNN-FPN.yaml:10-36 and pairwise_iou at retina_rotated.py:276): boxes (n,5) = (cx,cy,w,h,angle_deg). */
int sod_nms_rotated(const float* boxes, const long long* order, int n, float iou_threshold, long long* keep, int* num_keep,
                    void* mask_ws, void* stream);
int sod_box_iou_rotated(const float* boxes1, int n1, const float* boxes2, int n2, float* iou_out, void* stream);
/* detectron2 ROIAlign(output_size, spatial_scale, sampling_ratio, aligned=True) / ROIAlignRotated (roi_heads/roi_heads.py:48-53).
 * x: NHWC bf16; rois (R,5) [batch,x1,y1,x2,y2] or, rotated, (R,6) [batch,cx,cy,w,h,angle_deg]; out (R,PH,PW,C) fp32.
 * bwd accumulates atomically into dx (N,H,W,C) fp32 (zero it first). */
int sod_roi_align_fwd(const void* x, const float* rois, float* out, int R, int N, int H, int W, int C, int PH, int PW,
                      float spatial_scale, int sampling_ratio, int rotated, void* stream);
/* the same with fp32 features (fp32 validation mode of the two-stage path; the backward already works on fp32 rows) */
int sod_roi_align_fwd_f32(const float* x, const float* rois, float* out, int R, int N, int H, int W, int C, int PH, int PW,
                          float spatial_scale, int sampling_ratio, int rotated, void* stream);
int sod_roi_align_bwd(const float* dout, const float* rois, float* dx, int R, int N, int H, int W, int C, int PH, int PW,
                      float spatial_scale, int sampling_ratio, int rotated, void* stream);
/* fvcore.nn.giou_loss(boxes1, boxes2, eps) on XYXY (retina_rotated.py:240): per-row loss, optional sum, optional gradient
 * w.r.t. boxes1 scaled by grad_scale[0]. */
int sod_giou_loss_xyxy(const float* boxes1, const float* boxes2, long long P, float eps, float* elem_out, float* sum_out,
                       const float* grad_scale, float* dboxes1, float* ws, void* stream);
/* fvcore.nn.smooth_l1_loss(input, target, beta) (retina_rotated.py:229, rpd.py:389-395) */
int sod_smooth_l1_loss(const float* input, const float* target, long long n, float beta, float* elem_out, float* sum_out,
                       const float* grad_scale, float* dinput, float* ws, void* stream);
/* pairwise_iou + Matcher([lo,hi],[l0,l1,l2],allow_low_quality_matches) of RetinaNet.label_anchors (retina_rotated.py:251-295)
 * without materialising the G x A matrix. gt_best_ws: G unsigned words. G <= 4096. */
int sod_anchor_match(const float* gt_boxes, int G, const float* anchors, int A, float thr_lo, float thr_hi,
                     int label_below, int label_between, int label_above, int allow_low_quality,
                     float* matched_vals, int* matches, signed char* labels, unsigned* gt_best_ws, void* stream);

/* DeformConv / ModulatedDeformConv FORWARD as one implicit-GEMM kernel: the bilinear gather feeds the MFMA loop through LDS, no
 * (N*Ho*Wo, KH*KW*C) column buffer in HBM (detectron2.layers.DeformConv.forward at df_conv.py:67-78, rpd.py:637-642).
 * x (N,H,W,C) bf16, offset / mask fp32 rows as for sod_deform_im2col, w [K][KH*KW][C] bf16 (the GEMM view of the KRSC weights),
 * bias [K] or NULL, y (N,Ho,Wo,K) bf16 = act(conv).  Needs C % 64 == 0, (C / deformable_groups) % 64 == 0, K % 8 == 0.
 * The sampled values are computed exactly as sod_deform_im2col computes them (fp32, one rounding to bf16). */
int sod_deform_conv_fwd_fused(const void* x, const float* offset, const float* mask, const void* w, const float* bias, void* y,
                              int N, int H, int W, int C, int K, int KH, int KW, int stride, int pad, int dil, int deformable_groups,
                              int off_ld, int mask_ld, int mask_is_logit, int relu, void* stream);
/* fp32 reference-precision forward of the same op (TEST MODE: one thread per output, no matrix cores): x (N,H,W,C), w [K][KH*KW][C],
 * y (N,Ho,Wo,K) all fp32.  Lets the sampling / contraction semantics be checked on the device at fp32 tolerance (the reference's
 * known-answer vectors, tests/test_deformable_conv.py:67-87) independently of the bf16 operand rounding of the fast paths. */
int sod_deform_conv_fwd_f32(const float* x, const float* offset, const float* mask, const float* w, const float* bias, float* y,
                            int N, int H, int W, int C, int K, int KH, int KW, int stride, int pad, int dil, int deformable_groups,
                            int off_ld, int mask_ld, int mask_is_logit, void* stream);
/* ... and its WEIGHT GRADIENT the same way: dw[k][tap][c] (fp32, [K][KH*KW][C], accumulated) += sum over pixels of dy * sample, the
 * sampled rows gathered into LDS tiles inside the kernel (detectron2 deform_conv_backward_filter).  Pixel splits meet in fp32 slabs
 * in ws (sod_conv2d_wgrad_workspace_bytes() suffices) summed in a fixed order: deterministic.  dy (N,Ho,Wo,K) bf16.  Needs
 * C % 128 == 0 (or C == 64) and, with deformable_groups > 1, (C / deformable_groups) % 128 == 0.  qscale (optional, [K]): each row k
 * of the sum is multiplied by qscale[k] before it is added (detectron2 DeformBottleneckBlock: FrozenBN folded into conv2's weights). */
int sod_deform_conv_wgrad_fused(const void* dy, const void* x, const float* offset, const float* mask, float* dw, const float* qscale,
                                int N, int H, int W, int C, int K, int KH, int KW, int stride, int pad, int dil, int deformable_groups,
                                int off_ld, int mask_ld, int mask_is_logit, void* ws, long long ws_bytes, void* stream);
/* ---------------------------------------------------------------------------------------------------------
 * Deformable convolution v1/v2 — detectron2.layers.DeformConv / ModulatedDeformConv behind DFConv2d
 * (slender_det/layers/df_conv.py:6-78; rpd.py:147-154,637-642). NHWC; offset fp32 rows of pitch off_ld with channel
 * 2k = dy, 2k+1 = dx of tap k = (g*KH+i)*KW+j; mask fp32 rows of pitch mask_ld (mask_is_logit: sigmoid applied inside,
 * as DFConv2d does at df_conv.py:76).  cols = (N,Ho,Wo,KH*KW*C) bf16, channel = tap*C + c; the product with the weights is
 * sod_conv2d_fwd(cols, w viewed as [K][1][1][KH*KW*C]) and its gradients are the 1x1 dgrad/wgrad on the same view.
 * col2im: dx_f32 (N,H,W,C) fp32 atomically accumulated (zero it), doffset/dmask pitched like offset/mask (zero them).
 * --------------------------------------------------------------------------------------------------------- */
int sod_deform_im2col(const void* x, const float* offset, const float* mask, void* cols,
                      int N, int H, int W, int C, int KH, int KW, int stride, int pad, int dil, int deformable_groups,
                      int off_ld, int mask_ld, int mask_is_logit, void* stream);
/* Backward of DeformConv / ModulatedDeformConv w.r.t. input, offsets and mask in ONE pass, without the column-gradient tensor
 * (detectron2 deform_conv_backward_input + deform_conv_backward_parameters; call sites rpd.py:637-642, df_conv.py:67-78): the workgroup
 * of an 8x8 output-pixel tile x 32 input channels computes its slice of dcols = dY x W^T tap by tap on the matrix cores (dY rows as
 * register fragments, W^T rows staged through LDS) and scatters it straight into the LDS fixed-point window / the offset and mask
 * gradients.  dy (N,Ho,Wo,K) bf16; wt = the CRSK copy of the weights viewed as [(KH*KW*C)][K] bf16 (row = tap*C + c); dx_f32, doffset,
 * dmask as for sod_deform_col2im (zero them); wnorm_ws: KH*KW*C floats of scratch (column norms of W for the fixed-point scale).
 * K in {128, 256, 512}, C % 32 == 0 and (C / deformable_groups) % 32 == 0; other shapes: sod_conv2d_dgrad + sod_deform_col2im. */
int sod_deform_conv_bwd_fused(const void* dy, const void* wt, const void* x, const float* offset, const float* mask, float* dx_f32,
                              float* doffset, float* dmask, float* wnorm_ws, int N, int H, int W, int C, int K, int KH, int KW, int stride,
                              int pad, int dil, int deformable_groups, int off_ld, int mask_ld, int mask_is_logit, void* stream);
/* Diagnostic counter of the tiled DeformConv backward kernels (sod_deform_conv_bwd_fused, sod_deform_col2im): they accumulate dX in an
 * LDS window around each 8x8 output tile; a sample whose bilinear footprint lies outside the window (offsets larger than the window's
 * slack, SOD_DCN_FUSED_R = 2 px beyond the receptive field) takes 32 global float atomics instead - a data-dependent performance cliff
 * (RepPoints' learned offsets, rpd.py:637-647).  While a DEVICE pointer is registered, every such sample lane (one pixel x tap x 8
 * channels with a non-zero gradient) adds 1 to *device_counter; NULL (default) switches the counting off.  Process-wide. */
int sod_deform_conv_set_window_counter(unsigned long long* device_counter);
/* Slack (pixels beyond the tile's receptive field, 0 .. 16) of the LDS window of the sod_deform_conv_bwd_fused launches that follow;
 * -1 = the default (env SOD_DCN_FUSED_R or 2).  A wider window keeps larger offsets off the global-atomic path and admits fewer
 * workgroups per CU; sod_deform_conv_bwd_fused_supported answers for the current setting.  layers/deform_conv.py raises it per layer
 * and level when the counter above shows more than a few per cent of the samples outside.  Process-wide; results do not depend on it. */
int sod_deform_conv_set_window_slack(int pixels);
/* 1 if sod_deform_conv_bwd_fused takes this layer, 0 if it must go through sod_conv2d_dgrad + sod_deform_col2im: the shape rule above AND
 * the workgroup's LDS window (8x8 output tile + receptive field + slack) within 96 KB - a 3x3, K = 512 layer with stride 2 (res5's first
 * block under STRIDE_IN_1X1 = False + DEFORM_ON_PER_STAGE) or stride 2 with dilation 2 does not fit.  Host-only, no GPU call. */
int sod_deform_conv_bwd_fused_supported(int C, int K, int KH, int KW, int stride, int dil, int deformable_groups);
int sod_deform_col2im(const void* dcols, const void* x, const float* offset, const float* mask,
                      float* dx_f32, float* doffset, float* dmask,
                      int N, int H, int W, int C, int KH, int KW, int stride, int pad, int dil, int deformable_groups,
                      int off_ld, int mask_ld, int mask_is_logit, void* stream);

/* fp32 validation mode of the two (SOD_PRECISION=fp32): fp32 activations, columns and column gradients; the GEMMs in between run on
   sod_conv2d_*_f32 over the column tensor.  Same sampling rule (detectron2 deform_conv; df_conv.py:67-78). */
int sod_deform_im2col_f32(const float* x, const float* offset, const float* mask, float* cols,
                          int N, int H, int W, int C, int KH, int KW, int stride, int pad, int dil, int deformable_groups,
                          int off_ld, int mask_ld, int mask_is_logit, void* stream);
int sod_deform_col2im_f32(const float* dcols, const float* x, const float* offset, const float* mask,
                          float* dx_f32, float* doffset, float* dmask,
                          int N, int H, int W, int C, int KH, int KW, int stride, int pad, int dil, int deformable_groups,
                          int off_ld, int mask_ld, int mask_is_logit, void* stream);
int sod_f32_to_bf16(const float* x, void* y, long long n, void* stream);

/* RetinaNet (detectron2 RetinaNet; in-tree mirror slender_det/modeling/meta_arch/retina/retina_rotated.py):
 * sod_retina_targets = tail of label_anchors (:279-291) + Box2BoxTransform.get_deltas (:203-206) for ONE image (weights4 host);
 * sod_retina_box_loss_* = smooth_l1_loss(pred_deltas[pos], gt_deltas[pos], beta, "sum") (:229) on the pitched prediction buffer
 * (anchor a of pixel p at p*pitch + a*4) + the EMA loss normaliser (:210-214) kept on the device:
 * normalizer <- momentum*normalizer + (1-momentum)*max(num_pos,1); sums2 = {loss sum, num_pos}. */
int sod_retina_targets(const float* anchors, int A, const float* gt_boxes, const int* gt_classes, int G, const int* matches,
                       const signed char* match_labels, int num_classes, const float* weights4, int* gt_labels, float* gt_deltas,
                       void* stream);
int sod_retina_box_loss_fwd(const float* pred, int pitch, const int* gt_labels, const float* gt_deltas, int N, int R, int A,
                            int num_classes, float beta, float* sums2, float* normalizer, float momentum, float* ws, void* stream);
int sod_retina_box_loss_bwd(const float* pred, int pitch, const int* gt_labels, const float* gt_deltas, int N, int R, int A,
                            int num_classes, float beta, const float* grad_num, const float* grad_den, void* dpred_bf16, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Two-stage path — detectron2 GeneralizedRCNN + RPN/RRPN + StandardROIHeads/RROIHeads as configured by the reference's
 * configs/rotated/Base-RRCNN-FPN.yaml (BASELINE config 5) and subclassed by proposal_generator/rpn.py:26-356,
 * meta_arch/rcnn/pvrcnn.py:67-97, roi_heads/roi_heads.py:48-53.  box_dim 4 = XYXY, 5 = (cx, cy, w, h, angle_deg).
 * sod_anchor_match_rotated: sod_anchor_match with pairwise_iou_rotated (RRPN.label_and_sample_anchors, RROIHeads sampling).
 * sod_box2box_get_deltas / apply_deltas: Box2BoxTransform / Box2BoxTransformRotated (weights host array of box_dim floats);
 *   apply: deltas rows of pitch ld hold k class-specific box_dim-vectors, boxes (n, box_dim), out (n, k*box_dim).
 * sod_bce_logits_loss_*: F.binary_cross_entropy_with_logits(x[l>=0], l[l>=0], "sum"), labels int8 in {-1,0,1} (loss_rpn_cls).
 * sod_rpn_loc_loss_*: smooth_l1_loss(pred[l==1], target[l==1], beta, "sum") (loss_rpn_loc).
 * sod_softmax_ce_*: F.cross_entropy(scores, labels, "sum") over rows of pitch ld (labels < 0 ignored); bwd fills the whole pitch.
 * sod_fastrcnn_box_loss_*: FastRCNNOutputs.smooth_l1_loss — foreground rows (0 <= class < K) use the box_dim columns of their class.
 * All bwd entry points scale by grad_scale[0] (device) * scale_mul (host: 1/normaliser).
 * --------------------------------------------------------------------------------------------------------- */
int sod_anchor_match_rotated(const float* gt_boxes, int G, const float* anchors, int A, float thr_lo, float thr_hi,
                             int label_below, int label_between, int label_above, int allow_low_quality,
                             float* matched_vals, int* matches, signed char* labels, unsigned* gt_best_ws, void* stream);
/* ROIHeads.label_and_sample_proposals' matching + labelling (detectron2 roi_heads.py; reference subclass roi_heads/roi_heads.py:48-53) for
 * the whole batch in ONE launch: boxes (N, R, box_dim) padded proposal rows (counts[n] valid), the images' ground truth concatenated
 * (gt_off: N + 1 offsets), Matcher([iou_threshold], [label_below, label_above], allow_low_quality_matches=False).  Out: matches (N, R) =
 * index of the best box inside its image (first maximum wins), classes (N, R) int8 = its class / num_classes (label 0) / -1 (label -1 and
 * padding rows).  num_classes <= 126. */
int sod_roi_label_batched(const float* boxes, const int* counts, int N, int R, int box_dim, const float* gt_boxes, const int* gt_classes,
                          const int* gt_off, float iou_threshold, int label_below, int label_above, int num_classes, int* matches,
                          signed char* classes, void* stream);
int sod_box2box_get_deltas(const float* src, const float* tgt, long long n, int box_dim, const float* weights, float* deltas, void* stream);
int sod_box2box_apply_deltas(const float* deltas, const float* boxes, long long n, int k, int box_dim, int ld, const float* weights,
                             float scale_clamp, float* out, void* stream);
int sod_bce_logits_loss_fwd(const float* logits, const signed char* labels, long long n, float* sum_out, float* ws, void* stream);
int sod_bce_logits_loss_bwd(const float* logits, const signed char* labels, long long n, const float* grad_scale, float scale_mul,
                            float* dlogits, void* stream);
/* F.binary_cross_entropy_with_logits(x[fg], t[fg], "sum") with float targets, fg = rows with 0 <= label != bg_label: the centerness
 * loss of the LRTB head (meta/heads/lrtb_head.py:236-238; FCOSV2 has it fused in sod_fcos_regctr_loss_*) */
int sod_bce_logits_soft_fwd(const float* logits, const float* targets, const int* labels, int bg_label, long long n, float* sum_out,
                            float* ws, void* stream);
int sod_bce_logits_soft_bwd(const float* logits, const float* targets, const int* labels, int bg_label, long long n,
                            const float* grad_scale, float scale_mul, float* dlogits, void* stream);
int sod_rpn_loc_loss_fwd(const float* pred, const float* target, const signed char* labels, long long n, int box_dim, float beta,
                         float* sum_out, float* ws, void* stream);
int sod_rpn_loc_loss_bwd(const float* pred, const float* target, const signed char* labels, long long n, int box_dim, float beta,
                         const float* grad_scale, float scale_mul, float* dpred, void* stream);
int sod_softmax_ce_fwd(const float* scores, const int* labels, int R, int C, int ld, float* sum_out, float* ws, void* stream);
int sod_softmax_ce_bwd(const float* scores, const int* labels, int R, int C, int ld, const float* grad_scale, float scale_mul,
                       float* dscores, void* stream);
int sod_fastrcnn_box_loss_fwd(const float* pred, const int* gt_classes, const float* gt_deltas, int R, int K, int box_dim, int ld,
                              float beta, float* sum_out, float* ws, void* stream);
int sod_fastrcnn_box_loss_bwd(const float* pred, const int* gt_classes, const float* gt_deltas, int R, int K, int box_dim, int ld,
                              float beta, const float* grad_scale, float scale_mul, float* dpred, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * RepPointsDetector (slender_det/modeling/meta_arch/reppoints/rpd.py:45-798), X = sum of level pixels, point rows pitched by ld.
 * sod_reppoints_dcn_offset: out[r,2k] = scale*pts[r,2k+1] - base_y[k], out[r,2k+1] = scale*pts[r,2k] - base_x[k] (xy->yx flip and
 *   dcn_base_offset of rpd.py:105-110,621-635; subtract_base=0, scale=gradient_mul gives the backward of the same expression);
 *   flip_xy=0 keeps the channel order (PointSetHead "Supervised Offset", meta/heads/pointset_head.py:137-140, subtracts the base
 *   from the un-flipped points).
 * sod_points2bbox_*: "minmax" transform of rpd.py:221-249 for ONE level: point k = (pts[2k] (+add[2k]))*point_stride + w*grid_stride, ...;
 *   boxes/argidx are level slices of (N,X,4)/(N,X) buffers (img strides in elements); bwd scatters to the arg points and writes
 *   whole rows (fp32 and/or bf16).
 * sod_reppoints_point_match: init-box labels for the batch, mode 0 rep_points_match (matchers/rep_matcher.py:9-101, pos_num=1),
 *   1 nearest_point_match (:199-223), 2 inside_match (:226-248) with structures/points.py:6-45; gt boxes concatenated with
 *   box_offsets (N+1); objectness (N,X) int {0,1}, box_labels (N,X,4).  max_gt >= every image's gt count, <= 4096.
 * sod_reppoints_labels: rpd.py:303-323 after pairwise_iou+Matcher (sod_anchor_match per image, init boxes as anchors):
 *   cls = gt_classes[match]; matcher label 0 -> num_classes; centre outside the image -> -1 (and objectness 0); refine = gt_boxes[match].
 * sod_reppoints_box_loss_*: smooth_l1(pred/(4*stride), target/(4*stride), beta, "sum") over selected rows (rpd.py:383-396);
 *   row selected when bg_label<0 ? label>0 : (0<=label!=bg_label); sums2 = {loss sum, #rows};
 *   bwd: dpred = mul*grad_num[0]/max(grad_den[0],den_min) * dloss/dpred (zeros on unselected rows).
 * sod_reppoints_finalize: normalizer <- m*normalizer + (1-m)*refine_sums2[1]/num_images (rpd.py:366-376);
 *   out3 = {focal/max(1,normalizer), init_weight*init_sum/max(1,init_rows), refine_sum/max(1,normalizer)}.
 * --------------------------------------------------------------------------------------------------------- */
int sod_reppoints_dcn_offset(const float* pts, float* out, long long rows, int ld, int num_points, float scale, int subtract_base,
                             int flip_xy, void* stream);
int sod_points2bbox_fwd(const float* pts, const float* add, int ld, int N, int H, int W, float grid_stride, float point_stride,
                        int num_points, float* boxes, long long box_img_stride, unsigned* argidx, long long arg_img_stride, void* stream);
int sod_points2bbox_bwd(const float* dboxes, long long box_img_stride, const unsigned* argidx, long long arg_img_stride, int ld,
                        int N, int H, int W, float point_stride, int num_points, float* dpts_f32, void* dpts_bf16, void* stream);
/* TRANSFORM_METHOD "moment" (meta/heads/pointset_head.py:328-343): box = mean -+ std * exp(moment_transfer[0|1]) over the num_points
 * points (torch.std: unbiased); "partial_minmax" (:322-327) is sod_points2bbox_* with num_points = 4 and the full row pitch ld.
 * bwd recomputes the moments from pts (+ add), writes d(pts) rows of ld floats and ACCUMULATES moment_mul * d(moment_transfer) into
 * dmoment2 (the reference scales that gradient with grad_mul, :333-334). */
int sod_points2bbox_moment_fwd(const float* pts, const float* add, int ld, int N, int H, int W, float grid_stride, float point_stride,
                               int num_points, const float* moment_transfer, float* boxes, long long box_img_stride, void* stream);
int sod_points2bbox_moment_bwd(const float* dboxes, long long box_img_stride, const float* pts, const float* add, int ld, int N, int H,
                               int W, float grid_stride, float point_stride, int num_points, const float* moment_transfer,
                               float moment_mul, float* dpts_f32, void* dpts_bf16, float* dmoment2, void* stream);
int sod_reppoints_point_match(const float* centers, const float* strides, int X, const int* lvl_start, int num_levels,
                              const float* gt_boxes, const int* box_offsets, int N, int max_gt, int mode, float scale,
                              int* objectness, float* box_labels, void* stream);
int sod_reppoints_labels(const int* matches, const signed char* match_labels, const float* gt_boxes, const int* gt_classes,
                         const int* box_offsets, const float* centers, const float* image_hw, int N, int X, int num_classes,
                         int* cls_labels, float* refine_boxes, int* objectness, void* stream);
int sod_reppoints_box_loss_fwd(const float* pred, const float* target, const int* labels, const float* strides, int N, int X,
                               int bg_label, float beta, float* sums2, float* ws, void* stream);
int sod_reppoints_box_loss_bwd(const float* pred, const float* target, const int* labels, const float* strides, int N, int X,
                               int bg_label, float beta, const float* grad_num, const float* grad_den, float den_min, float mul,
                               float* dpred, void* stream);
int sod_reppoints_finalize(const float* focal_sum, const float* init_sums2, const float* refine_sums2, float* normalizer,
                           float momentum, int num_images, float init_weight, float* out3, void* stream);

/* BBOX_REG_LOSS_TYPE "giou" of RetinaNet (retina_rotated.py:236-245) and AnchorHead (meta/heads/anchor_head.py:366-374): positives decode
 * their deltas against the anchor (Box2BoxTransform.apply_deltas) and take fvcore giou_loss (eps 1e-7) against the matched gt box
 * (matched_boxes (N,R,4)); same sums2 / EMA-normaliser / pitched bf16 gradient contract as sod_retina_box_loss_*. */
int sod_retina_giou_loss_fwd(const float* pred, int pitch, const int* gt_labels, const float* anchors, const float* matched_boxes,
                             int N, int R, int A, int num_classes, const float* weights4, float scale_clamp, float* sums2,
                             float* normalizer, float momentum, float* ws, void* stream);
int sod_retina_giou_loss_bwd(const float* pred, int pitch, const int* gt_labels, const float* anchors, const float* matched_boxes,
                             int N, int R, int A, int num_classes, const float* weights4, float scale_clamp,
                             const float* grad_num, const float* grad_den, void* dpred_bf16, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * The reference's own native operators (slender_det._C; bindings slender_det/layers/csrc/vision.cpp:64-80), fp32 NCHW.
 * BorderAlign: feature (B,4C,H,W), boxes (B,K,4) XYXY in feature coordinates -> out (B,C,K,4)
 *   (BorderAlign_cuda.cu:94-146; wh is derived from boxes as layers/border_align.py:38 does). bwd accumulates into dfeature (zero it).
 * CornerPool: directional running max, mode 0 bottom / 1 top / 2 right / 3 left (layers/corner_pool.py:90-116);
 *   bwd scatters dy to the arg-max (tie_latest=1: torch.cummax indices, the path torch>=1.5 takes; 0: the strict '>' of
 *   corner_pool.cpp:30-69). dx must be zeroed by the caller.
 * --------------------------------------------------------------------------------------------------------- */
int sod_border_align_fwd(const float* feature, const float* boxes, float* out, int B, int C, int K, int H, int W, int pool_size,
                         void* stream);
int sod_border_align_bwd(const float* dout, const float* feature, const float* boxes, float* dfeature, int B, int C, int K, int H,
                         int W, int pool_size, void* stream);
int sod_corner_pool_fwd(const float* x, float* y, long long planes, int H, int W, int mode, void* stream);
int sod_corner_pool_bwd(const float* x, const float* dy, float* dx, long long planes, int H, int W, int mode, int tie_latest,
                        void* stream);

/* detectron2.modeling.sampling.subsample_labels for a whole batch on the device (RPN.label_and_sample_anchors,
 * slender_det/modeling/proposal_generator/rpn.py:137-191; ROIHeads.label_and_sample_proposals behind roi_heads/roi_heads.py:30-66): per
 * image, up to int(num_samples * positive_fraction) positives (label != -1 and != bg_label) and the remaining quota of negatives
 * (label == bg_label), each a uniform random subset (keys from splitmix64(seed, image, kind, index); radix select of the k-th smallest).
 * labels / out: (N, R) int8; out = 1 sampled positive, 0 sampled negative, -1 otherwise; counts: (N, 2) int32 = drawn positives /
 * negatives.  No host synchronisation (the reference: two nonzero() + two randperm() per image). */
int sod_sample_labels(const signed char* labels, int N, int R, int num_samples, float positive_fraction, int bg_label,
                      unsigned long long seed, signed char* out, int* counts, void* stream);
/* the same draw + the drawn indices per image as an unordered list idx (N, num_samples) int32 (-1 padded); list_n: (N,) int32 scratch */
int sod_sample_labels_list(const signed char* labels, int N, int R, int num_samples, float positive_fraction, int bg_label,
                           unsigned long long seed, signed char* out, int* counts, int* idx, int* list_n, void* stream);
/* indices of the sampled elements (mask == 1 first, then mask == 0, each in index order) into S slots per image, -1 padded; num (N) */
int sod_compact_samples(const signed char* mask, int N, int R, int S, int* idx, int* num, void* stream);

/* RPN.losses on the sampled anchors only (detectron2 rpn.py losses(): the sums run over the <= BATCH_SIZE_PER_IMAGE sampled anchors of an
   image).  idx (N, S) int32 = sod_compact_samples' output over the sampled labels (anchor index in the concatenated (level, h, w, a)
   order, -1 = empty slot).  gather: rows of the per-level padded NHWC head outputs (level l: hw[l] pixels per image, logit_pitch[l] >= A
   and delta_pitch[l] >= A*D floats per pixel) -> row_logits (N, S), row_deltas (N, S, D).  scatter: the gradients of those rows back into
   zero-initialised per-level tensors of the same shapes. */
int sod_rpn_gather_sampled(int nlev, const void* const* logits, const void* const* deltas, const int* hw, const int* logit_pitch,
                           const int* delta_pitch, const int* idx, int N, int S, int A, int D, float* row_logits, float* row_deltas,
                           void* stream);
int sod_rpn_scatter_sampled(int nlev, void* const* dlogits, void* const* ddeltas, const int* hw, const int* logit_pitch,
                            const int* delta_pitch, const int* idx, int N, int S, int A, int D, const float* row_dlogits,
                            const float* row_ddeltas, void* stream);

/* ------------------------------------------------------------------------------------------------------------------------------
 * fp32-STORAGE validation path (csrc/f32_path.hip; Python: SOD_PRECISION=fp32 / layers.functional.set_precision("fp32")).
 * The operators of the training step with fp32 activations, weight copies and gradients and fp32 FMA accumulation, untuned: the
 * configuration in which north_star's "total-loss delta < 1e-3 vs the reference's CPU path after 100 iterations" is asserted
 * (tests/test_gpu_parity100.py).  Same operand conventions as the bf16 entry points they shadow (NHWC, KRSC weights, image strides
 * in elements, flags SOD_CONV_RELU | SOD_CONV_RES_UP2); every reduction has a fixed order.
 *   conv fwd / dgrad / wgrad: ATen conv2d of detectron2 ResNet / FPN and of FCOSHead (slender_det/modeling/meta_arch/fcos/fcosv2.py:277-381).
 *   dgrad reads the KRSC weights (no transposed copy); `accum` is added (at the even pixels only with accum_even, shape (N,H/2,W/2,C))
 *   before `relu_mask` (> 0 keeps) is applied; wgrad ADDS qscale[k] * sum into dw. */
int sod_conv2d_fwd_f32(const float* x, const float* w_krsc, const float* bias, const float* res, float* y, int N, int H, int W, int C, int K,
                       int R, int S, int stride, int pad, int dil, long long x_img_stride, long long y_img_stride, int flags, void* stream);
int sod_conv2d_dgrad_f32(const float* dy, const float* w_krsc, const float* accum, const float* relu_mask, float* dx, int N, int H, int W, int C,
                         int K, int R, int S, int stride, int pad, int dil, long long dy_img_stride, int accum_even, void* stream);
int sod_conv2d_wgrad_f32(const float* dy, const float* x, float* dw, const float* qscale, int N, int H, int W, int C, int K, int R, int S,
                         int stride, int pad, int dil, long long dy_img_stride, long long x_img_stride, void* stream);
/* nn.GroupNorm(G, C) (+ ReLU) of the towers (fcosv2.py:315-336): two-pass statistics; backward also adds dgamma / dbeta and (optional)
 * the per-channel sum of dx (the bias gradient of the preceding conv) in place.  256 % (C / G) == 0. */
int sod_groupnorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean_rstd, int N, int HW, int C, int G,
                          float eps, int relu, void* stream);
int sod_groupnorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean_rstd, float* dx,
                          float* dgamma, float* dbeta, float* dxsum, int N, int HW, int C, int G, int relu, void* stream);
/* op 0: out = relu(a); 1: out = b > 0 ? a : 0 (ReLU backward, a = dy, b = y); 2: out = a + b */
int sod_eltwise_f32(int op, const float* a, const float* b, float* out, long long n, void* stream);
/* FPN top-down sum a + nearest2x(b) and the gradient of the up-sampling; BasicStem max-pool (C.9) */
int sod_add_up2_f32(const float* a, const float* b, float* out, int N, int H, int W, int C, void* stream);
int sod_upsample2x_bwd_f32(const float* g, float* dprev, int N, int Hc, int Wc, int C, void* stream);
int sod_maxpool3x3s2_f32(const float* x, float* y, int N, int H, int W, int C, void* stream);
/* dbias[c] += sum over images and pixels of dy (image stride in elements, 0 = dense) */
int sod_bias_grad_f32(const float* dy, float* dbias, int N, int HW, int C, long long img_stride, void* stream);
/* FCOSV2.preprocess_image (fcosv2.py:268-275) into an fp32 NHWC(Cpad) image; mean3 / std3 are HOST pointers */
int sod_preprocess_image_f32(const void* img, int is_uint8, int C, int H, int W, float* out, int Hp, int Wp, int Cpad, const float* mean3,
                             const float* std3, void* stream);
/* fp32 compute copies of a master weight: KRSC times scale[k] (FrozenBN fold), input channels zero-padded to Cpad; optional CRSK */
int sod_weight_prep_f32(const float* w, const float* scale, float* w_krsc, float* w_crsk, int K, int RS, int C, int Cpad, void* stream);
/* sod_retina_box_loss_bwd / sod_retina_giou_loss_bwd with fp32 gradient rows (fp32 validation mode of RetinaNet, retina_rotated.py:185-249) */
int sod_retina_box_loss_bwd_f32(const float* pred, int pitch, const int* gt_labels, const float* gt_deltas, int N, int R, int A,
                                int num_classes, float beta, const float* grad_num, const float* grad_den, float* dpred, void* stream);
int sod_retina_giou_loss_bwd_f32(const float* pred, int pitch, const int* gt_labels, const float* anchors, const float* matched_boxes,
                                 int N, int R, int A, int num_classes, const float* weights4, float scale_clamp,
                                 const float* grad_num, const float* grad_den, float* dpred, void* stream);
/* sod_fcos_regctr_loss_bwd with fp32 gradient rows */
int sod_fcos_regctr_loss_bwd_f32(const float* box_raw, int ld_box, const float* ctr_logit, int ld_ctr, const int* labels,
                                 const float* reg_targets, const float* ctr_targets, const float* scales, int N, int nlevels, const int* lvl_h,
                                 const int* lvl_w, const int* lvl_stride, int num_classes, int loss_type, int norm_reg_targets,
                                 const float* grad_reg, const float* grad_ctr, const float* norm, float inv_world, void* dbox, int ld_out,
                                 int ctr_col, void* dctr, int ld_dctr, int dctr_col, float* dscales, float* ws, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SLENDER_HIP_H_ */
