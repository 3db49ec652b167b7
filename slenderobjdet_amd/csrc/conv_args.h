// Argument blocks shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv_igemm256.hip).
#pragma once
#include "common.h"
#include "../../include/slender_hip.h"
#include <stdlib.h>

namespace sodconv {

enum { MODE_FWD = 0, MODE_DGRAD = 1 };
enum {
  F_BIAS = 1, F_RELU = 2, F_RES = 4, F_RES_UP2 = 8, F_MASK = 16,
  F_WBITS = 64,        // bf16 forward output: bit (dst element index) of LevelGeo::bits = stored value > 0 (1-bit ReLU mask for backward)
  F_MASKBITS = 128,    // like F_MASK, but LevelGeo::mask is such a bit array (1/16 of the bytes of the bf16 tensor)
  F_GNSTATS = 32,
  F_REVERSE = 1024,    // walk the tiles last to first: a consumer that starts where its producer stopped finds that part of the tensor in the
                       // Infinity Cache (sod_conv_set_reverse)
  F_RES_EVEN = 512,    // with F_RES_UP2 (dgrad): the half-resolution residual is added at EVEN (h, w) only = the compact data gradient of a
                       // stride-2 1x1 consumer scattered back, without materialising the zero-stuffed tensor
};
constexpr int MAXLEV = SOD_CONV_MAX_LEVELS;

// One "level" = one (N,H,W,C) tensor; a launch may cover several levels that share the weights (the FPN levels of
// the FCOS towers), so that the small levels do not pay a launch + tail each.
struct LevelGeo {
  const void* src;     // fwd: x (N,Hs,Ws,Cred); dgrad: dy (N,Hs,Ws,Cred)
  void* dst;           // (N,Hp,Wp,Nout) rows at dst_img_stride
  const void* res;     // bf16, indexed like dst (or half-resolution with F_RES_UP2)
  const void* mask;    // bf16, indexed like dst: dst = mask>0 ? v : 0 (ReLU backward)
  void* bits;          // F_WBITS: uint8 [N * dst_img_stride / 8]
  float* gn_sum;       // F_GNSTATS: [N][gn_G][2] running (sum, sum of squares), accumulated with float atomics
  uint32_t src_bytes;
  int Hs, Ws, Hp, Wp, P;
  int tile0;           // first pixel tile of this level
  int pstart;          // first pixel this launch covers (a launch may handle only the tail of a level, see dispatch_conv)
  int src_img_stride, dst_img_stride, res_img_stride;  // elements
  FastDiv div_hw, div_w;
};

struct ConvArgs {
  LevelGeo lev[MAXLEV];
  int nlev;
  const void* w;       // [Nout][R*S*Cred]
  const float* bias;   // [Nout] or null
  uint32_t w_bytes;
  int N, Cred, Nout;
  int R, S, stride, pad, dil;
  int Kred, T;         // R*S*Cred, #K-steps
  int flags;
  int nq_tiles, np_tiles;
  FastDiv div_cpt /* Cred/64 (fast) or Cred/8 (generic) */, div_s, div_stride;
  FastDiv div_rs;      // R*S
  int gn_G;            // F_GNSTATS: number of groups (Nout / gn_G == 8: the 8 channels a lane stores are one group)
  int tap_inner;       // linear path: K-step order (channel chunk outer, tap inner) - the taps of one chunk re-read the same cache lines
  int Cpitch;          // channels per pixel of the SOURCE tensor (= Cred except in window mode)
  int cwin;            // channel WINDOW (grouped convolutions, ResNeXt): the 128 output channels of a q-tile contract over the 128 source
                       // channels at the same offset only; the weights are [Nout][R*S][128] (block-diagonal inside the window), Cred = 128
};

// K-step order of the linear staging path: (channel chunk outer, tap inner) makes the R*S taps of one 64-channel chunk re-read the
// same cache lines back to back.  Measured on the FCOS head (16 x 5 levels, 256 -> 256 3x3): L2 fetch traffic of the 256x256 kernel
// 707 -> 205 MB per launch (algorithmic: 184 MB) at equal time; the 16-channel variant of the 128x128 kernel (box / centerness
// prediction, Nout = 8) halves its time (1.59 GB of L2 fetches per launch before); the other variants measure equal and keep the
// (tap outer) order their model-level parity tests were pinned with.  (Tap-inner for EVERY variant of the 128x128 kernel measured +0.24 % on the step in round 4 and moves the fp32 summation
// order of every 3x3 convolution - one sampling-sensitive gradient of the bf16 RepPoints test then leaves its bar; not taken, knob removed.)
inline int conv_tap_inner(int dflt) { return dflt; }

// GroupNorm statistics gathered in the conv epilogue (F_GNSTATS): the FCOS tower unit is conv3x3 -> GroupNorm(32) -> ReLU
// (slender_det/modeling/meta_arch/fcos/fcosv2.py:300-336) and the statistics pass would re-read the tensor the epilogue just held in
// registers.  A lane owns 8 consecutive channels of a pixel = one group of GroupNorm(32, 256); it accumulates over its pixels (flushing
// when the image index changes), then the lanes that own the same channels are summed with shuffles and one lane per group issues the
// two atomics - unless the wave's 64 pixels straddle two images, in which case every lane flushes its own partial.
struct GnAcc {
  float s, ss;
  int n;
};
__device__ __forceinline__ void gn_acc_flush(const GnAcc& g, float* sums, int G, int grp) {
  if (g.n >= 0) {
    atomicAdd(sums + ((size_t)g.n * G + grp) * 2, g.s);
    atomicAdd(sums + ((size_t)g.n * G + grp) * 2 + 1, g.ss);
  }
}
__device__ __forceinline__ void gn_acc_add(GnAcc& g, int n, const bf16x8_t& o, float* sums, int G, int grp) {
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) { const float f = (float)o[e]; s += f; ss += f * f; }     // the STORED (rounded) values, as gn_stats reads them
  if (n != g.n) { gn_acc_flush(g, sums, G, grp); g.n = n; g.s = 0.f; g.ss = 0.f; }
  g.s += s; g.ss += ss;
}
// LPR = lanes per pixel row (lanes l, l + LPR, l + 2 LPR, ... own the same channels); pa .. pb = the wave's pixel range
template <int LPR>
__device__ __forceinline__ void gn_acc_finish(GnAcc& g, uint32_t pa, uint32_t pb, uint32_t P, const FastDiv& div_hw, float* sums, int G, int grp,
                                              bool qok, int lane) {
  if (pa >= P) return;                                   // wave-uniform
  if (pb >= P) pb = P - 1;
  const int na = (int)fd_div(pa, div_hw), nb = (int)fd_div(pb, div_hw);
  if (na == nb) {                                        // wave-uniform: every partial belongs to image na (or is empty)
    float s = (g.n >= 0) ? g.s : 0.f, ss = (g.n >= 0) ? g.ss : 0.f;
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
    if (lane < LPR && qok) {
      atomicAdd(sums + ((size_t)na * G + grp) * 2, s);
      atomicAdd(sums + ((size_t)na * G + grp) * 2 + 1, ss);
    }
  } else if (qok) {
    gn_acc_flush(g, sums, G, grp);
  }
}

// conv_igemm256.hip: 256x256x64 tile, 8 waves, 8-phase main loop.  Returns SOD_EARG when the shape is outside its fast path.
bool conv256_supported(const ConvArgs& a, int mode);
// max_pt_tiles > 0 launches only the first max_pt_tiles pixel tiles (the caller covers the rest with the 128x128 kernel).
int launch_conv256(const ConvArgs& a, int mode, bool out_f32, int max_pt_tiles, hipStream_t st);

// conv_pw.hip: persistent weight-stationary kernel for the expanding 1x1 convolutions of the bottleneck blocks (and their data gradients)
bool pw_supported(const ConvArgs& a, int mode, bool out_f32, int cus);
int launch_pw(const ConvArgs& a, int mode, hipStream_t st);

// conv_ws3.hip: persistent weight-stationary kernel for the 3x3 / stride-1 128 -> 128 convolutions (conv2 of the res3 bottleneck blocks and its
// data gradient): 295 KB of weights in registers, a 10 x 16 halo window per 8 x 14 output tile staged once for the nine taps
bool ws3_supported(const ConvArgs& a, int mode, bool out_f32, int cus, bool any_size);
int launch_ws3(const ConvArgs& a, int mode, hipStream_t st);

// --------------------------------------------------------------------------------------------
// wgrad: dW[q][tap][c] += sum_p dY[p][q] * X[p shifted by tap][c]
// The contraction runs over a VIRTUAL pixel index that concatenates the levels (each padded to a multiple of 64), so
// one launch reduces over all FPN levels that share the weights.  (conv_igemm.hip: 128x128 tiles; conv_wgrad256.hip: 256x256.)
// --------------------------------------------------------------------------------------------
struct WLevel {
  const void* dy;      // (N,Ho,Wo,K) bf16 rows at dy_img_stride
  const void* x;       // (N,Hx,Wx,C) bf16
  uint32_t dy_bytes, x_bytes;
  int Hx, Wx, Ho, Wo, P;
  int v0;              // first virtual pixel of this level (multiple of 64)
  int dy_img_stride, x_img_stride;
  FastDiv div_hw, div_w;
};

struct WgradArgs {
  WLevel lev[MAXLEV];
  int nlev;
  float* dw;           // [K][R][S][C] fp32, accumulated
  const float* qscale; // optional per-output-channel factor (folded FrozenBN scale)
  int dbg_plain_store; // timing experiment only (SOD_WGRAD_PLAIN=1): racy plain stores instead of atomics
  float* partial;      // optional fp32 slabs: blocks store their partial tile, a reduce kernel sums the splits
  int det;             // deterministic: the reduce kernel adds into dw with plain read-modify-writes in a fixed order
  int N, C, K;
  int R, S, stride, pad, dil;
  int V, nz, v_per_split;   // total virtual pixels; v_per_split multiple of 64
  int QT, CT;
  FastDiv div_s;
  int diag;            // channel window (grouped convolutions): only the tiles with c-tile == q-tile exist, dw is [K][R*S][128]
};

enum { WGRAD_DETERMINISTIC = 1, WGRAD_DIAG = 2 };   // sod_conv2d_wgrad flags

// conv_wgrad256.hip: 256(q) x 256(c) output tile per workgroup of 8 waves, one workgroup per CU, split over pixels with fp32 slabs in
// the caller's workspace and a fixed-order reduce kernel (no atomics anywhere).
bool wgrad256_supported(const WgradArgs& a);
long long wgrad256_workspace_bytes(const WgradArgs& a, int cus);
int launch_wgrad256(WgradArgs& a, int cus, float* ws, long long ws_bytes, hipStream_t st);

// conv_wgrad9.hip: 3x3 / stride 1 / pad 1 with the nine taps in ONE workgroup (128 q x 64 c x 9 taps), both operands enumerated in a padded
// pixel order so that a tap is an LDS row offset; slabs + fixed-order reduce like the 256 kernel.
bool wgrad9_supported(const WgradArgs& a);
long long wgrad9_workspace_bytes(const WgradArgs& a, int cus);
int wgrad9_tiles_per_block(const WgradArgs& a, int cus);
int launch_wgrad9(const WgradArgs& a, int cus, float* ws, long long ws_bytes, hipStream_t st);

// conv_wgrad_fold.hip: few output channels (K <= 80 for 3x3) - the taps folded into the rows of the 128 x 128 tile, X staged once for all
// taps; stride 1, "same" geometry, float atomics (not for the deterministic mode).
bool wgrad_fold_supported(const WgradArgs& a);
int launch_wgrad_fold(WgradArgs& a, int cus, hipStream_t st);

// conv_wgrad_ring.hip: 128 x 128 tile, G groups of 4 waves per workgroup that split the tile's pixel range among themselves and sum
// their partial tiles through LDS.  variant = G * 1000 + NSTAGE * 100 + EPI * 10 + FDB (EPI 0: float atomics in 256-B runs, 1: [128][128]
// fp32 slabs in a.partial for wgrad_reduce_kernel; FDB: fragment reads one K-step ahead).  a.nz counts z-BLOCKS of G pixel ranges.
int launch_wgrad_ring(const WgradArgs& a, int variant, hipStream_t st);

}  // namespace sodconv
