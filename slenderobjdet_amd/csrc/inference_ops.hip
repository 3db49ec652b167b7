// On-device post-processing of the dense detectors: everything between the head outputs and the final
// detections of FCOS.inference / inference_single_image (slender_det/modeling/meta_arch/fcos/fcosv2.py:174-249, fcos.py:385-464)
// for the WHOLE batch without a host round trip:
//
//   fcos_decode_kernel      one workgroup per (image, level): sigmoid -> threshold -> x sigmoid(centerness) -> per-level top-k (exact
//                           radix select over the score bits) -> decode l/t/r/b into boxes -> sqrt; candidates come out in the order
//                           torch's nonzero() gives them (location-major, class-minor), padded to top_n slots per level.
//   nms_class_shift_kernel  detectron2.layers.batched_nms' class-offset trick per image: boxes + class * (max coordinate + 1).
//   nms_mask / nms_scan     the batched form of the kernels in detection_ops.hip: per-image candidate counts are read from device
//                           memory, the scan stops after max_keep survivors (keep[: max_detections_per_image]).
// The reference does all of this per image and per level with boolean indexing, .nonzero(), .item() and topk (one host sync each).
#include "common.h"
#include "../../include/slender_hip.h"

namespace {

constexpr int DEC_THREADS = 1024;
constexpr int DEC_V = 8;                        // consecutive elements per thread in the ordered compaction
constexpr int DEC_CHUNK = DEC_THREADS * DEC_V;

struct DecodeArgs {
  const float* cls;       // (N, L, ld_cls) logits
  const float* box;       // (N, L, ld_box) raw regression (+ centerness logit in column ctr_col_box when >= 0)
  const float* scales;    // [nlev] Scale values (fcos.py:532)
  int N, L, nlev, K, ld_cls, ld_box;
  int ctr_col_box, ctr_col_cls;
  int norm_reg, top_n;
  float thresh;
  int H[SOD_MAX_LEVELS], W[SOD_MAX_LEVELS], stride[SOD_MAX_LEVELS], loc0[SOD_MAX_LEVELS];
  float* out_boxes;       // (N, nlev*top_n, 4)
  float* out_scores;      // (N, nlev*top_n), -1 in unused slots
  int* out_classes;       // (N, nlev*top_n), -1 in unused slots
  int* out_counts;        // (N, nlev)
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// exclusive prefix sum of one unsigned value per thread over the 1024-thread block; *total = sum of all values
__device__ __forceinline__ unsigned block_exscan(unsigned v, unsigned* lds_waves /* >= 17 */, unsigned* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();                       // lds_waves may still be read by the previous call
  if (lane == 63) lds_waves[wave] = inc;
  __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < DEC_THREADS / 64; ++w) {
    const unsigned t = lds_waves[w];
    if (w < wave) base += t;
    tot += t;
  }
  *total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(DEC_THREADS) void fcos_decode_kernel(const DecodeArgs a) {
  __shared__ unsigned hist[256];
  __shared__ unsigned wv[32];
  __shared__ unsigned sel[4];            // [0] bucket, [1] remaining k, [2] block count
  const int l = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int HW = a.H[l] * a.W[l], K = a.K;
  const long long total = (long long)HW * K;
  const float* cls = a.cls + ((long long)n * a.L + a.loc0[l]) * a.ld_cls;
  const float* box = a.box + ((long long)n * a.L + a.loc0[l]) * a.ld_box;
  const bool dense = a.ld_cls == K;      // logits of consecutive (location, class) pairs are consecutive in memory

  auto logit_at = [&](long long e) -> float {
    if (dense) return cls[e];
    const int loc = (int)(e / K);
    return cls[(long long)loc * a.ld_cls + (int)(e - (long long)loc * K)];
  };
  auto ctr_at = [&](int loc) -> float {
    return a.ctr_col_box >= 0 ? box[(long long)loc * a.ld_box + a.ctr_col_box] : cls[(long long)loc * a.ld_cls + a.ctr_col_cls];
  };
  // candidate test and score key of element e (fcosv2.py:206-212: keep = sigmoid(cls) > thresh; score = sigmoid(cls) * sigmoid(ctr))
  auto key_of = [&](long long e, bool& cand) -> unsigned {
    const float p = sigmoidf_(logit_at(e));
    cand = p > a.thresh;
    if (!cand) return 0u;
    const float s = p * sigmoidf_(ctr_at((int)(e / K)));
    return __float_as_uint(s);           // s >= 0: the bit pattern orders like the value
  };

  // ---- pass 1: number of candidates
  unsigned mine = 0;
  for (long long e = tid; e < total; e += DEC_THREADS) {
    bool c;
    (void)key_of(e, c);
    mine += c ? 1u : 0u;
  }
  unsigned cnt;
  (void)block_exscan(mine, wv, &cnt);

  // ---- pass 2 (only when more than top_n candidates): the top_n-th largest score key T and how many elements equal to T are taken
  unsigned T = 0u, quota = 0xffffffffu;
  if (cnt > (unsigned)a.top_n) {
    unsigned prefix = 0u, krem = (unsigned)a.top_n;
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (tid < 256) hist[tid] = 0u;
      __syncthreads();
      for (long long e = tid; e < total; e += DEC_THREADS) {
        bool c;
        const unsigned key = key_of(e, c);
        if (c && (shift == 24 || (key >> (shift + 8)) == prefix)) atomicAdd(&hist[(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        unsigned cum = 0u;
        int b = 255;
        for (; b > 0; --b) {
          if (cum + hist[b] >= krem) break;
          cum += hist[b];
        }
        sel[0] = (unsigned)b; sel[1] = krem - cum;     // krem - cum of the elements in bucket b are still wanted
      }
      __syncthreads();
      prefix = (prefix << 8) | sel[0];
      krem = sel[1];
      __syncthreads();
    }
    T = prefix; quota = krem;
  }
  const unsigned count = cnt > (unsigned)a.top_n ? (unsigned)a.top_n : cnt;

  // ---- pass 3: ordered compaction + decode.  Selected = candidate with key > T, or key == T among the first `quota` such elements.
  const long long slot0 = ((long long)n * a.nlev + l) * a.top_n;
  const float scale = a.scales[l];
  const int W = a.W[l], stride = a.stride[l];
  unsigned out_base = 0u, eq_seen = 0u;
  for (long long c0 = 0; c0 < total; c0 += DEC_CHUNK) {
    const long long e0 = c0 + (long long)tid * DEC_V;
    unsigned keys[DEC_V];
    unsigned gt_bits = 0u, eq_bits = 0u;
#pragma unroll
    for (int v = 0; v < DEC_V; ++v) {
      bool c = false;
      keys[v] = (e0 + v < total) ? key_of(e0 + v, c) : 0u;
      if (c) {
        if (cnt <= (unsigned)a.top_n || keys[v] > T) gt_bits |= 1u << v;
        else if (keys[v] == T) eq_bits |= 1u << v;
      }
    }
    const unsigned ngt = __popc(gt_bits), neq = __popc(eq_bits);
    unsigned tot;
    const unsigned ex = block_exscan((neq << 16) | ngt, wv, &tot);
    const unsigned ex_gt = ex & 0xffffu, ex_eq = ex >> 16, tot_gt = tot & 0xffffu, tot_eq = tot >> 16;
    const unsigned q_rem = quota > eq_seen ? quota - eq_seen : 0u;                 // equal-key elements still wanted at chunk start
    unsigned pos = out_base + ex_gt + (ex_eq < q_rem ? ex_eq : q_rem);
    unsigned eq_rank = ex_eq;
#pragma unroll
    for (int v = 0; v < DEC_V; ++v) {
      bool take = (gt_bits >> v) & 1u;
      if ((eq_bits >> v) & 1u) { take = eq_rank < q_rem; ++eq_rank; }
      if (take) {
        const long long e = e0 + v;
        const int loc = (int)(e / K), c = (int)(e - (long long)loc * K);
        const int i = loc / W, j = loc - i * W;
        const float x = (float)(j * stride + stride / 2), y = (float)(i * stride + stride / 2);
        const float* br = box + (long long)loc * a.ld_box;
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float z = br[k] * scale;
          r[k] = a.norm_reg ? fmaxf(z, 0.f) * (float)stride : expf(z);
        }
        float* ob = a.out_boxes + (slot0 + pos) * 4;
        ob[0] = x - r[0]; ob[1] = y - r[1]; ob[2] = x + r[2]; ob[3] = y + r[3];
        a.out_scores[slot0 + pos] = sqrtf(__uint_as_float(keys[v]));
        a.out_classes[slot0 + pos] = c;
        ++pos;
      }
    }
    out_base += tot_gt + (tot_eq < q_rem ? tot_eq : q_rem);
    eq_seen += tot_eq;
  }
  // unused slots: score -1 sorts behind every real candidate
  for (unsigned s = count + tid; s < (unsigned)a.top_n; s += DEC_THREADS) {
    float* ob = a.out_boxes + (slot0 + s) * 4;
    ob[0] = ob[1] = ob[2] = ob[3] = 0.f;
    a.out_scores[slot0 + s] = -1.f;
    a.out_classes[slot0 + s] = -1;
  }
  if (tid == 0) a.out_counts[n * a.nlev + l] = (int)count;
}

// batched_nms' class offsets (torchvision / detectron2.layers.batched_nms): shifted = boxes + class * (max over the image's boxes + 1)
__global__ __launch_bounds__(1024) void nms_class_shift_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                               const int* __restrict__ classes, int M, float* __restrict__ shifted,
                                                               int* __restrict__ nvalid) {
  __shared__ float red[16];
  __shared__ unsigned cntw[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* bx = boxes + (long long)b * M * 4;
  const float* sc = scores + (long long)b * M;
  float mx = -3.0e38f;
  unsigned cnt = 0;
  for (int i = tid; i < M; i += 1024)
    if (sc[i] >= 0.f) {
      ++cnt;
#pragma unroll
      for (int k = 0; k < 4; ++k) mx = fmaxf(mx, bx[i * 4 + k]);
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o, 64)); cnt += __shfl_xor(cnt, o, 64); }
  if ((tid & 63) == 0) { red[tid >> 6] = mx; cntw[tid >> 6] = cnt; }
  __syncthreads();
  mx = red[0]; cnt = cntw[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) { mx = fmaxf(mx, red[w]); cnt += cntw[w]; }
  const float step = mx + 1.f;
  for (int i = tid; i < M; i += 1024) {
    const float off = sc[i] >= 0.f ? (float)classes[(long long)b * M + i] * step : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) shifted[((long long)b * M + i) * 4 + k] = bx[i * 4 + k] + off;
  }
  if (tid == 0) nvalid[b] = (int)cnt;
}

__device__ __forceinline__ bool iou_gt4(const float* a, const float* b, float thr) {       // as detection_ops.hip:iou_gt
  const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
  const float inter = width * height;
  const float sa = (a[2] - a[0]) * (a[3] - a[1]);
  const float sb = (b[2] - b[0]) * (b[3] - b[1]);
  return inter / (sa + sb - inter) > thr;
}

// mask[b][i][w] bit j = IoU(box order[i], box order[64 w + j]) > thr for j > i; grid (words, words, B); n read per image
__global__ __launch_bounds__(64) void nms_mask_batched_kernel(const float* __restrict__ boxes, const long long* __restrict__ order,
                                                              const int* __restrict__ nvalid, int M, float thr,
                                                              unsigned long long* __restrict__ mask, int words) {
  const int b = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x;
  const int n = min(nvalid[b], M);
  if (cb < rb || rb * 64 >= n || cb * 64 >= n) return;
  boxes += (long long)b * M * 4; order += (long long)b * M; mask += (long long)b * M * words;
  __shared__ float cbox[64 * 4];
  const int lane = threadIdx.x;
  const int cj = cb * 64 + lane;
  if (cj < n) {
    const long long o = order[cj];
#pragma unroll
    for (int e = 0; e < 4; ++e) cbox[lane * 4 + e] = boxes[o * 4 + e];
  }
  __syncthreads();
  const int i = rb * 64 + lane;
  if (i >= n) return;
  float a4[4];
  const long long oi = order[i];
#pragma unroll
  for (int e = 0; e < 4; ++e) a4[e] = boxes[oi * 4 + e];
  unsigned long long bits = 0;
  const int cnt = min(64, n - cb * 64);
  for (int j = (rb == cb) ? lane + 1 : 0; j < cnt; ++j)
    if (iou_gt4(a4, cbox + j * 4, thr)) bits |= 1ull << j;
  mask[(long long)i * words + cb] = bits;
}

// one workgroup per image; as nms_scan_kernel (detection_ops.hip) with the count read from device memory and an early stop once
// max_keep boxes survived (the reference slices keep[: max_detections_per_image])
__global__ __launch_bounds__(1024) void nms_scan_batched_kernel(const unsigned long long* __restrict__ mask, const long long* __restrict__ order,
                                                                const int* __restrict__ nvalid, int M, int words, int max_keep,
                                                                long long* __restrict__ keep, int* __restrict__ nkeep) {
  __shared__ unsigned long long removed[1024];
  __shared__ unsigned long long chunk_keep;
  const int b = blockIdx.x, w = threadIdx.x;
  const int n = min(nvalid[b], M);
  mask += (long long)b * M * words; order += (long long)b * M; keep += (long long)b * max_keep;
  const int nw = (n + 63) / 64;
  if (w < words) removed[w] = 0;
  __syncthreads();
  int kept = 0;
  for (int c = 0; c < nw && kept < max_keep; ++c) {
    const int cnt = min(64, n - c * 64);
    if (w < 64) {
      const long long i = (long long)c * 64 + w;
      const unsigned long long diag = (w < cnt) ? mask[i * words + c] : 0ull;
      const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
      unsigned long long rem = removed[c], kb = 0ull;
      for (int j = 0; j < cnt; ++j) {
        const unsigned long long dj = ((unsigned long long)__shfl(dhi, j, 64) << 32) | (unsigned long long)__shfl(dlo, j, 64);
        if (!((rem >> j) & 1ull)) { kb |= 1ull << j; rem |= dj; }
      }
      if (w == 0) chunk_keep = kb;
    }
    __syncthreads();
    const unsigned long long kb = chunk_keep;
    {
      const int slices = 1024 / words;
      const int ww = w % words, sl = w / words;
      if (sl < slices && ww > c && ww < nw) {
        unsigned long long acc = 0ull;
        for (int j = sl; j < cnt; j += slices)
          if ((kb >> j) & 1ull) acc |= mask[((long long)c * 64 + j) * words + ww];
        if (acc) atomicOr(&removed[ww], acc);
      }
    }
    if (w < 64 && ((kb >> w) & 1ull)) {
      const int pos = kept + __popcll(kb & ((1ull << w) - 1ull));
      if (pos < max_keep) keep[pos] = order[(long long)c * 64 + w];
    }
    kept += __popcll(kb);
    __syncthreads();
  }
  if (w == 0) nkeep[b] = kept < max_keep ? kept : max_keep;
}

}  // namespace

extern "C" int sod_fcos_decode(const float* cls_logits, int ld_cls, const float* box_raw, int ld_box, const float* scales,
                               int N, int nlev, const int* H, const int* W, const int* strides, int num_classes,
                               int ctr_col_box, int ctr_col_cls, int norm_reg_targets, float pre_nms_thresh, int pre_nms_top_n,
                               float* out_boxes, float* out_scores, int* out_classes, int* out_counts, void* stream) {
  if (!cls_logits || !box_raw || !scales || !H || !W || !strides || !out_boxes || !out_scores || !out_classes || !out_counts) return SOD_EARG;
  if (N <= 0 || nlev <= 0 || nlev > SOD_MAX_LEVELS || num_classes <= 0 || ld_cls < num_classes || ld_box < 4 || pre_nms_top_n <= 0) return SOD_EARG;
  if ((ctr_col_box >= 0) == (ctr_col_cls >= 0) || ctr_col_box >= ld_box || ctr_col_cls >= ld_cls) return SOD_EARG;
  if (pre_nms_top_n >= 65536) return SOD_EARG;       // packed 16-bit block counts
  DecodeArgs a{};
  a.cls = cls_logits; a.box = box_raw; a.scales = scales;
  a.N = N; a.nlev = nlev; a.K = num_classes; a.ld_cls = ld_cls; a.ld_box = ld_box;
  a.ctr_col_box = ctr_col_box; a.ctr_col_cls = ctr_col_cls; a.norm_reg = norm_reg_targets; a.top_n = pre_nms_top_n; a.thresh = pre_nms_thresh;
  long long L = 0;
  for (int l = 0; l < nlev; ++l) {
    if (H[l] <= 0 || W[l] <= 0 || strides[l] <= 0) return SOD_EARG;
    a.H[l] = H[l]; a.W[l] = W[l]; a.stride[l] = strides[l]; a.loc0[l] = (int)L;
    L += (long long)H[l] * W[l];
    if ((long long)H[l] * W[l] * num_classes >= (1ll << 31)) return SOD_ESIZE;
  }
  if (L * N >= (1ll << 31)) return SOD_ESIZE;
  a.L = (int)L;
  a.out_boxes = out_boxes; a.out_scores = out_scores; a.out_classes = out_classes; a.out_counts = out_counts;
  SOD_LAUNCH(fcos_decode_kernel, dim3(nlev, N), dim3(DEC_THREADS), 0, (hipStream_t)stream, a);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" long long sod_batched_nms_workspace_bytes(int B, int M) {
  const long long words = (M + 63) / 64;
  return (long long)B * M * words * 8 + (long long)B * M * 4 * (long long)sizeof(float) + (long long)B * (long long)sizeof(int);
}

// shifted boxes + per-image candidate count (first half of batched NMS); the caller sorts the scores (any stable descending sort)
// and then calls sod_batched_nms_run with the order.  ws layout: [mask][shifted boxes][nvalid].
extern "C" int sod_batched_nms_prepare(const float* boxes, const float* scores, const int* classes, int B, int M, void* ws, void* stream) {
  if (!boxes || !scores || !classes || !ws || B <= 0 || M <= 0 || M > 65536) return SOD_EARG;
  const long long words = (M + 63) / 64;
  float* shifted = (float*)((char*)ws + (long long)B * M * words * 8);
  int* nvalid = (int*)(shifted + (long long)B * M * 4);
  SOD_LAUNCH(nms_class_shift_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, boxes, scores, classes, M, shifted, nvalid);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_batched_nms_run(const long long* order, int B, int M, float iou_threshold, int max_keep, long long* keep, int* num_keep,
                                   void* ws, void* stream) {
  if (!order || !keep || !num_keep || !ws || B <= 0 || M <= 0 || M > 65536 || max_keep <= 0) return SOD_EARG;
  const int words = (M + 63) / 64;
  if (words > 1024) return SOD_ESIZE;
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* mask = (unsigned long long*)ws;
  const float* shifted = (const float*)((char*)ws + (long long)B * M * words * 8);
  const int* nvalid = (const int*)(shifted + (long long)B * M * 4);
  hipError_t e = hipMemsetAsync(mask, 0, (size_t)B * M * words * 8, st);
  if (e != hipSuccess) return (int)e;
  SOD_LAUNCH(nms_mask_batched_kernel, dim3(words, words, B), dim3(64), 0, st, shifted, order, nvalid, M, iou_threshold, mask, words);
  SOD_LAUNCH(nms_scan_batched_kernel, dim3(B), dim3(1024), 0, st, (const unsigned long long*)mask, order, nvalid, M, words, max_keep, keep, num_keep);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}
