// On-device post-processing of the dense detectors: everything between the head outputs and the final
// detections of FCOS.inference / inference_single_image (slender_det/modeling/meta_arch/fcos/fcosv2.py:174-249, fcos.py:385-464)
// for the WHOLE batch without a host round trip:
//
//   fcos_decode_kernel      one workgroup per (image, level): sigmoid -> threshold -> x sigmoid(centerness) -> per-level top-k (exact
//                           radix select over the score bits) -> decode l/t/r/b into boxes -> sqrt; candidates come out in the order
//                           torch's nonzero() gives them (location-major, class-minor), padded to top_n slots per level.
//   (the batched class-aware NMS that consumes these candidates lives in detection_ops.hip: sod_batched_nms_*)
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
  int mode;               // 0: FCOS (rows = locations, score x centerness, boxes decoded, sqrt)
                          // 1: generic (row, class) selection: rows of K logits, score = sigmoid(logit), emits row indices
                          // 2: generic row selection: score = max over the K classes, class = first argmax, emits row indices
  int* out_rows;          // modes 1 / 2: (N, nlev*top_n) row index inside the level
  float* out_boxes;       // (N, nlev*top_n, 4)
  float* out_scores;      // (N, nlev*top_n), -inf in unused slots
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
  const int HW = a.H[l] * a.W[l], K = a.K;     // modes 1 / 2: H = rows of the level, W = 1
  const long long total = a.mode == 2 ? (long long)HW : (long long)HW * K;
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
    if (a.mode == 2) {                   // rpd.py:741-748: scores, classes = logits.sigmoid().max(1); keep = score > threshold
      const float* row = cls + e * a.ld_cls;
      float m = row[0];
      for (int k = 1; k < K; ++k) m = fmaxf(m, row[k]);
      const float p = sigmoidf_(m);
      cand = p > a.thresh;
      return cand ? __float_as_uint(p) : 0u;
    }
    const float p = sigmoidf_(logit_at(e));
    cand = p > a.thresh;
    if (!cand) return 0u;
    const float s = a.mode == 0 ? p * sigmoidf_(ctr_at((int)(e / K))) : p;
    return __float_as_uint(s);           // s >= 0: the bit pattern orders like the value
  };

  // RetinaNet takes k = min(topk_candidates, number of ANCHORS of the level) of the anchor x class scores (retina_rotated.py:318)
  const int top_n = (a.mode == 1 && HW < a.top_n) ? HW : a.top_n;

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
  if (cnt > (unsigned)top_n) {
    unsigned prefix = 0u, krem = (unsigned)top_n;
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
  const unsigned count = cnt > (unsigned)top_n ? (unsigned)top_n : cnt;

  // ---- pass 3: ordered compaction + decode.  Selected = candidate with key > T, or key == T among the first `quota` such elements.
  const long long slot0 = ((long long)n * a.nlev + l) * a.top_n;
  const float scale = a.mode == 0 ? a.scales[l] : 1.f;      // the generic modes have no Scale / box inputs
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
        if (cnt <= (unsigned)top_n || keys[v] > T) gt_bits |= 1u << v;
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
      if (take && a.mode != 0) {
        const long long e = e0 + v;
        int row = (int)e, c = 0;
        if (a.mode == 1) { row = (int)(e / K); c = (int)(e - (long long)row * K); }
        else {
          const float* lr = cls + e * a.ld_cls;
          float m = lr[0];
          for (int k = 1; k < K; ++k) if (lr[k] > m) { m = lr[k]; c = k; }
        }
        a.out_rows[slot0 + pos] = row;
        a.out_scores[slot0 + pos] = __uint_as_float(keys[v]);
        a.out_classes[slot0 + pos] = c;
        ++pos;
      } else if (take) {
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
  // unused slots: score -inf sorts behind every real candidate and marks the slot empty for the batched NMS
  for (unsigned s = count + tid; s < (unsigned)a.top_n; s += DEC_THREADS) {
    if (a.mode == 0) {
      float* ob = a.out_boxes + (slot0 + s) * 4;
      ob[0] = ob[1] = ob[2] = ob[3] = 0.f;
    } else {
      a.out_rows[slot0 + s] = 0;
    }
    a.out_scores[slot0 + s] = -__builtin_inff();
    a.out_classes[slot0 + s] = -1;
  }
  if (tid == 0) a.out_counts[n * a.nlev + l] = (int)count;
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

/* Generic per-level "threshold + top-k" selection on rows of K class logits (RetinaNet: retina_rotated.py:296-340 / d2
 * inference_single_image; RepPoints: rpd.py:717-765), for the whole batch in one launch.  logits (N, R, ld) with R = sum rows[l]
 * (level-major).  by_row_max = 0: candidates are (row, class) pairs with sigmoid(logit) > thresh, top_n best per (image, level);
 * by_row_max = 1: candidates are rows, scored by their best class.  Outputs as sod_fcos_decode, with the selected row index inside
 * its level instead of a decoded box. */
extern "C" int sod_dense_topk_select(const float* logits, int ld, int N, int nlev, const int* rows, int num_classes, int by_row_max,
                                     float score_thresh, int top_n, int* out_rows, float* out_scores, int* out_classes, int* out_counts,
                                     void* stream) {
  if (!logits || !rows || !out_rows || !out_scores || !out_classes || !out_counts) return SOD_EARG;
  if (N <= 0 || nlev <= 0 || nlev > SOD_MAX_LEVELS || num_classes <= 0 || ld < num_classes || top_n <= 0 || top_n >= 65536) return SOD_EARG;
  DecodeArgs a{};
  a.cls = logits; a.N = N; a.nlev = nlev; a.K = num_classes; a.ld_cls = ld; a.ld_box = 0;
  a.ctr_col_box = a.ctr_col_cls = -1; a.top_n = top_n; a.thresh = score_thresh; a.mode = by_row_max ? 2 : 1;
  long long R = 0;
  for (int l = 0; l < nlev; ++l) {
    if (rows[l] <= 0 || (long long)rows[l] * num_classes >= (1ll << 31)) return SOD_EARG;
    a.H[l] = rows[l]; a.W[l] = 1; a.stride[l] = 1; a.loc0[l] = (int)R;
    R += rows[l];
  }
  if (R * N >= (1ll << 31)) return SOD_ESIZE;
  a.L = (int)R;
  a.out_rows = out_rows; a.out_scores = out_scores; a.out_classes = out_classes; a.out_counts = out_counts;
  SOD_LAUNCH(fcos_decode_kernel, dim3(nlev, N), dim3(DEC_THREADS), 0, (hipStream_t)stream, a);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}
