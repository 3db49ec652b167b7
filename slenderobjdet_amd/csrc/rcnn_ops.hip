// Two-stage (R-CNN) support kernels — detectron2 GeneralizedRCNN / RPN / RRPN / StandardROIHeads / RROIHeads as selected by
// configs/rotated/Base-RRCNN-FPN.yaml and the reference's proposal_generator/rpn.py:26-356, roi_heads/roi_heads.py:48-53
// (third-party sources absent: semantics restated in SURVEY.md §2.3 / Appendix C.4-C.6).  HBM/latency-bound fp32 work:
//   * Box2BoxTransform / Box2BoxTransformRotated get_deltas and apply_deltas (4- and 5-parameter boxes);
//   * RPN objectness BCE-with-logits and localisation smooth-L1 over labelled anchors (labels -1 ignore / 0 / 1);
//   * Fast R-CNN softmax cross-entropy and the class-specific box regression loss.
#include "common.h"
#include "../../include/slender_hip.h"
#include <math.h>

namespace {

constexpr int RC_RED = 1024;
constexpr float RC_PI = 3.14159265358979323846f;

inline int rc_nblk(long long n, int cap = RC_RED) {
  long long g = (n + 255) / 256;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

__global__ void rc_finish(const float* __restrict__ part, int nb, int nsum, float* __restrict__ out) {
  __shared__ float red[4];
  for (int s = 0; s < nsum; ++s) {
    float v = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) v += part[s * RC_RED + i];
    v = block_sum_256(v, red);
    if (threadIdx.x == 0) out[s] = v;
    __syncthreads();
  }
}

struct W5 { float w[5]; };

// d2 Box2BoxTransform.get_deltas (XYXY) / Box2BoxTransformRotated.get_deltas (cx, cy, w, h, angle in degrees)
__global__ __launch_bounds__(256) void get_deltas_kernel(const float* __restrict__ src, const float* __restrict__ tgt, long long n, int D, W5 w,
                                                         float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float* s = src + i * D;
    const float* t = tgt + i * D;
    float* o = out + i * D;
    if (D == 4) {
      const float sw = s[2] - s[0], sh = s[3] - s[1], scx = s[0] + 0.5f * sw, scy = s[1] + 0.5f * sh;
      const float tw = t[2] - t[0], th = t[3] - t[1], tcx = t[0] + 0.5f * tw, tcy = t[1] + 0.5f * th;
      o[0] = w.w[0] * (tcx - scx) / sw; o[1] = w.w[1] * (tcy - scy) / sh;
      o[2] = w.w[2] * logf(tw / sw); o[3] = w.w[3] * logf(th / sh);
    } else {
      o[0] = w.w[0] * (t[0] - s[0]) / s[2]; o[1] = w.w[1] * (t[1] - s[1]) / s[3];
      o[2] = w.w[2] * logf(t[2] / s[2]); o[3] = w.w[3] * logf(t[3] / s[3]);
      float da = t[4] - s[4];
      da = fmodf(da + 180.f, 360.f);
      if (da < 0.f) da += 360.f;            // python's % (result takes the sign of the divisor)
      da -= 180.f;
      o[4] = da * w.w[4] * RC_PI / 180.f;
    }
  }
}

// apply_deltas: deltas (n, k*D), boxes (n, D) -> out (n, k*D)
__global__ __launch_bounds__(256) void apply_deltas_kernel(const float* __restrict__ deltas, const float* __restrict__ boxes, long long n, int k,
                                                           int D, int ld, W5 w, float clampv, float* __restrict__ out) {
  const long long total = n * k;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / k;
    const int c = (int)(i - r * k);
    const float* b = boxes + r * D;
    const float* d = deltas + r * ld + c * D;
    float* o = out + (r * k + c) * D;
    if (D == 4) {
      const float bw = b[2] - b[0], bh = b[3] - b[1], cx = b[0] + 0.5f * bw, cy = b[1] + 0.5f * bh;
      const float dx = d[0] / w.w[0], dy = d[1] / w.w[1], dw = fminf(d[2] / w.w[2], clampv), dh = fminf(d[3] / w.w[3], clampv);
      const float pcx = dx * bw + cx, pcy = dy * bh + cy, pw = expf(dw) * bw, ph = expf(dh) * bh;
      o[0] = pcx - 0.5f * pw; o[1] = pcy - 0.5f * ph; o[2] = pcx + 0.5f * pw; o[3] = pcy + 0.5f * ph;
    } else {
      const float dx = d[0] / w.w[0], dy = d[1] / w.w[1], dw = fminf(d[2] / w.w[2], clampv), dh = fminf(d[3] / w.w[3], clampv);
      const float da = d[4] / w.w[4];
      o[0] = dx * b[2] + b[0]; o[1] = dy * b[3] + b[1];
      o[2] = expf(dw) * b[2]; o[3] = expf(dh) * b[3];
      float a = da * 180.f / RC_PI + b[4];
      a = fmodf(a + 180.f, 360.f);
      if (a < 0.f) a += 360.f;
      o[4] = a - 180.f;
    }
  }
}

// F.binary_cross_entropy_with_logits(logits[valid], labels[valid], reduction="sum") with labels in {-1 (ignored), 0, 1}
template <bool BWD>
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ x, const signed char* __restrict__ lab, long long n,
                                                  float* __restrict__ part, const float* __restrict__ gs, float mul, float* __restrict__ dx) {
  __shared__ float red[4];
  float acc = 0.f;
  const float sc = BWD ? gs[0] * mul : 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int l = lab[i];
    const float v = x[i];
    if (!BWD) {
      if (l >= 0) acc += fmaxf(v, 0.f) - v * (float)l + log1pf(expf(-fabsf(v)));
    } else {
      dx[i] = l >= 0 ? (1.f / (1.f + expf(-v)) - (float)l) * sc : 0.f;
    }
  }
  if (!BWD) {
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
  }
}

// F.binary_cross_entropy_with_logits(x[fg], t[fg], reduction="sum") with float targets; fg = rows with 0 <= label != bg
// (the centerness loss of the LRTB / FCOS heads, meta/heads/lrtb_head.py:236-238)
template <bool BWD>
__global__ __launch_bounds__(256) void bce_soft_kernel(const float* __restrict__ x, const float* __restrict__ t, const int* __restrict__ lab, int bg,
                                                       long long n, float* __restrict__ part, const float* __restrict__ gs, float mul,
                                                       float* __restrict__ dx) {
  __shared__ float red[4];
  float acc = 0.f;
  const float sc = BWD ? gs[0] * mul : 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int l = lab[i];
    const bool sel = l >= 0 && l != bg;
    const float v = x[i];
    if (!BWD) {
      if (sel) acc += fmaxf(v, 0.f) - v * t[i] + log1pf(expf(-fabsf(v)));
    } else {
      dx[i] = sel ? (1.f / (1.f + expf(-v)) - t[i]) * sc : 0.f;
    }
  }
  if (!BWD) {
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
  }
}

// smooth_l1_loss(pred[label == 1], target[label == 1], beta, "sum") over D-vectors
template <bool BWD>
__global__ __launch_bounds__(256) void loc_kernel(const float* __restrict__ p, const float* __restrict__ t, const signed char* __restrict__ lab,
                                                  long long n, int D, float beta, float* __restrict__ part, const float* __restrict__ gs, float mul,
                                                  float* __restrict__ dp) {
  __shared__ float red[4];
  float acc = 0.f;
  const float sc = BWD ? gs[0] * mul : 0.f;
  const long long total = n * D;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const bool pos = lab[i / D] == 1;
    float g = 0.f;
    if (pos) {
      const float d = p[i] - t[i], ad = fabsf(d);
      if (beta < 1e-5f) { acc += ad; g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
      else if (ad < beta) { acc += 0.5f * d * d / beta; g = d / beta; }
      else { acc += ad - 0.5f * beta; g = d > 0.f ? 1.f : -1.f; }
    }
    if (BWD) dp[i] = g * sc;
  }
  if (!BWD) {
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
  }
}

// F.cross_entropy(scores, labels, reduction="sum"): one wave per row (C <= 4096), labels < 0 ignored
template <bool BWD>
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ x, const int* __restrict__ lab, int R, int C, int ld,
                                                 float* __restrict__ part, const float* __restrict__ gs, float mul, float* __restrict__ dx) {
  __shared__ float red[4];
  float acc = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float sc = BWD ? gs[0] * mul : 0.f;
  for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
    const float* row = x + (long long)r * ld;
    const int l = lab[r];
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, row[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(row[c] - m);
    s = wave_sum(s);
    if (!BWD) {
      if (l >= 0 && lane == 0) acc += logf(s) + m - row[l];
    } else {
      float* drow = dx + (long long)r * ld;
      for (int c = lane; c < ld; c += 64) {
        float g = 0.f;
        if (l >= 0 && c < C) g = (expf(row[c] - m) / s - (c == l ? 1.f : 0.f)) * sc;
        drow[c] = g;
      }
    }
  }
  if (!BWD) {
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
  }
}

// FastRCNNOutputs.smooth_l1_loss: rows with 0 <= class < K use the D columns of their class
template <bool BWD>
__global__ __launch_bounds__(256) void frcnn_box_kernel(const float* __restrict__ pred, const int* __restrict__ cls, const float* __restrict__ tgt,
                                                        int R, int K, int D, int ld, float beta, float* __restrict__ part,
                                                        const float* __restrict__ gs, float mul, float* __restrict__ dp) {
  __shared__ float red[4];
  float acc = 0.f;
  const float sc = BWD ? gs[0] * mul : 0.f;
  if (BWD) {
    const long long total = (long long)R * ld;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
      const int r = (int)(i / ld), c = (int)(i - (long long)r * ld);
      const int k = cls[r];
      float g = 0.f;
      if (k >= 0 && k < K && c >= k * D && c < (k + 1) * D) {
        const float d = pred[i] - tgt[(long long)r * D + (c - k * D)], ad = fabsf(d);
        if (beta < 1e-5f) g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        else if (ad < beta) g = d / beta;
        else g = d > 0.f ? 1.f : -1.f;
      }
      dp[i] = g * sc;
    }
  } else {
    const long long total = (long long)R * D;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
      const int r = (int)(i / D), j = (int)(i - (long long)r * D);
      const int k = cls[r];
      if (k >= 0 && k < K) {
        const float d = pred[(long long)r * ld + k * D + j] - tgt[i], ad = fabsf(d);
        if (beta < 1e-5f) acc += ad;
        else if (ad < beta) acc += 0.5f * d * d / beta;
        else acc += ad - 0.5f * beta;
      }
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
  }
}

inline bool fill_w(W5& w, const float* weights, int D) {
  if (!weights || (D != 4 && D != 5)) return false;
  for (int i = 0; i < D; ++i) { w.w[i] = weights[i]; if (!(w.w[i] > 0.f)) return false; }
  return true;
}

}  // namespace


// ------------------------------------------------------------------------------------------------------------------------------
// detectron2.modeling.sampling.subsample_labels for a whole batch, on the device (the reference reaches it through RPN
// label_and_sample_anchors, proposal_generator/rpn.py:137-191, and ROIHeads.label_and_sample_proposals): per image, up to
// int(num_samples * positive_fraction) POSITIVES (label != -1 and != bg) and the remaining quota of NEGATIVES (label == bg), each drawn
// uniformly without replacement.  The reference does this with two nonzero() + two randperm() per image (four host syncs per image);
// here one workgroup per (image, kind) gives every candidate a 64-bit key = (splitmix64(seed, image, kind, index) high word, index) -
// unique, so "the k smallest keys" is a uniform k-subset - and finds the k-th smallest key by an 8-pass radix select over LDS histograms.
// out[i] = 1 sampled positive, 0 sampled negative, -1 everything else; counts[n] = {positives, negatives} drawn.
__device__ __forceinline__ unsigned long long sample_key(unsigned long long seed, int n, int kind, uint32_t i) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(2 * n + kind + 1) + (unsigned long long)i * 0xD6E8FEB86659FD93ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (z & 0xFFFFFFFF00000000ull) | i;
}

// ``list`` (optional, (N, S) int32 filled with -1 by the caller) + ``list_n`` ((N,) int32, zeroed): the indices of the sampled elements
// as they are found (unordered: positives and negatives are drawn by different workgroups); S = num_samples.
__global__ __launch_bounds__(1024) void sample_labels_kernel(const signed char* __restrict__ lab, int R, int num_samples, int max_pos, int bg,
                                                            unsigned long long seed, signed char* __restrict__ out, int* __restrict__ counts,
                                                            int* __restrict__ list, int* __restrict__ list_n) {
  __shared__ int hist[256];
  __shared__ int red[32];
  __shared__ int s_bin, s_k;
  const int n = blockIdx.x, kind = blockIdx.y, tid = threadIdx.x;
  const signed char* L = lab + (long long)n * R;
  signed char* O = out + (long long)n * R;
  // members of both kinds (the negatives' quota depends on how many positives exist)
  int cp = 0, cn = 0;
  for (int i = tid; i < R; i += 1024) {
    const int v = L[i];
    cp += (v != -1 && v != bg);
    cn += (v == bg);
  }
  for (int o = 32; o > 0; o >>= 1) { cp += __shfl_xor(cp, o, 64); cn += __shfl_xor(cn, o, 64); }
  if ((tid & 63) == 0) { red[tid >> 6] = cp; red[16 + (tid >> 6)] = cn; }
  __syncthreads();
  cp = 0; cn = 0;
  for (int w = 0; w < 16; ++w) { cp += red[w]; cn += red[16 + w]; }
  const int num_pos = cp < max_pos ? cp : max_pos;
  int num_neg = num_samples - num_pos;
  if (num_neg > cn) num_neg = cn;
  const int members = kind == 0 ? cp : cn;
  const int k = kind == 0 ? num_pos : num_neg;
  if (tid == 0) counts[2 * n + kind] = k;
  // radix select of the k-th smallest key among this kind's members (k < members; otherwise every member is taken)
  unsigned long long prefix = 0, T = ~0ull;
  bool done = false;
  if (k < members && k > 0) {
    // k is tiny against the member count (256 of 1.6 M RPN anchors): ONE pass collects the members whose key falls under a threshold
    // that ~8k of them are expected to pass, and the k-th smallest is selected among those in LDS.  The keys are uniform 64-bit hashes,
    // so the count is Poisson(8k); should it come out below k or above the buffer, the full 8-pass select below runs instead - the
    // threshold changes only the work, never T.
    constexpr int CAP = 4096;
    __shared__ unsigned long long cand[CAP];
    __shared__ int ncand;
    const double frac = 8.0 * (double)k / (double)members;
    if (frac < 0.5 && 8 * k + 64 <= CAP) {
      const unsigned long long thr = (unsigned long long)(frac * 18446744073709551616.0);
      if (tid == 0) ncand = 0;
      __syncthreads();
      for (int i = tid; i < R; i += 1024) {
        const int v = L[i];
        const bool mem = kind == 0 ? (v != -1 && v != bg) : (v == bg);
        if (!mem) continue;
        const unsigned long long key = sample_key(seed, n, kind, (uint32_t)i);
        if (key < thr) {
          const int p = atomicAdd(&ncand, 1);
          if (p < CAP) cand[p] = key;
        }
      }
      __syncthreads();
      const int m = ncand;
      if (m >= k && m <= CAP) {
        int kk = k;
        for (int pass = 7; pass >= 0; --pass) {
          for (int b = tid; b < 256; b += 1024) hist[b] = 0;
          __syncthreads();
          const int shift = pass * 8;
          for (int i = tid; i < m; i += 1024) {
            const unsigned long long key = cand[i];
            if (pass == 7 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(int)((key >> shift) & 255ull)], 1);
          }
          __syncthreads();
          if (tid == 0) {
            int acc = 0, b = 0;
            for (; b < 256; ++b) {
              if (acc + hist[b] >= kk) break;
              acc += hist[b];
            }
            s_bin = b; s_k = kk - acc;
          }
          __syncthreads();
          prefix |= (unsigned long long)s_bin << shift;
          kk = s_k;
          __syncthreads();
        }
        T = prefix;
        done = true;
      } else {
        prefix = 0;
      }
    }
  }
  if (k < members && k > 0 && !done) {
    int kk = k;      // rank (1-based) still to find inside the current prefix
    for (int pass = 7; pass >= 0; --pass) {
      for (int b = tid; b < 256; b += 1024) hist[b] = 0;
      __syncthreads();
      const int shift = pass * 8;
      for (int i = tid; i < R; i += 1024) {
        const int v = L[i];
        const bool mem = kind == 0 ? (v != -1 && v != bg) : (v == bg);
        if (!mem) continue;
        const unsigned long long key = sample_key(seed, n, kind, (uint32_t)i);
        if (pass == 7 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(int)((key >> shift) & 255ull)], 1);
      }
      __syncthreads();
      if (tid == 0) {
        int acc = 0, b = 0;
        for (; b < 256; ++b) {
          if (acc + hist[b] >= kk) break;
          acc += hist[b];
        }
        s_bin = b; s_k = kk - acc;
      }
      __syncthreads();
      prefix |= (unsigned long long)s_bin << shift;
      kk = s_k;
      __syncthreads();
    }
    T = prefix;        // the k-th smallest key itself (keys are unique)
  }
  for (int i = tid; i < R; i += 1024) {
    const int v = L[i];
    const bool pos = (v != -1 && v != bg), neg = (v == bg);
    bool taken = false;
    if (kind == 0) {
      if (!neg) { taken = pos && k > 0 && sample_key(seed, n, 0, (uint32_t)i) <= T; O[i] = taken ? 1 : -1; }     // positives and "other" elements
    } else if (neg) {
      taken = k > 0 && sample_key(seed, n, 1, (uint32_t)i) <= T;
      O[i] = taken ? 0 : -1;
    }
    if (taken && list) {
      const int p = atomicAdd(list_n + n, 1);
      if (p < num_samples) list[(long long)n * num_samples + p] = i;
    }
  }
}

// Indices of the sampled elements of every image, positives (mask == 1) first, then negatives (mask == 0), each in index order, into S
// slots per image (-1 padded); num[n] = how many.  One workgroup per image, ordered compaction by a block-wide scan over contiguous chunks.
__global__ __launch_bounds__(1024) void compact_samples_kernel(const signed char* __restrict__ mask, int R, int S, int* __restrict__ idx,
                                                              int* __restrict__ num) {
  __shared__ int sc[1024];
  const int n = blockIdx.x, tid = threadIdx.x;
  const signed char* M = mask + (long long)n * R;
  int* I = idx + (long long)n * S;
  for (int i = tid; i < S; i += 1024) I[i] = -1;
  const int chunk = (R + 1023) / 1024, lo = tid * chunk, hi = (lo + chunk < R) ? lo + chunk : R;
  int base = 0;
  for (int want = 1; want >= 0; --want) {
    int c = 0;
    for (int i = lo; i < hi; ++i) c += (M[i] == want);
    sc[tid] = c;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {            // inclusive Hillis-Steele scan
      const int v = tid >= o ? sc[tid - o] : 0;
      __syncthreads();
      sc[tid] += v;
      __syncthreads();
    }
    int pos = base + sc[tid] - c;
    const int total = sc[1023];
    __syncthreads();
    for (int i = lo; i < hi; ++i)
      if (M[i] == want) { if (pos < S) I[pos] = i; ++pos; }
    base += total;
  }
  if (tid == 0) num[n] = base < S ? base : S;
}

extern "C" int sod_box2box_get_deltas(const float* src, const float* tgt, long long n, int box_dim, const float* weights, float* deltas,
                                      void* stream) {
  W5 w{};
  if (!src || !tgt || !deltas || n < 0 || !fill_w(w, weights, box_dim)) return SOD_EARG;
  if (n == 0) return SOD_OK;
  SOD_LAUNCH(get_deltas_kernel, dim3(rc_nblk(n, 4096)), dim3(256), 0, (hipStream_t)stream, src, tgt, n, box_dim, w, deltas);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_box2box_apply_deltas(const float* deltas, const float* boxes, long long n, int k, int box_dim, int ld, const float* weights,
                                        float scale_clamp, float* out, void* stream) {
  W5 w{};
  if (!deltas || !boxes || !out || n < 0 || k <= 0 || !fill_w(w, weights, box_dim)) return SOD_EARG;
  if (ld <= 0) ld = k * box_dim;
  if (ld < k * box_dim) return SOD_EARG;
  if (n == 0) return SOD_OK;
  SOD_LAUNCH(apply_deltas_kernel, dim3(rc_nblk(n * k, 4096)), dim3(256), 0, (hipStream_t)stream, deltas, boxes, n, k, box_dim, ld, w, scale_clamp, out);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_bce_logits_loss_fwd(const float* logits, const signed char* labels, long long n, float* sum_out, float* ws, void* stream) {
  if (!logits || !labels || !sum_out || !ws || n < 0) return SOD_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int g = rc_nblk(n);
  SOD_LAUNCH(bce_kernel<false>, dim3(g), dim3(256), 0, st, logits, labels, n, ws, nullptr, 0.f, nullptr);
  SOD_LAUNCH(rc_finish, dim3(1), dim3(256), 0, st, ws, g, 1, sum_out);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_bce_logits_loss_bwd(const float* logits, const signed char* labels, long long n, const float* grad_scale, float scale_mul,
                                       float* dlogits, void* stream) {
  if (!logits || !labels || !grad_scale || !dlogits || n < 0) return SOD_EARG;
  SOD_LAUNCH(bce_kernel<true>, dim3(rc_nblk(n, 4096)), dim3(256), 0, (hipStream_t)stream, logits, labels, n, nullptr, grad_scale, scale_mul, dlogits);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_bce_logits_soft_fwd(const float* logits, const float* targets, const int* labels, int bg_label, long long n, float* sum_out,
                                       float* ws, void* stream) {
  if (!logits || !targets || !labels || !sum_out || !ws || n < 0) return SOD_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int g = rc_nblk(n);
  SOD_LAUNCH(bce_soft_kernel<false>, dim3(g), dim3(256), 0, st, logits, targets, labels, bg_label, n, ws, nullptr, 0.f, nullptr);
  SOD_LAUNCH(rc_finish, dim3(1), dim3(256), 0, st, ws, g, 1, sum_out);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_bce_logits_soft_bwd(const float* logits, const float* targets, const int* labels, int bg_label, long long n,
                                       const float* grad_scale, float scale_mul, float* dlogits, void* stream) {
  if (!logits || !targets || !labels || !grad_scale || !dlogits || n < 0) return SOD_EARG;
  SOD_LAUNCH(bce_soft_kernel<true>, dim3(rc_nblk(n, 4096)), dim3(256), 0, (hipStream_t)stream, logits, targets, labels, bg_label, n, nullptr,
             grad_scale, scale_mul, dlogits);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_rpn_loc_loss_fwd(const float* pred, const float* target, const signed char* labels, long long n, int box_dim, float beta,
                                    float* sum_out, float* ws, void* stream) {
  if (!pred || !target || !labels || !sum_out || !ws || n < 0 || box_dim <= 0) return SOD_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int g = rc_nblk(n * box_dim);
  SOD_LAUNCH(loc_kernel<false>, dim3(g), dim3(256), 0, st, pred, target, labels, n, box_dim, beta, ws, nullptr, 0.f, nullptr);
  SOD_LAUNCH(rc_finish, dim3(1), dim3(256), 0, st, ws, g, 1, sum_out);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_rpn_loc_loss_bwd(const float* pred, const float* target, const signed char* labels, long long n, int box_dim, float beta,
                                    const float* grad_scale, float scale_mul, float* dpred, void* stream) {
  if (!pred || !target || !labels || !grad_scale || !dpred || n < 0 || box_dim <= 0) return SOD_EARG;
  SOD_LAUNCH(loc_kernel<true>, dim3(rc_nblk(n * box_dim, 4096)), dim3(256), 0, (hipStream_t)stream, pred, target, labels, n, box_dim, beta, nullptr,
             grad_scale, scale_mul, dpred);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_softmax_ce_fwd(const float* scores, const int* labels, int R, int C, int ld, float* sum_out, float* ws, void* stream) {
  if (!scores || !labels || !sum_out || !ws || R < 0 || C <= 0 || ld < C) return SOD_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int g = rc_nblk((long long)R * 64);
  SOD_LAUNCH(ce_kernel<false>, dim3(g), dim3(256), 0, st, scores, labels, R, C, ld, ws, nullptr, 0.f, nullptr);
  SOD_LAUNCH(rc_finish, dim3(1), dim3(256), 0, st, ws, g, 1, sum_out);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_softmax_ce_bwd(const float* scores, const int* labels, int R, int C, int ld, const float* grad_scale, float scale_mul,
                                  float* dscores, void* stream) {
  if (!scores || !labels || !grad_scale || !dscores || R < 0 || C <= 0 || ld < C) return SOD_EARG;
  if (R == 0) return SOD_OK;
  SOD_LAUNCH(ce_kernel<true>, dim3(rc_nblk((long long)R * 64, 4096)), dim3(256), 0, (hipStream_t)stream, scores, labels, R, C, ld, nullptr, grad_scale,
             scale_mul, dscores);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_fastrcnn_box_loss_fwd(const float* pred, const int* gt_classes, const float* gt_deltas, int R, int K, int box_dim, int ld,
                                         float beta, float* sum_out, float* ws, void* stream) {
  if (!pred || !gt_classes || !gt_deltas || !sum_out || !ws || R < 0 || K <= 0 || box_dim <= 0 || ld < K * box_dim) return SOD_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int g = rc_nblk((long long)R * box_dim);
  SOD_LAUNCH(frcnn_box_kernel<false>, dim3(g), dim3(256), 0, st, pred, gt_classes, gt_deltas, R, K, box_dim, ld, beta, ws, nullptr, 0.f, nullptr);
  SOD_LAUNCH(rc_finish, dim3(1), dim3(256), 0, st, ws, g, 1, sum_out);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_fastrcnn_box_loss_bwd(const float* pred, const int* gt_classes, const float* gt_deltas, int R, int K, int box_dim, int ld,
                                         float beta, const float* grad_scale, float scale_mul, float* dpred, void* stream) {
  if (!pred || !gt_classes || !gt_deltas || !grad_scale || !dpred || R < 0 || K <= 0 || box_dim <= 0 || ld < K * box_dim) return SOD_EARG;
  if (R == 0) return SOD_OK;
  SOD_LAUNCH(frcnn_box_kernel<true>, dim3(rc_nblk((long long)R * ld, 4096)), dim3(256), 0, (hipStream_t)stream, pred, gt_classes, gt_deltas, R, K,
             box_dim, ld, beta, nullptr, grad_scale, scale_mul, dpred);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_sample_labels(const signed char* labels, int N, int R, int num_samples, float positive_fraction, int bg_label,
                                 unsigned long long seed, signed char* out, int* counts, void* stream) {
  if (!labels || !out || !counts || N <= 0 || R <= 0 || num_samples <= 0 || !(positive_fraction >= 0.f && positive_fraction <= 1.f)) return SOD_EARG;
  if (N > 65535) return SOD_ESIZE;
  const int max_pos = (int)(num_samples * positive_fraction);
  SOD_LAUNCH(sample_labels_kernel, dim3(N, 2), dim3(1024), 0, (hipStream_t)stream, labels, R, num_samples, max_pos, bg_label, seed, out, counts,
             (int*)nullptr, (int*)nullptr);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

// The same draw, and the indices of the drawn elements of every image as a list: idx (N, num_samples) int32, -1 padded, UNORDERED (sort the
// rows for a run-to-run stable order); list_n (N,) int32 scratch.  For row counts where sod_compact_samples' scan over every element costs
// more than the draw itself (RPN: 1.6 M anchors per image for 256 samples).
extern "C" int sod_sample_labels_list(const signed char* labels, int N, int R, int num_samples, float positive_fraction, int bg_label,
                                      unsigned long long seed, signed char* out, int* counts, int* idx, int* list_n, void* stream) {
  if (!labels || !out || !counts || !idx || !list_n || N <= 0 || R <= 0 || num_samples <= 0 || !(positive_fraction >= 0.f && positive_fraction <= 1.f)) return SOD_EARG;
  if (N > 65535) return SOD_ESIZE;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(idx, 0xff, sizeof(int) * (size_t)N * num_samples, st);
  if (e == hipSuccess) e = hipMemsetAsync(list_n, 0, sizeof(int) * (size_t)N, st);
  if (e != hipSuccess) return (int)e;
  const int max_pos = (int)(num_samples * positive_fraction);
  SOD_LAUNCH(sample_labels_kernel, dim3(N, 2), dim3(1024), 0, st, labels, R, num_samples, max_pos, bg_label, seed, out, counts, idx, list_n);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

// RPN losses on the SAMPLED anchors only (detectron2 RPN.losses sums over the <= BATCH_SIZE_PER_IMAGE sampled anchors of an image; the
// other ~1.6 M rows of the dense (N, R) formulation are label -1).  Anchor r of the concatenated (level, h, w, a) order lives in level l
// with start[l] <= r < start[l + 1], at pixel (r - start[l]) / A, anchor (r - start[l]) % A of that level's padded NHWC head output.
constexpr int RPN_MAXLEV = 8;
struct RpnLevelPtrs {
  const float* logit[RPN_MAXLEV]; const float* delta[RPN_MAXLEV];     // gather sources (forward)
  float* dlogit[RPN_MAXLEV]; float* ddelta[RPN_MAXLEV];               // scatter destinations (backward), zero-initialised by the caller
  int start[RPN_MAXLEV + 1], hw[RPN_MAXLEV], pl[RPN_MAXLEV], pd[RPN_MAXLEV];
  int nlev;
};

template <bool SCATTER>
__global__ __launch_bounds__(256) void rpn_sampled_rows_kernel(const RpnLevelPtrs L, const int* __restrict__ idx, int N, int S, int A, int D,
                                                               float* __restrict__ row_logit, float* __restrict__ row_delta) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * S) return;
  const int n = i / S, r = idx[i];
  if (r < 0) {                       // padding slot of an image with fewer samples
    if (!SCATTER) { row_logit[i] = 0.f; for (int d = 0; d < D; ++d) row_delta[(long long)i * D + d] = 0.f; }
    return;
  }
  int l = 0;
  for (int k = 1; k < L.nlev; ++k)
    if (r >= L.start[k]) l = k;
  const int local = r - L.start[l], pix = local / A, a = local - pix * A;
  const long long row = (long long)n * L.hw[l] + pix;
  if (SCATTER) {
    L.dlogit[l][row * L.pl[l] + a] = row_logit[i];
    for (int d = 0; d < D; ++d) L.ddelta[l][row * L.pd[l] + a * D + d] = row_delta[(long long)i * D + d];
  } else {
    row_logit[i] = L.logit[l][row * L.pl[l] + a];
    for (int d = 0; d < D; ++d) row_delta[(long long)i * D + d] = L.delta[l][row * L.pd[l] + a * D + d];
  }
}

static int rpn_levels_fill(RpnLevelPtrs& L, int nlev, const void* const* logits, const void* const* deltas, const int* hw, const int* pl, const int* pd,
                           int A, bool scatter) {
  if (nlev <= 0 || nlev > RPN_MAXLEV || !logits || !deltas || !hw || !pl || !pd || A <= 0) return SOD_EARG;
  L.nlev = nlev;
  int start = 0;
  for (int l = 0; l < nlev; ++l) {
    if (!logits[l] || !deltas[l] || hw[l] <= 0 || pl[l] < A || pd[l] <= 0) return SOD_EARG;
    if (scatter) { L.dlogit[l] = (float*)logits[l]; L.ddelta[l] = (float*)deltas[l]; }
    else { L.logit[l] = (const float*)logits[l]; L.delta[l] = (const float*)deltas[l]; }
    L.start[l] = start; L.hw[l] = hw[l]; L.pl[l] = pl[l]; L.pd[l] = pd[l];
    start += hw[l] * A;
  }
  L.start[nlev] = start;
  return SOD_OK;
}

extern "C" int sod_rpn_gather_sampled(int nlev, const void* const* logits, const void* const* deltas, const int* hw, const int* logit_pitch,
                                      const int* delta_pitch, const int* idx, int N, int S, int A, int D, float* row_logits, float* row_deltas,
                                      void* stream) {
  RpnLevelPtrs L{};
  int rc = rpn_levels_fill(L, nlev, logits, deltas, hw, logit_pitch, delta_pitch, A, false);
  if (rc || !idx || !row_logits || !row_deltas || N <= 0 || S <= 0 || D <= 0) return rc ? rc : SOD_EARG;
  for (int l = 0; l < nlev; ++l) if (delta_pitch[l] < A * D) return SOD_EARG;
  SOD_LAUNCH(rpn_sampled_rows_kernel<false>, dim3((N * S + 255) / 256), dim3(256), 0, (hipStream_t)stream, L, idx, N, S, A, D, row_logits, row_deltas);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_rpn_scatter_sampled(int nlev, void* const* dlogits, void* const* ddeltas, const int* hw, const int* logit_pitch,
                                       const int* delta_pitch, const int* idx, int N, int S, int A, int D, const float* row_dlogits,
                                       const float* row_ddeltas, void* stream) {
  RpnLevelPtrs L{};
  int rc = rpn_levels_fill(L, nlev, (const void* const*)dlogits, (const void* const*)ddeltas, hw, logit_pitch, delta_pitch, A, true);
  if (rc || !idx || !row_dlogits || !row_ddeltas || N <= 0 || S <= 0 || D <= 0) return rc ? rc : SOD_EARG;
  for (int l = 0; l < nlev; ++l) if (delta_pitch[l] < A * D) return SOD_EARG;
  SOD_LAUNCH(rpn_sampled_rows_kernel<true>, dim3((N * S + 255) / 256), dim3(256), 0, (hipStream_t)stream, L, idx, N, S, A, D,
             const_cast<float*>(row_dlogits), const_cast<float*>(row_ddeltas));
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_compact_samples(const signed char* mask, int N, int R, int S, int* idx, int* num, void* stream) {
  if (!mask || !idx || !num || N <= 0 || R <= 0 || S <= 0) return SOD_EARG;
  SOD_LAUNCH(compact_samples_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, mask, R, S, idx, num);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}
