// RepPoints detector support kernels (slender_det/modeling/meta_arch/reppoints/rpd.py, matchers/rep_matcher.py,
// structures/points.py).  All of this is HBM/latency-bound integer and fp32 work over (N, X = sum of level pixels) rows:
//   * dcn offsets from point offsets (rpd.py:621-635): xy -> yx swap, minus the y-major kernel grid, gradient multiplier;
//   * points2bbox "minmax" (rpd.py:221-249) with the arg indices kept for the backward scatter;
//   * the three init-box matchers (rep_matcher.py:9-101, :199-223, :226-248) — one workgroup per image, no (P, M) matrices;
//   * classification / refine labels after pairwise_iou + Matcher (rpd.py:311-323);
//   * stride-normalised smooth-L1 over labelled rows (rpd.py:383-396) and the loss finalisation with the EMA normaliser
//     kept on the device (rpd.py:364-381 calls .item()).
#include "common.h"
#include "../../include/slender_hip.h"
#include <math.h>

namespace {

constexpr int RP_RED = 1024;

inline int rp_nblk(long long n, int cap = RP_RED) {
  long long g = (n + 255) / 256;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// ------------------------------------------------------------------------------------------------ dcn offsets
// out[r, 2k] = scale * in[r, 2k+1] - by[k] ; out[r, 2k+1] = scale * in[r, 2k] - bx[k]   (by = k / ks - pad, bx = k % ks - pad)
// Forward: scale 1, subtract_base 1.  Backward is the same permutation with scale = gradient_mul and no base.
__global__ __launch_bounds__(256) void rp_dcn_offset_kernel(const float* __restrict__ in, float* __restrict__ out, long long rows, int ld,
                                                            int npts, int ks, float scale, int sub_base, int flip) {
  const long long total = rows * ld;
  const int pad = (ks - 1) / 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int j = (int)(i % ld);
    float v = 0.f;
    if (j < 2 * npts) {
      v = scale * in[i - j + (flip ? (j ^ 1) : j)];
      if (sub_base) {
        const int k = j >> 1;
        v -= (float)(((j & 1) == 0 ? k / ks : k % ks) - pad);
      }
    }
    out[i] = v;
  }
}

// ------------------------------------------------------------------------------------------------ points2bbox (minmax)
struct P2BArgs {
  const float* pts;     // (N, H*W, ld): channel 2k = x offset, 2k+1 = y offset of point k (rpd.py:239-243)
  const float* add;     // optional second addend with the same layout (offsets_refine + offsets_init.detach(), rpd.py:642)
  float* boxes;         // level slice of (N, X, 4): boxes[n*box_img_stride + p*4 + c]
  unsigned* arg;        // level slice of (N, X): byte c = index of the point that produced box coordinate c
  long long box_img_stride, arg_img_stride;
  int N, H, W, ld, npts;
  float grid_stride, pt_stride;
};

__global__ __launch_bounds__(256) void p2b_fwd_kernel(const P2BArgs a) {
  const long long HW = (long long)a.H * a.W, total = HW * a.N;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / HW, p = i - n * HW;
    const int h = (int)(p / a.W), w = (int)(p - (long long)h * a.W);
    const float cx = (float)w * a.grid_stride, cy = (float)h * a.grid_stride;
    const float* r = a.pts + i * a.ld;
    const float* q = a.add ? a.add + i * a.ld : nullptr;
    float xmin = 0, xmax = 0, ymin = 0, ymax = 0;
    unsigned ixmin = 0, ixmax = 0, iymin = 0, iymax = 0;
    for (int k = 0; k < a.npts; ++k) {
      float vx = r[2 * k], vy = r[2 * k + 1];
      if (q) { vx += q[2 * k]; vy += q[2 * k + 1]; }
      const float x = vx * a.pt_stride + cx, y = vy * a.pt_stride + cy;
      if (k == 0) { xmin = xmax = x; ymin = ymax = y; }
      else {
        if (x < xmin) { xmin = x; ixmin = k; }
        if (x > xmax) { xmax = x; ixmax = k; }
        if (y < ymin) { ymin = y; iymin = k; }
        if (y > ymax) { ymax = y; iymax = k; }
      }
    }
    f32x4_t b = {xmin, ymin, xmax, ymax};
    *reinterpret_cast<f32x4_t*>(a.boxes + n * a.box_img_stride + p * 4) = b;
    if (a.arg) a.arg[n * a.arg_img_stride + p] = ixmin | (iymin << 8) | (ixmax << 16) | (iymax << 24);
  }
}

__global__ __launch_bounds__(256) void p2b_bwd_kernel(const P2BArgs a, const float* __restrict__ dboxes, float* __restrict__ dpts32,
                                                      __bf16* __restrict__ dpts16) {
  const long long HW = (long long)a.H * a.W, total = HW * a.N;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / HW, p = i - n * HW;
    const f32x4_t d = *reinterpret_cast<const f32x4_t*>(dboxes + n * a.box_img_stride + p * 4);
    const unsigned ar = a.arg[n * a.arg_img_stride + p];
    const int j0 = 2 * (int)(ar & 255), j1 = 2 * (int)((ar >> 8) & 255) + 1, j2 = 2 * (int)((ar >> 16) & 255), j3 = 2 * (int)(ar >> 24) + 1;
    for (int j = 0; j < a.ld; ++j) {
      float v = 0.f;
      if (j == j0) v += d[0];
      if (j == j1) v += d[1];
      if (j == j2) v += d[2];
      if (j == j3) v += d[3];
      v *= a.pt_stride;
      if (dpts32) dpts32[i * a.ld + j] = v;
      if (dpts16) dpts16[i * a.ld + j] = (__bf16)v;
    }
  }
}

// "moment" transform (meta/heads/pointset_head.py:328-343): box = mean -+ std * exp(moment_transfer), torch.std = unbiased.
// The backward recomputes the moments from the points (nothing but the boxes is stored by the forward).
__device__ __forceinline__ void p2b_moments(const P2BArgs& a, long long i, float cx, float cy, float& mx, float& my, float& sx, float& sy) {
  const float* r = a.pts + i * a.ld;
  const float* q = a.add ? a.add + i * a.ld : nullptr;
  float s1x = 0.f, s1y = 0.f;
  for (int k = 0; k < a.npts; ++k) {
    float vx = r[2 * k], vy = r[2 * k + 1];
    if (q) { vx += q[2 * k]; vy += q[2 * k + 1]; }
    s1x += vx * a.pt_stride + cx; s1y += vy * a.pt_stride + cy;
  }
  mx = s1x / (float)a.npts; my = s1y / (float)a.npts;
  float s2x = 0.f, s2y = 0.f;
  for (int k = 0; k < a.npts; ++k) {
    float vx = r[2 * k], vy = r[2 * k + 1];
    if (q) { vx += q[2 * k]; vy += q[2 * k + 1]; }
    const float dx = vx * a.pt_stride + cx - mx, dy = vy * a.pt_stride + cy - my;
    s2x += dx * dx; s2y += dy * dy;
  }
  sx = sqrtf(s2x / (float)(a.npts - 1)); sy = sqrtf(s2y / (float)(a.npts - 1));
}

__global__ __launch_bounds__(256) void p2b_moment_fwd_kernel(const P2BArgs a, const float* __restrict__ mt) {
  const long long HW = (long long)a.H * a.W, total = HW * a.N;
  const float ew = expf(mt[0]), eh = expf(mt[1]);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / HW, p = i - n * HW;
    const int h = (int)(p / a.W), w = (int)(p - (long long)h * a.W);
    float mx, my, sx, sy;
    p2b_moments(a, i, (float)w * a.grid_stride, (float)h * a.grid_stride, mx, my, sx, sy);
    const f32x4_t b = {mx - sx * ew, my - sy * eh, mx + sx * ew, my + sy * eh};
    *reinterpret_cast<f32x4_t*>(a.boxes + n * a.box_img_stride + p * 4) = b;
  }
}

__global__ __launch_bounds__(256) void p2b_moment_bwd_kernel(const P2BArgs a, const float* __restrict__ dboxes, const float* __restrict__ mt,
                                                             float moment_mul, float* __restrict__ dpts32, __bf16* __restrict__ dpts16,
                                                             float* __restrict__ dmt) {
  __shared__ float red[4];
  const long long HW = (long long)a.H * a.W, total = HW * a.N;
  const float ew = expf(mt[0]), eh = expf(mt[1]);
  float gw = 0.f, gh = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / HW, p = i - n * HW;
    const int h = (int)(p / a.W), w = (int)(p - (long long)h * a.W);
    const float cx = (float)w * a.grid_stride, cy = (float)h * a.grid_stride;
    float mx, my, sx, sy;
    p2b_moments(a, i, cx, cy, mx, my, sx, sy);
    const f32x4_t d = *reinterpret_cast<const f32x4_t*>(dboxes + n * a.box_img_stride + p * 4);
    const float dmx = d[0] + d[2], dmy = d[1] + d[3], dhw = d[2] - d[0], dhh = d[3] - d[1];
    gw += dhw * sx * ew; gh += dhh * sy * eh;
    const float inv_n = 1.f / (float)a.npts;
    const float kx = sx > 0.f ? dhw * ew / ((float)(a.npts - 1) * sx) : 0.f, ky = sy > 0.f ? dhh * eh / ((float)(a.npts - 1) * sy) : 0.f;
    const float* r = a.pts + i * a.ld;
    const float* q = a.add ? a.add + i * a.ld : nullptr;
    for (int j = 0; j < a.ld; ++j) {
      float v = 0.f;
      if (j < 2 * a.npts) {
        float pv = r[j];
        if (q) pv += q[j];
        if (j & 1) v = dmy * inv_n + ky * (pv * a.pt_stride + cy - my);
        else       v = dmx * inv_n + kx * (pv * a.pt_stride + cx - mx);
        v *= a.pt_stride;
      }
      if (dpts32) dpts32[i * a.ld + j] = v;
      if (dpts16) dpts16[i * a.ld + j] = (__bf16)v;
    }
  }
  if (dmt) {      // d(moment_transfer): grad_mul scales the gradient by moment_mul (pointset_head.py:333-334)
    gw = block_sum_256(gw, red);
    gh = block_sum_256(gh, red);
    if (threadIdx.x == 0) { atomicAdd(dmt, gw * moment_mul); atomicAdd(dmt + 1, gh * moment_mul); }
  }
}

// ------------------------------------------------------------------------------------------------ init-box matchers
struct RpMatchArgs {
  const float* centers;   // (X, 2) x, y
  const float* strides;   // (X,)
  const int* lvl_start;   // (nl + 1,)
  const float* boxes;     // (sum G, 4)
  const int* box_off;     // (N + 1,)
  int* obj;               // (N, X)
  float* blab;            // (N, X, 4)
  int X, nl, mode;
  float scale;
};

__device__ __forceinline__ void block_argmin(float& d, int& idx, float* sd, int* si) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float od = __shfl_xor(d, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (od < d || (od == d && oi < idx)) { d = od; idx = oi; }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { sd[wave] = d; si[wave] = idx; }
  __syncthreads();
  d = sd[0]; idx = si[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (sd[w] < d || (sd[w] == d && si[w] < idx)) { d = sd[w]; idx = si[w]; }
}

__device__ __forceinline__ float rp_dist(float px, float py, float cx, float cy, float w, float h) {
  const float dx = (px - cx) / w, dy = (py - cy) / h;
  return sqrtf(dx * dx + dy * dy);
}

// stride_match (structures/points.py:31-45): the FPN stride a box belongs to
__device__ __forceinline__ float rp_box_stride(float w, float h, float smin, float smax) {
  const int e = (int)((log2f(w) + log2f(h)) / 2.f);
  return fminf(fmaxf(exp2f((float)e), smin), smax);
}

__global__ __launch_bounds__(256) void rp_match_kernel(const RpMatchArgs a) {
  extern __shared__ float smem[];
  __shared__ float sd[4];
  __shared__ int si[4];
  __shared__ float scnt[4];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int g0 = a.box_off[n], G = a.box_off[n + 1] - g0;
  float* gd = smem;                  // (G) best distance of each gt
  int* gi = (int*)(smem + G);        // (G) its point
  float* gs = smem + 2 * G;          // (G) box stride (modes 1, 2)
  int* obj = a.obj + (long long)n * a.X;
  float* blab = a.blab + (long long)n * a.X * 4;
  const float* boxes = a.boxes + (long long)g0 * 4;
  for (int p = tid; p < a.X; p += 256) {
    obj[p] = 0;
    *reinterpret_cast<f32x4_t*>(blab + (long long)p * 4) = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  float smin = a.strides[a.lvl_start[0]], smax = smin;
  for (int l = 1; l < a.nl; ++l) { const float s = a.strides[a.lvl_start[l]]; smin = fminf(smin, s); smax = fmaxf(smax, s); }
  int mode = a.mode;
  __syncthreads();

  if (mode == 2) {   // inside_match (rep_matcher.py:226-248)
    float cnt = 0.f;
    for (int p = tid; p < a.X; p += 256) {
      const float px = a.centers[2 * p], py = a.centers[2 * p + 1], s = a.strides[p];
      bool ins = false;
      float best = INFINITY; int bg = 0;
      for (int g = 0; g < G; ++g) {
        const f32x4_t b = *reinterpret_cast<const f32x4_t*>(boxes + g * 4);
        const float w = b[2] - b[0], h = b[3] - b[1];
        const float d = rp_dist(px, py, (b[0] + b[2]) * 0.5f, (b[1] + b[3]) * 0.5f, w, h);
        if (d < best) { best = d; bg = g; }
        const bool in = (px + s >= b[0]) && (py + s >= b[1]) && (px <= b[2]) && (py <= b[3]);
        ins = ins || (in && s == rp_box_stride(w, h, smin, smax));
      }
      if (ins) { obj[p] = 1; cnt += 1.f; }
      if (G > 0) *reinterpret_cast<f32x4_t*>(blab + (long long)p * 4) = *reinterpret_cast<const f32x4_t*>(boxes + bg * 4);
    }
    cnt = block_sum_256(cnt, scnt);
    if (cnt > 0.f) return;
    __syncthreads();
    for (int p = tid; p < a.X; p += 256) {   // nothing inside: fall back to nearest_point_match (rep_matcher.py:242-243)
      obj[p] = 0;
      *reinterpret_cast<f32x4_t*>(blab + (long long)p * 4) = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    mode = 1;
    __syncthreads();
  }

  if (mode == 0) {   // rep_points_match (rep_matcher.py:9-101), pos_num = 1
    int lmin = (int)log2f(smin), lmax = (int)log2f(smax);
    for (int g = 0; g < G; ++g) {
      const f32x4_t b = *reinterpret_cast<const f32x4_t*>(boxes + g * 4);
      const float cx = (b[0] + b[2]) * 0.5f, cy = (b[1] + b[3]) * 0.5f;
      const float w = fmaxf(b[2] - b[0], 1e-6f), h = fmaxf(b[3] - b[1], 1e-6f);
      int lvl = (int)((log2f(w / a.scale) + log2f(h / a.scale)) / 2.f);
      lvl = min(max(lvl, lmin), lmax);
      int li = -1;
      for (int l = 0; l < a.nl; ++l)
        if ((int)log2f(a.strides[a.lvl_start[l]]) == lvl) li = l;
      float best = INFINITY; int bi = 0x7fffffff;
      if (li >= 0) {
        for (int p = a.lvl_start[li] + tid; p < a.lvl_start[li + 1]; p += 256) {
          const float d = rp_dist(a.centers[2 * p], a.centers[2 * p + 1], cx, cy, w, h);
          if (d < best) { best = d; bi = p; }
        }
      }
      block_argmin(best, bi, sd, si);
      if (tid == 0) { gd[g] = best; gi[g] = (bi == 0x7fffffff) ? -1 : bi; }
    }
    __syncthreads();
    for (int g = tid; g < G; g += 256) {
      const int p = gi[g];
      if (p < 0) continue;
      bool win = true;   // sequential rule (:74-88): strictly smaller distance replaces, so the earliest minimum keeps the point
      for (int o = 0; o < G; ++o)
        if (o != g && gi[o] == p && (gd[o] < gd[g] || (gd[o] == gd[g] && o < g))) win = false;
      if (win) {
        obj[p] = 1;
        *reinterpret_cast<f32x4_t*>(blab + (long long)p * 4) = *reinterpret_cast<const f32x4_t*>(boxes + g * 4);
      }
    }
    return;
  }

  // nearest_point_match (rep_matcher.py:199-223)
  for (int g = 0; g < G; ++g) {
    const f32x4_t b = *reinterpret_cast<const f32x4_t*>(boxes + g * 4);
    const float cx = (b[0] + b[2]) * 0.5f, cy = (b[1] + b[3]) * 0.5f, w = b[2] - b[0], h = b[3] - b[1];
    const float bs = rp_box_stride(w, h, smin, smax);
    float best = INFINITY; int bi = 0x7fffffff;
    for (int p = tid; p < a.X; p += 256) {
      const float d = rp_dist(a.centers[2 * p], a.centers[2 * p + 1], cx, cy, w, h) + (a.strides[p] == bs ? 0.f : 1e5f);
      if (d < best) { best = d; bi = p; }
    }
    block_argmin(best, bi, sd, si);
    if (tid == 0) { gd[g] = best; gi[g] = (bi == 0x7fffffff) ? -1 : bi; gs[g] = bs; }
  }
  __syncthreads();
  for (int g = tid; g < G; g += 256) {
    const int p = gi[g];
    if (p < 0) continue;
    const float px = a.centers[2 * p], py = a.centers[2 * p + 1], s = a.strides[p];
    float pmin = INFINITY;   // nearest object of this point
    for (int o = 0; o < G; ++o) {
      const f32x4_t b = *reinterpret_cast<const f32x4_t*>(boxes + o * 4);
      const float d = rp_dist(px, py, (b[0] + b[2]) * 0.5f, (b[1] + b[3]) * 0.5f, b[2] - b[0], b[3] - b[1]) + (s == gs[o] ? 0.f : 1e5f);
      pmin = fminf(pmin, d);
    }
    const bool lost = pmin < gd[g];
    gd[g] = lost ? -1.f : gd[g];   // mark; read by the tie rule below after the barrier
  }
  __syncthreads();
  for (int g = tid; g < G; g += 256) {
    const int p = gi[g];
    if (p < 0 || gd[g] < 0.f) continue;
    bool win = true;   // the reference's loop (:217-222) lets the LAST surviving box overwrite a shared point
    for (int o = g + 1; o < G; ++o)
      if (gi[o] == p && gd[o] >= 0.f) win = false;
    if (win) {
      obj[p] = 1;
      *reinterpret_cast<f32x4_t*>(blab + (long long)p * 4) = *reinterpret_cast<const f32x4_t*>(boxes + g * 4);
    }
  }
}

// ------------------------------------------------------------------------------------------------ labels (rpd.py:303-323)
__global__ __launch_bounds__(256) void rp_labels_kernel(const int* __restrict__ matches, const signed char* __restrict__ mlab,
                                                        const float* __restrict__ boxes, const int* __restrict__ classes,
                                                        const int* __restrict__ box_off, const float* __restrict__ centers,
                                                        const float* __restrict__ image_hw, int N, int X, int K,
                                                        int* __restrict__ cls, float* __restrict__ rbox, int* __restrict__ obj) {
  const long long total = (long long)N * X;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int n = (int)(i / X), p = (int)(i - (long long)n * X);
    const int g0 = box_off[n], G = box_off[n + 1] - g0;
    const bool invalid = centers[2 * p] >= image_hw[2 * n + 1] || centers[2 * p + 1] >= image_hw[2 * n];
    int c = K;
    f32x4_t b = {0.f, 0.f, 0.f, 0.f};
    if (G > 0) {
      const int m = g0 + matches[i];
      c = classes[m];                       // matcher label -1 ("ignore") keeps the gt class: the reference only rewrites label 0
      if (mlab[i] == 0) c = K;
      b = *reinterpret_cast<const f32x4_t*>(boxes + (long long)m * 4);
    }
    if (invalid) { c = -1; if (obj) obj[i] = 0; }
    cls[i] = c;
    *reinterpret_cast<f32x4_t*>(rbox + i * 4) = b;
  }
}

// ------------------------------------------------------------------------------------------------ box loss (rpd.py:383-396)
struct RpBoxArgs {
  const float* pred;     // (N, X, 4)
  const float* target;   // (N, X, 4)
  const int* labels;     // (N, X): bg_label < 0 -> row selected when label > 0; else when 0 <= label != bg_label
  const float* strides;  // (X,)
  int N, X, bg;
  float beta;
};

template <bool BWD>
__global__ __launch_bounds__(256) void rp_box_loss_kernel(const RpBoxArgs a, float* __restrict__ part, const float* __restrict__ gnum,
                                                          const float* __restrict__ gden, float den_min, float mul, float* __restrict__ dpred) {
  __shared__ float red[4];
  float acc = 0.f, cnt = 0.f;
  const float sc = BWD ? mul * gnum[0] / fmaxf(gden[0], den_min) : 0.f;
  const long long total = (long long)a.N * a.X;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int lab = a.labels[i];
    const bool sel = a.bg < 0 ? lab > 0 : (lab >= 0 && lab != a.bg);
    f32x4_t g = {0.f, 0.f, 0.f, 0.f};
    if (sel) {
      cnt += 1.f;
      const float nrm = a.strides[i % a.X] * 4.f;
      const f32x4_t p = *reinterpret_cast<const f32x4_t*>(a.pred + i * 4), t = *reinterpret_cast<const f32x4_t*>(a.target + i * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = p[e] / nrm - t[e] / nrm, ad = fabsf(d);
        if (a.beta < 1e-5f) { acc += ad; g[e] = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
        else if (ad < a.beta) { acc += 0.5f * d * d / a.beta; g[e] = d / a.beta; }
        else { acc += ad - 0.5f * a.beta; g[e] = d > 0.f ? 1.f : -1.f; }
        g[e] *= sc / nrm;
      }
    }
    if (BWD) *reinterpret_cast<f32x4_t*>(dpred + i * 4) = g;
  }
  if (!BWD) {
    acc = block_sum_256(acc, red);
    cnt = block_sum_256(cnt, red);
    if (threadIdx.x == 0) { part[blockIdx.x] = acc; part[RP_RED + blockIdx.x] = cnt; }
  }
}

__global__ void rp_box_finish_kernel(const float* __restrict__ part, int nb, float* __restrict__ sums) {
  __shared__ float red[4];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) { a += part[i]; b += part[RP_RED + i]; }
  a = block_sum_256(a, red);
  b = block_sum_256(b, red);
  if (threadIdx.x == 0) { sums[0] = a; sums[1] = b; }
}

__global__ void rp_finalize_kernel(const float* __restrict__ focal_sum, const float* __restrict__ init2, const float* __restrict__ refine2,
                                   float* __restrict__ normalizer, float momentum, float inv_images, float init_weight, float* __restrict__ out3) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float nrm = momentum * normalizer[0] + (1.f - momentum) * refine2[1] * inv_images;   // rpd.py:373-376
    normalizer[0] = nrm;
    const float den = fmaxf(1.f, nrm);
    out3[0] = focal_sum[0] / den;
    out3[1] = init2[0] / fmaxf(1.f, init2[1]) * init_weight;
    out3[2] = refine2[0] / den;
  }
}

}  // namespace

extern "C" int sod_reppoints_dcn_offset(const float* pts, float* out, long long rows, int ld, int num_points, float scale, int subtract_base,
                                        int flip_xy, void* stream) {
  if (!pts || !out || rows < 0 || num_points <= 0 || ld < 2 * num_points) return SOD_EARG;
  int ks = 1;
  while (ks * ks < num_points) ++ks;
  if (ks * ks != num_points || !(ks & 1)) return SOD_EARG;
  if (rows == 0) return SOD_OK;
  SOD_LAUNCH(rp_dcn_offset_kernel, dim3(rp_nblk(rows * ld, 8192)), dim3(256), 0, (hipStream_t)stream, pts, out, rows, ld, num_points, ks, scale,
             subtract_base, flip_xy);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

static int p2b_fill(P2BArgs& a, int ld, int N, int H, int W, float grid_stride, float point_stride, int num_points, long long box_img_stride,
                    long long arg_img_stride) {
  if (N <= 0 || H <= 0 || W <= 0 || num_points <= 0 || num_points > 255 || ld < 2 * num_points) return SOD_EARG;
  a.N = N; a.H = H; a.W = W; a.ld = ld; a.npts = num_points; a.grid_stride = grid_stride; a.pt_stride = point_stride;
  a.box_img_stride = box_img_stride > 0 ? box_img_stride : (long long)H * W * 4;
  a.arg_img_stride = arg_img_stride > 0 ? arg_img_stride : (long long)H * W;
  if (a.box_img_stride & 3) return SOD_EARG;
  return SOD_OK;
}

extern "C" int sod_points2bbox_fwd(const float* pts, const float* add, int ld, int N, int H, int W, float grid_stride, float point_stride,
                                   int num_points, float* boxes, long long box_img_stride, unsigned* argidx, long long arg_img_stride,
                                   void* stream) {
  if (!pts || !boxes) return SOD_EARG;
  P2BArgs a{};
  int rc = p2b_fill(a, ld, N, H, W, grid_stride, point_stride, num_points, box_img_stride, arg_img_stride);
  if (rc) return rc;
  a.pts = pts; a.add = add; a.boxes = boxes; a.arg = argidx;
  SOD_LAUNCH(p2b_fwd_kernel, dim3(rp_nblk((long long)N * H * W, 8192)), dim3(256), 0, (hipStream_t)stream, a);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_points2bbox_bwd(const float* dboxes, long long box_img_stride, const unsigned* argidx, long long arg_img_stride, int ld,
                                   int N, int H, int W, float point_stride, int num_points, float* dpts_f32, void* dpts_bf16, void* stream) {
  if (!dboxes || !argidx || (!dpts_f32 && !dpts_bf16)) return SOD_EARG;
  P2BArgs a{};
  int rc = p2b_fill(a, ld, N, H, W, 0.f, point_stride, num_points, box_img_stride, arg_img_stride);
  if (rc) return rc;
  a.arg = const_cast<unsigned*>(argidx);
  SOD_LAUNCH(p2b_bwd_kernel, dim3(rp_nblk((long long)N * H * W, 8192)), dim3(256), 0, (hipStream_t)stream, a, dboxes, dpts_f32, (__bf16*)dpts_bf16);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_points2bbox_moment_fwd(const float* pts, const float* add, int ld, int N, int H, int W, float grid_stride, float point_stride,
                                          int num_points, const float* moment_transfer, float* boxes, long long box_img_stride, void* stream) {
  if (!pts || !boxes || !moment_transfer || num_points < 2) return SOD_EARG;
  P2BArgs a{};
  int rc = p2b_fill(a, ld, N, H, W, grid_stride, point_stride, num_points, box_img_stride, 0);
  if (rc) return rc;
  a.pts = pts; a.add = add; a.boxes = boxes; a.arg = nullptr;
  SOD_LAUNCH(p2b_moment_fwd_kernel, dim3(rp_nblk((long long)N * H * W, 8192)), dim3(256), 0, (hipStream_t)stream, a, moment_transfer);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_points2bbox_moment_bwd(const float* dboxes, long long box_img_stride, const float* pts, const float* add, int ld, int N, int H,
                                          int W, float grid_stride, float point_stride, int num_points, const float* moment_transfer,
                                          float moment_mul, float* dpts_f32, void* dpts_bf16, float* dmoment2, void* stream) {
  if (!dboxes || !pts || !moment_transfer || (!dpts_f32 && !dpts_bf16) || num_points < 2) return SOD_EARG;
  P2BArgs a{};
  int rc = p2b_fill(a, ld, N, H, W, grid_stride, point_stride, num_points, box_img_stride, 0);
  if (rc) return rc;
  a.pts = pts; a.add = add;
  SOD_LAUNCH(p2b_moment_bwd_kernel, dim3(rp_nblk((long long)N * H * W, 1024)), dim3(256), 0, (hipStream_t)stream, a, dboxes, moment_transfer,
             moment_mul, dpts_f32, (__bf16*)dpts_bf16, dmoment2);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_reppoints_point_match(const float* centers, const float* strides, int X, const int* lvl_start, int num_levels,
                                         const float* gt_boxes, const int* box_offsets, int N, int max_gt, int mode, float scale,
                                         int* objectness, float* box_labels, void* stream) {
  if (!centers || !strides || !lvl_start || !box_offsets || !objectness || !box_labels || X <= 0 || num_levels <= 0 || N <= 0) return SOD_EARG;
  if (mode < 0 || mode > 2 || max_gt < 0 || max_gt > 4096 || (max_gt > 0 && !gt_boxes) || !(scale > 0.f)) return SOD_EARG;
  RpMatchArgs a{centers, strides, lvl_start, gt_boxes, box_offsets, objectness, box_labels, X, num_levels, mode, scale};
  SOD_LAUNCH(rp_match_kernel, dim3(N), dim3(256), sizeof(float) * 3 * (max_gt > 0 ? max_gt : 1), (hipStream_t)stream, a);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_reppoints_labels(const int* matches, const signed char* match_labels, const float* gt_boxes, const int* gt_classes,
                                    const int* box_offsets, const float* centers, const float* image_hw, int N, int X, int num_classes,
                                    int* cls_labels, float* refine_boxes, int* objectness, void* stream) {
  if (!matches || !match_labels || !box_offsets || !centers || !image_hw || !cls_labels || !refine_boxes || N <= 0 || X <= 0) return SOD_EARG;
  SOD_LAUNCH(rp_labels_kernel, dim3(rp_nblk((long long)N * X, 4096)), dim3(256), 0, (hipStream_t)stream, matches, match_labels, gt_boxes, gt_classes,
             box_offsets, centers, image_hw, N, X, num_classes, cls_labels, refine_boxes, objectness);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_reppoints_box_loss_fwd(const float* pred, const float* target, const int* labels, const float* strides, int N, int X,
                                          int bg_label, float beta, float* sums2, float* ws, void* stream) {
  if (!pred || !target || !labels || !strides || !sums2 || !ws || N <= 0 || X <= 0) return SOD_EARG;
  RpBoxArgs a{pred, target, labels, strides, N, X, bg_label, beta};
  hipStream_t st = (hipStream_t)stream;
  const int g = rp_nblk((long long)N * X);
  SOD_LAUNCH(rp_box_loss_kernel<false>, dim3(g), dim3(256), 0, st, a, ws, nullptr, nullptr, 0.f, 0.f, nullptr);
  SOD_LAUNCH(rp_box_finish_kernel, dim3(1), dim3(256), 0, st, ws, g, sums2);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_reppoints_box_loss_bwd(const float* pred, const float* target, const int* labels, const float* strides, int N, int X,
                                          int bg_label, float beta, const float* grad_num, const float* grad_den, float den_min, float mul,
                                          float* dpred, void* stream) {
  if (!pred || !target || !labels || !strides || !grad_num || !grad_den || !dpred || N <= 0 || X <= 0) return SOD_EARG;
  RpBoxArgs a{pred, target, labels, strides, N, X, bg_label, beta};
  SOD_LAUNCH(rp_box_loss_kernel<true>, dim3(rp_nblk((long long)N * X, 4096)), dim3(256), 0, (hipStream_t)stream, a, nullptr, grad_num, grad_den,
             den_min, mul, dpred);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_reppoints_finalize(const float* focal_sum, const float* init_sums2, const float* refine_sums2, float* normalizer,
                                      float momentum, int num_images, float init_weight, float* out3, void* stream) {
  if (!focal_sum || !init_sums2 || !refine_sums2 || !normalizer || !out3 || num_images <= 0) return SOD_EARG;
  SOD_LAUNCH(rp_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, focal_sum, init_sums2, refine_sums2, normalizer, momentum,
             1.f / (float)num_images, init_weight, out3);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}
