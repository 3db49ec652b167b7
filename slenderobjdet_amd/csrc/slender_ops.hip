// The reference's OWN native operators (slender_det._C), as HIP kernels for gfx950:
//   BorderAlign   slender_det/layers/csrc/border_align/BorderAlign_cuda.cu:94-146 (fwd), :209-276 (bwd); binding vision.cpp:77-79
//   CornerPool    slender_det/layers/csrc/corner_pool/corner_pool.cpp:11-256 (directional running max + scatter of the gradient to
//                 the arg-max); on torch >= 1.5 the reference runs torch.cummax instead (layers/corner_pool.py:106-116)
// Same tensor contract as the reference ops: fp32, NCHW.  Both are latency/HBM-bound gathers and scans; no MFMA.
#include "common.h"
#include "../../include/slender_hip.h"

namespace {

struct Bilin { int yl, xl, yh, xh; float w1, w2, w3, w4; };

// bilinear setup with the reference's clamping rule: only the upper edge is clamped (BorderAlign_cuda.cu:16-45)
__device__ __forceinline__ Bilin bilin_setup(float y, float x, int H, int W) {
  Bilin b;
  b.yl = (int)y; b.xl = (int)x;
  if (b.yl >= H - 1) { b.yh = b.yl = H - 1; y = (float)b.yl; } else b.yh = b.yl + 1;
  if (b.xl >= W - 1) { b.xh = b.xl = W - 1; x = (float)b.xl; } else b.xh = b.xl + 1;
  const float ly = y - b.yl, lx = x - b.xl, hy = 1.f - ly, hx = 1.f - lx;
  b.w1 = hy * hx; b.w2 = hy * lx; b.w3 = ly * hx; b.w4 = ly * lx;
  return b;
}

__device__ __forceinline__ float bilin_at(const float* f, int W, const Bilin& b) {
  return b.w1 * f[b.yl * W + b.xl] + b.w2 * f[b.yl * W + b.xh] + b.w3 * f[b.yh * W + b.xl] + b.w4 * f[b.yh * W + b.xh];
}

// One thread per (batch, channel, box): its four borders e in turn, channel e*C + c of the feature map, pool_size+1 samples along border e,
// max.  Neighbouring threads hold neighbouring boxes (one box per feature-map location in BorderDet), so the bilinear corners of a wave
// fall into a few cache lines of ONE channel plane, and the four results of a thread are one 16-byte store.  (Until round 5 the border
// index was the fastest thread index: four neighbouring threads read four different channel planes, H*W*4 bytes apart.)
template <bool BWD>
__global__ __launch_bounds__(256) void border_align_kernel(const float* __restrict__ feature, const float* __restrict__ boxes,
                                                           int B, int C, int K, int H, int W, int pool, float* __restrict__ out,
                                                           const float* __restrict__ dout, float* __restrict__ dfeat) {
  const long long total = (long long)B * C * K;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int k = (int)(i % K);
    long long t = i / K;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    const f32x4_t bx = *reinterpret_cast<const f32x4_t*>(boxes + ((long long)b * K + k) * 4);
    const float bw = bx[2] - bx[0], bh = bx[3] - bx[1];
    f32x4_t res = {0.f, 0.f, 0.f, 0.f};
    f32x4_t g4 = {0.f, 0.f, 0.f, 0.f};
    if (BWD) g4 = *reinterpret_cast<const f32x4_t*>(dout + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = bx[(e >> 1) * 2], y = bx[(e >> 1) * 2 + 1];
      float xs = 0.f, ys = 0.f;
      if (e == 0) xs = bw / pool; else if (e == 1) ys = bh / pool; else if (e == 2) xs = -(bw / pool); else ys = -(bh / pool);
      const long long plane = ((long long)b * 4 * C + (long long)e * C + c) * H * W;
      const float* f = feature + plane;
      float best = bilin_at(f, W, bilin_setup(y, x, H, W));
      float ax = x, ay = y;
      for (int s = 1; s <= pool; ++s) {
        x += xs; y += ys;
        const float v = bilin_at(f, W, bilin_setup(y, x, H, W));
        if (v > best) { best = v; ax = x; ay = y; }
      }
      if (!BWD) {
        res[e] = best;
      } else {
        const Bilin bl = bilin_setup(ay, ax, H, W);
        const float g = g4[e];
        float* d = dfeat + plane;
        atomicAdd(d + bl.yl * W + bl.xl, g * bl.w1); atomicAdd(d + bl.yl * W + bl.xh, g * bl.w2);
        atomicAdd(d + bl.yh * W + bl.xl, g * bl.w3); atomicAdd(d + bl.yh * W + bl.xh, g * bl.w4);
      }
    }
    if (!BWD) *reinterpret_cast<f32x4_t*>(out + i * 4) = res;
  }
}

// directional running max: mode 0 bottom (scan h up->down), 1 top (h reversed), 2 right (scan w), 3 left (w reversed).
// Scans along H: one thread per column - the lines of neighbouring threads are neighbours in memory, every step of the scan is a coalesced
// row access.
template <bool BWD>
__global__ __launch_bounds__(256) void corner_pool_h_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ dy,
                                                            float* __restrict__ dx, long long planes, int H, int W, int rev, int tie_latest) {
  const long long total = planes * W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long pl = i / W;
    const int ln = (int)(i - pl * W);
    const long long base = pl * H * W + ln;
    float best = 0.f, run = 0.f;
    int arg = 0;
    for (int j = 0; j < H; ++j) {
      const int jj = rev ? H - 1 - j : j;
      const float v = x[base + (long long)jj * W];
      if (j == 0 || v > best || (tie_latest && v == best)) {
        // the maximum moves: what the old position collected goes out with ONE read-modify-write (the column belongs to this thread: no
        // atomics; until round 5 every element was an atomicAdd - 1.37 ms instead of 0.06 ms at 16 x 128 x 128 x 128)
        if (BWD && j > 0) dx[base + (long long)arg * W] += run;
        best = v; arg = jj; run = 0.f;
      }
      if (!BWD) y[base + (long long)jj * W] = best;
      else run += dy[base + (long long)jj * W];
    }
    if (BWD) dx[base + (long long)arg * W] += run;
  }
}

// Scans along W: one WAVE per line, 64 consecutive elements per step (one 256-byte access), the running maximum as a wave-level inclusive
// scan (six shuffle steps) carried from chunk to chunk.  Backward: the scan runs on (value, position) pairs - a later element replaces the
// running maximum if it is greater (or equal, with tie_latest) - so every lane knows where the maximum of its prefix sits; those positions
// are non-decreasing along the scan, so the gradient a position receives is the sum over a contiguous run of lanes: a segmented sum
// (six more shuffle steps), one atomicAdd per run and chunk.  (Until round 5: one thread per line, i.e. a wave read 64 addresses W floats
// apart per step.)
template <bool BWD>
__global__ __launch_bounds__(256) void corner_pool_w_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ dy,
                                                            float* __restrict__ dx, long long lines, int W, int rev, int tie_latest) {
  const int lane = threadIdx.x & 63;
  const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
  for (long long ln = wave0; ln < lines; ln += nwaves) {
    const long long base = ln * W;
    float cbest = 0.f;       // carry: running maximum and its scan position after the previous chunks
    int carg = -1;
    for (int p0 = 0; p0 < W; p0 += 64) {
      const int p = p0 + lane;                       // scan position
      const bool live = p < W;
      const int m = rev ? W - 1 - p : p;             // memory index
      float v = live ? x[base + m] : 0.f;
      int a = p;
      // inclusive scan of (value, position): earlier aggregate (from lane - d) combined with this lane's later aggregate
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const float ov = __shfl_up(v, d, 64);
        const int oa = __shfl_up(a, d, 64);
        if (lane >= d && !(v > ov || (tie_latest && v == ov))) { v = ov; a = oa; }
      }
      if (carg >= 0 && !(v > cbest || (tie_latest && v == cbest))) { v = cbest; a = carg; }
      if (!BWD) {
        if (live) y[base + m] = v;
      } else {
        // segmented inclusive sum of dy over runs of equal arg; a run starts where arg == own position (or at lane 0 of the chunk)
        float s = live ? dy[base + m] : 0.f;
        bool head = (a == p) || lane == 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const float os = __shfl_up(s, d, 64);
          const int oh = __shfl_up((int)head, d, 64);
          if (lane >= d && !head) { s += os; head = oh != 0; }
        }
        const int an = __shfl_down(a, 1, 64);
        const bool last = live && (lane == 63 || p == W - 1 || an != a);
        if (last) atomicAdd(dx + base + (rev ? W - 1 - a : a), s);
      }
      // carry = the aggregate of the last live lane of this chunk
      const int src = (W - p0 >= 64) ? 63 : (W - p0 - 1);
      cbest = __shfl(v, src, 64);
      carg = __shfl(a, src, 64);
    }
  }
}

inline int grid_for(long long n) {
  long long g = (n + 255) / 256;
  if (g > 8192) g = 8192;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int sod_border_align_fwd(const float* feature, const float* boxes, float* out, int B, int C, int K, int H, int W, int pool_size,
                                    void* stream) {
  if (!feature || !boxes || !out || B <= 0 || C <= 0 || K <= 0 || H <= 0 || W <= 0 || pool_size <= 0) return SOD_EARG;
  SOD_LAUNCH(border_align_kernel<false>, dim3(grid_for((long long)B * C * K)), dim3(256), 0, (hipStream_t)stream, feature, boxes, B, C, K, H, W,
             pool_size, out, nullptr, nullptr);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_border_align_bwd(const float* dout, const float* feature, const float* boxes, float* dfeature, int B, int C, int K, int H,
                                    int W, int pool_size, void* stream) {
  if (!dout || !feature || !boxes || !dfeature || B <= 0 || C <= 0 || K <= 0 || H <= 0 || W <= 0 || pool_size <= 0) return SOD_EARG;
  SOD_LAUNCH(border_align_kernel<true>, dim3(grid_for((long long)B * C * K)), dim3(256), 0, (hipStream_t)stream, feature, boxes, B, C, K, H, W,
             pool_size, nullptr, dout, dfeature);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_corner_pool_fwd(const float* x, float* y, long long planes, int H, int W, int mode, void* stream) {
  if (!x || !y || planes <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 3) return SOD_EARG;
  if (mode < 2) {
    SOD_LAUNCH(corner_pool_h_kernel<false>, dim3(grid_for(planes * W)), dim3(256), 0, (hipStream_t)stream, x, y, nullptr, nullptr, planes, H, W, mode & 1, 1);
  } else {
    SOD_LAUNCH(corner_pool_w_kernel<false>, dim3(grid_for(planes * H * 64)), dim3(256), 0, (hipStream_t)stream, x, y, nullptr, nullptr, planes * H, W,
               mode & 1, 1);
  }
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_corner_pool_bwd(const float* x, const float* dy, float* dx, long long planes, int H, int W, int mode, int tie_latest,
                                   void* stream) {
  if (!x || !dy || !dx || planes <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 3) return SOD_EARG;
  if (mode < 2) {
    SOD_LAUNCH(corner_pool_h_kernel<true>, dim3(grid_for(planes * W)), dim3(256), 0, (hipStream_t)stream, x, nullptr, dy, dx, planes, H, W, mode & 1,
               tie_latest);
  } else {
    SOD_LAUNCH(corner_pool_w_kernel<true>, dim3(grid_for(planes * H * 64)), dim3(256), 0, (hipStream_t)stream, x, nullptr, dy, dx, planes * H, W, mode & 1,
               tie_latest);
  }
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}
