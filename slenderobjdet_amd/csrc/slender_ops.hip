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

// one thread per (batch, channel, box, border e): channel e*C + c of the feature map, pool_size+1 samples along border e, max
template <bool BWD>
__global__ __launch_bounds__(256) void border_align_kernel(const float* __restrict__ feature, const float* __restrict__ boxes,
                                                           int B, int C, int K, int H, int W, int pool, float* __restrict__ out,
                                                           const float* __restrict__ dout, float* __restrict__ dfeat) {
  const long long total = (long long)B * C * K * 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int e = (int)(i & 3);
    long long t = i >> 2;
    const int k = (int)(t % K); t /= K;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    const float* bx = boxes + ((long long)b * K + k) * 4;
    float x = bx[(e >> 1) * 2], y = bx[(e >> 1) * 2 + 1];
    const float bw = bx[2] - bx[0], bh = bx[3] - bx[1];
    float xs = 0.f, ys = 0.f;
    if (e == 0) xs = bw / pool; else if (e == 1) ys = bh / pool; else if (e == 2) xs = -(bw / pool); else ys = -(bh / pool);
    const long long plane = ((long long)b * 4 * C + (long long)e * C + c) * H * W;
    const float* f = feature + plane;
    float best = bilin_at(f, W, bilin_setup(y, x, H, W));
    int arg = 0;
    float ax = x, ay = y;
    for (int s = 1; s <= pool; ++s) {
      x += xs; y += ys;
      const float v = bilin_at(f, W, bilin_setup(y, x, H, W));
      if (v > best) { best = v; arg = s; ax = x; ay = y; }
    }
    if (!BWD) {
      out[i] = best;
    } else {
      (void)arg;
      const Bilin bl = bilin_setup(ay, ax, H, W);
      const float g = dout[i];
      float* d = dfeat + plane;
      atomicAdd(d + bl.yl * W + bl.xl, g * bl.w1); atomicAdd(d + bl.yl * W + bl.xh, g * bl.w2);
      atomicAdd(d + bl.yh * W + bl.xl, g * bl.w3); atomicAdd(d + bl.yh * W + bl.xh, g * bl.w4);
    }
  }
}

// directional running max: mode 0 bottom (scan h up->down), 1 top (h reversed), 2 right (scan w), 3 left (w reversed).
// One thread per scan line; lines are adjacent in memory for the h-scans (coalesced), strided for the w-scans.
template <bool BWD>
__global__ __launch_bounds__(256) void corner_pool_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ dy,
                                                          float* __restrict__ dx, long long planes, int H, int W, int mode, int tie_latest) {
  const bool along_h = mode < 2, rev = (mode & 1) != 0;
  const int len = along_h ? H : W, lines_per_plane = along_h ? W : H;
  const long long total = planes * lines_per_plane;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long pl = i / lines_per_plane;
    const int ln = (int)(i - pl * lines_per_plane);
    const long long base = pl * H * W + (along_h ? ln : (long long)ln * W);
    const int step = along_h ? W : 1;
    float best = 0.f;
    int arg = 0;
    for (int j = 0; j < len; ++j) {
      const int jj = rev ? len - 1 - j : j;
      const float v = x[base + (long long)jj * step];
      if (j == 0 || v > best || (tie_latest && v == best)) { best = v; arg = jj; }
      if (!BWD) y[base + (long long)jj * step] = best;
      else atomicAdd(dx + base + (long long)arg * step, dy[base + (long long)jj * step]);   // same-thread line: no contention
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
  SOD_LAUNCH(border_align_kernel<false>, dim3(grid_for((long long)B * C * K * 4)), dim3(256), 0, (hipStream_t)stream, feature, boxes, B, C, K, H, W,
             pool_size, out, nullptr, nullptr);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_border_align_bwd(const float* dout, const float* feature, const float* boxes, float* dfeature, int B, int C, int K, int H,
                                    int W, int pool_size, void* stream) {
  if (!dout || !feature || !boxes || !dfeature || B <= 0 || C <= 0 || K <= 0 || H <= 0 || W <= 0 || pool_size <= 0) return SOD_EARG;
  SOD_LAUNCH(border_align_kernel<true>, dim3(grid_for((long long)B * C * K * 4)), dim3(256), 0, (hipStream_t)stream, feature, boxes, B, C, K, H, W,
             pool_size, nullptr, dout, dfeature);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_corner_pool_fwd(const float* x, float* y, long long planes, int H, int W, int mode, void* stream) {
  if (!x || !y || planes <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 3) return SOD_EARG;
  SOD_LAUNCH(corner_pool_kernel<false>, dim3(grid_for(planes * (mode < 2 ? W : H))), dim3(256), 0, (hipStream_t)stream, x, y, nullptr, nullptr,
             planes, H, W, mode, 1);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_corner_pool_bwd(const float* x, const float* dy, float* dx, long long planes, int H, int W, int mode, int tie_latest,
                                   void* stream) {
  if (!x || !dy || !dx || planes <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 3) return SOD_EARG;
  SOD_LAUNCH(corner_pool_kernel<true>, dim3(grid_for(planes * (mode < 2 ? W : H))), dim3(256), 0, (hipStream_t)stream, x, nullptr, dy, dx, planes, H,
             W, mode, tie_latest);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}
