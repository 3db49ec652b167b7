// fp32-STORAGE validation path (SOD_PRECISION=fp32): the same operators as the bf16 product path with every activation, weight copy
// and gradient kept in fp32 and every product accumulated in fp32 FMA.  It exists to make north_star's "total-loss delta < 1e-3 vs the
// reference's CPU path after 100 iterations" a statement the implementation can be held to: with bf16 storage ANY two runs (the CPU
// emulation included) drift by a few 1e-3 over 100 SGD steps, so that bound can only be asserted where storage
// rounding is out of the picture.  Untuned by design (tiled SGEMM-style implicit GEMM, ~2-4 TFLOP/s): it is never what bench.py times.
//
// Operators and the reference code they stand for:
//   conv fwd / dgrad / wgrad            ATen conv2d in detectron2 ResNet / FPN and FCOSHead (fcosv2.py:277-381)
//   GroupNorm(+ReLU) fwd / bwd          nn.GroupNorm(32, C) + ReLU of the towers (fcosv2.py:315-336)
//   relu, add, add_up2, upsample bwd    F.relu, residual / FPN top-down sums
//   max-pool 3x3 s2, preprocess         BasicStem pool, FCOSV2.preprocess_image (fcosv2.py:268-275)
// Every reduction is a fixed-order tree or a sequential loop: results are bit-identical from run to run.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/slender_hip.h"
#include "common.h"

namespace {

struct CG {
  const float* a_src;   // fwd: x     dgrad: dy    wgrad: dy
  const float* b_src;   // fwd: w     dgrad: w     wgrad: x
  float* out;           // fwd: y     dgrad: dx    wgrad: dw (+=)
  const float* bias;    // fwd
  const float* res;     // fwd residual / dgrad accumulate
  const float* mask;    // dgrad: ReLU mask (tensor of dx's shape, > 0 keeps)
  const float* qscale;  // wgrad: per-output-channel factor
  int N, H, W, C, K, R, S, stride, pad, dil, Ho, Wo;
  long long x_img, y_img;   // image strides (elements) of the (N,H,W,C) tensor and of the (N,Ho,Wo,K) tensor
  int relu, res_mode;       // res_mode: 0 none, 1 same shape, 2 half resolution (nearest 2x up-sampling / even-pixel accumulate)
  int M, Nn, Kd;
};

// element (m, kk) of the gathered left operand and (kk, n) of the right one
template <int MODE>
__device__ __forceinline__ float load_a(const CG& g, int m, int kk) {
  if (m >= g.M || kk >= g.Kd) return 0.f;
  if (MODE == 0) {          // x[n, ho*s-p+r*d, wo*s-p+q*d, c]
    const int c = kk % g.C, rs = kk / g.C, q = rs % g.S, r = rs / g.S;
    const int wo = m % g.Wo, t = m / g.Wo, ho = t % g.Ho, n = t / g.Ho;
    const int h = ho * g.stride - g.pad + r * g.dil, w = wo * g.stride - g.pad + q * g.dil;
    if (h < 0 || h >= g.H || w < 0 || w >= g.W) return 0.f;
    return g.a_src[(long long)n * g.x_img + ((long long)h * g.W + w) * g.C + c];
  } else if (MODE == 1) {   // dy[n, (h+p-r*d)/s, (w+p-q*d)/s, ko]
    const int ko = kk % g.K, rs = kk / g.K, q = rs % g.S, r = rs / g.S;
    const int w = m % g.W, t = m / g.W, h = t % g.H, n = t / g.H;
    const int hn = h + g.pad - r * g.dil, wn = w + g.pad - q * g.dil;
    if (hn < 0 || wn < 0 || hn % g.stride || wn % g.stride) return 0.f;
    const int ho = hn / g.stride, wo = wn / g.stride;
    if (ho >= g.Ho || wo >= g.Wo) return 0.f;
    return g.a_src[(long long)n * g.y_img + ((long long)ho * g.Wo + wo) * g.K + ko];
  } else {                  // dy[pixel kk, channel m]
    const int wo = kk % g.Wo, t = kk / g.Wo, ho = t % g.Ho, n = t / g.Ho;
    return g.a_src[(long long)n * g.y_img + ((long long)ho * g.Wo + wo) * g.K + m];
  }
}

template <int MODE>
__device__ __forceinline__ float load_b(const CG& g, int kk, int n) {
  if (n >= g.Nn || kk >= g.Kd) return 0.f;
  if (MODE == 0) {          // w[n][kk]   (K, R*S*C)
    return g.b_src[(long long)n * g.Kd + kk];
  } else if (MODE == 1) {   // w[ko][rs][c]
    const int ko = kk % g.K, rs = kk / g.K;
    return g.b_src[((long long)ko * g.R * g.S + rs) * g.C + n];
  } else {                  // x at pixel kk, tap / channel n
    const int c = n % g.C, rs = n / g.C, q = rs % g.S, r = rs / g.S;
    const int wo = kk % g.Wo, t = kk / g.Wo, ho = t % g.Ho, im = t / g.Ho;
    const int h = ho * g.stride - g.pad + r * g.dil, w = wo * g.stride - g.pad + q * g.dil;
    if (h < 0 || h >= g.H || w < 0 || w >= g.W) return 0.f;
    return g.b_src[(long long)im * g.x_img + ((long long)h * g.W + w) * g.C + c];
  }
}

// C[M, Nn] = sum_kk A(m, kk) B(kk, n): 64x64 tile, 16-deep chunks through LDS, 256 threads x (4x4) fp32 FMA accumulators;
// the contraction runs in index order within a thread (a plain sequential fp32 sum)
template <int MODE>
__global__ __launch_bounds__(256) void conv_f32_kernel(const CG g) {
  __shared__ float As[16][68];
  __shared__ float Bs[16][68];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  // loader mappings: "kc" = 4 consecutive kk of one row (contiguous when the contraction index is the memory-contiguous one),
  // "rc" = one kk-quad of 64 consecutive rows (contiguous when the row index is)
  const bool a_kc = MODE != 2, b_kc = MODE == 0;
  const int a_row = a_kc ? (t >> 2) : (t & 63), a_k = a_kc ? (t & 3) * 4 : (t >> 6) * 4;
  const int b_row = b_kc ? (t >> 2) : (t & 63), b_k = b_kc ? (t & 3) * 4 : (t >> 6) * 4;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < g.Kd; k0 += 16) {
    float a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i] = load_a<MODE>(g, m0 + a_row, k0 + a_k + i);
      b[i] = load_b<MODE>(g, k0 + b_k + i, n0 + b_row);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      As[a_k + i][a_row] = a[i];
      Bs[b_k + i][b_row] = b[i];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        av[i] = As[kk][ty * 4 + i];
        bv[i] = Bs[kk][tx * 4 + i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= g.Nn) continue;
      float v = acc[i][j];
      if (MODE == 0) {
        const int wo = m % g.Wo, tt = m / g.Wo, ho = tt % g.Ho, im = tt / g.Ho;
        if (g.bias) v += g.bias[n];
        if (g.res_mode == 1) v += g.res[((long long)im * g.Ho * g.Wo + (long long)ho * g.Wo + wo) * g.K + n];
        else if (g.res_mode == 2) v += g.res[(((long long)im * (g.Ho / 2) + ho / 2) * (g.Wo / 2) + wo / 2) * g.K + n];
        if (g.relu) v = v > 0.f ? v : 0.f;
        g.out[(long long)im * g.y_img + ((long long)ho * g.Wo + wo) * g.K + n] = v;
      } else if (MODE == 1) {
        const int w = m % g.W, tt = m / g.W, h = tt % g.H, im = tt / g.H;
        const long long o = ((long long)im * g.H * g.W + (long long)h * g.W + w) * g.C + n;
        if (g.res_mode == 1) v += g.res[o];
        else if (g.res_mode == 2 && !(h & 1) && !(w & 1)) v += g.res[(((long long)im * (g.H / 2) + h / 2) * (g.W / 2) + w / 2) * g.C + n];
        if (g.mask) v = g.mask[o] > 0.f ? v : 0.f;
        g.out[o] = v;
      } else {
        if (g.qscale) v *= g.qscale[m];
        g.out[(long long)m * g.Nn + n] += v;
      }
    }
  }
}

int out_dim(int H, int pad, int dil, int R, int stride) { return (H + 2 * pad - dil * (R - 1) - 1) / stride + 1; }

int fill(CG& g, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil, long long x_img, long long y_img) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || R <= 0 || S <= 0 || stride <= 0 || dil <= 0 || pad < 0) return SOD_EARG;
  g.N = N; g.H = H; g.W = W; g.C = C; g.K = K; g.R = R; g.S = S; g.stride = stride; g.pad = pad; g.dil = dil;
  g.Ho = out_dim(H, pad, dil, R, stride); g.Wo = out_dim(W, pad, dil, S, stride);
  if (g.Ho <= 0 || g.Wo <= 0) return SOD_EARG;
  g.x_img = x_img > 0 ? x_img : (long long)H * W * C;
  g.y_img = y_img > 0 ? y_img : (long long)g.Ho * g.Wo * K;
  if (g.x_img < (long long)H * W * C || g.y_img < (long long)g.Ho * g.Wo * K) return SOD_EARG;
  if ((long long)N * g.Ho * g.Wo >= (1ll << 31) || (long long)N * H * W >= (1ll << 31) || (long long)R * S * C >= (1ll << 31) ||
      (long long)R * S * K >= (1ll << 31))
    return SOD_ESIZE;
  return SOD_OK;
}

template <int MODE>
int launch(CG& g, hipStream_t st) {
  const dim3 grid((g.M + 63) / 64, (g.Nn + 63) / 64);
  if (grid.y > 65535u) return SOD_ESIZE;
  SOD_LAUNCH(conv_f32_kernel<MODE>, grid, dim3(256), 0, st, g);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

// ------------------------------------------------------------------------------------------------ GroupNorm
// One workgroup per group g, images in sequence: statistics by a two-pass mean / variance, fixed-order block reductions.
__device__ __forceinline__ float block_sum(float v, float* sm) {
  const int t = threadIdx.x;
  sm[t] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) sm[t] += sm[t + s];
    __syncthreads();
  }
  const float r = sm[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void gn_fwd_f32_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ y, float* __restrict__ stats, int N, int HW, int C, int G, float eps,
                                                         int relu) {
  __shared__ float sm[256];
  const int grp = blockIdx.x, n = blockIdx.y, Cg = C / G;
  const long long base = (long long)n * HW * C + grp * Cg;
  const long long cnt = (long long)HW * Cg;
  float s = 0.f;
  for (long long e = threadIdx.x; e < cnt; e += 256) s += x[base + (e / Cg) * C + e % Cg];
  const float mean = block_sum(s, sm) / (float)cnt;
  float q = 0.f;
  for (long long e = threadIdx.x; e < cnt; e += 256) {
    const float d = x[base + (e / Cg) * C + e % Cg] - mean;
    q += d * d;
  }
  const float var = block_sum(q, sm) / (float)cnt;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    stats[((long long)n * G + grp) * 2] = mean;
    stats[((long long)n * G + grp) * 2 + 1] = rstd;
  }
  for (long long e = threadIdx.x; e < cnt; e += 256) {
    const int c = grp * Cg + (int)(e % Cg);
    const long long o = base + (e / Cg) * C + e % Cg;
    float v = (x[o] - mean) * rstd * gamma[c] + beta[c];
    if (relu) v = v > 0.f ? v : 0.f;
    y[o] = v;
  }
}

// backward: thread t owns channel (t % Cg) of the group (256 % Cg == 0), so dgamma / dbeta / dxsum are per-thread sums over the thread's
// elements of every image, combined across the threads of one channel in a fixed order at the end
__global__ __launch_bounds__(256) void gn_bwd_f32_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ stats, float* __restrict__ dx,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dxsum, int N, int HW,
                                                         int C, int G, int relu) {
  __shared__ float sm[256];
  __shared__ float ch[3][256];
  const int grp = blockIdx.x, Cg = C / G, t = threadIdx.x;
  const int c = grp * Cg + t % Cg;
  const float gam = gamma[c], bet = beta[c];
  const long long cnt = (long long)HW * Cg;
  float dg = 0.f, db = 0.f, ds = 0.f;
  for (int n = 0; n < N; ++n) {
    const long long base = (long long)n * HW * C + grp * Cg;
    const float mean = stats[((long long)n * G + grp) * 2], rstd = stats[((long long)n * G + grp) * 2 + 1];
    float a = 0.f, b = 0.f;
    for (long long e = t; e < cnt; e += 256) {
      const long long o = base + (e / Cg) * C + e % Cg;
      const float xh = (x[o] - mean) * rstd;
      float gv = dy[o];
      if (relu && !(xh * gam + bet > 0.f)) gv = 0.f;
      a += gv * gam;
      b += gv * gam * xh;
      dg += gv * xh;
      db += gv;
    }
    const float A = block_sum(a, sm) / (float)cnt, B = block_sum(b, sm) / (float)cnt;
    for (long long e = t; e < cnt; e += 256) {
      const long long o = base + (e / Cg) * C + e % Cg;
      const float xh = (x[o] - mean) * rstd;
      float gv = dy[o];
      if (relu && !(xh * gam + bet > 0.f)) gv = 0.f;
      const float d = rstd * (gv * gam - A - xh * B);
      dx[o] = d;
      ds += d;
    }
  }
  ch[0][t] = dg; ch[1][t] = db; ch[2][t] = ds;
  __syncthreads();
  if (t < Cg) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int u = t; u < 256; u += Cg) { s0 += ch[0][u]; s1 += ch[1][u]; s2 += ch[2][u]; }
    dgamma[c] += s0;
    dbeta[c] += s1;
    if (dxsum) dxsum[c] += s2;
  }
}

// ------------------------------------------------------------------------------------------------ element-wise
__global__ __launch_bounds__(256) void eltwise_f32_kernel(int op, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o,
                                                          long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float av = a[i];
    float v;
    if (op == 0) v = av > 0.f ? av : 0.f;                  // relu
    else if (op == 1) v = b[i] > 0.f ? av : 0.f;           // relu backward: a = dy, b = y
    else v = av + b[i];                                    // add
    o[i] = v;
  }
}

__global__ __launch_bounds__(256) void add_up2_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int N, int H,
                                                          int W, int C) {
  const long long total = (long long)N * H * W * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long long p = i / C;
    const int w = (int)(p % W); p /= W;
    const int h = (int)(p % H);
    const int n = (int)(p / H);
    o[i] = a[i] + b[(((long long)n * (H / 2) + h / 2) * (W / 2) + w / 2) * C + c];
  }
}

// gradient of the nearest 2x up-sampling: the four fine pixels of a coarse one, summed in a fixed order
__global__ __launch_bounds__(256) void upsample2x_bwd_f32_kernel(const float* __restrict__ g, float* __restrict__ d, int N, int Hc, int Wc, int C) {
  const long long total = (long long)N * Hc * Wc * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long long p = i / C;
    const int w = (int)(p % Wc); p /= Wc;
    const int h = (int)(p % Hc);
    const int n = (int)(p / Hc);
    const float* s = g + (((long long)n * 2 * Hc + 2 * h) * 2 * Wc + 2 * w) * C + c;
    d[i] = (s[0] + s[C]) + (s[(long long)2 * Wc * C] + s[(long long)2 * Wc * C + C]);
  }
}

__global__ __launch_bounds__(256) void maxpool3x3s2_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C, int Ho,
                                                               int Wo) {
  const long long total = (long long)N * Ho * Wo * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long long p = i / C;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    float m = -INFINITY;
    for (int r = 0; r < 3; ++r)
      for (int q = 0; q < 3; ++q) {
        const int h = 2 * ho - 1 + r, w = 2 * wo - 1 + q;
        if (h < 0 || h >= H || w < 0 || w >= W) continue;
        const float v = x[(((long long)n * H + h) * W + w) * C + c];
        m = v > m ? v : m;
      }
    y[i] = m;
  }
}

// per-channel sum over the pixels of every image: one workgroup per channel, fixed-order tree
__global__ __launch_bounds__(256) void bias_grad_f32_kernel(const float* __restrict__ dy, float* __restrict__ db, int N, int HW, int C,
                                                            long long img_stride) {
  __shared__ float sm[256];
  const int c = blockIdx.x;
  float s = 0.f;
  for (int n = 0; n < N; ++n)
    for (int p = threadIdx.x; p < HW; p += 256) s += dy[(long long)n * img_stride + (long long)p * C + c];
  const float r = block_sum(s, sm);
  if (threadIdx.x == 0) db[c] += r;
}

template <typename T>
__global__ __launch_bounds__(256) void preprocess_f32_kernel(const T* __restrict__ img, int C, int H, int W, float* __restrict__ out, int Hp, int Wp,
                                                             int Cpad, float m0, float m1, float m2, float s0, float s1, float s2) {
  const long long total = (long long)Hp * Wp * Cpad;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % Cpad);
    const long long p = i / Cpad;
    const int w = (int)(p % Wp), h = (int)(p / Wp);
    float v = 0.f;
    if (c < C && h < H && w < W) {
      const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
      v = ((float)img[((long long)c * H + h) * W + w] - mean) / sd;
    }
    out[i] = v;
  }
}

// fp32 compute copies of a weight: KRSC (x scale[k], input channels zero-padded to Cpad) and CRSK
__global__ __launch_bounds__(256) void weight_prep_f32_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ krsc,
                                                              float* __restrict__ crsk, int K, int RS, int C, int Cpad) {
  const long long total = (long long)K * RS * Cpad;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % Cpad);
    const long long t = i / Cpad;
    const int rs = (int)(t % RS), k = (int)(t / RS);
    float v = 0.f;
    if (c < C) {
      v = w[((long long)k * RS + rs) * C + c];
      if (scale) v *= scale[k];
      if (crsk) crsk[((long long)c * RS + rs) * K + k] = v;
    }
    if (krsc) krsc[i] = v;
  }
}

int grid_for(long long n) {
  const long long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" int sod_conv2d_fwd_f32(const float* x, const float* w, const float* bias, const float* res, float* y, int N, int H, int W, int C, int K, int R,
                                  int S, int stride, int pad, int dil, long long x_img_stride, long long y_img_stride, int flags, void* stream) {
  if (!x || !w || !y) return SOD_EARG;
  CG g{};
  int rc = fill(g, N, H, W, C, K, R, S, stride, pad, dil, x_img_stride, y_img_stride);
  if (rc) return rc;
  g.a_src = x; g.b_src = w; g.out = y; g.bias = bias; g.res = res;
  g.relu = flags & 1;
  g.res_mode = res ? ((flags & 2) ? 2 : 1) : 0;
  if (g.res_mode == 2 && ((g.Ho & 1) || (g.Wo & 1))) return SOD_EARG;
  if (g.res_mode && y_img_stride > 0 && y_img_stride != (long long)g.Ho * g.Wo * K) return SOD_EARG;
  g.M = N * g.Ho * g.Wo; g.Nn = K; g.Kd = R * S * C;
  return launch<0>(g, (hipStream_t)stream);
}

extern "C" int sod_conv2d_dgrad_f32(const float* dy, const float* w, const float* accum, const float* relu_mask, float* dx, int N, int H, int W, int C,
                                    int K, int R, int S, int stride, int pad, int dil, long long dy_img_stride, int accum_even, void* stream) {
  if (!dy || !w || !dx) return SOD_EARG;
  CG g{};
  int rc = fill(g, N, H, W, C, K, R, S, stride, pad, dil, 0, dy_img_stride);
  if (rc) return rc;
  g.a_src = dy; g.b_src = w; g.out = dx; g.res = accum; g.mask = relu_mask;
  g.res_mode = accum ? (accum_even ? 2 : 1) : 0;
  if (g.res_mode == 2 && ((H & 1) || (W & 1))) return SOD_EARG;
  g.M = N * H * W; g.Nn = C; g.Kd = R * S * K;
  return launch<1>(g, (hipStream_t)stream);
}

extern "C" int sod_conv2d_wgrad_f32(const float* dy, const float* x, float* dw, const float* qscale, int N, int H, int W, int C, int K, int R, int S,
                                    int stride, int pad, int dil, long long dy_img_stride, long long x_img_stride, void* stream) {
  if (!dy || !x || !dw) return SOD_EARG;
  CG g{};
  int rc = fill(g, N, H, W, C, K, R, S, stride, pad, dil, x_img_stride, dy_img_stride);
  if (rc) return rc;
  g.a_src = dy; g.b_src = x; g.out = dw; g.qscale = qscale;
  g.M = K; g.Nn = R * S * C; g.Kd = N * g.Ho * g.Wo;
  return launch<2>(g, (hipStream_t)stream);
}

extern "C" int sod_groupnorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean_rstd, int N, int HW, int C, int G,
                                     float eps, int relu, void* stream) {
  if (!x || !gamma || !beta || !y || !mean_rstd || N <= 0 || HW <= 0 || G <= 0 || C <= 0 || C % G) return SOD_EARG;
  SOD_LAUNCH(gn_fwd_f32_kernel, dim3(G, N), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean_rstd, N, HW, C, G, eps, relu);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_groupnorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean_rstd, float* dx,
                                     float* dgamma, float* dbeta, float* dxsum, int N, int HW, int C, int G, int relu, void* stream) {
  if (!dy || !x || !gamma || !beta || !mean_rstd || !dx || !dgamma || !dbeta || N <= 0 || HW <= 0 || G <= 0 || C <= 0 || C % G) return SOD_EARG;
  if (256 % (C / G)) return SOD_EARG;
  SOD_LAUNCH(gn_bwd_f32_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, beta, mean_rstd, dx, dgamma, dbeta, dxsum, N, HW, C, G, relu);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_eltwise_f32(int op, const float* a, const float* b, float* out, long long n, void* stream) {
  if (!a || !out || n <= 0 || op < 0 || op > 2 || (op > 0 && !b)) return SOD_EARG;
  SOD_LAUNCH(eltwise_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, op, a, b, out, n);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_add_up2_f32(const float* a, const float* b, float* out, int N, int H, int W, int C, void* stream) {
  if (!a || !b || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (H & 1) || (W & 1)) return SOD_EARG;
  SOD_LAUNCH(add_up2_f32_kernel, dim3(grid_for((long long)N * H * W * C)), dim3(256), 0, (hipStream_t)stream, a, b, out, N, H, W, C);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_upsample2x_bwd_f32(const float* g, float* dprev, int N, int Hc, int Wc, int C, void* stream) {
  if (!g || !dprev || N <= 0 || Hc <= 0 || Wc <= 0 || C <= 0) return SOD_EARG;
  SOD_LAUNCH(upsample2x_bwd_f32_kernel, dim3(grid_for((long long)N * Hc * Wc * C)), dim3(256), 0, (hipStream_t)stream, g, dprev, N, Hc, Wc, C);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_maxpool3x3s2_f32(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0) return SOD_EARG;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  SOD_LAUNCH(maxpool3x3s2_f32_kernel, dim3(grid_for((long long)N * Ho * Wo * C)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C, Ho, Wo);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_bias_grad_f32(const float* dy, float* dbias, int N, int HW, int C, long long img_stride, void* stream) {
  if (!dy || !dbias || N <= 0 || HW <= 0 || C <= 0) return SOD_EARG;
  if (img_stride <= 0) img_stride = (long long)HW * C;
  SOD_LAUNCH(bias_grad_f32_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, dy, dbias, N, HW, C, img_stride);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_preprocess_image_f32(const void* img, int is_uint8, int C, int H, int W, float* out, int Hp, int Wp, int Cpad, const float* mean3,
                                        const float* std3, void* stream) {
  if (!img || !out || !mean3 || !std3 || C <= 0 || C > 3 || Cpad < C || H <= 0 || W <= 0 || Hp < H || Wp < W) return SOD_EARG;
  const int g = grid_for((long long)Hp * Wp * Cpad);
  if (is_uint8)
    SOD_LAUNCH(preprocess_f32_kernel<uint8_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)img, C, H, W, out, Hp, Wp, Cpad, mean3[0],
               mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  else
    SOD_LAUNCH(preprocess_f32_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)img, C, H, W, out, Hp, Wp, Cpad, mean3[0], mean3[1],
               mean3[2], std3[0], std3[1], std3[2]);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_weight_prep_f32(const float* w, const float* scale, float* w_krsc, float* w_crsk, int K, int RS, int C, int Cpad, void* stream) {
  if (!w || (!w_krsc && !w_crsk) || K <= 0 || RS <= 0 || C <= 0 || Cpad < C) return SOD_EARG;
  SOD_LAUNCH(weight_prep_f32_kernel, dim3(grid_for((long long)K * RS * Cpad)), dim3(256), 0, (hipStream_t)stream, w, scale, w_krsc, w_crsk, K, RS, C,
             Cpad);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}
