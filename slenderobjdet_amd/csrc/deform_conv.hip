// Deformable convolution v1 / v2 (DCN), forward and backward, NHWC bf16.
//
// Replaces detectron2.layers.DeformConv / ModulatedDeformConv (CUDA-only csrc/deformable/*, absent from the reference
// tree; sampling rule restated in SURVEY.md Appendix C.11) as used by slender_det/layers/df_conv.py:6-78 (DFConv2d),
// reppoints/rpd.py:147-154,637-642 and the meta heads.
//
// Decomposition (round 1): the bilinear gather is a bandwidth-bound kernel that writes the sampled columns
// cols[n,ho,wo,(g,tap,c)] in bf16; the contraction with the weights, its data-gradient and its weight-gradient then run as
// 1x1 convolutions on the MFMA implicit-GEMM kernels (conv_igemm.hip) over 9*C "channels".  The column buffer is
// transient (recomputed in backward, never saved).  Fusing the gather into the GEMM's LDS staging is the next step.
//
// Offsets are NHWC fp32: off[n,ho,wo, 2*k] = dy, off[..., 2*k+1] = dx for k = (g*KH + i)*KW + j; mask[n,ho,wo,k] (v2).
#include "common.h"
#include "../../include/slender_hip.h"

namespace {

struct DcnArgs {
  const __bf16* x;        // (N,H,W,C)
  const float* off;       // (N,Ho,Wo,2*KH*KW*DG)
  const float* mask;      // (N,Ho,Wo,KH*KW*DG) or null
  __bf16* cols;           // (N,Ho,Wo,KH*KW*C)  channel index = tap*C + c   (tap = i*KW + j; deformable group = c / (C/DG))
  const __bf16* dcols;    // backward
  float* dx;              // (N,H,W,C) fp32, atomically accumulated
  float* doff;            // like off
  float* dmask;           // like mask
  int N, H, W, C, Ho, Wo, KH, KW, stride, pad, dil, DG;
  int off_ld, mask_ld;    // row pitch (elements) of offset / mask (and of their gradients)
  int mask_logit;         // mask holds logits: m = sigmoid(mask) here, dmask is the gradient w.r.t. the logit
};

struct Samp {
  bool valid;
  int yl, xl, yh, xh;
  float w00, w01, w10, w11, ly, lx;
  bool ok00, ok01, ok10, ok11;
};

__device__ __forceinline__ Samp make_samp(float py, float px, int H, int W) {
  Samp s;
  s.valid = (py > -1.f) && (px > -1.f) && (py < (float)H) && (px < (float)W);
  const float fy = floorf(py), fx = floorf(px);
  s.yl = (int)fy; s.xl = (int)fx; s.yh = s.yl + 1; s.xh = s.xl + 1;
  s.ly = py - fy; s.lx = px - fx;
  const float hy = 1.f - s.ly, hx = 1.f - s.lx;
  s.w00 = hy * hx; s.w01 = hy * s.lx; s.w10 = s.ly * hx; s.w11 = s.ly * s.lx;
  s.ok00 = s.valid && s.yl >= 0 && s.xl >= 0;
  s.ok01 = s.valid && s.yl >= 0 && s.xh <= W - 1;
  s.ok10 = s.valid && s.yh <= H - 1 && s.xl >= 0;
  s.ok11 = s.valid && s.yh <= H - 1 && s.xh <= W - 1;
  return s;
}

// work item = (pixel, tap, 8-channel vector); the 8-channel vectors of one (pixel, tap) are consecutive threads
__global__ __launch_bounds__(256) void dcn_im2col_kernel(const DcnArgs a) {
  const int c8n = a.C >> 3, taps = a.KH * a.KW, cpg = a.C / a.DG;
  const long long total = (long long)a.N * a.Ho * a.Wo * taps * c8n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c8 = (int)(i % c8n);
    long long t = i / c8n;
    const int tap = (int)(t % taps); t /= taps;
    const int wo = (int)(t % a.Wo); t /= a.Wo;
    const int ho = (int)(t % a.Ho);
    const int n = (int)(t / a.Ho);
    const int g = (c8 * 8) / cpg;
    const int k = g * taps + tap;
    const long long pix = ((long long)n * a.Ho + ho) * a.Wo + wo;
    const float dy = a.off[pix * a.off_ld + 2 * k], dxo = a.off[pix * a.off_ld + 2 * k + 1];
    const int ki = tap / a.KW, kj = tap - ki * a.KW;
    const Samp s = make_samp((float)(ho * a.stride - a.pad + ki * a.dil) + dy, (float)(wo * a.stride - a.pad + kj * a.dil) + dxo, a.H, a.W);
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long base = (long long)n * a.H * a.W;
    auto add = [&](bool ok, int yy, int xx, float w) {
      if (!ok) return;
      const bf16x8_t q = *reinterpret_cast<const bf16x8_t*>(a.x + ((base + (long long)yy * a.W + xx) * a.C) + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += w * (float)q[e];
    };
    add(s.ok00, s.yl, s.xl, s.w00); add(s.ok01, s.yl, s.xh, s.w01); add(s.ok10, s.yh, s.xl, s.w10); add(s.ok11, s.yh, s.xh, s.w11);
    float m = a.mask ? a.mask[pix * a.mask_ld + k] : 1.f;
    if (a.mask && a.mask_logit) m = 1.f / (1.f + expf(-m));
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)(v[e] * m);
    *reinterpret_cast<bf16x8_t*>(a.cols + (pix * taps + tap) * a.C + c8 * 8) = o;
  }
}

// backward of the gather: dX (atomic, fp32), dOffset, dMask.  Reduction over channels of one (pixel, tap, group) runs over
// `red` consecutive lanes with shuffles (red = (C/DG)/8, a power of two <= 64), else falls back to atomics.
__global__ __launch_bounds__(256) void dcn_col2im_kernel(const DcnArgs a, int red) {
  const int c8n = a.C >> 3, taps = a.KH * a.KW, cpg = a.C / a.DG;
  const long long total = (long long)a.N * a.Ho * a.Wo * taps * c8n;
  const long long padded = (total + 255) / 256 * 256;   // keep whole waves active for the shuffles
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < padded; i += (long long)gridDim.x * 256) {
    const bool live = i < total;
    float g_dy = 0.f, g_dx = 0.f, g_m = 0.f;
    long long pix = 0; int k = 0;
    if (live) {
      const int c8 = (int)(i % c8n);
      long long t = i / c8n;
      const int tap = (int)(t % taps); t /= taps;
      const int wo = (int)(t % a.Wo); t /= a.Wo;
      const int ho = (int)(t % a.Ho);
      const int n = (int)(t / a.Ho);
      const int g = (c8 * 8) / cpg;
      k = g * taps + tap;
      pix = ((long long)n * a.Ho + ho) * a.Wo + wo;
      const float dy = a.off[pix * a.off_ld + 2 * k], dxo = a.off[pix * a.off_ld + 2 * k + 1];
      const int ki = tap / a.KW, kj = tap - ki * a.KW;
      const Samp s = make_samp((float)(ho * a.stride - a.pad + ki * a.dil) + dy, (float)(wo * a.stride - a.pad + kj * a.dil) + dxo, a.H, a.W);
      float m = a.mask ? a.mask[pix * a.mask_ld + k] : 1.f;
      if (a.mask && a.mask_logit) m = 1.f / (1.f + expf(-m));
      const bf16x8_t dcv = *reinterpret_cast<const bf16x8_t*>(a.dcols + (pix * taps + tap) * a.C + c8 * 8);
      const long long base = (long long)n * a.H * a.W;
      float v00[8], v01[8], v10[8], v11[8];
      auto ld = [&](bool ok, int yy, int xx, float* dst) {
        if (ok) {
          const bf16x8_t q = *reinterpret_cast<const bf16x8_t*>(a.x + ((base + (long long)yy * a.W + xx) * a.C) + c8 * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) dst[e] = (float)q[e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) dst[e] = 0.f;
        }
      };
      ld(s.ok00, s.yl, s.xl, v00); ld(s.ok01, s.yl, s.xh, v01); ld(s.ok10, s.yh, s.xl, v10); ld(s.ok11, s.yh, s.xh, v11);
      const float hy = 1.f - s.ly, hx = 1.f - s.lx;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (float)dcv[e];
        const float dm = d * m;     // gradient w.r.t. the un-masked sample
        if (s.ok00) atomicAdd(a.dx + ((base + (long long)s.yl * a.W + s.xl) * a.C) + c8 * 8 + e, dm * s.w00);
        if (s.ok01) atomicAdd(a.dx + ((base + (long long)s.yl * a.W + s.xh) * a.C) + c8 * 8 + e, dm * s.w01);
        if (s.ok10) atomicAdd(a.dx + ((base + (long long)s.yh * a.W + s.xl) * a.C) + c8 * 8 + e, dm * s.w10);
        if (s.ok11) atomicAdd(a.dx + ((base + (long long)s.yh * a.W + s.xh) * a.C) + c8 * 8 + e, dm * s.w11);
        // d(sample)/d(py) = hx*(v10 - v00) + lx*(v11 - v01) ; d/d(px) = hy*(v01 - v00) + ly*(v11 - v10)
        g_dy += dm * (hx * (v10[e] - v00[e]) + s.lx * (v11[e] - v01[e]));
        g_dx += dm * (hy * (v01[e] - v00[e]) + s.ly * (v11[e] - v10[e]));
        g_m += d * (s.w00 * v00[e] + s.w01 * v01[e] + s.w10 * v10[e] + s.w11 * v11[e]);
      }
      if (!s.valid) { g_dy = 0.f; g_dx = 0.f; g_m = 0.f; }
      if (a.mask && a.mask_logit) g_m *= m * (1.f - m);
    }
    if (red > 0) {
      for (int o = red >> 1; o > 0; o >>= 1) {
        g_dy += __shfl_xor(g_dy, o, 64); g_dx += __shfl_xor(g_dx, o, 64); g_m += __shfl_xor(g_m, o, 64);
      }
      if (live && ((threadIdx.x & (red - 1)) == 0)) {
        a.doff[pix * a.off_ld + 2 * k] = g_dy;
        a.doff[pix * a.off_ld + 2 * k + 1] = g_dx;
        if (a.dmask) a.dmask[pix * a.mask_ld + k] = g_m;
      }
    } else if (live) {
      atomicAdd(a.doff + pix * a.off_ld + 2 * k, g_dy);
      atomicAdd(a.doff + pix * a.off_ld + 2 * k + 1, g_dx);
      if (a.dmask) atomicAdd(a.dmask + pix * a.mask_ld + k, g_m);
    }
  }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y, long long n8) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const f32x4_t a = reinterpret_cast<const f32x4_t*>(x)[i * 2], b = reinterpret_cast<const f32x4_t*>(x)[i * 2 + 1];
    bf16x8_t o = {(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
    reinterpret_cast<bf16x8_t*>(y)[i] = o;
  }
}

int dcn_fill(DcnArgs& a, int N, int H, int W, int C, int KH, int KW, int stride, int pad, int dil, int DG, int off_ld, int mask_ld,
             int mask_logit) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || dil <= 0 || pad < 0 || DG <= 0) return SOD_EARG;
  if ((C & 7) || C % DG || ((C / DG) & 7)) return SOD_EARG;
  a.N = N; a.H = H; a.W = W; a.C = C; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.dil = dil; a.DG = DG;
  a.Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  a.Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  if (a.Ho <= 0 || a.Wo <= 0) return SOD_EARG;
  a.off_ld = off_ld > 0 ? off_ld : 2 * KH * KW * DG;
  a.mask_ld = mask_ld > 0 ? mask_ld : KH * KW * DG;
  a.mask_logit = mask_logit;
  if (a.off_ld < 2 * KH * KW * DG || a.mask_ld < KH * KW * DG) return SOD_EARG;
  return SOD_OK;
}

inline int grid_for(long long n) {
  long long g = (n + 255) / 256;
  if (g > 16384) g = 16384;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int sod_deform_im2col(const void* x, const float* offset, const float* mask, void* cols,
                                 int N, int H, int W, int C, int KH, int KW, int stride, int pad, int dil, int deformable_groups,
                                 int off_ld, int mask_ld, int mask_is_logit, void* stream) {
  if (!x || !offset || !cols) return SOD_EARG;
  DcnArgs a{};
  int rc = dcn_fill(a, N, H, W, C, KH, KW, stride, pad, dil, deformable_groups, off_ld, mask_ld, mask_is_logit);
  if (rc) return rc;
  a.x = (const __bf16*)x; a.off = offset; a.mask = mask; a.cols = (__bf16*)cols;
  SOD_LAUNCH(dcn_im2col_kernel, dim3(grid_for((long long)N * a.Ho * a.Wo * KH * KW * (C / 8))), dim3(256), 0, (hipStream_t)stream, a);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_deform_col2im(const void* dcols, const void* x, const float* offset, const float* mask,
                                 float* dx_f32, float* doffset, float* dmask,
                                 int N, int H, int W, int C, int KH, int KW, int stride, int pad, int dil, int deformable_groups,
                                 int off_ld, int mask_ld, int mask_is_logit, void* stream) {
  if (!dcols || !x || !offset || !dx_f32 || !doffset || (mask && !dmask)) return SOD_EARG;
  DcnArgs a{};
  int rc = dcn_fill(a, N, H, W, C, KH, KW, stride, pad, dil, deformable_groups, off_ld, mask_ld, mask_is_logit);
  if (rc) return rc;
  a.x = (const __bf16*)x; a.off = offset; a.mask = mask; a.dcols = (const __bf16*)dcols; a.dx = dx_f32; a.doff = doffset; a.dmask = dmask;
  hipStream_t st = (hipStream_t)stream;
  const int per = (C / deformable_groups) / 8;
  int red = 0;
  if (per <= 64 && (per & (per - 1)) == 0 && (C / 8) % per == 0) red = per;
  // the atomic fallback (red == 0) accumulates: the caller passes zero-initialised doffset / dmask in every case (they are
  // pitched buffers whose padding columns must be zero anyway)
  SOD_LAUNCH(dcn_col2im_kernel, dim3(grid_for((long long)N * a.Ho * a.Wo * KH * KW * (C / 8))), dim3(256), 0, st, a, red);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_f32_to_bf16(const float* x, void* y, long long n, void* stream) {
  if (!x || !y || n < 0 || (n & 7)) return SOD_EARG;
  SOD_LAUNCH(f32_to_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, x, (__bf16*)y, n / 8);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}
