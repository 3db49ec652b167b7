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
#include <stdlib.h>

namespace {

struct DcnArgs {
  const __bf16* x;        // (N,H,W,C)   (x / cols / dcols hold fp32 for the <float> instantiations of the plain kernels: validation mode)
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
  unsigned long long* oow; // optional counter (sod_deform_conv_set_window_counter): sample lanes of the tiled backward kernels whose bilinear
                           // footprint left the LDS window and took the global float-atomic path
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

// 8 consecutive channels of element type T (bf16: the product path; float: the fp32 validation mode, SOD_PRECISION=fp32) as floats
template <typename T>
__device__ __forceinline__ void load8(const void* base, long long idx, float (&v)[8]) {
  if constexpr (sizeof(T) == 2) {
    const bf16x8_t q = *reinterpret_cast<const bf16x8_t*>((const __bf16*)base + idx);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)q[e];
  } else {
    const f32x4_t q0 = *reinterpret_cast<const f32x4_t*>((const float*)base + idx), q1 = *reinterpret_cast<const f32x4_t*>((const float*)base + idx + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = q0[e]; v[4 + e] = q1[e]; }
  }
}

template <typename T>
__device__ __forceinline__ void store8(void* base, long long idx, const float (&v)[8]) {
  if constexpr (sizeof(T) == 2) {
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
    *reinterpret_cast<bf16x8_t*>((__bf16*)base + idx) = o;
  } else {
    *reinterpret_cast<f32x4_t*>((float*)base + idx) = f32x4_t{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4_t*>((float*)base + idx + 4) = f32x4_t{v[4], v[5], v[6], v[7]};
  }
}

// work item = (pixel, tap, 8-channel vector); the 8-channel vectors of one (pixel, tap) are consecutive threads
template <typename T>
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
      float q[8];
      load8<T>(a.x, ((base + (long long)yy * a.W + xx) * a.C) + c8 * 8, q);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += w * q[e];
    };
    add(s.ok00, s.yl, s.xl, s.w00); add(s.ok01, s.yl, s.xh, s.w01); add(s.ok10, s.yh, s.xl, s.w10); add(s.ok11, s.yh, s.xh, s.w11);
    float m = a.mask ? a.mask[pix * a.mask_ld + k] : 1.f;
    if (a.mask && a.mask_logit) m = 1.f / (1.f + expf(-m));
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= m;
    store8<T>(a.cols, (pix * taps + tap) * a.C + c8 * 8, v);
  }
}

// backward of the gather: dX (atomic, fp32), dOffset, dMask.  Reduction over channels of one (pixel, tap, group) runs over
// `red` consecutive lanes with shuffles (red = (C/DG)/8, a power of two <= 64), else falls back to atomics.
template <typename T>
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
      float dcv[8];
      load8<T>(a.dcols, (pix * taps + tap) * a.C + c8 * 8, dcv);
      const long long base = (long long)n * a.H * a.W;
      float v00[8], v01[8], v10[8], v11[8];
      auto ld = [&](bool ok, int yy, int xx, float (&dst)[8]) {
        if (ok) {
          load8<T>(a.x, ((base + (long long)yy * a.W + xx) * a.C) + c8 * 8, dst);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) dst[e] = 0.f;
        }
      };
      ld(s.ok00, s.yl, s.xl, v00); ld(s.ok01, s.yl, s.xh, v01); ld(s.ok10, s.yh, s.xl, v10); ld(s.ok11, s.yh, s.xh, v11);
      const float hy = 1.f - s.ly, hx = 1.f - s.lx;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = dcv[e];
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

// Tiled backward of the gather.  The scatter of the plain kernel above costs 4 corners x 8 channels global atomics per work
// item (3.3 G per call at 16 x 800x1344, and all nine taps of a pixel hit the SAME address when the learned offsets cancel the
// kernel grid, which is exactly RepPoints' initial state).  Here a workgroup owns an 8x8 tile of output pixels x CC channels and
// accumulates dX in an LDS window covering the tile's receptive field plus R pixels of slack; samples that land outside the
// window fall back to global atomics.  The window is flushed once (zeros skipped).  dOffset / dMask are reduced over the CC/8
// lanes of a (pixel, tap) and added atomically (C/CC workgroups contribute; the caller zero-fills them).
//
// The window accumulates in 32-bit FIXED POINT with ds_add_u32: measured on MI355X (tools/micro/lds_atomic.hip), ds_add_f32 runs
// at 0.33 lanes/clk/CU (204 G/s chip-wide) while ds_add_u32 runs at 13.7 lanes/clk/CU (8.4 T/s), 40x faster.  The scale is a
// power of two chosen per workgroup from max|dcols * mask| of its tile so that the <= 576 contributions an element can receive
// cannot overflow: quantum = 2^-20 of the tile maximum, far below the bf16 rounding dX gets afterwards, and the sum is
// order-independent (bit-reproducible within a window).
template <int CC>
__global__ __launch_bounds__(256) void dcn_col2im_tile_kernel(const DcnArgs a, int tiles_x, int WH, int WW, int R) {
  extern __shared__ int win[];     // [WH][WW][PS], PS = CC + 1: pixel rows start on rotating LDS banks (CC is a multiple of the bank count)
  __shared__ float smax[4];
  constexpr int PS = CC + 1;
  constexpr int L = CC / 8;        // lanes per (pixel, tap)
  constexpr int PPI = 256 / L;     // pixels per pass
  const int tid = threadIdx.x;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int c0 = blockIdx.y * CC, n = blockIdx.z;
  const int taps = a.KH * a.KW, cpg = a.C / a.DG;
  const int ho0 = ty * 8, wo0 = tx * 8;
  const int wy0 = ho0 * a.stride - a.pad - R, wx0 = wo0 * a.stride - a.pad - R;
  const int wsize = WH * WW * CC;
  for (int i = tid; i < WH * WW * PS; i += 256) win[i] = 0;
  const int cl = tid % L, pl = tid / L;
  const int cch = c0 + cl * 8;            // first of this lane's 8 channels
  const int g = cch / cpg;
  const long long base = (long long)n * a.H * a.W;
  // pass 1: fixed-point scale of this workgroup = 2^(20 - ex) with max|d * m| < 2^ex
  float dmax = 0.f;
  for (int p = pl; p < 64; p += PPI) {
    const int ho = ho0 + (p >> 3), wo = wo0 + (p & 7);
    if (ho >= a.Ho || wo >= a.Wo) continue;
    const long long pix = ((long long)n * a.Ho + ho) * a.Wo + wo;
    for (int tap = 0; tap < taps; ++tap) {
      float m = a.mask ? a.mask[pix * a.mask_ld + g * taps + tap] : 1.f;
      if (a.mask && a.mask_logit) m = 1.f / (1.f + expf(-m));
      const bf16x8_t dcv = *reinterpret_cast<const bf16x8_t*>(a.dcols + (pix * taps + tap) * a.C + cch);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = (float)dcv[e] * m;
        dmax = fmaxf(dmax, fabsf(v));      // fmaxf drops NaNs: they are tracked separately (a NaN gradient must stay a NaN)
        if (v != v) dmax = __builtin_inff();
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, o, 64));
  if ((tid & 63) == 0) smax[tid >> 6] = dmax;
  __syncthreads();
  dmax = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
  int ex = 0;
  (void)frexpf(dmax, &ex);
  // fixed-point quantum: a window cell receives at most 64 pixels x taps contributions of magnitude < 2^ex each, and their sum
  // must fit an int32: 2^fbits * 64 * taps <= 2^30  (3x3: fbits = 20)
  int fbits = 30;
  for (int c = 64 * taps - 1; c > 0; c >>= 1) --fbits;
  const float S = ldexpf(1.f, fbits - ex), invS = ldexpf(1.f, ex - fbits);
  if (dmax == 0.f) return;   // all-zero gradient tile (e.g. the regression branch away from the few positive locations): nothing to add
  const bool finite_scale = dmax < 3.0e38f;   // inf / NaN in the tile: the float atomic path propagates them
  for (int p = pl; p < 64; p += PPI) {
    const int ho = ho0 + (p >> 3), wo = wo0 + (p & 7);
    const bool live = ho < a.Ho && wo < a.Wo;
    const long long pix = ((long long)n * a.Ho + ho) * a.Wo + wo;
    for (int tap = 0; tap < taps; ++tap) {
      const int k = g * taps + tap;
      float g_dy = 0.f, g_dx = 0.f, g_m = 0.f;
      if (live) {
        const float dy = a.off[pix * a.off_ld + 2 * k], dxo = a.off[pix * a.off_ld + 2 * k + 1];
        const int ki = tap / a.KW, kj = tap - ki * a.KW;
        const Samp s = make_samp((float)(ho * a.stride - a.pad + ki * a.dil) + dy, (float)(wo * a.stride - a.pad + kj * a.dil) + dxo, a.H, a.W);
        float m = a.mask ? a.mask[pix * a.mask_ld + k] : 1.f;
        if (a.mask && a.mask_logit) m = 1.f / (1.f + expf(-m));
        const bf16x8_t dcv = *reinterpret_cast<const bf16x8_t*>(a.dcols + (pix * taps + tap) * a.C + cch);
        const s16x8_t dbits = __builtin_bit_cast(s16x8_t, dcv);
        bool any = false;
#pragma unroll
        for (int e = 0; e < 8; ++e) any = any || ((dbits[e] & 0x7fff) != 0);
        if (s.valid && any) {
          float v00[8], v01[8], v10[8], v11[8];
          auto ld = [&](bool ok, int yy, int xx, float* dst) {
            if (ok) {
              const bf16x8_t q = *reinterpret_cast<const bf16x8_t*>(a.x + ((base + (long long)yy * a.W + xx) * a.C) + cch);
#pragma unroll
              for (int e = 0; e < 8; ++e) dst[e] = (float)q[e];
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) dst[e] = 0.f;
            }
          };
          ld(s.ok00, s.yl, s.xl, v00); ld(s.ok01, s.yl, s.xh, v01); ld(s.ok10, s.yh, s.xl, v10); ld(s.ok11, s.yh, s.xh, v11);
          const float hy = 1.f - s.ly, hx = 1.f - s.lx;
          const int wy = s.yl - wy0, wx = s.xl - wx0;
          const bool inwin0 = wy >= 0 && wx >= 0 && wy + 1 < WH && wx + 1 < WW;
          const bool inwin = inwin0 && finite_scale;
          if (!inwin0 && a.oow) atomicAdd(a.oow, 1ull);      // (the slow path: 32 global float atomics follow)
          int* w00p = win + ((wy * WW + wx) * PS) + cl * 8;
          float* g00p = a.dx + ((base + (long long)s.yl * a.W + s.xl) * a.C) + cch;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = (float)dcv[e];
            const float dm = d * m;
            if (inwin) {
              const float ds = dm * S;
              if (s.ok00) atomicAdd(w00p + e, __float2int_rn(ds * s.w00));
              if (s.ok01) atomicAdd(w00p + PS + e, __float2int_rn(ds * s.w01));
              if (s.ok10) atomicAdd(w00p + WW * PS + e, __float2int_rn(ds * s.w10));
              if (s.ok11) atomicAdd(w00p + WW * PS + PS + e, __float2int_rn(ds * s.w11));
            } else {
              if (s.ok00) atomicAdd(g00p + e, dm * s.w00);
              if (s.ok01) atomicAdd(g00p + a.C + e, dm * s.w01);
              if (s.ok10) atomicAdd(g00p + (long long)a.W * a.C + e, dm * s.w10);
              if (s.ok11) atomicAdd(g00p + (long long)a.W * a.C + a.C + e, dm * s.w11);
            }
            g_dy += dm * (hx * (v10[e] - v00[e]) + s.lx * (v11[e] - v01[e]));
            g_dx += dm * (hy * (v01[e] - v00[e]) + s.ly * (v11[e] - v10[e]));
            g_m += d * (s.w00 * v00[e] + s.w01 * v01[e] + s.w10 * v10[e] + s.w11 * v11[e]);
          }
          if (a.mask && a.mask_logit) g_m *= m * (1.f - m);
        }
      }
#pragma unroll
      for (int o = L >> 1; o > 0; o >>= 1) {
        g_dy += __shfl_xor(g_dy, o, 64); g_dx += __shfl_xor(g_dx, o, 64); g_m += __shfl_xor(g_m, o, 64);
      }
      if (live && cl == 0) {
        atomicAdd(a.doff + pix * a.off_ld + 2 * k, g_dy);
        atomicAdd(a.doff + pix * a.off_ld + 2 * k + 1, g_dx);
        if (a.dmask) atomicAdd(a.dmask + pix * a.mask_ld + k, g_m);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < wsize; i += 256) {
    const int c = i % CC, r = i / CC;
    const int q = win[r * PS + c];
    if (q != 0) {
      const float v = (float)q * invS;
      const int wx = r % WW, wy = r / WW;
      atomicAdd(a.dx + ((base + (long long)(wy0 + wy) * a.W + (wx0 + wx)) * a.C) + c0 + c, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Fused backward w.r.t. input / offsets / mask WITHOUT the column-gradient buffer.  dcols = dY x W^T was a 1x1 data gradient writing a
// (N*Ho*Wo, KH*KW*C) bf16 tensor to HBM (1.24 GB for the P3 level of RepPoints at batch 16) that dcn_col2im_tile_kernel read back.
// Here the workgroup that owns an 8x8 tile of output pixels x CC = 32 input channels computes its slice of dcols itself, tap by tap,
// on the matrix cores: A = the tile's dY rows (64 px x K, held in registers as MFMA fragments for all taps), B = W^T[(tap, c0..c0+31)]
// (32 rows of K contiguous bf16 in the CRSK copy: 16 KB at K = 256, staged through LDS, the next tap's rows prefetched into registers
// during the scatter), D = 64 px x 32 ch fp32 -> rounded to bf16 into an LDS tile (the value the column buffer used to hold) ->
// the scatter of dcn_col2im_tile_kernel (bilinear weights, LDS fixed-point window for dX, offset / mask gradients reduced over the
// channel lanes) runs on that tile.  The fixed-point scale needs max|dcols * mask| BEFORE the first tap is scattered; it is bounded by
// Cauchy-Schwarz: |dcols[p, tap, c]| <= ||dY[p, :]|| * ||W[:, tap, c]|| (column norms from dcn_wnorm_kernel), typically a few times the
// true maximum, i.e. the quantum is 2^-17 .. 2^-20 of the tile maximum instead of 2^-20: still far below the bf16 rounding of dX.
__global__ __launch_bounds__(256) void dcn_wnorm_kernel(const __bf16* __restrict__ wt, int rows, int K, float* __restrict__ out) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float s = 0.f;
  for (int k = 0; k < K; k += 8) {
    const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(wt + (long long)r * K + k);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += (float)v[e] * (float)v[e];
  }
  out[r] = sqrtf(s);
}

// KS: K = 32 * KS output channels of the convolution = contraction length of the dcols GEMM.  GM: the mask gradient is wanted (modulated
// DeformConv); without it - RepPoints' plain DeformConv - a ninth of the scatter's VALU work is not compiled in.
// STG: the offset gradients of the tile are staged in LDS ([64 px][2 * taps] floats behind the dcols tile) and leave in one coalesced pass
// after the tap loop instead of two scattered 4-byte atomics per (pixel, tap).  Measured on the P3 level of RepPoints (us per launch at
// offset spreads 0.5 / 4 / 8 px): 2 px of slack 2474 / 4485 / 6993 direct, 2760 / 3693 / 5596 staged; 4 px of slack 3008 / 3866 / 6218 direct,
// 2848 / 3608 / 4642 staged - the flush is a serial tail that only the two-workgroup configuration hides, so the launcher stages from 4 px
// of slack on (where layers/deform_conv.py::_WindowPolicy goes once offsets leave the window).
template <int KS, bool GM, bool STG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KS <= 8 ? 3 : 1))) void dcn_bwd_fused_kernel(const DcnArgs a, const __bf16* __restrict__ dy, const __bf16* __restrict__ wt,
                                                            const float* __restrict__ wnorm, int tiles_x, int WH, int WW, int R) {
  constexpr int CC = 32, PS = CC + 1, L = CC / 8, K = 32 * KS;
  constexpr int WROW = K * 2 + 16;        // bytes per staged weight row (16 B pad: the 16 rows of a fragment read start on different banks)
  constexpr int NPIECE = KS / 2;          // 16-B pieces of the weight chunk per thread (32 rows x K bf16 / 256 threads)
  extern __shared__ int win[];            // [WH][WW][PS] fixed-point window, then the weight chunk, then the dcols tile
  __shared__ float smax[8];
  __shared__ unsigned char slow_lane[4][64];   // per wave: the lanes whose sample fell outside the window, compacted (see the scatter)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int c0 = blockIdx.y * CC, n = blockIdx.z;
  const int taps = a.KH * a.KW, cpg = a.C / a.DG;
  const int ho0 = ty * 8, wo0 = tx * 8;
  const int wy0 = ho0 * a.stride - a.pad - R, wx0 = wo0 * a.stride - a.pad - R;
  const int wsize = WH * WW * CC;
  const int win_bytes = (WH * WW * PS * 4 + 15) & ~15;
  char* wl = reinterpret_cast<char*>(win) + win_bytes;
  __bf16* dcl = reinterpret_cast<__bf16*>(wl + 32 * WROW);
  float* dofl = reinterpret_cast<float*>(dcl + 64 * CC);      // STG only (the launcher sizes the allocation)
  for (int i = tid; i < WH * WW * PS; i += 256) win[i] = 0;

  // ---- A fragments: dY rows of this wave's 16 pixels, all K channels (lane: pixel = lane % 16, channels (lane / 16) * 8 .. + 8 of a step)
  const int pa = wave * 16 + (lane & 15);
  const int hoa = ho0 + (pa >> 3), woa = wo0 + (pa & 7);
  const bool alive = hoa < a.Ho && woa < a.Wo;
  const long long pixa = ((long long)n * a.Ho + hoa) * a.Wo + woa;
  bf16x8_t af[KS];
  float ss = 0.f;
  bool bad = false;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8_t v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
    if (alive) v = *reinterpret_cast<const bf16x8_t*>(dy + pixa * K + ks * 32 + (lane >> 4) * 8);
    af[ks] = v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float f = (float)v[e];
      ss += f * f;
      bad = bad || (f != f) || (fabsf(f) > 3.0e38f);
    }
  }
  // the weight rows of tap 0 start to arrive while the scale is worked out
  bf16x8_t wreg[NPIECE];
  auto w_issue = [&](int tap) {
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      const int q = i * 256 + tid, row = q / (K / 8), col = q % (K / 8);
      wreg[i] = *reinterpret_cast<const bf16x8_t*>(wt + ((long long)(tap * a.C + c0 + row)) * K + col * 8);
    }
  };
  auto w_store = [&]() {
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      const int q = i * 256 + tid, row = q / (K / 8), col = q % (K / 8);
      *reinterpret_cast<bf16x8_t*>(wl + row * WROW + col * 16) = wreg[i];
    }
  };
  w_issue(0);
  // max over the tile of ||dY[p]||, of the column norms of this channel chunk, and of |mask|
  ss += __shfl_xor(ss, 16, 64);
  ss += __shfl_xor(ss, 32, 64);
  float nmax = bad ? __builtin_inff() : sqrtf(ss);
  float wmax = 0.f, mmax = a.mask ? 0.f : 1.f;
  for (int i = tid; i < taps * CC; i += 256) wmax = fmaxf(wmax, wnorm[(i / CC) * a.C + c0 + (i % CC)]);
  if (a.mask) {
    const int g0 = c0 / cpg;
    for (int i = tid; i < 64 * taps; i += 256) {
      const int p = i / taps, tap = i - p * taps;
      const int ho = ho0 + (p >> 3), wo = wo0 + (p & 7);
      if (ho < a.Ho && wo < a.Wo) {
        float m = a.mask[(((long long)n * a.Ho + ho) * a.Wo + wo) * a.mask_ld + g0 * taps + tap];
        if (a.mask_logit) m = 1.f / (1.f + expf(-m));
        mmax = fmaxf(mmax, fabsf(m));
        if (m != m) mmax = __builtin_inff();
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    nmax = fmaxf(nmax, __shfl_xor(nmax, o, 64));
    wmax = fmaxf(wmax, __shfl_xor(wmax, o, 64));
    mmax = fmaxf(mmax, __shfl_xor(mmax, o, 64));
  }
  if (lane == 0) { smax[wave] = nmax; smax[4 + wave] = wmax * mmax; }
  __syncthreads();
  nmax = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
  const float wm = fmaxf(fmaxf(smax[4], smax[5]), fmaxf(smax[6], smax[7]));
  if (nmax == 0.f) return;     // all-zero gradient tile (the regression branch away from the few positive locations): nothing to add
  const float dmax = nmax * wm;
  int ex = 0;
  (void)frexpf(dmax, &ex);
  int fbits = 30;
  for (int c = 64 * taps - 1; c > 0; c >>= 1) --fbits;
  const float S = ldexpf(1.f, fbits - ex), invS = ldexpf(1.f, ex - fbits);
  const bool finite_scale = dmax < 3.0e38f && dmax > 0.f;   // inf / NaN in the tile: the float atomic path propagates them
  w_store();
  __syncthreads();

  const int cl = tid % L, pl = tid / L;           // scatter phase: 4 lanes x 8 channels per pixel, the 64 pixels of the tile at once
  const int cch = c0 + cl * 8;
  const int g = cch / cpg;
  const long long base = (long long)n * a.H * a.W;
  const int ho = ho0 + (pl >> 3), wo = wo0 + (pl & 7);
  const bool live = ho < a.Ho && wo < a.Wo;
  const long long pix = ((long long)n * a.Ho + ho) * a.Wo + wo;
  for (int tap = 0; tap < taps; ++tap) {
    // this tap's offsets (and mask value) are requested BEFORE the matrix-core phase and consumed after its barrier: their latency - the
    // first of two dependent memory round trips of the scatter (offsets -> corner addresses -> corner loads) - hides behind the MFMAs
    float dyo = 0.f, dxo = 0.f, mraw = 1.f;
    if (live) {
      const int kq = g * taps + tap;
      dyo = a.off[pix * a.off_ld + 2 * kq]; dxo = a.off[pix * a.off_ld + 2 * kq + 1];
      if (a.mask) mraw = a.mask[pix * a.mask_ld + kq];
    }
    // ---- dcols tile of this tap on the matrix cores: 16 px (this wave) x 32 ch
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(wl + (lane & 15) * WROW + (ks * 32 + (lane >> 4) * 8) * 2);
      const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(wl + (16 + (lane & 15)) * WROW + (ks * 32 + (lane >> 4) * 8) * 2);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], b1, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                  // D[m = (lane / 16) * 4 + i][n = lane % 16]
      __bf16* row = dcl + (wave * 16 + (lane >> 4) * 4 + i) * CC;
      row[lane & 15] = (__bf16)acc0[i];
      row[16 + (lane & 15)] = (__bf16)acc1[i];
    }
    if (tap + 1 < taps) w_issue(tap + 1);
    __syncthreads();                               // the tile is complete; every wave is done with this tap's weight rows
    // ---- scatter (dcn_col2im_tile_kernel's body, with the column gradients read from the LDS tile)
    const int k = g * taps + tap;
    float g_dy = 0.f, g_dx = 0.f, g_m = 0.f;
    // a lane whose sample fell outside the window parks what the scatter needs here and the WAVE adds it after the branch (below)
    // (four registers: the kernel sits at 168 VGPRs = three workgroups per CU, and 176 measured 21 % slower on in-window offsets)
    bool slow = false, outw = false;
    unsigned s_pk = 0;                 // (xl + 1) << 18 | (yl + 1) << 4 | corner bits; H, W < 16383 is checked by the launcher
    float s_m = 0.f, s_ly = 0.f, s_lx = 0.f;
    if (live) {
      const int ki = tap / a.KW, kj = tap - ki * a.KW;
      const Samp s = make_samp((float)(ho * a.stride - a.pad + ki * a.dil) + dyo, (float)(wo * a.stride - a.pad + kj * a.dil) + dxo, a.H, a.W);
      float m = mraw;
      if (a.mask && a.mask_logit) m = 1.f / (1.f + expf(-m));
      const bf16x8_t dcv = *reinterpret_cast<const bf16x8_t*>(dcl + pl * CC + cl * 8);
      const s16x8_t dbits = __builtin_bit_cast(s16x8_t, dcv);
      bool any = false;
#pragma unroll
      for (int e = 0; e < 8; ++e) any = any || ((dbits[e] & 0x7fff) != 0);
      if (s.valid && any) {
        float v00[8], v01[8], v10[8], v11[8];
        auto ld = [&](bool ok, int yy, int xx, float* dst) {
          if (ok) {
            const bf16x8_t q = *reinterpret_cast<const bf16x8_t*>(a.x + ((base + (long long)yy * a.W + xx) * a.C) + cch);
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[e] = (float)q[e];
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[e] = 0.f;
          }
        };
        ld(s.ok00, s.yl, s.xl, v00); ld(s.ok01, s.yl, s.xh, v01); ld(s.ok10, s.yh, s.xl, v10); ld(s.ok11, s.yh, s.xh, v11);
        const float hy = 1.f - s.ly, hx = 1.f - s.lx;
        const int wy = s.yl - wy0, wx = s.xl - wx0;
        const bool inwin0 = wy >= 0 && wx >= 0 && wy + 1 < WH && wx + 1 < WW;
        const bool inwin = inwin0 && finite_scale;
        outw = !inwin0;
        int* w00p = win + ((wy * WW + wx) * PS) + cl * 8;
        // the corner tests are per LANE, not per channel: one branch region per corner around its eight atomics (with the tests inside
        // the channel loop hipcc emitted 64 exec-mask save / branch pairs per tap - as many cycles as the arithmetic they guard)
        float dmv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = (float)dcv[e];
          const float dm = d * m;
          dmv[e] = dm;
          g_dy += dm * (hx * (v10[e] - v00[e]) + s.lx * (v11[e] - v01[e]));
          g_dx += dm * (hy * (v01[e] - v00[e]) + s.ly * (v11[e] - v10[e]));
          if constexpr (GM) g_m += d * (s.w00 * v00[e] + s.w01 * v01[e] + s.w10 * v10[e] + s.w11 * v11[e]);
        }
        if (inwin) {
          // (dm * S) once per channel instead of once per update: the same product, the same rounding
#pragma unroll
          for (int e = 0; e < 8; ++e) dmv[e] *= S;
          const float q00 = s.w00, q01 = s.w01, q10 = s.w10, q11 = s.w11;
          if (s.ok00) {
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(w00p + e, __float2int_rn(dmv[e] * q00));
          }
          if (s.ok01) {
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(w00p + PS + e, __float2int_rn(dmv[e] * q01));
          }
          if (s.ok10) {
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(w00p + WW * PS + e, __float2int_rn(dmv[e] * q10));
          }
          if (s.ok11) {
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(w00p + WW * PS + PS + e, __float2int_rn(dmv[e] * q11));
          }
        } else {
          slow = true;
          s_pk = ((unsigned)(s.xl + 1) << 18) | ((unsigned)(s.yl + 1) << 4) | (unsigned)s.ok00 | ((unsigned)s.ok01 << 1) | ((unsigned)s.ok10 << 2) |
                 ((unsigned)s.ok11 << 3);
          s_m = m; s_ly = s.ly; s_lx = s.lx;
        }
        if (a.mask && a.mask_logit) g_m *= m * (1.f - m);
      }
    }
#pragma unroll
    for (int o = L >> 1; o > 0; o >>= 1) {
      g_dy += __shfl_xor(g_dy, o, 64); g_dx += __shfl_xor(g_dx, o, 64); g_m += __shfl_xor(g_m, o, 64);
    }
    if constexpr (STG) {
      if (cl == 0) {
        dofl[pl * (2 * taps) + 2 * tap] = live ? g_dy : 0.f;
        dofl[pl * (2 * taps) + 2 * tap + 1] = live ? g_dx : 0.f;
        if (live && a.dmask) atomicAdd(a.dmask + pix * a.mask_ld + k, g_m);
      }
    } else if (live && cl == 0) {
      atomicAdd(a.doff + pix * a.off_ld + 2 * k, g_dy);
      atomicAdd(a.doff + pix * a.off_ld + 2 * k + 1, g_dx);
      if (a.dmask) atomicAdd(a.dmask + pix * a.mask_ld + k, g_m);
    }
    // ---- samples outside the window: global float atomics, issued by the WAVE instead of by the lane that owns the sample.  A lane
    // adding its own 4 corners x 8 channels issues 32 atomic instructions whose 64 lanes touch 64 different 32-B sectors each (16 pixels x
    // 4 channel groups); here eight parked lanes are served at a time, lane l of the wave adding channel l % 8 of parked lane l / 8: one
    // instruction per corner whose addresses form eight 32-B runs (two 128-B lines when the four lanes of a pixel are all parked) -
    // 1/8 of the instructions and 1/8 .. 1/32 of the memory transactions.  The products are formed exactly as the lane would have.
    if (a.oow) {                                    // one count per wave, not one same-address atomic per lane
      const unsigned long long om = __ballot(outw);
      if (om && lane == 0) atomicAdd(a.oow, (unsigned long long)__popcll(om));
    }
    const unsigned long long smask = __ballot(slow);
    if (smask) {                                    // wave-uniform
      if (slow) slow_lane[wave][__popcll(smask & ((1ull << lane) - 1ull))] = (unsigned char)lane;
      const int nslow = __popcll(smask);
      for (int j0 = 0; j0 < nslow; j0 += 8) {
        const int j = j0 + (lane >> 3);
        const int src = slow_lane[wave][j < nslow ? j : 0];
        const unsigned pk = (unsigned)__shfl((int)s_pk, src, 64);
        const int yl = (int)((pk >> 4) & 0x3fffu) - 1, xl = (int)(pk >> 18) - 1, okb = (int)(pk & 15u);
        const float mm = __shfl(s_m, src, 64), ly = __shfl(s_ly, src, 64), lx = __shfl(s_lx, src, 64);
        const float hy = 1.f - ly, hx = 1.f - lx;                                       // the weights as make_samp() forms them
        const float q00 = hy * hx, q01 = hy * lx, q10 = ly * hx, q11 = ly * lx;
        if (j < nslow) {
          const int ch = (src & 3) * 8 + (lane & 7);
          const float dm = (float)dcl[(wave * 16 + (src >> 2)) * CC + ch] * mm;
          float* gp = a.dx + ((base + (long long)yl * a.W + xl) * a.C) + c0 + ch;
          if (okb & 1) atomicAdd(gp, dm * q00);
          if (okb & 2) atomicAdd(gp + a.C, dm * q01);
          if (okb & 4) atomicAdd(gp + (long long)a.W * a.C, dm * q10);
          if (okb & 8) atomicAdd(gp + (long long)a.W * a.C + a.C, dm * q11);
        }
      }
    }
    if (tap + 1 < taps) w_store();                 // (this wave's own reads of the old rows finished before the barrier above)
    __syncthreads();                               // next tap's weight rows visible; the dcols tile may be overwritten
  }
  for (int i = tid; i < wsize; i += 256) {
    const int c = i % CC, r = i / CC;
    const int q = win[r * PS + c];
    if (q != 0) {
      const float v = (float)q * invS;
      const int wx = r % WW, wy = r / WW;
      atomicAdd(a.dx + ((base + (long long)(wy0 + wy) * a.W + (wx0 + wx)) * a.C) + c0 + c, v);
    }
  }
  if constexpr (STG) {      // 2 * taps consecutive floats per pixel, the eight pixels of a tile row one after the other (off_ld apart)
    for (int i = tid; i < 64 * 2 * taps; i += 256) {
      const int p = i / (2 * taps), j = i - p * (2 * taps);
      const int hop = ho0 + (p >> 3), wop = wo0 + (p & 7);
      const float v = dofl[i];
      if (hop < a.Ho && wop < a.Wo && v != 0.f) atomicAdd(a.doff + (((long long)n * a.Ho + hop) * a.Wo + wop) * a.off_ld + 2 * g * taps + j, v);
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

unsigned long long* g_dcn_oow = nullptr;      // sod_deform_conv_set_window_counter

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
  a.oow = g_dcn_oow;
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
  SOD_LAUNCH(dcn_im2col_kernel<__bf16>, dim3(grid_for((long long)N * a.Ho * a.Wo * KH * KW * (C / 8))), dim3(256), 0, (hipStream_t)stream, a);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

// fp32 validation mode (SOD_PRECISION=fp32): the same gather with fp32 activations and columns; the contraction with the weights and
// both of its gradients then run on the fp32 convolution kernels (f32_path.hip) over the column tensor.
extern "C" int sod_deform_im2col_f32(const float* x, const float* offset, const float* mask, float* cols,
                                     int N, int H, int W, int C, int KH, int KW, int stride, int pad, int dil, int deformable_groups,
                                     int off_ld, int mask_ld, int mask_is_logit, void* stream) {
  if (!x || !offset || !cols) return SOD_EARG;
  DcnArgs a{};
  int rc = dcn_fill(a, N, H, W, C, KH, KW, stride, pad, dil, deformable_groups, off_ld, mask_ld, mask_is_logit);
  if (rc) return rc;
  a.x = (const __bf16*)x; a.off = offset; a.mask = mask; a.cols = (__bf16*)cols;
  SOD_LAUNCH(dcn_im2col_kernel<float>, dim3(grid_for((long long)N * a.Ho * a.Wo * KH * KW * (C / 8))), dim3(256), 0, (hipStream_t)stream, a);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

// dX (fp32 atomics), dOffset, dMask from fp32 column gradients: the plain scatter kernel (summation order of dX varies run to run in
// the last bit; everything else of the validation mode is order-fixed).
extern "C" int sod_deform_col2im_f32(const float* dcols, const float* x, const float* offset, const float* mask,
                                     float* dx_f32, float* doffset, float* dmask,
                                     int N, int H, int W, int C, int KH, int KW, int stride, int pad, int dil, int deformable_groups,
                                     int off_ld, int mask_ld, int mask_is_logit, void* stream) {
  if (!dcols || !x || !offset || !dx_f32 || !doffset || (mask && !dmask)) return SOD_EARG;
  DcnArgs a{};
  int rc = dcn_fill(a, N, H, W, C, KH, KW, stride, pad, dil, deformable_groups, off_ld, mask_ld, mask_is_logit);
  if (rc) return rc;
  a.x = (const __bf16*)x; a.off = offset; a.mask = mask; a.dcols = (const __bf16*)dcols; a.dx = dx_f32; a.doff = doffset; a.dmask = dmask;
  const int per = (C / deformable_groups) / 8;
  int red = 0;
  if (per <= 64 && (per & (per - 1)) == 0 && (C / 8) % per == 0) red = per;
  SOD_LAUNCH(dcn_col2im_kernel<float>, dim3(grid_for((long long)N * a.Ho * a.Wo * KH * KW * (C / 8))), dim3(256), 0, (hipStream_t)stream, a, red);
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
  {   // tiled LDS-window kernel when the window fits (always for the 3x3 / stride-1 layers of this path)
    constexpr int r_env = 3, CC = 32;       // 3 px of window slack, 32 channels per workgroup (64: fewer workgroups per CU, measured slower)
    const int cpg = C / deformable_groups;
    const int WH = 7 * stride + (KH - 1) * dil + 2 + 2 * r_env, WW = 7 * stride + (KW - 1) * dil + 2 + 2 * r_env;
    const size_t lds = (size_t)WH * WW * (CC + 1) * sizeof(float);
    if (cpg % CC == 0 && C % CC == 0 && lds <= 64 * 1024 && N <= 65535 && C / CC <= 65535) {
      const int tiles_x = (a.Wo + 7) / 8, tiles_y = (a.Ho + 7) / 8;
      dim3 grid(tiles_x * tiles_y, C / CC, N);
      SOD_LAUNCH(dcn_col2im_tile_kernel<32>, grid, dim3(256), lds, st, a, tiles_x, WH, WW, r_env);
      SOD_CHECK_LAUNCH();
      return SOD_OK;
    }
  }
  const int per = (C / deformable_groups) / 8;
  int red = 0;
  if (per <= 64 && (per & (per - 1)) == 0 && (C / 8) % per == 0) red = per;
  // the atomic fallback (red == 0) accumulates: the caller passes zero-initialised doffset / dmask in every case (they are
  // pitched buffers whose padding columns must be zero anyway)
  SOD_LAUNCH(dcn_col2im_kernel<__bf16>, dim3(grid_for((long long)N * a.Ho * a.Wo * KH * KW * (C / 8))), dim3(256), 0, st, a, red);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_deform_conv_set_window_counter(unsigned long long* device_counter) {
  g_dcn_oow = device_counter;
  return SOD_OK;
}

// Slack of the fused backward's LDS window in pixels beyond the tile's receptive field: -1 = the default, 2.  Larger windows keep
// larger offsets off the global-atomic path and cost workgroups per CU (29.7 KB at 2, 47.6 KB at 4, 69.8 KB at 6 for a 3x3 layer).
static int g_dcn_fused_slack = -1;
static int dcn_fused_slack() {
  return g_dcn_fused_slack >= 0 ? g_dcn_fused_slack : 2;
}

extern "C" int sod_deform_conv_set_window_slack(int pixels) {
  if (pixels < -1 || pixels > 16) return SOD_EARG;
  g_dcn_fused_slack = pixels;
  return SOD_OK;
}

// LDS bytes of one dcn_bwd_fused_kernel workgroup: the fixed-point dX window (8x8 output tile + receptive field + slack, 33-float pitch),
// 32 weight rows of K bf16 (+16 B pad), the 64 x 32 bf16 column-gradient tile.
static size_t dcn_bwd_fused_lds(int K, int KH, int KW, int stride, int dil, bool staged = false) {
  const int r_env = dcn_fused_slack();
  const int WH = 7 * stride + (KH - 1) * dil + 2 + 2 * r_env, WW = 7 * stride + (KW - 1) * dil + 2 + 2 * r_env;
  return (((size_t)WH * WW * 33 * 4 + 15) & ~(size_t)15) + (size_t)32 * (K * 2 + 16) + (size_t)64 * 32 * 2 + (staged ? (size_t)64 * 2 * KH * KW * sizeof(float) : 0);
}

extern "C" int sod_deform_conv_bwd_fused_supported(int C, int K, int KH, int KW, int stride, int dil, int deformable_groups) {
  if (C <= 0 || K <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || dil <= 0 || deformable_groups <= 0 || C % deformable_groups) return 0;
  const int cpg = C / deformable_groups;
  if ((K != 128 && K != 256 && K != 512) || (C & 31) || (cpg & 31) || C / 32 > 65535) return 0;
  return dcn_bwd_fused_lds(K, KH, KW, stride, dil) <= 96 * 1024 ? 1 : 0;
}

extern "C" int sod_deform_conv_bwd_fused(const void* dy, const void* wt, const void* x, const float* offset, const float* mask, float* dx_f32,
                                         float* doffset, float* dmask, float* wnorm_ws, int N, int H, int W, int C, int K, int KH, int KW, int stride,
                                         int pad, int dil, int deformable_groups, int off_ld, int mask_ld, int mask_is_logit, void* stream) {
  if (!dy || !wt || !x || !offset || !dx_f32 || !doffset || !wnorm_ws || (mask && !dmask)) return SOD_EARG;
  DcnArgs a{};
  int rc = dcn_fill(a, N, H, W, C, KH, KW, stride, pad, dil, deformable_groups, off_ld, mask_ld, mask_is_logit);
  if (rc) return rc;
  const int cpg = C / deformable_groups;
  if ((K != 128 && K != 256 && K != 512) || (C & 31) || (cpg & 31) || N > 65535 || C / 32 > 65535 || H > 16382 || W > 16382) return SOD_EARG;
  const int r_env = dcn_fused_slack();
  const int WH = 7 * stride + (KH - 1) * dil + 2 + 2 * r_env, WW = 7 * stride + (KW - 1) * dil + 2 + 2 * r_env;
  // offset gradients staged in LDS from 4 px of slack on (see dcn_bwd_fused_kernel), if the larger allocation still fits
  const bool staged = r_env >= 4 && KH * KW <= 9 && dcn_bwd_fused_lds(K, KH, KW, stride, dil, true) <= 96 * 1024;
  const size_t lds = dcn_bwd_fused_lds(K, KH, KW, stride, dil, staged);
  if (lds > 96 * 1024 || (unsigned long long)N * a.Ho * a.Wo * K * 2ull >= 0x80000000ull * 4ull) return SOD_EARG;
  a.x = (const __bf16*)x; a.off = offset; a.mask = mask; a.dx = dx_f32; a.doff = doffset; a.dmask = dmask;
  hipStream_t st = (hipStream_t)stream;
  const int rows = KH * KW * C;
  SOD_LAUNCH(dcn_wnorm_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, (const __bf16*)wt, rows, K, wnorm_ws);
  const int tiles_x = (a.Wo + 7) / 8, tiles_y = (a.Ho + 7) / 8;
  const dim3 grid(tiles_x * tiles_y, C / 32, N);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipSuccess;
    const void* kernels[12] = {
        (const void*)dcn_bwd_fused_kernel<4, false, false>, (const void*)dcn_bwd_fused_kernel<8, false, false>, (const void*)dcn_bwd_fused_kernel<16, false, false>,
        (const void*)dcn_bwd_fused_kernel<4, true, false>,  (const void*)dcn_bwd_fused_kernel<8, true, false>,  (const void*)dcn_bwd_fused_kernel<16, true, false>,
        (const void*)dcn_bwd_fused_kernel<4, false, true>,  (const void*)dcn_bwd_fused_kernel<8, false, true>,  (const void*)dcn_bwd_fused_kernel<16, false, true>,
        (const void*)dcn_bwd_fused_kernel<4, true, true>,   (const void*)dcn_bwd_fused_kernel<8, true, true>,   (const void*)dcn_bwd_fused_kernel<16, true, true>};
    for (int i = 0; i < 12 && e == hipSuccess; ++i) e = hipFuncSetAttribute(kernels[i], hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
#define SOD_DCN_BWD_LAUNCH(KS_, GM_, STG_) \
  SOD_LAUNCH((dcn_bwd_fused_kernel<KS_, GM_, STG_>), grid, dim3(256), lds, st, a, (const __bf16*)dy, (const __bf16*)wt, wnorm_ws, tiles_x, WH, WW, r_env)
#define SOD_DCN_BWD_PICK(GM_, STG_) \
  do { if (K == 128) SOD_DCN_BWD_LAUNCH(4, GM_, STG_); else if (K == 256) SOD_DCN_BWD_LAUNCH(8, GM_, STG_); else SOD_DCN_BWD_LAUNCH(16, GM_, STG_); } while (0)
  if (a.dmask) { if (staged) SOD_DCN_BWD_PICK(true, true); else SOD_DCN_BWD_PICK(true, false); }
  else         { if (staged) SOD_DCN_BWD_PICK(false, true); else SOD_DCN_BWD_PICK(false, false); }
#undef SOD_DCN_BWD_PICK
#undef SOD_DCN_BWD_LAUNCH
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_f32_to_bf16(const float* x, void* y, long long n, void* stream) {
  if (!x || !y || n < 0 || (n & 7)) return SOD_EARG;
  SOD_LAUNCH(f32_to_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, x, (__bf16*)y, n / 8);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}
