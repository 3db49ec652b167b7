// Implicit-GEMM convolution on MFMA for gfx950 (CDNA4): forward, data-gradient and weight-gradient.
//
// Replaces the ATen/cuDNN convolutions the reference reaches through detectron2's ResNet/FPN and through
// FCOSHead (reference: slender_det/modeling/meta_arch/fcos/fcosv2.py:277-381, backbone/fpn.py:94-115).
//
// Layout: activations NHWC bf16, weights [Cout][R][S][Cin] bf16 ("KRSC"; for dgrad the host supplies the
// transposed copy [Cin][R][S][Cout]).  GEMM view: rows of the MFMA "A" operand are output channels (q),
// columns of the "B" operand are pixels (p), the contraction index is (tap, channel), 64 per K-step.
// Both operands are contraction-contiguous in memory, so tiles are staged HBM->LDS with
// buffer_load_dwordx4 ... lds (no VGPR round trip); padding/out-of-range lanes use an out-of-bounds
// buffer offset and the hardware returns zeros.  LDS rows are 128 B; the 16-B chunk index is XOR-swizzled
// on the SOURCE side ((row>>1)&7) so that ds_read_b128 fragment reads are bank-conflict free.
//
// wgrad contracts over pixels, which is the slow axis of both operands: tiles are staged [pixel][128 ch]
// and fragments are fetched with ds_read_b64_tr_b16 (hardware transpose), swizzled at 32-B granularity.
#include "common.h"

namespace {

enum { MODE_FWD = 0, MODE_DGRAD = 1 };
enum {
  F_BIAS = 1, F_RELU = 2, F_RES = 4, F_RES_UP2 = 8, F_MASK = 16,
};

struct ConvArgs {
  const void* src;     // fwd: x (N,Hs,Ws,Cred); dgrad: dy (N,Hs,Ws,Cred)
  const void* w;       // [Nout][R*S*Cred]
  void* dst;           // (N,Hp,Wp,Nout) rows at dst_img_stride
  const float* bias;   // [Nout] or null
  const void* res;     // bf16, indexed like dst (or half-resolution with F_RES_UP2)
  const void* mask;    // bf16, indexed like dst: dst = mask>0 ? v : 0 (ReLU backward)
  uint32_t src_bytes, w_bytes;
  int N, Hs, Ws, Cred;
  int Hp, Wp, Nout;
  int R, S, stride, pad, dil;
  int src_img_stride, dst_img_stride, res_img_stride;  // elements
  int Kred, T, P;      // R*S*Cred, #K-steps, N*Hp*Wp
  int flags;
  int nq_tiles, np_tiles;
  FastDiv div_hw, div_w, div_cpt /* Cred/64 (fast) or Cred/8 (generic) */, div_s, div_stride;
};

template <int MODE, bool GENERIC, int WQ, int WP, int FQ, int FP, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvArgs a) {
  constexpr int BQ = WQ * FQ * 16, BP = WP * FP * 16;
  constexpr int W_TILE = BQ * 128, X_TILE = BP * 128, STAGE = W_TILE + X_TILE;
  constexpr int XI = BP / 32;   // X rows per thread
  static_assert(WQ * WP == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = bid % a.nq_tiles, pt = bid / a.nq_tiles;   // q fastest: neighbours share the X tile
  const int q0 = qt * BQ, p0 = pt * BP;

  auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.w_bytes, 0x00020000);
  auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.src), 0, a.src_bytes, 0x00020000);

  // ---- per-thread staging geometry (rows are fixed for the whole K loop) ----
  const int srow = lane >> 3;                                   // row inside one 8-row wave instruction
  const int spos = lane & 7;                                    // 16-B slot inside the 128-B row
  const int sswz = (lane >> 4) | ((wave & 1) << 2);             // ((tile_row>>1)&7), see header
  const int schunk = spos ^ sswz;                               // logical chunk this lane fetches
  int xh[XI], xw[XI];
  uint32_t xoff[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int row = (i * 4 + wave) * 8 + srow;
    const uint32_t p = p0 + row;
    if (p < (uint32_t)a.P) {
      const uint32_t n = fd_div(p, a.div_hw);
      const uint32_t rem = p - n * a.div_hw.d;
      const uint32_t ph = fd_div(rem, a.div_w);
      const uint32_t pw = rem - ph * a.div_w.d;
      if (MODE == MODE_FWD) { xh[i] = (int)ph * a.stride - a.pad; xw[i] = (int)pw * a.stride - a.pad; }
      else                  { xh[i] = (int)ph + a.pad;            xw[i] = (int)pw + a.pad; }
      xoff[i] = n * (uint32_t)a.src_img_stride;
    } else {
      xh[i] = -(1 << 28); xw[i] = -(1 << 28); xoff[i] = 0;
    }
  }

  // branch-free source coordinate: returns all-ones mask when valid
  auto src_coord = [&](int base, int tapo, int lim, int& out) -> uint32_t {
    if (MODE == MODE_FWD) {
      out = base + tapo * a.dil;
      return (uint32_t) - (int)((unsigned)out < (unsigned)lim);
    } else {
      const int t = base - tapo * a.dil;
      if (a.stride == 1) {   // wave-uniform
        out = t;
        return (uint32_t) - (int)((unsigned)t < (unsigned)lim);
      }
      const uint32_t tt = (uint32_t)(t < 0 ? 0 : t);
      const uint32_t o = fd_div(tt, a.div_stride);
      out = (int)o;
      return (uint32_t) - (int)((t >= 0) & (o * (uint32_t)a.stride == tt) & (o < (uint32_t)lim));
    }
  };

  auto stage = [&](int t, char* buf) {
    // weights: row-major [q][Kred]; K-step t covers flattened contraction [t*64, t*64+64)
#pragma unroll
    for (int rb = 0; rb < BQ; rb += 32) {
      if (rb + wave * 8 < BQ) {
        const int row = rb + wave * 8 + srow;
        const int q = q0 + row;
        const int kk = t * 64 + schunk * 8;
        const uint32_t m = (uint32_t) - (int)((q < a.Nout) & (kk < a.Kred));
        const uint32_t off = ((uint32_t)q * (uint32_t)a.Kred + (uint32_t)kk) * 2u;
        const uint32_t voff = (off & m) | (SOD_OOB & ~m);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, SOD_LDS(buf + (rb + wave * 8) * 128), 16, voff, 0, 0, 0);
      }
    }
    int r_u = 0, s_u = 0, c0_u = 0;
    if (!GENERIC) {
      const uint32_t tap = fd_div((uint32_t)t, a.div_cpt);
      c0_u = (t - (int)(tap * a.div_cpt.d)) << 6;
      r_u = (int)fd_div(tap, a.div_s);
      s_u = (int)tap - r_u * a.S;
    }
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      int r, s, c;
      uint32_t m = 0xFFFFFFFFu;
      if (GENERIC) {
        const uint32_t g = (uint32_t)t * 8u + (uint32_t)schunk;     // 8-channel chunk index
        const uint32_t tap = fd_div(g, a.div_cpt);
        c = (int)(g - tap * a.div_cpt.d) << 3;
        r = (int)fd_div(tap, a.div_s);
        s = (int)tap - r * a.S;
        m = (uint32_t) - (int)(r < a.R);
      } else {
        r = r_u; s = s_u; c = c0_u + schunk * 8;
      }
      int h, w;
      m &= src_coord(xh[i], r, a.Hs, h);
      m &= src_coord(xw[i], s, a.Ws, w);
      const uint32_t off = (xoff[i] + ((uint32_t)h * (uint32_t)a.Ws + (uint32_t)w) * (uint32_t)a.Cred + (uint32_t)c) * 2u;
      const uint32_t voff = (off & m) | (SOD_OOB & ~m);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, SOD_LDS(buf + W_TILE + (i * 4 + wave) * 1024), 16, voff, 0, 0, 0);
    }
  };

  // ---- fragment read offsets ----
  const int wq = wave / WP, wp = wave % WP;
  const int fr = lane & 15, fg = lane >> 4;
  uint32_t aoff[FQ], boff[FP];
#pragma unroll
  for (int i = 0; i < FQ; ++i) {
    const int row = (wq * FQ + i) * 16 + fr;
    aoff[i] = row * 128 + ((fg ^ ((row >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < FP; ++j) {
    const int row = (wp * FP + j) * 16 + fr;
    boff[j] = W_TILE + row * 128 + ((fg ^ ((row >> 1) & 7)) << 4);
  }

  f32x4_t acc[FQ][FP];
#pragma unroll
  for (int i = 0; i < FQ; ++i)
#pragma unroll
    for (int j = 0; j < FP; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  stage(0, smem);
  for (int t = 0; t < a.T; ++t) {
    char* cur = smem + (t & 1) * STAGE;
    __syncthreads();   // drains this wave's LDS-DMA (vmcnt(0)) and orders all waves' reads of the other buffer
    if (t + 1 < a.T) stage(t + 1, smem + ((t + 1) & 1) * STAGE);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t af[FQ], bf[FP];
#pragma unroll
      for (int i = 0; i < FQ; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(cur + (aoff[i] ^ (ks << 6)));
#pragma unroll
      for (int j = 0; j < FP; ++j) bf[j] = *reinterpret_cast<const bf16x8_t*>(cur + (boff[j] ^ (ks << 6)));
#pragma unroll
      for (int i = 0; i < FQ; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: D[row=q][col=p]; lane holds 4 consecutive q for one pixel ----
  const bool vec_ok = (a.Nout & 3) == 0;
#pragma unroll
  for (int j = 0; j < FP; ++j) {
    const uint32_t p = p0 + (wp * FP + j) * 16 + fr;
    if (p >= (uint32_t)a.P) continue;
    const uint32_t n = fd_div(p, a.div_hw);
    const uint32_t rem = p - n * a.div_hw.d;
    size_t res_row = 0;
    if (a.flags & F_RES_UP2) {
      const uint32_t ph = fd_div(rem, a.div_w);
      const uint32_t pw = rem - ph * a.div_w.d;
      res_row = (size_t)n * a.res_img_stride + (size_t)((ph >> 1) * (a.Wp >> 1) + (pw >> 1)) * a.Nout;
    } else if (a.flags & F_RES) {
      res_row = (size_t)n * a.res_img_stride + (size_t)rem * a.Nout;
    }
    const size_t dst_row = (size_t)n * a.dst_img_stride + (size_t)rem * a.Nout;
#pragma unroll
    for (int i = 0; i < FQ; ++i) {
      const int q = q0 + (wq * FQ + i) * 16 + fg * 4;
      if (q >= a.Nout) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (vec_ok) {
        if (a.flags & F_BIAS) {
          const f32x4_t b = *reinterpret_cast<const f32x4_t*>(a.bias + q);
          v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3];
        }
        if (a.flags & (F_RES | F_RES_UP2)) {
          const bf16x4_t rv = *reinterpret_cast<const bf16x4_t*>((const __bf16*)a.res + res_row + q);
          v[0] += (float)rv[0]; v[1] += (float)rv[1]; v[2] += (float)rv[2]; v[3] += (float)rv[3];
        }
        if (a.flags & F_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (a.flags & F_MASK) {
          const bf16x4_t mv = *reinterpret_cast<const bf16x4_t*>((const __bf16*)a.mask + dst_row + q);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ((float)mv[e] > 0.f) ? v[e] : 0.f;
        }
        if (OUT_F32) {
          *reinterpret_cast<f32x4_t*>((float*)a.dst + dst_row + q) = f32x4_t{v[0], v[1], v[2], v[3]};
        } else {
          bf16x4_t o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
          *reinterpret_cast<bf16x4_t*>((__bf16*)a.dst + dst_row + q) = o;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (q + e >= a.Nout) break;
          float x = v[e];
          if (a.flags & F_BIAS) x += a.bias[q + e];
          if (a.flags & (F_RES | F_RES_UP2)) x += (float)((const __bf16*)a.res)[res_row + q + e];
          if (a.flags & F_RELU) x = fmaxf(x, 0.f);
          if (a.flags & F_MASK) x = ((float)((const __bf16*)a.mask)[dst_row + q + e] > 0.f) ? x : 0.f;
          if (OUT_F32) ((float*)a.dst)[dst_row + q + e] = x;
          else ((__bf16*)a.dst)[dst_row + q + e] = (__bf16)x;
        }
      }
    }
  }
}

// --------------------------------------------------------------------------------------------
// wgrad: dW[q][tap][c] += sum_p dY[p][q] * X[p shifted by tap][c]
// --------------------------------------------------------------------------------------------
struct WgradArgs {
  const void* dy;      // (N,Ho,Wo,K) bf16 rows at dy_img_stride
  const void* x;       // (N,Hx,Wx,C) bf16
  float* dw;           // [K][R][S][C] fp32, accumulated atomically
  const float* qscale; // optional per-output-channel factor (folded FrozenBN scale)
  uint32_t dy_bytes, x_bytes;
  int N, Hx, Wx, C, Ho, Wo, K;
  int R, S, stride, pad, dil;
  int dy_img_stride, x_img_stride;
  int P, nz, p_per_split;   // p_per_split multiple of 64
  int QT, CT;
  FastDiv div_hw, div_w, div_s;
};

__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs a) {
  constexpr int TILE = 64 * 256, STAGE = 2 * TILE;   // [64 pixels][128 ch] bf16, two operands
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
  const int RS = a.R * a.S;
  const int tap = bid % RS; bid /= RS;
  const int ct = bid % a.CT; bid /= a.CT;
  const int qt = bid % a.QT; bid /= a.QT;
  const int z = bid;
  const int r = tap / a.S, s = tap - r * a.S;
  const int q0 = qt * 128, c0 = ct * 128;
  const int pbeg = z * a.p_per_split;
  int pend = pbeg + a.p_per_split; if (pend > a.P) pend = a.P;
  const int nsteps = (pend - pbeg + 63) >> 6;

  auto yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, a.dy_bytes, 0x00020000);
  auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);

  // staging: one wave instruction = 4 pixel rows x 256 B; lane -> (row_in, 16-B slot)
  const int srow = lane >> 4, spos = lane & 15;
  const int sswz = srow | (((wave >> 1) & 1) << 2);       // (row&3) | ((row>>3)&1)<<2, see header
  const int schunk = spos ^ (sswz << 1);                  // logical 16-B chunk (8 channels)
  const bool qok = (q0 + schunk * 8) < a.K;
  const bool cok = (c0 + schunk * 8) < a.C;

  auto stage = [&](int it, char* buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (i * 4 + wave) * 4 + srow;
      const int p = pbeg + it * 64 + row;
      uint32_t vy = SOD_OOB, vx = SOD_OOB;
      if (p < pend) {
        const uint32_t n = fd_div((uint32_t)p, a.div_hw);
        const uint32_t rem = (uint32_t)p - n * a.div_hw.d;
        const uint32_t ho = fd_div(rem, a.div_w);
        const uint32_t wo = rem - ho * a.div_w.d;
        if (qok) vy = (n * (uint32_t)a.dy_img_stride + rem * (uint32_t)a.K + (uint32_t)(q0 + schunk * 8)) * 2u;
        const int hi = (int)ho * a.stride - a.pad + r * a.dil;
        const int wi = (int)wo * a.stride - a.pad + s * a.dil;
        if (cok && (unsigned)hi < (unsigned)a.Hx && (unsigned)wi < (unsigned)a.Wx)
          vx = (n * (uint32_t)a.x_img_stride + (uint32_t)(hi * a.Wx + wi) * (uint32_t)a.C + (uint32_t)(c0 + schunk * 8)) * 2u;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(yrsrc, SOD_LDS(buf + (i * 4 + wave) * 1024), 16, vy, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, SOD_LDS(buf + TILE + (i * 4 + wave) * 1024), 16, vx, 0, 0, 0);
    }
  };

  // transposed fragment reads: lane 4q+p of a 16-lane group addresses pixel row q, channels 4p..4p+3
  const int wq = wave >> 1, wc = wave & 1;
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = lane >> 4;
  const int tswz = tq | ((tg & 1) << 2);
  uint32_t aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    aoff[i] = (uint32_t)(8 * tg + tq) * 256u + (uint32_t)(((wq * 4 + i) ^ tswz) * 32) + tp * 8;
    boff[i] = TILE + (uint32_t)(8 * tg + tq) * 256u + (uint32_t)(((wc * 4 + i) ^ tswz) * 32) + tp * 8;
  }

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  if (nsteps > 0) stage(0, smem);
  for (int it = 0; it < nsteps; ++it) {
    char* cur = smem + (it & 1) * STAGE;
    __syncthreads();
    if (it + 1 < nsteps) stage(it + 1, smem + ((it + 1) & 1) * STAGE);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)SOD_LDS(cur + aoff[i] + ks * 8192));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)SOD_LDS(cur + aoff[i] + ks * 8192 + 1024));
        s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        af[i] = __builtin_bit_cast(bf16x8_t, v);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)SOD_LDS(cur + boff[j] + ks * 8192));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)SOD_LDS(cur + boff[j] + ks * 8192 + 1024));
        s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        bf[j] = __builtin_bit_cast(bf16x8_t, v);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }

  // D[row=q][col=c]
  const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = q0 + (wq * 4 + i) * 16 + fg * 4 + e;
      if (q >= a.K) continue;
      const float qs = a.qscale ? a.qscale[q] : 1.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = c0 + (wc * 4 + j) * 16 + fr;
        if (c < a.C) atomicAdd(a.dw + ((size_t)q * RS + tap) * a.C + c, acc[i][j][e] * qs);
      }
    }
  }
}

template <int MODE, bool GENERIC, int WQ, int WP, int FQ, int FP, bool OUT_F32>
int launch_conv(const ConvArgs& a0, hipStream_t st) {
  constexpr int BQ = WQ * FQ * 16, BP = WP * FP * 16;
  ConvArgs a = a0;
  a.nq_tiles = (a.Nout + BQ - 1) / BQ;
  a.np_tiles = (a.P + BP - 1) / BP;
  const size_t lds = 2 * (size_t)(BQ + BP) * 128;
  auto kern = conv_igemm_kernel<MODE, GENERIC, WQ, WP, FQ, FP, OUT_F32>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  SOD_LAUNCH(kern, dim3(a.nq_tiles * a.np_tiles), dim3(256), lds, st, a);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

template <int MODE, bool OUT_F32>
int dispatch_conv(const ConvArgs& a, hipStream_t st) {
  const bool generic = (a.Cred & 63) != 0;
  if (a.Nout <= 16) {
    return generic ? launch_conv<MODE, true, 1, 4, 1, 4, OUT_F32>(a, st) : launch_conv<MODE, false, 1, 4, 1, 4, OUT_F32>(a, st);
  } else if (a.Nout <= 64) {
    return generic ? launch_conv<MODE, true, 1, 4, 4, 4, OUT_F32>(a, st) : launch_conv<MODE, false, 1, 4, 4, 4, OUT_F32>(a, st);
  }
  return generic ? launch_conv<MODE, true, 2, 2, 4, 4, OUT_F32>(a, st) : launch_conv<MODE, false, 2, 2, 4, 4, OUT_F32>(a, st);
}

int fill_common(ConvArgs& a, int N, int Hs, int Ws, int Cred, int Hp, int Wp, int Nout, int R, int S, int stride,
                int pad, int dil, long long src_img_stride, long long dst_img_stride, size_t dst_elt) {
  if (N <= 0 || Hs <= 0 || Ws <= 0 || Hp <= 0 || Wp <= 0 || Nout <= 0 || R <= 0 || S <= 0 || stride <= 0 || dil <= 0 || pad < 0)
    return SOD_EARG;
  if (Cred <= 0 || (Cred & 7)) return SOD_EARG;
  if (src_img_stride < (long long)Hs * Ws * Cred || dst_img_stride < (long long)Hp * Wp * Nout) return SOD_EARG;
  const unsigned long long sb = (unsigned long long)N * src_img_stride * 2ull;
  const unsigned long long wb = (unsigned long long)Nout * R * S * Cred * 2ull;
  const unsigned long long db = (unsigned long long)N * dst_img_stride * dst_elt;
  if (sb >= 0x80000000ull || wb >= 0x80000000ull || db >= 0x200000000ull) return SOD_ESIZE;
  if ((long long)N * Hp * Wp >= (1ll << 31)) return SOD_ESIZE;
  a.src_bytes = (uint32_t)sb; a.w_bytes = (uint32_t)wb;
  a.N = N; a.Hs = Hs; a.Ws = Ws; a.Cred = Cred; a.Hp = Hp; a.Wp = Wp; a.Nout = Nout;
  a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.dil = dil;
  a.src_img_stride = (int)src_img_stride; a.dst_img_stride = (int)dst_img_stride;
  a.Kred = R * S * Cred; a.T = (a.Kred + 63) / 64; a.P = N * Hp * Wp;
  a.div_hw = make_fastdiv((uint32_t)(Hp * Wp));
  a.div_w = make_fastdiv((uint32_t)Wp);
  a.div_cpt = make_fastdiv((uint32_t)((Cred & 63) ? Cred / 8 : Cred / 64));
  a.div_s = make_fastdiv((uint32_t)S);
  a.div_stride = make_fastdiv((uint32_t)stride);
  return SOD_OK;
}

}  // namespace

#include "../../include/slender_hip.h"

extern "C" int sod_conv2d_fwd(const void* x, const void* w, const float* bias, const void* res, void* y,
                              int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                              long long x_img_stride, long long y_img_stride, long long res_img_stride,
                              int flags, int out_f32, void* stream) {
  if (!x || !w || !y) return SOD_EARG;
  const int Ho = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1;
  const int Wo = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return SOD_EARG;
  if (x_img_stride <= 0) x_img_stride = (long long)H * W * C;
  if (y_img_stride <= 0) y_img_stride = (long long)Ho * Wo * K;
  ConvArgs a{};
  int rc = fill_common(a, N, H, W, C, Ho, Wo, K, R, S, stride, pad, dil, x_img_stride, y_img_stride, out_f32 ? 4 : 2);
  if (rc) return rc;
  a.src = x; a.w = w; a.dst = y; a.bias = bias; a.res = res; a.mask = nullptr;
  a.flags = 0;
  if (bias) a.flags |= F_BIAS;
  if (flags & SOD_CONV_RELU) a.flags |= F_RELU;
  if (res) {
    if (flags & SOD_CONV_RES_UP2) {
      if ((Ho & 1) || (Wo & 1)) return SOD_EARG;
      a.flags |= F_RES_UP2;
      a.res_img_stride = (int)(res_img_stride > 0 ? res_img_stride : (long long)(Ho / 2) * (Wo / 2) * K);
    } else {
      a.flags |= F_RES;
      a.res_img_stride = (int)(res_img_stride > 0 ? res_img_stride : y_img_stride);
    }
  }
  hipStream_t st = (hipStream_t)stream;
  return out_f32 ? dispatch_conv<MODE_FWD, true>(a, st) : dispatch_conv<MODE_FWD, false>(a, st);
}

extern "C" int sod_conv2d_dgrad(const void* dy, const void* wt, const void* accum, const void* relu_mask, void* dx,
                                int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                long long dy_img_stride, long long dx_img_stride, void* stream) {
  if (!dy || !wt || !dx) return SOD_EARG;
  const int Ho = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1;
  const int Wo = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return SOD_EARG;
  if (dy_img_stride <= 0) dy_img_stride = (long long)Ho * Wo * K;
  if (dx_img_stride <= 0) dx_img_stride = (long long)H * W * C;
  ConvArgs a{};
  // GEMM rows are the INPUT pixels (H,W); the gather source is dY (Ho,Wo,K); output channels = C.
  int rc = fill_common(a, N, Ho, Wo, K, H, W, C, R, S, stride, pad, dil, dy_img_stride, dx_img_stride, 2);
  if (rc) return rc;
  a.src = dy; a.w = wt; a.dst = dx; a.bias = nullptr; a.res = accum; a.mask = relu_mask;
  a.flags = 0;
  if (accum) { a.flags |= F_RES; a.res_img_stride = (int)dx_img_stride; }
  if (relu_mask) a.flags |= F_MASK;
  return dispatch_conv<MODE_DGRAD, false>(a, (hipStream_t)stream);
}

extern "C" int sod_conv2d_wgrad(const void* dy, const void* x, float* dw, const float* qscale,
                                int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                long long dy_img_stride, long long x_img_stride, int splits, void* stream) {
  if (!dy || !x || !dw) return SOD_EARG;
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || (C & 7) || (K & 7)) return SOD_EARG;
  const int Ho = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1;
  const int Wo = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return SOD_EARG;
  if (dy_img_stride <= 0) dy_img_stride = (long long)Ho * Wo * K;
  if (x_img_stride <= 0) x_img_stride = (long long)H * W * C;
  const unsigned long long yb = (unsigned long long)N * dy_img_stride * 2ull, xb = (unsigned long long)N * x_img_stride * 2ull;
  if (yb >= 0x80000000ull || xb >= 0x80000000ull) return SOD_ESIZE;
  WgradArgs a{};
  a.dy = dy; a.x = x; a.dw = dw; a.qscale = qscale; a.dy_bytes = (uint32_t)yb; a.x_bytes = (uint32_t)xb;
  a.N = N; a.Hx = H; a.Wx = W; a.C = C; a.Ho = Ho; a.Wo = Wo; a.K = K;
  a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.dil = dil;
  a.dy_img_stride = (int)dy_img_stride; a.x_img_stride = (int)x_img_stride;
  a.P = N * Ho * Wo;
  a.QT = (K + 127) / 128; a.CT = (C + 127) / 128;
  const int tiles = a.QT * a.CT * R * S;
  if (splits <= 0) {
    // enough blocks for ~3 waves of the 256 CUs x 2 resident blocks, at least 256 pixels per block
    splits = (1536 + tiles - 1) / tiles;
    const int maxs = (a.P + 255) / 256;
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
  }
  int pps = (a.P + splits - 1) / splits;
  pps = (pps + 63) / 64 * 64;
  a.p_per_split = pps;
  a.nz = (a.P + pps - 1) / pps;
  a.div_hw = make_fastdiv((uint32_t)(Ho * Wo));
  a.div_w = make_fastdiv((uint32_t)Wo);
  a.div_s = make_fastdiv((uint32_t)S);
  const size_t lds = 2 * 2 * 64 * 256;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  SOD_LAUNCH(conv_wgrad_kernel, dim3(a.nz * tiles), dim3(256), lds, (hipStream_t)stream, a);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}
