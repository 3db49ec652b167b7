// Deformable convolution v1 / v2 as ONE implicit-GEMM kernel per pass (gfx950): the bilinear gather feeds the MFMA loop through
// LDS, the (N*Ho*Wo, KH*KW*C) column buffer of deform_conv.hip never exists in HBM.
//
// Replaces detectron2.layers.DeformConv / ModulatedDeformConv (CUDA-only, absent from the reference tree; sampling rule in SURVEY.md
// Appendix C.11) at slender_det/layers/df_conv.py:67-78 and slender_det/modeling/meta_arch/reppoints/rpd.py:637-642.
//
// dcn_fwd_fused_kernel: y[p][q] = act(bias[q] + sum_{tap, c} W[q][tap][c] * m(p,tap) * bilinear(x, p, tap)[c])
//   * output tile 128 pixels x 256 output channels per workgroup of 8 waves (wave (wr, wc) = 128 q x 32 px = 8x2 MFMA 16x16x32
//     accumulators), one workgroup per CU, double-buffered LDS (2 x 48 KB);
//   * a K-step = 64 channels of one tap.  Weights [256 q][64 c] arrive by LDS-DMA (rows of 128 B, 16-B chunks XOR-swizzled on the
//     source side).  The sampled tile [128 px][64 c] is built by the VALU: 8 consecutive threads own the 8 16-B chunks of one
//     pixel's 128-B run, so each of the 4 bilinear corners is ONE full cache line per pixel; corners outside the image use the
//     out-of-range buffer offset (hardware zero fill), i.e. the gather is branch-free.  Interpolation is fp32 in the order of
//     dcn_im2col_kernel (v = w00*q00 + w01*q01 + w10*q10 + w11*q11, then * mask, then ONE rounding to bf16), so the tile equals the
//     column buffer of the unfused path bit for bit;
//   * per iteration: issue the corner loads and the weight DMA of step k+1, run the 32 MFMAs of step k, then interpolate and
//     ds_write_b128 step k+1 (the loads had the MFMA phase to land), one barrier.  The two waves of a SIMD drift apart by a phase,
//     so one interpolates while the other feeds the matrix core;
//   * sampling positions (4 corner offsets + 4 weights + mask per pixel) are recomputed only when the tap (or the deformable group)
//     changes: once per C/64 K-steps;
//   * epilogue: bias + ReLU, transposed through LDS so that every lane stores 16 contiguous bytes.
//
// dcn_wgrad_fused_kernel: dW[q][tap][c] += sum_p dY[p][q] * m * bilinear(x, p, tap)[c]   (contraction over pixels)
//   * 256 q x 128 c output tile per workgroup (wave = 128 q x 32 c... see the kernel), K-tile = 64 pixels: dY rows by LDS-DMA, the
//     sampled rows [64 px][128 c] by the same VALU gather, both read back with ds_read_b64_tr_b16 (hardware transpose);
//   * split over pixels with fp32 slabs in the caller's workspace + the fixed-order reduce (as conv_wgrad256.hip): deterministic.
#include "common.h"
#include "../../include/slender_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

struct DcnFArgs {
  const __bf16* x;        // (N,H,W,C)
  const float* off;       // (N,Ho,Wo,off_ld)
  const float* mask;      // (N,Ho,Wo,mask_ld) or null
  const __bf16* w;        // [K][KH*KW][C] bf16 (the (K,1,1,KH*KW*C) GEMM view of the KRSC weights)
  const float* bias;      // [K] or null
  __bf16* y;              // (N,Ho,Wo,K)
  const __bf16* dy;       // wgrad: (N,Ho,Wo,K)
  float* partial;         // wgrad: slabs
  float* dw;              // wgrad: [K][KH*KW][C] fp32, accumulated
  const float* qscale;    // wgrad: optional per-output-channel factor (chain rule through a folded FrozenBN scale)
  uint32_t x_bytes, w_bytes, dy_bytes;
  int N, H, W, C, Ho, Wo, K, KH, KW, stride, pad, dil, DG;
  int off_ld, mask_ld, mask_logit, relu;
  int P;                  // N*Ho*Wo
  int QT, CT, nz, kt_per_split;   // wgrad
  FastDiv div_hw, div_w, div_kw;
};

struct Tap {               // sampling state of one pixel for the current (tap, group)
  uint32_t o00, o01, o10, o11;     // byte offsets of the 4 corner pixels in x (SOD_OOB = outside -> zero fill)
  float w00, w01, w10, w11, m;
};

// dy / dx / mask of pixel `pix` for sampling point k (loaded one tap ahead of their use, see issue_gather)
struct RawOff { float dy, dx, m; };
__device__ __forceinline__ RawOff load_off(const DcnFArgs& a, bool pvalid, long long pix, int k) {
  RawOff r{0.f, 0.f, 1.f};
  if (pvalid) {
    r.dy = a.off[pix * a.off_ld + 2 * k]; r.dx = a.off[pix * a.off_ld + 2 * k + 1];
    if (a.mask) r.m = a.mask[pix * a.mask_ld + k];
  }
  return r;
}

__device__ __forceinline__ Tap make_tap(const DcnFArgs& a, bool pvalid, uint32_t n, int ho, int wo, const RawOff& ro, int tap) {
  Tap t;
  t.o00 = t.o01 = t.o10 = t.o11 = SOD_OOB;
  t.w00 = t.w01 = t.w10 = t.w11 = 0.f; t.m = 1.f;
  if (!pvalid) return t;
  const int ki = (int)fd_div((uint32_t)tap, a.div_kw), kj = tap - ki * a.KW;
  const float py = (float)(ho * a.stride - a.pad + ki * a.dil) + ro.dy, px = (float)(wo * a.stride - a.pad + kj * a.dil) + ro.dx;
  const bool valid = (py > -1.f) && (px > -1.f) && (py < (float)a.H) && (px < (float)a.W);      // make_samp (deform_conv.hip)
  const float fy = floorf(py), fx = floorf(px);
  const int yl = (int)fy, xl = (int)fx, yh = yl + 1, xh = xl + 1;
  const float ly = py - fy, lx = px - fx, hy = 1.f - ly, hx = 1.f - lx;
  t.w00 = hy * hx; t.w01 = hy * lx; t.w10 = ly * hx; t.w11 = ly * lx;
  const uint32_t base = n * (uint32_t)(a.H * a.W);
  const uint32_t rowb = (uint32_t)a.C * 2u;
  if (valid && yl >= 0 && xl >= 0) t.o00 = (base + (uint32_t)(yl * a.W + xl)) * rowb;
  if (valid && yl >= 0 && xh <= a.W - 1) t.o01 = (base + (uint32_t)(yl * a.W + xh)) * rowb;
  if (valid && yh <= a.H - 1 && xl >= 0) t.o10 = (base + (uint32_t)(yh * a.W + xl)) * rowb;
  if (valid && yh <= a.H - 1 && xh <= a.W - 1) t.o11 = (base + (uint32_t)(yh * a.W + xh)) * rowb;
  float m = ro.m;
  if (a.mask && a.mask_logit) m = 1.f / (1.f + expf(-m));
  t.m = m;
  return t;
}

// the four corner chunks (8 channels each) of one sample -> the bf16 chunk of the column tile, arithmetic of dcn_im2col_kernel
__device__ __forceinline__ bf16x8_t interp8(const Tap& t, u32x4_t r00, u32x4_t r01, u32x4_t r10, u32x4_t r11) {
  const bf16x8_t q00 = __builtin_bit_cast(bf16x8_t, r00), q01 = __builtin_bit_cast(bf16x8_t, r01);
  const bf16x8_t q10 = __builtin_bit_cast(bf16x8_t, r10), q11 = __builtin_bit_cast(bf16x8_t, r11);
  bf16x8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = 0.f;
    v += t.w00 * (float)q00[e];
    v += t.w01 * (float)q01[e];
    v += t.w10 * (float)q10[e];
    v += t.w11 * (float)q11[e];
    o[e] = (__bf16)(v * t.m);
  }
  return o;
}

constexpr int F_ROWB = 128;                    // bytes per LDS row: 64 bf16 contraction elements
constexpr int F_A = 256 * F_ROWB;              // weights [256 q][64 c]           32 KB
constexpr int F_B = 128 * F_ROWB;              // samples [128 px][64 c]          16 KB
constexpr int F_STAGE = F_A + F_B;
constexpr int F_LDS = 2 * F_STAGE;             // 96 KB
constexpr int F_EROW = 256 + 16;               // epilogue: [32 px][128 q] bf16 rows, padded

__global__ __launch_bounds__(512, 2) void dcn_fwd_fused_kernel(const DcnFArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
  const int nq = (a.K + 255) / 256;
  const int qt = (int)(bid % (uint32_t)nq), pt = (int)(bid / (uint32_t)nq);
  const int q0 = qt * 256, p0 = pt * 128;
  const int taps = a.KH * a.KW, ccn = a.C >> 6, cpg = a.C / a.DG;
  const int T = taps * ccn;
  const int Kred = taps * a.C;

  auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x), 0, a.x_bytes, 0x00020000);
  auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.w), 0, a.w_bytes, 0x00020000);

  // ---- gather ownership: chunk (8 channels) gch of pixels gpx0 and gpx0 + 64 of the tile
  const int gch = tid & 7, gpx0 = tid >> 3;
  bool pv[2]; uint32_t pn[2]; int pho[2], pwo[2]; long long ppix[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const uint32_t gp = (uint32_t)(p0 + gpx0 + 64 * s);
    pv[s] = gp < (uint32_t)a.P;
    const uint32_t gc = pv[s] ? gp : 0u;
    pn[s] = fd_div(gc, a.div_hw);
    const uint32_t rem = gc - pn[s] * a.div_hw.d;
    pho[s] = (int)fd_div(rem, a.div_w);
    pwo[s] = (int)(rem - (uint32_t)pho[s] * a.div_w.d);
    ppix[s] = (long long)gc;
  }
  Tap tp[2];
  RawOff nxt[2];                             // offsets of the NEXT (tap, group), requested a whole tap (C/64 K-steps) ahead
  int cur_tap = -1, cur_g = -1;
  const int steps_per_change = cpg >> 6;     // K-steps that share one sampling position
#pragma unroll
  for (int s = 0; s < 2; ++s) nxt[s] = load_off(a, pv[s], ppix[s], 0);
  u32x4_t rg[2][4];
  auto issue_gather = [&](int k) {           // corner loads of K-step k (tap = k / ccn, channel chunk = k % ccn)
    const int tap = k / ccn, cc = k - tap * ccn;
    const int g = (cc * 64) / cpg;
    if (tap != cur_tap || g != cur_g) {
      cur_tap = tap; cur_g = g;
#pragma unroll
      for (int s = 0; s < 2; ++s) tp[s] = make_tap(a, pv[s], pn[s], pho[s], pwo[s], nxt[s], tap);
      const int kn = k + steps_per_change;   // first K-step of the next sampling position
      if (kn < T) {
        const int tapn = kn / ccn, gn = ((kn - tapn * ccn) * 64) / cpg;
#pragma unroll
        for (int s = 0; s < 2; ++s) nxt[s] = load_off(a, pv[s], ppix[s], gn * taps + tapn);
      }
    }
    const uint32_t cadd = (uint32_t)(cc * 64 + gch * 8) * 2u;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      rg[s][0] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, tp[s].o00 + cadd, 0, 0);
      rg[s][1] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, tp[s].o01 + cadd, 0, 0);
      rg[s][2] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, tp[s].o10 + cadd, 0, 0);
      rg[s][3] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, tp[s].o11 + cadd, 0, 0);
    }
  };
  auto write_gather = [&](int stage) {       // interpolate the loaded corners and store the two chunks of the sample tile
    char* bt = smem + stage * F_STAGE + F_A;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int px = gpx0 + 64 * s;
      const bf16x8_t o = interp8(tp[s], rg[s][0], rg[s][1], rg[s][2], rg[s][3]);
      *reinterpret_cast<bf16x8_t*>(bt + px * F_ROWB + ((gch ^ ((px >> 1) & 7)) << 4)) = o;
    }
  };
  // ---- weight staging: 32 wave instructions of 8 rows x 128 B per K-step, 4 per wave
  const int srow = lane >> 3, sslot = lane & 7;
  const int sswz = (srow >> 1) | ((wave & 1) << 2);          // (row >> 1) & 7 with row = (j*8 + wave)*8 + srow
  const int schunk = sslot ^ sswz;
  auto issue_weights = [&](int k, int stage) {
    const int tap = k / ccn, cc = k - tap * ccn;
    const uint32_t koff = (uint32_t)(tap * a.C + cc * 64 + schunk * 8) * 2u;
    char* dst = smem + stage * F_STAGE + wave * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = q0 + (j * 8 + wave) * 8 + srow;
      const uint32_t voff = q < a.K ? (uint32_t)q * (uint32_t)Kred * 2u + koff : SOD_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, SOD_LDS(dst + j * 8192), 16, voff, 0, 0, 0);
    }
  };

  // ---- fragment offsets
  const int fr = lane & 15, fg = lane >> 4;
  uint32_t aoff[8], boff[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = wr * 128 + i * 16 + fr;
    aoff[i] = (uint32_t)(row * F_ROWB + ((fg ^ ((row >> 1) & 7)) << 4));
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = wc * 32 + j * 16 + fr;
    boff[j] = (uint32_t)(F_A + row * F_ROWB + ((fg ^ ((row >> 1) & 7)) << 4));
  }
  f32x4_t acc[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: K-step 0 complete in stage 0
  issue_gather(0);
  issue_weights(0, 0);
  write_gather(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int k = 0; k < T; ++k) {
    const char* cur = smem + (k & 1) * F_STAGE;
    const bool more = k + 1 < T;
    if (more) {
      issue_gather(k + 1);
      issue_weights(k + 1, (k + 1) & 1);
    }
    bf16x8_t bf[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bf[j][ks] = *reinterpret_cast<const bf16x8_t*>(cur + (boff[j] ^ (ks << 6)));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(cur + aoff[i]);
      const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(cur + (aoff[i] ^ 64u));
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bf[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bf[j][1], acc[i][j], 0, 0, 0);
      }
    }
    if (more) write_gather((k + 1) & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the weight DMA of step k+1 has landed before anybody reads it
    __syncthreads();
  }

  // ---- epilogue: bias + ReLU, [px][q] bf16 through LDS, 16-B stores
  char* wl = smem + wave * (32 * F_EROW);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int q = q0 + wr * 128 + i * 16 + fg * 4 + e;
        bv[e] = q < a.K ? a.bias[q] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf16x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[i][j][e] + bv[e];
        if (a.relu) v = fmaxf(v, 0.f);
        o[e] = (__bf16)v;
      }
      *reinterpret_cast<bf16x4_t*>(wl + (j * 16 + fr) * F_EROW + (i * 16 + fg * 4) * 2) = o;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave reads back only its own region
  const int erow = lane >> 4, ech = lane & 15;
  const int q = q0 + wr * 128 + ech * 8;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int px = it * 4 + erow;
    const uint32_t gp = (uint32_t)(p0 + wc * 32 + px);
    if (gp < (uint32_t)a.P && q < a.K) {
      const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(wl + px * F_EROW + ech * 16);
      *reinterpret_cast<bf16x8_t*>(a.y + (long long)gp * a.K + q) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------ weight gradient
constexpr int G_UNIT = 64 * 256;               // [64 px][128 ch] bf16                                16 KB
constexpr int G_STAGE = 3 * G_UNIT;            // dY q 0..127 | dY q 128..255 | sampled 128 channels    48 KB
constexpr int G_TAPS = 2 * G_STAGE;            // then 2 x [64 px][12 dwords] sampling states
constexpr int G_TAPROW = 12;
constexpr int G_LDS = G_TAPS + 2 * 64 * G_TAPROW * 4;
constexpr int G_SLAB = 256 * 128;              // floats per partial tile

template <int OFF>
__device__ __forceinline__ s16x4_t g_tr_read(uint32_t addr) {
  s16x4_t r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
__device__ __forceinline__ bf16x8_t g_pack8(s16x4_t lo, s16x4_t hi) {
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// One workgroup = (output tile [256 q][tap][128 c], pixel split z).  K-tile = 64 pixels: dY rows by LDS-DMA, the sampled rows by
// the VALU gather of the forward kernel (16 consecutive threads = the 256-B run of one pixel), both consumed through
// ds_read_b64_tr_b16.  The sampling state of a pixel is computed ONCE (threads 0..63, one K-tile ahead) and shared through LDS.
__global__ __launch_bounds__(512, 2) void dcn_wgrad_fused_kernel(const DcnFArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;                 // wave tile: q wr*64 + [0,64), c wc*64 + [0,64)
  uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
  const int taps = a.KH * a.KW, cpg = a.C / a.DG;
  const int tiles = a.QT * a.CT * taps;
  const int tile = (int)(bid % (uint32_t)tiles), z = (int)(bid / (uint32_t)tiles);
  const int tap = tile % taps, ct = (tile / taps) % a.CT, qt = tile / (taps * a.CT);
  const int q0 = qt * 256, c0 = ct * 128;
  const int g = c0 / cpg;
  const int kidx = g * taps + tap;
  const int KT = (a.P + 63) >> 6;
  const int kt0 = z * a.kt_per_split;
  int T = KT - kt0; if (T > a.kt_per_split) T = a.kt_per_split;

  auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x), 0, a.x_bytes, 0x00020000);
  auto yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.dy), 0, a.dy_bytes, 0x00020000);
  uint32_t* tapbuf = reinterpret_cast<uint32_t*>(smem + G_TAPS);

  // ---- sampling states: thread t < 64 owns pixel row t of the K-tile two steps ahead
  RawOff ro{0.f, 0.f, 1.f};
  auto load_raw = [&](int kt) {              // request dy / dx / mask of pixel row `tid` of K-tile kt
    if (tid < 64) {
      const uint32_t gp = (uint32_t)((kt0 + kt) * 64 + tid);
      ro = load_off(a, kt < T && gp < (uint32_t)a.P, (long long)gp, kidx);
    }
  };
  auto publish_tap = [&](int kt) {           // sampling state of pixel row `tid` of K-tile kt -> tapbuf[kt & 1]
    if (tid < 64) {
      const uint32_t gp = (uint32_t)((kt0 + kt) * 64 + tid);
      const bool pvalid = kt < T && gp < (uint32_t)a.P;
      const uint32_t gc = pvalid ? gp : 0u;
      const uint32_t n = fd_div(gc, a.div_hw);
      const uint32_t rem = gc - n * a.div_hw.d;
      const int ho = (int)fd_div(rem, a.div_w), wo = (int)(rem - (uint32_t)ho * a.div_w.d);
      const Tap t = make_tap(a, pvalid, n, ho, wo, ro, tap);
      uint32_t* d = tapbuf + ((kt & 1) * 64 + tid) * G_TAPROW;
      d[0] = t.o00; d[1] = t.o01; d[2] = t.o10; d[3] = t.o11;
      d[4] = __float_as_uint(t.w00); d[5] = __float_as_uint(t.w01); d[6] = __float_as_uint(t.w10); d[7] = __float_as_uint(t.w11);
      d[8] = __float_as_uint(t.m);
    }
  };
  // ---- gather ownership: 16-B chunk gch of pixel rows gpx0 and gpx0 + 32
  const int gch = tid & 15, gpx0 = tid >> 4;
  const uint32_t cadd = (uint32_t)(c0 + gch * 8) * 2u;
  Tap tp[2];
  u32x4_t rg[2][4];
  auto issue_gather = [&](int kt) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const uint32_t* d = tapbuf + ((kt & 1) * 64 + gpx0 + 32 * s) * G_TAPROW;
      tp[s].o00 = d[0]; tp[s].o01 = d[1]; tp[s].o10 = d[2]; tp[s].o11 = d[3];
      tp[s].w00 = __uint_as_float(d[4]); tp[s].w01 = __uint_as_float(d[5]); tp[s].w10 = __uint_as_float(d[6]); tp[s].w11 = __uint_as_float(d[7]);
      tp[s].m = __uint_as_float(d[8]);
      rg[s][0] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, tp[s].o00 + cadd, 0, 0);
      rg[s][1] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, tp[s].o01 + cadd, 0, 0);
      rg[s][2] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, tp[s].o10 + cadd, 0, 0);
      rg[s][3] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, tp[s].o11 + cadd, 0, 0);
    }
  };
  auto write_gather = [&](int stage) {
    char* bt = smem + stage * G_STAGE + 2 * G_UNIT;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int row = gpx0 + 32 * s;
      const int tswz = (row & 3) | (((row >> 3) & 1) << 2);
      const bf16x8_t o = interp8(tp[s], rg[s][0], rg[s][1], rg[s][2], rg[s][3]);
      *reinterpret_cast<bf16x8_t*>(bt + row * 256 + (((((gch >> 1) ^ tswz) << 1) | (gch & 1)) << 4)) = o;
    }
  };
  // ---- dY staging: two units [64 px][128 q], one wave instruction = 4 pixel rows x 256 B; 32 instructions per K-tile, 4 per wave
  const int srow = lane >> 4, spos = lane & 15;
  auto issue_dy = [&](int kt, int stage) {
    char* dst = smem + stage * G_STAGE + wave * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ii = j * 8 + wave;               // 0..31: unit = ii >> 4, rows (ii & 15) * 4 + srow
      const int row = (ii & 15) * 4 + srow, u = ii >> 4;
      const int tswz = (row & 3) | (((row >> 3) & 1) << 2);
      const int chunk = spos ^ (tswz << 1);
      const uint32_t gp = (uint32_t)((kt0 + kt) * 64 + row);
      const int q = q0 + u * 128 + chunk * 8;
      const uint32_t voff = (kt < T && gp < (uint32_t)a.P && q < a.K) ? (gp * (uint32_t)a.K + (uint32_t)q) * 2u : SOD_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(yrsrc, SOD_LDS(dst + j * 8192), 16, voff, 0, 0, 0);
    }
  };

  // ---- transposed fragment reads
  const int tq = (lane & 15) >> 2, tpp = lane & 3, tg = lane >> 4;
  const int rswz = tq | ((tg & 1) << 2);
  const uint32_t lds0 = (uint32_t)(uintptr_t)SOD_LDS(smem);
  uint32_t aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    aoff[i] = (uint32_t)((wr >> 1) * G_UNIT) + (uint32_t)(8 * tg + tq) * 256u + (uint32_t)(((((wr & 1) * 4 + i) ^ rswz)) * 32) + tpp * 8;
    boff[i] = (uint32_t)(2 * G_UNIT) + (uint32_t)(8 * tg + tq) * 256u + (uint32_t)((((wc * 4 + i) ^ rswz)) * 32) + tpp * 8;
  }
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- prologue
  load_raw(0);
  publish_tap(0);
  load_raw(1);
  __syncthreads();
  issue_gather(0);
  issue_dy(0, 0);
  publish_tap(1);
  load_raw(2);
  write_gather(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int k = 0; k < T; ++k) {
    const uint32_t cur = lds0 + (uint32_t)((k & 1) * G_STAGE);
    const bool more = k + 1 < T;
    if (more) {
      issue_gather(k + 1);                       // reads tapbuf[(k+1)&1], published during iteration k-1
      issue_dy(k + 1, (k + 1) & 1);
    }
    s16x4_t ar[4][2][2], br[4][2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ar[i][0][0] = g_tr_read<0>(cur + aoff[i]); ar[i][0][1] = g_tr_read<1024>(cur + aoff[i]);
      ar[i][1][0] = g_tr_read<8192>(cur + aoff[i]); ar[i][1][1] = g_tr_read<8192 + 1024>(cur + aoff[i]);
      br[i][0][0] = g_tr_read<0>(cur + boff[i]); br[i][0][1] = g_tr_read<1024>(cur + boff[i]);
      br[i][1][0] = g_tr_read<8192>(cur + boff[i]); br[i][1][1] = g_tr_read<8192 + 1024>(cur + boff[i]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    bf16x8_t af[4][2], bf[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      af[i][0] = g_pack8(ar[i][0][0], ar[i][0][1]); af[i][1] = g_pack8(ar[i][1][0], ar[i][1][1]);
      bf[i][0] = g_pack8(br[i][0][0], br[i][0][1]); bf[i][1] = g_pack8(br[i][1][0], br[i][1][1]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
      }
    publish_tap(k + 2);                          // into tapbuf[k & 1]: its old content (K-tile k) is in registers since iteration k-1
    load_raw(k + 3);
    if (more) write_gather((k + 1) & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: the partial tile goes to the workspace in fragment order
  float* slab = a.partial + ((size_t)z * (size_t)tiles + (size_t)tile) * G_SLAB + (size_t)wave * (16 * 256) + (size_t)lane * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4_t*>(slab + (i * 4 + j) * 256) = acc[i][j];
}

__global__ __launch_bounds__(256) void dcn_wgrad_reduce_kernel(const DcnFArgs a) {
  const int taps = a.KH * a.KW, tiles = a.QT * a.CT * taps;
  const uint32_t idx = blockIdx.x * 256u + threadIdx.x;        // float4 index: < tiles * 8192
  const int lane = idx & 63, frag = (idx >> 6) & 127, tile = (int)(idx >> 13);
  if (tile >= tiles) return;
  const float* src = a.partial + (size_t)tile * G_SLAB + (size_t)(idx & 8191u) * 4;
  const size_t zstride = (size_t)tiles * G_SLAB;
  f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  int zz = 0;
  for (; zz + 2 <= a.nz; zz += 2) {
    s0 += *reinterpret_cast<const f32x4_t*>(src + (size_t)zz * zstride);
    s1 += *reinterpret_cast<const f32x4_t*>(src + (size_t)(zz + 1) * zstride);
  }
  for (; zz < a.nz; ++zz) s0 += *reinterpret_cast<const f32x4_t*>(src + (size_t)zz * zstride);
  const f32x4_t sum = s0 + s1;
  const int wave = frag >> 4, i = (frag >> 2) & 3, j = frag & 3;
  const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fg = lane >> 4;
  const int tap = tile % taps, ct = (tile / taps) % a.CT, qt = tile / (taps * a.CT);
  const int c = ct * 128 + wc * 64 + j * 16 + fr;
  if (c >= a.C) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int q = qt * 256 + wr * 64 + i * 16 + fg * 4 + e;
    if (q < a.K) a.dw[((size_t)q * taps + tap) * a.C + c] += a.qscale ? sum[e] * a.qscale[q] : sum[e];
  }
}

// fp32 reference-precision forward (test mode): one thread per output element, fp32 input / weights / accumulation, the sampling
// rule of make_tap.  It exists so that the op's SEMANTICS can be checked on the device at fp32 tolerance (the reference's known-answer
// vectors, tests/test_deformable_conv.py:67-87) independently of bf16 operand rounding; it is not a fast path.
__global__ __launch_bounds__(256) void dcn_fwd_f32_kernel(const float* __restrict__ x, const float* __restrict__ off, const float* __restrict__ mask,
                                                          const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
                                                          int N, int H, int W, int C, int Ho, int Wo, int K, int KH, int KW, int stride, int pad, int dil,
                                                          int DG, int off_ld, int mask_ld, int mask_logit) {
  const long long total = (long long)N * Ho * Wo * K;
  const int taps = KH * KW, cpg = C / DG;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int k = (int)(i % K);
    const long long pix = i / K;
    const int wo = (int)(pix % Wo), ho = (int)((pix / Wo) % Ho), n = (int)(pix / ((long long)Wo * Ho));
    float acc = bias ? bias[k] : 0.f;
    for (int g = 0; g < DG; ++g)
      for (int tap = 0; tap < taps; ++tap) {
        const int kk = g * taps + tap, ki = tap / KW, kj = tap - ki * KW;
        const float py = (float)(ho * stride - pad + ki * dil) + off[pix * off_ld + 2 * kk];
        const float px = (float)(wo * stride - pad + kj * dil) + off[pix * off_ld + 2 * kk + 1];
        if (!(py > -1.f && px > -1.f && py < (float)H && px < (float)W)) continue;
        float m = mask ? mask[pix * mask_ld + kk] : 1.f;
        if (mask && mask_logit) m = 1.f / (1.f + expf(-m));
        const float fy = floorf(py), fx = floorf(px);
        const int yl = (int)fy, xl = (int)fx, yh = yl + 1, xh = xl + 1;
        const float ly = py - fy, lx = px - fx, hy = 1.f - ly, hx = 1.f - lx;
        const float* xb = x + (long long)n * H * W * C;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
          float v = 0.f;
          if (yl >= 0 && xl >= 0) v += hy * hx * xb[((long long)yl * W + xl) * C + c];
          if (yl >= 0 && xh <= W - 1) v += hy * lx * xb[((long long)yl * W + xh) * C + c];
          if (yh <= H - 1 && xl >= 0) v += ly * hx * xb[((long long)yh * W + xl) * C + c];
          if (yh <= H - 1 && xh <= W - 1) v += ly * lx * xb[((long long)yh * W + xh) * C + c];
          acc += w[((long long)k * taps + tap) * C + c] * (v * m);
        }
      }
    y[i] = acc;
  }
}

int dcn_fill(DcnFArgs& a, const void* x, const float* offset, const float* mask, int N, int H, int W, int C, int K, int KH, int KW,
             int stride, int pad, int dil, int DG, int off_ld, int mask_ld, int mask_is_logit) {
  if (!x || !offset || N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || dil <= 0 || pad < 0 || DG <= 0) return SOD_EARG;
  if ((C & 63) || (K & 7) || C % DG || ((C / DG) & 63) || KH * KW > 49) return SOD_EARG;
  a.x = (const __bf16*)x; a.off = offset; a.mask = mask;
  a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.dil = dil; a.DG = DG;
  a.Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  a.Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  if (a.Ho <= 0 || a.Wo <= 0) return SOD_EARG;
  a.off_ld = off_ld > 0 ? off_ld : 2 * KH * KW * DG;
  a.mask_ld = mask_ld > 0 ? mask_ld : KH * KW * DG;
  if (a.off_ld < 2 * KH * KW * DG || a.mask_ld < KH * KW * DG) return SOD_EARG;
  a.mask_logit = mask_is_logit;
  const unsigned long long xb = (unsigned long long)N * H * W * C * 2ull, wb = (unsigned long long)K * KH * KW * C * 2ull;
  const long long P = (long long)N * a.Ho * a.Wo;
  if (xb >= 0x80000000ull || wb >= 0x80000000ull || P >= (1ll << 30) || (unsigned long long)P * K * 2ull >= 0x80000000ull) return SOD_ESIZE;
  a.x_bytes = (uint32_t)xb; a.w_bytes = (uint32_t)wb; a.dy_bytes = (uint32_t)((unsigned long long)P * K * 2ull);
  a.P = (int)P;
  a.div_hw = make_fastdiv((uint32_t)(a.Ho * a.Wo));
  a.div_w = make_fastdiv((uint32_t)a.Wo);
  a.div_kw = make_fastdiv((uint32_t)KW);
  return SOD_OK;
}

}  // namespace

extern "C" int sod_deform_conv_fwd_fused(const void* x, const float* offset, const float* mask, const void* w, const float* bias, void* y,
                                         int N, int H, int W, int C, int K, int KH, int KW, int stride, int pad, int dil, int deformable_groups,
                                         int off_ld, int mask_ld, int mask_is_logit, int relu, void* stream) {
  if (!w || !y) return SOD_EARG;
  DcnFArgs a{};
  const int rc = dcn_fill(a, x, offset, mask, N, H, W, C, K, KH, KW, stride, pad, dil, deformable_groups, off_ld, mask_ld, mask_is_logit);
  if (rc) return rc;
  a.w = (const __bf16*)w; a.bias = bias; a.y = (__bf16*)y; a.relu = relu;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)dcn_fwd_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  const int nq = (K + 255) / 256, np = (a.P + 127) / 128;
  SOD_LAUNCH(dcn_fwd_fused_kernel, dim3(nq * np), dim3(512), F_LDS, (hipStream_t)stream, a);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_deform_conv_wgrad_fused(const void* dy, const void* x, const float* offset, const float* mask, float* dw, const float* qscale,
                                           int N, int H, int W, int C, int K, int KH, int KW, int stride, int pad, int dil, int deformable_groups,
                                           int off_ld, int mask_ld, int mask_is_logit, void* ws, long long ws_bytes, void* stream) {
  if (!dy || !dw || !ws || ws_bytes <= 0 || ((uintptr_t)ws & 15)) return SOD_EARG;
  DcnFArgs a{};
  const int rc = dcn_fill(a, x, offset, mask, N, H, W, C, K, KH, KW, stride, pad, dil, deformable_groups, off_ld, mask_ld, mask_is_logit);
  if (rc) return rc;
  if ((C & 127) && C != 64) return SOD_EARG;          // 128-channel tiles (a lone 64-channel input runs as one half-empty tile)
  if (((C / deformable_groups) & 127) && deformable_groups != 1) return SOD_EARG;    // a tile must not straddle deformable groups
  a.dy = (const __bf16*)dy; a.dw = dw; a.qscale = qscale; a.partial = (float*)ws;
  a.QT = (K + 255) / 256; a.CT = (C + 127) / 128;
  const int tiles = a.QT * a.CT * KH * KW;
  const int KT = (a.P + 63) / 64;
  int cus = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  int nz = cus / tiles;
  if (nz < 1) nz = 1;
  if (nz > KT) nz = KT;
  const int per = (KT + nz - 1) / nz;
  nz = (KT + per - 1) / per;
  a.nz = nz; a.kt_per_split = per;
  if ((long long)nz * tiles * G_SLAB * (long long)sizeof(float) > ws_bytes) return SOD_EARG;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)dcn_wgrad_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  hipStream_t st = (hipStream_t)stream;
  SOD_LAUNCH(dcn_wgrad_fused_kernel, dim3(nz * tiles), dim3(512), G_LDS, st, a);
  SOD_LAUNCH(dcn_wgrad_reduce_kernel, dim3(tiles * 32), dim3(256), 0, st, a);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_deform_conv_fwd_f32(const float* x, const float* offset, const float* mask, const float* w, const float* bias, float* y,
                                       int N, int H, int W, int C, int K, int KH, int KW, int stride, int pad, int dil, int deformable_groups,
                                       int off_ld, int mask_ld, int mask_is_logit, void* stream) {
  if (!x || !offset || !w || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || dil <= 0 || pad < 0) return SOD_EARG;
  if (deformable_groups <= 0 || C % deformable_groups) return SOD_EARG;
  const int Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return SOD_EARG;
  if (off_ld <= 0) off_ld = 2 * KH * KW * deformable_groups;
  if (mask_ld <= 0) mask_ld = KH * KW * deformable_groups;
  const long long total = (long long)N * Ho * Wo * K;
  long long g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  SOD_LAUNCH(dcn_fwd_f32_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, offset, mask, w, bias, y, N, H, W, C, Ho, Wo, K, KH, KW, stride, pad,
             dil, deformable_groups, off_ld, mask_ld, mask_is_logit);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}
