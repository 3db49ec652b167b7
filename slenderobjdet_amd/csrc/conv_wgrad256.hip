// Convolution weight gradient, 256(q) x 256(c) output tile per workgroup, 8 waves, 4-phase K-tile loop (gfx950).
//
//   dW[q][tap][c] += sum_p dY[p][q] * X[p shifted by tap][c]          (replaces ATen's conv backward-weight under the d2 ResNet/FPN
//   and FCOSHead convolutions: slender_det/modeling/backbone/fpn.py:94-115, slender_det/modeling/meta_arch/fcos/fcosv2.py:277-381)
//
// The contraction runs over pixels, the slow axis of both NHWC operands.  Structure = conv_igemm256.hip (one workgroup of 8 waves per
// CU, LDS-DMA staging that is never drained inside the loop, two wave rows staggered by one barrier) with transposed operand reads:
//
//   * wave (wr, wc) = (wave>>2, wave&3) owns 128(q) x 64(c) outputs = 8x4 MFMA 16x16x32 accumulators:
//       q in {u*128 + wr*64 + [0,64)}, c in {u*128 + wc*32 + [0,32)}, u = 0, 1;
//   * a K-tile = 64 pixels.  It is staged as FOUR 16-KB units [64 pixels][128 channels] (rows of 256 B, each one contiguous 256-B run
//     of global memory): Ya0 / Ya1 = dY channels q0 + {0..127} / {128..255}, Xb0 / Xb1 = X channels c0 + {0..127} / {128..255};
//     LDS = 2 K-tiles x 4 units = 128 KB; the 32-B chunks of a row are XOR-swizzled with (row&3) | ((row>>3)&1)<<2 on the SOURCE
//     side, which makes the transposing reads conflict-free;
//   * fragments come from ds_read_b64_tr_b16 (hardware transpose: a 16-lane group fetches 4 pixel rows x 16 channels and every lane
//     receives 4 consecutive pixels of ONE channel = half an MFMA operand register pair);
//   * phases and hazards as in conv_igemm256.hip: quadrants (a0,b0) (a0,b1) (a1,b1) (a1,b0), 16 MFMAs each, one unit (two LDS-DMA
//     instructions per thread) issued inside every MFMA cluster, `s_waitcnt vmcnt(8)` + raw s_barrier per phase; a unit issued in
//     phase p is retired by the wait of phase p+5 at the latest and first read in phase >= p+6 (round 6: one unit more in flight).
//
// Split over pixels: the launch is ONE workgroup per CU, tiles x nz blocks; every block stores its 256x256 fp32 partial tile as a
// slab (fragment order: 16-B stores, 1 KB per wave instruction) into the caller's workspace and wgrad256_reduce_kernel sums the nz
// slabs of a tile in fixed order and adds the result (x folded FrozenBN scale) into dW with plain read-modify-writes: no atomics,
// bit-identical from run to run.  The 9 taps of one pixel range are consecutive block ids (same XCD after the remap): they share the
// dY tile and overlapping X rows in that XCD's L2.
#include "conv_args.h"
#include <stdlib.h>

namespace sodconv {
namespace {

constexpr int WROWB = 256;             // bytes per LDS row: 128 channels of one pixel
constexpr int WUNIT = 64 * WROWB;      // 16 KB
constexpr int WBUF = 4 * WUNIT;        // one K-tile: [Ya0][Ya1][Xb0][Xb1]
constexpr int WLDS_BYTES = 2 * WBUF;   // 128 KB
constexpr int SLAB = 256 * 256;        // floats per partial tile

// SOD_W256_ABL (measurement builds only, tools/bench_wgrad_shapes.py): 1 = no LDS-DMA, 2 = no fragment reads,
// 4 = no MFMAs, 8 = no slab stores.  Never defined in the shipped library.
#ifndef SOD_W256_ABL
#define SOD_W256_ABL 0
#endif

template <int OFF>
__device__ __forceinline__ s16x4_t tr_read(uint32_t addr) {
  s16x4_t r;
#if SOD_W256_ABL & 2
  asm volatile("; no read %0 %1" : "=v"(r) : "v"(addr));
#else
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
#endif
  return r;
}

__device__ __forceinline__ bf16x8_t pack8(s16x4_t lo, s16x4_t hi) {
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

template <int SA, int SB, int KS>
__device__ __forceinline__ void wmma_half(f32x4_t (&acc)[8][4], const bf16x8_t (&af)[4][2], const bf16x8_t (&bf)[2][2]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#if SOD_W256_ABL & 4
      asm volatile("; no mfma %0 %1 %2" : "+v"(acc[SA * 4 + i][SB * 2 + j]) : "v"(af[i][KS]), "v"(bf[j][KS]));
#else
      acc[SA * 4 + i][SB * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][KS], bf[j][KS], acc[SA * 4 + i][SB * 2 + j], 0, 0, 0);
#endif
}

// PACK runs after the lgkmcnt(0) + sched_barrier: nothing that touches the raw fragment registers may be scheduled above the wait
// (the transposing reads are inline asm, invisible to the compiler's own wait insertion).
#if SOD_W256_ABL & 32
#define SODW_PRIO(x)
#else
#define SODW_PRIO(x) __builtin_amdgcn_s_setprio(x)
#endif
#if SOD_W256_ABL & 128
#define SODW_BAR_B
#else
#define SODW_BAR_B __builtin_amdgcn_s_barrier()
#endif
#define SODW_PHASE(SA, SB, BFR, PACK, STAGE)                              \
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                        \
  __builtin_amdgcn_s_barrier();                                           \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      \
  __builtin_amdgcn_sched_barrier(0);                                      \
  PACK;                                                                   \
  SODW_PRIO(1);                                                           \
  wmma_half<SA, SB, 0>(acc, af, BFR);                                     \
  STAGE;                                                                  \
  wmma_half<SA, SB, 1>(acc, af, BFR);                                     \
  SODW_PRIO(0);                                                           \
  SODW_BAR_B;

__global__ __launch_bounds__(512, 2) void conv_wgrad256_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
  const int RS = a.R * a.S;
  const int tile = (int)(bid % (uint32_t)(a.QT * a.CT * RS));
  const int z = (int)(bid / (uint32_t)(a.QT * a.CT * RS));
  const int tap = tile % RS;
  const int ct = (tile / RS) % a.CT, qt = tile / (RS * a.CT);
  const int r = tap / a.S, s = tap - r * a.S;
  const int q0 = qt * 256, c0 = ct * 256;
  const int vbeg = z * a.v_per_split;
  int vend = vbeg + a.v_per_split; if (vend > a.V) vend = a.V;
  const int T = (vend - vbeg) >> 6;

  // ---- staging geometry: one wave instruction = 4 pixel rows x 256 B; lane -> (row in the group of 4, 16-B slot)
  const int srow = lane >> 4, spos = lane & 15;
  const int sswz = srow | (((wave >> 1) & 1) << 2);          // (L&3) | ((L>>3)&1)<<2 for L = (j*8+wave)*4+srow
  const int schunk = spos ^ (sswz << 1);                     // logical 16-B chunk (8 channels) this lane fetches
  const uint32_t qadd0 = (uint32_t)(q0 + schunk * 8) * 2u, cadd0 = (uint32_t)(c0 + schunk * 8) * 2u;   // unit 1: + 256 B

  // ---- pixel rows.  A thread stages pixel rows L_j = (j*8+wave)*4+srow, j = 0, 1, of every K-tile and needs their byte offsets (dY row, X
  // row shifted by the tap; out-of-range = zero fill).  Tracked incrementally per thread for both rows they cost ~100 vector instructions
  // per wave and K-tile - 2 000 of the loop's 4 800 cycles per K-tile, more than the MFMA cluster they ran in can hide (in-kernel
  // stamps, tools/bench_wgrad256_stamps.py).  Now every lane computes ONE row from scratch - two divisions, no carried state: even 16-B
  // slots row L_0, odd slots row L_1 - and takes the other one from its neighbour lane (DPP quad permute).
  int cur_lv = 0;
#pragma unroll
  for (int i = 1; i < MAXLEV; ++i)
    if (i < a.nlev && vbeg >= a.lev[i].v0) cur_lv = i;
  WLevel g = a.lev[cur_lv];
  int next_v0 = (cur_lv + 1 < a.nlev) ? a.lev[cur_lv + 1].v0 : 0x7fffffff;
  auto yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.dy), 0, g.dy_bytes, 0x00020000);
  auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.x), 0, g.x_bytes, 0x00020000);
  const int dh = r * a.dil - a.pad, dw = s * a.dil - a.pad;
  const bool odd = spos & 1;
  const int myrow = ((odd ? 8 : 0) + wave) * 4 + srow;
  uint32_t voy[2], vox[2];
  auto rows = [&](int kt) {              // row offsets of K-tile kt (+ the level switch, wave-uniform)
#if SOD_W256_ABL & 64
    if (kt > 1) return;
#endif
    uint32_t oy = SOD_OOB, ox = SOD_OOB;
    const int v = vbeg + kt * 64;
    if (kt < T) {
      if (v >= next_v0) {
        while (v >= next_v0) { ++cur_lv; next_v0 = (cur_lv + 1 < a.nlev) ? a.lev[cur_lv + 1].v0 : 0x7fffffff; }
        g = a.lev[cur_lv];
        yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.dy), 0, g.dy_bytes, 0x00020000);
        xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.x), 0, g.x_bytes, 0x00020000);
      }
      const int p = v - g.v0 + myrow;
      if (p < g.P) {
        const uint32_t n = fd_div((uint32_t)p, g.div_hw);
        const uint32_t rem = (uint32_t)p - n * g.div_hw.d;
        const uint32_t ho = fd_div(rem, g.div_w);
        const uint32_t wo = rem - ho * g.div_w.d;
        const int hi = (int)ho * a.stride + dh, wi = (int)wo * a.stride + dw;
        oy = (n * (uint32_t)g.dy_img_stride + rem * (uint32_t)a.K) * 2u;
        if (((unsigned)hi < (unsigned)g.Hx) & ((unsigned)wi < (unsigned)g.Wx))
          ox = (n * (uint32_t)g.x_img_stride + (uint32_t)(hi * g.Wx + wi) * (uint32_t)a.C) * 2u;
      }
    }
    // neighbour lane's pair: quad_perm [1, 0, 3, 2] = 0xB1
    const uint32_t ny = (uint32_t)__builtin_amdgcn_mov_dpp((int)oy, 0xB1, 0xF, 0xF, true);
    const uint32_t nx = (uint32_t)__builtin_amdgcn_mov_dpp((int)ox, 0xB1, 0xF, 0xF, true);
    voy[0] = odd ? ny : oy; vox[0] = odd ? nx : ox;
    voy[1] = odd ? oy : ny; vox[1] = odd ? ox : nx;
  };

  // K need not be a multiple of 256 (RetinaNet / AnchorHead class scores: 9 anchors x 80 classes = 720): a lane whose 8 output channels lie
  // beyond K requests the out-of-range offset (zero fill) - the next channels in memory belong to the following PIXEL.
  const bool qv[2] = {q0 + schunk * 8 < a.K, q0 + 128 + schunk * 8 < a.K};
  auto stage_a = [&](int u, int kt) {     // dY channels q0 + u*128 + [0,128) of K-tile kt (row state must describe kt)
    char* dst = smem + (kt & 1) * WBUF + u * WUNIT + wave * 1024;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t off = (qv[u] && voy[j] != SOD_OOB) ? voy[j] + qadd0 + (uint32_t)(u * 256) : SOD_OOB;
#if SOD_W256_ABL & 1
      asm volatile("; no dma %0 %1" :: "v"(off), "v"(dst));
#else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(yrsrc, SOD_LDS(dst + j * 8192), 16, off, 0, 0, 0);
#endif
    }
  };
  auto stage_b = [&](int u, int kt) {     // X channels c0 + u*128 + [0,128)
    char* dst = smem + (kt & 1) * WBUF + (2 + u) * WUNIT + wave * 1024;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#if SOD_W256_ABL & 1
      asm volatile("; no dma %0 %1" :: "v"(vox[j] + cadd0), "v"(dst));
#else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, SOD_LDS(dst + j * 8192), 16, vox[j] + cadd0 + (uint32_t)(u * 256), 0, 0, 0);
#endif
    }
  };

  // ---- transposed fragment reads: lane 4q+p of a 16-lane group addresses pixel row q, channels 4p..4p+3 of the 16-channel block
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = lane >> 4;
  const int tswz = tq | ((tg & 1) << 2);
  const uint32_t lds0 = (uint32_t)(uintptr_t)SOD_LDS(smem);
  uint32_t aoff[4], boff[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = (uint32_t)(8 * tg + tq) * 256u + (uint32_t)(((wr * 4 + i) ^ tswz) * 32) + tp * 8;
#pragma unroll
  for (int j = 0; j < 2; ++j) boff[j] = 2 * WUNIT + (uint32_t)(8 * tg + tq) * 256u + (uint32_t)(((wc * 2 + j) ^ tswz) * 32) + tp * 8;

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t af[4][2], bf0[2][2], bf1[2][2];
  s16x4_t ar[4][2][2], br[2][2][2];       // raw halves [tile][k-step][lo/hi]

#define SODW_READ_A(BASE, U)                                                                   \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                              \
    const uint32_t ad = (BASE) + (U) * WUNIT + aoff[i];                                        \
    ar[i][0][0] = tr_read<0>(ad); ar[i][0][1] = tr_read<1024>(ad);                             \
    ar[i][1][0] = tr_read<8192>(ad); ar[i][1][1] = tr_read<8192 + 1024>(ad);                   \
  }
#define SODW_READ_B(BASE, U)                                                                   \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                              \
    const uint32_t ad = (BASE) + (U) * WUNIT + boff[j];                                        \
    br[j][0][0] = tr_read<0>(ad); br[j][0][1] = tr_read<1024>(ad);                             \
    br[j][1][0] = tr_read<8192>(ad); br[j][1][1] = tr_read<8192 + 1024>(ad);                   \
  }
#define SODW_PACK_A                                                                            \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) { af[i][0] = pack8(ar[i][0][0], ar[i][0][1]); af[i][1] = pack8(ar[i][1][0], ar[i][1][1]); }
#define SODW_PACK_B(BFR)                                                                       \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) { BFR[j][0] = pack8(br[j][0][0], br[j][0][1]); BFR[j][1] = pack8(br[j][1][0], br[j][1][1]); }

#if SOD_W256_ABL & 16      // measurement build: wall-clock stamps (100 MHz) around the prologue, the K loop and the epilogue
  const unsigned long long st0 = __builtin_amdgcn_s_memrealtime(), sc0 = __builtin_amdgcn_s_memtime();
#endif
  // ---- prologue: K-tile 0 complete, first two units of K-tile 1
  rows(0);
  // Every unit is re-requested one phase after its last read (Xb0 / Ya0 are last read in phase 0, Xb1 in 1, Ya1 in 2), i.e. six to seven
  // phases before its first read, and four units (64 KB) stay in flight behind every wait.  (Until round 6: five phases, three units behind
  // vmcnt(6); the deeper schedule measured 384.5 vs 394.8 us on the tower shape, results bit-identical.  Safe one phase after the read: the
  // other wave row runs one barrier behind, and a phase has two.)
  stage_b(0, 0); stage_a(0, 0); stage_b(1, 0); stage_a(1, 0);
  rows(1);
  stage_b(0, 1); stage_a(0, 1); stage_b(1, 1);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();       // wave row 1 runs one barrier behind wave row 0

#if SOD_W256_ABL & 16
  const unsigned long long st1 = __builtin_amdgcn_s_memrealtime(), sc1 = __builtin_amdgcn_s_memtime();
#endif
  for (int k = 0; k < T; ++k) {
    const uint32_t cur = lds0 + (uint32_t)((k & 1) * WBUF);
    // phase 0: quadrant (a0, b0)
    SODW_READ_B(cur, 0)
    SODW_READ_A(cur, 0)
    SODW_PHASE(0, 0, bf0, SODW_PACK_B(bf0) SODW_PACK_A, stage_a(1, k + 1))
    // phase 1: quadrant (a0, b1); the rows of K-tile k+2 are computed inside the MFMA cluster
    SODW_READ_B(cur, 1)
    SODW_PHASE(0, 1, bf1, SODW_PACK_B(bf1), rows(k + 2); stage_b(0, k + 2))
    // phase 2: quadrant (a1, b1)
    SODW_READ_A(cur, 1)
    SODW_PHASE(1, 1, bf1, SODW_PACK_A, stage_a(0, k + 2))
    // phase 3: quadrant (a1, b0), b0 still in registers
    SODW_PHASE(1, 0, bf0, , stage_b(1, k + 2))
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if SOD_W256_ABL & 16
  const unsigned long long st2 = __builtin_amdgcn_s_memrealtime(), sc2 = __builtin_amdgcn_s_memtime();
#endif

  // ---- epilogue: the partial tile goes to the workspace in FRAGMENT order (16 B per lane, 1 KB per wave instruction)
  float* slab = a.partial + ((size_t)z * (size_t)(a.QT * a.CT * RS) + (size_t)tile) * SLAB + (size_t)wave * (32 * 256) + (size_t)lane * 4;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#if SOD_W256_ABL & 8
      if (acc[i][j][0] == 123.456f) *reinterpret_cast<f32x4_t*>(slab + (i * 4 + j) * 256) = acc[i][j];
#else
      *reinterpret_cast<f32x4_t*>(slab + (i * 4 + j) * 256) = acc[i][j];
#endif
    }
#if SOD_W256_ABL & 16
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0) {
    const unsigned long long st3 = __builtin_amdgcn_s_memrealtime();
    float* o = a.partial + ((size_t)z * (size_t)(a.QT * a.CT * RS) + (size_t)tile) * SLAB;
    o[0] = (float)(st1 - st0); o[1] = (float)(st2 - st1); o[2] = (float)(st3 - st2); o[3] = (float)T;
    o[64 * 4] = (float)(sc2 - sc1);      // shader cycles of the K loop (lane 0 of the NEXT wave instruction's row: still wave 0's slab part)
    o[64 * 4 + 1] = (float)(st0 & 0xffffff);
  }
#endif
}

// Sums the nz slabs of every tile in z order and adds the result into dW.  One thread = one float4 of the fragment-ordered slab.
__global__ __launch_bounds__(256) void wgrad256_reduce_kernel(const WgradArgs a) {
  const int RS = a.R * a.S, tiles = a.QT * a.CT * RS;
  const uint32_t idx = blockIdx.x * 256u + threadIdx.x;        // < tiles * 16384
  const int lane = idx & 63, frag = (idx >> 6) & 255, tile = (int)(idx >> 14);
  if (tile >= tiles) return;
  const float* src = a.partial + (size_t)tile * SLAB + (size_t)(idx & 16383u) * 4;
  const size_t zstride = (size_t)tiles * SLAB;
  f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  int zz = 0;
  for (; zz + 4 <= a.nz; zz += 4) {     // four independent loads in flight; the summation ORDER is fixed by the code, not by timing
    const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(src + (size_t)zz * zstride);
    const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(src + (size_t)(zz + 1) * zstride);
    const f32x4_t v2 = *reinterpret_cast<const f32x4_t*>(src + (size_t)(zz + 2) * zstride);
    const f32x4_t v3 = *reinterpret_cast<const f32x4_t*>(src + (size_t)(zz + 3) * zstride);
    s0 += v0; s1 += v1; s2 += v2; s3 += v3;
  }
  for (; zz < a.nz; ++zz) s0 += *reinterpret_cast<const f32x4_t*>(src + (size_t)zz * zstride);
  const f32x4_t sum = (s0 + s1) + (s2 + s3);
  const int wave = frag >> 5, i = (frag >> 2) & 7, j = frag & 3;
  const int wr = wave >> 2, wc = wave & 3, fr = lane & 15, fg = lane >> 4;
  const int tap = tile % RS, ct = (tile / RS) % a.CT, qt = tile / (RS * a.CT);
  const int c = ct * 256 + (j >> 1) * 128 + wc * 32 + (j & 1) * 16 + fr;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int q = qt * 256 + (i >> 2) * 128 + wr * 64 + (i & 3) * 16 + fg * 4 + e;
    if (q >= a.K) continue;                 // rows of the last q-tile beyond K (their slab entries are zeros)
    float* dst = a.dw + ((size_t)q * RS + tap) * a.C + c;
    *dst += sum[e] * (a.qscale ? a.qscale[q] : 1.f);
  }
}

int splits_for(const WgradArgs& a, int cus, int* vps_out) {
  const int tiles = ((a.K + 255) / 256) * (a.C / 256) * a.R * a.S;
  int V = 0;
  for (int l = 0; l < a.nlev; ++l) V += (a.lev[l].P + 63) / 64 * 64;
  const int kt = V / 64;
  int nz = cus / tiles;
  if (nz < 1) nz = 1;
  if (nz > kt) nz = kt;
  int per = (kt + nz - 1) / nz;          // K-tiles per block
  nz = (kt + per - 1) / per;
  *vps_out = per * 64;
  return nz;
}

}  // namespace

bool wgrad256_supported(const WgradArgs& a) {
  if ((a.K & 7) || a.K < 256 || (a.C & 255)) return false;      // K: any multiple of 8 from 256 up (the last q-tile is masked)
  long long V = 0;
  for (int l = 0; l < a.nlev; ++l) V += (a.lev[l].P + 63) / 64 * 64;
  return V >= 64 && V < (1ll << 30);
}

long long wgrad256_workspace_bytes(const WgradArgs& a, int cus) {
  int vps = 0;
  const int nz = splits_for(a, cus, &vps);
  const long long tiles = (long long)((a.K + 255) / 256) * (a.C / 256) * a.R * a.S;
  return (long long)nz * tiles * SLAB * (long long)sizeof(float);
}

int launch_wgrad256(WgradArgs& a, int cus, float* ws, long long ws_bytes, hipStream_t st) {
  if (!wgrad256_supported(a)) return SOD_EARG;
  a.QT = (a.K + 255) / 256; a.CT = a.C / 256;
  const int tiles = a.QT * a.CT * a.R * a.S;
  int V = 0;
  for (int l = 0; l < a.nlev; ++l) {
    a.lev[l].v0 = V;
    V += (a.lev[l].P + 63) / 64 * 64;
  }
  a.V = V;
  int vps = 0;
  a.nz = splits_for(a, cus, &vps);
  a.v_per_split = vps;
  a.div_s = make_fastdiv((uint32_t)a.S);
  const long long need = (long long)a.nz * tiles * SLAB * (long long)sizeof(float);
  if (!ws || need > ws_bytes) return SOD_EARG;
  if ((long long)tiles * 16384 >= (1ll << 31)) return SOD_ESIZE;
  a.partial = ws;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_wgrad256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WLDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  SOD_LAUNCH(conv_wgrad256_kernel, dim3(a.nz * tiles), dim3(512), WLDS_BYTES, st, a);
  SOD_LAUNCH(wgrad256_reduce_kernel, dim3(tiles * 64), dim3(256), 0, st, a);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

}  // namespace sodconv
