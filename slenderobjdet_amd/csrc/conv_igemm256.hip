// Implicit-GEMM convolution, 256(q) x 256(p) x 64 tile, 8 waves, 8-phase main loop (gfx950).
//
// Same math, layouts and epilogue contract as conv_igemm.hip (forward and stride-1 data gradient of the FCOS/RetinaNet
// towers and the other large 3x3 convolutions; reference: slender_det/modeling/meta_arch/fcos/fcosv2.py:277-381), for the
// compute-bound shapes only.  The 128x128 kernel stalls once per K-step on vmcnt(0)+barrier with two workgroups per CU; this
// one keeps ONE workgroup of 8 waves per CU and never drains the LDS-DMA queue inside the loop:
//
//   * wave (wr, wc) = (wave>>2, wave&3) owns a 128(q) x 64(p) output block = 8x4 MFMA 16x16x32 accumulators;
//   * a K-tile (64 contraction elements) is staged as FOUR 16-KB units, cut along the order the waves consume them:
//       Ua0 / Ua1 = weight rows {0..63} / {64..127} of both wave rows,  Ub0 / Ub1 = pixel rows {0..31} / {32..63} of all four
//       wave columns;  LDS = 2 K-tiles x 4 units = 128 KB, rows of 128 B, 16-B chunks XOR-swizzled on the SOURCE side;
//   * a K-tile is computed in 4 phases (quadrants (a0,b0) (a0,b1) (a1,b1) (a1,b0), 16 MFMAs each); every phase
//       { ds_read the operands first needed now | s_waitcnt vmcnt(6) | s_barrier | lgkmcnt(0) |
//         8 MFMA | issue ONE unit of a later K-tile (2 LDS-DMA per thread) | 8 MFMA | s_barrier }
//     so three to four units (most of a K-tile) are always in flight across the barriers;
//   * the two wave rows run staggered by one barrier: while waves 0-3 issue MFMAs, waves 4-7 (their SIMD partners) read LDS and
//     issue loads, and vice versa.
//
// Hazard bookkeeping (phases numbered globally, p = 4*tile + i):
//   RAW: a unit issued in phase p (inside its MFMA cluster) is retired by the vmcnt(6) at the top of phase p+4 (the 3 younger
//        units = 6 loads may remain) and first read in phase >= p+5 - one phase after the wait, which holds for both staggered
//        wave rows.
//        issue/first-read: Ua0 4k+2 / 4k+8, Ub0 4k+3 / 4k+8, Ub1 4k+4 / 4k+9, Ua1 4k+5 / 4k+10.
//   WAR: a slot is restaged >= 2 phases after its last ds_read (last reads: Ua0, Ub0 phase 4k (b0 stays in registers for
//        phase 4k+3), Ub1 4k+1, Ua1 4k+2; restaged at 4k+2, 4k+3, 4k+4, 4k+5).
// Every thread issues exactly two loads per phase (tiles past the end use the out-of-range offset and write zeros into a
// slot nobody reads), so the counted wait is uniform from prologue to tail.
#include "conv_args.h"
#include <stdlib.h>
#include <type_traits>

namespace sodconv {
namespace {

constexpr int ROWB = 128;              // bytes per LDS row: 64 bf16 contraction elements
constexpr int UNIT = 128 * ROWB;       // 16 KB
constexpr int BUF = 4 * UNIT;          // one K-tile: [Ua0][Ua1][Ub0][Ub1]
constexpr int LDS_BYTES = 2 * BUF;     // 128 KB
constexpr int E16_PITCH = 256;         // bytes per pixel row of the bf16 epilogue tile (128 channels, no padding: 8 waves x 64 rows = exactly the ring's 128 KB -
                                       // a workgroup that claims more than that loses its co-residents on the CU); 8-byte units XOR-swizzled with 2 * (row & 15)

template <int SA, int SB, int KS>
__device__ __forceinline__ void mma_half(f32x4_t (&acc)[8][4], const bf16x8_t (&af)[4][2], const bf16x8_t (&bf)[2][2]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      acc[SA * 4 + i][SB * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][KS], bf[j][KS], acc[SA * 4 + i][SB * 2 + j], 0, 0, 0);
}

// One phase after its ds_reads: retire the unit issued four phases ago, meet the other wave row, then 16 MFMAs with this phase's
// unit (two LDS-DMA pieces) issued inside the cluster, where the MFMA pipe hides their issue cost (measured on the FCOS head:
// +2 % over issuing them in the load segment; s_setprio around the cluster +5 %).
// (Round 6 measured a deeper schedule - every unit re-requested one phase after its last read, four units in flight behind vmcnt(8) -
// on the tower shape: forward 352.1 vs 352.8 us, data gradient 355.5 vs 356.8 us per launch, results bit-identical: no gain, not kept.
// The same change is worth 2.7 % on conv_wgrad256.hip, which stages twice the bytes per MFMA.)
// The scheduling barriers pin the phase's staging arithmetic (tap / channel-chunk division, validity selects: ~25 scalar and vector
// instructions) BEHIND the first eight MFMAs: hipcc otherwise hoists it between the lgkmcnt wait and the first MFMA, where the matrix pipe
// waits for it.  Results bit-identical; stand-alone no difference (380 us either way on the tower shape), in the step +0.35 %
// (664.7 -> 667.0 img/s, three alternating pairs, round 6).
#define SOD256_SB __builtin_amdgcn_sched_barrier(0);
#define SOD256_PHASE(SA, SB, BFR, STAGE)                                  \
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                        \
  __builtin_amdgcn_s_barrier();                                           \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      \
  SOD256_SB                                                               \
  __builtin_amdgcn_s_setprio(1);                                          \
  mma_half<SA, SB, 0>(acc, af, BFR);                                      \
  SOD256_SB                                                               \
  STAGE;                                                                  \
  SOD256_SB                                                               \
  mma_half<SA, SB, 1>(acc, af, BFR);                                      \
  __builtin_amdgcn_s_setprio(0);                                          \
  __builtin_amdgcn_s_barrier();

template <int MODE, bool OUT_F32>
__global__ __launch_bounds__(512, 2) void conv_igemm256_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  // (One workgroup per tile.  A persistent variant - one workgroup per CU looping over tiles - measured equal, 344.3 vs 345.6 us per
  // P3-sized launch: of the 12 us a tower-conv tile spends outside its 54-us K loop [s_memtime stamps: address set-up + first LDS-DMA
  // requests 2.4 us, waiting for them 0.7 us, epilogue 6.9 us] none is workgroup dispatch.  tools/bench_tower_tile.py.)
  uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
  if (a.flags & F_REVERSE) bid = gridDim.x - 1 - bid;
  const int qt = bid % a.nq_tiles;
  int pt = bid / a.nq_tiles;
  int lv = 0;
#pragma unroll
  for (int i = 1; i < MAXLEV; ++i)
    if (i < a.nlev && pt >= a.lev[i].tile0) lv = i;
  const LevelGeo& g = a.lev[lv];
  pt -= g.tile0;
  const int q0 = qt * 256, p0 = pt * 256;
  const int gP = g.P, gHs = g.Hs, gWs = g.Ws;
  const int T = a.T;

  auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.w_bytes, 0x00020000);
  auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.src), 0, g.src_bytes, 0x00020000);

  // ---- staging geometry: a wave instruction covers 8 LDS rows x 128 B; thread -> (row, 16-B slot); unit row L = (j*8+wave)*8+srow
  const int srow = lane >> 3, spos = lane & 7;
  const int sswz = (lane >> 4) | ((wave & 1) << 2);          // (L >> 1) & 7
  const int schunk = spos ^ sswz;
  uint32_t wbase[2][2], rowbase[2][2], tapmask[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int L = (j * 8 + wave) * 8 + srow;
      const int q = q0 + (L >> 6) * 128 + u * 64 + (L & 63);
      wbase[u][j] = (q < a.Nout) ? ((uint32_t)q * (uint32_t)a.Kred + (uint32_t)schunk * 8u) * 2u : SOD_OOB;
    }
  // The weight units of K-tile 0 need nothing but wbase: they are requested BEFORE the pixel geometry below is worked out (two divisions
  // and the tap masks for four rows per thread: ~2 us with nothing else on the CU), so their latency runs beside that arithmetic.
  const uint32_t rs_mul = a.div_rs.mul, rs_shr = a.div_rs.shr, rs_d = a.div_rs.d;
  const int aCred = a.Cred;
  const bool pitched = a.Cpitch < a.Cred;      // wave-uniform
  auto stage_a = [&](int u, int kt) {     // weight rows of sub-block u (a0 / a1) of K-tile kt
    char* dst = smem + (kt & 1) * BUF + u * UNIT + wave * 1024;
    const bool live = kt < T;
    const uint32_t kk = live ? (uint32_t)kt : 0u;
    // K-tile kk = (channel chunk cc, tap): weights [Nout][tap][C] -> byte offset (tap*C + cc*64)*2
    const uint32_t cc = (__umulhi(kk, rs_mul) + kk) >> rs_shr;
    const uint32_t woff = ((kk - cc * rs_d) * (uint32_t)aCred + cc * 64u) * 2u;
    const uint32_t dead = live ? 0u : SOD_OOB;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t voff = (wbase[u][j] + woff) | dead;     // invalid rows: OOB + woff stays out of range
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, SOD_LDS(dst + j * 8192), 16, voff, 0, 0, 0);
    }
  };
  stage_a(0, 0); stage_a(1, 0);
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int L = (j * 8 + wave) * 8 + srow;
      const uint32_t p = (uint32_t)(p0 + (L >> 5) * 64 + u * 32 + (L & 31));
      uint32_t m = 0, rb = 0;
      if (p < (uint32_t)gP) {
        const uint32_t n = fd_div(p, g.div_hw);
        const uint32_t rem = p - n * g.div_hw.d;
        const uint32_t ph = fd_div(rem, g.div_w);
        const uint32_t pw = rem - ph * g.div_w.d;
        int xh, xw;
        if (MODE == MODE_FWD) { xh = (int)ph * a.stride - a.pad; xw = (int)pw * a.stride - a.pad; }
        else                  { xh = (int)ph + a.pad;            xw = (int)pw + a.pad; }
        rb = (n * (uint32_t)g.src_img_stride + ((uint32_t)xh * (uint32_t)gWs + (uint32_t)xw) * (uint32_t)a.Cpitch + (uint32_t)schunk * 8u) * 2u;
        // tap (r, s) is valid iff its row AND its column lie inside the source: R + S comparisons and an outer product of the two bit
        // rows instead of R * S (this runs once per tile, with nothing else on the CU to hide it)
        uint32_t colbits = 0;
        for (int s2 = 0; s2 < a.S; ++s2) {
          const int w = (MODE == MODE_FWD) ? xw + s2 * a.dil : xw - s2 * a.dil;
          colbits |= (uint32_t)((unsigned)w < (unsigned)gWs) << s2;
        }
        for (int r = 0; r < a.R; ++r) {
          const int h = (MODE == MODE_FWD) ? xh + r * a.dil : xh - r * a.dil;
          if ((unsigned)h < (unsigned)gHs) m |= colbits << (r * a.S);
        }
      }
      rowbase[u][j] = rb; tapmask[u][j] = m;
    }
  const int tap_sign = (MODE == MODE_FWD) ? 1 : -1;

  // Loop-invariant scalars in registers (no kernarg reloads, i.e. no lgkmcnt waits, inside the K loop); branch-free staging:
  // a dead K-tile (kt >= T) ORs the out-of-range bit into the weight offset and selects tap 31, whose mask bit is never set.
  const uint32_t s_mul = a.div_s.mul, s_shr = a.div_s.shr;
  const int aS = a.S, row_step = a.dil * gWs * a.Cpitch * 2 * tap_sign, col_step = a.dil * a.Cpitch * 2 * tap_sign;      // Cpitch: source channels per pixel (>= the contraction's own width only in sod_conv2d_dgrad_ml_kpitch)

  auto stage_b = [&](int u, int kt) {     // pixel rows of sub-block u (b0 / b1) of K-tile kt
    char* dst = smem + (kt & 1) * BUF + (2 + u) * UNIT + wave * 1024;
    const bool live = kt < T;
    const uint32_t kk = live ? (uint32_t)kt : 0u;
    const uint32_t cc = (__umulhi(kk, rs_mul) + kk) >> rs_shr;
    const uint32_t tap = kk - cc * rs_d;
    const int c0 = (int)cc * 128;
    const int r = (int)((__umulhi(tap, s_mul) + tap) >> s_shr);
    const int s2 = (int)tap - r * aS;
    const uint32_t tapoff = (uint32_t)(r * row_step + s2 * col_step + c0);
    const uint32_t tbit = live ? tap : 31u;
    // a source pitch below the contraction width (sod_conv2d_dgrad_ml_kpitch): chunks past the pixel's last channel are zero fill
    const bool cin = !pitched || ((int)cc * 64 + schunk * 8 < a.Cpitch);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t voff = (((tapmask[u][j] >> tbit) & 1u) && cin) ? rowbase[u][j] + tapoff : SOD_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, SOD_LDS(dst + j * 8192), 16, voff, 0, 0, 0);
    }
  };

  // ---- fragment read offsets (inside a unit)
  const int fr = lane & 15, fg = lane >> 4;
  uint32_t aoff[4], boff[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int L = wr * 64 + i * 16 + fr;
    aoff[i] = L * ROWB + ((fg ^ ((L >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int L = wc * 32 + j * 16 + fr;
    boff[j] = 2 * UNIT + L * ROWB + ((fg ^ ((L >> 1) & 7)) << 4);
  }

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t af[4][2], bf0[2][2], bf1[2][2];

  auto read_a = [&](const char* buf, int s) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[i][ks] = *reinterpret_cast<const bf16x8_t*>(buf + s * UNIT + (aoff[i] ^ (ks << 6)));
  };
  auto read_b = [&](const char* buf, int s, bf16x8_t (&bfr)[2][2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bfr[j][ks] = *reinterpret_cast<const bf16x8_t*>(buf + s * UNIT + (boff[j] ^ (ks << 6)));
  };

  // ---- prologue: K-tile 0 complete, first two units of K-tile 1
  stage_b(0, 0); stage_b(1, 0);          // (both weight units of K-tile 0 are on their way since the top of the tile)
  stage_a(0, 1); stage_b(0, 1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();       // wave row 1 runs one barrier behind wave row 0

  for (int k = 0; k < T; ++k) {
    const char* cur = smem + (k & 1) * BUF;
    // phase 0: quadrant (a0, b0)
    read_b(cur, 0, bf0);
    read_a(cur, 0);
    SOD256_PHASE(0, 0, bf0, stage_b(1, k + 1))
    // phase 1: quadrant (a0, b1)
    read_b(cur, 1, bf1);
    SOD256_PHASE(0, 1, bf1, stage_a(1, k + 1))
    // phase 2: quadrant (a1, b1)
    read_a(cur, 1);
    SOD256_PHASE(1, 1, bf1, stage_a(0, k + 2))
    // phase 3: quadrant (a1, b0), b0 still in registers
    SOD256_PHASE(1, 0, bf0, stage_b(0, k + 2))
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                    // all LDS-DMA (incl. the dummy tail units) landed; all fragment reads done

  // ---- epilogue: as conv_igemm.hip - the wave's accumulators go through LDS (fp32, [pixel][q]) so that every lane owns 16
  // contiguous output bytes of one pixel; residual / mask loads and the stores are 256-512-B runs.  (Storing straight from the
  // accumulator layout instead - 4 consecutive channels = 8 bytes per lane, no LDS passes, twice the store instructions - measured
  // SLOWER: 352.5 vs 341.2 us per P3-sized launch, tools/bench_tower_tile.py; the 32-byte runs per pixel cost more in the memory
  // pipeline than the 64 LDS passes per lane they save.)
  const int Nout = a.Nout;
  constexpr int QW = 128;
  constexpr int EROWB = QW * 4 + 16;
  constexpr int EPL = OUT_F32 ? 4 : 8;
  constexpr int LPR = QW / EPL;                    // lanes per pixel row (16 / 32)
  constexpr int ERPP = 64 / LPR;                   // pixel rows per pass (4 / 2)
  char* wl = smem + wave * (16 * EROWB);
  const int erow = lane / LPR, eq = (lane % LPR) * EPL;
  const int q = q0 + wr * QW + eq;
  constexpr int NP = 16 / ERPP;                    // passes per 16-pixel fragment column
  using RV = typename std::conditional<EPL == 8, bf16x8_t, bf16x4_t>::type;
  const bool qok = q < Nout;
  float bv[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) bv[e] = 0.f;
  if ((a.flags & F_BIAS) && qok) {
#pragma unroll
    for (int e = 0; e < EPL; e += 4) {
      const f32x4_t b = *reinterpret_cast<const f32x4_t*>(a.bias + q + e);
      bv[e] = b[0]; bv[e + 1] = b[1]; bv[e + 2] = b[2]; bv[e + 3] = b[3];
    }
  }
  GnAcc ga{0.f, 0.f, -1};
  // ---- bf16 whole-tile path (round 4): without a residual operand, bias and ReLU commute with the layout change, so they are applied in
  // the ACCUMULATOR layout and the wave's whole 64-pixel x 128-channel tile goes through LDS once as bf16 (16 KB per wave: 32 8-byte writes,
  // one wait, 16 16-byte reads, one wait) instead of four fp32 passes of 16 pixel rows with a write -> wait -> read -> wait chain each: the
  // epilogue of a tile is a latency chain on two waves per SIMD, not a bandwidth problem.  Same values bit for bit (one rounding either way).
  if constexpr (!OUT_F32) {
    if (!(a.flags & (F_RES | F_RES_UP2))) {
      char* w16 = smem + wave * (64 * E16_PITCH);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f32x4_t b4 = {0.f, 0.f, 0.f, 0.f};
        const int qa = q0 + wr * QW + i * 16 + fg * 4;
        if ((a.flags & F_BIAS) && qa < Nout) b4 = *reinterpret_cast<const f32x4_t*>(a.bias + qa);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          f32x4_t v = acc[i][jj] + b4;
          if (a.flags & F_RELU) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
          bf16x4_t o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
          // 8-byte unit i*4+fg of row jj*16+fr, swizzled: the 32 lanes of a half wave (16 rows x 2 units) hit 32 distinct units = 64 banks
          *reinterpret_cast<bf16x4_t*>(w16 + (jj * 16 + fr) * E16_PITCH + (((i * 4 + fg) ^ (2 * fr)) << 3)) = o;
        }
      }
      // (wave-private region: the compiler's lgkmcnt wait between the writes above and the reads below is the only ordering needed)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const uint32_t pfirst = (uint32_t)(p0 + wc * 64 + jj * 16 + erow);
        const uint32_t pc0 = pfirst < (uint32_t)gP ? pfirst : 0u;
        uint32_t n_run = fd_div(pc0, g.div_hw);
        uint32_t rem_run = pc0 - n_run * g.div_hw.d;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const uint32_t p = pfirst + (uint32_t)(k * ERPP);
          const bool okk = (p < (uint32_t)gP) && qok;
          const uint32_t n = n_run, rem = rem_run;
          rem_run += (uint32_t)ERPP;
          while (rem_run >= g.div_hw.d) { rem_run -= g.div_hw.d; ++n_run; }
          if (okk) {
            const size_t drow = (size_t)n * g.dst_img_stride + (size_t)rem * Nout;
            // (an even XOR keeps the two 8-byte units of a lane's 16 bytes adjacent: one ds_read_b128, the 16 lanes of a row permuted)
            bf16x8_t o = *reinterpret_cast<const bf16x8_t*>(w16 + (jj * 16 + k * ERPP + erow) * E16_PITCH + (((eq >> 3) ^ (k * ERPP + erow)) << 4));
            if (a.flags & F_MASK) {
              const bf16x8_t mv = *reinterpret_cast<const bf16x8_t*>((const __bf16*)g.mask + drow + q);
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = ((float)mv[e] > 0.f) ? o[e] : (__bf16)0.f;
            }
            sod_store16((__bf16*)g.dst + drow + q, o);
            if constexpr (MODE == MODE_FWD) { if (a.flags & F_GNSTATS) gn_acc_add(ga, (int)n, o, g.gn_sum, a.gn_G, q >> 3); }
          }
        }
      }
      if constexpr (MODE == MODE_FWD) {
        if (a.flags & F_GNSTATS)
          gn_acc_finish<LPR>(ga, (uint32_t)(p0 + wc * 64), (uint32_t)(p0 + wc * 64 + 63), (uint32_t)gP, g.div_hw, g.gn_sum, a.gn_G, q >> 3, qok, lane);
      }
      return;
    }
  }
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    // residual / mask operands of the 16 pixel rows are requested before the accumulators go through LDS (one exposed latency)
    RV resv[NP], maskv[NP];
    size_t drow[NP];
    bool ok[NP];
    int nimg[NP];
    // (image, pixel-in-image) of this lane's first row by one division, the following rows (ERPP pixels further each) by carry
    const uint32_t pfirst = (uint32_t)(p0 + wc * 64 + jj * 16 + erow);
    const uint32_t pc0 = pfirst < (uint32_t)gP ? pfirst : 0u;
    uint32_t n_run = fd_div(pc0, g.div_hw);
    uint32_t rem_run = pc0 - n_run * g.div_hw.d;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const uint32_t p = pfirst + (uint32_t)(k * ERPP);
      ok[k] = (p < (uint32_t)gP) && qok;
      drow[k] = 0;
      const uint32_t n = n_run, rem = rem_run;
      rem_run += (uint32_t)ERPP;
      while (rem_run >= g.div_hw.d) { rem_run -= g.div_hw.d; ++n_run; }      // (levels of a few pixels: more than one wrap)
      if (ok[k]) {
        nimg[k] = (int)n;
        drow[k] = (size_t)n * g.dst_img_stride + (size_t)rem * Nout;
        if (a.flags & (F_RES | F_RES_UP2)) {
          size_t res_row;
          if (a.flags & F_RES_UP2) {
            const uint32_t ph = fd_div(rem, g.div_w);
            const uint32_t pw = rem - ph * g.div_w.d;
            res_row = (size_t)n * g.res_img_stride + (size_t)((ph >> 1) * (g.Wp >> 1) + (pw >> 1)) * Nout;
          } else {
            res_row = (size_t)n * g.res_img_stride + (size_t)rem * Nout;
          }
          resv[k] = *reinterpret_cast<const RV*>((const __bf16*)g.res + res_row + q);
        }
        if (a.flags & F_MASK) maskv[k] = *reinterpret_cast<const RV*>((const __bf16*)g.mask + drow[k] + q);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
      *reinterpret_cast<f32x4_t*>(wl + fr * EROWB + (i * 16 + fg * 4) * 4) = acc[i][jj];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int row = k * ERPP + erow;
      if (ok[k]) {
        float v[EPL];
#pragma unroll
        for (int e = 0; e < EPL; e += 4) {
          const f32x4_t t4 = *reinterpret_cast<const f32x4_t*>(wl + row * EROWB + (eq + e) * 4);
          v[e] = t4[0]; v[e + 1] = t4[1]; v[e + 2] = t4[2]; v[e + 3] = t4[3];
        }
        if (a.flags & F_BIAS) {
#pragma unroll
          for (int e = 0; e < EPL; ++e) v[e] += bv[e];
        }
        if (a.flags & (F_RES | F_RES_UP2)) {
#pragma unroll
          for (int e = 0; e < EPL; ++e) v[e] += (float)resv[k][e];
        }
        if (a.flags & F_RELU) {
#pragma unroll
          for (int e = 0; e < EPL; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (a.flags & F_MASK) {
#pragma unroll
          for (int e = 0; e < EPL; ++e) v[e] = ((float)maskv[k][e] > 0.f) ? v[e] : 0.f;
        }
        if constexpr (OUT_F32) {
          sod_store16((float*)g.dst + drow[k] + q, f32x4_t{v[0], v[1], v[2], v[3]});
        } else {
          bf16x8_t o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
          sod_store16((__bf16*)g.dst + drow[k] + q, o);
          if constexpr (MODE == MODE_FWD) { if (a.flags & F_GNSTATS) gn_acc_add(ga, nimg[k], o, g.gn_sum, a.gn_G, q >> 3); }
        }
      }
    }
  }
  if constexpr (!OUT_F32 && MODE == MODE_FWD) {
    if (a.flags & F_GNSTATS)
      gn_acc_finish<LPR>(ga, (uint32_t)(p0 + wc * 64), (uint32_t)(p0 + wc * 64 + 63), (uint32_t)gP, g.div_hw, g.gn_sum, a.gn_G, q >> 3, qok, lane);
  }
}

template <int MODE, bool OUT_F32>
int launch256(const ConvArgs& a0, int max_pt_tiles, hipStream_t st) {
  ConvArgs a = a0;
  a.T = a.Kred / 64;
  a.div_cpt = make_fastdiv((uint32_t)(a.Cred / 64));
  a.div_rs = make_fastdiv((uint32_t)(a.R * a.S));
  a.tap_inner = 1;      // this kernel always walks K as (channel chunk outer, tap inner), see conv_args.h
  a.nq_tiles = (a.Nout + 255) / 256;
  int tiles = 0;
  for (int l = 0; l < a.nlev; ++l) {
    a.lev[l].tile0 = tiles;
    tiles += (a.lev[l].P + 255) / 256;
  }
  a.np_tiles = (max_pt_tiles > 0 && max_pt_tiles < tiles) ? max_pt_tiles : tiles;
  auto kern = conv_igemm256_kernel<MODE, OUT_F32>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  SOD_LAUNCH(kern, dim3(a.nq_tiles * a.np_tiles), dim3(512), LDS_BYTES, st, a);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

}  // namespace

bool conv256_supported(const ConvArgs& a, int mode) {
  if ((a.Cred & 63) || (a.Nout & 7) || a.R * a.S > 31 || a.cwin) return false;   // tap-validity masks are 32-bit, bit 31 = "dead tile"
  if (mode == MODE_DGRAD && a.stride != 1) return false;
  return true;
}

int launch_conv256(const ConvArgs& a, int mode, bool out_f32, int max_pt_tiles, hipStream_t st) {
  if (!conv256_supported(a, mode)) return SOD_EARG;
  if (mode == MODE_FWD) return out_f32 ? launch256<MODE_FWD, true>(a, max_pt_tiles, st) : launch256<MODE_FWD, false>(a, max_pt_tiles, st);
  return out_f32 ? launch256<MODE_DGRAD, true>(a, max_pt_tiles, st) : launch256<MODE_DGRAD, false>(a, max_pt_tiles, st);
}

}  // namespace sodconv
