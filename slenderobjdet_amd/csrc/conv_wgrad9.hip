// Weight gradient of a 3x3 / stride 1 / pad 1 convolution with the NINE TAPS IN ONE WORKGROUP (gfx950), round 6.
//
//   dW[q][tap][c] += sum_p dY[p][q] * X[p shifted by tap][c]      (the d2 ResNet / FPN 3x3 convolutions and the FCOSHead towers:
//   slender_det/modeling/backbone/fpn.py:94-115, slender_det/modeling/meta_arch/fcos/fcosv2.py:277-381; replaces ATen's conv backward-weight)
//
// conv_wgrad256.hip gives every tap its own workgroup: the nine workgroups of a pixel range stage the same dY rows nine times and nine
// shifted copies of the X rows (64 KB of LDS-DMA per 64-pixel K-tile and workgroup), and its stamps put a quarter of the loop into
// lock-step waits for those rows.  Here one workgroup of 8 waves owns 128(q) x 64(c) x 9 taps (36 MFMA 16x16x32 accumulators per wave:
// wave (wr, wc) = (wave >> 2, wave & 3) -> q in wr * 64 + [0, 64), c in wc * 16 + [0, 16), all nine taps) and stages per K-tile ONE
// [64][128 q] tile of dY (16 KB) and 64 NEW rows [64 c] of X (8 KB): 24 KB instead of 64 KB for 1.125 x the MACs, three K-tiles in flight.
//
// The shift is an LDS row offset.  Both operands are enumerated in a PADDED pixel order per level: every image gets a zero row in front
// and every row a zero pixel behind it, position j = n (H+1)(W+1) + (h+1)(W+1) + w, so that the neighbour (h + dh, w + dw) of a pixel is
// position j + dh (W+1) + dw and every out-of-image neighbour IS one of the zero positions (left of column 0 = the pad behind the previous
// row, below the last row = the zero row of the next image; positions outside [0, N (H+1)(W+1)) are zero too).  Pad positions are staged
// as zeros (out-of-range buffer offsets) in BOTH operands, so they add nothing: no border masks anywhere, and a tap's B fragment is the
// same transposed LDS read as the centre tap's at a different row.  Cost: (H+1)(W+1) / (H W) more K-tiles (2.4 % on the FCOS head).
//
// LDS: X ring [704 rows][64 c] (row = X-stream index u mod 640; rows 640..703 mirror rows 0..63 so that a 64-row window never wraps) at
// offset 0, dY ring of four [64][128 q] tiles behind it: 152 KB, one workgroup per CU.  32-B chunks are XOR-swizzled on the SOURCE side of
// the LDS-DMA (X: chunk ^ ((row >> 1) & 3), dY: chunk ^ (row & 7)); the K index of an MFMA operand is PERMUTED the same way in both
// operands (k-group g of a 32-pixel step = rows 4g..4g+3 and 16+4g..16+4g+3) so that a half wave of ds_read_b64_tr_b16 takes 8 consecutive
// rows: conflict-free at every shift.  Rows +16 / +32 / +48 are instruction offsets (they do not change the swizzle bits).
//
// Split over pixels: grid = tiles x nz, one workgroup per CU; every block stores its 128 x 64 x 9 fp32 partial tile as a slab in fragment
// order and wgrad9_reduce_kernel sums the nz slabs of a tile in fixed order into dW (x folded FrozenBN scale): no atomics, bit-identical
// from run to run.  A block whose K-tile range crosses a level boundary restarts its rings per level.
#include "conv_args.h"
#include <stdlib.h>

namespace sodconv {
namespace {

constexpr int G9_D = 3;                           // K-tiles in flight
constexpr int G9_XROWB = 128, G9_YROWB = 256;
constexpr int G9_NXC = 10, G9_RX = 64 * G9_NXC;   // X ring: ten 64-row chunks
constexpr int G9_XBYTES = (G9_RX + 64) * G9_XROWB;          // + mirror of chunk slot 0: 90 112 B
constexpr int G9_YTILE = 64 * G9_YROWB;           // 16 KB
constexpr int G9_LDS = G9_XBYTES + 4 * G9_YTILE;  // 155 648 B (+ 1 KB offset table behind it)
constexpr int G9_SLAB = 128 * 64 * 9;             // floats per partial tile

// SOD_W9_ABL (measurement builds only, tools/bench_wgrad9_abl.py): 1 = every LDS-DMA request of the loop out of range (zero fill, no memory
// traffic), 2 = no fragment reads, 4 = no MFMAs, 8 = no barrier in the loop, 16 = no vmcnt wait in the loop (8 / 16: wrong results, timing only),
// 32 = no s_setprio.  Never defined in the shipped library.
#ifndef SOD_W9_ABL
#define SOD_W9_ABL 0
#endif

template <int OFF_LO, int OFF_HI>
__device__ __forceinline__ bf16x8_t tr_read2(uint32_t addr) {
  s16x4_t lo, hi;
#if SOD_W9_ABL & 2
  asm volatile("; no read %0 %1 %2" : "=&v"(lo), "=&v"(hi) : "v"(addr));
#else
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
               : "=&v"(lo), "=&v"(hi) : "v"(addr), "n"(OFF_LO), "n"(OFF_HI));
#endif
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

#if SOD_W9_ABL & 4
#define G9_MFMA(ACC, A, B) asm volatile("; no mfma %0 %1 %2" : "+v"(ACC) : "v"(A), "v"(B))
#else
#define G9_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, ACC, 0, 0, 0)
#endif
#define G9_WAIT_LGKM(N) do { asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

}  // namespace

struct W9Level {
  const void* dy;      // (N, H, W, K) bf16 rows at dy_img_stride
  const void* x;       // (N, H, W, C) bf16
  uint32_t dy_bytes, x_bytes;
  int H, W, Wp, HWp;   // Wp = W + 1, HWp = (H + 1) (W + 1)
  int Npad;            // N * HWp padded positions
  int T, t0;           // K-tiles of this level (ceil(Npad / 64)), first global K-tile
  int E;               // X chunks a K-tile reaches ahead: (2 (W + 2) + 63) / 64
  int dy_img_stride, x_img_stride;
  FastDiv div_hwp, div_wp;
};

struct W9Args {
  W9Level lev[MAXLEV];
  int nlev;
  float* dw;
  const float* qscale;
  float* partial;
  int N, C, K, QT, CT;
  int Ttot, nz, t_per_split;
};

namespace {

__global__ __launch_bounds__(512, 2) void conv_wgrad9_kernel(const W9Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles = a.QT * a.CT;
  const int tile = (int)(bid % (uint32_t)tiles), z = (int)(bid / (uint32_t)tiles);
  const int ct = tile % a.CT, qt = tile / a.CT;
  const int q0 = qt * 128, c0 = ct * 64;
  const int kt_beg = z * a.t_per_split;
  int kt_end = kt_beg + a.t_per_split; if (kt_end > a.Ttot) kt_end = a.Ttot;

  // ---- staging geometry (loop-invariant per lane).  X: a wave instruction = 8 rows x 128 B; dY: 4 rows x 256 B.
  const int x_row = wave * 8 + (lane >> 3);                                  // row inside the 64-row chunk
  const int x_ps = lane & 7;                                                 // physical 16-B slot
  const int x_log = ((((x_ps >> 1) ^ ((x_row >> 1) & 3)) << 1) | (x_ps & 1));  // logical 16-B chunk (8 channels) this lane fetches
  const uint32_t x_cadd = (uint32_t)(c0 + x_log * 8) * 2u;
  const int y_row0 = wave * 4 + (lane >> 4);                                 // row inside the tile, + 32 for the second instruction
  const int y_ps = lane & 15;
  const int y_log = ((((y_ps >> 1) ^ (y_row0 & 7)) << 1) | (y_ps & 1));
  const uint32_t y_qadd = (uint32_t)(q0 + y_log * 8) * 2u;
  // K need not be a multiple of 128 (RetinaNet's class scores: 9 x 80 = 720): a lane whose 8 output channels lie beyond K requests the
  // out-of-range offset in every tile (the next bytes in memory belong to the following PIXEL); the reduce kernel skips those rows
  const bool y_live = q0 + y_log * 8 < a.K;

  // ---- fragment read geometry: lane 4q+p of a 16-lane group addresses row q, 8 bytes at p * 8 of the 32-B chunk
  const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  const int il = 4 * tg + tq;                                                // row inside the 16-row block (k permutation: see the header)
  uint32_t aoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = (uint32_t)G9_XBYTES + (uint32_t)il * G9_YROWB + (uint32_t)(((wr * 4 + i) ^ (il & 7)) << 5) + (uint32_t)tp * 8u;
  const uint32_t bxor = (uint32_t)(wc << 5) | (uint32_t)(tp << 3);

  f32x4_t acc[4][9];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[i][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  for (int lv = 0; lv < a.nlev; ++lv) {
    const W9Level& g = a.lev[lv];
    int sa = (kt_beg > g.t0 ? kt_beg : g.t0) - g.t0;
    int sb = (kt_end < g.t0 + g.T ? kt_end : g.t0 + g.T) - g.t0;
    if (sa >= sb) continue;                                                  // wave-uniform
    const int T = sb - sa, j0 = sa * 64, E = g.E;
    const int W = g.W, Wp = g.Wp, HWp = g.HWp, sh0 = g.W + 2;
    auto yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.dy), 0, g.dy_bytes, 0x00020000);
    auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.x), 0, g.x_bytes, 0x00020000);

    // padded position -> pixel (n, h, w) or "zero": two divisions per row
    auto pix = [&](int j, uint32_t& n, uint32_t& hw) -> bool {
      if ((unsigned)j >= (unsigned)g.Npad) return false;
      n = fd_div((uint32_t)j, g.div_hwp);
      const uint32_t rem = (uint32_t)j - n * (uint32_t)HWp;
      const uint32_t hp = fd_div(rem, g.div_wp);
      const uint32_t wp = rem - hp * (uint32_t)Wp;
      hw = (hp - 1u) * (uint32_t)W + wp;
      return hp != 0u && wp < (uint32_t)W;
    };
    // X chunk m of this segment's stream (u = 64 m + row; position j0 + u - (W + 2)) -> ring slot `slot` (= m mod 10, tracked by the caller)
    auto stage_x = [&](int m, int slot) {
      uint32_t n, hw, off = SOD_OOB;
      if (pix(j0 + 64 * m + x_row - sh0, n, hw)) off = (n * (uint32_t)g.x_img_stride + hw * (uint32_t)a.C) * 2u + x_cadd;
      char* dst = smem + (slot * 64 + wave * 8) * G9_XROWB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, SOD_LDS(dst), 16, off, 0, 0, 0);
      if (slot == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, SOD_LDS(dst + G9_RX * G9_XROWB), 16, off, 0, 0, 0);   // the mirror rows
    };
    auto stage_y = [&](int t) {            // dY K-tile t (positions j0 + 64 t + row), ring slot t & 3; tiles past the end are zeros
      char* dst = smem + G9_XBYTES + (t & 3) * G9_YTILE + wave * 4 * G9_YROWB;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        uint32_t n, hw, off = SOD_OOB;
        if (t < T && y_live && pix(j0 + 64 * t + k * 32 + y_row0, n, hw)) off = (n * (uint32_t)g.dy_img_stride + hw * (uint32_t)a.K) * 2u + y_qadd;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(yrsrc, SOD_LDS(dst + k * 32 * G9_YROWB), 16, off, 0, 0, 0);
      }
    };

    // tap shifts in rows of the X stream: (W + 2) + dh (W + 1) + dw, 0 .. 2 (W + 2)
    int sh[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) sh[t] = sh0 + (t / 3 - 1) * Wp + (t % 3 - 1);

    // ---- prologue: X chunks 0 .. E + D - 1, dY tiles 0 .. D - 1, issued in the order the loop's counted wait assumes
    int xslot = 0;                         // ring slot of the next X chunk to stage
    for (int m = 0; m <= E; ++m) { stage_x(m, xslot); xslot = (xslot + 1 == G9_NXC) ? 0 : xslot + 1; }
    stage_y(0);
#pragma unroll
    for (int d = 1; d < G9_D; ++d) {
      stage_x(E + d, xslot); xslot = (xslot + 1 == G9_NXC) ? 0 : xslot + 1;
      stage_y(d);
    }
    // The loop is bound by instruction ISSUE, not by the matrix pipe alone: an MFMA 16x16x32 leaves the SIMD two vector-issue slots, and
    // the two waves of a SIMD share them - 288 slots per K-tile for 104 transposed reads and everything else.  So what a K-tile needs besides
    // its MFMAs and reads is (a) kept small and (b) placed behind the MFMAs of the 18 steps, one tile AHEAD:
    //   * the nine B addresses of the next tile: steps 0 - 8, five instructions each;
    //   * the source offsets of the three LDS-DMA requests of the next tile (two divisions per staged row): ONE wave per tile (wave t mod 8)
    //     works them out for all 64 + 64 rows, a lane per row, and leaves them in a 1-KB table in LDS (steps 9 - 14); after the next
    //     tile's barrier every lane picks up its three rows' offsets and adds its chunk.
    const uint32_t hwp_mul = g.div_hwp.mul, hwp_shr = g.div_hwp.shr, wp_mul = g.div_wp.mul, wp_shr = g.div_wp.shr;
    const uint32_t Npad = (uint32_t)g.Npad, xis = (uint32_t)g.x_img_stride, yis = (uint32_t)g.dy_img_stride, aC = (uint32_t)a.C, aK = (uint32_t)a.K;
#define G9_PIX1(J, N_, REM, INR) { const uint32_t j_ = (uint32_t)(J); INR = j_ < Npad; N_ = (__umulhi(j_, hwp_mul) + j_) >> hwp_shr; REM = j_ - N_ * (uint32_t)HWp; }
#define G9_PIX2(REM, HW, OK) { const uint32_t hp_ = (__umulhi(REM, wp_mul) + REM) >> wp_shr; const uint32_t wq_ = REM - hp_ * (uint32_t)Wp; \
                               HW = (hp_ - 1u) * (uint32_t)W + wq_; OK = OK && hp_ != 0u && wq_ < (uint32_t)W; }
#define G9_PIX3(TAB, N_, HW, OK, IS, CH) { const uint32_t v_ = OK ? (N_ * (IS) + HW * (CH)) * 2u : SOD_OOB;                                \
                                           asm volatile("ds_write_b32 %0, %1" :: "v"((uint32_t)(uintptr_t)SOD_LDS(TAB)), "v"(v_) : "memory"); }
    uint32_t* const tab = reinterpret_cast<uint32_t*>(smem + G9_LDS);      // [2][128]: X rows 0..63, dY rows 0..63 of the group to request
    uint32_t pn, prem, phw; bool pok;
    if (wave == 0) {       // the group the first iteration requests: G(D) = X chunk D + E, dY tile D
      G9_PIX1(j0 + 64 * (G9_D + E) + lane - sh0, pn, prem, pok) G9_PIX2(prem, phw, pok) G9_PIX3(tab + lane, pn, phw, pok, xis, aC)
      G9_PIX1(j0 + 64 * G9_D + lane, pn, prem, pok) pok = pok && G9_D < T; G9_PIX2(prem, phw, pok) G9_PIX3(tab + 64 + lane, pn, phw, pok, yis, aK)
    }
#define G9_BADDR(DST, XB, TAP) { int sbt_ = (XB) + sh[TAP]; if (sbt_ >= G9_RX) sbt_ -= G9_RX; const uint32_t row_ = (uint32_t)sbt_ + (uint32_t)il; \
                                 DST = ((row_ << 7) | ((row_ << 4) & 0x60u)) ^ bxor; }
    uint32_t baddr[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) G9_BADDR(baddr[tap], 0, tap)
    int xb = 0;                            // (64 t) mod 640: ring row of the stream index 64 t
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (wave 0's table entries are in LDS before it reaches the barrier)

    for (int t = 0; t < T; ++t) {
      // everything K-tile t reads was requested G9_D iterations ago: at most the 3 (D - 1) younger requests may still be in flight
#if !(SOD_W9_ABL & 16)
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
#endif
#if !(SOD_W9_ABL & 8)
      __builtin_amdgcn_s_barrier();
#endif
      const uint32_t* tb = tab + (t & 1) * 128;
      uint32_t ox, oy0, oy1;
      asm volatile("ds_read_b32 %0, %1" : "=v"(ox) : "v"((uint32_t)(uintptr_t)SOD_LDS(tb + x_row)));
      asm volatile("ds_read_b32 %0, %1" : "=v"(oy0) : "v"((uint32_t)(uintptr_t)SOD_LDS(tb + 64 + y_row0)));
      asm volatile("ds_read_b32 %0, %1 offset:128" : "=v"(oy1) : "v"((uint32_t)(uintptr_t)SOD_LDS(tb + 64 + y_row0)));
      const uint32_t ybase = (uint32_t)((t & 3) * G9_YTILE);
      const int xb_prev = xb;
      xb += 64; if (xb >= G9_RX) xb -= G9_RX;          // now the ring row of the NEXT tile's stream index
      const int jn = j0 + 64 * (t + 1);                // first position of the next tile
      const bool ylive = t + 1 + G9_D < T;
      const bool mine = (wave == ((t + 1) & 7));       // this wave fills the table of the next iteration (wave-uniform)
      uint32_t* const tn = tab + ((t + 1) & 1) * 128 + lane;
      // the next tile's B addresses are this tile's + 64 rows (a multiple of 8 rows: the swizzle bits stay), minus the ring where the
      // tap's window start wraps - a scalar per tap
      int binc[9];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        int s0_ = xb_prev + sh[tap]; if (s0_ >= G9_RX) s0_ -= G9_RX;
        binc[tap] = (s0_ + 64 >= G9_RX) ? (64 - G9_RX) * G9_XROWB : 64 * G9_XROWB;
      }

      bf16x8_t af0[4], af1[4], b0, b1, b2;
#pragma unroll
      for (int i = 0; i < 4; ++i) af0[i] = tr_read2<0, 4096>(ybase + aoff[i]);
      b0 = tr_read2<0, 2048>(baddr[0]);
      b1 = tr_read2<0, 2048>(baddr[1]);

      // one step: [request the B fragment of step s + 2] [wait for this step's] [4 MFMAs] [a piece of next tile's bookkeeping]
#define G9_MMA(AF, B, TAP, WORK)                                                                                   \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                   \
      G9_MFMA(acc[i][TAP], AF[i], B);                                                                             \
  __builtin_amdgcn_sched_barrier(0);                                                                               \
  WORK                                                                                                             \
  __builtin_amdgcn_sched_barrier(0);
      // lgkmcnt counts ds instructions (two per fragment): the numbers are the requests YOUNGER than the one needed.
#if !(SOD_W9_ABL & 32)
      __builtin_amdgcn_s_setprio(1);
#endif
      b2 = tr_read2<0, 2048>(baddr[2]);       G9_WAIT_LGKM(4);  G9_MMA(af0, b0, 0, if (mine) G9_PIX1(jn + 64 * (G9_D + E) + lane - sh0, pn, prem, pok))
      {   // X chunk t + D + E and dY tile t + D (the table reads above are older than b0: they have landed)
        char* xd = smem + (xslot * 64 + wave * 8) * G9_XROWB;
#if SOD_W9_ABL & 1
        ox = oy0 = oy1 = SOD_OOB;
#endif
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, SOD_LDS(xd), 16, ox + x_cadd, 0, 0, 0);
        if (xslot == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, SOD_LDS(xd + G9_RX * G9_XROWB), 16, ox + x_cadd, 0, 0, 0);   // the mirror rows
        xslot = (xslot + 1 == G9_NXC) ? 0 : xslot + 1;
        char* yd = smem + G9_XBYTES + ((t + G9_D) & 3) * G9_YTILE + wave * 4 * G9_YROWB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(yrsrc, SOD_LDS(yd), 16, y_live ? oy0 + y_qadd : SOD_OOB, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(yrsrc, SOD_LDS(yd + 32 * G9_YROWB), 16, y_live ? oy1 + y_qadd : SOD_OOB, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      b0 = tr_read2<0, 2048>(baddr[3]);       G9_WAIT_LGKM(4);  G9_MMA(af0, b1, 1, if (mine) G9_PIX2(prem, phw, pok))
      b1 = tr_read2<0, 2048>(baddr[4]);       G9_WAIT_LGKM(4);  G9_MMA(af0, b2, 2, if (mine) G9_PIX3(tn, pn, phw, pok, xis, aC))
      b2 = tr_read2<0, 2048>(baddr[5]);       G9_WAIT_LGKM(4);  G9_MMA(af0, b0, 3, if (mine) { G9_PIX1(jn + 64 * G9_D + lane, pn, prem, pok) pok = pok && ylive; })
      b0 = tr_read2<0, 2048>(baddr[6]);
#pragma unroll
      for (int i = 0; i < 4; ++i) af1[i] = tr_read2<8192, 8192 + 4096>(ybase + aoff[i]);
                                              G9_WAIT_LGKM(12); G9_MMA(af0, b1, 4, if (mine) G9_PIX2(prem, phw, pok))
      b1 = tr_read2<0, 2048>(baddr[7]);       G9_WAIT_LGKM(12); G9_MMA(af0, b2, 5, if (mine) G9_PIX3(tn + 64, pn, phw, pok, yis, aK))
      b2 = tr_read2<0, 2048>(baddr[8]);       G9_WAIT_LGKM(12); G9_MMA(af0, b0, 6, )
      b0 = tr_read2<4096, 6144>(baddr[0]);    G9_WAIT_LGKM(4);  G9_MMA(af0, b1, 7, )
      b1 = tr_read2<4096, 6144>(baddr[1]);    G9_WAIT_LGKM(4);  G9_MMA(af0, b2, 8, )
      b2 = tr_read2<4096, 6144>(baddr[2]);    G9_WAIT_LGKM(4);  G9_MMA(af1, b0, 0, baddr[0] += (uint32_t)binc[0];)
      b0 = tr_read2<4096, 6144>(baddr[3]);    G9_WAIT_LGKM(4);  G9_MMA(af1, b1, 1, baddr[1] += (uint32_t)binc[1];)
      b1 = tr_read2<4096, 6144>(baddr[4]);    G9_WAIT_LGKM(4);  G9_MMA(af1, b2, 2, baddr[2] += (uint32_t)binc[2];)
      b2 = tr_read2<4096, 6144>(baddr[5]);    G9_WAIT_LGKM(4);  G9_MMA(af1, b0, 3, baddr[3] += (uint32_t)binc[3];)
      b0 = tr_read2<4096, 6144>(baddr[6]);    G9_WAIT_LGKM(4);  G9_MMA(af1, b1, 4, baddr[4] += (uint32_t)binc[4];)
      b1 = tr_read2<4096, 6144>(baddr[7]);    G9_WAIT_LGKM(4);  G9_MMA(af1, b2, 5, baddr[5] += (uint32_t)binc[5];)
      b2 = tr_read2<4096, 6144>(baddr[8]);    G9_WAIT_LGKM(4);  G9_MMA(af1, b0, 6, baddr[6] += (uint32_t)binc[6];)
                                              G9_WAIT_LGKM(2);  G9_MMA(af1, b1, 7, baddr[7] += (uint32_t)binc[7];)
                                              G9_WAIT_LGKM(0);  G9_MMA(af1, b2, 8, baddr[8] += (uint32_t)binc[8];)
      __builtin_amdgcn_s_setprio(0);
#undef G9_MMA
    }
#undef G9_PIX1
#undef G9_PIX2
#undef G9_PIX3
#undef G9_BADDR
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the dead prefetches behind the segment's last K-tile
    __builtin_amdgcn_s_barrier();                          // every wave is done with the rings before the next level refills them
  }

  // ---- epilogue: the partial tile in FRAGMENT order (16 B per lane, 1 KB per wave instruction)
  float* slab = a.partial + ((size_t)z * (size_t)tiles + (size_t)tile) * G9_SLAB + (size_t)wave * (36 * 256) + (size_t)lane * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 9; ++t) *reinterpret_cast<f32x4_t*>(slab + (i * 9 + t) * 256) = acc[i][t];
}

// Sums the nz slabs of every tile in z order and adds the result into dW.  One thread = one float4 of the fragment-ordered slab.
__global__ __launch_bounds__(256) void wgrad9_reduce_kernel(const W9Args a) {
  const int tiles = a.QT * a.CT;
  const uint32_t idx = blockIdx.x * 256u + threadIdx.x;        // < tiles * 18432
  const int lane = idx & 63;
  const uint32_t fi = idx >> 6;                                // fragment index over all tiles: 288 per tile
  const int tile = (int)(fi / 288u), frag = (int)(fi % 288u);
  if (tile >= tiles) return;
  const float* src = a.partial + (size_t)tile * G9_SLAB + (size_t)frag * 256 + (size_t)lane * 4;
  const size_t zstride = (size_t)tiles * G9_SLAB;
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
  const int wave = frag / 36, i = (frag % 36) / 9, tap = frag % 9;
  const int wr = wave >> 2, wc = wave & 3, fr = lane & 15, fg = lane >> 4;
  const int ct = tile % a.CT, qt = tile / a.CT;
  const int c = ct * 64 + wc * 16 + fr;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int q = qt * 128 + wr * 64 + i * 16 + fg * 4 + e;
    if (q >= a.K) continue;                 // rows of the last q-tile beyond K (their slab entries are zeros)
    float* dst = a.dw + ((size_t)q * 9 + tap) * a.C + c;
    *dst += sum[e] * (a.qscale ? a.qscale[q] : 1.f);
  }
}

int w9_fill(const WgradArgs& a, W9Args& w) {
  w.nlev = a.nlev; w.dw = a.dw; w.qscale = a.qscale; w.N = a.N; w.C = a.C; w.K = a.K;
  w.QT = (a.K + 127) / 128; w.CT = a.C / 64;
  int t0 = 0;
  for (int l = 0; l < a.nlev; ++l) {
    const WLevel& s = a.lev[l];
    W9Level& g = w.lev[l];
    g.dy = s.dy; g.x = s.x; g.dy_bytes = s.dy_bytes; g.x_bytes = s.x_bytes;
    g.H = s.Hx; g.W = s.Wx; g.Wp = s.Wx + 1; g.HWp = (s.Hx + 1) * (s.Wx + 1);
    g.Npad = a.N * g.HWp;
    g.T = (g.Npad + 63) / 64; g.t0 = t0; t0 += g.T;
    g.E = (2 * (s.Wx + 2) + 63) / 64;
    g.dy_img_stride = s.dy_img_stride; g.x_img_stride = s.x_img_stride;
    g.div_hwp = make_fastdiv((uint32_t)g.HWp); g.div_wp = make_fastdiv((uint32_t)g.Wp);
  }
  w.Ttot = t0;
  return t0;
}

}  // namespace

// 3x3, stride 1, pad 1, no dilation, same-size output; K a multiple of 8 (>= 128), C of 64; rows short enough for the X ring (W <= 190)
bool wgrad9_supported(const WgradArgs& a) {
  if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.dil != 1 || a.diag) return false;
  if ((a.K & 7) || (a.C & 63) || a.K < 128 || a.C < 64) return false;      // K: any multiple of 8 from 128 up (the last q-tile is masked)
  long long tot = 0;
  for (int l = 0; l < a.nlev; ++l) {
    const WLevel& s = a.lev[l];
    if (s.Ho != s.Hx || s.Wo != s.Wx || s.Wx > 190) return false;
    if ((2 * (s.Wx + 2) + 63) / 64 + G9_D + 1 > G9_NXC) return false;
    const long long np = (long long)a.N * (s.Hx + 1) * (s.Wx + 1);
    if (np >= (1ll << 30)) return false;
    tot += np;
  }
  return tot >= 1 && tot < (1ll << 30);
}

static int w9_splits(const W9Args& w, int cus, int* per_out) {
  const int tiles = w.QT * w.CT;
  int nz = cus / tiles; if (nz < 1) nz = 1;
  if (nz > w.Ttot) nz = w.Ttot;
  const int per = (w.Ttot + nz - 1) / nz;
  *per_out = per;
  return (w.Ttot + per - 1) / per;
}

long long wgrad9_workspace_bytes(const WgradArgs& a, int cus) {
  W9Args w{};
  w9_fill(a, w);
  int per = 0;
  const int nz = w9_splits(w, cus, &per);
  return (long long)nz * w.QT * w.CT * G9_SLAB * (long long)sizeof(float);
}

// K-tiles every block gets: the dispatcher's measure of whether the launch amortises its prologues and slabs
int wgrad9_tiles_per_block(const WgradArgs& a, int cus) {
  W9Args w{};
  w9_fill(a, w);
  int per = 0;
  w9_splits(w, cus, &per);
  return per;
}

int launch_wgrad9(const WgradArgs& a, int cus, float* ws, long long ws_bytes, hipStream_t st) {
  if (!wgrad9_supported(a)) return SOD_EARG;
  W9Args w{};
  w9_fill(a, w);
  int per = 0;
  w.nz = w9_splits(w, cus, &per);
  w.t_per_split = per;
  const int tiles = w.QT * w.CT;
  const long long need = (long long)w.nz * tiles * G9_SLAB * (long long)sizeof(float);
  if (!ws || need > ws_bytes) return SOD_EARG;
  if ((long long)tiles * 18432 >= (1ll << 31)) return SOD_ESIZE;
  w.partial = ws;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_wgrad9_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G9_LDS + 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  SOD_LAUNCH(conv_wgrad9_kernel, dim3(w.nz * tiles), dim3(512), G9_LDS + 1024, st, w);
  SOD_LAUNCH(wgrad9_reduce_kernel, dim3(tiles * 72), dim3(256), 0, st, w);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

}  // namespace sodconv
