// Convolution weight gradient, 128(q) x 128(c) output tile, split over pixels INSIDE the workgroup (gfx950).
//
//   dW[q][tap][c] += sum_p dY[p][q] * X[p shifted by tap][c]          (the weight gradients of the d2 ResNet / FPN convolutions under
//   slender_det/modeling/backbone/fpn.py:94-115 and of the prediction convolutions of fcosv2.py:277-381: every shape the 256x256
//   kernel of conv_wgrad256.hip does not take)
//
// The backbone shapes have few output tiles (4 ... 72 of 128 x 128) for 256 CUs, so the contraction over pixels is split and the
// partial tiles have to meet somewhere.  conv_wgrad_kernel (conv_igemm.hip) runs two 4-wave workgroups per CU and lets all 512 of them
// add their 64-KB tile into dW with float atomics: 32 MB of atomics per launch at the memory side's 1.3 TB/s, issued by every block at
// the same moment at the end of the launch, as 64-B segments (one 16x16 accumulator register = 4 rows x 64 B).  Here a workgroup is G
// groups of 4 waves; every group runs the SAME ring loop as that kernel (three or four 16-KB LDS slots per group, filled by
// buffer_load ... lds with a counted vmcnt, fragments by ds_read_b64_tr_b16) over its own pixel range of ONE output tile, and
//   * the G partial tiles are summed THROUGH LDS (the ring is dead by then): each group writes its accumulators as a [128][132] fp32
//     tile, then all 4 G waves walk the rows - half the partial bytes per CU leave the CU for G = 2;
//   * that pass also changes the layout: a wave instruction of the global epilogue covers 256 CONTIGUOUS bytes of one dW row (the shape
//     MI355X_MICROARCH.md measures at the full atomic rate), or, with a workspace (EPI = 1), the combined tile goes out as a [128][128]
//     slab with 16-B stores and wgrad_reduce_kernel (conv_igemm.hip) sums the slabs of a tile in fixed order: no atomics;
//   (Round 4 also built fragment double-buffering - the 16 transposing reads of K-step it+1 issued before the MFMAs of step it - and a
//   four-slot ring; both measured neutral, profiles/r4_wgrad_variants.txt, and left the tree in round 5.)
// One barrier per K-step for the whole workgroup; groups whose pixel range is shorter (the last split) keep staging dead (zero-fill)
// tiles so that every thread issues the same number of LDS-DMA loads per step and the counted wait stays valid.
#include "conv_args.h"
#include <stdlib.h>

namespace sodconv {
namespace {

constexpr int RKP = 32;                 // pixels per K-step
constexpr int RTILE = RKP * 256;        // one operand tile [32 px][128 ch] bf16
constexpr int RSTAGE = 2 * RTILE;       // dY tile + X tile
constexpr int RNI = RKP / 16;           // staged rows per thread and operand
constexpr int TPITCH = 132;             // floats per row of the combine tile (16-B aligned rows, fragment writes conflict-free)

template <int OFF>
__device__ __forceinline__ s16x4_t rtr_read(uint32_t addr) {
  s16x4_t r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
__device__ __forceinline__ bf16x8_t rpack8(s16x4_t lo, s16x4_t hi) {
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

struct RFrag {
  s16x4_t alo[4], ahi[4], blo[4], bhi[4];
};

// ABL (measurement builds only, -DSOD_RING_ABLATION): 1 = no MFMAs, 2 = no fragment reads, 4 = no global epilogue, 8 = no X loads,
// 16 = no dY loads - what the K loop costs without one of its parts (tools/bench_wgrad_backbone.py).
template <int G, int NSTAGE, int EPI, int ABL = 0>
__global__ __launch_bounds__(256 * G, 2) void conv_wgrad_ring_kernel(const WgradArgs a) {
  static_assert(G == 1 || G == 2, "one or two groups of four waves");
  static_assert(NSTAGE == 3 || NSTAGE == 4, "ring of three or four slots");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wv = wave & 3;
  uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
  const int RS = a.R * a.S;
  const int tap = bid % RS; bid /= RS;
  const int ct = bid % a.CT; bid /= a.CT;
  const int qt = bid % a.QT; bid /= a.QT;
  const int z = bid;
  const int r = tap / a.S, s = tap - r * a.S;
  const int q0 = qt * 128, c0 = ct * 128;
  // pixel range of this group, and the step count of the workgroup's LONGEST group (group 0: ranges are handed out in order)
  const int vbeg = (z * G + grp) * a.v_per_split;
  int vend = vbeg + a.v_per_split; if (vend > a.V) vend = a.V;
  const int nsteps = vend > vbeg ? (vend - vbeg) / RKP : 0;
  int vend0 = z * G * a.v_per_split + a.v_per_split; if (vend0 > a.V) vend0 = a.V;
  const int nmax = (vend0 - z * G * a.v_per_split) / RKP;
  char* ring = smem + grp * (NSTAGE * RSTAGE);

  // staging: one wave instruction = 4 pixel rows x 256 B; lane -> (row_in, 16-B slot)
  const int srow = lane >> 4, spos = lane & 15;
  const int sswz = srow | (((wv >> 1) & 1) << 2);         // (row&3) | ((row>>3)&1)<<2
  const int schunk = spos ^ (sswz << 1);                  // logical 16-B chunk (8 channels)
  const uint32_t qadd = (uint32_t)(q0 + schunk * 8) * 2u, cadd = (uint32_t)(c0 + schunk * 8) * 2u;
  const bool qok = (q0 + schunk * 8) < a.K, cok = (c0 + schunk * 8) < a.C;

  int cur_lv = 0;
#pragma unroll
  for (int i = 1; i < MAXLEV; ++i)
    if (i < a.nlev && vbeg >= a.lev[i].v0) cur_lv = i;
  WLevel g = a.lev[cur_lv];
  int next_v0 = (cur_lv + 1 < a.nlev) ? a.lev[cur_lv + 1].v0 : 0x7fffffff;
  auto yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.dy), 0, g.dy_bytes, 0x00020000);
  auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.x), 0, g.x_bytes, 0x00020000);

  // Row state of the INCREMENTAL path (stride 1, "same" geometry, Wo >= 32): a K-step advances every row by 32 pixels, i.e. at most one
  // column wrap and one image wrap - a handful of adds / compares instead of two divisions per row and step.
  uint32_t r_oy[RNI], r_ox[RNI];
  int r_ho[RNI], r_wo[RNI], r_p[RNI];
  bool fast = false, linear = false;
  int dh = 0, dw = 0;
  uint32_t ycorr = 0, xcorr = 0;
  auto init_rows = [&](int pbase) {
    fast = (a.stride == 1) && (g.Wo >= RKP) && (g.Ho == g.Hx) && (g.Wo == g.Wx);
    dh = r * a.dil - a.pad; dw = s * a.dil - a.pad;
    ycorr = (uint32_t)(g.dy_img_stride - g.Ho * g.Wo * a.K) * 2u;
    xcorr = (uint32_t)(g.x_img_stride - g.Hx * g.Wx * a.C) * 2u;
    // LINEAR: an un-shifted tap over dense tensors (every 1x1 convolution of the bottleneck blocks) - both byte offsets are affine in
    // the pixel index, no (row, column) to carry: 6 instead of 22 vector instructions per row and K-step
    linear = fast && dh == 0 && dw == 0 && ycorr == 0 && xcorr == 0;
    const uint32_t tapshift = (uint32_t)((dh * g.Wx + dw) * a.C * 2);
#pragma unroll
    for (int i = 0; i < RNI; ++i) {
      const int row = (i * 4 + wv) * 4 + srow;
      const int p = pbase + row;
      const uint32_t n = fd_div((uint32_t)p, g.div_hw);
      const uint32_t rem = (uint32_t)p - n * g.div_hw.d;
      const uint32_t ho = fd_div(rem, g.div_w);
      const uint32_t wo = rem - ho * g.div_w.d;
      r_p[i] = p; r_ho[i] = (int)ho; r_wo[i] = (int)wo;
      r_oy[i] = (n * (uint32_t)g.dy_img_stride + rem * (uint32_t)a.K) * 2u + qadd;
      r_ox[i] = (n * (uint32_t)g.x_img_stride + rem * (uint32_t)a.C) * 2u + cadd + tapshift;
    }
  };
  init_rows(vbeg - g.v0);
  uint32_t abl_y0[RNI], abl_x0[RNI];
#pragma unroll
  for (int i = 0; i < RNI; ++i) { abl_y0[i] = r_oy[i]; abl_x0[i] = r_ox[i]; }

  // Requests tile `it` of this group into `buf`.  EVERY call issues RLPS LDS-DMA loads per thread (dead tiles: out-of-range offsets, zero
  // fill into a slot nobody reads) - the counted vmcnt below depends on it.  Called with increasing `it`.
  auto stage = [&](int it, char* buf) {
    const bool live = it < nsteps;                      // group-uniform
    if (live) {
      const int v = vbeg + it * RKP;
      if (v >= next_v0) {
        ++cur_lv;
        g = a.lev[cur_lv];
        next_v0 = (cur_lv + 1 < a.nlev) ? a.lev[cur_lv + 1].v0 : 0x7fffffff;
        yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.dy), 0, g.dy_bytes, 0x00020000);
        xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.x), 0, g.x_bytes, 0x00020000);
        init_rows(v - g.v0);
      }
    }
    if (linear) {
      const uint32_t ystep = (uint32_t)(2 * RKP * a.K), xstep = (uint32_t)(2 * RKP * a.C);
#pragma unroll
      for (int i = 0; i < RNI; ++i) {
        const bool pv = live && r_p[i] < g.P;
        const uint32_t vy = (pv && qok) ? r_oy[i] : SOD_OOB;
        const uint32_t vx = (pv && cok) ? r_ox[i] : SOD_OOB;
        if constexpr (!(ABL & 16)) __builtin_amdgcn_raw_ptr_buffer_load_lds(yrsrc, SOD_LDS(buf + (i * 4 + wv) * 1024), 16, vy, 0, 0, 0);
        if constexpr (!(ABL & 8)) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, SOD_LDS(buf + RTILE + (i * 4 + wv) * 1024), 16, vx, 0, 0, 0);
        r_p[i] += RKP; r_oy[i] += ystep; r_ox[i] += xstep;
      }
      return;
    }
    if (!live || fast) {
      const uint32_t ystep = (uint32_t)(2 * RKP * a.K), xstep = (uint32_t)(2 * RKP * a.C);
#pragma unroll
      for (int i = 0; i < RNI; ++i) {
        const bool pv = live && r_p[i] < g.P;
        const bool tv = ((unsigned)(r_ho[i] + dh) < (unsigned)g.Hx) & ((unsigned)(r_wo[i] + dw) < (unsigned)g.Wx);
        uint32_t vy = (pv && qok) ? r_oy[i] : SOD_OOB;
        uint32_t vx = (pv && tv && cok) ? r_ox[i] : SOD_OOB;
        if constexpr (ABL & 64) {      // measurement: the row arithmetic runs, the loads keep the rows of the first step (L2-hot)
          asm volatile("; keep %0 %1" :: "v"(vy), "v"(vx));
          vy = abl_y0[i]; vx = abl_x0[i];
        }
        if constexpr (!(ABL & 16)) __builtin_amdgcn_raw_ptr_buffer_load_lds(yrsrc, SOD_LDS(buf + (i * 4 + wv) * 1024), 16, vy, 0, 0, 0);
        if constexpr (!(ABL & 8)) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, SOD_LDS(buf + RTILE + (i * 4 + wv) * 1024), 16, vx, 0, 0, 0);
        if constexpr (ABL & 32) continue;      // measurement: rows never advance (no row arithmetic, L2-hot loads)
        r_p[i] += RKP; r_oy[i] += ystep; r_ox[i] += xstep;
        int wo = r_wo[i] + RKP, ho = r_ho[i];
        if (wo >= g.Wo) { wo -= g.Wo; ho += 1; }
        if (ho >= g.Ho) { ho -= g.Ho; r_oy[i] += ycorr; r_ox[i] += xcorr; }
        r_wo[i] = wo; r_ho[i] = ho;
      }
      return;
    }
    const int pbase = vbeg + it * RKP - g.v0;
#pragma unroll
    for (int i = 0; i < RNI; ++i) {
      const int row = (i * 4 + wv) * 4 + srow;
      const int p = pbase + row;
      const bool pv = p < g.P;
      const uint32_t pc = pv ? (uint32_t)p : 0u;                    // padding rows: pixel 0, masked below
      const uint32_t n = fd_div(pc, g.div_hw);
      const uint32_t rem = pc - n * g.div_hw.d;
      const uint32_t ho = fd_div(rem, g.div_w);
      const uint32_t wo = rem - ho * g.div_w.d;
      const uint32_t oy = (n * (uint32_t)g.dy_img_stride + rem * (uint32_t)a.K) * 2u + qadd;
      const int hi = (int)ho * a.stride - a.pad + r * a.dil;
      const int wi = (int)wo * a.stride - a.pad + s * a.dil;
      const bool xv = ((unsigned)hi < (unsigned)g.Hx) & ((unsigned)wi < (unsigned)g.Wx);
      const uint32_t ox = (n * (uint32_t)g.x_img_stride + ((uint32_t)hi * (uint32_t)g.Wx + (uint32_t)wi) * (uint32_t)a.C) * 2u + cadd;
      const uint32_t vy = (pv && qok) ? oy : SOD_OOB;
      const uint32_t vx = (pv && xv && cok) ? ox : SOD_OOB;
      if constexpr (!(ABL & 16)) __builtin_amdgcn_raw_ptr_buffer_load_lds(yrsrc, SOD_LDS(buf + (i * 4 + wv) * 1024), 16, vy, 0, 0, 0);
      if constexpr (!(ABL & 8)) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, SOD_LDS(buf + RTILE + (i * 4 + wv) * 1024), 16, vx, 0, 0, 0);
    }
  };

  // transposed fragment reads: lane 4q+p of a 16-lane group addresses pixel row q, channels 4p..4p+3
  const int wq = wv >> 1, wc = wv & 1;
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = lane >> 4;
  const int tswz = tq | ((tg & 1) << 2);
  uint32_t aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    aoff[i] = (uint32_t)(8 * tg + tq) * 256u + (uint32_t)(((wq * 4 + i) ^ tswz) * 32) + tp * 8;
    boff[i] = RTILE + (uint32_t)(8 * tg + tq) * 256u + (uint32_t)(((wc * 4 + i) ^ tswz) * 32) + tp * 8;
  }
  const uint32_t ring0 = (uint32_t)(uintptr_t)SOD_LDS(ring);
  auto read_frags = [&](RFrag& f, int slot) {
    const uint32_t cb = ring0 + (uint32_t)(slot * RSTAGE);
    if constexpr (ABL & 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { f.alo[i] = f.ahi[i] = f.blo[i] = f.bhi[i] = s16x4_t{(short)slot, 1, 2, 3}; }
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f.alo[i] = rtr_read<0>(cb + aoff[i]);
      f.ahi[i] = rtr_read<1024>(cb + aoff[i]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f.blo[j] = rtr_read<0>(cb + boff[j]);
      f.bhi[j] = rtr_read<1024>(cb + boff[j]);
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto mfma16 = [&](const RFrag& f) {
    if constexpr (ABL & 1) { acc[0][0][0] += (float)f.alo[0][0] + (float)f.bhi[3][1]; return; }
    bf16x8_t af[4], bf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { af[i] = rpack8(f.alo[i], f.ahi[i]); bf[i] = rpack8(f.blo[i], f.bhi[i]); }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
  };

  constexpr int RLPS = ((ABL & 8) ? 0 : RNI) + ((ABL & 16) ? 0 : RNI);       // LDS-DMA loads per thread and stage (2 * RNI)
  constexpr int D = NSTAGE - 1;            // tiles in flight
#pragma unroll
  for (int d = 0; d < D; ++d) stage(d, ring + d * RSTAGE);

  {
    int slot = 0;
    for (int it = 0; it < nmax; ++it) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * RLPS) : "memory");
      __builtin_amdgcn_s_barrier();      // tile `it` has landed for every wave; every wave has finished reading tile it-1
      RFrag f;
      read_frags(f, slot);
      int ns = slot + D; if (ns >= NSTAGE) ns -= NSTAGE;
      stage(it + D, ring + ns * RSTAGE);                       // the slot of tile it-1
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (it < nsteps) mfma16(f);
      slot = (slot == NSTAGE - 1) ? 0 : slot + 1;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // dead tiles still in flight would land in the combine tile
  __builtin_amdgcn_s_barrier();

  // ---- combine through LDS, in two halves of 64 rows so that the G half tiles [64][TPITCH] fp32 fit into the dead ring (the workgroup
  // must not claim more LDS than the K loop needs: in the training step the data-gradient kernels of the other stream share the CU)
  const int fr = lane & 15, fg = lane >> 4;
  float* Tbase = reinterpret_cast<float*>(smem);
  float* T = Tbase + grp * (64 * TPITCH);
  if constexpr (EPI == 0) {
    if (a.qscale) {          // folded FrozenBN scale per output channel: one batch of loads, nothing below depends on a load
      float qsv[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int q = q0 + (wq * 4 + i) * 16 + fg * 4 + e;
          qsv[i][e] = q < a.K ? a.qscale[q] : 1.f;
        }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][e] *= qsv[i][e];
    }
  }
  if constexpr (ABL & 4) { if (acc[0][0][0] != 12345.678f) return; }
  constexpr int RPW = 64 / (4 * G);        // rows per wave and half
  const bool full = (q0 + 128 <= a.K) && (c0 + 128 <= a.C);
  const size_t qstride = (size_t)RS * a.C;
  float* slab = EPI == 1 ? a.partial + ((size_t)z * (size_t)(a.QT * a.CT * RS) + ((size_t)qt * a.CT + ct) * RS + tap) * (128 * 128) : nullptr;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h) __syncthreads();                 // the rows of half 0 have been read
    if (wq == h) {                          // this wave's 64 rows are the half's rows
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < 4; ++j) T[(i * 16 + fg * 4 + e) * TPITCH + (wc * 4 + j) * 16 + fr] = acc[i][j][e];
    }
    __syncthreads();
    if constexpr (EPI == 0) {
      // 256 contiguous bytes of one dW row per wave instruction
      float* dbase = a.dw + ((size_t)(q0 + h * 64 + wave * RPW) * RS + tap) * a.C + c0 + lane;
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int row = wave * RPW + rr;
        float v0 = Tbase[row * TPITCH + lane], v1 = Tbase[row * TPITCH + 64 + lane];
        if constexpr (G == 2) { v0 += Tbase[64 * TPITCH + row * TPITCH + lane]; v1 += Tbase[64 * TPITCH + row * TPITCH + 64 + lane]; }
        if (full) {
          atomicAdd(dbase + rr * qstride, v0);
          atomicAdd(dbase + rr * qstride + 64, v1);
        } else if (q0 + h * 64 + row < a.K) {
          if (c0 + lane < a.C) atomicAdd(dbase + rr * qstride, v0);
          if (c0 + 64 + lane < a.C) atomicAdd(dbase + rr * qstride + 64, v1);
        }
      }
    } else {
      // [128][128] fp32 slab of (z, tile): 16-B stores, a wave instruction = two rows of 512 B
      const int c4 = (lane & 31) * 4, rsub = lane >> 5;
#pragma unroll
      for (int rr = 0; rr < RPW; rr += 2) {
        const int row = wave * RPW + rr + rsub;
        f32x4_t v = *reinterpret_cast<const f32x4_t*>(Tbase + row * TPITCH + c4);
        if constexpr (G == 2) v += *reinterpret_cast<const f32x4_t*>(Tbase + 64 * TPITCH + row * TPITCH + c4);
        *reinterpret_cast<f32x4_t*>(slab + (h * 64 + row) * 128 + c4) = v;
      }
    }
  }
}

template <int G, int NSTAGE, int EPI, int ABL = 0>
int launch_one(const WgradArgs& a, int tiles, hipStream_t st) {
  constexpr int ring = G * NSTAGE * RSTAGE, comb = G * 64 * TPITCH * 4;
  constexpr int lds = ring > comb ? ring : comb;
  auto kern = conv_wgrad_ring_kernel<G, NSTAGE, EPI, ABL>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  SOD_LAUNCH(kern, dim3(a.nz * tiles), dim3(256 * G), lds, st, a);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

}  // namespace

// variant = G * 1000 + NSTAGE * 100 + EPI * 10 (2300: atomic epilogue, 2310: slabs).  a.nz = number of z-BLOCKS (each covers G consecutive pixel ranges of
// a.v_per_split virtual pixels); a.partial must be set for EPI = 1.
int launch_wgrad_ring(const WgradArgs& a, int variant, hipStream_t st) {
  const int tiles = a.QT * a.CT * a.R * a.S;
  switch (variant) {
#define SOD_RING_CASE(G, NS, EPI) case G * 1000 + NS * 100 + EPI * 10: return launch_one<G, NS, EPI>(a, tiles, st);
    SOD_RING_CASE(2, 3, 0) SOD_RING_CASE(2, 3, 1)
#undef SOD_RING_CASE
#ifdef SOD_RING_ABLATION
#define SOD_ABL_CASE(ABL) case 2300 + 10000 * (ABL): return launch_one<2, 3, 0, (ABL)>(a, tiles, st);
    SOD_ABL_CASE(1) SOD_ABL_CASE(3) SOD_ABL_CASE(4) SOD_ABL_CASE(5) SOD_ABL_CASE(7) SOD_ABL_CASE(8 + 7) SOD_ABL_CASE(16 + 7) SOD_ABL_CASE(8 + 4) SOD_ABL_CASE(16 + 4) SOD_ABL_CASE(24 + 4) SOD_ABL_CASE(32) SOD_ABL_CASE(64)
#undef SOD_ABL_CASE
#endif
    default: return SOD_EARG;
  }
}

}  // namespace sodconv
