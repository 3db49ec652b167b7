// Convolution weight gradient for FEW output channels: the taps are folded into the rows of the 128 x 128 output tile (gfx950).
//
//   dW[q][tap][c] += sum_p dY[p][q] * X[p shifted by tap][c]       (the prediction convolutions of the heads: FCOSHead bbox_pred + centerness,
//   8 padded channels, and cls_logits, 80 - slender_det/modeling/meta_arch/fcos/fcosv2.py:338-361; RetinaNetHead.bbox_pred, 36 -> 40 -
//   retina_rotated.py:432-437)
//
// conv_wgrad_kernel gives every (q-tile, tap) its own 128 x 128 tile: with K = 8 output channels 8 of the 128 tile rows carry data, and the
// X tile is staged and multiplied nine times.  Substituting p' = p + shift(tap),
//
//   dW[q][tap][c] = sum_p' dY[p' - shift(tap)][q] * X[p'][c],
//
// the X operand is the same for every tap, so ONE tile can hold rows rho = tap * K + q for all taps (72 rows for K = 8; 720 rows = 6 tiles
// instead of 9 for K = 80): a 16-byte staging slot (8 channels) belongs to one tap, so the tap - and with it the pixel shift and the
// border test - is a per-LANE constant of the dY staging, and the K loop is the ring loop of conv_wgrad_kernel<32, 3> unchanged
// (three 16-KB LDS slots filled by buffer_load ... lds with a counted vmcnt, fragments by ds_read_b64_tr_b16, 16 MFMAs per 32-pixel step).
// Stride 1, "same" geometry (Ho = Hx, Wo = Wx), K a multiple of 8; float atomics into dW (the deterministic mode keeps the un-folded kernel).
#include "conv_args.h"
#include <stdlib.h>
#include <algorithm>

namespace sodconv {
namespace {

constexpr int FKP = 32;                 // pixels per K-step
constexpr int FTILE = FKP * 256;        // one operand tile [32 px][128 ch] bf16
constexpr int FSTAGE = 2 * FTILE;
constexpr int FNI = FKP / 16;           // staged rows per thread and operand

template <int OFF>
__device__ __forceinline__ s16x4_t ftr_read(uint32_t addr) {
  s16x4_t r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

__global__ __launch_bounds__(256, 2) void conv_wgrad_fold_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
  const int RS = a.R * a.S;
  const int ct = bid % a.CT; bid /= a.CT;
  const int qt = bid % a.QT; bid /= a.QT;      // a.QT = tiles over the folded rows rho = tap * K + q
  const int z = bid;
  const int rho0 = qt * 128, c0 = ct * 128;
  const int rows = RS * a.K;                   // folded rows in total
  const int vbeg = z * a.v_per_split;
  int vend = vbeg + a.v_per_split; if (vend > a.V) vend = a.V;
  const int nsteps = (vend - vbeg) / FKP;

  // staging: one wave instruction = 4 pixel rows x 256 B; lane -> (row_in, 16-B slot)
  const int srow = lane >> 4, spos = lane & 15;
  const int sswz = srow | (((wave >> 1) & 1) << 2);
  const int schunk = spos ^ (sswz << 1);                  // logical 16-B chunk (8 folded rows / 8 channels)
  // this lane's folded rows rho0 + schunk*8 .. +7 lie in ONE tap (K is a multiple of 8)
  const int rho_l = rho0 + schunk * 8;
  const bool rok = rho_l < rows;
  const int tap_l = rok ? rho_l / a.K : 0;
  const int q_l = rho_l - tap_l * a.K;
  const int r_l = tap_l / a.S, s_l = tap_l - r_l * a.S;
  const int dh = r_l * a.dil - a.pad, dw = s_l * a.dil - a.pad;      // X pixel (hi, wi) pairs with output pixel (hi - dh, wi - dw)
  const uint32_t cadd = (uint32_t)(c0 + schunk * 8) * 2u;
  const bool cok = (c0 + schunk * 8) < a.C;

  int cur_lv = 0;
#pragma unroll
  for (int i = 1; i < MAXLEV; ++i)
    if (i < a.nlev && vbeg >= a.lev[i].v0) cur_lv = i;
  WLevel g = a.lev[cur_lv];
  int next_v0 = (cur_lv + 1 < a.nlev) ? a.lev[cur_lv + 1].v0 : 0x7fffffff;
  auto yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.dy), 0, g.dy_bytes, 0x00020000);
  auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.x), 0, g.x_bytes, 0x00020000);

  // Requests tile `it` into `buf`: every call issues 2 * FNI LDS-DMA loads per thread (the counted vmcnt depends on it).
  auto stage = [&](int it, char* buf) {
    const int v = vbeg + it * FKP;
    if (v >= next_v0) {
      while (v >= next_v0) { ++cur_lv; next_v0 = (cur_lv + 1 < a.nlev) ? a.lev[cur_lv + 1].v0 : 0x7fffffff; }
      g = a.lev[cur_lv];
      yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.dy), 0, g.dy_bytes, 0x00020000);
      xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.x), 0, g.x_bytes, 0x00020000);
    }
    const int pbase = v - g.v0;
#pragma unroll
    for (int i = 0; i < FNI; ++i) {
      const int p = pbase + (i * 4 + wave) * 4 + srow;              // X pixel of this row
      const bool pv = p < g.P;
      const uint32_t pc = pv ? (uint32_t)p : 0u;
      const uint32_t n = fd_div(pc, g.div_hw);
      const uint32_t rem = pc - n * g.div_hw.d;
      const uint32_t hi = fd_div(rem, g.div_w);
      const uint32_t wi = rem - hi * g.div_w.d;
      const int ho = (int)hi - dh, wo = (int)wi - dw;
      const bool yv = ((unsigned)ho < (unsigned)g.Ho) & ((unsigned)wo < (unsigned)g.Wo);
      const uint32_t oy = (n * (uint32_t)g.dy_img_stride + (uint32_t)(ho * g.Wo + wo) * (uint32_t)a.K + (uint32_t)q_l) * 2u;
      const uint32_t ox = (n * (uint32_t)g.x_img_stride + rem * (uint32_t)a.C) * 2u + cadd;
      const uint32_t vy = (pv && yv && rok) ? oy : SOD_OOB;
      const uint32_t vx = (pv && cok) ? ox : SOD_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(yrsrc, SOD_LDS(buf + (i * 4 + wave) * 1024), 16, vy, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, SOD_LDS(buf + FTILE + (i * 4 + wave) * 1024), 16, vx, 0, 0, 0);
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
    boff[i] = FTILE + (uint32_t)(8 * tg + tq) * 256u + (uint32_t)(((wc * 4 + i) ^ tswz) * 32) + tp * 8;
  }

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  constexpr int LPS = FNI * 2;
  const uint32_t lds0 = (uint32_t)(uintptr_t)SOD_LDS(smem);
  // the q-rows of this wave that lie beyond the folded rows need no MFMAs (K = 8: 72 of 128 rows -> waves with wq = 1 own rows 64..127)
  const bool wave_live = rho0 + wq * 64 < rows;
  if (nsteps > 0) stage(0, smem);
  if (nsteps > 1) stage(1, smem + FSTAGE);
  int slot = 0;
  for (int it = 0; it < nsteps; ++it) {
    if (it + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // tile `it` has landed for every wave; every wave has finished reading tile it-1
    const uint32_t cb = lds0 + (uint32_t)(slot * FSTAGE);
    s16x4_t alo[4], ahi[4], blo[4], bhi[4];
    if (wave_live) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        alo[i] = ftr_read<0>(cb + aoff[i]);
        ahi[i] = ftr_read<1024>(cb + aoff[i]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        blo[j] = ftr_read<0>(cb + boff[j]);
        bhi[j] = ftr_read<1024>(cb + boff[j]);
      }
    }
    int ns = slot + 2; if (ns >= 3) ns -= 3;
    if (it + 2 < nsteps) stage(it + 2, smem + ns * FSTAGE);   // overwrites the slot of tile it-1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);                       // nothing that uses the fragments may move above the wait
    if (wave_live) {
      bf16x8_t af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s16x8_t v = {alo[i][0], alo[i][1], alo[i][2], alo[i][3], ahi[i][0], ahi[i][1], ahi[i][2], ahi[i][3]};
        af[i] = __builtin_bit_cast(bf16x8_t, v);
        s16x8_t w = {blo[i][0], blo[i][1], blo[i][2], blo[i][3], bhi[i][0], bhi[i][1], bhi[i][2], bhi[i][3]};
        bf[i] = __builtin_bit_cast(bf16x8_t, w);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    slot = (slot == 2) ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!wave_live) return;

  // D[row = rho][col = c] -> dW[q][tap][c], rho = tap * K + q
  const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int rho = rho0 + (wq * 4 + i) * 16 + fg * 4 + e;
      if (rho >= rows) continue;
      const int tap = rho / a.K, q = rho - tap * a.K;
      const float qs = a.qscale ? a.qscale[q] : 1.f;
      float* drow = a.dw + ((size_t)q * RS + tap) * a.C;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = c0 + (wc * 4 + j) * 16 + fr;
        if (c < a.C) atomicAdd(drow + c, acc[i][j][e] * qs);
      }
    }
  }
}

}  // namespace

// Folding pays when it needs fewer 128-row tiles than one tile set per tap.
bool wgrad_fold_supported(const WgradArgs& a) {
  if (a.stride != 1 || (a.K & 7) || a.R * a.S <= 1 || a.diag || a.det) return false;
  for (int l = 0; l < a.nlev; ++l)
    if (a.lev[l].Ho != a.lev[l].Hx || a.lev[l].Wo != a.lev[l].Wx) return false;
  const int folded = (a.R * a.S * a.K + 127) / 128, plain = a.R * a.S * ((a.K + 127) / 128);
  return folded * 3 <= plain * 2;            // at least a third fewer tile rows (K <= 80 for 3x3)
}

int launch_wgrad_fold(WgradArgs& a, int cus, hipStream_t st) {
  if (!wgrad_fold_supported(a)) return SOD_EARG;
  a.QT = (a.R * a.S * a.K + 127) / 128;
  a.CT = (a.C + 127) / 128;
  const int tiles = a.QT * a.CT;
  int V = 0;
  long long Ptot = 0;
  for (int l = 0; l < a.nlev; ++l) {
    a.lev[l].v0 = V;
    V += (a.lev[l].P + 63) / 64 * 64;
    Ptot += a.lev[l].P;
  }
  a.V = V;
  // one resident wave of workgroups: three per CU (48 KB of LDS each), at least 256 pixels per workgroup
  int splits = std::max(1, 3 * cus / tiles);
  const int maxs = (int)((Ptot + 255) / 256);
  if (splits > maxs) splits = maxs;
  int vps = (V + splits - 1) / splits;
  vps = (vps + 63) / 64 * 64;
  a.v_per_split = vps;
  a.nz = (V + vps - 1) / vps;
  a.partial = nullptr;
  SOD_LAUNCH(conv_wgrad_fold_kernel, dim3(a.nz * tiles), dim3(256), 3 * FSTAGE, st, a);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

}  // namespace sodconv
