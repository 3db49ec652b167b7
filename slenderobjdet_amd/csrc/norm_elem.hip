// HBM-bound NHWC bf16 kernels around the convolutions: GroupNorm(+ReLU) fwd/bwd, ReLU backward mask,
// bias gradient, max-pool, nearest-2x upsample backward, weight preparation (FrozenBN folding + transposed copy),
// fused SGD over the flat parameter arena, image normalisation / batching.
//
// Reference call sites: nn.GroupNorm(32,C)+ReLU in FCOSHead (fcosv2.py:315-336), detectron2 FrozenBatchNorm2d /
// BasicStem max_pool2d / FPN top-down path, torch.optim.SGD built by
// slender_det/solver/build.py:8-33, preprocess_image (fcosv2.py:268-275).
#include <cstdlib>

#include "common.h"
#include "../../include/slender_hip.h"

namespace {

// ------------------------------------------------------------------ GroupNorm
// x: (N, HW, C) bf16; thread owns one 8-channel vector column (c8) and strides over pixels.
struct GnArgs {
  const __bf16* x; const __bf16* dy; const float* gamma; const float* beta;
  __bf16* y; __bf16* dx; float* stats;      // stats[N][G][2] : (sum,sumsq) then (mean,rstd)
  float* red;                               // bwd: red[N][G][2] = (sum dy*g, sum dy*g*xhat)
  float* dgamma; float* dbeta;
  float* dxsum;                             // optional: per-channel sum of dx (= bias gradient of the producing conv)
  int N, HW, C, G, cpg, relu;
  int n0;                                   // first image of this launch (blockIdx.y counts from it): image-chunked backward
  long long img_stride;                     // elements between images
  float eps;
  int pix_per_block;
  // deterministic mode (caller passed a workspace): blocks store their partial sums, indexed by block, and the gn_det_* kernels add
  // them up in a fixed order; null = float atomics
  float* part_grp;   // [gridDim.x][N][C/8][2]   per channel-vector (sum, sumsq) or (s1, s2)
  float* part_gb;    // [gridDim.x][N][2*C]      dgamma | dbeta
  float* part_dx;    // [gridDim.x][N][C]        per-channel sum of dx
};

__device__ __forceinline__ void gn_stats_body(const GnArgs& a, const int bx) {
  // Block reduction through plain LDS stores: ds_add_f32 runs at 0.33 lanes/clk/CU on gfx950 (tools/micro/lds_atomic.hip), 40x
  // below ds_write/ds_read, and these kernels ended every block with 2-18 of them per thread.
  extern __shared__ float lsum[];   // [256][2] per-thread partials
  const int n = blockIdx.y + a.n0;
  const int c8n = a.C >> 3;                 // vectors per pixel
  const int rows_per_iter = 256 / c8n;      // host guarantees c8n divides 256
  const int c8 = threadIdx.x % c8n, prow = threadIdx.x / c8n;
  const int p0 = bx * a.pix_per_block;
  int p1 = p0 + a.pix_per_block; if (p1 > a.HW) p1 = a.HW;
  float s = 0.f, ss = 0.f;
  const __bf16* base = a.x + (long long)n * a.img_stride + c8 * 8;
  for (int p = p0 + prow; p < p1; p += rows_per_iter) {
    const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(base + (long long)p * a.C);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s += f; ss += f * f; }
  }
  lsum[threadIdx.x * 2] = s;
  lsum[threadIdx.x * 2 + 1] = ss;
  __syncthreads();
  if ((int)threadIdx.x < c8n) {             // thread i sums the rows of channel vector i: one global atomic pair per vector
    float ts = 0.f, tss = 0.f;
    for (int r = 0; r < rows_per_iter; ++r) { ts += lsum[(r * c8n + threadIdx.x) * 2]; tss += lsum[(r * c8n + threadIdx.x) * 2 + 1]; }
    const int g = ((int)threadIdx.x * 8) / a.cpg;
    if (a.part_grp) {
      float* dst = a.part_grp + (((size_t)blockIdx.x * a.N + n) * c8n + threadIdx.x) * 2;
      dst[0] = ts; dst[1] = tss;
    } else {
      atomicAdd(a.stats + ((long long)n * a.G + g) * 2, ts);
      atomicAdd(a.stats + ((long long)n * a.G + g) * 2 + 1, tss);
    }
  }
}

// ---- multi-level wrappers: one launch covers all FPN levels of a shared-weight GroupNorm (levels differ in HW only) ----
constexpr int GN_MAX_LEVELS = 6;
struct GnLevel {
  const __bf16* x; const __bf16* dy; __bf16* y; __bf16* dx; float* stats; float* red;
  long long img_stride; int HW, pix_per_block, blk0, nblk; float inv_m;
};
struct GnML {
  GnLevel lev[GN_MAX_LEVELS];
  int nlev;
  const float* gamma; const float* beta; float* dgamma; float* dbeta; float* dxsum;
  int N, C, G, cpg, relu; float eps;
  float* part_grp; float* part_gb; float* part_dx;
  int n0;
  int rev;     // walk blocks / images last to first (a pass that re-reads what the previous pass just read starts where that one ended)
};

__device__ __forceinline__ int gn_bx(const GnML& m) { return m.rev ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x; }

__device__ __forceinline__ int gn_pick(const GnML& m, GnArgs& a) {
  int l = 0;
  const int bxx = gn_bx(m);
  while (l + 1 < m.nlev && bxx >= m.lev[l + 1].blk0) ++l;
  const GnLevel& L = m.lev[l];
  a.x = L.x; a.dy = L.dy; a.y = L.y; a.dx = L.dx; a.stats = L.stats; a.red = L.red; a.img_stride = L.img_stride; a.HW = L.HW;
  a.pix_per_block = L.pix_per_block;
  a.gamma = m.gamma; a.beta = m.beta; a.dgamma = m.dgamma; a.dbeta = m.dbeta; a.dxsum = m.dxsum;
  a.N = m.N; a.C = m.C; a.G = m.G; a.cpg = m.cpg; a.relu = m.relu; a.eps = m.eps;
  a.n0 = m.rev ? m.n0 + (int)gridDim.y - 1 - 2 * (int)blockIdx.y : m.n0;      // n = blockIdx.y + a.n0  ->  n0 + gridDim.y - 1 - blockIdx.y
  a.part_grp = m.part_grp; a.part_gb = m.part_gb; a.part_dx = m.part_dx;
  return l;
}

__global__ void gn_finalize_stats_kernel(const GnML m) {
  const GnLevel& L = m.lev[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < m.N * m.G) {
    const float mean = L.stats[i * 2] * L.inv_m;
    float var = L.stats[i * 2 + 1] * L.inv_m - mean * mean;
    var = fmaxf(var, 0.f);
    L.stats[i * 2] = mean;
    L.stats[i * 2 + 1] = rsqrtf(var + m.eps);
  }
}

__device__ __forceinline__ void gn_apply_body(const GnArgs& a, const int bx) {
  const int n = blockIdx.y + a.n0;
  const int c8n = a.C >> 3;
  const int rows_per_iter = 256 / c8n;
  const int c8 = threadIdx.x % c8n, prow = threadIdx.x / c8n;
  const int g = (c8 * 8) / a.cpg;
  const float mean = a.stats[((long long)n * a.G + g) * 2], rstd = a.stats[((long long)n * a.G + g) * 2 + 1];
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float gm = a.gamma[c8 * 8 + e], bt = a.beta[c8 * 8 + e];
    sc[e] = rstd * gm; sh[e] = bt - mean * rstd * gm;
  }
  const int p0 = bx * a.pix_per_block;
  int p1 = p0 + a.pix_per_block; if (p1 > a.HW) p1 = a.HW;
  const long long base = (long long)n * a.img_stride + c8 * 8;
  for (int p = p0 + prow; p < p1; p += rows_per_iter) {
    const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(a.x + base + (long long)p * a.C);
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float f = (float)v[e] * sc[e] + sh[e];
      if (a.relu) f = fmaxf(f, 0.f);
      o[e] = (__bf16)f;
    }
    sod_store16(a.y + base + (long long)p * a.C, o);
  }
}

// The two backward passes stream their operands through wave-private LDS landing buffers (LDS-DMA: `buffer_load ... lds`, 16 bytes per lane,
// GN_S stages of [x | dy] per wave) instead of through registers: the loads in flight cost no VGPRs, so the kernels stay below 64 registers
// and ONE of their waves fits on a SIMD beside two waves of the 256 x 256 convolution kernels (2 x 224 of the 512 registers; 24 of the
// 32 KB of LDS those leave) - the passes of one tower then run BESIDE the other tower's data gradient on the same CUs instead of between
// its workgroups, and with three stages (6 KB per wave) in flight they stream there at a useful rate.  Every wave issues the same number
// of vector-memory operations per trip (dead rows use the out-of-range offset: zero fill / dropped stores), so the counted waits hold.
constexpr int GN_S = 3;
constexpr int GN_RING = 4 * GN_S * 2048;      // bytes of LDS per block

template <int N>
__device__ __forceinline__ void gn_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef __attribute__((ext_vector_type(4))) unsigned int gn_u32x4_t;

__device__ __forceinline__ void gn_bwd_reduce_body(const GnArgs& a, const int bx) {
  extern __shared__ __attribute__((aligned(16))) char gsm[];   // landing buffers, then [256][18] per-thread partials: dgamma[8], dbeta[8], s1, s2
  float* lsum = reinterpret_cast<float*>(gsm);
  const int n = blockIdx.y + a.n0;
  const int c8n = a.C >> 3;
  const int rows_per_iter = 256 / c8n;
  const int c8 = threadIdx.x % c8n, prow = threadIdx.x / c8n;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int g = (c8 * 8) / a.cpg;
  const float mean = a.stats[((long long)n * a.G + g) * 2], rstd = a.stats[((long long)n * a.G + g) * 2 + 1];
  float gm[8], bt[8], dg[8], db[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { gm[e] = a.gamma[c8 * 8 + e]; bt[e] = a.beta[c8 * 8 + e]; dg[e] = 0.f; db[e] = 0.f; }
  const int p0 = bx * a.pix_per_block;
  int p1 = p0 + a.pix_per_block; if (p1 > a.HW) p1 = a.HW;
  const uint32_t img_bytes = (uint32_t)a.HW * (uint32_t)a.C * 2u;
  auto xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x + (long long)n * a.img_stride), 0, img_bytes, 0x00020000);
  auto gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.dy + (long long)n * a.img_stride), 0, img_bytes, 0x00020000);
  char* ring = gsm + wave * (GN_S * 2048);
  const uint32_t col = (uint32_t)c8 * 16u, rowb = (uint32_t)a.C * 2u;
  auto issue = [&](int it, int slot) {
    const int p = p0 + prow + it * rows_per_iter;
    const uint32_t off = p < p1 ? (uint32_t)p * rowb + col : SOD_OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, SOD_LDS(ring + slot * 2048), 16, off, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, SOD_LDS(ring + slot * 2048 + 1024), 16, off, 0, 0, 0);
  };
  // Per element only what depends on it: the mask, xhat, dgamma += d * xhat, dbeta += d.  The group sums are linear in those per-channel
  // sums - s1 = sum_e gamma[e] * dbeta[e], s2 = sum_e gamma[e] * dgamma[e] (a thread's 8 channels are one group, its pixels one image) -
  // and are formed once per thread after the loop (round 5: 15 -> 9 VALU operations per element of a pass that was not far from VALU-bound).
  const int nit = (p1 - p0 + rows_per_iter - 1) / rows_per_iter;      // block-uniform; rows are accumulated in pixel order
#pragma unroll
  for (int st = 0; st < GN_S - 1; ++st) issue(st, st);
  int slot = 0;
  for (int it = 0; it < nit; ++it) {
    int nslot = slot + GN_S - 1; if (nslot >= GN_S) nslot -= GN_S;
    issue(it + GN_S - 1, nslot);                 // (past the end: out-of-range requests, zero fill into a slot nobody reads)
    gn_wait_vm<2 * (GN_S - 1)>();
    const bf16x8_t xv = *reinterpret_cast<const bf16x8_t*>(ring + slot * 2048 + lane * 16);
    const bf16x8_t gv = *reinterpret_cast<const bf16x8_t*>(ring + slot * 2048 + 1024 + lane * 16);
    if (p0 + prow + it * rows_per_iter < p1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = ((float)xv[e] - mean) * rstd;
        float d = (float)gv[e];
        if (a.relu && !(xh * gm[e] + bt[e] > 0.f)) d = 0.f;
        dg[e] += d * xh; db[e] += d;
      }
    }
    slot = (slot == GN_S - 1) ? 0 : slot + 1;
  }
  gn_wait_vm<0>();
  __syncthreads();                               // every wave is done with its landing buffers: the space becomes lsum
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1 += db[e] * gm[e]; s2 += dg[e] * gm[e]; }
  // block-level reduction through plain LDS stores (see gn_stats_body), then one global atomic per channel per block
  float* mine = lsum + threadIdx.x * 18;
#pragma unroll
  for (int e = 0; e < 8; ++e) { mine[e] = dg[e]; mine[8 + e] = db[e]; }
  mine[16] = s1; mine[17] = s2;
  __syncthreads();
  for (int i = threadIdx.x; i < a.C; i += 256) {   // channel i = vector i/8, element i%8; rows r*c8n + vector
    float tg = 0.f, tb = 0.f;
    for (int r = 0; r < rows_per_iter; ++r) {
      const float* src = lsum + (r * c8n + (i >> 3)) * 18 + (i & 7);
      tg += src[0]; tb += src[8];
    }
    if (a.part_gb) {
      float* dst = a.part_gb + ((size_t)blockIdx.x * a.N + n) * (2 * a.C);
      dst[i] = tg; dst[a.C + i] = tb;
    } else {
      atomicAdd(a.dgamma + i, tg);                 // fp32 accumulation into the grad arena
      atomicAdd(a.dbeta + i, tb);
    }
  }
  if ((int)threadIdx.x < c8n) {
    float t1 = 0.f, t2 = 0.f;
    for (int r = 0; r < rows_per_iter; ++r) { t1 += lsum[(r * c8n + threadIdx.x) * 18 + 16]; t2 += lsum[(r * c8n + threadIdx.x) * 18 + 17]; }
    const int gg = ((int)threadIdx.x * 8) / a.cpg;
    if (a.part_grp) {
      float* dst = a.part_grp + (((size_t)blockIdx.x * a.N + n) * c8n + threadIdx.x) * 2;
      dst[0] = t1; dst[1] = t2;
    } else {
      atomicAdd(a.red + ((long long)n * a.G + gg) * 2, t1);
      atomicAdd(a.red + ((long long)n * a.G + gg) * 2 + 1, t2);
    }
  }
}


__device__ __forceinline__ void gn_bwd_apply_body(const GnArgs& a, const int bx, const float inv_m) {
  extern __shared__ __attribute__((aligned(16))) char gsm[];   // landing buffers, then [256][8] per-thread sums of dx when a.dxsum
  float* lsum = reinterpret_cast<float*>(gsm);
  const int n = blockIdx.y + a.n0;
  const int c8n = a.C >> 3;
  const int rows_per_iter = 256 / c8n;
  const int c8 = threadIdx.x % c8n, prow = threadIdx.x / c8n;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int g = (c8 * 8) / a.cpg;
  const float mean = a.stats[((long long)n * a.G + g) * 2], rstd = a.stats[((long long)n * a.G + g) * 2 + 1];
  const float m1 = a.red[((long long)n * a.G + g) * 2] * inv_m, m2 = a.red[((long long)n * a.G + g) * 2 + 1] * inv_m;
  float gm[8], bt[8], sx[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { gm[e] = a.gamma[c8 * 8 + e]; bt[e] = a.beta[c8 * 8 + e]; sx[e] = 0.f; }
  const int p0 = bx * a.pix_per_block;
  int p1 = p0 + a.pix_per_block; if (p1 > a.HW) p1 = a.HW;
  const uint32_t img_bytes = (uint32_t)a.HW * (uint32_t)a.C * 2u;
  auto xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x + (long long)n * a.img_stride), 0, img_bytes, 0x00020000);
  auto gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.dy + (long long)n * a.img_stride), 0, img_bytes, 0x00020000);
  auto orr = __builtin_amdgcn_make_buffer_rsrc(a.dx + (long long)n * a.img_stride, 0, img_bytes, 0x00020000);
  char* ring = gsm + wave * (GN_S * 2048);
  const uint32_t col = (uint32_t)c8 * 16u, rowb = (uint32_t)a.C * 2u;
  auto issue = [&](int it, int slot) {
    const int p = p0 + prow + it * rows_per_iter;
    const uint32_t off = p < p1 ? (uint32_t)p * rowb + col : SOD_OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, SOD_LDS(ring + slot * 2048), 16, off, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, SOD_LDS(ring + slot * 2048 + 1024), 16, off, 0, 0, 0);
  };
  // trip `it`: requests of trip it + 2, wait for trip it's two loads, compute, ONE store (dropped for rows past the end).  Younger than
  // trip it's loads at its wait: the loads of trips it + 1 and it + 2 (4) and the stores of trips it - 2 and it - 1 (fewer at the start).
  const int nit = (p1 - p0 + rows_per_iter - 1) / rows_per_iter;
#pragma unroll
  for (int st = 0; st < GN_S - 1; ++st) issue(st, st);
  int slot = 0;
  for (int it = 0; it < nit; ++it) {
    int nslot = slot + GN_S - 1; if (nslot >= GN_S) nslot -= GN_S;
    issue(it + GN_S - 1, nslot);
    if (it == 0) gn_wait_vm<2 * (GN_S - 1)>(); else if (it == 1) gn_wait_vm<2 * (GN_S - 1) + 1>(); else gn_wait_vm<3 * (GN_S - 1)>();
    const bf16x8_t xv = *reinterpret_cast<const bf16x8_t*>(ring + slot * 2048 + lane * 16);
    const bf16x8_t gv = *reinterpret_cast<const bf16x8_t*>(ring + slot * 2048 + 1024 + lane * 16);
    const int p = p0 + prow + it * rows_per_iter;
    const bool live = p < p1;
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = ((float)xv[e] - mean) * rstd;
      float d = (float)gv[e];
      if (a.relu && !(xh * gm[e] + bt[e] > 0.f)) d = 0.f;
      o[e] = (__bf16)(rstd * (d * gm[e] - m1 - xh * m2));
      sx[e] += live ? (float)o[e] : 0.f;       // the bias gradient is the sum of what the conv's wgrad/dgrad see (the stored bf16 values)
    }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(gn_u32x4_t, o), orr, live ? (uint32_t)p * rowb + col : SOD_OOB, 0, 0);
    slot = (slot == GN_S - 1) ? 0 : slot + 1;
  }
  gn_wait_vm<0>();
  __syncthreads();                               // every wave is done with its landing buffers: the space becomes lsum
  if (a.dxsum) {
#pragma unroll
    for (int e = 0; e < 8; ++e) lsum[threadIdx.x * 8 + e] = sx[e];
    __syncthreads();
    for (int i = threadIdx.x; i < a.C; i += 256) {
      float t = 0.f;
      for (int r = 0; r < rows_per_iter; ++r) t += lsum[(r * c8n + (i >> 3)) * 8 + (i & 7)];
      if (a.part_dx) a.part_dx[((size_t)blockIdx.x * a.N + n) * a.C + i] = t;
      else atomicAdd(a.dxsum + i, t);
    }
  }
}

// ---- deterministic second stages (one thread per output; the loops fix the summation order) ----
// out[(l, n, g)][w] = sum over the blocks of level l and the channel vectors of group g of part_grp[block][n][c8][w];
// finalize != 0 turns (sum, sumsq) into (mean, rstd) as gn_finalize_stats_kernel does.
__global__ void gn_det_group_kernel(const GnML m, int finalize) {
  const GnLevel& L = m.lev[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m.N * m.G) return;
  const int n = i / m.G, g = i - n * m.G;
  const int c8n = m.C >> 3, vpg = m.cpg >> 3;
  float s0 = 0.f, s1 = 0.f;
  for (int b = L.blk0; b < L.blk0 + L.nblk; ++b) {
    const float* src = m.part_grp + (((size_t)b * m.N + n) * c8n + (size_t)g * vpg) * 2;
    for (int v = 0; v < vpg; ++v) { s0 += src[v * 2]; s1 += src[v * 2 + 1]; }
  }
  float* dst = (finalize ? L.stats : L.red) + (size_t)i * 2;
  if (finalize) {
    const float mean = s0 * L.inv_m;
    const float var = fmaxf(s1 * L.inv_m - mean * mean, 0.f);
    dst[0] = mean; dst[1] = rsqrtf(var + m.eps);
  } else {
    dst[0] = s0; dst[1] = s1;
  }
}

// out[c] += sum_r part[r * ld + c]  for c < cols: 64 columns x 4 row lanes per block, lanes combined as (0+1)+(2+3)
__global__ __launch_bounds__(256) void col_accumulate_kernel(const float* __restrict__ part, int rows, int cols, int ld, float* __restrict__ out) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  float s = 0.f;
  if (c < cols)
    for (int r = rl; r < rows; r += 4) s += part[(size_t)r * ld + c];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < cols) out[c] += (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void gn_stats_kernel(const GnML m) {
  GnArgs a;
  const int l = gn_pick(m, a);
  gn_stats_body(a, gn_bx(m) - m.lev[l].blk0);
}
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnML m) {
  GnArgs a;
  const int l = gn_pick(m, a);
  gn_apply_body(a, gn_bx(m) - m.lev[l].blk0);
}
__global__ __launch_bounds__(256, 8) void gn_bwd_reduce_kernel(const GnML m) {
  GnArgs a;
  const int l = gn_pick(m, a);
  gn_bwd_reduce_body(a, gn_bx(m) - m.lev[l].blk0);
}
__global__ __launch_bounds__(256, 8) void gn_bwd_apply_kernel(const GnML m) {
  GnArgs a;
  const int l = gn_pick(m, a);
  gn_bwd_apply_body(a, gn_bx(m) - m.lev[l].blk0, m.lev[l].inv_m);
}

// ------------------------------------------------------------------ elementwise (8 x bf16 per lane)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const __bf16* __restrict__ dy, const __bf16* __restrict__ y,
                                                       __bf16* __restrict__ dx, long long n8) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const bf16x8_t g = reinterpret_cast<const bf16x8_t*>(dy)[i];
    const bf16x8_t v = reinterpret_cast<const bf16x8_t*>(y)[i];
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = ((float)v[e] > 0.f) ? g[e] : (__bf16)0.f;
    sod_store16(reinterpret_cast<bf16x8_t*>(dx) + i, o);
  }
}

__global__ __launch_bounds__(256) void relu_fwd_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ y, long long n8) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const bf16x8_t v = reinterpret_cast<const bf16x8_t*>(x)[i];
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = ((float)v[e] > 0.f) ? v[e] : (__bf16)0.f;
    sod_store16(reinterpret_cast<bf16x8_t*>(y) + i, o);
  }
}

__global__ __launch_bounds__(256) void add_kernel(const __bf16* __restrict__ a, const __bf16* __restrict__ b,
                                                  __bf16* __restrict__ o, long long n8) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const bf16x8_t x = reinterpret_cast<const bf16x8_t*>(a)[i];
    const bf16x8_t y = reinterpret_cast<const bf16x8_t*>(b)[i];
    bf16x8_t r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (__bf16)((float)x[e] + (float)y[e]);
    reinterpret_cast<bf16x8_t*>(o)[i] = r;
  }
}

// o[n,h,w,:] = a[n,h,w,:] + b[n,h/2,w/2,:]  (FPN top-down path when a normalisation sits between the lateral conv and the sum)
__global__ __launch_bounds__(256) void add_up2_kernel(const __bf16* __restrict__ a, const __bf16* __restrict__ b, __bf16* __restrict__ o,
                                                      int N, int H, int W, int C) {
  const int c8n = C >> 3, Hc = H >> 1, Wc = W >> 1;
  const long long total = (long long)N * H * W * c8n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c8 = (int)(i % c8n);
    long long r = i / c8n;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    const bf16x8_t x = reinterpret_cast<const bf16x8_t*>(a)[i];
    const bf16x8_t y = *reinterpret_cast<const bf16x8_t*>(b + (((long long)n * Hc + (h >> 1)) * Wc + (w >> 1)) * C + c8 * 8);
    bf16x8_t q;
#pragma unroll
    for (int e = 0; e < 8; ++e) q[e] = (__bf16)((float)x[e] + (float)y[e]);
    reinterpret_cast<bf16x8_t*>(o)[i] = q;
  }
}

// bias gradient: db[c] += sum over rows of dy[row][c]; dy rows may be strided per image
__global__ __launch_bounds__(256) void channel_sum_kernel(const __bf16* __restrict__ dy, float* __restrict__ db,
                                                          int HW, int C, long long img_stride, int pix_per_block, float* __restrict__ part,
                                                          const float* __restrict__ scale_num, const float* __restrict__ scale_den,
                                                          float den_mul, float den_min) {
  extern __shared__ float lsum[];   // [256][8] per-thread partials
  const int n = blockIdx.y;
  const int c8n = C >> 3;
  const int rows_per_iter = 256 / c8n;
  const int c8 = threadIdx.x % c8n, prow = threadIdx.x / c8n;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int p0 = blockIdx.x * pix_per_block;
  int p1 = p0 + pix_per_block; if (p1 > HW) p1 = HW;
  if (threadIdx.x < rows_per_iter * c8n) {
    const long long base = (long long)n * img_stride + c8 * 8;
    // four independent 16-byte loads in flight per thread (one per trip left the pass at 0.7-1.1 TB/s on the P2 / P3-sized tensors)
    int p = p0 + prow;
    for (; p + 3 * rows_per_iter < p1; p += 4 * rows_per_iter) {
      const bf16x8_t v0 = *reinterpret_cast<const bf16x8_t*>(dy + base + (long long)p * C);
      const bf16x8_t v1 = *reinterpret_cast<const bf16x8_t*>(dy + base + (long long)(p + rows_per_iter) * C);
      const bf16x8_t v2 = *reinterpret_cast<const bf16x8_t*>(dy + base + (long long)(p + 2 * rows_per_iter) * C);
      const bf16x8_t v3 = *reinterpret_cast<const bf16x8_t*>(dy + base + (long long)(p + 3 * rows_per_iter) * C);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += ((float)v0[e] + (float)v1[e]) + ((float)v2[e] + (float)v3[e]);
    }
    for (; p < p1; p += rows_per_iter) {
      const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(dy + base + (long long)p * C);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) lsum[threadIdx.x * 8 + e] = s[e];      // plain stores, not ds_add_f32 (see gn_stats_body)
  }
  __syncthreads();
  float sc = scale_num ? scale_num[0] : 1.f;      // sod_bias_grad_scaled: dy is an UN-scaled gradient (sod_sigmoid_focal_loss_fwd_grad)
  if (scale_den) sc /= fmaxf(scale_den[0] * den_mul, den_min);
  for (int i = threadIdx.x; i < C; i += 256) {
    float t = 0.f;
    for (int r = 0; r < rows_per_iter; ++r) t += lsum[(r * c8n + (i >> 3)) * 8 + (i & 7)];
    t *= sc;
    if (part) part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * C + i] = t;      // deterministic mode: col_accumulate_kernel adds the rows
    else atomicAdd(db + i, t);
  }
}

// channel_sum over several dense (N, HW_l, C) tensors in ONE launch (bias gradient of a conv shared by all FPN levels)
struct ChanSumML {
  const __bf16* dy[GN_MAX_LEVELS];
  int HW[GN_MAX_LEVELS], blk0[GN_MAX_LEVELS], ppb[GN_MAX_LEVELS];
  int nlev, C;
};
__global__ __launch_bounds__(256) void channel_sum_ml_kernel(const ChanSumML m, float* __restrict__ db) {
  extern __shared__ float lsum[];   // [256][8] per-thread partials
  int l = 0;
  while (l + 1 < m.nlev && (int)blockIdx.x >= m.blk0[l + 1]) ++l;
  const int n = blockIdx.y, C = m.C, HW = m.HW[l];
  const int c8n = C >> 3;
  const int rows_per_iter = 256 / c8n;
  const int c8 = threadIdx.x % c8n, prow = threadIdx.x / c8n;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int p0 = ((int)blockIdx.x - m.blk0[l]) * m.ppb[l];
  int p1 = p0 + m.ppb[l]; if (p1 > HW) p1 = HW;
  if (threadIdx.x < rows_per_iter * c8n) {
    const __bf16* __restrict__ dy = m.dy[l] + (long long)n * HW * C + c8 * 8;
    int p = p0 + prow;
    for (; p + 3 * rows_per_iter < p1; p += 4 * rows_per_iter) {      // four loads in flight per thread (channel_sum_kernel)
      const bf16x8_t v0 = *reinterpret_cast<const bf16x8_t*>(dy + (long long)p * C);
      const bf16x8_t v1 = *reinterpret_cast<const bf16x8_t*>(dy + (long long)(p + rows_per_iter) * C);
      const bf16x8_t v2 = *reinterpret_cast<const bf16x8_t*>(dy + (long long)(p + 2 * rows_per_iter) * C);
      const bf16x8_t v3 = *reinterpret_cast<const bf16x8_t*>(dy + (long long)(p + 3 * rows_per_iter) * C);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += ((float)v0[e] + (float)v1[e]) + ((float)v2[e] + (float)v3[e]);
    }
    for (; p < p1; p += rows_per_iter) {
      const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(dy + (long long)p * C);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) lsum[threadIdx.x * 8 + e] = s[e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) {
    float t = 0.f;
    for (int r = 0; r < rows_per_iter; ++r) t += lsum[(r * c8n + (i >> 3)) * 8 + (i & 7)];
    atomicAdd(db + i, t);
  }
}

// 3x3 stride-2 pad-1 max pool, NHWC bf16 (detectron2 BasicStem)
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ y,
                                                           int N, int H, int W, int C, int Ho, int Wo) {
  const int c8n = C >> 3;
  const long long total = (long long)N * Ho * Wo * c8n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c8 = (int)(i % c8n);
    long long r = i / c8n;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -3.0e38f;
    for (int dh = 0; dh < 3; ++dh) {
      const int h = ho * 2 - 1 + dh;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int dw = 0; dw < 3; ++dw) {
        const int w = wo * 2 - 1 + dw;
        if ((unsigned)w >= (unsigned)W) continue;
        const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(x + (((long long)n * H + h) * W + w) * C + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], (float)v[e]);
      }
    }
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)m[e];
    *reinterpret_cast<bf16x8_t*>(y + i * 8) = o;
  }
}

// backward of nearest-2x upsample: dprev[n,h,w,:] = sum of the 2x2 block of g
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const __bf16* __restrict__ g, __bf16* __restrict__ d,
                                                             int N, int Hc, int Wc, int C) {
  const int c8n = C >> 3;
  const long long total = (long long)N * Hc * Wc * c8n;
  const int Wf = Wc * 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c8 = (int)(i % c8n);
    long long r = i / c8n;
    const int w = (int)(r % Wc); r /= Wc;
    const int h = (int)(r % Hc);
    const int n = (int)(r / Hc);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) {
        const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(g + (((long long)n * Hc * 2 + h * 2 + dh) * Wf + w * 2 + dw) * C + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
      }
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)s[e];
    *reinterpret_cast<bf16x8_t*>(d + i * 8) = o;
  }
}

// ------------------------------------------------------------------ weights
// master fp32 KRSC -> bf16 KRSC (scaled per output channel) and bf16 CRSK (for dgrad)
__global__ __launch_bounds__(256) void weight_prep_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                          __bf16* __restrict__ w_krsc, __bf16* __restrict__ w_crsk,
                                                          int K, int RS, int C, int Cpad /* fwd row pitch in channels */) {
  const long long total = (long long)K * RS * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long long r = i / C;
    const int t = (int)(r % RS);
    const int k = (int)(r / RS);
    float v = w[i];
    if (scale) v *= scale[k];
    const __bf16 b = (__bf16)v;
    if (w_krsc) w_krsc[((long long)k * RS + t) * Cpad + c] = b;
    if (w_crsk) w_crsk[((long long)c * RS + t) * K + k] = b;
  }
}

// The same for EVERY trainable convolution of a model in one launch (61 launches of ~7 us each per FCOS step otherwise):
// entries sorted by their first global element index; a thread finds its entry by binary search.
struct PrepEntry {
  long long elem0;      // first global TILE of this entry (prefix sum of tiles: ceil(K/64) * RS * ceil(C/64))
  long long src_off;    // fp32 offset into the parameter arena
  long long krsc_off;   // bf16 offset into the KRSC compute arena
  long long crsk_off;   // bf16 offset into the CRSK compute arena
  long long scale_off;  // fp32 offset into the scale array, or -1
  int K, RS, C, Cpad;
};
// One workgroup = one 64(k) x 64(c) tile of one tap of one weight: coalesced fp32 reads, coalesced KRSC bf16 writes, transpose through
// LDS, coalesced CRSK writes (the element-wise version wrote 2-byte values with a stride of K elements: 0.42 ms per step for 32 M
// parameters; this one moves the same 256 MB at HBM speed).
__global__ __launch_bounds__(256) void weight_prep_batched_kernel(const float* __restrict__ params, const float* __restrict__ scales,
                                                                  const PrepEntry* __restrict__ tab, int n, long long total_tiles,
                                                                  __bf16* __restrict__ krsc, __bf16* __restrict__ crsk) {
  __shared__ __bf16 tile[64][66];
  for (long long tb = blockIdx.x; tb < total_tiles; tb += gridDim.x) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (tab[mid].elem0 <= tb) lo = mid; else hi = mid - 1;
    }
    const PrepEntry e = tab[lo];
    const int ct_n = (e.C + 63) >> 6;
    long long j = tb - e.elem0;
    const int ct = (int)(j % ct_n); j /= ct_n;
    const int t = (int)(j % e.RS);
    const int kt = (int)(j / e.RS);
    const int k0 = kt * 64, c0 = ct * 64;
    // load + KRSC store: thread -> (row = tid / 16 + 16 * pass, 4 consecutive c)
    const int lc = (threadIdx.x & 15) * 4, lr = threadIdx.x >> 4;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = lr + pass * 16, k = k0 + r;
      if (k < e.K) {
        const float sc = e.scale_off >= 0 ? scales[e.scale_off + k] : 1.f;
        const long long rowoff = ((long long)k * e.RS + t) * e.C + c0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = lc + q;
          __bf16 b = (__bf16)0.f;
          if (c0 + c < e.C) {
            b = (__bf16)(params[e.src_off + rowoff + c] * sc);
            krsc[e.krsc_off + ((long long)k * e.RS + t) * e.Cpad + c0 + c] = b;
          }
          tile[r][c] = b;
        }
      }
    }
    __syncthreads();
    // CRSK store: thread -> (c row = tid / 16 + 16 * pass, 4 consecutive k)
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int cr = lr + pass * 16, c = c0 + cr;
      if (c < e.C) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = k0 + lc + q;
          if (k < e.K) crsk[e.crsk_off + ((long long)c * e.RS + t) * e.K + k] = tile[lc + q][cr];
        }
      }
    }
    __syncthreads();
  }
}

// per-output-channel scaling of a weight gradient (chain rule through the folded FrozenBN scale)
__global__ __launch_bounds__(256) void scale_rows_kernel(float* __restrict__ g, const float* __restrict__ scale, int K, long long row) {
  const long long total = (long long)K * row;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) g[i] *= scale[i / row];
}

// ------------------------------------------------------------------ fused SGD over the flat arena
// torch.optim.SGD semantics: d = g + wd*p ; buf = mom*buf + d (buf = d on the first step) ; p -= lr*d'
struct SgdSeg { long long begin, end; float lr_mult, wd; };
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  const SgdSeg* __restrict__ segs, int nseg, const float* __restrict__ lr_dev,
                                                  float lr_host, float momentum, int nesterov, int first_step, float grad_scale) {
  const float base_lr = lr_dev ? lr_dev[0] : lr_host;
  const SgdSeg sg = segs[blockIdx.y];
  const float step = base_lr * sg.lr_mult;
  if (((sg.begin | sg.end) & 3) == 0) {      // arena segments are 64-float aligned: 16-B accesses
    for (long long i = sg.begin + ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < sg.end; i += (long long)gridDim.x * 1024) {
      const f32x4_t gv = *reinterpret_cast<const f32x4_t*>(g + i);
      f32x4_t pv = *reinterpret_cast<const f32x4_t*>(p + i);
      f32x4_t d = gv * grad_scale + sg.wd * pv;
      if (momentum != 0.f) {
        f32x4_t b = d;
        if (!first_step) b = momentum * *reinterpret_cast<const f32x4_t*>(m + i) + d;
        *reinterpret_cast<f32x4_t*>(m + i) = b;
        d = nesterov ? d + momentum * b : b;
      }
      pv -= step * d;
      *reinterpret_cast<f32x4_t*>(p + i) = pv;
    }
    return;
  }
  for (long long i = sg.begin + (long long)blockIdx.x * 256 + threadIdx.x; i < sg.end; i += (long long)gridDim.x * 256) {
    float d = g[i] * grad_scale + sg.wd * p[i];
    if (momentum != 0.f) {
      const float b = first_step ? d : momentum * m[i] + d;
      m[i] = b;
      d = nesterov ? d + momentum * b : b;
    }
    p[i] -= step * d;
  }
}

// ------------------------------------------------------------------ fused Adam / AdamW / Adagrad over the flat arena
// torch.optim.Adam / AdamW / Adagrad semantics (slender_det/solver/build.py:26-31), one launch over all parameters, per-segment
// learning-rate multiplier and weight decay exactly as sgd_kernel.  mode 0 = Adam (L2 decay added to the gradient), 1 = AdamW
// (decoupled: p *= 1 - lr*wd), 2 = Adagrad (L2 decay added to the gradient; `m` unused, `v` = the running sum of squares).
// bc1 / bc2 = 1 - beta^t, computed in double on the host as torch does; `clr` (Adagrad) = lr / (1 + (t-1) * lr_decay).
__global__ __launch_bounds__(256) void adaptive_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, const SgdSeg* __restrict__ segs, int mode, float lr,
                                                       float beta1, float beta2, float eps, float bc1, float bc2_sqrt, float grad_scale) {
  const SgdSeg sg = segs[blockIdx.y];
  const float step = lr * sg.lr_mult;
  for (long long i = sg.begin + (long long)blockIdx.x * 256 + threadIdx.x; i < sg.end; i += (long long)gridDim.x * 256) {
    float pv = p[i];
    float d = g[i] * grad_scale;
    if (mode == 1) pv *= 1.f - step * sg.wd;
    else d += sg.wd * pv;
    if (mode == 2) {
      const float s = v[i] + d * d;
      v[i] = s;
      pv -= step * (d / (sqrtf(s) + eps));
    } else {
      const float mi = m[i] + (d - m[i]) * (1.f - beta1);          // torch: exp_avg.lerp_(grad, 1 - beta1)
      const float vi = beta2 * v[i] + (1.f - beta2) * d * d;       // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
      m[i] = mi; v[i] = vi;
      const float denom = sqrtf(vi) / bc2_sqrt + eps;
      pv -= (step / bc1) * (mi / denom);
    }
    p[i] = pv;
  }
}

// ------------------------------------------------------------------ image batching
// uint8/float CHW image -> (x - mean)/std -> NHWC bf16 with Cpad channels, zero padded to (Hp, Wp)
template <typename T>
__global__ __launch_bounds__(256) void preprocess_kernel(const T* __restrict__ img, int C, int H, int W,
                                                         __bf16* __restrict__ out, int Hp, int Wp, int Cpad,
                                                         float m0, float m1, float m2, float s0, float s1, float s2) {
  const long long total = (long long)Hp * Wp;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int w = (int)(i % Wp), h = (int)(i / Wp);
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)0.f;
    if (h < H && w < W) {
      const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
      for (int c = 0; c < C && c < 3; ++c) o[c] = (__bf16)(((float)img[((long long)c * H + h) * W + w] - mean[c]) / stdv[c]);
    }
    *reinterpret_cast<bf16x8_t*>(out + i * Cpad) = o;   // Cpad == 8
  }
}

// The whole batch in one launch (blockIdx.y = image): 16 launches of 10 us each (2 TB/s, latency-bound) -> one.
constexpr int PRE_MAX_IMAGES = 64;
struct PreBatch {
  const void* img[PRE_MAX_IMAGES];
  int H[PRE_MAX_IMAGES], W[PRE_MAX_IMAGES];
};
template <typename T>
__global__ __launch_bounds__(256) void preprocess_batch_kernel(const PreBatch b, int C, __bf16* __restrict__ out, int Hp, int Wp,
                                                               float m0, float m1, float m2, float s0, float s1, float s2) {
  const int n = blockIdx.y;
  const T* __restrict__ img = (const T*)b.img[n];
  const int H = b.H[n], W = b.W[n];
  __bf16* __restrict__ o_n = out + (long long)n * Hp * Wp * 8;
  const int total = Hp * Wp;
  const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int h = i / Wp, w = i - h * Wp;
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)0.f;
    if (h < H && w < W) {
      for (int c = 0; c < C && c < 3; ++c) o[c] = (__bf16)(((float)img[((long long)c * H + h) * W + w] - mean[c]) / stdv[c]);
    }
    *reinterpret_cast<bf16x8_t*>(o_n + (long long)i * 8) = o;
  }
}

// ---- input pipeline on the device: ResizeShortestEdge + RandomFlip (slender_det/data/utils.py:29-50 ->
// detectron2 ResizeTransform = PIL.Image.resize(BILINEAR) on the uint8 image, HFlipTransform) fused with preprocess_image's
// normalise + pad + NHWC(8) bf16 (fcosv2.py:268-275).  PIL's bilinear resize is a separable triangle filter whose support grows with
// the down-scaling factor, evaluated in fixed point (22 fractional bits) with a ROUNDED uint8 image between the horizontal and the
// vertical pass; the per-axis tap ranges and integer coefficients are computed on the host exactly as Pillow's precompute_coeffs does
// (data/transforms.py) and the kernel repeats Pillow's integer arithmetic, so resized pixels match PIL bit for bit.
struct ResizeImg {
  const uint8_t* src;      // (H, W, 3) uint8, interleaved channels (the decoded image as detectron2's mapper holds it)
  const int* xb; const int* xk;   // [newW][2] (first tap, taps), [newW][kx] coefficients
  const int* yb; const int* yk;   // [newH][2], [newH][ky]
  int H, W, newH, newW, kx, ky, flip;
};
struct ResizeBatch { ResizeImg im[PRE_MAX_IMAGES]; };

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

__global__ __launch_bounds__(256) void resize_flip_preprocess_kernel(const ResizeBatch b, __bf16* __restrict__ out, int Hp, int Wp,
                                                                     float m0, float m1, float m2, float s0, float s1, float s2) {
  const ResizeImg& im = b.im[blockIdx.y];
  __bf16* __restrict__ o_n = out + (long long)blockIdx.y * Hp * Wp * 8;
  const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
  const int total = Hp * Wp;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int y = i / Wp, xo = i - y * Wp;
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)0.f;
    if (y < im.newH && xo < im.newW) {
      const int x = im.flip ? im.newW - 1 - xo : xo;            // HFlipTransform of the RESIZED image
      const int x0 = im.xb[2 * x], nx = im.xb[2 * x + 1], y0 = im.yb[2 * y], ny = im.yb[2 * y + 1];
      int acc[3] = {1 << 21, 1 << 21, 1 << 21};                 // vertical pass accumulators, 1 << (PRECISION_BITS - 1)
      for (int r = 0; r < ny; ++r) {
        const uint8_t* row = im.src + ((long long)(y0 + r) * im.W + x0) * 3;
        int h[3] = {1 << 21, 1 << 21, 1 << 21};                 // horizontal pass of source row y0 + r at output column x
        for (int t = 0; t < nx; ++t) {
          const int k = im.xk[(long long)x * im.kx + t];
          h[0] += (int)row[3 * t] * k; h[1] += (int)row[3 * t + 1] * k; h[2] += (int)row[3 * t + 2] * k;
        }
        const int kv = im.yk[(long long)y * im.ky + r];
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += clip8(h[c] >> 22) * kv;      // the intermediate image is uint8 in Pillow
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) o[c] = (__bf16)(((float)clip8(acc[c] >> 22) - mean[c]) / stdv[c]);
    }
    *reinterpret_cast<bf16x8_t*>(o_n + (long long)i * 8) = o;
  }
}

// NHWC bf16 <-> NCHW f32 layout conversion (API boundary only)
__global__ __launch_bounds__(256) void nchw_f32_to_nhwc_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y,
                                                                    int N, int C, int HW) {
  const long long total = (long long)N * C * HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long long r = i / C;
    const int p = (int)(r % HW);
    const int n = (int)(r / HW);
    y[i] = (__bf16)x[((long long)n * C + c) * HW + p];
  }
}

inline int blocks_for(long long n, int cap = 4096) {
  long long g = (n + 255) / 256;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

int gn_check(int N, int HW, int C, int G) {
  if (N <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G) return SOD_EARG;
  const int cpg = C / G;
  if (cpg % 8) return SOD_EARG;
  const int c8n = C / 8;
  if (c8n > 256 || 256 % c8n) return SOD_EARG;
  return SOD_OK;
}

int gn_grid(int HW, int N, int& ppb, int C = 256) {
  // Every block ends with C float atomics on the same few lines, and they, not the reads, set the time of the short passes (round 6, 16 x
  // 100 x 168 x 256: 1024 / 512 / 256 blocks in all -> 42.0 / 30.9 / 25.9 us; five levels in one launch 67.4 / 45.6 / 35.7 us), while a
  // tensor of half a gigabyte wants 512 blocks to keep enough loads in flight (16 x 200 x 336 x 256: 103.6 / 96.1 / 106.8 us).
  const long long bytes = (long long)N * HW * C * 2;
  const int tot = bytes > (256ll << 20) ? 512 : 256;
  int gx = tot / (N > 0 ? N : 1);
  if (gx < 1) gx = 1;
  ppb = (HW + gx - 1) / gx;
  if (ppb < 64) ppb = 64;
  return (HW + ppb - 1) / ppb;
}

}  // namespace

// SOD_GN_REVERSE bit mask: 1 = the forward apply pass, 2 = the backward reduce pass, 4 = the backward apply pass walk their blocks and
// images last to first.  The pass then starts on the part of the tensor its predecessor touched last (what the 256 MB Infinity Cache
// still holds): the backward reduce follows the data gradient that wrote dy, the backward apply follows the reduce pass over the same two
// tensors.  Measured on the FCOS R50 step (one gpurun call, 60 timed steps): mask 0 629.6 / 630.2 img/s, 4: 631.7, 6: 632.3, 5: 631.0,
// 7: 631.7 - default 6.  Results do not depend on the order (float atomics aside; deterministic mode keeps its own fixed order).
static int gn_reverse_mask() {
  static const int v = getenv("SOD_GN_REVERSE") ? atoi(getenv("SOD_GN_REVERSE")) : 6;
  return v;
}

static int gn_fill(GnML& m, int nlev, const int* hw, int N, int C, int G, float eps, int relu, const long long* img_strides, int tot = 1024) {
  if (nlev <= 0 || nlev > GN_MAX_LEVELS || !hw) return SOD_EARG;
  m.nlev = nlev; m.N = N; m.C = C; m.G = G; m.cpg = C / G; m.relu = relu; m.eps = eps;
  int blk = 0;
  for (int l = 0; l < nlev; ++l) {
    int rc = gn_check(N, hw[l], C, G);
    if (rc) return rc;
    GnLevel& L = m.lev[l];
    L.HW = hw[l];
    L.img_stride = (img_strides && img_strides[l] > 0) ? img_strides[l] : (long long)hw[l] * C;
    // ~1024 / N blocks per level for the largest level, proportionally fewer for the small ones (>= 64 pixels per block)
    int gx = tot / (N > 0 ? N : 1);
    if (gx < 1) gx = 1;
    int ppb = (hw[0] + gx - 1) / gx;
    if (ppb < 64) ppb = 64;
    L.pix_per_block = ppb;
    L.blk0 = blk;
    L.nblk = (hw[l] + ppb - 1) / ppb;
    blk += L.nblk;
    L.inv_m = 1.f / ((float)hw[l] * (float)m.cpg);
  }
  return blk;     // total blocks in x (positive) or a negative status
}

// Floats of deterministic-mode scratch a GroupNorm call over these levels needs (0 blocks -> 0).
static long long gn_det_floats(int gx, int N, int C, bool bwd) {
  const long long rows = (long long)gx * N;
  return rows * (C / 8) * 2 + (bwd ? rows * 3 * C : 0);
}

extern "C" int sod_groupnorm_fwd_ml(int nlev, const void* const* x, const float* gamma, const float* beta, void* const* y, float* mean_rstd,
                                    int N, const int* hw, int C, int G, float eps, int relu, float* det_ws, long long det_ws_bytes,
                                    void* stream) {
  if (!x || !gamma || !beta || !y || !mean_rstd) return SOD_EARG;
  GnML m{};
  const int gx = gn_fill(m, nlev, hw, N, C, G, eps, relu, nullptr);
  if (gx <= 0) return gx ? gx : SOD_EARG;
  m.gamma = gamma; m.beta = beta;
  for (int l = 0; l < nlev; ++l) {
    if (!x[l] || !y[l]) return SOD_EARG;
    m.lev[l].x = (const __bf16*)x[l]; m.lev[l].y = (__bf16*)y[l]; m.lev[l].stats = mean_rstd + (size_t)l * N * G * 2;
  }
  hipStream_t st = (hipStream_t)stream;
  if (det_ws) {
    if (gn_det_floats(gx, N, C, false) * (long long)sizeof(float) > det_ws_bytes) return SOD_EARG;
    m.part_grp = det_ws;
    SOD_LAUNCH(gn_stats_kernel, dim3(gx, N), dim3(256), sizeof(float) * 2 * 256, st, m);
    SOD_LAUNCH(gn_det_group_kernel, dim3((N * G + 255) / 256, nlev), dim3(256), 0, st, m, 1);
  } else {
    hipError_t e = hipMemsetAsync(mean_rstd, 0, sizeof(float) * 2 * N * G * nlev, st);
    if (e != hipSuccess) return (int)e;
    SOD_LAUNCH(gn_stats_kernel, dim3(gx, N), dim3(256), sizeof(float) * 2 * 256, st, m);
    SOD_LAUNCH(gn_finalize_stats_kernel, dim3((N * G + 255) / 256, nlev), dim3(256), 0, st, m);
  }
  SOD_LAUNCH(gn_apply_kernel, dim3(gx, N), dim3(256), 0, st, m);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_groupnorm_apply_ml(int nlev, const void* const* x, const float* gamma, const float* beta, void* const* y, float* sums_to_mean_rstd,
                                      int N, const int* hw, int C, int G, float eps, int relu, void* stream) {
  if (!x || !gamma || !beta || !y || !sums_to_mean_rstd) return SOD_EARG;
  GnML m{};
  const int gx = gn_fill(m, nlev, hw, N, C, G, eps, relu, nullptr);
  if (gx <= 0) return gx ? gx : SOD_EARG;
  m.gamma = gamma; m.beta = beta;
  for (int l = 0; l < nlev; ++l) {
    if (!x[l] || !y[l]) return SOD_EARG;
    m.lev[l].x = (const __bf16*)x[l]; m.lev[l].y = (__bf16*)y[l]; m.lev[l].stats = sums_to_mean_rstd + (size_t)l * N * G * 2;
  }
  hipStream_t st = (hipStream_t)stream;
  SOD_LAUNCH(gn_finalize_stats_kernel, dim3((N * G + 255) / 256, nlev), dim3(256), 0, st, m);
  m.rev = gn_reverse_mask() & 1;
  SOD_LAUNCH(gn_apply_kernel, dim3(gx, N), dim3(256), 0, st, m);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_groupnorm_bwd_ml(int nlev, const void* const* dy, const void* const* x, const float* gamma, const float* beta,
                                    const float* mean_rstd, void* const* dx, float* dgamma, float* dbeta, float* dxsum,
                                    float* red_ws /* 2*N*G*nlev floats */, int N, const int* hw, int C, int G, int relu,
                                    float* det_ws, long long det_ws_bytes, void* stream) {
  if (!dy || !x || !gamma || !beta || !mean_rstd || !dx || !dgamma || !dbeta || !red_ws) return SOD_EARG;
  if (!hw || N <= 0 || nlev <= 0 || nlev > GN_MAX_LEVELS) return SOD_EARG;
  // (Round 3 measured the two passes in chunks of images whose dy + x fit the 256 MB Infinity Cache: slower in every configuration -
  // 619.6 / 621.9 img/s without chunks, 617.4 / 620.0 at 184 MB, 612.3 / 618.4 at 128 MB, 609.9 / 617.7 at 96 MB: the other streams move
  // ~1 GB through the same cache between a chunk's two passes, and four dependent kernel pairs expose four grid tails.  Removed in round 5.)
  GnML m{};
  const int gx = gn_fill(m, nlev, hw, N, C, G, 0.f, relu, nullptr);
  if (gx <= 0) return gx ? gx : SOD_EARG;
  m.gamma = gamma; m.beta = beta; m.dgamma = dgamma; m.dbeta = dbeta; m.dxsum = dxsum;
  for (int l = 0; l < nlev; ++l) {
    if (!x[l] || !dy[l] || !dx[l]) return SOD_EARG;
    m.lev[l].x = (const __bf16*)x[l]; m.lev[l].dy = (const __bf16*)dy[l]; m.lev[l].dx = (__bf16*)dx[l];
    m.lev[l].stats = const_cast<float*>(mean_rstd) + (size_t)l * N * G * 2; m.lev[l].red = red_ws + (size_t)l * N * G * 2;
  }
  hipStream_t st = (hipStream_t)stream;
  if (det_ws) {
    if (gn_det_floats(gx, N, C, true) * (long long)sizeof(float) > det_ws_bytes) return SOD_EARG;
    const long long rows = (long long)gx * N;
    m.part_grp = det_ws;
    m.part_gb = det_ws + rows * (C / 8) * 2;
    m.part_dx = dxsum ? m.part_gb + rows * 2 * C : nullptr;
    SOD_LAUNCH(gn_bwd_reduce_kernel, dim3(gx, N), dim3(256), GN_RING, st, m);
    SOD_LAUNCH(gn_det_group_kernel, dim3((N * G + 255) / 256, nlev), dim3(256), 0, st, m, 0);
    SOD_LAUNCH(col_accumulate_kernel, dim3((C + 63) / 64), dim3(256), 0, st, m.part_gb, (int)rows, C, 2 * C, dgamma);
    SOD_LAUNCH(col_accumulate_kernel, dim3((C + 63) / 64), dim3(256), 0, st, m.part_gb + C, (int)rows, C, 2 * C, dbeta);
    SOD_LAUNCH(gn_bwd_apply_kernel, dim3(gx, N), dim3(256), GN_RING, st, m);
    if (dxsum) SOD_LAUNCH(col_accumulate_kernel, dim3((C + 63) / 64), dim3(256), 0, st, m.part_dx, (int)rows, C, C, dxsum);
    SOD_CHECK_LAUNCH();
    return SOD_OK;
  }
  hipError_t e = hipMemsetAsync(red_ws, 0, sizeof(float) * 2 * N * G * nlev, st);
  if (e != hipSuccess) return (int)e;
  // The reduce pass ends every block with 2 C + 2 C / 8 float atomics on the same few lines: half as many blocks of twice the pixels for it
  // (round 6, five tower levels at batch 16: 4096 / 2048 / 1024 / 512 blocks in all -> 309 / 223 / 216 / 204 us for the two passes).
  {
    GnML mr = m;
    const int gxr = gn_fill(mr, nlev, hw, N, C, G, 0.f, relu, nullptr, 512);
    if (gxr <= 0) return gxr ? gxr : SOD_EARG;
    mr.rev = (gn_reverse_mask() >> 1) & 1;
    SOD_LAUNCH(gn_bwd_reduce_kernel, dim3(gxr, N), dim3(256), GN_RING, st, mr);
  }
  m.rev = (gn_reverse_mask() >> 2) & 1;      // (the apply pass keeps the 1024-block grid: 512 / 768 blocks measured 205 / 200 us against 191 for the two passes)
  SOD_LAUNCH(gn_bwd_apply_kernel, dim3(gx, N), dim3(256), GN_RING, st, m);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean_rstd,
                                 int N, int HW, int C, int G, long long img_stride, float eps, int relu, float* det_ws, long long det_ws_bytes,
                                 void* stream) {
  if (!x || !y) return SOD_EARG;
  if (img_stride > 0 && img_stride != (long long)HW * C) return SOD_EARG;     // dense images only
  const void* xs[1] = {x};
  void* ys[1] = {y};
  return sod_groupnorm_fwd_ml(1, xs, gamma, beta, ys, mean_rstd, N, &HW, C, G, eps, relu, det_ws, det_ws_bytes, stream);
}

extern "C" int sod_groupnorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean_rstd,
                                 void* dx, float* dgamma, float* dbeta, float* dxsum, float* red_ws /* 2*N*G floats */,
                                 int N, int HW, int C, int G, long long img_stride, int relu, float* det_ws, long long det_ws_bytes, void* stream) {
  if (!dy || !x || !dx) return SOD_EARG;
  if (img_stride > 0 && img_stride != (long long)HW * C) return SOD_EARG;
  const void* dys[1] = {dy};
  const void* xs[1] = {x};
  void* dxs[1] = {dx};
  return sod_groupnorm_bwd_ml(1, dys, xs, gamma, beta, mean_rstd, dxs, dgamma, dbeta, dxsum, red_ws, N, &HW, C, G, relu, det_ws, det_ws_bytes, stream);
}

extern "C" int sod_relu_bwd(const void* dy, const void* y, void* dx, long long n, void* stream) {
  if (!dy || !y || !dx || n < 0 || (n & 7)) return SOD_EARG;
  SOD_LAUNCH(relu_bwd_kernel, dim3(blocks_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)dy, (const __bf16*)y, (__bf16*)dx, n / 8);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_relu_fwd(const void* x, void* y, long long n, void* stream) {
  if (!x || !y || n < 0 || (n & 7)) return SOD_EARG;
  SOD_LAUNCH(relu_fwd_kernel, dim3(blocks_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, (__bf16*)y, n / 8);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_add_bf16(const void* a, const void* b, void* out, long long n, void* stream) {
  if (!a || !b || !out || n < 0 || (n & 7)) return SOD_EARG;
  SOD_LAUNCH(add_kernel, dim3(blocks_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)a, (const __bf16*)b, (__bf16*)out, n / 8);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_add_up2_bf16(const void* a, const void* b, void* out, int N, int H, int W, int C, void* stream) {
  if (!a || !b || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || (H & 1) || (W & 1)) return SOD_EARG;
  SOD_LAUNCH(add_up2_kernel, dim3(blocks_for((long long)N * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const __bf16*)a, (const __bf16*)b,
             (__bf16*)out, N, H, W, C);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_bias_grad_scaled(const void* dy, float* dbias, const float* scale_num, const float* scale_den, float den_mul, float den_min,
                                    int N, int HW, int C, long long img_stride, float* det_ws, long long det_ws_bytes, void* stream);

extern "C" int sod_bias_grad(const void* dy, float* dbias, int N, int HW, int C, long long img_stride, float* det_ws, long long det_ws_bytes,
                             void* stream) {
  return sod_bias_grad_scaled(dy, dbias, nullptr, nullptr, 1.f, 1.f, N, HW, C, img_stride, det_ws, det_ws_bytes, stream);
}

extern "C" int sod_bias_grad_scaled(const void* dy, float* dbias, const float* scale_num, const float* scale_den, float den_mul, float den_min,
                                    int N, int HW, int C, long long img_stride, float* det_ws, long long det_ws_bytes, void* stream) {
  if (!dy || !dbias || N <= 0 || HW <= 0 || C <= 0 || (C & 7) || C > 2048) return SOD_EARG;
  if (img_stride <= 0) img_stride = (long long)HW * C;
  const int c8n = C / 8;
  if (c8n > 256) return SOD_EARG;
  int ppb;
  const int gx = gn_grid(HW, N, ppb, C);
  if (det_ws && (long long)gx * N * C * (long long)sizeof(float) > det_ws_bytes) return SOD_EARG;
  SOD_LAUNCH(channel_sum_kernel, dim3(gx, N), dim3(256), sizeof(float) * 8 * 256, (hipStream_t)stream, (const __bf16*)dy, dbias, HW, C, img_stride, ppb,
             det_ws, scale_num, scale_den, den_mul, den_min);
  if (det_ws) SOD_LAUNCH(col_accumulate_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, det_ws, gx * N, C, C, dbias);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_bias_grad_ml(int nlev, const void* const* dy, float* dbias, int N, const int* hw, int C, void* stream) {
  if (!dy || !dbias || !hw || nlev <= 0 || nlev > GN_MAX_LEVELS || N <= 0 || C <= 0 || (C & 7) || C > 2048) return SOD_EARG;
  ChanSumML m{};
  m.nlev = nlev; m.C = C;
  int blk = 0, hw_max = 0;
  for (int l = 0; l < nlev; ++l) {
    if (!dy[l] || hw[l] <= 0) return SOD_EARG;
    // pixels per block from the largest level for all of them: the small levels add a block or two per image, not sixteen
    if (l == 0 || hw[l] > hw_max) { hw_max = hw[l]; }
    m.dy[l] = (const __bf16*)dy[l]; m.HW[l] = hw[l]; m.blk0[l] = 0;
  }
  int ppb;
  long long hw_all = 0;
  for (int l = 0; l < nlev; ++l) hw_all += hw[l];
  gn_grid(hw_max, N, ppb, (int)((long long)C * hw_all / hw_max));      // (sized by the bytes of all levels)
  for (int l = 0; l < nlev; ++l) {
    m.ppb[l] = ppb; m.blk0[l] = blk;
    blk += (hw[l] + ppb - 1) / ppb;
  }
  SOD_LAUNCH(channel_sum_ml_kernel, dim3(blk, N), dim3(256), sizeof(float) * 8 * 256, (hipStream_t)stream, m, dbias);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_maxpool3x3s2(const void* x, void* y, int N, int H, int W, int C, void* stream) {
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return SOD_EARG;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  SOD_LAUNCH(maxpool3x3s2_kernel, dim3(blocks_for((long long)N * Ho * Wo * (C / 8), 16384)), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16*)x, (__bf16*)y, N, H, W, C, Ho, Wo);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_upsample2x_bwd(const void* g, void* dprev, int N, int Hc, int Wc, int C, void* stream) {
  if (!g || !dprev || N <= 0 || Hc <= 0 || Wc <= 0 || C <= 0 || (C & 7)) return SOD_EARG;
  SOD_LAUNCH(upsample2x_bwd_kernel, dim3(blocks_for((long long)N * Hc * Wc * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16*)g, (__bf16*)dprev, N, Hc, Wc, C);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_weight_prep(const float* w, const float* scale, void* w_krsc, void* w_crsk, int K, int RS, int C, int Cpad, void* stream) {
  if (!w || (!w_krsc && !w_crsk) || K <= 0 || RS <= 0 || C <= 0 || Cpad < C) return SOD_EARG;
  SOD_LAUNCH(weight_prep_kernel, dim3(blocks_for((long long)K * RS * C)), dim3(256), 0, (hipStream_t)stream, w, scale,
                     (__bf16*)w_krsc, (__bf16*)w_crsk, K, RS, C, Cpad);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_weight_prep_batched(const float* params, const float* scales, const void* table_dev, int n, long long total_elems,
                                       void* krsc_arena, void* crsk_arena, void* stream) {
  if (!params || !table_dev || n <= 0 || total_elems <= 0 || !krsc_arena || !crsk_arena) return SOD_EARG;
  SOD_LAUNCH(weight_prep_batched_kernel, dim3((unsigned)(total_elems < 16384 ? total_elems : 16384)), dim3(256), 0, (hipStream_t)stream, params, scales,
             (const PrepEntry*)table_dev, n, total_elems, (__bf16*)krsc_arena, (__bf16*)crsk_arena);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_scale_rows(float* g, const float* scale, int K, long long row, void* stream) {
  if (!g || !scale || K <= 0 || row <= 0) return SOD_EARG;
  SOD_LAUNCH(scale_rows_kernel, dim3(blocks_for((long long)K * row)), dim3(256), 0, (hipStream_t)stream, g, scale, K, row);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_sgd_step(float* params, const float* grads, float* momentum_buf, const void* segments_dev, int nseg,
                            const float* lr_dev, float lr, float momentum, int nesterov, int first_step, float grad_scale, void* stream) {
  if (!params || !grads || !segments_dev || nseg <= 0 || (momentum != 0.f && !momentum_buf)) return SOD_EARG;
  SOD_LAUNCH(sgd_kernel, dim3(512, nseg), dim3(256), 0, (hipStream_t)stream, params, grads, momentum_buf,
                     (const SgdSeg*)segments_dev, nseg, lr_dev, lr, momentum, nesterov, first_step, grad_scale);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_adaptive_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const void* segments_dev, int nseg,
                                 int mode, float lr, float beta1, float beta2, float eps, float bias_correction1, float bias_correction2_sqrt,
                                 float grad_scale, void* stream) {
  if (!params || !grads || !exp_avg_sq || !segments_dev || nseg <= 0 || mode < 0 || mode > 2 || (mode != 2 && !exp_avg)) return SOD_EARG;
  if (mode != 2 && !(bias_correction1 > 0.f && bias_correction2_sqrt > 0.f)) return SOD_EARG;
  SOD_LAUNCH(adaptive_kernel, dim3(512, nseg), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
                     (const SgdSeg*)segments_dev, mode, lr, beta1, beta2, eps, bias_correction1, bias_correction2_sqrt, grad_scale);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_preprocess_image(const void* img, int is_uint8, int C, int H, int W, void* out, int Hp, int Wp, int Cpad,
                                    const float* mean3, const float* std3, void* stream) {
  if (!img || !out || C <= 0 || C > 3 || H <= 0 || W <= 0 || Hp < H || Wp < W || Cpad != 8 || !mean3 || !std3) return SOD_EARG;
  const int g = blocks_for((long long)Hp * Wp);
  if (is_uint8)
    SOD_LAUNCH(preprocess_kernel<uint8_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)img, C, H, W, (__bf16*)out, Hp, Wp, Cpad,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  else
    SOD_LAUNCH(preprocess_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)img, C, H, W, (__bf16*)out, Hp, Wp, Cpad,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_preprocess_batch(int n, const void* const* imgs, int is_uint8, int C, const int* H, const int* W, void* out, int Hp, int Wp,
                                    int Cpad, const float* mean3, const float* std3, void* stream) {
  if (n <= 0 || n > PRE_MAX_IMAGES || !imgs || !H || !W || !out || C <= 0 || C > 3 || Cpad != 8 || !mean3 || !std3 || Hp <= 0 || Wp <= 0) return SOD_EARG;
  if ((long long)Hp * Wp >= (1ll << 31)) return SOD_ESIZE;
  PreBatch b;
  for (int i = 0; i < n; ++i) {
    if (!imgs[i] || H[i] <= 0 || W[i] <= 0 || H[i] > Hp || W[i] > Wp) return SOD_EARG;
    b.img[i] = imgs[i]; b.H[i] = H[i]; b.W[i] = W[i];
  }
  const dim3 grid(blocks_for((long long)Hp * Wp, 1024), n);
  if (is_uint8)
    SOD_LAUNCH(preprocess_batch_kernel<uint8_t>, grid, dim3(256), 0, (hipStream_t)stream, b, C, (__bf16*)out, Hp, Wp, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  else
    SOD_LAUNCH(preprocess_batch_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, b, C, (__bf16*)out, Hp, Wp, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_resize_flip_preprocess_batch(int n, const void* const* imgs, const int* H, const int* W, const int* newH, const int* newW,
                                               const int* const* xbounds, const int* const* xcoef, const int* kx,
                                               const int* const* ybounds, const int* const* ycoef, const int* ky, const int* flip,
                                               void* out, int Hp, int Wp, int Cpad, const float* mean3, const float* std3, void* stream) {
  if (!imgs || !H || !W || !newH || !newW || !xbounds || !xcoef || !kx || !ybounds || !ycoef || !ky || !flip || !out || !mean3 || !std3) return SOD_EARG;
  if (n <= 0 || n > PRE_MAX_IMAGES || Cpad != 8 || Hp <= 0 || Wp <= 0 || (long long)Hp * Wp >= (1ll << 31)) return SOD_EARG;
  ResizeBatch b;
  for (int i = 0; i < n; ++i) {
    if (!imgs[i] || !xbounds[i] || !xcoef[i] || !ybounds[i] || !ycoef[i]) return SOD_EARG;
    if (H[i] <= 0 || W[i] <= 0 || newH[i] <= 0 || newW[i] <= 0 || newH[i] > Hp || newW[i] > Wp || kx[i] <= 0 || ky[i] <= 0) return SOD_EARG;
    ResizeImg& im = b.im[i];
    im.src = (const uint8_t*)imgs[i]; im.xb = xbounds[i]; im.xk = xcoef[i]; im.yb = ybounds[i]; im.yk = ycoef[i];
    im.H = H[i]; im.W = W[i]; im.newH = newH[i]; im.newW = newW[i]; im.kx = kx[i]; im.ky = ky[i]; im.flip = flip[i] ? 1 : 0;
  }
  SOD_LAUNCH(resize_flip_preprocess_kernel, dim3(blocks_for((long long)Hp * Wp, 2048), n), dim3(256), 0, (hipStream_t)stream, b, (__bf16*)out, Hp, Wp,
             mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_nchw_f32_to_nhwc_bf16(const float* x, void* y, int N, int C, int HW, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || HW <= 0) return SOD_EARG;
  SOD_LAUNCH(nchw_f32_to_nhwc_bf16_kernel, dim3(blocks_for((long long)N * C * HW)), dim3(256), 0, (hipStream_t)stream, x, (__bf16*)y, N, C, HW);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}
