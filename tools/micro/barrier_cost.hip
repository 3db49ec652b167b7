// Micro-benchmark: what one s_barrier costs an 8-wave (512-thread) workgroup that owns its CU, gfx950.
//   mode 0: barrier only                     mode 1: barrier + 16 MFMAs per wave between barriers
//   mode 2: 16 MFMAs per wave, no barrier    mode 3: two barriers per 16 MFMAs (the phase structure of the 256x256 kernels)
//   mode 4: as 3, with the two wave rows staggered by one barrier (row 1 runs one barrier behind row 0)
//   mode 5: as 3 + 6 VALU adds right AFTER each barrier release (the address arithmetic of the fragment reads)
//   mode 6: as 3 + the same 12 VALU adds placed in the middle of the MFMA cluster instead
//   mode 7: as 5 + 8 ds_read_b64 after the VALU adds and an lgkmcnt(0) before the MFMAs (a phase of the weight-gradient kernel)
//   mode 8: as 7 with the reads addressed by loop-invariant registers (no VALU between barrier and reads)
//   mode 9: as 8 with the two wave rows staggered by one barrier: one row's reads run beside the other row's MFMAs
//   mode 10: as 9 with 12 ds_read_b128 per phase (the forward / data-gradient kernel's fragment bytes) instead of 8 ds_read_b64
// Reports cycles (s_memtime) per loop trip, median over workgroups.   hipcc -O3 --offload-arch=gfx950 barrier_cost.hip -o barrier_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* cyc, float* sink, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x4_t acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(e + 1); }
  if ((MODE == 4 || MODE >= 9) && wave >= 4) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __shared__ unsigned long long lbuf[4096];
  for (int i = threadIdx.x; i < 4096; i += 512) lbuf[i] = i;
  __syncthreads();
  unsigned v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5;
  unsigned long long rsum = 0;
  const unsigned lbase = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned long long*)lbuf + (threadIdx.x & 63) * 8;
#define VALU6 asm volatile("v_add_u32 %0, %0, %6\n v_add_u32 %1, %1, %6\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %6\n v_add_u32 %4, %4, %6\n v_add_u32 %5, %5, %6" \
                           : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5) : "s"(it));
#define READS8(ADDR) { unsigned long long r0, r1, r2, r3, r4, r5, r6, r7; \
      asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n" \
                   "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n s_waitcnt lgkmcnt(0)" \
                   : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(ADDR)); \
      rsum += r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7; }
#define READS12(ADDR) { f32x4_t q0, q1, q2, q3, q4, q5, q6, q7, q8, q9, q10, q11; \
      asm volatile("ds_read_b128 %0, %12\n ds_read_b128 %1, %12 offset:1024\n ds_read_b128 %2, %12 offset:2048\n ds_read_b128 %3, %12 offset:3072\n" \
                   "ds_read_b128 %4, %12 offset:4096\n ds_read_b128 %5, %12 offset:5120\n ds_read_b128 %6, %12 offset:6144\n ds_read_b128 %7, %12 offset:7168\n" \
                   "ds_read_b128 %8, %12 offset:8192\n ds_read_b128 %9, %12 offset:9216\n ds_read_b128 %10, %12 offset:10240\n ds_read_b128 %11, %12 offset:11264\n s_waitcnt lgkmcnt(0)" \
                   : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7), "=v"(q8), "=v"(q9), "=v"(q10), "=v"(q11) : "v"(ADDR16)); \
      rsum += (unsigned long long)(q0[0] + q1[1] + q2[2] + q3[3] + q4[0] + q5[1] + q6[2] + q7[3] + q8[0] + q9[1] + q10[2] + q11[3]); }
  const unsigned ADDR16 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned long long*)lbuf + (threadIdx.x & 63) * 16;
  for (int it = 0; it < iters; ++it) {
    if (MODE != 2) __builtin_amdgcn_s_barrier();
    if (MODE == 5 || MODE == 7) { VALU6 }
    if (MODE == 7) READS8(lbase + ((v0 - threadIdx.x) & 0))
    if (MODE == 8 || MODE == 9) READS8(lbase)
    if (MODE == 10) READS12(lbase)
    if (MODE >= 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      if (MODE == 6) { VALU6 VALU6 }
#pragma unroll
      for (int i = 8; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    if (MODE >= 3) __builtin_amdgcn_s_barrier();
    if (MODE == 5 || MODE == 7) { VALU6 }
  }
  if (rsum == 0x123456789ull || v0 + v1 + v2 + v3 + v4 + v5 == 0x7fffffffu) sink[1] = 1.f;
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((MODE == 4 || MODE >= 9) && wave < 4) __builtin_amdgcn_s_barrier();
  f32x4_t s = acc[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) s += acc[i];
  if (s[0] == 123.456f) sink[threadIdx.x] = s[1];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const int blocks = 256, iters = 4000;
  unsigned long long* cyc;
  float* sink;
  hipMalloc(&cyc, blocks * 8);
  hipMalloc(&sink, 4096);
  std::vector<unsigned long long> h(blocks);
  const char* names[11] = {"barrier only", "barrier + 16 MFMA", "16 MFMA, no barrier", "barrier + 16 MFMA + barrier", "as before, wave rows staggered",
                          "6 VALU after each barrier", "12 VALU inside the MFMA cluster", "6 VALU + 8 LDS reads + wait, then MFMAs", "8 LDS reads (no VALU) + wait",
                          "8 LDS reads + wait, rows staggered", "12 ds_read_b128 + wait, rows staggered"};
  for (int mode = 0; mode < 11; ++mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, cyc, sink, iters);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, cyc, sink, iters);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, cyc, sink, iters);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), 0, 0, cyc, sink, iters);
      if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(512), 0, 0, cyc, sink, iters);
      if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(512), 0, 0, cyc, sink, iters);
      if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(512), 0, 0, cyc, sink, iters);
      if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(512), 0, 0, cyc, sink, iters);
      if (mode == 8) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(512), 0, 0, cyc, sink, iters);
      if (mode == 9) hipLaunchKernelGGL(k<9>, dim3(blocks), dim3(512), 0, 0, cyc, sink, iters);
      if (mode == 10) hipLaunchKernelGGL(k<10>, dim3(blocks), dim3(512), 0, 0, cyc, sink, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("mode %d %-34s: %.1f shader cycles per trip (s_memtime, median workgroup), wall %.1f ns per trip\n", mode, names[mode],
           (double)h[blocks / 2] / iters, ms * 1e6 / iters);
  }
  return 0;
}
