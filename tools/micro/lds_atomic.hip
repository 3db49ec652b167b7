// Micro-benchmark: LDS atomic throughput on gfx950 (ds_add_f32 vs ds_add_u32 vs plain read-modify-write), conflict-free lane-linear
// addresses and a strided (bank-conflicting) pattern.  hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics lds_atomic.hip -o lds_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int stride) {
  __shared__ float buf[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) buf[i] = 0.f;
  __syncthreads();
  const int base = (threadIdx.x * stride) & 8191;
  float v = 1.f + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = (base + e * 256 + it) & 8191;
      if (MODE == 0) atomicAdd(&buf[idx], v);
      else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(buf) + idx, (unsigned)threadIdx.x);
      else buf[idx] += v;
    }
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = buf[threadIdx.x];
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, blocks = 2048;
  for (int stride = 1; stride <= 32; stride *= 32)
    for (int mode = 0; mode < 3; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, stride);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, stride);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, stride);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      double ops = (double)blocks * 256 * iters * 8;
      printf("stride %2d mode %d (%s): %.3f ms, %.1f G lane-ops/s, %.2f lanes/clk/CU @2.4GHz 256CU\n", stride, mode,
             mode == 0 ? "ds_add_f32" : mode == 1 ? "ds_add_u32" : "plain rmw", ms, ops / ms / 1e6, ops / (ms * 1e-3) / 2.4e9 / 256);
    }
  return 0;
}
