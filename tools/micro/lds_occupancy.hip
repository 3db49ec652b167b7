// How many 256-thread workgroups with a given dynamic LDS allocation does a CU of this GPU hold?  (hipOccupancyMaxActiveBlocksPerMultiprocessor;
// the answer decides whether the fused DeformConv backward keeps three workgroups per CU when its allocation grows.)
//   hipcc --offload-arch=gfx950 -o tools/micro/lds_occupancy tools/micro/lds_occupancy.hip && tools/micro/lds_occupancy
#include <hip/hip_runtime.h>
#include <stdio.h>
extern __shared__ int dyn[];
__global__ __launch_bounds__(256) void probe(int* out) { dyn[threadIdx.x] = threadIdx.x; __syncthreads(); if (out) out[threadIdx.x] = dyn[255 - threadIdx.x]; }
int main() {
  int dev = 0, lds_cu = 0, lds_blk = 0;
  hipDeviceGetAttribute(&lds_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev);
  hipDeviceGetAttribute(&lds_blk, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
  printf("LDS per CU %d B, per block (default limit) %d B\n", lds_cu, lds_blk);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  const int sizes[] = {40960, 47104, 48228, 51200, 51760, 52224, 53248, 54272, 54613, 56000, 65536, 81920};
  for (int s : sizes) {
    int n = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)probe, 256, (size_t)s);
    printf("dynamic LDS %6d B: %d workgroups per CU (%s)\n", s, n, hipGetErrorString(e));
  }
  return 0;
}
