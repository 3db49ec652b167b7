// Shared device/host helpers for the slenderobjdet_amd HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

#define SOD_LDS(p) ((__attribute__((address_space(3))) void*)(p))
// Byte offset handed to buffer loads for "this lane is padding": always beyond num_records,
// so the hardware range check returns zeros (tensors are limited to < 2 GiB by the host side).
#define SOD_OOB 0x80000000u

// status codes of the C ABI
#define SOD_OK 0
#define SOD_EARG (-1)      // bad argument / unsupported shape
#define SOD_ESIZE (-2)     // tensor too large for 32-bit buffer addressing
#define SOD_EALIGN (-3)    // pointer alignment

// Division by a runtime constant via multiply-high; exact for n < 2^31.
struct FastDiv {
  uint32_t mul, shr, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  if (d <= 1) { f.mul = 0; f.shr = 0; return f; }
  uint32_t l = 0;
  while ((1u << l) < d) ++l;             // l = ceil(log2 d)
  uint64_t m = ((uint64_t(1) << 32) * ((uint64_t(1) << l) - d)) / d + 1;
  f.mul = (uint32_t)m;
  f.shr = l;
  return f;
}
__device__ __forceinline__ uint32_t fd_div(uint32_t n, const FastDiv& f) {
  // d == 1 is encoded as mul = 0, shr = 0 and needs no special case: a branch here ends up inside every K loop that divides
  const uint32_t t = __umulhi(n, f.mul);
  return (t + n) >> f.shr;               // n < 2^31 => no overflow
}

__device__ __forceinline__ float bf16_to_f32(__bf16 v) { return (float)v; }
__device__ __forceinline__ __bf16 f32_to_bf16(float v) { return (__bf16)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread.
__device__ __forceinline__ float block_sum_256(float v, float* red /* >=4 floats LDS */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// Bijective XCD-aware remap of a 1-D block id (guide T1): consecutive logical ids land on one XCD.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblk) {
  const uint32_t q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  const uint32_t base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// 16-byte global store of a kernel's OUTPUT tensor.  (Round 3 measured write-through "sc1" stores here - the end-of-kernel release then
// finds no dirty lines in the eight L2s -: 616-619 -> 611-612 img/s, slower, because consumers on the same XCD lose their L2 hits; removed.)
template <typename V>
__device__ __forceinline__ void sod_store16(void* p, V v) {
  static_assert(sizeof(V) == 16, "16-byte vectors only");
  *reinterpret_cast<V*>(p) = v;
}

#define SOD_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

// Launch + error attribution: clear any stale (sticky-less) error left by other HIP users in this thread first.
#define SOD_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
