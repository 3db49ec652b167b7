// Stream construction the torch stream API does not offer: a HIP stream restricted to a subset of the 256 CUs
// (hipExtStreamCreateWithCUMask).  Used by the training step's side streams (weight gradients, frozen-prefix look-ahead, box tower)
// so that whole-CU workgroups of one stream cannot evict the other stream's: see DESIGN.md section 4 "CU-partitioned streams".
#include "common.h"
#include "../../include/slender_hip.h"

extern "C" int sod_stream_create_cumask(const unsigned* mask_words, int nwords, void** out_stream) {
  if (mask_words == nullptr || out_stream == nullptr || nwords <= 0 || nwords > 32) return SOD_EARG;
  hipStream_t s = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)nwords, mask_words);
  if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
  *out_stream = (void*)s;
  return SOD_OK;
}

extern "C" int sod_stream_destroy(void* stream) {
  if (stream == nullptr) return SOD_EARG;
  hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
  return SOD_OK;
}

// Rehearsal aid for the N-GPU path on a one-GPU box (bench.py --rccl-rehearsal --rehearsal-occupancy): `wgs` workgroups of 256 threads that
// stay resident for `usec` microseconds, as RCCL's channel kernels do while a gradient bucket is on the wire.  A one-rank all-reduce moves
// no data and occupies nothing, so it cannot show what the whole-CU workgroups of this library (one per CU, all registers of the CU) lose
// while some CUs hold a foreign wave.  Every wave leaves after the interval (wall_clock64 runs at 100 MHz, independent of the shader clock).
__global__ __launch_bounds__(256) void occupy_kernel(long long ticks) {
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

extern "C" int sod_debug_occupy(int wgs, int usec, void* stream) {
  if (wgs <= 0 || wgs > 256 || usec <= 0 || usec > 100000) return SOD_EARG;
  SOD_LAUNCH(occupy_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, (long long)usec * 100ll);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}
