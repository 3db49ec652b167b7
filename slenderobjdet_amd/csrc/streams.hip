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
