// The frozen ResNet stem as ONE kernel: preprocess_image (normalise + zero pad) -> conv 7x7 stride 2 (+ folded FrozenBatchNorm2d)
// -> ReLU -> max-pool 3x3 stride 2, from the decoded uint8 image straight to the 64-channel stride-4 feature map.
//
// Replaces FCOSV2.preprocess_image (slender_det/modeling/meta_arch/fcos/fcosv2.py:268-275) + detectron2 BasicStem (source absent;
// SURVEY.md Appendix C.9) = three HBM-bound passes of the un-fused path (per 16 x 800x1344 batch: 275 MB normalised NHWC(8) input
// written and read, 550 MB conv output written and read by the pool) with 51 MB of uint8 in and 138 MB out.
//
// One workgroup (4 waves) = an 8x8 tile of POOLED pixels of one image:
//   * the 39x39x3 input patch it depends on is loaded as uint8, normalised ((v - mean) / std, one rounding to bf16 as
//     sod_preprocess_batch does; zero outside the image = the padding of ImageList.from_tensors and of the convolution) into LDS,
//     channel-planar, rows of 48 bf16;
//   * the 17x17 conv outputs under the tile (1-pixel halo for the pool: 13 % recomputation) are an implicit GEMM on MFMA 16x16x32:
//     M = 64 output channels (4 tiles, the weights live in REGISTERS for the whole workgroup: 96 VGPRs), N = 289 conv pixels
//     (19 tiles, round-robin over the waves), K = (channel, kernel row, 8 kernel columns) = 24 chunks of 8 (7 columns + 1 zero,
//     21 (c, r) pairs + 3 zero chunks) so that a B fragment is 8 CONSECUTIVE patch columns: four ds_read_b32 (the patch column of a
//     stride-2 conv pixel is 4-byte aligned);
//   * epilogue: + folded BN shift, ReLU, bf16, into an LDS tile [289 px][64 ch]; then the 3x3 stride-2 max over that tile (conv
//     pixels outside the conv output are skipped, as -inf padding does) and 16-B stores of the pooled rows.
#include "common.h"
#include "../../include/slender_hip.h"

namespace {

constexpr int ST_MAX_IMAGES = 64;
constexpr int ST_PT = 39;                 // patch side: 2 * 16 + 7
constexpr int ST_PSTR = 48;               // patch row pitch (bf16): 39 columns + 9 zero columns (the 8th kernel column reads them)
constexpr int ST_CT = 17;                 // conv tile side: 2 * 8 + 1
constexpr int ST_NPX = ST_CT * ST_CT;     // 289
constexpr int ST_OSTR = 72;               // conv-out row pitch (bf16): 64 channels + 8 pad (144 B: rows shift by 4 banks)
constexpr int ST_PATCH_BYTES = 3 * ST_PT * ST_PSTR * 2;                 // 11 232
constexpr int ST_LDS = ST_PATCH_BYTES + ST_NPX * ST_OSTR * 2;           // + 41 616 = 52 848: three workgroups per CU

struct StemArgs {
  const uint8_t* img[ST_MAX_IMAGES];      // (3, H, W) uint8, channel planes
  int H[ST_MAX_IMAGES], W[ST_MAX_IMAGES];
  const __bf16* w;                        // [64][24][8]: w[k][c*7 + r][s] (s = 7 and chunks 21..23 zero)
  const float* bias;                      // [64] folded FrozenBN shift
  __bf16* out;                            // (N, Hq, Wq, 64)
  int N, Hc, Wc, Hq, Wq;                  // conv output / pooled output sizes of the padded batch
  float mean[3], stdv[3];
};

__global__ __launch_bounds__(256) void stem_fused_kernel(const StemArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* patch = reinterpret_cast<__bf16*>(smem);
  __bf16* cout = reinterpret_cast<__bf16*>(smem + ST_PATCH_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tx = blockIdx.x, ty = blockIdx.y, n = blockIdx.z;
  const int H = a.H[n], W = a.W[n];
  const uint8_t* __restrict__ img = a.img[n];
  const int cy0 = 16 * ty - 1, cx0 = 16 * tx - 1;          // conv coordinates of the tile's first row / column
  const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;          // input coordinates of the patch's first row / column

  // ---- weights -> registers (A operand: lane = output channel m, 8 consecutive k)
  const int fr = lane & 15, fg = lane >> 4;
  bf16x8_t af[4][6];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) af[mi][ks] = *reinterpret_cast<const bf16x8_t*>(a.w + ((mi * 16 + fr) * 24 + ks * 4 + fg) * 8);

  // ---- patch: uint8 -> normalised bf16, zero outside the image; columns 39..47 of every row are zero (the 8th kernel column).
  // The byte loads are UNCONDITIONAL (out-of-image positions read byte 0 of the image and are zeroed afterwards): a load inside a
  // lane-dependent branch makes hipcc wait vmcnt(0) per element (22 serial memory latencies per workgroup, measured).  The
  // normalisation is a 3 x 256 table in LDS ((v - mean) / std with the division of sod_preprocess_batch, one rounding to bf16).
  __bf16* lut = cout;                          // the conv-out tile is not live yet
  for (int i = tid; i < 3 * 256; i += 256) lut[i] = (__bf16)(((float)(i & 255) - a.mean[i >> 8]) / a.stdv[i >> 8]);
  // thread -> patch column tid % 48 (threads 240..255 idle here), rows (tid / 48) + 5 j of the 117 (channel, row) pairs
  constexpr int PER_THREAD = (3 * ST_PT + 4) / 5;
  const int px = tid % 48, r0 = tid / 48;
  const int ix = ix0 + px;
  const bool colok = tid < 240 && px < ST_PT && (unsigned)ix < (unsigned)W;
  auto irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, (uint32_t)(3 * H * W), 0x00020000);
  uint8_t raw[PER_THREAD];
#pragma unroll
  for (int j = 0; j < PER_THREAD; ++j) {      // buffer loads: out-of-image positions use the out-of-range offset (returns 0), no branch
    const int t = r0 + 5 * j;                 // (channel, patch row) pair
    const int c = t >= 2 * ST_PT ? 2 : (t >= ST_PT ? 1 : 0), py = t - c * ST_PT;
    const int iy = iy0 + py;
    const bool ok = colok && t < 3 * ST_PT && (unsigned)iy < (unsigned)H;
    raw[j] = __builtin_amdgcn_raw_buffer_load_b8(irsrc, ok ? (uint32_t)((c * H + iy) * W + ix) : SOD_OOB, 0, 0);
  }
  __syncthreads();                              // the table is complete
#pragma unroll
  for (int j = 0; j < PER_THREAD; ++j) {
    const int t = r0 + 5 * j;
    if (tid < 240 && t < 3 * ST_PT) {
      const int c = t >= 2 * ST_PT ? 2 : (t >= ST_PT ? 1 : 0), py = t - c * ST_PT;
      const bool ok = colok && (unsigned)(iy0 + py) < (unsigned)H;
      const __bf16 v = lut[(c << 8) + raw[j]];
      patch[t * ST_PSTR + px] = ok ? v : (__bf16)0.f;
    }
  }
  __syncthreads();

  // ---- conv: 19 pixel tiles of 16, round-robin over the 4 waves
  float bv[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[mi][e] = a.bias[mi * 16 + fg * 4 + e];
  for (int nt = wave; nt < (ST_NPX + 15) / 16; nt += 4) {
    const int px = nt * 16 + fr;
    const int pc = px < ST_NPX ? px : ST_NPX - 1;          // lanes past the tile compute a duplicate that is never stored
    const int cy = pc / ST_CT, cx = pc - cy * ST_CT;
    f32x4_t acc[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) acc[mi] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
      const int kc = ks * 4 + fg;                          // (channel, kernel row) pair of this lane's 8 contraction elements
      bf16x8_t bf;
      if (kc < 21) {
        const int c = kc / 7, r = kc - c * 7;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(patch + (c * ST_PT + 2 * cy + r) * ST_PSTR + 2 * cx);
        typedef __attribute__((ext_vector_type(4))) unsigned int u4;
        const u4 raw = {src[0], src[1], src[2], src[3]};
        bf = __builtin_bit_cast(bf16x8_t, raw);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bf[e] = (__bf16)0.f;
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi][ks], bf, acc[mi], 0, 0, 0);
    }
    if (px < ST_NPX) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        bf16x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)fmaxf(acc[mi][e] + bv[mi][e], 0.f);
        *reinterpret_cast<bf16x4_t*>(cout + px * ST_OSTR + mi * 16 + fg * 4) = o;
      }
    }
  }
  __syncthreads();

  // ---- max-pool 3x3 stride 2 pad 1 over the conv tile: item = (pooled pixel, 8-channel chunk)
  for (int item = tid; item < 64 * 8; item += 256) {
    const int ch = item & 7, pp = item >> 3;
    const int py = pp >> 3, pxx = pp & 7;
    const int gy = ty * 8 + py, gx = tx * 8 + pxx;
    if (gy >= a.Hq || gx >= a.Wq) continue;
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -3.0e38f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int ly = 2 * py + dy;
      if ((unsigned)(cy0 + ly) >= (unsigned)a.Hc) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int lx = 2 * pxx + dx;
        if ((unsigned)(cx0 + lx) >= (unsigned)a.Wc) continue;
        const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(cout + (ly * ST_CT + lx) * ST_OSTR + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], (float)v[e]);
      }
    }
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)m[e];
    *reinterpret_cast<bf16x8_t*>(a.out + (((long long)n * a.Hq + gy) * a.Wq + gx) * 64 + ch * 8) = o;
  }
}

}  // namespace

extern "C" int sod_stem_fused(int n, const void* const* imgs, const int* H, const int* W, const void* w_packed, const float* bias,
                              void* out, int Hp, int Wp, const float* mean3, const float* std3, void* stream) {
  if (!imgs || !H || !W || !w_packed || !bias || !out || !mean3 || !std3) return SOD_EARG;
  if (n <= 0 || n > ST_MAX_IMAGES || Hp <= 0 || Wp <= 0 || (Hp & 3) || (Wp & 3)) return SOD_EARG;
  StemArgs a{};
  for (int i = 0; i < n; ++i) {
    if (!imgs[i] || H[i] <= 0 || W[i] <= 0 || H[i] > Hp || W[i] > Wp) return SOD_EARG;
    a.img[i] = (const uint8_t*)imgs[i]; a.H[i] = H[i]; a.W[i] = W[i];
  }
  a.w = (const __bf16*)w_packed; a.bias = bias; a.out = (__bf16*)out; a.N = n;
  a.Hc = (Hp + 6 - 7) / 2 + 1; a.Wc = (Wp + 6 - 7) / 2 + 1;
  a.Hq = (a.Hc + 2 - 3) / 2 + 1; a.Wq = (a.Wc + 2 - 3) / 2 + 1;
  for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.stdv[c] = std3[c]; }
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)stem_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  SOD_LAUNCH(stem_fused_kernel, dim3((a.Wq + 7) / 8, (a.Hq + 7) / 8, n), dim3(256), ST_LDS, (hipStream_t)stream, a);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}
