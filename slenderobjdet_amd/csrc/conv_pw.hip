// Expanding 1x1 convolutions of the bottleneck blocks as a PERSISTENT, WEIGHT-STATIONARY kernel (gfx950).
//
//   forward   conv3 of a detectron2 BottleneckBlock: y = relu(x W^T + b + shortcut) (+ 1-bit ReLU mask)       128 / 256 / 512 -> 512 / 1024 / 2048
//   backward  conv1's data gradient: dx = bits(block input) ? dy W + (gradient of the identity path) : 0      same shapes
//   (build_resnet_backbone reached from slender_det/modeling/backbone/fpn.py:103)
//
// These launches are the slowest per FLOP of the step (250 - 630 TFLOP/s) and are memory-bound: a 128x128-tile workgroup of conv_igemm.hip
// re-reads the pixel tile once per output-channel tile and the weight tile once per pixel tile (res4 conv3: 825 MB through the CUs for
// 310 MB of HBM traffic), and what a CU can keep in flight bounds the rate.  Here:
//   * ONE workgroup of 8 waves per CU owns ONE 128-channel slice of the output for the whole launch; its weights [128][C] live in
//     registers as MFMA A operands (32 / 64 / 128 VGPRs per lane for C = 128 / 256 / 512): no weight byte is re-read;
//   * it walks over pixel tiles of PT = 16384 / C pixels; the nq = Nout / 128 workgroups that share a pixel tile sit on ONE XCD and
//     take the tile from that XCD's L2 (one HBM read per tile);
//   * everything a tile needs arrives by LDS-DMA one or two tiles ahead - the pixel tile [PT][C] (32 KB, ring of NX), the shortcut /
//     accumulate tile [PT][128] (ring of 2, it doubles as the staging tile of the output) and the mask bits - under ONE counted
//     vmcnt per tile, so ~100 KB per CU are in flight all the time; stores are buffer stores (out-of-range offsets for dead rows);
//   * wave (wq, wp) = (wave >> 1, wave & 1) computes 32 channels x PT/2 pixels: a B fragment read from LDS feeds two MFMAs;
//   * epilogue in the accumulator layout (bias + shortcut + ReLU, one rounding) into the staging tile, then rows of 256 B out.
// Same MFMA instruction, same K order and same epilogue arithmetic as conv_igemm_kernel: results are bit-identical.
// LDS rows are XOR-swizzled by (row & 15) on 16-byte chunks, applied on the SOURCE side of the LDS-DMA (the DMA writes linearly).
#include "conv_args.h"

namespace sodconv {
namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

template <int N>
__device__ __forceinline__ void pw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// byte offset of a 16-byte piece (pixel row `row` of pixel tile t, `coff` bytes into the slice's 256-byte channel run) in the (P, Nout)
// bf16 tensors, or the out-of-range offset for dead tiles / rows.  (A free function: a lambda called from inside the staging lambdas made
// hipcc drop the kernel's HOST stub without a diagnostic.)
__device__ __forceinline__ uint32_t pw_row_off(int t, int ntiles, uint32_t pt, uint32_t P, uint32_t Nout, uint32_t q0, uint32_t row, uint32_t coff) {
  const uint32_t p = (uint32_t)t * pt + row;
  return (t < ntiles && p < P) ? (p * Nout + q0) * 2u + coff : SOD_OOB;
}

}  // namespace

// CK = C / 32 (4, 8, 16); KS = 128-channel slices of the output per workgroup (its channel range QW = 128 * KS); PT = pixels per tile;
// NX = slots of the X ring.  (External linkage on purpose, and fixed-size index arrays below: hipcc 7.2 silently dropped this kernel's
// HOST stub and registration when it sat in the anonymous namespace with a dependent-size array captured by the staging lambdas.)
template <int MODE, int CK, int KS, int PT, int NX>
__global__ __launch_bounds__(512, 1) void conv_pw_kernel(const ConvArgs a, const int tiles_per_wg_stride) {
  constexpr int C = CK * 32, QW = 128 * KS;
  constexpr int XROW = C * 2;                   // bytes per pixel row of the X tile
  constexpr int XT = PT * XROW;                 // X tile bytes (8 or 32 KB)
  constexpr int LX = XT / 8192;                 // LDS-DMA loads per thread of an X tile
  constexpr int RROW = QW * 2;                  // bytes per pixel row of the shortcut / staging tile
  constexpr int RT = PT * RROW;
  constexpr int LS = RT / 8192;                 // 16-byte pieces per thread of that tile
  constexpr int BROW = QW / 8;                  // mask-bit bytes per pixel
  constexpr int BT = PT * BROW < 1024 ? 1024 : PT * BROW;   // bit tile (at least one wave instruction's 1 KB)
  constexpr int P_R = NX * XT, P_B = P_R + 2 * RT, P_DUMMY = P_B + 2 * BT;
  constexpr int FQ = 2 * KS, FP = PT / 32;      // MFMA tiles per wave: QW/4 channels x PT/2 pixels
  constexpr int NST = 2 * LS;                   // vector-memory stores per thread and tile: LS x 16 B + LS bit bytes (dead ones included)
  constexpr int WAITN = (NX == 3 ? LX : 0) + NST;   // operations that may stay in flight at the top of a tile
  static_assert(XT % 8192 == 0 && RT % 8192 == 0 && LX <= 4 && LS <= 4 && BT % 1024 == 0 && PT * BROW <= 8192, "tile geometry");
  static_assert(NX == 2 || NX == 3, "X ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wave >> 1, wp = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int Nout = a.Nout, P = a.lev[0].P;
  const int nq = Nout / QW;
  // workgroup -> (XCD, slot, channel range): the nq workgroups of one pixel-tile stream are neighbours on one XCD
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int qt = local % nq, slot = local / nq;
  const int S = (int)(gridDim.x >> 3) / nq;                 // streams per XCD
  const int stream0 = xcd * S + slot;                       // first pixel tile of this workgroup; stride = 8 * S
  const int stride_t = tiles_per_wg_stride;
  const int ntiles = (P + PT - 1) / PT;
  const bool rev = (a.flags & F_REVERSE) != 0;
  const int q0 = qt * QW;
  const LevelGeo& g = a.lev[0];

  auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.src), 0, g.src_bytes, 0x00020000);
  const uint32_t obytes = (uint32_t)P * (uint32_t)Nout * 2u;
  auto orsrc = __builtin_amdgcn_make_buffer_rsrc(g.dst, 0, obytes, 0x00020000);
  const bool has_res = (a.flags & F_RES) != 0;
  // backward with a bf16 ReLU MASK TENSOR of dx's shape (conv3's data gradient: dx = b > 0 ? dy W : 0) and no accumulate operand: the mask
  // tile travels in the shortcut slot and is applied where the shortcut would be added
  const bool has_mask = (MODE == MODE_DGRAD) && (a.flags & F_MASK) != 0;
  const void* rsrc_ptr = has_res ? g.res : (has_mask ? g.mask : (const void*)g.dst);
  auto rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(rsrc_ptr), 0, (has_res || has_mask) ? obytes : 0u, 0x00020000);
  const bool has_bits = (MODE == MODE_FWD) ? (a.flags & F_WBITS) != 0 : (a.flags & F_MASKBITS) != 0;
  void* bits_ptr = (MODE == MODE_FWD) ? g.bits : (has_mask ? g.dst : const_cast<void*>(g.mask));
  auto brsrc = __builtin_amdgcn_make_buffer_rsrc(has_bits ? bits_ptr : g.dst, 0, has_bits ? (obytes >> 4) : 0u, 0x00020000);

  // ---- weights of this channel range -> registers (A operands), once.  a.w is [Nout][C] row-major.
  bf16x8_t af[FQ][CK];
  {
    const __bf16* wbase = (const __bf16*)a.w + (size_t)(q0 + wq * (QW / 4) + fr) * C + fg * 8;
#pragma unroll
    for (int i = 0; i < FQ; ++i)
#pragma unroll
      for (int kb = 0; kb < CK; ++kb) af[i][kb] = *reinterpret_cast<const bf16x8_t*>(wbase + (size_t)i * 16 * C + kb * 32);
  }
  f32x4_t bv[FQ];
#pragma unroll
  for (int i = 0; i < FQ; ++i) {
    bv[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (a.flags & F_BIAS) bv[i] = *reinterpret_cast<const f32x4_t*>(a.bias + q0 + wq * (QW / 4) + i * 16 + fg * 4);
  }

  // ---- per-thread constants of the LDS-DMA / row-layout mapping: piece k of a thread covers LDS bytes (k * 8 + wave) * 1024 + lane * 16
  uint32_t x_row[4], x_coff[4];          // X tile: pixel row inside the tile, byte offset of the LOGICAL chunk inside the pixel's row
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t o = (uint32_t)((k * 8 + wave) * 1024 + lane * 16);
    const uint32_t row = o / XROW, phys = (o % XROW) >> 4;
    x_row[k] = row;
    x_coff[k] = ((phys ^ (row & 15u)) << 4);
  }
  uint32_t r_row[4], r_coff[4];          // staging tile: pixel row, byte offset (in the QW-channel run of the range) of the logical chunk
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t o = (uint32_t)((k * 8 + wave) * 1024 + lane * 16);
    const uint32_t row = o / RROW, phys = (o % RROW) >> 4;
    r_row[k] = row;
    r_coff[k] = ((phys ^ (row & 15u)) << 4);
  }

  auto tile_of = [&](int it) -> int {      // pixel tile of iteration it (>= ntiles: dead)
    const int t = stream0 + it * stride_t;
    return (rev && t < ntiles) ? ntiles - 1 - t : t;
  };
  auto issue_x = [&](int it) {
    const int t = tile_of(it);
    char* dst = smem + (it % NX) * XT;
    const uint32_t p0 = (uint32_t)t * PT;
#pragma unroll
    for (int k = 0; k < LX; ++k) {
      const uint32_t p = p0 + x_row[k];
      const uint32_t voff = (t < ntiles && p < (uint32_t)P) ? p * (uint32_t)XROW + x_coff[k] : SOD_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, SOD_LDS(dst + (k * 8 + wave) * 1024), 16, voff, 0, 0, 0);
    }
  };
  auto issue_res = [&](int it) {           // shortcut / accumulate tile and (dgrad) the mask bits of iteration it
    const int t = tile_of(it);
    char* dst = smem + P_R + (it & 1) * RT;
#pragma unroll
    for (int k = 0; k < LS; ++k)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rrsrc, SOD_LDS(dst + (k * 8 + wave) * 1024), 16,
                                               pw_row_off(t, ntiles, PT, (uint32_t)P, (uint32_t)Nout, (uint32_t)q0, r_row[k], r_coff[k]), 0, 0, 0);
    if constexpr (MODE == MODE_DGRAD) {
      // bits of the tile: BROW bytes per pixel, contiguous per pixel at (p * Nout + q0) / 8; 16-byte piece `piece` = (pixel, part)
      const int piece = wave * 64 + lane;
      const bool live = piece * 16 < PT * BROW;
      const uint32_t p = (uint32_t)t * PT + (uint32_t)(piece / KS);
      const uint32_t voff = (live && t < ntiles && p < (uint32_t)P) ? ((p * (uint32_t)Nout + (uint32_t)q0) >> 3) + (uint32_t)(piece % KS) * 16u : SOD_OOB;
      char* bdst = (wave * 1024 < PT * BROW) ? smem + P_B + (it & 1) * BT + wave * 1024 : smem + P_DUMMY;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(brsrc, SOD_LDS(bdst), 16, voff, 0, 0, 0);
    }
  };

  // ---- prologue
  issue_res(0);
  issue_x(0);
  if constexpr (NX == 3) issue_x(1);
  pw_wait_vm<(NX == 3 ? LX : 0)>();

  const uint32_t t_sw = (uint32_t)(fg ^ fr);            // X fragment swizzle: physical chunk = (kb * 4 + fg) ^ fr = (kb << 2) ^ t_sw
  for (int it = 0; stream0 + it * stride_t < ntiles; ++it) {
    const int t = tile_of(it);
    // ---- everything of tile `it` has landed (the younger X tile and the previous tile's stores may still be in flight)
    pw_wait_vm<WAITN>();
    __builtin_amdgcn_s_barrier();
    issue_res(it + 1);
    issue_x(it + NX - 1);

    // ---- K loop: acc[i][j] = sum over kb of A(i, kb) x B(j, kb)
    f32x4_t acc[FQ][FP];
#pragma unroll
    for (int i = 0; i < FQ; ++i)
#pragma unroll
      for (int j = 0; j < FP; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const char* xb = smem + (it % NX) * XT + (wp * (PT / 2) + fr) * XROW;
    // B fragments of K block kb + 1 are requested before the FQ * FP independent MFMAs of block kb issue (the compiler otherwise
    // emits read -> wait -> MFMA one by one)
    bf16x8_t bcur[FP], bnxt[FP];
#pragma unroll
    for (int j = 0; j < FP; ++j) bcur[j] = *reinterpret_cast<const bf16x8_t*>(xb + j * 16 * XROW + (t_sw << 4));
#pragma unroll
    for (int kb = 0; kb < CK; ++kb) {
      if (kb + 1 < CK) {
#pragma unroll
        for (int j = 0; j < FP; ++j) bnxt[j] = *reinterpret_cast<const bf16x8_t*>(xb + j * 16 * XROW + ((((uint32_t)(kb + 1) << 2) ^ t_sw) << 4));
      }
#pragma unroll
      for (int j = 0; j < FP; ++j)
#pragma unroll
        for (int i = 0; i < FQ; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][kb], bcur[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < FP; ++j) bcur[j] = bnxt[j];
    }

    // ---- epilogue 1, accumulator layout: + bias + shortcut, ReLU, one rounding, into the staging tile (in place of the shortcut)
    char* stg = smem + P_R + (it & 1) * RT;
#pragma unroll
    for (int j = 0; j < FP; ++j) {
      const int r = wp * (PT / 2) + j * 16 + fr;
#pragma unroll
      for (int i = 0; i < FQ; ++i) {
        const int chunk = wq * (4 * KS) + i * 2 + (fg >> 1);
        char* p8 = stg + r * RROW + ((chunk ^ fr) << 4) + (fg & 1) * 8;
        const u32x2_t rv = *reinterpret_cast<const u32x2_t*>(p8);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e];
        if (a.flags & F_BIAS) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bv[i][e];
        }
        if (has_res) {
          v[0] += __uint_as_float(rv[0] << 16); v[1] += __uint_as_float(rv[0] & 0xffff0000u);
          v[2] += __uint_as_float(rv[1] << 16); v[3] += __uint_as_float(rv[1] & 0xffff0000u);
        }
        if (a.flags & F_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (has_mask) {
          v[0] = (__uint_as_float(rv[0] << 16) > 0.f) ? v[0] : 0.f; v[1] = (__uint_as_float(rv[0] & 0xffff0000u) > 0.f) ? v[1] : 0.f;
          v[2] = (__uint_as_float(rv[1] << 16) > 0.f) ? v[2] : 0.f; v[3] = (__uint_as_float(rv[1] & 0xffff0000u) > 0.f) ? v[3] : 0.f;
        }
        bf16x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
        *reinterpret_cast<bf16x4_t*>(p8) = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- epilogue 2, row layout: 16 bytes = 8 channels of one pixel per lane; a wave instruction stores 1 KB of full channel runs
    const char* bitb = smem + P_B + (it & 1) * BT;
#pragma unroll
    for (int k = 0; k < LS; ++k) {
      const uint32_t o = (uint32_t)((k * 8 + wave) * 1024 + lane * 16);
      u32x4_t v = *reinterpret_cast<const u32x4_t*>(stg + o);
      const uint32_t goff = pw_row_off(t, ntiles, PT, (uint32_t)P, (uint32_t)Nout, (uint32_t)q0, r_row[k], r_coff[k]);
      if constexpr (MODE == MODE_DGRAD) {
        uint32_t m = has_bits ? (uint32_t)(*reinterpret_cast<const uint8_t*>(bitb + r_row[k] * BROW + (r_coff[k] >> 4))) : 0xffu;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t lo = (m >> (2 * e)) & 1u, hi = (m >> (2 * e + 1)) & 1u;
          v[e] &= (lo ? 0x0000ffffu : 0u) | (hi ? 0xffff0000u : 0u);
        }
        __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, goff, 0, 0);
        // keep the store count of the two modes equal (NST): the bit byte of the forward mode has no counterpart here
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)0, brsrc, SOD_OOB, 0, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, goff, 0, 0);
        uint32_t b = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // stored value > 0: a positive bf16 has a clear sign bit and a non-zero magnitude
          const uint32_t lo = v[e] & 0xffffu, hi = v[e] >> 16;
          b |= ((lo != 0u && lo <= 0x7f80u) ? 1u : 0u) << (2 * e);          // (0, +inf]; NaN compares false as in the tiled kernel
          b |= ((hi != 0u && hi <= 0x7f80u) ? 1u : 0u) << (2 * e + 1);
        }
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)b, brsrc, (has_bits && goff != SOD_OOB) ? (goff >> 4) : SOD_OOB, 0, 0);
      }
    }
  }
  pw_wait_vm<0>();      // dead prefetches must have landed before the LDS allocation goes back
}

namespace {

template <int MODE, int CK, int KS, int PT, int NX>
int launch_pw_one(const ConvArgs& a, hipStream_t st) {
  constexpr int C = CK * 32, QW = 128 * KS;
  constexpr int lds = NX * PT * C * 2 + 2 * PT * QW * 2 + 2 * (PT * QW / 8 < 1024 ? 1024 : PT * QW / 8) + 1024;
  auto kern = conv_pw_kernel<MODE, CK, KS, PT, NX>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  const int nq = a.Nout / QW;
  const int grid = 256;                         // one workgroup per CU (pw_supported checks the device)
  const int stride = grid / nq;                 // pixel-tile streams
  SOD_LAUNCH(kern, dim3(grid), dim3(512), lds, st, a, stride);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

// channel range of a workgroup per contraction width: the weights of the range must fit the register file (C * QW * 2 B over 512 lanes)
constexpr int pw_qw(int C) { return C == 128 ? 512 : (C == 256 ? 256 : 128); }

}  // namespace

// Shapes and epilogues the persistent kernel takes: 1x1, stride 1, no padding, one dense level, C in {128, 256, 512}, Nout a multiple of
// the workgroup's channel range (512 / 256 / 128 channels for C = 128 / 256 / 512: 128 KB of weights in registers) with 1 ... 32 ranges:
// the EXPANDING convolutions of the bottleneck blocks, and the contracting 512 -> 128 ones of res3 (one range: the wide tensor is read
// exactly once).  bf16 output; forward: bias / shortcut / ReLU / bit mask; backward: accumulate / bit mask, or a bf16 mask tensor without
// an accumulate operand.  256 CUs (the grid is the chip).
bool pw_supported(const ConvArgs& a, int mode, bool out_f32, int cus) {
  if (out_f32 || cus != 256 || a.nlev != 1 || a.cwin) return false;
  if (a.R != 1 || a.S != 1 || a.stride != 1 || a.pad != 0) return false;
  if (!(a.Cred == 128 || a.Cred == 256 || a.Cred == 512) || a.Cpitch != a.Cred) return false;
  const int qw = pw_qw(a.Cred);
  const int nq = a.Nout / qw;
  if ((a.Nout % qw) || nq < 1 || nq > 32 || (32 % nq)) return false;
  const LevelGeo& g = a.lev[0];
  if (g.pstart != 0 || g.Hs != g.Hp || g.Ws != g.Wp) return false;
  if (g.src_img_stride != g.Hs * g.Ws * a.Cred || g.dst_img_stride != g.Hp * g.Wp * a.Nout) return false;
  if ((long long)g.P * a.Nout * 2 >= (1ll << 31)) return false;
  const int allowed = (mode == MODE_FWD) ? (F_BIAS | F_RELU | F_RES | F_WBITS | F_REVERSE) : (F_RES | F_MASKBITS | F_MASK | F_REVERSE);
  if (a.flags & ~allowed) return false;
  if ((a.flags & F_MASK) && (a.flags & (F_RES | F_MASKBITS))) return false;      // the mask tensor travels in the shortcut slot
  if ((a.flags & F_RES) && g.res_img_stride != g.dst_img_stride) return false;
  if (g.P < 16384) return false;                // a launch this small does not fill the persistent grid: the tiled kernel is as good
  return true;
}

int launch_pw(const ConvArgs& a, int mode, hipStream_t st) {
  if (mode == MODE_FWD) {
    if (a.Cred == 128) return launch_pw_one<MODE_FWD, 4, 4, 32, 3>(a, st);
    if (a.Cred == 256) return launch_pw_one<MODE_FWD, 8, 2, 64, 2>(a, st);
    return launch_pw_one<MODE_FWD, 16, 1, 32, 3>(a, st);
  }
  if (a.Cred == 128) return launch_pw_one<MODE_DGRAD, 4, 4, 32, 3>(a, st);
  if (a.Cred == 256) return launch_pw_one<MODE_DGRAD, 8, 2, 64, 2>(a, st);
  return launch_pw_one<MODE_DGRAD, 16, 1, 32, 3>(a, st);
}

}  // namespace sodconv
