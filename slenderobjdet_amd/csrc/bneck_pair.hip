// Two chained 1x1 convolutions of the ResNet body in ONE kernel: EXPAND (narrow -> wide, + add operand, + nonlinearity, wide tensor
// stored) followed by CONTRACT (wide -> narrow) on the tile that has just been produced (gfx950).
//
//   forward  (mode 0):  y  = relu(conv3_i(h2) + b3 + x)            (block i's last conv + residual; 1-bit mask of y > 0 recorded)
//                       h1 = relu(conv1_{i+1}(y) + b1)             (the NEXT block's first conv)
//   backward (mode 1):  g  = bits_{k-1} ? conv1_k^T(da) + g_skip : 0   (data gradient of block k's first conv + identity-path gradient,
//                                                                       masked with block k-1's output ReLU bits)
//                       db = (b > 0) ? conv3_{k-1}^T(g) : 0            (data gradient of block k-1's last conv, masked with its input's ReLU)
//   (detectron2 BottleneckBlock chains under slender_det/modeling/backbone/fpn.py:94-115)
//
// Why: the un-fused pair moves the wide tensor three times (conv3 writes it, conv1 reads it back, plus the add operand), and the round-4
// ablations show these kernels are the SUM of their memory phases - the matrix cores contribute nothing to their
// time.  Here the wide tile never comes back: each 128-channel chunk of it is produced into LDS, stored once, and consumed from LDS as
// the contraction operand of the second GEMM (a rank-128 update of the narrow accumulator held in registers for the whole tile).
//
// Structure: one workgroup of 8 waves per pixel tile of BP <= 144 pixels (tile sizes are chosen on the host so that every CU gets the
// same number of tiles).  LDS: the narrow input tile [BP][CN] (LDS-DMA, XOR-swizzled rows), the wide chunk Y [BP][128], the add-operand
// chunk R [BP][128] (LDS-DMA, requested one chunk ahead).  Waves split the OUTPUT ROWS: wave w owns rows w*16..+15 of the 128-row chunk
// in GEMM 1 and rows w*CN/8..+CN/8-1 of the narrow output in GEMM 2, for ALL pixels of the tile, so every weight byte is fetched by
// exactly one wave - straight from L2 into MFMA A-operand registers (global_load_dwordx4 in fragment layout, no LDS staging); the
// pixel operands are ds_read_b128 fragments shared by the eight waves.  Results are bit-identical to the two-launch path: same MFMA
// instruction, same k order, same epilogue arithmetic (tests/test_gpu_bottleneck.py).
//
// STATUS (round 4): correct and tested, OFF by default (SOD_PAIR_FWD / SOD_PAIR_BWD, modeling/backbone/resnet.py).  Stand-alone on
// operands from HBM the res3 pair is faster (forward 258 -> 215 us, backward 244 -> 228 us), the res4 pair slower (155 -> 165 us:
// 96-pixel tiles stream every weight once per tile), and in the training step neither moves img/s: a workgroup claims a whole CU
// (155 KB LDS), so nothing of the other streams runs beside it, the un-fused conv1 finds the wide tensor in the Infinity Cache,
// and a tile's life is a drained pipeline at both ends (ablation: with both GEMMs, the stores and epilogue 1 removed the res3 launch
// still takes 149 of 228 us).  What it needs to pay is in DESIGN.md section 6: a persistent workgroup with the next tile's narrow
// input and add chunks in flight across tile boundaries.
#include "conv_args.h"
#include <stdlib.h>

namespace sodconv {
namespace {

// pixel blocks of 16 per tile: at most 9 (144 pixels) with 128 narrow channels, 8 with 256 (the narrow accumulator is CN x BP fp32 in
// the registers of eight waves)
constexpr int pnb_of(int cn) { return cn == 128 ? 9 : 8; }

struct PairArgs {
  const __bf16* xin;       // [P][CN]
  const __bf16* add;       // [P][CW] or null
  const __bf16* we;        // [CW][CN]   expand weights, row = wide channel
  const float* bias_e;     // [CW] or null
  const __bf16* wc;        // [CN][CW]   contract weights, row = narrow output channel
  const float* bias_c;     // [CN] or null
  const uint8_t* bits_in;  // mode 1: [P * CW / 8]
  const __bf16* mask2;     // mode 1: [P][CN]
  __bf16* wide;            // [P][CW]
  uint8_t* bits_out;       // mode 0: [P * CW / 8] or null
  __bf16* narrow;          // [P][CN]
  int P, CW;
  int n_big, bp_big, bp_small;
  uint32_t xin_bytes, add_bytes;
};

// LDS reads as inline asm: hipcc puts an s_waitcnt vmcnt(0) in front of every LDS read it knows of while an LDS-DMA load is
// outstanding (it cannot tell the LDS ranges apart), and the add-operand chunk of the NEXT step is in flight during the whole compute
// phase.  The reader waits with lds_wait<N> (counted lgkmcnt, tied to the register so that no use can be scheduled above it).
template <int OFF = 0>
__device__ __forceinline__ bf16x8_t lds_read128(uint32_t addr) {      // OFF: the instruction's immediate offset (< 65536)
  bf16x8_t r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
template <int N>
__device__ __forceinline__ void lds_wait(bf16x8_t& r) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r) : "n"(N));
}
// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <int I>
struct IC { static constexpr int value = I; };
template <int N>
struct Unroll {
  template <class F>
  static __device__ __forceinline__ void run(F&& f) { Unroll<N - 1>::run(f); f(IC<N - 1>{}); }
};
template <>
struct Unroll<0> {
  template <class F>
  static __device__ __forceinline__ void run(F&&) {}
};

template <int CN, int MODE>
__global__ __launch_bounds__(512, 2) void bneck_pair_kernel(const PairArgs a) {
  constexpr int PNB = pnb_of(CN), PBPMAX = PNB * 16;
  constexpr int K1 = CN / 32;              // k-steps of GEMM 1 (contraction over the narrow channels)
  constexpr int R2 = CN / 128;             // 16-row blocks of the narrow output per wave (8 waves x R2 x 16 = CN rows)
  constexpr int XROW = CN * 2;             // bytes per pixel row of the narrow tile
  constexpr int XBYTES = PBPMAX * XROW;
  constexpr int YBYTES = PBPMAX * 256;
  constexpr int RINST = (PBPMAX * 256 + 8191) / 8192;      // LDS-DMA instructions per wave and add chunk (fixed count: 5 or 4)
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* const XN = smem;                   // [BP][CN] bf16, 16-B chunks XOR-swizzled with (row & 15); every region is 1-KB aligned
  char* const Y = smem + XBYTES;           // [BP][128]
  constexpr int RBYTES = 8 * RINST * 1024; // one add-operand chunk image (whole LDS-DMA instructions)
  char* const RR = Y + YBYTES;             // 2 x [BP][128]: chunk j in buffer j & 1, requested two steps ahead
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;

  const int t = blockIdx.x;
  int p0, bp;
  if (t < a.n_big) { p0 = t * a.bp_big; bp = a.bp_big; }
  else { p0 = a.n_big * a.bp_big + (t - a.n_big) * a.bp_small; bp = a.bp_small; }
  if (p0 + bp > a.P) bp = a.P - p0;                        // ragged last tile
  const int nb = (bp + 15) >> 4;                           // pixel blocks in use (wave-uniform)
  const int J = a.CW >> 7;                                 // 128-channel chunks of the wide tensor

  auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.xin), 0, a.xin_bytes, 0x00020000);
  auto rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.add ? a.add : a.xin), 0, a.add ? a.add_bytes : 0u, 0x00020000);

  // ---- narrow input tile: LDS image is linear per wave instruction (1 KB), the SOURCE chunk is swizzled
  {
    constexpr int CPR = XROW / 16;                         // 16-B chunks per row (16 or 32)
    const int ninst = (nb * 16 * XROW) >> 10;
    for (int i = wave; i < ninst; i += 8) {
      const int lin = i * 64 + lane;                       // 16-B slot index in the tile image
      const int row = lin / CPR, cp = lin % CPR;
      const int chunk = cp ^ (row & 15);
      const uint32_t off = (p0 + row < a.P) ? (uint32_t)(((size_t)(p0 + row) * CN + chunk * 8) * 2) : SOD_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, SOD_LDS(XN + i * 1024), 16, off, 0, 0, 0);
    }
  }
  // Mode 1 with room behind the add rows (CN = 128: a 40-KB image for 36 KB of rows): the ReLU bits of the chunk - 16 B per pixel -
  // ride along in the LAST LDS-DMA instruction slots of the image (rows >= PBPMAX of the "add" image are bit rows: slot s of the tail
  // holds the 16 bytes of pixel s).  One request per pixel for all eight waves instead of a byte load per lane and pixel block.
  constexpr bool BITS_LDS = MODE == 1 && (RBYTES - PBPMAX * 256) >= PBPMAX * 16;
  auto bitsrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.bits_in ? a.bits_in : (const uint8_t*)a.xin), 0,
                                                    a.bits_in ? a.add_bytes >> 4 : 0u, 0x00020000);
  auto stage_add = [&](int j) {                            // add-operand chunk j -> RR: exactly RINST loads per thread (rows beyond the
    if (!a.add) return;                                    // tile / the tensor: out-of-range offsets, zero fill into rows nobody uses)
#pragma unroll
    for (int u = 0; u < RINST; ++u) {
      const int i = wave + 8 * u;
      const int lin = i * 64 + lane;
      const int row = lin >> 4, cp = lin & 15;
      const int chunk = cp ^ (row & 15);
      const uint32_t off = (row < bp) ? (uint32_t)(((size_t)(p0 + row) * a.CW + j * 128 + chunk * 8) * 2) : SOD_OOB;
      if (BITS_LDS && i * 64 >= PBPMAX * 16) {             // (wave-uniform) an instruction wholly behind the rows: the pixels' bit rows
        const int px = lin - PBPMAX * 16;
        const uint32_t boff = (px < bp) ? (uint32_t)((((size_t)(p0 + px) * a.CW + j * 128) >> 3)) : SOD_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(bitsrsrc, SOD_LDS(RR + (j & 1) * RBYTES + i * 1024), 16, boff, 0, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rrsrc, SOD_LDS(RR + (j & 1) * RBYTES + i * 1024), 16, off, 0, 0, 0);
      }
    }
  };
  stage_add(0);

  // ---- weight fragments straight from global memory in MFMA A-operand layout: lane (fr, fg) holds row fr, k = 8 fg .. 8 fg + 7
  bf16x8_t a1[K1];
  auto load_a1 = [&](int j) {
    const __bf16* src = a.we + (size_t)(j * 128 + wave * 16 + fr) * CN + fg * 8;
#pragma unroll
    for (int ks = 0; ks < K1; ++ks) a1[ks] = *reinterpret_cast<const bf16x8_t*>(src + ks * 32);
  };
  bf16x8_t a2[R2][4];
  auto load_a2 = [&](int j) {
#pragma unroll
    for (int rb = 0; rb < R2; ++rb) {
      const __bf16* src = a.wc + (size_t)((wave * R2 + rb) * 16 + fr) * a.CW + j * 128 + fg * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a2[rb][ks] = *reinterpret_cast<const bf16x8_t*>(src + ks * 32);
    }
  };
  load_a1(0);

  const uint32_t xn0 = (uint32_t)(uintptr_t)SOD_LDS(XN), y0 = (uint32_t)(uintptr_t)SOD_LDS(Y), r0 = (uint32_t)(uintptr_t)SOD_LDS(RR);
  f32x4_t acc1[PNB], acc2[R2][PNB];
#pragma unroll
  for (int n = 0; n < PNB; ++n) {
    acc1[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rb = 0; rb < R2; ++rb) acc2[rb][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  // Both GEMMs run over all PNB pixel blocks (compile-time trip counts and wait counts); blocks beyond the tile read stale LDS
  // bytes and their columns are never stored.  Fragment addresses: ONE register per operand tile - row fr of pixel block 0, logical
  // chunk fg, swizzled with the row (chunk ^ fr) - the k-step moves the chunk by 4 (an XOR of byte-address bit 6 upwards: the tiles
  // are 1-KB aligned and a row is 256 / 512 B) and the pixel block is the instruction's immediate offset.
  static_assert((PNB - 1) * 16 * XROW < 65536, "immediate offset of ds_read_b128");
  const uint32_t xfrag = xn0 + (uint32_t)(fr * XROW) + (uint32_t)((fg ^ fr) << 4);
  const uint32_t yfrag = y0 + (uint32_t)(fr * 256) + (uint32_t)((fg ^ fr) << 4);
  auto gemm1 = [&]() {                                     // acc1[n] = We[chunk rows of this wave] x XN^T
#pragma unroll
    for (int n = 0; n < PNB; ++n) acc1[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < K1; ++ks) {
      bf16x8_t b[PNB];
      const uint32_t ad = xfrag ^ (uint32_t)(ks << 6);
      Unroll<PNB>::run([&](auto ic) { constexpr int n = decltype(ic)::value; b[n] = lds_read128<n * 16 * XROW>(ad); });
      Unroll<PNB>::run([&](auto ic) {
        constexpr int n = decltype(ic)::value;
        lds_wait<PNB - 1 - n>(b[n]);
        acc1[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[ks], b[n], acc1[n], 0, 0, 0);
      });
    }
  };
  auto gemm2 = [&]() {                                     // acc2[rb][n] += Wc[rows of this wave][chunk] x Y^T
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8_t b[PNB];
      const uint32_t ad = yfrag ^ (uint32_t)(ks << 6);
      Unroll<PNB>::run([&](auto ic) { constexpr int n = decltype(ic)::value; b[n] = lds_read128<n * 16 * 256>(ad); });
      Unroll<PNB>::run([&](auto ic) {
        constexpr int n = decltype(ic)::value;
        lds_wait<PNB - 1 - n>(b[n]);
#pragma unroll
        for (int rb = 0; rb < R2; ++rb) acc2[rb][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[rb][ks], b[n], acc2[rb][n], 0, 0, 0);
      });
    }
  };

  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
  auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(a.wide, 0, a.add_bytes, 0x00020000);                    // P * CW * 2 bytes
  auto brsrc = __builtin_amdgcn_make_buffer_rsrc(a.bits_out ? a.bits_out : (uint8_t*)a.wide, 0, a.bits_out ? a.add_bytes >> 4 : 0u, 0x00020000);
  constexpr int ST = (PBPMAX * 256 + 8191) / 8192;         // 16-B slots of the wide chunk per thread (fixed trip count: 4 or 5)
  // per-step scalars of epilogue 1, requested a step ahead like everything else: the expand bias of the lane's 4 rows (mode 0) or
  // the ReLU bits of its (pixel, 4 rows) cells (mode 1: one byte per pixel block)
  const int q = wave * 16 + fg * 4;                        // first of this lane's 4 rows inside a chunk
  f32x4_t bias4 = {0.f, 0.f, 0.f, 0.f};
  uint32_t mbits[PNB];
  auto load_step_scalars = [&](int j) {
    if (MODE == 0) {
      if (a.bias_e) bias4 = *reinterpret_cast<const f32x4_t*>(a.bias_e + j * 128 + q);
    } else if (!BITS_LDS) {
#pragma unroll
      for (int n = 0; n < PNB; ++n) {
        const int p = p0 + n * 16 + fr;
        mbits[n] = (n * 16 + fr < bp) ? (uint32_t)a.bits_in[((size_t)p * a.CW + j * 128 + q) >> 3] : 0u;
      }
    }
  };
  constexpr int NSCAL = MODE == 0 ? 1 : (BITS_LDS ? 0 : PNB);               // VMEM operations of load_step_scalars (mode 0 without bias: none - see below)

  // VMEM operations a thread issues AFTER the add-chunk DMA it waits for at the end of a step: the wide stores (+ bit stores in mode
  // 0), the next step's contract weights, the expand weights and scalars of the step after, and the add chunk two steps ahead.
  const bool has_bias = MODE == 0 && a.bias_e;             // (uniform) decides between two wait constants
  constexpr int YOUNGER = ST * (MODE == 0 ? 2 : 1) + R2 * 4 + K1 + RINST;

  load_a2(0);
  load_step_scalars(0);
  if (J > 1) stage_add(1); else stage_add(0);              // (a one-chunk problem re-requests chunk 0 into the other buffer: fixed counts)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // prologue: everything has landed
  __syncthreads();
  gemm1();
  load_a1(J > 1 ? 1 : 0);

  for (int j = 0; j < J; ++j) {
    // ---- epilogue 1: the wave's 16 rows x all pixels of chunk j, D layout (lane: rows 4 fg .. 4 fg + 3 of pixel column fr) -> Y
    const uint32_t rbuf = r0 + (uint32_t)((j & 1) * RBYTES);
    const uint32_t sw0 = (uint32_t)(fr * 256) + (uint32_t)((((q >> 3) ^ fr) << 4) + (q & 4) * 2);
#pragma unroll
    for (int n = 0; n < PNB; ++n) {
      const uint32_t sw = sw0 + (uint32_t)(n * 4096);
      float v[4] = {acc1[n][0], acc1[n][1], acc1[n][2], acc1[n][3]};
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bias4[e];
      }
      if (a.add) {
        const bf16x4_t r4 = *(__attribute__((address_space(3))) const bf16x4_t*)(uintptr_t)(rbuf + sw);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)r4[e];
      }
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      } else {
        uint32_t mb;
        if (BITS_LDS) mb = *(__attribute__((address_space(3))) const uint8_t*)(uintptr_t)(rbuf + PBPMAX * 256 + (n * 16 + fr) * 16 + (q >> 3));
        else mb = mbits[n];
        mb >>= (q & 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ((mb >> e) & 1u) ? v[e] : 0.f;
      }
      bf16x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
      *(__attribute__((address_space(3))) bf16x4_t*)(uintptr_t)(y0 + sw) = o;
    }
    const int j1 = j + 1 < J ? j + 1 : j, j2 = j + 2 < J ? j + 2 : j1;      // (the last steps re-request operands: fixed operation counts)
    __syncthreads();                                       // Y complete, add buffer j & 1 consumed
    stage_add(j2);                                         // -> buffer (j + 2) & 1 = the one just consumed   [RINST operations]
    // ---- the wide chunk leaves through coalesced 16-B stores: 16 lanes = one pixel's 256 contiguous bytes.  Every thread issues
    // exactly ST stores (+ ST bit-mask bytes): rows beyond the tile take an out-of-range offset, which the hardware drops.
    {
      bf16x8_t o[ST];
      const uint32_t ya = y0 + (uint32_t)(tid * 16);
      Unroll<ST>::run([&](auto ic) { constexpr int it = decltype(ic)::value; o[it] = lds_read128<it * 8192>(ya); });
      Unroll<ST>::run([&](auto ic) {
        constexpr int it = decltype(ic)::value;
        lds_wait<ST - 1 - it>(o[it]);
        const int sl = tid + it * 512;
        const int row = sl >> 4, cp = sl & 15;
        const int chunk = cp ^ (row & 15);
        const bool ok = row < bp;
        const uint32_t e0 = (uint32_t)(p0 + row) * (uint32_t)a.CW + (uint32_t)(j * 128 + chunk * 8);        // < 2^30 elements
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o[it]), wrsrc, ok ? e0 * 2u : SOD_OOB, 0, 0);
        if (MODE == 0) {
          uint32_t b = 0;
#pragma unroll
          for (int e = 0; e < 8; ++e) b |= ((float)o[it][e] > 0.f ? 1u : 0u) << e;
          __builtin_amdgcn_raw_buffer_store_b8((unsigned char)b, brsrc, ok ? e0 >> 3 : SOD_OOB, 0, 0);
        }
      });
    }
    gemm2();
    load_a2(j1);                                           // a2 is free: the next step's contract columns            [R2 * 4]
    if (j + 1 < J) gemm1();                                // (uses a1(j + 1), requested a whole step ago)
    load_a1(j2);                                           // a1 is free: the expand rows of the step after next      [K1]
    load_step_scalars(j1);                                 //                                                          [NSCAL or 0]
    // the add chunk of step j + 1 (requested a step ago) has landed; everything issued after it stays in flight
    if (MODE == 0 && !has_bias) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER + NSCAL) : "memory");
    __syncthreads();                                       // every wave has finished reading Y
  }

  // ---- epilogue 2: narrow output through the (dead) XN region, then coalesced stores
  {
    float bc[R2][4];
#pragma unroll
    for (int rb = 0; rb < R2; ++rb)
#pragma unroll
      for (int e = 0; e < 4; ++e) bc[rb][e] = (MODE == 0 && a.bias_c) ? a.bias_c[(wave * R2 + rb) * 16 + fg * 4 + e] : 0.f;
#pragma unroll
    for (int rb = 0; rb < R2; ++rb) {
      const int q2 = (wave * R2 + rb) * 16 + fg * 4;
#pragma unroll
      for (int n = 0; n < PNB; ++n) {
        const int row = n * 16 + fr;
        bf16x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc2[rb][n][e] + bc[rb][e];
          if (MODE == 0) v = fmaxf(v, 0.f);
          o[e] = (__bf16)v;
        }
        const uint32_t sw = (uint32_t)(row * XROW) + (uint32_t)((((q2 >> 3) ^ fr) << 4) + (q2 & 4) * 2);
        *(__attribute__((address_space(3))) bf16x4_t*)(uintptr_t)(xn0 + sw) = o;
      }
    }
    __syncthreads();
    constexpr int CPR = XROW / 16;
    for (int s = tid; s < nb * 16 * CPR; s += 512) {
      const int row = s / CPR, cp = s % CPR;
      const int p = p0 + row;
      if (p < a.P) {
        bf16x8_t o = lds_read128(xn0 + (uint32_t)(row * XROW + cp * 16));
        lds_wait<0>(o);
        const int chunk = cp ^ (row & 15);
        const size_t e0 = (size_t)p * CN + chunk * 8;
        if (MODE == 1 && a.mask2) {
          const bf16x8_t m = *reinterpret_cast<const bf16x8_t*>(a.mask2 + e0);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = ((float)m[e] > 0.f) ? o[e] : (__bf16)0.f;
        }
        *reinterpret_cast<bf16x8_t*>(a.narrow + e0) = o;
      }
    }
  }
}

template <int CN, int MODE>
int launch_pair(const PairArgs& a, int tiles, hipStream_t st) {
  constexpr int PBPMAX = pnb_of(CN) * 16;
  constexpr int lds = PBPMAX * CN * 2 + PBPMAX * 256 + 2 * 8 * ((PBPMAX * 256 + 8191) / 8192) * 1024;
  auto kern = bneck_pair_kernel<CN, MODE>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  SOD_LAUNCH(kern, dim3(tiles), dim3(512), lds, st, a);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

}  // namespace
}  // namespace sodconv

using namespace sodconv;

// Tile sizes such that the CUs get equal shares: r = rounds needed at the largest tile; cus * r tiles of 16-pixel granularity, the
// first n_big one block larger.  Returns the number of tiles.
static int pair_tiling(long long P, int cus, int PNB, int* n_big, int* bp_big, int* bp_small) {
  const long long blocks = (P + 15) / 16;                  // 16-pixel blocks
  long long r = (blocks + (long long)cus * PNB - 1) / ((long long)cus * PNB);
  if (r < 1) r = 1;
  long long tiles = (long long)cus * r;
  if (tiles > blocks) tiles = blocks;
  const long long small = blocks / tiles, big = blocks - small * tiles;      // `big` tiles carry one block more
  *n_big = (int)big; *bp_big = (int)(small + 1) * 16; *bp_small = (int)small * 16;
  return (int)tiles;
}

extern "C" int sod_bottleneck_pair_supported(int CN, int CW) {
  return ((CN == 128 || CN == 256) && CW >= 128 && (CW & 127) == 0) ? 1 : 0;
}

extern "C" int sod_bottleneck_pair(const void* xin, const void* add, const void* we, const float* bias_e, const void* wc, const float* bias_c,
                                   const void* bits_in, const void* mask2, void* wide, void* bits_out, void* narrow,
                                   long long P, int CN, int CW, int mode, void* stream) {
  if (!xin || !we || !wc || !wide || !narrow || P <= 0 || (mode != 0 && mode != 1)) return SOD_EARG;
  if (!sod_bottleneck_pair_supported(CN, CW)) return SOD_EARG;
  if (mode == 1 && (!bits_in || !add)) return SOD_EARG;      // the backward form always carries the identity-path gradient
  if ((unsigned long long)P * CW * 2ull >= 0x80000000ull) return SOD_ESIZE;
  const uintptr_t al = (uintptr_t)xin | (uintptr_t)add | (uintptr_t)we | (uintptr_t)wc | (uintptr_t)wide | (uintptr_t)narrow | (uintptr_t)mask2 |
                       (uintptr_t)bias_e | (uintptr_t)bias_c;
  if (al & 15) return SOD_EALIGN;
  PairArgs a{};
  a.xin = (const __bf16*)xin; a.add = (const __bf16*)add; a.we = (const __bf16*)we; a.bias_e = bias_e; a.wc = (const __bf16*)wc;
  a.bias_c = bias_c; a.bits_in = (const uint8_t*)bits_in; a.mask2 = (const __bf16*)mask2; a.wide = (__bf16*)wide;
  a.bits_out = (uint8_t*)bits_out; a.narrow = (__bf16*)narrow;
  a.P = (int)P; a.CW = CW;
  a.xin_bytes = (uint32_t)((unsigned long long)P * CN * 2ull); a.add_bytes = (uint32_t)((unsigned long long)P * CW * 2ull);
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  const int tiles = pair_tiling(P, cus, pnb_of(CN), &a.n_big, &a.bp_big, &a.bp_small);
  hipStream_t st = (hipStream_t)stream;
  if (CN == 128) return mode == 0 ? launch_pair<128, 0>(a, tiles, st) : launch_pair<128, 1>(a, tiles, st);
  return mode == 0 ? launch_pair<256, 0>(a, tiles, st) : launch_pair<256, 1>(a, tiles, st);
}
