// 3x3 / stride 1 / pad 1 convolution 128 -> 128 channels as a PERSISTENT, WEIGHT-STATIONARY kernel (gfx950), round 6.
//
//   forward   conv2 of a detectron2 BottleneckBlock of res3: y = relu(x * W + b)              (FrozenBN folded into W and b)
//   backward  its data gradient: dx = mask > 0 ? dy * W^T (taps flipped) : 0                  (mask = conv1's ReLU output)
//   (build_resnet_backbone reached from slender_det/modeling/backbone/fpn.py:103; replaces ATen's conv / conv backward-input)
//
// On the 128 x 128-tile kernel these launches are bound by what a CU can ingest: every workgroup re-stages its 128-pixel tile once per TAP and
// the 128 x 64 weight tile of every K-step once per PIXEL tile - 32 B through L2 -> LDS per thousand MACs, 725 TFLOP/s.  Here the WEIGHTS never
// move: 128 x 9 x 128 bf16 = 295 KB live in the registers of one workgroup per CU for the whole launch (4 waves x 1 per SIMD; wave w owns output
// channels 32 w .. 32 w + 31: 2 x 9 x 4 MFMA A operands = 288 VGPRs per lane), and a pixel tile is staged ONCE for all nine taps: the workgroup
// walks over tiles of 8 rows x 14 columns of output pixels; the 10 x 16 halo window (all 128 channels: 160 LDS rows of 256 B = 40 KB, double
// buffered) arrives by LDS-DMA one tile ahead, out-of-image pixels as out-of-range offsets (hardware zero fill: no border masks).  The compute
// window is the 16 columns of the halo (two of its 16 outputs per row are halo columns and are dropped: 12.5 % of the MFMAs, paid so that the
// window pitch is 16 rows = 4 KB - a tap's row shift is an instruction OFFSET and the XOR swizzle of a pixel's LDS row depends on its column
// only, so the whole K loop runs on 24 base addresses computed once per launch).  B fragment = 16 pixels x 32 channels, one ds_read_b128 per
// lane, feeding two MFMAs (the wave's two 16-channel output tiles); 16-byte chunks XOR-swizzled by (row & 15) on the SOURCE side of the
// LDS-DMA: conflict-free at every tap.  Epilogue in the accumulator layout (bias + ReLU, one rounding) into the consumed halo buffer, then
// 256-byte channel runs out (the data gradient applies the bf16 mask tensor there).  res3 geometry (16 x 100 x 168): 16 x 13 x 12 tiles.
#include "conv_args.h"

namespace sodconv {
namespace {

constexpr int W3_TH = 8, W3_TWC = 16, W3_TW = 14;
constexpr int W3_ROWS = (W3_TH + 2) * W3_TWC;          // 160 halo pixels
constexpr int W3_BUF = W3_ROWS * 256;                  // 40 960 B
constexpr int W3_GUARD = 256;                          // a halo column's neighbour reads one row in front of / behind a buffer
constexpr int W3_LDS = 2 * W3_GUARD + 2 * W3_BUF;      // 82 432 B
constexpr int W3_STEPS = W3_TH * 9 * 4;                // (pixel row, tap, 32-channel block)
// fragment reads in flight ahead of the MFMAs (one wave per SIMD: nothing else hides the LDS round trip); the data gradient carries its eight
// mask loads across the epilogue and has the registers for three only (five spill 45 registers)
template <int MODE> struct W3Dist { static constexpr int v = (MODE == MODE_FWD) ? 5 : 2; };
constexpr int W3_DMAX = 5;

template <int OFF>
__device__ __forceinline__ bf16x8_t w3_read(uint32_t addr) {
  bf16x8_t r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

struct W3State {
  f32x4_t acc[2][W3_TH];
  bf16x8_t af[2][9][4];
  bf16x8_t b[W3_DMAX + 1];
  uint32_t rb[3][4];           // base address of (column shift, channel block) in the CURRENT buffer
};

// step S = (pixel row j, tap, channel block kb); MODE_DGRAD reads the taps flipped
template <int MODE, int S>
__device__ __forceinline__ void w3_issue(W3State& st) {
  if constexpr (S < W3_STEPS) {
    constexpr int j = S / 36, tap = (S % 36) / 4, kb = S % 4;
    constexpr int sg = (MODE == MODE_FWD) ? 1 : -1;
    constexpr int dh = sg * (tap / 3 - 1), dw = sg * (tap % 3 - 1);
    st.b[S % (W3Dist<MODE>::v + 1)] = w3_read<(j + 1 + dh) * W3_TWC * 256>(st.rb[dw + 1][kb]);
  }
}

template <int MODE, int S>
__device__ __forceinline__ void w3_prime(W3State& st) {          // the first DIST requests
  if constexpr (S < W3Dist<MODE>::v) { w3_issue<MODE, S>(st); w3_prime<MODE, S + 1>(st); }
}

template <int MODE, int S>
__device__ __forceinline__ void w3_steps(W3State& st) {
  if constexpr (S < W3_STEPS) {
    constexpr int j = S / 36, tap = (S % 36) / 4, kb = S % 4;
    constexpr int W3_DIST = W3Dist<MODE>::v;
    w3_issue<MODE, S + W3_DIST>(st);
    // the fragment of this step is the oldest of at most DIST + 1 requests
    constexpr int younger = (W3_STEPS - 1 - S) < W3_DIST ? (W3_STEPS - 1 - S) : W3_DIST;
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(younger) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    // (the first MFMA of a pixel row starts from a constant zero: no accumulator clearing between tiles)
    constexpr bool first = tap == 0 && kb == 0;
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    st.acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.af[0][tap][kb], st.b[S % (W3_DIST + 1)], first ? zero : st.acc[0][j], 0, 0, 0);
    st.acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.af[1][tap][kb], st.b[S % (W3_DIST + 1)], first ? zero : st.acc[1][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    w3_steps<MODE, S + 1>(st);
  }
}

}  // namespace


template <int MODE>
__global__ __launch_bounds__(256, 1) void conv_ws3_kernel(const ConvArgs a, const int tiles_y, const int tiles_x) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const LevelGeo& g = a.lev[0];
  const int H = g.Hs, W = g.Ws;
  const int ntiles = a.N * tiles_y * tiles_x;
  const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
  auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.src), 0, g.src_bytes, 0x00020000);

  W3State st;
  // ---- weights of this wave's 32 output channels -> registers (MFMA A operands), once.  a.w is [128][9][128].
  {
    const __bf16* wbase = (const __bf16*)a.w + (size_t)(wave * 32 + fr) * (9 * 128) + fg * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) st.af[i][tap][kb] = *reinterpret_cast<const bf16x8_t*>(wbase + (size_t)i * 16 * (9 * 128) + tap * 128 + kb * 32);
  }
  f32x4_t bv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    bv[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (MODE == MODE_FWD && (a.flags & F_BIAS)) bv[i] = *reinterpret_cast<const f32x4_t*>(a.bias + wave * 32 + i * 16 + fg * 4);
  }

  // ---- fragment base addresses (buffer 0): pixel column fr + dw of the 16-wide window, channel block kb, swizzled by the column
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int col = fr + d - 1;
      st.rb[d][kb] = (uint32_t)(W3_GUARD + col * 256 + (((kb * 4 + fg) ^ (col & 15)) << 4));
    }

  // ---- staging: instruction k of a wave = halo row k, columns wave * 4 + fg (a wave instruction covers 4 pixels x 256 B)
  const int s_col = wave * 4 + fg;
  const uint32_t s_chunk = (uint32_t)(((lane & 15) ^ s_col) << 4);             // logical 16-B chunk this lane fetches (source-side swizzle)
  auto stage = [&](int t, int buf) {
    uint32_t base = 0;
    int h0 = 0;
    bool cok = false;                      // this lane's column lies inside the image (and the tile exists)
    if (t < ntiles) {
      const int tx = t % tiles_x, r = t / tiles_x;
      const int ty = r % tiles_y, n = r / tiles_y;
      h0 = ty * W3_TH;
      const int wc = tx * W3_TW - 1 + s_col;
      cok = (unsigned)wc < (unsigned)W;
      base = (uint32_t)(((n * H + h0 - 1) * W + wc) * 256) + s_chunk;      // row h0 - 1 (may lie outside the image: tested per k; arithmetic mod 2^32)
    }
    char* dst = smem + W3_GUARD + buf * W3_BUF + wave * 1024;
#pragma unroll
    for (int k = 0; k < W3_TH + 2; ++k) {
      const bool rok = (unsigned)(h0 - 1 + k) < (unsigned)H;                    // scalar
      const uint32_t off = (rok && cok) ? base + (uint32_t)(k * W * 256) : SOD_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, SOD_LDS(dst + k * 4096), 16, off, 0, 0, 0);
    }
  };

  stage((int)bid, 0);
  stage((int)bid + (int)gridDim.x, 1);
  for (int it = 0, t = (int)bid; t < ntiles; ++it, t += (int)gridDim.x) {
    const int buf = it & 1;
    // tile `it` has landed: younger are the 8 stores of the previous tile and the 10 requests of the next one (dgrad: its mask loads were waited for)
    if (it == 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (it > 0) {                                        // the base addresses follow the buffer (kept in place: no second set of registers)
      const uint32_t delta = buf ? (uint32_t)W3_BUF : (uint32_t)(-W3_BUF);
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) st.rb[d][kb] += delta;
    }
    w3_prime<MODE, 0>(st);
    w3_steps<MODE, 0>(st);

    // ---- epilogue 1, accumulator layout -> the consumed halo buffer as a [128 px][128 ch] bf16 tile (8-byte units XOR-swizzled by 2 * column)
    __builtin_amdgcn_s_barrier();                       // every wave has read its last fragment of this buffer
    char* stg = smem + W3_GUARD + buf * W3_BUF;
#pragma unroll
    for (int j = 0; j < W3_TH; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x4_t v = st.acc[i][j] + bv[i];
        if (MODE == MODE_FWD && (a.flags & F_RELU)) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        const bf16x4_t o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        const int unit = wave * 8 + i * 4 + fg;          // 8-byte unit (4 channels) inside the pixel's 256-byte row
        *reinterpret_cast<bf16x4_t*>(stg + (j * 16 + fr) * 256 + ((unit ^ (2 * fr)) << 3)) = o;
      }
    // ---- epilogue 2, row layout: 16 bytes = 8 channels of one pixel per lane; a wave instruction stores four pixels' full 256-byte runs.
    // Thread (c16, col) = (tid & 15, tid >> 4) handles column col of every window row k.  The data gradient's mask values are requested
    // BEFORE the barrier (the accumulators are dead: their registers carry the eight loads), so their latency runs beside it.
    const int tx = t % tiles_x, tr = t / tiles_x;
    const int ty = tr % tiles_y, n = tr / tiles_y;
    const int c16 = tid & 15, col = tid >> 4;
    const int wq = tx * W3_TW - 1 + col;
    const bool cok = col >= 1 && col <= W3_TW && wq < W;
    const size_t off0 = ((size_t)(n * H + ty * W3_TH) * W + wq) * 128 + c16 * 8;
    bf16x8_t mv[4];
    __builtin_amdgcn_sched_barrier(0);       // (the loads below must not be scheduled in front of the staging writes: the accumulators would still be live)
    const bool masked = MODE == MODE_DGRAD && (a.flags & F_MASK) != 0;
    if (masked) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (cok && ty * W3_TH + k < H) mv[k] = *reinterpret_cast<const bf16x8_t*>((const __bf16*)g.mask + off0 + (size_t)k * W * 128);
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      bf16x8_t o = *reinterpret_cast<const bf16x8_t*>(stg + (k * 16 + col) * 256 + (((c16 * 2) ^ (2 * col)) << 3));
      if (cok && ty * W3_TH + k < H) {
        if (masked) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = ((float)mv[k & 3][e] > 0.f) ? o[e] : (__bf16)0.f;
        }
        sod_store16((__bf16*)g.dst + off0 + (size_t)k * W * 128, o);
      }
      // rows 4 .. 7: their mask values take the registers of rows 0 .. 3 as those are consumed
      if (masked && k < 4 && cok && ty * W3_TH + k + 4 < H)
        mv[k] = *reinterpret_cast<const bf16x8_t*>((const __bf16*)g.mask + off0 + (size_t)(k + 4) * W * 128);
    }
    __builtin_amdgcn_s_barrier();                       // the staging tile has been read: the buffer may be refilled
    stage(t + 2 * (int)gridDim.x, buf);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // dead prefetches must have landed before the LDS allocation goes back
}

// 3x3, stride 1, pad 1, no dilation, 128 -> 128 channels, one dense level, bf16 output; forward: bias / ReLU; backward: a bf16 mask tensor
bool ws3_supported(const ConvArgs& a, int mode, bool out_f32, int cus, bool any_size) {
  if (out_f32 || cus != 256 || a.nlev != 1 || a.cwin) return false;
  if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.dil != 1) return false;
  if (a.Cred != 128 || a.Cpitch != 128 || a.Nout != 128) return false;
  const LevelGeo& g = a.lev[0];
  if (g.pstart != 0 || g.Hs != g.Hp || g.Ws != g.Wp) return false;
  if (g.src_img_stride != g.Hs * g.Ws * 128 || g.dst_img_stride != g.Hp * g.Wp * 128) return false;
  if ((long long)g.P * 128 * 2 >= (1ll << 31)) return false;
  const int allowed = (mode == MODE_FWD) ? (F_BIAS | F_RELU | F_REVERSE) : (F_MASK | F_REVERSE);
  if (a.flags & ~allowed) return false;
  // the persistent grid wants a few tiles per workgroup
  const long long tiles = (long long)a.N * ((g.Hs + W3_TH - 1) / W3_TH) * ((g.Ws + W3_TW - 1) / W3_TW);
  return any_size || tiles >= 4 * 256;
}

int launch_ws3(const ConvArgs& a, int mode, hipStream_t st) {
  const LevelGeo& g = a.lev[0];
  const int ty = (g.Hs + W3_TH - 1) / W3_TH, tx = (g.Ws + W3_TW - 1) / W3_TW;
  static bool attr_done[2] = {false, false};
  const void* kern = mode == MODE_FWD ? (const void*)conv_ws3_kernel<MODE_FWD> : (const void*)conv_ws3_kernel<MODE_DGRAD>;
  if (!attr_done[mode]) {
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, W3_LDS);
    if (e != hipSuccess) return (int)e;
    attr_done[mode] = true;
  }
  if (mode == MODE_FWD) SOD_LAUNCH(conv_ws3_kernel<MODE_FWD>, dim3(256), dim3(256), W3_LDS, st, a, ty, tx);
  else SOD_LAUNCH(conv_ws3_kernel<MODE_DGRAD>, dim3(256), dim3(256), W3_LDS, st, a, ty, tx);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

}  // namespace sodconv
