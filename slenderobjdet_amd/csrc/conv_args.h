// Argument blocks shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv_igemm256.hip).
#pragma once
#include "common.h"
#include "../../include/slender_hip.h"
#include <stdlib.h>

namespace sodconv {

enum { MODE_FWD = 0, MODE_DGRAD = 1 };
enum {
  F_BIAS = 1, F_RELU = 2, F_RES = 4, F_RES_UP2 = 8, F_MASK = 16,
};
constexpr int MAXLEV = SOD_CONV_MAX_LEVELS;

// One "level" = one (N,H,W,C) tensor; a launch may cover several levels that share the weights (the FPN levels of
// the FCOS towers), so that the small levels do not pay a launch + tail each.
struct LevelGeo {
  const void* src;     // fwd: x (N,Hs,Ws,Cred); dgrad: dy (N,Hs,Ws,Cred)
  void* dst;           // (N,Hp,Wp,Nout) rows at dst_img_stride
  const void* res;     // bf16, indexed like dst (or half-resolution with F_RES_UP2)
  const void* mask;    // bf16, indexed like dst: dst = mask>0 ? v : 0 (ReLU backward)
  uint32_t src_bytes;
  int Hs, Ws, Hp, Wp, P;
  int tile0;           // first pixel tile of this level
  int pstart;          // first pixel this launch covers (a launch may handle only the tail of a level, see dispatch_conv)
  int src_img_stride, dst_img_stride, res_img_stride;  // elements
  FastDiv div_hw, div_w;
};

struct ConvArgs {
  LevelGeo lev[MAXLEV];
  int nlev;
  const void* w;       // [Nout][R*S*Cred]
  const float* bias;   // [Nout] or null
  uint32_t w_bytes;
  int N, Cred, Nout;
  int R, S, stride, pad, dil;
  int Kred, T;         // R*S*Cred, #K-steps
  int flags;
  int nq_tiles, np_tiles;
  FastDiv div_cpt /* Cred/64 (fast) or Cred/8 (generic) */, div_s, div_stride;
  FastDiv div_rs;      // R*S
  int tap_inner;       // linear path: K-step order (channel chunk outer, tap inner) - the taps of one chunk re-read the same cache lines
};

// K-step order of the linear staging path: (channel chunk outer, tap inner) makes the R*S taps of one 64-channel chunk re-read the
// same cache lines back to back.  Measured on the FCOS head (16 x 5 levels, 256 -> 256 3x3): L2 fetch traffic of the 256x256 kernel
// 707 -> 205 MB per launch (algorithmic: 184 MB) at equal time; the 16-channel variant of the 128x128 kernel (box / centerness
// prediction, Nout = 8) halves its time (1.59 GB of L2 fetches per launch before); the other variants measure equal and keep the
// (tap outer) order their model-level parity tests were pinned with.  SOD_CONV_TAP_INNER=0|1 forces one order for every variant of the 128x128 kernel (the 256x256 kernel always uses tap-inner).
inline int conv_tap_inner(int dflt) {
  static int v = -2;
  if (v == -2) { const char* e = getenv("SOD_CONV_TAP_INNER"); v = e ? atoi(e) : -1; }
  return v < 0 ? dflt : v;
}

// conv_igemm256.hip: 256x256x64 tile, 8 waves, 8-phase main loop.  Returns SOD_EARG when the shape is outside its fast path.
bool conv256_supported(const ConvArgs& a, int mode);
// max_pt_tiles > 0 launches only the first max_pt_tiles pixel tiles (the caller covers the rest with the 128x128 kernel).
int launch_conv256(const ConvArgs& a, int mode, bool out_f32, int max_pt_tiles, hipStream_t st);

// --------------------------------------------------------------------------------------------
// wgrad: dW[q][tap][c] += sum_p dY[p][q] * X[p shifted by tap][c]
// The contraction runs over a VIRTUAL pixel index that concatenates the levels (each padded to a multiple of 64), so
// one launch reduces over all FPN levels that share the weights.  (conv_igemm.hip: 128x128 tiles; conv_wgrad256.hip: 256x256.)
// --------------------------------------------------------------------------------------------
struct WLevel {
  const void* dy;      // (N,Ho,Wo,K) bf16 rows at dy_img_stride
  const void* x;       // (N,Hx,Wx,C) bf16
  uint32_t dy_bytes, x_bytes;
  int Hx, Wx, Ho, Wo, P;
  int v0;              // first virtual pixel of this level (multiple of 64)
  int dy_img_stride, x_img_stride;
  FastDiv div_hw, div_w;
};

struct WgradArgs {
  WLevel lev[MAXLEV];
  int nlev;
  float* dw;           // [K][R][S][C] fp32, accumulated
  const float* qscale; // optional per-output-channel factor (folded FrozenBN scale)
  int dbg_plain_store; // timing experiment only (SOD_WGRAD_PLAIN=1): racy plain stores instead of atomics
  float* partial;      // optional fp32 slabs: blocks store their partial tile, a reduce kernel sums the splits
  int det;             // deterministic: the reduce kernel adds into dw with plain read-modify-writes in a fixed order
  int N, C, K;
  int R, S, stride, pad, dil;
  int V, nz, v_per_split;   // total virtual pixels; v_per_split multiple of 64
  int QT, CT;
  FastDiv div_s;
};

enum { WGRAD_DETERMINISTIC = 1 };   // sod_conv2d_wgrad flags

// conv_wgrad256.hip: 256(q) x 256(c) output tile per workgroup of 8 waves, one workgroup per CU, split over pixels with fp32 slabs in
// the caller's workspace and a fixed-order reduce kernel (no atomics anywhere).
bool wgrad256_supported(const WgradArgs& a);
long long wgrad256_workspace_bytes(const WgradArgs& a, int cus);
int launch_wgrad256(WgradArgs& a, int cus, float* ws, long long ws_bytes, hipStream_t st);

}  // namespace sodconv
