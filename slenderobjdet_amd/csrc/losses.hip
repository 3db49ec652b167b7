// Loss and target-assignment kernels of the FCOS training path (HBM-bound, fp32 arithmetic).
//
// Replaces, for this path:
//   fvcore.nn.sigmoid_focal_loss_jit            (call site slender_det/modeling/meta_arch/fcos/fcosv2.py:124)
//   slender_det.layers.iou_loss                 (slender_det/layers/iou_loss.py:4-37)
//   compute_centerness_targets                  (slender_det/modeling/meta_arch/fcos/utils.py:295-300)
//   compute_targets_for_locations / get_sample_region (fcos/utils.py:108-212)
//   F.binary_cross_entropy_with_logits(sum)     (fcosv2.py:140)
// All reductions are two-stage (per-block partials, then one block adds them in index order) so results are
// bitwise reproducible from run to run.
#include "common.h"
#include "../../include/slender_hip.h"

namespace {

constexpr int RED_BLOCKS = 1024;   // max partials per reduction (workspace floats per reduced scalar)

__global__ void finish_sum_kernel(const float* __restrict__ part, int nblk, int nscalars, float* __restrict__ out,
                                  int accumulate) {
  // one wave per scalar; sums partials in a fixed order
  __shared__ float red[4];
  for (int s = 0; s < nscalars; ++s) {
    float v = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) v += part[s * RED_BLOCKS + i];
    v = block_sum_256(v, red);
    if (threadIdx.x == 0) out[s] = accumulate ? out[s] + v : v;
    __syncthreads();
  }
}

// sigmoid(x) and softplus(-|x|) = log1p(exp(-|x|)) from ONE hardware exponential (v_exp_f32) and, only where it is needed, one
// hardware logarithm: for e = exp(-|x|) < 2^-5 the alternating series e - e^2/2 + e^3/3 - e^4/4 is exact to < 1e-8 relative, above that
// 1 + e keeps >= 19 significant bits and v_log_f32 is accurate to ~1 ulp.  (The libm expf / log1pf / division sequences made these
// kernels VALU-bound: 140 us for 28.7 M logits against 23 us of HBM time.)
__device__ __forceinline__ void sigmoid_softplus(float x, float& p, float& sp) {
  const float e = __expf(-fabsf(x));
  const float r = __frcp_rn(1.f + e);
  p = (x >= 0.f) ? r : e * r;
  sp = (e < 0.03125f) ? e * (1.f - e * (0.5f - e * (0.33333334f - e * 0.25f))) : __logf(1.f + e);
}

__device__ __forceinline__ float focal_term(float x, float t, float alpha, float gamma) {
  float p, sp;
  sigmoid_softplus(x, p, sp);
  const float ce = fmaxf(x, 0.f) - x * t + sp;
  const float pt = p * t + (1.f - p) * (1.f - t);
  const float om = 1.f - pt;
  const float mod = (gamma == 2.f) ? om * om : powf(om, gamma);
  float l = ce * mod;
  if (alpha >= 0.f) l *= alpha * t + (1.f - alpha) * (1.f - t);
  return l;
}

__device__ __forceinline__ float focal_grad(float x, float t, float alpha, float gamma) {
  float p, sp;
  sigmoid_softplus(x, p, sp);
  const float ce = fmaxf(x, 0.f) - x * t + sp;
  const float pt = p * t + (1.f - p) * (1.f - t);
  const float om = 1.f - pt;
  const float mod = (gamma == 2.f) ? om * om : powf(om, gamma);
  float g = -(2.f * t - 1.f) * mod * (gamma * pt * ce + om);
  if (alpha >= 0.f) g *= alpha * t + (1.f - alpha) * (1.f - t);
  return g;
}

__global__ __launch_bounds__(256) void focal_fwd_kernel(const float* __restrict__ x, const int* __restrict__ labels,
                                                        const float* __restrict__ dense, long long M, int K, int ld,
                                                        float alpha, float gamma, float* __restrict__ elem,
                                                        float* __restrict__ part) {
  __shared__ float red[4];
  float acc = 0.f;
  const long long total = M * K;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long m = i / K;
    const int k = (int)(i - m * K);
    const float t = dense ? dense[m * K + k] : ((labels[m] == k) ? 1.f : 0.f);
    float l = focal_term(x[m * ld + k], t, alpha, gamma);
    if (!dense && labels[m] < 0) l = 0.f;     // negative label = ignored row (RetinaNet valid_mask, retina_rotated.py:208)
    if (elem) elem[m * K + k] = l;
    acc += l;
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// grad written as bf16 (training path, padded rows) or f32
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void focal_bwd_kernel(const float* __restrict__ x, const int* __restrict__ labels,
                                                        const float* __restrict__ dense, long long M, int K, int ld,
                                                        float alpha, float gamma, const float* __restrict__ scale_num,
                                                        const float* __restrict__ scale_den, float den_mul, float den_min, void* __restrict__ dx,
                                                        int ld_out) {
  float sc = scale_num ? scale_num[0] : 1.f;
  if (scale_den) sc /= fmaxf(scale_den[0] * den_mul, den_min);
  const long long total = M * ld_out;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long m = i / ld_out;
    const int k = (int)(i - m * ld_out);
    float g = 0.f;
    if (k < K) {
      const float t = dense ? dense[m * K + k] : ((labels[m] == k) ? 1.f : 0.f);
      g = (!dense && labels[m] < 0) ? 0.f : focal_grad(x[m * ld + k], t, alpha, gamma) * sc;
    }
    if (OUT_BF16) ((__bf16*)dx)[i] = (__bf16)g; else ((float*)dx)[i] = g;
  }
}

// Vectorised forms for the training path (class-index labels, K % 4 == 0, rows 16-B aligned, M*ld_out < 2^31): one thread = 4
// consecutive classes of one location; the row index comes from a 32-bit multiply-high division and the label is read once per 4
// elements.  (The scalar kernels above spend most of their time in a 64-bit division per element: 199 us for the 16 x 22400 x 80
// FCOS logits, 0.57 TB/s.)
__global__ __launch_bounds__(256) void focal_fwd_vec4_kernel(const float* __restrict__ x, const int* __restrict__ labels, uint32_t total4,
                                                             FastDiv div_k4, int ld, float alpha, float gamma, float* __restrict__ elem,
                                                             int K, float* __restrict__ part) {
  __shared__ float red[4];
  float acc = 0.f;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total4; i += gridDim.x * 256u) {
    const uint32_t m = fd_div(i, div_k4);
    const int k = (int)(i - m * div_k4.d) * 4;
    const int lab = labels[m];
    const f32x4_t xv = *reinterpret_cast<const f32x4_t*>(x + (size_t)m * ld + k);
    f32x4_t lv;
#pragma unroll
    for (int e = 0; e < 4; ++e) lv[e] = (lab < 0) ? 0.f : focal_term(xv[e], (lab == k + e) ? 1.f : 0.f, alpha, gamma);
    if (elem) *reinterpret_cast<f32x4_t*>(elem + (size_t)m * K + k) = lv;
    acc += (lv[0] + lv[1]) + (lv[2] + lv[3]);
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// Forward sum AND the un-scaled gradient in one pass over the logits (round 4): the loss term and its derivative share the sigmoid, the
// softplus, ce, p_t and the modulating factor, and the backward pass of the training step only differs from this gradient by ONE scalar
// (upstream gradient / normaliser), which the consumers apply (scaled weights in the data gradient, a per-channel factor in the weight
// gradient, a scalar in the bias gradient).  The 1 GB fp32 logits of RetinaNet are read once per step instead of twice.
__global__ __launch_bounds__(256) void focal_fwd_grad_vec4_kernel(const float* __restrict__ x, const int* __restrict__ labels, uint32_t total4,
                                                                  FastDiv div_k4, int K, int ld, float alpha, float gamma,
                                                                  float* __restrict__ part, __bf16* __restrict__ dx, int ld_out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total4; i += gridDim.x * 256u) {
    const uint32_t m = fd_div(i, div_k4);
    const int k = (int)(i - m * div_k4.d) * 4;       // column of the (padded) gradient row
    f32x4_t gv = {0.f, 0.f, 0.f, 0.f};
    if (k < K) {
      const int lab = labels[m];
      if (lab >= 0) {
        const f32x4_t xv = *reinterpret_cast<const f32x4_t*>(x + (size_t)m * ld + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = (lab == k + e) ? 1.f : 0.f;
          float p, sp;
          sigmoid_softplus(xv[e], p, sp);
          const float ce = fmaxf(xv[e], 0.f) - xv[e] * t + sp;
          const float pt = p * t + (1.f - p) * (1.f - t);
          const float om = 1.f - pt;
          const float mod = (gamma == 2.f) ? om * om : powf(om, gamma);
          const float at = (alpha >= 0.f) ? alpha * t + (1.f - alpha) * (1.f - t) : 1.f;
          acc += ce * mod * at;                                                   // focal_term
          gv[e] = -(2.f * t - 1.f) * mod * (gamma * pt * ce + om) * at;           // focal_grad
        }
      }
    }
    bf16x4_t o = {(__bf16)gv[0], (__bf16)gv[1], (__bf16)gv[2], (__bf16)gv[3]};
    *reinterpret_cast<bf16x4_t*>(dx + (size_t)m * ld_out + k) = o;
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

template <bool OUT_BF16>
__global__ __launch_bounds__(256) void focal_bwd_vec4_kernel(const float* __restrict__ x, const int* __restrict__ labels, uint32_t total4,
                                                             FastDiv div_k4, int K, int ld, float alpha, float gamma,
                                                             const float* __restrict__ scale_num, const float* __restrict__ scale_den,
                                                             float den_mul, float den_min, void* __restrict__ dx, int ld_out) {
  float sc = scale_num ? scale_num[0] : 1.f;
  if (scale_den) sc /= fmaxf(scale_den[0] * den_mul, den_min);
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total4; i += gridDim.x * 256u) {
    const uint32_t m = fd_div(i, div_k4);
    const int k = (int)(i - m * div_k4.d) * 4;       // column of the (padded) output row
    f32x4_t gv = {0.f, 0.f, 0.f, 0.f};
    if (k < K) {                                      // K % 4 == 0: a group is entirely inside or entirely padding
      const int lab = labels[m];
      if (lab >= 0) {
        const f32x4_t xv = *reinterpret_cast<const f32x4_t*>(x + (size_t)m * ld + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[e] = focal_grad(xv[e], (lab == k + e) ? 1.f : 0.f, alpha, gamma) * sc;
      }
    }
    if (OUT_BF16) {
      bf16x4_t o = {(__bf16)gv[0], (__bf16)gv[1], (__bf16)gv[2], (__bf16)gv[3]};
      *reinterpret_cast<bf16x4_t*>((__bf16*)dx + (size_t)m * ld_out + k) = o;
    } else {
      *reinterpret_cast<f32x4_t*>((float*)dx + (size_t)m * ld_out + k) = gv;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// IoU family loss on LTRB distances (reference: layers/iou_loss.py:4-37)
// ---------------------------------------------------------------------------------------------
struct IouOut { float loss; float g[4]; };

__device__ __forceinline__ void minmax_grad(float a, float b, float& dmin, float& dmax) {
  // torch.minimum/maximum split the gradient on ties
  dmin = (a < b) ? 1.f : ((a == b) ? 0.5f : 0.f);
  dmax = (a > b) ? 1.f : ((a == b) ? 0.5f : 0.f);
}

template <bool GRAD>
__device__ __forceinline__ IouOut iou_ltrb(const float p[4], const float t[4], int type) {
  IouOut o;
  const float pl = p[0], ptp = p[1], pr = p[2], pb = p[3];
  const float tl = t[0], tt = t[1], tr = t[2], tb = t[3];
  const float ta = (tl + tr) * (tt + tb);
  const float pa = (pl + pr) * (ptp + pb);
  const float wi = fminf(pl, tl) + fminf(pr, tr);
  const float gw = fmaxf(pl, tl) + fmaxf(pr, tr);
  const float hi = fminf(pb, tb) + fminf(ptp, tt);
  const float gh = fmaxf(pb, tb) + fmaxf(ptp, tt);
  const float ac = gw * gh + 1e-7f;
  const float ai = wi * hi;
  const float au = ta + pa - ai;
  const float iou = (ai + 1.0f) / (au + 1.0f);
  const float giou = iou - (ac - au) / ac;
  if (type == SOD_IOU_LOSS_IOU) o.loss = -logf(iou);
  else if (type == SOD_IOU_LOSS_LINEAR) o.loss = 1.f - iou;
  else o.loss = 1.f - giou;
  if (GRAD) {
    // order l,t,r,b ; l and r act on widths, t and b on heights
    const float pv[4] = {pl, ptp, pr, pb};
    const float tv[4] = {tl, tt, tr, tb};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float dmin, dmax;
      minmax_grad(pv[e], tv[e], dmin, dmax);
      const bool horiz = (e == 0 || e == 2);
      const float dpa = horiz ? (ptp + pb) : (pl + pr);
      const float dai = horiz ? hi * dmin : wi * dmin;
      const float dac = horiz ? gh * dmax : gw * dmax;
      const float dau = dpa - dai;
      const float diou = (dai * (au + 1.0f) - (ai + 1.0f) * dau) / ((au + 1.0f) * (au + 1.0f));
      float d;
      if (type == SOD_IOU_LOSS_IOU) d = -diou / iou;
      else if (type == SOD_IOU_LOSS_LINEAR) d = -diou;
      else d = -(diou - (au * dac - ac * dau) / (ac * ac));
      o.g[e] = d;
    }
  }
  return o;
}

__global__ __launch_bounds__(256) void iou_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                      const float* __restrict__ weight, const int* __restrict__ mask,
                                                      int mask_bg, long long P, int type, float* __restrict__ elem,
                                                      float* __restrict__ part) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long long)gridDim.x * 256) {
    float l = 0.f;
    if (!mask || (mask[i] >= 0 && mask[i] != mask_bg)) {
      const f32x4_t pv = *reinterpret_cast<const f32x4_t*>(pred + i * 4);
      const f32x4_t tv = *reinterpret_cast<const f32x4_t*>(target + i * 4);
      const float p[4] = {pv[0], pv[1], pv[2], pv[3]}, t[4] = {tv[0], tv[1], tv[2], tv[3]};
      l = iou_ltrb<false>(p, t, type).loss;
      if (weight) l *= weight[i];
    }
    if (elem) elem[i] = l;
    acc += l;
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void iou_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                      const float* __restrict__ weight, const int* __restrict__ mask,
                                                      int mask_bg, long long P, int type, const float* __restrict__ scale,
                                                      float* __restrict__ dpred) {
  const float sc = scale ? scale[0] : 1.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long long)gridDim.x * 256) {
    f32x4_t g = {0.f, 0.f, 0.f, 0.f};
    if (!mask || (mask[i] >= 0 && mask[i] != mask_bg)) {
      const f32x4_t pv = *reinterpret_cast<const f32x4_t*>(pred + i * 4);
      const f32x4_t tv = *reinterpret_cast<const f32x4_t*>(target + i * 4);
      const float p[4] = {pv[0], pv[1], pv[2], pv[3]}, t[4] = {tv[0], tv[1], tv[2], tv[3]};
      const IouOut o = iou_ltrb<true>(p, t, type);
      const float w = (weight ? weight[i] : 1.f) * sc;
      g = f32x4_t{o.g[0] * w, o.g[1] * w, o.g[2] * w, o.g[3] * w};
    }
    *reinterpret_cast<f32x4_t*>(dpred + i * 4) = g;
  }
}

// ---------------------------------------------------------------------------------------------
// FCOS target assignment (reference: fcos/utils.py:108-212, fcosv2.py:150-172)
// ---------------------------------------------------------------------------------------------
struct AssignArgs {
  const float* boxes;      // [sumG,4] XYXY
  const int* classes;      // [sumG]
  const int* box_off;      // [N+1]
  int N, L, num_classes;
  int lvl_off[SOD_MAX_LEVELS + 1];
  int lvl_w[SOD_MAX_LEVELS];
  int lvl_stride[SOD_MAX_LEVELS];
  float lvl_lo[SOD_MAX_LEVELS], lvl_hi[SOD_MAX_LEVELS], lvl_rad[SOD_MAX_LEVELS];  // rad = stride*radius (fp32), <=0: box test
  int nlevels;
  int* labels; float* reg; float* ctr;
};

__device__ __forceinline__ float centerness_of(float l, float t, float r, float b) {
  return sqrtf((fminf(l, r) / fmaxf(l, r)) * (fminf(t, b) / fmaxf(t, b)));
}

__global__ __launch_bounds__(256) void fcos_assign_kernel(const AssignArgs a, float* __restrict__ part) {
  __shared__ float red[4];
  const int n = blockIdx.y;
  const int g0 = a.box_off[n], g1 = a.box_off[n + 1];
  float npos = 0.f, sctr = 0.f;
  for (int loc = blockIdx.x * 256 + threadIdx.x; loc < a.L; loc += gridDim.x * 256) {
    int lv = 0;
    while (lv + 1 < a.nlevels && loc >= a.lvl_off[lv + 1]) ++lv;
    const int idx = loc - a.lvl_off[lv];
    const int iy = idx / a.lvl_w[lv], ix = idx - iy * a.lvl_w[lv];
    const int st = a.lvl_stride[lv];
    const float x = (float)(ix * st) + (float)(st / 2);
    const float y = (float)(iy * st) + (float)(st / 2);
    const float rad = a.lvl_rad[lv];
    float best = 100000000.f;   // INF of the reference
    int bi = 0;
    float bl = 0.f, bt = 0.f, br = 0.f, bb = 0.f;
    bool first_center_zero = false;
    if (rad > 0.f && g1 > g0) {
      const float cx0 = (a.boxes[g0 * 4 + 0] + a.boxes[g0 * 4 + 2]) / 2.f;
      first_center_zero = (cx0 == 0.f);   // reference quirk: get_sample_region returns all-false (utils.py:121-122)
    }
    for (int g = g0; g < g1; ++g) {
      const float x1 = a.boxes[g * 4 + 0], y1 = a.boxes[g * 4 + 1], x2 = a.boxes[g * 4 + 2], y2 = a.boxes[g * 4 + 3];
      const float l = x - x1, t = y - y1, r = x2 - x, b = y2 - y;
      bool inside;
      if (rad > 0.f) {
        const float cx = (x1 + x2) / 2.f, cy = (y1 + y2) / 2.f;
        const float xmin = cx - rad, ymin = cy - rad, xmax = cx + rad, ymax = cy + rad;
        const float c0 = (xmin > x1) ? xmin : x1;
        const float c1 = (ymin > y1) ? ymin : y1;
        const float c2 = (xmax > x2) ? x2 : xmax;
        const float c3 = (ymax > y2) ? y2 : ymax;
        const float m = fminf(fminf(x - c0, y - c1), fminf(c2 - x, c3 - y));
        inside = (m > 0.f) && !first_center_zero;
      } else {
        inside = fminf(fminf(l, t), fminf(r, b)) > 0.f;
      }
      const float mx = fmaxf(fmaxf(l, t), fmaxf(r, b));
      const bool cared = (mx >= a.lvl_lo[lv]) && (mx <= a.lvl_hi[lv]);
      float area = (x2 - x1) * (y2 - y1);
      if (!inside || !cared) area = 100000000.f;
      if (g == g0 || area < best) { best = area; bi = g; bl = l; bt = t; br = r; bb = b; }   // first minimum wins (torch.min)
    }
    int label = a.num_classes;
    float c = 0.f;
    if (g1 > g0 && best != 100000000.f) {
      label = a.classes[bi];
      if (label >= 0 && label != a.num_classes) { c = centerness_of(bl, bt, br, bb); npos += 1.f; sctr += c; }
    }
    const long long o = (long long)n * a.L + loc;
    a.labels[o] = label;
    *reinterpret_cast<f32x4_t*>(a.reg + o * 4) = f32x4_t{bl, bt, br, bb};
    a.ctr[o] = c;
  }
  npos = block_sum_256(npos, red);
  sctr = block_sum_256(sctr, red);
  if (threadIdx.x == 0) {
    const int b = blockIdx.y * gridDim.x + blockIdx.x;
    part[b] = npos;
    part[RED_BLOCKS + b] = sctr;
  }
}

// ---------------------------------------------------------------------------------------------
// Fused FCOS regression + centerness loss over all N*L locations (no gather, no host sync).
//   pred = exp(scale_l * raw)   or   relu(scale_l * raw) * stride_l     (fcosv2.py:372-378)
//   reg_loss = sum_pos iou_loss(pred, tgt) * ctr_tgt ;  ctr_loss = sum_pos BCEWithLogits(ctr_logit, ctr_tgt)
// ---------------------------------------------------------------------------------------------
struct RegCtrArgs {
  const float* box_raw; int ld_box;       // [M, ld_box], first 4 columns
  const float* ctr_logit; int ld_ctr;     // [M, ld_ctr], column 0
  const int* labels; const float* reg_t; const float* ctr_t;
  const float* scales;                    // [nlevels] device
  int M, L, num_classes, type, norm_reg;
  int nlevels; int lvl_off[SOD_MAX_LEVELS + 1]; int lvl_stride[SOD_MAX_LEVELS];
};

__device__ __forceinline__ int level_of(const RegCtrArgs& a, int loc) {
  int lv = 0;
  while (lv + 1 < a.nlevels && loc >= a.lvl_off[lv + 1]) ++lv;
  return lv;
}

__global__ __launch_bounds__(256) void regctr_fwd_kernel(const RegCtrArgs a, float* __restrict__ part) {
  __shared__ float red[4];
  float lreg = 0.f, lctr = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.M; i += gridDim.x * 256) {
    const int lab = a.labels[i];
    if (lab < 0 || lab == a.num_classes) continue;
    const int lv = level_of(a, i % a.L);
    const float s = a.scales[lv];
    float p[4], t[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float z = a.box_raw[(long long)i * a.ld_box + e] * s;
      p[e] = a.norm_reg ? fmaxf(z, 0.f) * (float)a.lvl_stride[lv] : expf(z);
      t[e] = a.reg_t[(long long)i * 4 + e];
    }
    const float c = a.ctr_t[i];
    lreg += iou_ltrb<false>(p, t, a.type).loss * c;
    const float x = a.ctr_logit[(long long)i * a.ld_ctr];
    lctr += fmaxf(x, 0.f) - x * c + log1pf(expf(-fabsf(x)));
  }
  lreg = block_sum_256(lreg, red);
  lctr = block_sum_256(lctr, red);
  if (threadIdx.x == 0) { part[blockIdx.x] = lreg; part[RED_BLOCKS + blockIdx.x] = lctr; }
}

// writes d(box_raw) and d(ctr_logit) as bf16 into a padded [M, ld_out] buffer (columns 0..3 box, ctr_col ctr,
// remaining columns zero) and per-level d(scale) partials.
template <typename T>      // T = __bf16 (product path) or float (fp32-storage validation path, f32_path.hip)
__global__ __launch_bounds__(256) void regctr_bwd_kernel(const RegCtrArgs a, const float* __restrict__ greg,
                                                         const float* __restrict__ gctr, const float* __restrict__ norm, float inv_world,
                                                         T* __restrict__ dbox, int ld_out, int ctr_col,
                                                         T* __restrict__ dctr, int ld_dctr, int dctr_col,
                                                         float* __restrict__ part) {
  // norm[0] = sum over ranks of num_pos, norm[1] = sum over ranks of sum(ctr targets); inv_world = 1/world
  __shared__ float red[4];
  const float inv_np = 1.f / fmaxf(norm[0] * inv_world, 1.f);
  const float sreg = greg[0] / (norm[1] * inv_world);
  const float sctr = gctr[0] * inv_np;
  float dsc[SOD_MAX_LEVELS];
#pragma unroll
  for (int l = 0; l < SOD_MAX_LEVELS; ++l) dsc[l] = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.M; i += gridDim.x * 256) {
    const int lab = a.labels[i];
    float gb[4] = {0.f, 0.f, 0.f, 0.f};
    float gc = 0.f;
    if (lab >= 0 && lab != a.num_classes) {
      const int lv = level_of(a, i % a.L);
      const float s = a.scales[lv];
      float p[4], t[4], z[4], raw[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        raw[e] = a.box_raw[(long long)i * a.ld_box + e];
        z[e] = raw[e] * s;
        p[e] = a.norm_reg ? fmaxf(z[e], 0.f) * (float)a.lvl_stride[lv] : expf(z[e]);
        t[e] = a.reg_t[(long long)i * 4 + e];
      }
      const float c = a.ctr_t[i];
      const IouOut o = iou_ltrb<true>(p, t, a.type);
      float ds = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dp = o.g[e] * c * sreg;
        const float dz = a.norm_reg ? ((z[e] > 0.f) ? dp * (float)a.lvl_stride[lv] : 0.f) : dp * p[e];
        gb[e] = dz * s;
        ds += dz * raw[e];
      }
#pragma unroll
      for (int l = 0; l < SOD_MAX_LEVELS; ++l) if (l == lv) dsc[l] += ds;
      const float x = a.ctr_logit[(long long)i * a.ld_ctr];
      gc = (1.f / (1.f + expf(-x)) - c) * sctr;
    }
    if (dctr == dbox && ld_dctr == ld_out) {
      for (int e = 0; e < ld_out; ++e) {
        float v = 0.f;
        if (e < 4) v = gb[e]; else if (e == ctr_col) v = gc;
        dbox[(long long)i * ld_out + e] = (T)v;
      }
    } else {
      for (int e = 0; e < ld_out; ++e) dbox[(long long)i * ld_out + e] = (T)(e < 4 ? gb[e] : 0.f);
      dctr[(long long)i * ld_dctr + dctr_col] = (T)gc;   // caller pre-fills the rest of that row
    }
  }
#pragma unroll
  for (int l = 0; l < SOD_MAX_LEVELS; ++l) {
    const float v = block_sum_256(dsc[l], red);
    if (threadIdx.x == 0) part[l * RED_BLOCKS + blockIdx.x] = v;
  }
}

__global__ void fcos_finalize_kernel(const float* __restrict__ focal_sum, const float* __restrict__ regctr_sums,
                                     const float* __restrict__ stats, float inv_world, float* __restrict__ out) {
  // fcosv2.py:115-145: cls/num_pos_avg, reg/sum_ctr_avg, ctr/num_pos_avg ; no positives -> reg = ctr = 0
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float np = fmaxf(stats[0] * inv_world, 1.f);
    out[0] = focal_sum[0] / np;
    const float sc = stats[1] * inv_world;
    out[1] = (sc > 0.f) ? regctr_sums[0] / sc : 0.f;
    out[2] = regctr_sums[1] / np;
  }
}

inline int grid_for(long long n) {
  long long g = (n + 255) / 256;
  if (g > RED_BLOCKS) g = RED_BLOCKS;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" long long sod_reduce_workspace_bytes(void) { return (long long)RED_BLOCKS * 8 * sizeof(float); }

extern "C" int sod_sigmoid_focal_loss_fwd(const float* logits, const int* labels, const float* dense_targets,
                                          long long M, int K, int ld, float alpha, float gamma, float* elem_out,
                                          float* sum_out, float* ws, void* stream) {
  if (!logits || (!labels && !dense_targets) || !sum_out || !ws || M < 0 || K <= 0 || ld < K) return SOD_EARG;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = labels && !dense_targets && (K & 3) == 0 && (ld & 3) == 0 && M * (long long)K < (1ll << 31) &&
                   ((uintptr_t)logits & 15) == 0 && ((uintptr_t)elem_out & 15) == 0;
  const int g = vec ? grid_for(M * K / 4) : grid_for(M * K);
  if (vec)
    SOD_LAUNCH(focal_fwd_vec4_kernel, dim3(g), dim3(256), 0, st, logits, labels, (uint32_t)(M * K / 4), make_fastdiv((uint32_t)(K / 4)), ld, alpha,
               gamma, elem_out, K, ws);
  else
    SOD_LAUNCH(focal_fwd_kernel, dim3(g), dim3(256), 0, st, logits, labels, dense_targets, M, K, ld, alpha, gamma, elem_out, ws);
  SOD_LAUNCH(finish_sum_kernel, dim3(1), dim3(256), 0, st, ws, g, 1, sum_out, 0);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_sigmoid_focal_loss_fwd_grad(const float* logits, const int* labels, long long M, int K, int ld, float alpha, float gamma,
                                               float* sum_out, float* ws, void* dlogits_bf16, int ld_out, void* stream) {
  if (!logits || !labels || !sum_out || !ws || !dlogits_bf16 || M < 0 || K <= 0 || ld < K || ld_out < K) return SOD_EARG;
  if ((K & 3) || (ld & 3) || (ld_out & 3) || M * (long long)ld_out >= (1ll << 31) || ((uintptr_t)logits & 15) || ((uintptr_t)dlogits_bf16 & 7))
    return SOD_EARG;      // the vectorised form only (class-index labels, 4 classes per lane): callers fall back to the two-pass entry points
  hipStream_t st = (hipStream_t)stream;
  const uint32_t total4 = (uint32_t)(M * ld_out / 4);
  const int g = grid_for(M * ld_out / 4);
  SOD_LAUNCH(focal_fwd_grad_vec4_kernel, dim3(g), dim3(256), 0, st, logits, labels, total4, make_fastdiv((uint32_t)(ld_out / 4)), K, ld, alpha, gamma,
             ws, (__bf16*)dlogits_bf16, ld_out);
  SOD_LAUNCH(finish_sum_kernel, dim3(1), dim3(256), 0, st, ws, g, 1, sum_out, 0);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_sigmoid_focal_loss_bwd(const float* logits, const int* labels, const float* dense_targets,
                                          long long M, int K, int ld, float alpha, float gamma,
                                          const float* scale_num, const float* scale_den, float den_mul, float den_min,
                                          void* dlogits, int ld_out, int out_bf16, void* stream) {
  if (!logits || (!labels && !dense_targets) || !dlogits || M < 0 || K <= 0 || ld < K || ld_out < K) return SOD_EARG;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = labels && !dense_targets && (K & 3) == 0 && (ld & 3) == 0 && (ld_out & 3) == 0 && M * (long long)ld_out < (1ll << 31) &&
                   ((uintptr_t)logits & 15) == 0 && ((uintptr_t)dlogits & 15) == 0;
  if (vec) {
    const uint32_t total4 = (uint32_t)(M * ld_out / 4);
    const int gv = grid_for(M * ld_out / 4) * 2;
    const FastDiv d = make_fastdiv((uint32_t)(ld_out / 4));
    if (out_bf16)
      SOD_LAUNCH(focal_bwd_vec4_kernel<true>, dim3(gv), dim3(256), 0, st, logits, labels, total4, d, K, ld, alpha, gamma, scale_num, scale_den, den_mul, den_min, dlogits, ld_out);
    else
      SOD_LAUNCH(focal_bwd_vec4_kernel<false>, dim3(gv), dim3(256), 0, st, logits, labels, total4, d, K, ld, alpha, gamma, scale_num, scale_den, den_mul, den_min, dlogits, ld_out);
    SOD_CHECK_LAUNCH();
    return SOD_OK;
  }
  const int g = grid_for(M * ld_out) * 2;
  if (out_bf16)
    SOD_LAUNCH(focal_bwd_kernel<true>, dim3(g), dim3(256), 0, st, logits, labels, dense_targets, M, K, ld, alpha, gamma, scale_num, scale_den, den_mul, den_min, dlogits, ld_out);
  else
    SOD_LAUNCH(focal_bwd_kernel<false>, dim3(g), dim3(256), 0, st, logits, labels, dense_targets, M, K, ld, alpha, gamma, scale_num, scale_den, den_mul, den_min, dlogits, ld_out);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_iou_loss_fwd(const float* pred, const float* target, const float* weight, const int* mask, int mask_bg,
                                long long P, int loss_type, float* elem_out, float* sum_out, float* ws, void* stream) {
  if (!pred || !target || !sum_out || !ws || P < 0 || loss_type < 0 || loss_type > 2) return SOD_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int g = grid_for(P);
  SOD_LAUNCH(iou_fwd_kernel, dim3(g), dim3(256), 0, st, pred, target, weight, mask, mask_bg, P, loss_type, elem_out, ws);
  SOD_LAUNCH(finish_sum_kernel, dim3(1), dim3(256), 0, st, ws, g, 1, sum_out, 0);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_iou_loss_bwd(const float* pred, const float* target, const float* weight, const int* mask, int mask_bg,
                                long long P, int loss_type, const float* grad_scale, float* dpred, void* stream) {
  if (!pred || !target || !dpred || P < 0 || loss_type < 0 || loss_type > 2) return SOD_EARG;
  SOD_LAUNCH(iou_bwd_kernel, dim3(grid_for(P)), dim3(256), 0, (hipStream_t)stream, pred, target, weight, mask, mask_bg, P, loss_type, grad_scale, dpred);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_fcos_assign(const float* boxes, const int* classes, const int* box_offsets, int N,
                               int nlevels, const int* lvl_h, const int* lvl_w, const int* lvl_stride,
                               const float* lvl_lo, const float* lvl_hi, float radius, int num_classes,
                               int* labels, float* reg_targets, float* ctr_targets, float* stats /*[2]*/, float* ws, void* stream) {
  if (!box_offsets || !labels || !reg_targets || !ctr_targets || !stats || !ws || N <= 0 || nlevels <= 0 || nlevels > SOD_MAX_LEVELS)
    return SOD_EARG;
  AssignArgs a{};
  a.boxes = boxes; a.classes = classes; a.box_off = box_offsets; a.N = N; a.num_classes = num_classes; a.nlevels = nlevels;
  int off = 0;
  for (int l = 0; l < nlevels; ++l) {
    if (lvl_h[l] <= 0 || lvl_w[l] <= 0 || lvl_stride[l] <= 0) return SOD_EARG;
    a.lvl_off[l] = off; a.lvl_w[l] = lvl_w[l]; a.lvl_stride[l] = lvl_stride[l];
    a.lvl_lo[l] = lvl_lo[l]; a.lvl_hi[l] = lvl_hi[l];
    a.lvl_rad[l] = radius > 0.f ? (float)((double)lvl_stride[l] * (double)radius) : 0.f;
    off += lvl_h[l] * lvl_w[l];
  }
  for (int l = nlevels; l <= SOD_MAX_LEVELS; ++l) a.lvl_off[l] = off;
  a.L = off; a.labels = labels; a.reg = reg_targets; a.ctr = ctr_targets;
  int gx = (a.L + 255) / 256;
  if (gx * N > RED_BLOCKS) gx = RED_BLOCKS / N;
  if (gx < 1) return SOD_EARG;
  hipStream_t st = (hipStream_t)stream;
  SOD_LAUNCH(fcos_assign_kernel, dim3(gx, N), dim3(256), 0, st, a, ws);
  SOD_LAUNCH(finish_sum_kernel, dim3(1), dim3(256), 0, st, ws, gx * N, 2, stats, 0);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

static int fill_regctr(RegCtrArgs& a, int nlevels, const int* lvl_h, const int* lvl_w, const int* lvl_stride) {
  if (nlevels <= 0 || nlevels > SOD_MAX_LEVELS) return SOD_EARG;
  int off = 0;
  for (int l = 0; l < nlevels; ++l) { a.lvl_off[l] = off; a.lvl_stride[l] = lvl_stride[l]; off += lvl_h[l] * lvl_w[l]; }
  for (int l = nlevels; l <= SOD_MAX_LEVELS; ++l) a.lvl_off[l] = off;
  a.L = off; a.nlevels = nlevels;
  return SOD_OK;
}

extern "C" int sod_fcos_regctr_loss_fwd(const float* box_raw, int ld_box, const float* ctr_logit, int ld_ctr,
                                        const int* labels, const float* reg_targets, const float* ctr_targets,
                                        const float* scales, int N, int nlevels, const int* lvl_h, const int* lvl_w,
                                        const int* lvl_stride, int num_classes, int loss_type, int norm_reg_targets,
                                        float* sums /*[2]: reg, ctr*/, float* ws, void* stream) {
  if (!box_raw || !ctr_logit || !labels || !reg_targets || !ctr_targets || !scales || !sums || !ws) return SOD_EARG;
  RegCtrArgs a{};
  int rc = fill_regctr(a, nlevels, lvl_h, lvl_w, lvl_stride);
  if (rc) return rc;
  a.box_raw = box_raw; a.ld_box = ld_box; a.ctr_logit = ctr_logit; a.ld_ctr = ld_ctr; a.labels = labels;
  a.reg_t = reg_targets; a.ctr_t = ctr_targets; a.scales = scales; a.M = N * a.L; a.num_classes = num_classes;
  a.type = loss_type; a.norm_reg = norm_reg_targets;
  hipStream_t st = (hipStream_t)stream;
  const int g = grid_for(a.M);
  SOD_LAUNCH(regctr_fwd_kernel, dim3(g), dim3(256), 0, st, a, ws);
  SOD_LAUNCH(finish_sum_kernel, dim3(1), dim3(256), 0, st, ws, g, 2, sums, 0);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

static int regctr_bwd_impl(int out_f32, const float* box_raw, int ld_box, const float* ctr_logit, int ld_ctr,
                                        const int* labels, const float* reg_targets, const float* ctr_targets,
                                        const float* scales, int N, int nlevels, const int* lvl_h, const int* lvl_w,
                                        const int* lvl_stride, int num_classes, int loss_type, int norm_reg_targets,
                                        const float* grad_reg, const float* grad_ctr, const float* norm /*[2]*/, float inv_world,
                                        void* dbox, int ld_out, int ctr_col, void* dctr, int ld_dctr, int dctr_col,
                                        float* dscales, float* ws, void* stream) {
  if (!box_raw || !ctr_logit || !labels || !reg_targets || !ctr_targets || !scales || !grad_reg || !grad_ctr || !norm || !dbox || !dctr || !dscales || !ws)
    return SOD_EARG;
  if (ld_out < 4 || (dctr == dbox && (ctr_col < 4 || ctr_col >= ld_out))) return SOD_EARG;
  RegCtrArgs a{};
  int rc = fill_regctr(a, nlevels, lvl_h, lvl_w, lvl_stride);
  if (rc) return rc;
  a.box_raw = box_raw; a.ld_box = ld_box; a.ctr_logit = ctr_logit; a.ld_ctr = ld_ctr; a.labels = labels;
  a.reg_t = reg_targets; a.ctr_t = ctr_targets; a.scales = scales; a.M = N * a.L; a.num_classes = num_classes;
  a.type = loss_type; a.norm_reg = norm_reg_targets;
  hipStream_t st = (hipStream_t)stream;
  const int g = grid_for(a.M);
  if (out_f32)
    SOD_LAUNCH(regctr_bwd_kernel<float>, dim3(g), dim3(256), 0, st, a, grad_reg, grad_ctr, norm, inv_world, (float*)dbox, ld_out, ctr_col,
               (float*)dctr, ld_dctr, dctr_col, ws);
  else
    SOD_LAUNCH(regctr_bwd_kernel<__bf16>, dim3(g), dim3(256), 0, st, a, grad_reg, grad_ctr, norm, inv_world, (__bf16*)dbox, ld_out, ctr_col,
               (__bf16*)dctr, ld_dctr, dctr_col, ws);
  SOD_LAUNCH(finish_sum_kernel, dim3(1), dim3(256), 0, st, ws, g, nlevels, dscales, 1);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_fcos_regctr_loss_bwd(const float* box_raw, int ld_box, const float* ctr_logit, int ld_ctr,
                                        const int* labels, const float* reg_targets, const float* ctr_targets,
                                        const float* scales, int N, int nlevels, const int* lvl_h, const int* lvl_w,
                                        const int* lvl_stride, int num_classes, int loss_type, int norm_reg_targets,
                                        const float* grad_reg, const float* grad_ctr, const float* norm /*[2]*/, float inv_world,
                                        void* dbox, int ld_out, int ctr_col, void* dctr, int ld_dctr, int dctr_col,
                                        float* dscales, float* ws, void* stream) {
  return regctr_bwd_impl(0, box_raw, ld_box, ctr_logit, ld_ctr, labels, reg_targets, ctr_targets, scales, N, nlevels, lvl_h, lvl_w, lvl_stride, num_classes,
                         loss_type, norm_reg_targets, grad_reg, grad_ctr, norm, inv_world, dbox, ld_out, ctr_col, dctr, ld_dctr, dctr_col, dscales, ws, stream);
}

extern "C" int sod_fcos_regctr_loss_bwd_f32(const float* box_raw, int ld_box, const float* ctr_logit, int ld_ctr,
                                        const int* labels, const float* reg_targets, const float* ctr_targets,
                                        const float* scales, int N, int nlevels, const int* lvl_h, const int* lvl_w,
                                        const int* lvl_stride, int num_classes, int loss_type, int norm_reg_targets,
                                        const float* grad_reg, const float* grad_ctr, const float* norm /*[2]*/, float inv_world,
                                        void* dbox, int ld_out, int ctr_col, void* dctr, int ld_dctr, int dctr_col,
                                        float* dscales, float* ws, void* stream) {
  return regctr_bwd_impl(1, box_raw, ld_box, ctr_logit, ld_ctr, labels, reg_targets, ctr_targets, scales, N, nlevels, lvl_h, lvl_w, lvl_stride, num_classes,
                         loss_type, norm_reg_targets, grad_reg, grad_ctr, norm, inv_world, dbox, ld_out, ctr_col, dctr, ld_dctr, dctr_col, dscales, ws, stream);
}

extern "C" int sod_fcos_finalize_losses(const float* focal_sum, const float* regctr_sums, const float* stats,
                                        float inv_world, float* out3, void* stream) {
  if (!focal_sum || !regctr_sums || !stats || !out3) return SOD_EARG;
  SOD_LAUNCH(fcos_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, focal_sum, regctr_sums, stats, inv_world, out3);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}
