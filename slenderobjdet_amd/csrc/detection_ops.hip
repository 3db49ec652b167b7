// Detection post-/mid-processing operators the reference reaches through detectron2 / torchvision / fvcore
// (sources absent from the reference tree; semantics restated in SURVEY.md Appendix C.3, C.4, C.5, C.12, C.13, C.14):
//   nms                      torchvision.ops.nms via detectron2.layers.batched_nms (call sites fcosv2.py:241, rpd.py:781,
//                            proposal_utils.py:115, roi_heads/fast_rcnn.py:103)
//   roi_align fwd/bwd        detectron2 ROIAlign(aligned=True) used by ROIPooler (roi_heads/roi_heads.py:48-53)
//   roi_align_rotated fwd    detectron2 ROIAlignRotated (configs/rotated/Base-RRCNN-FPN.yaml:31-36)
//   giou / smooth-L1         fvcore.nn.giou_loss / smooth_l1_loss (retina_rotated.py:229,240; rpd.py:389-395)
//   anchor labelling         pairwise_iou + Matcher(thresholds, labels, allow_low_quality_matches) (retina_rotated.py:251-295)
// All are HBM/latency-bound gather kernels: coalesced 16-B channel vectors, wavefront reductions, no MFMA.
#include "common.h"
#include "../../include/slender_hip.h"
#include <stdlib.h>

namespace {

// ---------------------------------------------------------------------------------------------- NMS
// boxes are given in descending-score order; mask[i][w] bit b = IoU(box i, box 64*w+b) > thr, only for j > i.
__device__ __forceinline__ bool iou_gt(const float* a, const float* b, float thr) {
  const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
  const float inter = width * height;
  const float sa = (a[2] - a[0]) * (a[3] - a[1]);
  const float sb = (b[2] - b[0]) * (b[3] - b[1]);
  return inter / (sa + sb - inter) > thr;
}

// ---- rotated boxes (cx, cy, w, h, angle_deg): detectron2 box_iou_rotated ----
struct P2 { float x, y; };
__device__ __forceinline__ P2 psub(P2 a, P2 b) { return P2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ float pcross(P2 a, P2 b) { return a.x * b.y - b.x * a.y; }
__device__ __forceinline__ float pdot(P2 a, P2 b) { return a.x * b.x + a.y * b.y; }

__device__ __forceinline__ void rot_vertices(float cx, float cy, float w, float h, float cs, float sn, P2* pts) {
  const float c2 = cs * 0.5f, s2 = sn * 0.5f;      // cs / sn = cosf / sinf of the angle in radians (computed once by the caller)
  pts[0] = P2{cx + s2 * h + c2 * w, cy + c2 * h - s2 * w};
  pts[1] = P2{cx - s2 * h + c2 * w, cy - c2 * h - s2 * w};
  pts[2] = P2{2.f * cx - pts[0].x, 2.f * cy - pts[0].y};
  pts[3] = P2{2.f * cx - pts[1].x, 2.f * cy - pts[1].y};
}

// Where the up-to-24 candidate points of the clipping live.  A private array is indexed dynamically and therefore sits in SCRATCH memory
// (400 B per lane): the bubble sort and the Graham scan below then make a few hundred trips to memory per box pair - measured ~50 000 cycles
// per pair and lane.  The hot kernels hand in a slice of LDS instead (point k of thread t at [k * stride + t]).
struct RotPtsPrivate {
  P2 v[24];
  __device__ __forceinline__ P2 get(int i) const { return v[i]; }
  __device__ __forceinline__ void set(int i, P2 p) { v[i] = p; }
};
struct RotPtsLds {
  P2* base; int stride;
  __device__ __forceinline__ P2 get(int i) const { return base[i * stride]; }
  __device__ __forceinline__ void set(int i, P2 p) { base[i * stride] = p; }
};

// detectron2's rotated-box intersection (box_iou_rotated_utils.h), the same candidate points, the same bubble sort by polar angle and the
// same Graham scan in the same order - the result must match the reference's to the bit wherever a threshold decides a label or a keep.
template <class PTS>
__device__ __forceinline__ float rot_intersection_area(const P2* p1, const P2* p2, PTS& q) {
  int num = 0;
  P2 v1[4], v2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { v1[i] = psub(p1[(i + 1) & 3], p1[i]); v2[i] = psub(p2[(i + 1) & 3], p2[i]); }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float det = pcross(v2[j], v1[i]);
      if (fabsf(det) <= 1e-14f) continue;
      const P2 v12 = psub(p2[j], p1[i]);
      const float t1 = pcross(v2[j], v12) / det, t2 = pcross(v1[i], v12) / det;
      if (t1 >= 0.f && t1 <= 1.f && t2 >= 0.f && t2 <= 1.f) q.set(num++, P2{p1[i].x + v1[i].x * t1, p1[i].y + v1[i].y * t1});
    }
  {
    const P2 AB = v2[0], DA = v2[3];
    const float ABAB = pdot(AB, AB), ADAD = pdot(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const P2 AP = psub(p1[i], p2[0]);
      const float apab = pdot(AP, AB), apad = -pdot(AP, DA);
      if (apab >= 0.f && apad >= 0.f && apab <= ABAB && apad <= ADAD) q.set(num++, p1[i]);
    }
  }
  {
    const P2 AB = v1[0], DA = v1[3];
    const float ABAB = pdot(AB, AB), ADAD = pdot(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const P2 AP = psub(p2[i], p1[0]);
      const float apab = pdot(AP, AB), apad = -pdot(AP, DA);
      if (apab >= 0.f && apad >= 0.f && apab <= ABAB && apad <= ADAD) q.set(num++, p2[i]);
    }
  }
  if (num <= 2) return 0.f;
  // Graham scan
  int t = 0;
  {
    P2 best = q.get(0);
    for (int i = 1; i < num; ++i) {
      const P2 c = q.get(i);
      if (c.y < best.y || (c.y == best.y && c.x < best.x)) { t = i; best = c; }
    }
    // q[i] = inter[i] - start, in place, then q[0] <-> q[t]
    for (int i = 0; i < num; ++i) q.set(i, psub(q.get(i), best));
    const P2 tmp = q.get(0); q.set(0, q.get(t)); q.set(t, tmp);
  }
  for (int i = 1; i < num - 1; ++i) {        // bubble sort by polar angle around q[0] (as the CUDA path of detectron2)
    P2 cur = q.get(1);
    for (int j = 1; j < num - i; ++j) {
      const P2 nxt = q.get(j + 1);
      const float c = pcross(cur, nxt);
      const bool swap = (c < -1e-6f) || (fabsf(c) < 1e-6f && pdot(cur, cur) > pdot(nxt, nxt));
      if (swap) { q.set(j, nxt); }            // cur moves up to j + 1
      else { q.set(j, cur); cur = nxt; }
    }
    q.set(num - i, cur);
  }
  int k = 1;
  for (; k < num; ++k) {
    const P2 c = q.get(k);
    if (pdot(c, c) > 1e-8f) break;
  }
  if (k == num) return 0.f;
  q.set(1, q.get(k));
  int m = 2;
  for (int i = k + 1; i < num; ++i) {
    const P2 qi = q.get(i);
    while (m > 1) {
      const P2 b2 = q.get(m - 2);
      if (pcross(psub(qi, b2), psub(q.get(m - 1), b2)) >= 0.f) --m; else break;
    }
    q.set(m++, qi);
  }
  if (m <= 2) return 0.f;
  float area = 0.f;
  const P2 q0 = q.get(0);
  P2 prev = q.get(1);
  for (int i = 1; i < m - 1; ++i) {
    const P2 nx = q.get(i + 1);
    area += fabsf(pcross(psub(prev, q0), psub(nx, q0)));
    prev = nx;
  }
  return area / 2.f;
}

template <class PTS>
__device__ __forceinline__ float iou_rotated_impl(const float* a, const float* b, PTS& pts) {
  const float area1 = a[2] * a[3], area2 = b[2] * b[3];
  if (area1 < 1e-14f || area2 < 1e-14f) return 0.f;
  {   // disjoint circumscribed circles => empty intersection => IoU exactly 0 (skips the polygon clipping for almost every pair)
    const float dx = a[0] - b[0], dy = a[1] - b[1];
    const float ra = 0.5f * sqrtf(a[2] * a[2] + a[3] * a[3]), rb = 0.5f * sqrtf(b[2] * b[2] + b[3] * b[3]);
    const float rs = ra + rb;
    if (dx * dx + dy * dy > rs * rs * 1.0001f) return 0.f;
  }
  const float tha = a[4] * 0.01745329251994329577f, thb = b[4] * 0.01745329251994329577f;
  const float ca = cosf(tha), sa = sinf(tha), cb = cosf(thb), sb = sinf(thb);
  {   // Separating-axis test on the four face normals (w axis (cos, -sin), h axis (sin, cos) as in rot_vertices): rectangles separated by a
      // margin have no edge crossing and no contained vertex, so the clipping below returns EXACTLY 0 - provided its own arithmetic cannot
      // invent a crossing: a computed crossing point is off by ~eps * |p2 - p1| / sin(angle between the edges), which stays below the
      // margin unless the edges are within ~1 degree of parallel.  Hence only for boxes at least a pixel thick whose axes are more than
      // ~3 degrees from parallel / perpendicular; everything else takes the full computation as before.
    const float c = ca * cb + sa * sb, s2 = sa * cb - ca * sb;        // cos / sin of (angle a - angle b)
    const float ac = fabsf(c), as = fabsf(s2);
    if (ac > 0.05f && as > 0.05f && fminf(fminf(a[2], a[3]), fminf(b[2], b[3])) >= 1.f) {
      const float dx = b[0] - a[0], dy = b[1] - a[1];
      const float m = 0.05f + 1e-4f * (a[2] + a[3] + b[2] + b[3]);
      const float hwa = 0.5f * a[2], hha = 0.5f * a[3], hwb = 0.5f * b[2], hhb = 0.5f * b[3];
      if (fabsf(dx * ca - dy * sa) > hwa + hwb * ac + hhb * as + m) return 0.f;
      if (fabsf(dx * sa + dy * ca) > hha + hwb * as + hhb * ac + m) return 0.f;
      if (fabsf(dx * cb - dy * sb) > hwb + hwa * ac + hha * as + m) return 0.f;
      if (fabsf(dx * sb + dy * cb) > hhb + hwa * as + hha * ac + m) return 0.f;
    }
  }
  const float sx = (a[0] + b[0]) / 2.f, sy = (a[1] + b[1]) / 2.f;   // centre shift for precision
  P2 p1[4], p2[4];
  rot_vertices(a[0] - sx, a[1] - sy, a[2], a[3], ca, sa, p1);
  rot_vertices(b[0] - sx, b[1] - sy, b[2], b[3], cb, sb, p2);
  const float inter = rot_intersection_area(p1, p2, pts);
  return inter / (area1 + area2 - inter);
}

__device__ float iou_rotated(const float* a, const float* b) {          // candidate points in a private (scratch) array
  RotPtsPrivate pts;
  return iou_rotated_impl(a, b, pts);
}

// candidate points in LDS: ``lds`` = this thread's first slot of a [24][stride] P2 array shared by the ``stride`` threads of the workgroup
__device__ float iou_rotated_lds(const float* a, const float* b, P2* lds, int stride) {
  RotPtsLds pts{lds, stride};
  return iou_rotated_impl(a, b, pts);
}

// Cheap NECESSARY conditions for IoU(a, b) > thr between two rotated boxes (cx, cy, w, h, angle) - the NMS kernels run the polygon clipping
// only for pairs that pass:
//   * area ratio: inter <= min(area), union >= max(area)  =>  IoU <= min / max;
//   * for thr >= 0.5: IoU > 1/2 means the intersection S covers more than half of EACH box; a rectangle K is convex and centrally
//     symmetric, so a convex subset that misses its centre c lies in a half-plane through c and has at most half of K's area - hence
//     each box's centre lies in the other box (rot_vertices' frame: w axis (cos, -sin), h axis (sin, cos)).
// Both are applied with a 1e-3 slack, far above the rounding of the float IoU they guard, so no pair the full computation would flag
// is dropped (checked against the oracle's keep sets: tests/test_gpu_rcnn.py, tests/test_gpu_detection_ops.py).  (cs = cos / sin of the angles, precomputed per box.)
__device__ __forceinline__ bool rot_pair_may_exceed(const float* a, float ca, float sa, const float* b, float cb, float sb, float thr) {
  // Boxes thinner than a pixel are left to the full computation: detectron2's clipping works with ABSOLUTE tolerances (1e-14, 1e-6, 1e-8
  // on quantities of order 1e6) and returns values unrelated to the true overlap there - which the oracle reproduces and the product must too.
  if (fminf(fminf(a[2], a[3]), fminf(b[2], b[3])) < 1.0f) return true;
  const float a1 = a[2] * a[3], a2 = b[2] * b[3];
  if (fminf(a1, a2) < thr * 0.999f * fmaxf(a1, a2)) return false;
  if (thr >= 0.5f) {
    const float dx = b[0] - a[0], dy = b[1] - a[1];
    if (fabsf(dx * ca - dy * sa) > 0.5005f * a[2] + 1e-3f || fabsf(dx * sa + dy * ca) > 0.5005f * a[3] + 1e-3f) return false;   // centre of b in a
    if (fabsf(dx * cb - dy * sb) > 0.5005f * b[2] + 1e-3f || fabsf(dx * sb + dy * cb) > 0.5005f * b[3] + 1e-3f) return false;   // centre of a in b
  }
  return true;
}

template <int BD>   // BD = 4 axis-aligned XYXY, 5 rotated
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, const long long* __restrict__ order, int n,
                                                      float thr, unsigned long long* __restrict__ mask, int words) {
  const int rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;   // only the upper triangle is ever read
  __shared__ float cbox[64 * BD];
  __shared__ float crad[64];      // rotated: circumscribed-circle radius of each column box
  __shared__ P2 rot_pts[BD == 5 ? 24 * 64 : 1];
  const int lane = threadIdx.x;
  const int cj = cb * 64 + lane;
  if (cj < n) {
    const long long o = order[cj];
#pragma unroll
    for (int e = 0; e < BD; ++e) cbox[lane * BD + e] = boxes[o * BD + e];
    if (BD == 5) crad[lane] = 0.5f * sqrtf(cbox[lane * BD + 2] * cbox[lane * BD + 2] + cbox[lane * BD + 3] * cbox[lane * BD + 3]);
  }
  __syncthreads();
  const int i = rb * 64 + lane;
  if (i >= n) return;
  float a[BD];
  const long long oi = order[i];
#pragma unroll
  for (int e = 0; e < BD; ++e) a[e] = boxes[oi * BD + e];
  const float ra = BD == 5 ? 0.5f * sqrtf(a[2] * a[2] + a[3] * a[3]) : 0.f;
  unsigned long long bits = 0;
  const int cnt = min(64, n - cb * 64);
  const int start = (rb == cb) ? lane + 1 : 0;
  for (int j = start; j < cnt; ++j) {
    bool hit;
    if (BD == 4) hit = iou_gt(a, cbox + j * BD, thr);
    else {
      const float dx = a[0] - cbox[j * BD], dy = a[1] - cbox[j * BD + 1], rs = ra + crad[j];
      hit = (dx * dx + dy * dy <= rs * rs * 1.0001f) && (iou_rotated_lds(a, cbox + j * BD, rot_pts + lane, 64) > thr);   // disjoint circles: IoU is exactly 0
    }
    if (hit) bits |= 1ull << j;
  }
  mask[(long long)i * words + cb] = bits;
}

__global__ __launch_bounds__(256) void pairwise_iou_rotated_kernel(const float* __restrict__ b1, int n1, const float* __restrict__ b2, int n2,
                                                                   float* __restrict__ out) {
  __shared__ P2 rot_pts[24 * 256];
  const long long total = (long long)n1 * n2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
    out[i] = iou_rotated_lds(b1 + (i / n2) * 5, b2 + (i % n2) * 5, rot_pts + threadIdx.x, 256);
}

// single workgroup, boxes visited in score order in chunks of 64: wave 0 resolves a chunk against the chunk's own 64x64 diagonal
// block of the suppression matrix with wave shuffles (no barrier per box), then thread w ORs the rows of the chunk's survivors
// into word w of the "removed" bitmap.  2 barriers per 64 boxes instead of 2 per box (10 000 candidates: 3.5 ms -> ~0.5 ms).
__global__ __launch_bounds__(1024) void nms_scan_kernel(const unsigned long long* __restrict__ mask, const long long* __restrict__ order,
                                                        int n, int words, long long* __restrict__ keep, int* __restrict__ nkeep) {
  __shared__ unsigned long long removed[1024];
  __shared__ unsigned long long chunk_keep;
  const int w = threadIdx.x;
  if (w < words) removed[w] = 0;
  __syncthreads();
  int kept = 0;
  for (int c = 0; c < words; ++c) {
    const int cnt = min(64, n - c * 64);
    if (w < 64) {
      const long long i = (long long)c * 64 + w;
      const unsigned long long diag = (w < cnt) ? mask[i * words + c] : 0ull;
      const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
      unsigned long long rem = removed[c], kb = 0ull;
      for (int j = 0; j < cnt; ++j) {
        const unsigned long long dj = ((unsigned long long)__shfl(dhi, j, 64) << 32) | (unsigned long long)__shfl(dlo, j, 64);
        if (!((rem >> j) & 1ull)) { kb |= 1ull << j; rem |= dj; }
      }
      if (w == 0) chunk_keep = kb;
    }
    __syncthreads();
    const unsigned long long kb = chunk_keep;
    {   // all 1024 threads: word ww = w % words, row slice sl = w / words of `slices`; independent loads, merged with ds_or_b64
      const int slices = 1024 / words;          // words <= 1024
      const int ww = w % words, sl = w / words;
      if (sl < slices && ww > c) {
        unsigned long long acc = 0ull;
#pragma unroll 4
        for (int j = sl; j < cnt; j += slices)
          if ((kb >> j) & 1ull) acc |= mask[((long long)c * 64 + j) * words + ww];
        if (acc) atomicOr(&removed[ww], acc);
      }
    }
    if (w < 64 && ((kb >> w) & 1ull)) keep[kept + __popcll(kb & ((1ull << w) - 1ull))] = order[(long long)c * 64 + w];
    kept += __popcll(kb);
    __syncthreads();
  }
  if (w == 0) *nkeep = kept;
}

// ---------------------------------------------------------------------------------------------- batched class-aware NMS
// detectron2.layers.batched_nms / batched_nms_rotated + keep[: max_keep] for B images of M candidate slots each, without a host round
// trip: empty slots carry score -inf, the per-image candidate counts stay on the device, and the scan stops after max_keep survivors.
// Class offsets: axis-aligned  boxes + class * (max coordinate of the image + 1)                       (torchvision batched_nms)
//                rotated       centres + class * (max - min + 1), max = max(max(cx, cy) + max(w, h) / 2), min = min(min(cx, cy) - max(w, h) / 2)
template <int BD>
__global__ __launch_bounds__(1024) void nms_class_shift_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                               const int* __restrict__ classes, int M, float* __restrict__ shifted,
                                                               int* __restrict__ nvalid) {
  __shared__ float redmx[16], redmn[16];
  __shared__ unsigned cntw[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* bx = boxes + (long long)b * M * BD;
  const float* sc = scores + (long long)b * M;
  float mx = -3.0e38f, mn = 3.0e38f;
  unsigned cnt = 0;
  for (int i = tid; i < M; i += 1024)
    if (sc[i] > -3.0e38f) {
      ++cnt;
      if (BD == 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) mx = fmaxf(mx, bx[i * 4 + k]);
      } else {
        const float half = fmaxf(bx[i * 5 + 2], bx[i * 5 + 3]) / 2.f;
        mx = fmaxf(mx, fmaxf(bx[i * 5], bx[i * 5 + 1]) + half);
        mn = fminf(mn, fminf(bx[i * 5], bx[i * 5 + 1]) - half);
      }
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o, 64)); mn = fminf(mn, __shfl_xor(mn, o, 64)); cnt += __shfl_xor(cnt, o, 64); }
  if ((tid & 63) == 0) { redmx[tid >> 6] = mx; redmn[tid >> 6] = mn; cntw[tid >> 6] = cnt; }
  __syncthreads();
  mx = redmx[0]; mn = redmn[0]; cnt = cntw[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) { mx = fmaxf(mx, redmx[w]); mn = fminf(mn, redmn[w]); cnt += cntw[w]; }
  const float step = BD == 4 ? mx + 1.f : mx - mn + 1.f;
  for (int i = tid; i < M; i += 1024) {
    const float off = sc[i] > -3.0e38f ? (float)classes[(long long)b * M + i] * step : 0.f;
#pragma unroll
    for (int k = 0; k < BD; ++k) shifted[((long long)b * M + i) * BD + k] = bx[i * BD + k] + ((BD == 4 || k < 2) ? off : 0.f);
  }
  if (tid == 0) nvalid[b] = (int)cnt;
}

// mask[b][i][w] bit j = IoU(box order[i], box order[64 w + j]) > thr for j > i; grid (words, words, B); n read per image
template <int BD>
__global__ __launch_bounds__(64) void nms_mask_batched_kernel(const float* __restrict__ boxes, const long long* __restrict__ order,
                                                              const int* __restrict__ nvalid, int M, float thr,
                                                              unsigned long long* __restrict__ mask, int words) {
  const int b = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x;
  const int n = min(nvalid[b], M);
  if (cb < rb || rb * 64 >= n || cb * 64 >= n) return;
  boxes += (long long)b * M * BD; order += (long long)b * M; mask += (long long)b * M * words;
  __shared__ float cbox[64 * BD];
  __shared__ float crad[64];
  const int lane = threadIdx.x;
  const int cj = cb * 64 + lane;
  if (cj < n) {
    const long long o = order[cj];
#pragma unroll
    for (int e = 0; e < BD; ++e) cbox[lane * BD + e] = boxes[o * BD + e];
    if (BD == 5) crad[lane] = 0.5f * sqrtf(cbox[lane * BD + 2] * cbox[lane * BD + 2] + cbox[lane * BD + 3] * cbox[lane * BD + 3]);
  }
  __syncthreads();
  const int i = rb * 64 + lane;
  const int cnt = min(64, n - cb * 64);
  if constexpr (BD == 4) {
    if (i >= n) return;
    float a[BD];
    const long long oi = order[i];
#pragma unroll
    for (int e = 0; e < BD; ++e) a[e] = boxes[oi * BD + e];
    unsigned long long bits = 0;
    for (int j = (rb == cb) ? lane + 1 : 0; j < cnt; ++j)
      if (iou_gt(a, cbox + j * BD, thr)) bits |= 1ull << j;
    mask[(long long)i * words + cb] = bits;
  } else {
    // Rotated boxes: the polygon-clipping IoU costs hundreds of instructions and only the few pairs that pass the circle test need it.
    // Looping "for j: if (near) iou" makes the WAVE pay it for every column some lane is near (RPN proposals of one level overlap
    // heavily: nearly all 64 columns, 9.1 ms per step for 16 x 10 000 candidates).  Instead: circle test for all 64 x 64 pairs (no
    // divergence), the surviving pairs compacted through LDS and dealt out evenly over the lanes, results OR-ed into the rows' words.
    __shared__ float rbox[64 * BD];
    __shared__ P2 rot_pts[24 * 64];
    __shared__ float ccs[64 * 2];
    __shared__ unsigned short pairs[64 * 64];
    __shared__ unsigned long long rbits[64];
    __shared__ int total_pairs;
    float a[BD] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < n) {
      const long long oi = order[i];
#pragma unroll
      for (int e = 0; e < BD; ++e) { a[e] = boxes[oi * BD + e]; rbox[lane * BD + e] = a[e]; }
    }
    rbits[lane] = 0ull;
    if (cj < n) {
      const float th = cbox[lane * BD + 4] * 0.01745329251994329577f;
      ccs[lane * 2] = cosf(th); ccs[lane * 2 + 1] = sinf(th);
    }
    __syncthreads();
    const float ra = 0.5f * sqrtf(a[2] * a[2] + a[3] * a[3]);
    const float tha = a[4] * 0.01745329251994329577f, ca = cosf(tha), sa = sinf(tha);
    unsigned long long near = 0ull;
    if (i < n)
      for (int j = (rb == cb) ? lane + 1 : 0; j < cnt; ++j) {
        const float dx = a[0] - cbox[j * BD], dy = a[1] - cbox[j * BD + 1], rs = ra + crad[j];
        if (dx * dx + dy * dy <= rs * rs * 1.0001f && rot_pair_may_exceed(a, ca, sa, cbox + j * BD, ccs[j * 2], ccs[j * 2 + 1], thr)) near |= 1ull << j;
      }
    // exclusive prefix sum of the per-row pair counts over the wave
    const int mine = __popcll(near);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) total_pairs = incl;
    int off = incl - mine;
    for (unsigned long long m = near; m; m &= m - 1ull) pairs[off++] = (unsigned short)((lane << 6) | __builtin_ctzll(m));
    __syncthreads();
    const int total = total_pairs;
    for (int t = lane; t < total; t += 64) {
      const int pr = pairs[t];
      const int r = pr >> 6, j = pr & 63;
      float ar[BD];
#pragma unroll
      for (int e = 0; e < BD; ++e) ar[e] = rbox[r * BD + e];
      if (iou_rotated_lds(ar, cbox + j * BD, rot_pts + lane, 64) > thr) atomicOr(&rbits[r], 1ull << j);
    }
    __syncthreads();
    if (i < n) mask[(long long)i * words + cb] = rbits[lane];
  }
}

// one workgroup per image; nms_scan_kernel with the count read from device memory and an early stop at max_keep survivors
__global__ __launch_bounds__(1024) void nms_scan_batched_kernel(const unsigned long long* __restrict__ mask, const long long* __restrict__ order,
                                                                const int* __restrict__ nvalid, int M, int words, int max_keep,
                                                                long long* __restrict__ keep, int* __restrict__ nkeep) {
  __shared__ unsigned long long removed[1024];
  __shared__ unsigned long long chunk_keep;
  const int b = blockIdx.x, w = threadIdx.x;
  const int n = min(nvalid[b], M);
  mask += (long long)b * M * words; order += (long long)b * M; keep += (long long)b * max_keep;
  const int nw = (n + 63) / 64;
  if (w < words) removed[w] = 0;
  __syncthreads();
  int kept = 0;
  for (int c = 0; c < nw && kept < max_keep; ++c) {
    const int cnt = min(64, n - c * 64);
    if (w < 64) {
      const long long i = (long long)c * 64 + w;
      const unsigned long long diag = (w < cnt) ? mask[i * words + c] : 0ull;
      const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
      unsigned long long rem = removed[c], kb = 0ull;
      for (int j = 0; j < cnt; ++j) {
        const unsigned long long dj = ((unsigned long long)__shfl(dhi, j, 64) << 32) | (unsigned long long)__shfl(dlo, j, 64);
        if (!((rem >> j) & 1ull)) { kb |= 1ull << j; rem |= dj; }
      }
      if (w == 0) chunk_keep = kb;
    }
    __syncthreads();
    const unsigned long long kb = chunk_keep;
    {
      const int slices = 1024 / words;
      const int ww = w % words, sl = w / words;
      if (sl < slices && ww > c && ww < nw) {
        unsigned long long acc = 0ull;
        for (int j = sl; j < cnt; j += slices)
          if ((kb >> j) & 1ull) acc |= mask[((long long)c * 64 + j) * words + ww];
        if (acc) atomicOr(&removed[ww], acc);
      }
    }
    if (w < 64 && ((kb >> w) & 1ull)) {
      const int pos = kept + __popcll(kb & ((1ull << w) - 1ull));
      if (pos < max_keep) keep[pos] = order[(long long)c * 64 + w];
    }
    kept += __popcll(kb);
    __syncthreads();
  }
  if (w == 0) nkeep[b] = kept < max_keep ? kept : max_keep;
}

// The per-image glue of find_top_rpn_proposals (detectron2 proposal_utils; reference copy slender_det/modeling/proposal_generator/
// proposal_utils.py:45-120) for the whole batch: drop non-finite entries, clip to the image, drop boxes not larger than min_size.
// Dropped slots get score -inf (= empty for the batched NMS); the number of non-finite entries is counted in *bad.
template <int BD>
__global__ __launch_bounds__(256) void rpn_clip_filter_kernel(float* __restrict__ boxes, float* __restrict__ scores, const float* __restrict__ image_hw,
                                                              int B, int M, float min_size, int* __restrict__ bad) {
  const long long total = (long long)B * M;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / M);
    float* q = boxes + i * BD;
    const float h = image_hw[2 * b], w = image_hw[2 * b + 1];
    bool fin = isfinite(scores[i]);
#pragma unroll
    for (int k = 0; k < BD; ++k) fin = fin && isfinite(q[k]);
    if (!fin) { atomicAdd(bad, 1); scores[i] = -__builtin_inff(); continue; }
    bool keep;
    if (BD == 4) {
      q[0] = fminf(fmaxf(q[0], 0.f), w); q[1] = fminf(fmaxf(q[1], 0.f), h);
      q[2] = fminf(fmaxf(q[2], 0.f), w); q[3] = fminf(fmaxf(q[3], 0.f), h);
      keep = (q[2] - q[0] > min_size) && (q[3] - q[1] > min_size);
    } else {
      float ang = fmodf(q[4] + 180.0f, 360.0f);          // RotatedBoxes.normalize_angles: (a + 180) % 360 - 180, Python modulo
      if (ang < 0.f) ang += 360.0f;
      q[4] = ang - 180.0f;
      if (fabsf(q[4]) <= 1.0f) {                          // RotatedBoxes.clip: only nearly horizontal boxes
        float x1 = q[0] - q[2] / 2.0f, y1 = q[1] - q[3] / 2.0f, x2 = q[0] + q[2] / 2.0f, y2 = q[1] + q[3] / 2.0f;
        x1 = fminf(fmaxf(x1, 0.f), w); y1 = fminf(fmaxf(y1, 0.f), h); x2 = fminf(fmaxf(x2, 0.f), w); y2 = fminf(fmaxf(y2, 0.f), h);
        q[0] = (x1 + x2) / 2.0f; q[1] = (y1 + y2) / 2.0f;
        q[2] = fminf(q[2], x2 - x1); q[3] = fminf(q[3], y2 - y1);
      }
      keep = (q[2] > min_size) && (q[3] > min_size);
    }
    if (!keep) scores[i] = -__builtin_inff();
  }
}

// ---------------------------------------------------------------------------------------------- ROIAlign (aligned=True)
struct RoiArgs {
  const __bf16* x;      // (N,H,W,C) NHWC bf16
  const float* rois;    // (R,5) [batch, x1,y1,x2,y2]  or (R,6) [batch, cx,cy,w,h,angle_deg] when rotated
  int R, N, H, W, C, PH, PW, sampling_ratio;
  float scale;
  int rotated;
};

__device__ __forceinline__ void bilinear_setup(float y, float x, int H, int W, int& yl, int& xl, int& yh, int& xh, float w[4], bool& valid) {
  valid = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  yl = (int)y; xl = (int)x;
  if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
  if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
  const float ly = y - yl, lx = x - xl, hy = 1.f - ly, hx = 1.f - lx;
  w[0] = hy * hx; w[1] = hy * lx; w[2] = ly * hx; w[3] = ly * lx;
}

// one thread = one (roi, ph, pw, 8-channel vector); XF32: the features are fp32 (validation mode, SOD_PRECISION=fp32)
template <bool BWD, bool XF32 = false>
__global__ __launch_bounds__(256) void roi_align_kernel(const RoiArgs a, float* __restrict__ out /* fwd (R,PH,PW,C) f32 */,
                                                        const float* __restrict__ dout, float* __restrict__ dx /* (N,H,W,C) f32 */) {
  const int c8n = a.C >> 3;
  const long long total = (long long)a.R * a.PH * a.PW * c8n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c8 = (int)(i % c8n);
    long long t = i / c8n;
    const int pw = (int)(t % a.PW); t /= a.PW;
    const int ph = (int)(t % a.PH);
    const int r = (int)(t / a.PH);
    const float* roi = a.rois + (long long)r * (a.rotated ? 6 : 5);
    const int b = (int)roi[0];
    float start_h, start_w, roi_h, roi_w, cth = 1.f, sth = 0.f, ctr_h = 0.f, ctr_w = 0.f;
    if (a.rotated) {
      ctr_w = roi[1] * a.scale - 0.5f; ctr_h = roi[2] * a.scale - 0.5f;
      roi_w = roi[3] * a.scale; roi_h = roi[4] * a.scale;
      const float theta = roi[5] * 3.14159265358979323846f / 180.0f;
      cth = cosf(theta); sth = sinf(theta);
      start_h = -roi_h / 2.0f; start_w = -roi_w / 2.0f;
    } else {
      start_w = roi[1] * a.scale - 0.5f; start_h = roi[2] * a.scale - 0.5f;
      roi_w = roi[3] * a.scale - 0.5f - start_w; roi_h = roi[4] * a.scale - 0.5f - start_h;
    }
    const float bin_h = roi_h / (float)a.PH, bin_w = roi_w / (float)a.PW;
    const int gh = a.sampling_ratio > 0 ? a.sampling_ratio : (int)ceilf(roi_h / (float)a.PH);
    const int gw = a.sampling_ratio > 0 ? a.sampling_ratio : (int)ceilf(roi_w / (float)a.PW);
    const float count = fmaxf((float)(gh * gw), 1.f);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float g[8];
    if (BWD) {
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = dout[i * 8 + e] / count;
    }
    for (int iy = 0; iy < gh; ++iy) {
      const float yy = start_h + ph * bin_h + (iy + 0.5f) * bin_h / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        const float xx = start_w + pw * bin_w + (ix + 0.5f) * bin_w / (float)gw;
        float y = yy, x = xx;
        if (a.rotated) { y = yy * cth - xx * sth + ctr_h; x = yy * sth + xx * cth + ctr_w; }
        int yl, xl, yh, xh; float w[4]; bool valid;
        bilinear_setup(y, x, a.H, a.W, yl, xl, yh, xh, w, valid);
        if (!valid) continue;
        const long long base = ((long long)b * a.H) * a.W;
        const long long o00 = ((base + (long long)yl * a.W + xl) * a.C) + c8 * 8, o01 = ((base + (long long)yl * a.W + xh) * a.C) + c8 * 8;
        const long long o10 = ((base + (long long)yh * a.W + xl) * a.C) + c8 * 8, o11 = ((base + (long long)yh * a.W + xh) * a.C) + c8 * 8;
        if (!BWD && XF32) {
          const float* xf = reinterpret_cast<const float*>(a.x);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += w[0] * xf[o00 + e] + w[1] * xf[o01 + e] + w[2] * xf[o10 + e] + w[3] * xf[o11 + e];
        } else if (!BWD) {
          const bf16x8_t v00 = *reinterpret_cast<const bf16x8_t*>(a.x + o00), v01 = *reinterpret_cast<const bf16x8_t*>(a.x + o01);
          const bf16x8_t v10 = *reinterpret_cast<const bf16x8_t*>(a.x + o10), v11 = *reinterpret_cast<const bf16x8_t*>(a.x + o11);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += w[0] * (float)v00[e] + w[1] * (float)v01[e] + w[2] * (float)v10[e] + w[3] * (float)v11[e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            atomicAdd(dx + o00 + e, g[e] * w[0]); atomicAdd(dx + o01 + e, g[e] * w[1]);
            atomicAdd(dx + o10 + e, g[e] * w[2]); atomicAdd(dx + o11 + e, g[e] * w[3]);
          }
        }
      }
    }
    if (!BWD) {
#pragma unroll
      for (int e = 0; e < 8; ++e) out[i * 8 + e] = acc[e] / count;
    }
  }
}

// Tiled ROIAlign backward: one workgroup per (roi, 32-channel chunk) accumulates dX for the roi's footprint in 16x16-pixel LDS
// tiles in fixed point (ds_add_u32, see deform_conv.hip for the measurement behind that choice) and flushes each tile with one
// global atomic per touched element, instead of 4 corners x samples x channels global atomics per bin.  Scale: an element
// receives at most sum_bins |g_bin| <= 64 * max|g| (the bilinear weights of a bin's samples, divided by the sample count, sum to
// <= 1), so 2^(24 - ex) with max|g| < 2^ex cannot overflow 31 bits.
constexpr int ROI_T = 16;    // tile edge (pixels)
constexpr int ROI_CC = 32;   // channels per workgroup
__global__ __launch_bounds__(256) void roi_align_bwd_tile_kernel(const RoiArgs a, const float* __restrict__ dout, float* __restrict__ dx) {
  constexpr int PS = ROI_CC + 1, L = ROI_CC / 8;
  __shared__ int win[ROI_T * ROI_T * PS];
  __shared__ float smax[4];
  const int tid = threadIdx.x, r = blockIdx.x, c0 = blockIdx.y * ROI_CC;
  const int cl = tid % L, bl = tid / L, nb = a.PH * a.PW;
  const float* roi = a.rois + (long long)r * (a.rotated ? 6 : 5);
  const int b = (int)roi[0];
  float start_h, start_w, roi_h, roi_w, cth = 1.f, sth = 0.f, ctr_h = 0.f, ctr_w = 0.f;
  float fx0, fx1, fy0, fy1;   // footprint bounds (feature coordinates) of all sample points
  if (a.rotated) {
    ctr_w = roi[1] * a.scale - 0.5f; ctr_h = roi[2] * a.scale - 0.5f;
    roi_w = roi[3] * a.scale; roi_h = roi[4] * a.scale;
    const float theta = roi[5] * 3.14159265358979323846f / 180.0f;
    cth = cosf(theta); sth = sinf(theta);
    start_h = -roi_h / 2.0f; start_w = -roi_w / 2.0f;
    const float ex = 0.5f * (fabsf(roi_w * cth) + fabsf(roi_h * sth)), ey = 0.5f * (fabsf(roi_w * sth) + fabsf(roi_h * cth));
    fx0 = ctr_w - ex; fx1 = ctr_w + ex; fy0 = ctr_h - ey; fy1 = ctr_h + ey;
  } else {
    start_w = roi[1] * a.scale - 0.5f; start_h = roi[2] * a.scale - 0.5f;
    roi_w = roi[3] * a.scale - 0.5f - start_w; roi_h = roi[4] * a.scale - 0.5f - start_h;
    fx0 = fminf(start_w, start_w + roi_w); fx1 = fmaxf(start_w, start_w + roi_w);
    fy0 = fminf(start_h, start_h + roi_h); fy1 = fmaxf(start_h, start_h + roi_h);
  }
  const float bin_h = roi_h / (float)a.PH, bin_w = roi_w / (float)a.PW;
  const int gh = a.sampling_ratio > 0 ? a.sampling_ratio : (int)ceilf(roi_h / (float)a.PH);
  const int gw = a.sampling_ratio > 0 ? a.sampling_ratio : (int)ceilf(roi_w / (float)a.PW);
  const float count = fmaxf((float)(gh * gw), 1.f);
  // pixel window that can receive anything (samples outside [-1, H] x [-1, W] are dropped; the rest clamp into the map)
  const int wx0 = max(0, (int)floorf(fmaxf(fx0, -1.f))), wx1 = min(a.W - 1, (int)floorf(fminf(fx1, (float)a.W)) + 1);
  const int wy0 = max(0, (int)floorf(fmaxf(fy0, -1.f))), wy1 = min(a.H - 1, (int)floorf(fminf(fy1, (float)a.H)) + 1);
  if (wx1 < wx0 || wy1 < wy0 || gh <= 0 || gw <= 0) return;
  // fixed-point scale from max |g| of this (roi, chunk)
  float gmax = 0.f;
  for (int bin = bl; bin < nb; bin += 256 / L) {
    const float* gp = dout + ((long long)r * nb + bin) * a.C + c0 + cl * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) gmax = fmaxf(gmax, fabsf(gp[e]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o, 64));
  if ((tid & 63) == 0) smax[tid >> 6] = gmax;
  __syncthreads();
  gmax = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
  if (gmax == 0.f || !(gmax < 3.0e38f)) return;   // nothing to add / non-finite gradient (the loss is already non-finite then)
  int ex = 0;
  (void)frexpf(gmax, &ex);
  const float S = ldexpf(1.f, 24 - ex) / count, invS = ldexpf(1.f, ex - 24);
  const long long base = ((long long)b * a.H) * a.W;
  for (int ty = wy0; ty <= wy1; ty += ROI_T)
    for (int tx = wx0; tx <= wx1; tx += ROI_T) {
      for (int i = tid; i < ROI_T * ROI_T * PS; i += 256) win[i] = 0;
      __syncthreads();
      // work item = one ROW of a bin's sample grid (bin, iy), dealt out over the 256 / L lane groups.  A row whose bounding box (+2 px: the
      // bilinear neighbour and the clamp into the map) misses this tile is skipped before anything is loaded: with the adaptive sampling
      // ratio (ceil(roi / 7) samples per bin side) a large ROI has ~1 sample per feature pixel, and visiting every sample for every
      // 16x16 tile of its footprint made the work quadratic in the ROI's area (3.4 ms per step for the rotated R-CNN's 8192 ROIs).
      for (int item = bl; item < nb * gh; item += 256 / L) {
        const int bin = item / gh, iy = item - bin * gh;
        const int ph = bin / a.PW, pw = bin - ph * a.PW;
        const float yy = start_h + ph * bin_h + (iy + 0.5f) * bin_h / (float)gh;
        {
          const float xa = start_w + pw * bin_w + 0.5f * bin_w / (float)gw, xb = start_w + pw * bin_w + ((float)gw - 0.5f) * bin_w / (float)gw;
          float ya0 = yy, xa0 = xa, yb0 = yy, xb0 = xb;
          if (a.rotated) { ya0 = yy * cth - xa * sth + ctr_h; xa0 = yy * sth + xa * cth + ctr_w; yb0 = yy * cth - xb * sth + ctr_h; xb0 = yy * sth + xb * cth + ctr_w; }
          if (fmaxf(xa0, xb0) + 2.f < (float)tx || fminf(xa0, xb0) - 2.f > (float)(tx + ROI_T - 1) ||
              fmaxf(ya0, yb0) + 2.f < (float)ty || fminf(ya0, yb0) - 2.f > (float)(ty + ROI_T - 1)) continue;
        }
        const float* gp = dout + ((long long)r * nb + bin) * a.C + c0 + cl * 8;
        float g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = gp[e] * S;
        {
          for (int ix = 0; ix < gw; ++ix) {
            const float xx = start_w + pw * bin_w + (ix + 0.5f) * bin_w / (float)gw;
            float y = yy, x = xx;
            if (a.rotated) { y = yy * cth - xx * sth + ctr_h; x = yy * sth + xx * cth + ctr_w; }
            int yl, xl, yh, xh; float w[4]; bool valid;
            bilinear_setup(y, x, a.H, a.W, yl, xl, yh, xh, w, valid);
            if (!valid) continue;
            const int ys[2] = {yl - ty, yh - ty}, xs[2] = {xl - tx, xh - tx};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int py = ys[k >> 1], px = xs[k & 1];
              if (py < 0 || py >= ROI_T || px < 0 || px >= ROI_T) continue;
              int* wp = win + (py * ROI_T + px) * PS + cl * 8;
#pragma unroll
              for (int e = 0; e < 8; ++e) atomicAdd(wp + e, __float2int_rn(g[e] * w[k]));
            }
          }
        }
      }
      __syncthreads();
      for (int i = tid; i < ROI_T * ROI_T * ROI_CC; i += 256) {
        const int c = i % ROI_CC, p = i / ROI_CC;
        const int q = win[p * PS + c];
        if (q != 0) {
          const int py = ty + p / ROI_T, px = tx + p % ROI_T;
          atomicAdd(dx + ((base + (long long)py * a.W + px) * a.C) + c0 + c, (float)q * invS);
        }
      }
      __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------- box losses on XYXY
constexpr int RED = 1024;
__global__ void finish_sum2(const float* __restrict__ part, int nblk, float* __restrict__ out) {
  __shared__ float red[4];
  float v = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) v += part[i];
  v = block_sum_256(v, red);
  if (threadIdx.x == 0) out[0] = v;
}

// fvcore giou_loss (eps 1e-7): per-row loss and gradient w.r.t. boxes1
__global__ __launch_bounds__(256) void giou_xyxy_kernel(const float* __restrict__ b1, const float* __restrict__ b2, long long P, float eps,
                                                        float* __restrict__ elem, float* __restrict__ part, const float* __restrict__ gscale,
                                                        float* __restrict__ d1) {
  __shared__ float red[4];
  float acc = 0.f;
  const float gs = gscale ? gscale[0] : 1.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long long)gridDim.x * 256) {
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(b1 + i * 4), b = *reinterpret_cast<const f32x4_t*>(b2 + i * 4);
    const float x1 = a[0], y1 = a[1], x2 = a[2], y2 = a[3], x1g = b[0], y1g = b[1], x2g = b[2], y2g = b[3];
    const float xk1 = fmaxf(x1, x1g), yk1 = fmaxf(y1, y1g), xk2 = fminf(x2, x2g), yk2 = fminf(y2, y2g);
    const bool ov = (yk2 > yk1) && (xk2 > xk1);
    const float inter = ov ? (xk2 - xk1) * (yk2 - yk1) : 0.f;
    const float uni = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter;
    const float iou = inter / (uni + eps);
    const float xc1 = fminf(x1, x1g), yc1 = fminf(y1, y1g), xc2 = fmaxf(x2, x2g), yc2 = fmaxf(y2, y2g);
    const float ac = (xc2 - xc1) * (yc2 - yc1);
    const float l = 1.f - (iou - (ac - uni) / (ac + eps));
    if (elem) elem[i] = l;
    acc += l;
    if (d1) {
      // derivatives of inter / union / hull w.r.t. (x1,y1,x2,y2) of box 1; max/min ties split like torch
      auto dmax = [](float p, float q) { return p > q ? 1.f : (p == q ? 0.5f : 0.f); };
      auto dmin = [](float p, float q) { return p < q ? 1.f : (p == q ? 0.5f : 0.f); };
      const float iw = xk2 - xk1, ih = yk2 - yk1, w1 = x2 - x1, h1 = y2 - y1, cw = xc2 - xc1, ch = yc2 - yc1;
      const float di[4] = {ov ? -dmax(x1, x1g) * ih : 0.f, ov ? -dmax(y1, y1g) * iw : 0.f, ov ? dmin(x2, x2g) * ih : 0.f, ov ? dmin(y2, y2g) * iw : 0.f};
      const float da[4] = {-h1, -w1, h1, w1};
      const float dc[4] = {-dmin(x1, x1g) * ch, -dmin(y1, y1g) * cw, dmax(x2, x2g) * ch, dmax(y2, y2g) * cw};
      f32x4_t g;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float du = da[e] - di[e];
        const float diou = (di[e] * (uni + eps) - inter * du) / ((uni + eps) * (uni + eps));
        const float dterm = ((dc[e] - du) * (ac + eps) - (ac - uni) * dc[e]) / ((ac + eps) * (ac + eps));
        g[e] = -(diou - dterm) * gs;
      }
      *reinterpret_cast<f32x4_t*>(d1 + i * 4) = g;
    }
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0 && part) part[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void smooth_l1_kernel(const float* __restrict__ x, const float* __restrict__ t, long long n, float beta,
                                                        float* __restrict__ elem, float* __restrict__ part, const float* __restrict__ gscale,
                                                        float* __restrict__ dx) {
  __shared__ float red[4];
  float acc = 0.f;
  const float gs = gscale ? gscale[0] : 1.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float d = x[i] - t[i], ad = fabsf(d);
    float l, g;
    if (beta < 1e-5f) { l = ad; g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
    else if (ad < beta) { l = 0.5f * d * d / beta; g = d / beta; }
    else { l = ad - 0.5f * beta; g = d > 0.f ? 1.f : -1.f; }
    if (elem) elem[i] = l;
    if (dx) dx[i] = g * gs;
    acc += l;
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0 && part) part[blockIdx.x] = acc;
}

// ---------------------------------------------------------------------------------------------- anchor labelling
// Matcher(thresholds=[lo,hi], labels=[l0,l1,l2], allow_low_quality_matches) over pairwise_iou(gt (G), anchors (A)) without the
// G x A matrix: pass 1 = per-anchor max/argmax over gts + per-gt max over anchors (atomicMax on float bits, IoU >= 0);
// pass 2 = thresholds + low-quality promotion (anchor ties with the per-gt best, as `Q == best_per_gt[:, None]`).
__device__ __forceinline__ float pair_iou(const float* g, const float* a) {
  const float w = fminf(g[2], a[2]) - fmaxf(g[0], a[0]), h = fminf(g[3], a[3]) - fmaxf(g[1], a[1]);
  const float inter = fmaxf(w, 0.f) * fmaxf(h, 0.f);
  const float ag = (g[2] - g[0]) * (g[3] - g[1]), aa = (a[2] - a[0]) * (a[3] - a[1]);
  return inter > 0.f ? inter / (ag + aa - inter) : 0.f;
}

// D = 4: XYXY boxes (pairwise_iou); D = 5: (cx, cy, w, h, angle) boxes (pairwise_iou_rotated, RRPN / RROIHeads)
template <int D>
__device__ __forceinline__ float match_iou(const float* g, const float* a) {
  if (D == 4) return pair_iou(g, a);
  return fmaxf(iou_rotated(g, a), 0.f);
}

template <int D>
__global__ __launch_bounds__(256) void anchor_match1_kernel(const float* __restrict__ gts, int G, const float* __restrict__ anchors, int A,
                                                            float* __restrict__ best_val, int* __restrict__ best_idx,
                                                            unsigned* __restrict__ gt_best_bits) {
  extern __shared__ unsigned lbest[];   // [G]
  __shared__ P2 rot_pts[D == 5 ? 24 * 256 : 1];
  for (int g = threadIdx.x; g < G; g += 256) lbest[g] = 0u;
  __syncthreads();
  for (int i = blockIdx.x * 256 + threadIdx.x; i < A; i += gridDim.x * 256) {
    float a[D];
#pragma unroll
    for (int e = 0; e < D; ++e) a[e] = anchors[(long long)i * D + e];
    float bv = -1.f; int bi = 0;
    for (int g = 0; g < G; ++g) {
      float v;
      if constexpr (D == 5) v = fmaxf(iou_rotated_lds(gts + g * D, a, rot_pts + threadIdx.x, 256), 0.f);
      else v = match_iou<D>(gts + g * D, a);
      if (v > bv) { bv = v; bi = g; }            // first maximum wins (torch.max(dim=0))
      if (v > 0.f) atomicMax(&lbest[g], __float_as_uint(v));  // v >= 0: uint order == float order; lbest starts at 0, so a zero changes nothing
                                                             // (and almost every pair is a zero: 256 threads on one LDS word serialise)
    }
    best_val[i] = bv; best_idx[i] = bi;
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += 256) atomicMax(&gt_best_bits[g], lbest[g]);
}

// Rotated boxes: anchor_match1_kernel<5> above pays the polygon clipping per WAVE for every box some lane is near (64 consecutive anchors =
// 3.5 locations x 18 shapes: near a box that is every wave, 150 of its 200 us for the 1.6 M RRPN anchors).  This variant tests all (anchor,
// box) pairs of a 256-anchor chunk with the circle test only, compacts the survivors through LDS and deals them out evenly over the
// threads; per-anchor maximum with "first maximum wins" = 64-bit LDS atomicMax of (IoU bits, ~box index).  Same values, same argmax.
__global__ __launch_bounds__(256) void anchor_match1_rot_kernel(const float* __restrict__ gts, int G, const float* __restrict__ anchors, int A,
                                                                float* __restrict__ best_val, int* __restrict__ best_idx,
                                                                unsigned* __restrict__ gt_best_bits) {
  extern __shared__ unsigned lbest[];   // [G]
  constexpr int GS = 8;                 // boxes per round: at most 256 * GS pairs
  __shared__ unsigned pairs[256 * GS];
  __shared__ P2 rot_pts[24 * 256];      // 48 KB: the clipping's candidate points (see RotPtsLds)
  __shared__ unsigned long long abest[256];
  __shared__ int npairs;
  const int tid = threadIdx.x;
  for (int g = tid; g < G; g += 256) lbest[g] = 0u;
  const int chunks = (A + 255) / 256;
  for (int ch = blockIdx.x; ch < chunks; ch += gridDim.x) {
    const int i = ch * 256 + tid;
    float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < A) {
#pragma unroll
      for (int e = 0; e < 5; ++e) a[e] = anchors[(long long)i * 5 + e];
    }
    const float ra = 0.5f * sqrtf(a[2] * a[2] + a[3] * a[3]);
    const bool live = i < A && a[2] * a[3] >= 1e-14f;
    abest[tid] = 0x00000000FFFFFFFFull;          // IoU 0 with box 0: what "v > bv" from bv = -1 leaves when every IoU is 0
    for (int g0 = 0; g0 < G; g0 += GS) {
      if (tid == 0) npairs = 0;
      __syncthreads();
      if (live) {
        const int g1 = min(G, g0 + GS);
        for (int g = g0; g < g1; ++g) {
          const float* b = gts + g * 5;
          // the two early returns of iou_rotated: a pair dropped here has IoU exactly 0 there
          const float dx = b[0] - a[0], dy = b[1] - a[1], rs = ra + 0.5f * sqrtf(b[2] * b[2] + b[3] * b[3]);
          if (b[2] * b[3] >= 1e-14f && dx * dx + dy * dy <= rs * rs * 1.0001f) pairs[atomicAdd(&npairs, 1)] = ((unsigned)tid << 16) | (unsigned)(g - g0);
        }
      }
      __syncthreads();
      const int np = npairs;
      for (int t = tid; t < np; t += 256) {
        const unsigned pr = pairs[t];
        const int la = (int)(pr >> 16), g = g0 + (int)(pr & 0xffffu);
        float aa[5];
#pragma unroll
        for (int e = 0; e < 5; ++e) aa[e] = anchors[((long long)ch * 256 + la) * 5 + e];
        const float v = fmaxf(iou_rotated_lds(gts + g * 5, aa, rot_pts + tid, 256), 0.f);
        if (v > 0.f) {
          atomicMax(&abest[la], ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)g));
          atomicMax(&lbest[g], __float_as_uint(v));
        }
      }
      __syncthreads();
    }
    if (i < A) {
      const unsigned long long bb = abest[tid];
      best_val[i] = __uint_as_float((unsigned)(bb >> 32));
      best_idx[i] = (int)(0xFFFFFFFFu - (unsigned)(bb & 0xFFFFFFFFull));
    }
    __syncthreads();
  }
  __syncthreads();
  for (int g = tid; g < G; g += 256) atomicMax(&gt_best_bits[g], lbest[g]);
}

template <int D>
__global__ __launch_bounds__(256) void anchor_match2_kernel(const float* __restrict__ gts, int G, const float* __restrict__ anchors, int A,
                                                            const float* __restrict__ best_val, const unsigned* __restrict__ gt_best_bits,
                                                            float lo, float hi, int l0, int l1, int l2, int low_quality,
                                                            signed char* __restrict__ labels) {
  __shared__ P2 rot_pts[D == 5 ? 24 * 256 : 1];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < A; i += gridDim.x * 256) {
    const float v = best_val[i];
    int lab = (v < lo) ? l0 : ((v < hi) ? l1 : l2);
    if (low_quality) {
      float a[D];
#pragma unroll
      for (int e = 0; e < D; ++e) a[e] = anchors[(long long)i * D + e];
      for (int g = 0; g < G; ++g) {
        float v;
        if constexpr (D == 5) v = fmaxf(iou_rotated_lds(gts + g * D, a, rot_pts + threadIdx.x, 256), 0.f);
        else v = match_iou<D>(gts + g * D, a);
        if (v == __uint_as_float(gt_best_bits[g])) { lab = 1; break; }
      }
    }
    labels[i] = (signed char)lab;
  }
}

// ---------------------------------------------------------------------------------------------- RetinaNet targets / box loss
// label_anchors tail (retina_rotated.py:279-291) + Box2BoxTransform.get_deltas for one image
__global__ __launch_bounds__(256) void retina_targets_kernel(const float* __restrict__ anchors, int A, const float* __restrict__ gts,
                                                             const int* __restrict__ gt_classes, int G, const int* __restrict__ matches,
                                                             const signed char* __restrict__ mlabels, int num_classes, float wx, float wy,
                                                             float ww, float wh, int* __restrict__ gt_labels, float* __restrict__ deltas) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < A; i += gridDim.x * 256) {
    int lab = num_classes;
    f32x4_t d = {0.f, 0.f, 0.f, 0.f};
    if (G > 0) {
      const int m = matches[i];
      const int ml = mlabels[i];
      lab = (ml == 0) ? num_classes : ((ml == -1) ? -1 : gt_classes[m]);
      const f32x4_t s = *reinterpret_cast<const f32x4_t*>(anchors + (long long)i * 4);
      const f32x4_t t = *reinterpret_cast<const f32x4_t*>(gts + (long long)m * 4);
      const float sw = s[2] - s[0], sh = s[3] - s[1], scx = s[0] + 0.5f * sw, scy = s[1] + 0.5f * sh;
      const float tw = t[2] - t[0], th = t[3] - t[1], tcx = t[0] + 0.5f * tw, tcy = t[1] + 0.5f * th;
      d = f32x4_t{wx * (tcx - scx) / sw, wy * (tcy - scy) / sh, ww * logf(tw / sw), wh * logf(th / sh)};
    }
    gt_labels[i] = lab;
    *reinterpret_cast<f32x4_t*>(deltas + (long long)i * 4) = d;
  }
}

struct RetinaBoxArgs {
  const float* pred;        // (N, sumHW, pitch) fp32: anchor a of pixel p at p*pitch + a*4
  const int* labels;        // (N, R)
  const float* deltas;      // (N, R, 4)
  int N, R, A, pitch, num_classes;
  float beta;
};

template <bool BWD, typename T = __bf16>      // T: storage type of the delta gradient (bf16 product path, float in the fp32 validation mode)
__global__ __launch_bounds__(256) void retina_box_kernel(const RetinaBoxArgs a, float* __restrict__ part, const float* __restrict__ gnum,
                                                         const float* __restrict__ gden, T* __restrict__ dpred) {
  __shared__ float red[4];
  float acc = 0.f, npos = 0.f;
  const float sc = BWD ? gnum[0] / gden[0] : 0.f;
  const long long total = (long long)a.N * a.R;
  const long long pix_per_img = a.R / a.A;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / a.R, r = i - n * a.R;
    const long long px = r / a.A;
    const int an = (int)(r - px * a.A);
    const long long po = (n * pix_per_img + px) * a.pitch + an * 4;
    const int lab = a.labels[i];
    const bool pos = lab >= 0 && lab != a.num_classes;
    f32x4_t g = {0.f, 0.f, 0.f, 0.f};
    if (pos) {
      npos += 1.f;
      const f32x4_t p = *reinterpret_cast<const f32x4_t*>(a.pred + po), t = *reinterpret_cast<const f32x4_t*>(a.deltas + i * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = p[e] - t[e], ad = fabsf(d);
        if (a.beta < 1e-5f) { acc += ad; g[e] = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
        else if (ad < a.beta) { acc += 0.5f * d * d / a.beta; g[e] = d / a.beta; }
        else { acc += ad - 0.5f * a.beta; g[e] = d > 0.f ? 1.f : -1.f; }
      }
    }
    if (BWD) {
#pragma unroll
      for (int e = 0; e < 4; ++e) dpred[po + e] = (T)(g[e] * sc);
    }
  }
  if (!BWD) {
    acc = block_sum_256(acc, red);
    npos = block_sum_256(npos, red);
    if (threadIdx.x == 0) { part[blockIdx.x] = acc; part[RED + blockIdx.x] = npos; }
  }
}

// fvcore giou_loss of ONE pair and its gradient w.r.t. the first box (same arithmetic as giou_xyxy_kernel)
__device__ __forceinline__ float giou_pair(const float* a, const float* b, float eps, float* g /* may be null */) {
  const float x1 = a[0], y1 = a[1], x2 = a[2], y2 = a[3], x1g = b[0], y1g = b[1], x2g = b[2], y2g = b[3];
  const float xk1 = fmaxf(x1, x1g), yk1 = fmaxf(y1, y1g), xk2 = fminf(x2, x2g), yk2 = fminf(y2, y2g);
  const bool ov = (yk2 > yk1) && (xk2 > xk1);
  const float inter = ov ? (xk2 - xk1) * (yk2 - yk1) : 0.f;
  const float uni = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter;
  const float iou = inter / (uni + eps);
  const float xc1 = fminf(x1, x1g), yc1 = fminf(y1, y1g), xc2 = fmaxf(x2, x2g), yc2 = fmaxf(y2, y2g);
  const float ac = (xc2 - xc1) * (yc2 - yc1);
  if (g) {
    auto dmax = [](float p, float q) { return p > q ? 1.f : (p == q ? 0.5f : 0.f); };
    auto dmin = [](float p, float q) { return p < q ? 1.f : (p == q ? 0.5f : 0.f); };
    const float iw = xk2 - xk1, ih = yk2 - yk1, w1 = x2 - x1, h1 = y2 - y1, cw = xc2 - xc1, ch = yc2 - yc1;
    const float di[4] = {ov ? -dmax(x1, x1g) * ih : 0.f, ov ? -dmax(y1, y1g) * iw : 0.f, ov ? dmin(x2, x2g) * ih : 0.f, ov ? dmin(y2, y2g) * iw : 0.f};
    const float da[4] = {-h1, -w1, h1, w1};
    const float dc[4] = {-dmin(x1, x1g) * ch, -dmin(y1, y1g) * cw, dmax(x2, x2g) * ch, dmax(y2, y2g) * cw};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float du = da[e] - di[e];
      const float diou = (di[e] * (uni + eps) - inter * du) / ((uni + eps) * (uni + eps));
      const float dterm = ((dc[e] - du) * (ac + eps) - (ac - uni) * dc[e]) / ((ac + eps) * (ac + eps));
      g[e] = -(diou - dterm);
    }
  }
  return 1.f - (iou - (ac - uni) / (ac + eps));
}

// RetinaNet BBOX_REG_LOSS_TYPE "giou" (retina_rotated.py:236-245; AnchorHead, meta/heads/anchor_head.py:366-374): positives decode
// their deltas against the anchor (Box2BoxTransform.apply_deltas) and take giou_loss against the matched gt box; backward chains
// d(giou)/d(box) through the decode into the pitched bf16 delta gradient.
struct RetinaGiouArgs {
  const float* pred; const int* labels; const float* anchors; const float* gt; // pred (N,P,pitch), labels (N,R), anchors (R,4), gt (N,R,4)
  int N, R, A, pitch, num_classes;
  float wx, wy, ww, wh, clampv;
};
template <bool BWD, typename T = __bf16>
__global__ __launch_bounds__(256) void retina_giou_kernel(const RetinaGiouArgs a, float* __restrict__ part, const float* __restrict__ gnum,
                                                          const float* __restrict__ gden, T* __restrict__ dpred) {
  __shared__ float red[4];
  float acc = 0.f, npos = 0.f;
  const float sc = BWD ? gnum[0] / gden[0] : 0.f;
  const long long total = (long long)a.N * a.R, pix_per_img = a.R / a.A;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / a.R, r = i - n * a.R, px = r / a.A;
    const int an = (int)(r - px * a.A);
    const long long po = (n * pix_per_img + px) * a.pitch + an * 4;
    const int lab = a.labels[i];
    const bool pos = lab >= 0 && lab != a.num_classes;
    f32x4_t gd = {0.f, 0.f, 0.f, 0.f};
    if (pos) {
      npos += 1.f;
      const f32x4_t d = *reinterpret_cast<const f32x4_t*>(a.pred + po), anc = *reinterpret_cast<const f32x4_t*>(a.anchors + r * 4);
      const f32x4_t gt = *reinterpret_cast<const f32x4_t*>(a.gt + i * 4);
      const float aw = anc[2] - anc[0], ah = anc[3] - anc[1], acx = anc[0] + 0.5f * aw, acy = anc[1] + 0.5f * ah;
      const float dw = d[2] / a.ww, dh = d[3] / a.wh;
      const bool cw_ = dw > a.clampv, ch_ = dh > a.clampv;
      const float pw = expf(cw_ ? a.clampv : dw) * aw, ph = expf(ch_ ? a.clampv : dh) * ah;
      const float pcx = d[0] / a.wx * aw + acx, pcy = d[1] / a.wy * ah + acy;
      const float box[4] = {pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph};
      const float gb[4] = {gt[0], gt[1], gt[2], gt[3]};
      float g[4];
      acc += giou_pair(box, gb, 1e-7f, BWD ? g : nullptr);
      if (BWD) {
        gd[0] = (g[0] + g[2]) * aw / a.wx;
        gd[1] = (g[1] + g[3]) * ah / a.wy;
        gd[2] = cw_ ? 0.f : (g[2] - g[0]) * 0.5f * pw / a.ww;
        gd[3] = ch_ ? 0.f : (g[3] - g[1]) * 0.5f * ph / a.wh;
      }
    }
    if (BWD) {
#pragma unroll
      for (int e = 0; e < 4; ++e) dpred[po + e] = (T)(gd[e] * sc);
    }
  }
  if (!BWD) {
    acc = block_sum_256(acc, red);
    npos = block_sum_256(npos, red);
    if (threadIdx.x == 0) { part[blockIdx.x] = acc; part[RED + blockIdx.x] = npos; }
  }
}

// sums[0] = smooth-L1 sum over positives, sums[1] = number of positives; normalizer <- m*normalizer + (1-m)*max(npos,1)
__global__ void retina_finish_kernel(const float* __restrict__ part, int nblk_, float* __restrict__ sums, float* __restrict__ normalizer,
                                     float momentum) {
  __shared__ float red[4];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < nblk_; i += 256) { a += part[i]; b += part[RED + i]; }
  a = block_sum_256(a, red);
  b = block_sum_256(b, red);
  if (threadIdx.x == 0) {
    sums[0] = a; sums[1] = b;
    if (normalizer) normalizer[0] = momentum * normalizer[0] + (1.f - momentum) * fmaxf(b, 1.f);
  }
}

inline int nblk(long long n, int cap = RED) {
  long long g = (n + 255) / 256;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" long long sod_nms_workspace_bytes(int n) {
  const long long words = (n + 63) / 64;
  return (long long)n * words * 8;
}

extern "C" int sod_nms(const float* boxes, const long long* order, int n, float iou_threshold, long long* keep, int* num_keep,
                       void* mask_ws, void* stream) {
  if (n < 0 || !num_keep || (n > 0 && (!boxes || !order || !keep || !mask_ws))) return SOD_EARG;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) return (int)hipMemsetAsync(num_keep, 0, sizeof(int), st);
  const int words = (n + 63) / 64;
  if (words > 1024) return SOD_ESIZE;   // 65536 boxes per call (detectron2 switches to per-class loops above 40 000)
  hipError_t e = hipMemsetAsync(mask_ws, 0, (size_t)n * words * 8, st);
  if (e != hipSuccess) return (int)e;
  SOD_LAUNCH(nms_mask_kernel<4>, dim3(words, words), dim3(64), 0, st, boxes, order, n, iou_threshold, (unsigned long long*)mask_ws, words);
  SOD_LAUNCH(nms_scan_kernel, dim3(1), dim3(1024), 0, st, (const unsigned long long*)mask_ws, order, n, words, keep, num_keep);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_nms_rotated(const float* boxes, const long long* order, int n, float iou_threshold, long long* keep, int* num_keep,
                               void* mask_ws, void* stream) {
  if (n < 0 || !num_keep || (n > 0 && (!boxes || !order || !keep || !mask_ws))) return SOD_EARG;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) return (int)hipMemsetAsync(num_keep, 0, sizeof(int), st);
  const int words = (n + 63) / 64;
  if (words > 1024) return SOD_ESIZE;
  hipError_t e = hipMemsetAsync(mask_ws, 0, (size_t)n * words * 8, st);
  if (e != hipSuccess) return (int)e;
  SOD_LAUNCH(nms_mask_kernel<5>, dim3(words, words), dim3(64), 0, st, boxes, order, n, iou_threshold, (unsigned long long*)mask_ws, words);
  SOD_LAUNCH(nms_scan_kernel, dim3(1), dim3(1024), 0, st, (const unsigned long long*)mask_ws, order, n, words, keep, num_keep);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_box_iou_rotated(const float* boxes1, int n1, const float* boxes2, int n2, float* iou_out, void* stream) {
  if (n1 < 0 || n2 < 0 || ((long long)n1 * n2 > 0 && (!boxes1 || !boxes2 || !iou_out))) return SOD_EARG;
  if ((long long)n1 * n2 == 0) return SOD_OK;
  SOD_LAUNCH(pairwise_iou_rotated_kernel, dim3(nblk((long long)n1 * n2, 4096)), dim3(256), 0, (hipStream_t)stream, boxes1, n1, boxes2, n2, iou_out);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

static int roi_fill(RoiArgs& a, const void* x, const float* rois, int R, int N, int H, int W, int C, int PH, int PW, float scale,
                    int sampling_ratio, int rotated) {
  if (!x || !rois || R < 0 || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || PH <= 0 || PW <= 0) return SOD_EARG;
  a.x = (const __bf16*)x; a.rois = rois; a.R = R; a.N = N; a.H = H; a.W = W; a.C = C; a.PH = PH; a.PW = PW;
  a.scale = scale; a.sampling_ratio = sampling_ratio; a.rotated = rotated;
  return SOD_OK;
}

extern "C" int sod_roi_align_fwd(const void* x, const float* rois, float* out, int R, int N, int H, int W, int C, int PH, int PW,
                                 float spatial_scale, int sampling_ratio, int rotated, void* stream) {
  RoiArgs a{};
  int rc = roi_fill(a, x, rois, R, N, H, W, C, PH, PW, spatial_scale, sampling_ratio, rotated);
  if (rc || !out) return rc ? rc : SOD_EARG;
  if (R == 0) return SOD_OK;
  SOD_LAUNCH((roi_align_kernel<false, false>), dim3(nblk((long long)R * PH * PW * (C / 8), 8192)), dim3(256), 0, (hipStream_t)stream, a, out, nullptr, nullptr);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_roi_align_fwd_f32(const float* x, const float* rois, float* out, int R, int N, int H, int W, int C, int PH, int PW,
                                     float spatial_scale, int sampling_ratio, int rotated, void* stream) {
  RoiArgs a{};
  int rc = roi_fill(a, x, rois, R, N, H, W, C, PH, PW, spatial_scale, sampling_ratio, rotated);
  if (rc || !out) return rc ? rc : SOD_EARG;
  if (R == 0) return SOD_OK;
  SOD_LAUNCH((roi_align_kernel<false, true>), dim3(nblk((long long)R * PH * PW * (C / 8), 8192)), dim3(256), 0, (hipStream_t)stream, a, out, nullptr, nullptr);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_roi_align_bwd(const float* dout, const float* rois, float* dx, int R, int N, int H, int W, int C, int PH, int PW,
                                 float spatial_scale, int sampling_ratio, int rotated, void* stream) {
  RoiArgs a{};
  int rc = roi_fill(a, dx, rois, R, N, H, W, C, PH, PW, spatial_scale, sampling_ratio, rotated);   // x unused in bwd
  if (rc || !dout || !dx) return rc ? rc : SOD_EARG;
  if (R == 0) return SOD_OK;
  if (C % ROI_CC == 0 && C / ROI_CC <= 65535) {
    SOD_LAUNCH(roi_align_bwd_tile_kernel, dim3(R, C / ROI_CC), dim3(256), 0, (hipStream_t)stream, a, dout, dx);
    SOD_CHECK_LAUNCH();
    return SOD_OK;
  }
  SOD_LAUNCH((roi_align_kernel<true, false>), dim3(nblk((long long)R * PH * PW * (C / 8), 8192)), dim3(256), 0, (hipStream_t)stream, a, nullptr, dout, dx);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_giou_loss_xyxy(const float* boxes1, const float* boxes2, long long P, float eps, float* elem_out, float* sum_out,
                                  const float* grad_scale, float* dboxes1, float* ws, void* stream) {
  if (!boxes1 || !boxes2 || P < 0 || (sum_out && !ws)) return SOD_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int g = nblk(P);
  SOD_LAUNCH(giou_xyxy_kernel, dim3(g), dim3(256), 0, st, boxes1, boxes2, P, eps, elem_out, sum_out ? ws : nullptr, grad_scale, dboxes1);
  if (sum_out) SOD_LAUNCH(finish_sum2, dim3(1), dim3(256), 0, st, ws, g, sum_out);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_smooth_l1_loss(const float* input, const float* target, long long n, float beta, float* elem_out, float* sum_out,
                                  const float* grad_scale, float* dinput, float* ws, void* stream) {
  if (!input || !target || n < 0 || (sum_out && !ws)) return SOD_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int g = nblk(n);
  SOD_LAUNCH(smooth_l1_kernel, dim3(g), dim3(256), 0, st, input, target, n, beta, elem_out, sum_out ? ws : nullptr, grad_scale, dinput);
  if (sum_out) SOD_LAUNCH(finish_sum2, dim3(1), dim3(256), 0, st, ws, g, sum_out);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

template <int D>
static int anchor_match_impl(const float* gt_boxes, int G, const float* anchors, int A, float thr_lo, float thr_hi,
                             int label_below, int label_between, int label_above, int allow_low_quality,
                             float* matched_vals, int* matches, signed char* labels, unsigned* gt_best_ws, void* stream) {
  if (!anchors || A <= 0 || !matched_vals || !matches || !labels || G < 0 || G > 4096) return SOD_EARG;
  hipStream_t st = (hipStream_t)stream;
  if (G == 0) {   // Matcher on an empty gt set: everything unmatched with the lowest label
    hipError_t e = hipMemsetAsync(matches, 0, sizeof(int) * A, st);
    if (e == hipSuccess) e = hipMemsetAsync(matched_vals, 0, sizeof(float) * A, st);
    if (e == hipSuccess) e = hipMemsetAsync(labels, label_below & 0xff, A, st);
    return (int)e;
  }
  if (!gt_boxes || !gt_best_ws) return SOD_EARG;
  hipError_t e = hipMemsetAsync(gt_best_ws, 0, sizeof(unsigned) * G, st);
  if (e != hipSuccess) return (int)e;
  const int g = nblk(A, 2048);
  if (D == 5) SOD_LAUNCH(anchor_match1_rot_kernel, dim3(g), dim3(256), sizeof(unsigned) * G, st, gt_boxes, G, anchors, A, matched_vals, matches, gt_best_ws);
  else SOD_LAUNCH(anchor_match1_kernel<D>, dim3(g), dim3(256), sizeof(unsigned) * G, st, gt_boxes, G, anchors, A, matched_vals, matches, gt_best_ws);
  SOD_LAUNCH(anchor_match2_kernel<D>, dim3(g), dim3(256), 0, st, gt_boxes, G, anchors, A, matched_vals, gt_best_ws, thr_lo, thr_hi,
             label_below, label_between, label_above, allow_low_quality, labels);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

// ROIHeads.label_and_sample_proposals' matching + labelling for the WHOLE batch in one launch: proposal r of image n (rows >= counts[n] are
// padding) against that image's boxes gt[gt_off[n] .. gt_off[n + 1]): best IoU with "first maximum wins", Matcher([thr], [l0, l1]) without
// low-quality matches, class = the matched box's class for label 1, num_classes for label 0, -1 for label -1 and for padding rows.
template <int D>
__global__ __launch_bounds__(256) void roi_label_batched_kernel(const float* __restrict__ boxes, const int* __restrict__ counts, int R,
                                                                const float* __restrict__ gt, const int* __restrict__ gt_classes,
                                                                const int* __restrict__ gt_off, float thr, int l0, int l1, int num_classes,
                                                                int* __restrict__ matches, signed char* __restrict__ cls) {
  __shared__ P2 rot_pts[D == 5 ? 24 * 256 : 1];
  const int n = blockIdx.y;
  const int g0 = gt_off[n], G = gt_off[n + 1] - g0, cnt = counts[n];
  for (int r = blockIdx.x * 256 + threadIdx.x; r < R; r += gridDim.x * 256) {
    const long long o = (long long)n * R + r;
    if (r >= cnt) { matches[o] = 0; cls[o] = -1; continue; }
    float a[D];
#pragma unroll
    for (int e = 0; e < D; ++e) a[e] = boxes[o * D + e];
    float bv = -1.f; int bi = 0;
    for (int g = 0; g < G; ++g) {
      float v;
      if constexpr (D == 5) v = fmaxf(iou_rotated_lds(gt + (long long)(g0 + g) * D, a, rot_pts + threadIdx.x, 256), 0.f);
      else v = match_iou<D>(gt + (long long)(g0 + g) * D, a);
      if (v > bv) { bv = v; bi = g; }
    }
    int c = num_classes;                            // no boxes: everything is background (matches stay 0)
    if (G > 0) {
      const int lab = (bv < thr) ? l0 : l1;
      c = (lab == 0) ? num_classes : ((lab == -1) ? -1 : gt_classes[g0 + bi]);
    }
    matches[o] = bi;
    cls[o] = (signed char)c;
  }
}

extern "C" int sod_roi_label_batched(const float* boxes, const int* counts, int N, int R, int box_dim, const float* gt_boxes, const int* gt_classes,
                                     const int* gt_off, float iou_threshold, int label_below, int label_above, int num_classes, int* matches,
                                     signed char* classes, void* stream) {
  if (!boxes || !counts || !gt_off || !matches || !classes || N <= 0 || N > 65535 || R <= 0 || (box_dim != 4 && box_dim != 5) || num_classes <= 0 ||
      num_classes > 126)
    return SOD_EARG;
  const dim3 grid((R + 255) / 256, N);
  if (box_dim == 4) SOD_LAUNCH(roi_label_batched_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, boxes, counts, R, gt_boxes, gt_classes, gt_off,
                               iou_threshold, label_below, label_above, num_classes, matches, classes);
  else SOD_LAUNCH(roi_label_batched_kernel<5>, grid, dim3(256), 0, (hipStream_t)stream, boxes, counts, R, gt_boxes, gt_classes, gt_off, iou_threshold,
                  label_below, label_above, num_classes, matches, classes);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_anchor_match(const float* gt_boxes, int G, const float* anchors, int A, float thr_lo, float thr_hi,
                                int label_below, int label_between, int label_above, int allow_low_quality,
                                float* matched_vals, int* matches, signed char* labels, unsigned* gt_best_ws, void* stream) {
  return anchor_match_impl<4>(gt_boxes, G, anchors, A, thr_lo, thr_hi, label_below, label_between, label_above, allow_low_quality, matched_vals,
                              matches, labels, gt_best_ws, stream);
}

extern "C" int sod_anchor_match_rotated(const float* gt_boxes, int G, const float* anchors, int A, float thr_lo, float thr_hi,
                                        int label_below, int label_between, int label_above, int allow_low_quality,
                                        float* matched_vals, int* matches, signed char* labels, unsigned* gt_best_ws, void* stream) {
  return anchor_match_impl<5>(gt_boxes, G, anchors, A, thr_lo, thr_hi, label_below, label_between, label_above, allow_low_quality, matched_vals,
                              matches, labels, gt_best_ws, stream);
}

extern "C" int sod_retina_targets(const float* anchors, int A, const float* gt_boxes, const int* gt_classes, int G, const int* matches,
                                  const signed char* match_labels, int num_classes, const float* weights4, int* gt_labels, float* gt_deltas,
                                  void* stream) {
  if (!anchors || A <= 0 || !gt_labels || !gt_deltas || !weights4 || (G > 0 && (!gt_boxes || !gt_classes || !matches || !match_labels))) return SOD_EARG;
  SOD_LAUNCH(retina_targets_kernel, dim3(nblk(A, 2048)), dim3(256), 0, (hipStream_t)stream, anchors, A, gt_boxes, gt_classes, G, matches, match_labels,
             num_classes, weights4[0], weights4[1], weights4[2], weights4[3], gt_labels, gt_deltas);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_retina_box_loss_fwd(const float* pred, int pitch, const int* gt_labels, const float* gt_deltas, int N, int R, int A,
                                       int num_classes, float beta, float* sums2, float* normalizer, float momentum, float* ws, void* stream) {
  if (!pred || !gt_labels || !gt_deltas || !sums2 || !ws || N <= 0 || R <= 0 || A <= 0 || R % A || pitch < A * 4) return SOD_EARG;
  RetinaBoxArgs a{pred, gt_labels, gt_deltas, N, R, A, pitch, num_classes, beta};
  hipStream_t st = (hipStream_t)stream;
  const int g = nblk((long long)N * R);
  SOD_LAUNCH((retina_box_kernel<false, __bf16>), dim3(g), dim3(256), 0, st, a, ws, nullptr, nullptr, nullptr);
  SOD_LAUNCH(retina_finish_kernel, dim3(1), dim3(256), 0, st, ws, g, sums2, normalizer, momentum);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

static int retina_giou_fill(RetinaGiouArgs& a, const float* pred, int pitch, const int* labels, const float* anchors, const float* gt, int N, int R,
                            int A, int K, const float* w4, float clampv) {
  if (!pred || !labels || !anchors || !gt || !w4 || N <= 0 || R <= 0 || A <= 0 || R % A || pitch < A * 4) return SOD_EARG;
  for (int i = 0; i < 4; ++i) if (!(w4[i] > 0.f)) return SOD_EARG;
  a = RetinaGiouArgs{pred, labels, anchors, gt, N, R, A, pitch, K, w4[0], w4[1], w4[2], w4[3], clampv};
  return SOD_OK;
}

extern "C" int sod_retina_giou_loss_fwd(const float* pred, int pitch, const int* gt_labels, const float* anchors, const float* matched_boxes,
                                        int N, int R, int A, int num_classes, const float* weights4, float scale_clamp, float* sums2,
                                        float* normalizer, float momentum, float* ws, void* stream) {
  RetinaGiouArgs a{};
  int rc = retina_giou_fill(a, pred, pitch, gt_labels, anchors, matched_boxes, N, R, A, num_classes, weights4, scale_clamp);
  if (rc || !sums2 || !ws) return rc ? rc : SOD_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int g = nblk((long long)N * R);
  SOD_LAUNCH((retina_giou_kernel<false, __bf16>), dim3(g), dim3(256), 0, st, a, ws, nullptr, nullptr, nullptr);
  SOD_LAUNCH(retina_finish_kernel, dim3(1), dim3(256), 0, st, ws, g, sums2, normalizer, momentum);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_retina_giou_loss_bwd(const float* pred, int pitch, const int* gt_labels, const float* anchors, const float* matched_boxes,
                                        int N, int R, int A, int num_classes, const float* weights4, float scale_clamp,
                                        const float* grad_num, const float* grad_den, void* dpred_bf16, void* stream) {
  RetinaGiouArgs a{};
  int rc = retina_giou_fill(a, pred, pitch, gt_labels, anchors, matched_boxes, N, R, A, num_classes, weights4, scale_clamp);
  if (rc || !grad_num || !grad_den || !dpred_bf16) return rc ? rc : SOD_EARG;
  SOD_LAUNCH((retina_giou_kernel<true, __bf16>), dim3(nblk((long long)N * R, 4096)), dim3(256), 0, (hipStream_t)stream, a, nullptr, grad_num, grad_den,
             (__bf16*)dpred_bf16);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_retina_box_loss_bwd(const float* pred, int pitch, const int* gt_labels, const float* gt_deltas, int N, int R, int A,
                                       int num_classes, float beta, const float* grad_num, const float* grad_den, void* dpred_bf16, void* stream) {
  if (!pred || !gt_labels || !gt_deltas || !grad_num || !grad_den || !dpred_bf16 || N <= 0 || R <= 0 || A <= 0 || R % A || pitch < A * 4) return SOD_EARG;
  RetinaBoxArgs a{pred, gt_labels, gt_deltas, N, R, A, pitch, num_classes, beta};
  SOD_LAUNCH((retina_box_kernel<true, __bf16>), dim3(nblk((long long)N * R, 4096)), dim3(256), 0, (hipStream_t)stream, a, nullptr, grad_num, grad_den, (__bf16*)dpred_bf16);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_retina_giou_loss_bwd_f32(const float* pred, int pitch, const int* gt_labels, const float* anchors, const float* matched_boxes,
                                            int N, int R, int A, int num_classes, const float* weights4, float scale_clamp,
                                            const float* grad_num, const float* grad_den, float* dpred, void* stream) {
  RetinaGiouArgs a{};
  int rc = retina_giou_fill(a, pred, pitch, gt_labels, anchors, matched_boxes, N, R, A, num_classes, weights4, scale_clamp);
  if (rc || !grad_num || !grad_den || !dpred) return rc ? rc : SOD_EARG;
  SOD_LAUNCH((retina_giou_kernel<true, float>), dim3(nblk((long long)N * R, 4096)), dim3(256), 0, (hipStream_t)stream, a, nullptr, grad_num, grad_den, dpred);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_retina_box_loss_bwd_f32(const float* pred, int pitch, const int* gt_labels, const float* gt_deltas, int N, int R, int A,
                                           int num_classes, float beta, const float* grad_num, const float* grad_den, float* dpred, void* stream) {
  if (!pred || !gt_labels || !gt_deltas || !grad_num || !grad_den || !dpred || N <= 0 || R <= 0 || A <= 0 || R % A || pitch < A * 4) return SOD_EARG;
  RetinaBoxArgs a{pred, gt_labels, gt_deltas, N, R, A, pitch, num_classes, beta};
  SOD_LAUNCH((retina_box_kernel<true, float>), dim3(nblk((long long)N * R, 4096)), dim3(256), 0, (hipStream_t)stream, a, nullptr, grad_num, grad_den, dpred);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" long long sod_batched_nms_workspace_bytes(int B, int M, int box_dim) {
  const long long words = (M + 63) / 64;
  return (long long)B * M * words * 8 + (long long)B * M * box_dim * (long long)sizeof(float) + (long long)B * (long long)sizeof(int);
}

// shifted boxes + per-image candidate count (first half of batched NMS); the caller sorts the scores (any stable descending sort)
// and then calls sod_batched_nms_run with the order.  ws layout: [mask][shifted boxes][nvalid].
extern "C" int sod_batched_nms_prepare(const float* boxes, const float* scores, const int* classes, int B, int M, int box_dim, void* ws, void* stream) {
  if (!boxes || !scores || !classes || !ws || B <= 0 || M <= 0 || M > 65536 || (box_dim != 4 && box_dim != 5)) return SOD_EARG;
  const long long words = (M + 63) / 64;
  float* shifted = (float*)((char*)ws + (long long)B * M * words * 8);
  int* nvalid = (int*)(shifted + (long long)B * M * box_dim);
  if (box_dim == 4) SOD_LAUNCH(nms_class_shift_kernel<4>, dim3(B), dim3(1024), 0, (hipStream_t)stream, boxes, scores, classes, M, shifted, nvalid);
  else SOD_LAUNCH(nms_class_shift_kernel<5>, dim3(B), dim3(1024), 0, (hipStream_t)stream, boxes, scores, classes, M, shifted, nvalid);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_batched_nms_run(const long long* order, int B, int M, int box_dim, float iou_threshold, int max_keep, long long* keep,
                                   int* num_keep, void* ws, void* stream) {
  if (!order || !keep || !num_keep || !ws || B <= 0 || M <= 0 || M > 65536 || max_keep <= 0 || (box_dim != 4 && box_dim != 5)) return SOD_EARG;
  const int words = (M + 63) / 64;
  if (words > 1024) return SOD_ESIZE;
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* mask = (unsigned long long*)ws;
  const float* shifted = (const float*)((char*)ws + (long long)B * M * words * 8);
  const int* nvalid = (const int*)(shifted + (long long)B * M * box_dim);
  hipError_t e = hipMemsetAsync(mask, 0, (size_t)B * M * words * 8, st);
  if (e != hipSuccess) return (int)e;
  if (box_dim == 4) SOD_LAUNCH(nms_mask_batched_kernel<4>, dim3(words, words, B), dim3(64), 0, st, shifted, order, nvalid, M, iou_threshold, mask, words);
  else SOD_LAUNCH(nms_mask_batched_kernel<5>, dim3(words, words, B), dim3(64), 0, st, shifted, order, nvalid, M, iou_threshold, mask, words);
  SOD_LAUNCH(nms_scan_batched_kernel, dim3(B), dim3(1024), 0, st, (const unsigned long long*)mask, order, nvalid, M, words, max_keep, keep, num_keep);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}

extern "C" int sod_rpn_clip_filter(float* boxes, float* scores, const float* image_hw, int B, int M, int box_dim, float min_size, int* bad_count,
                                   void* stream) {
  if (!boxes || !scores || !image_hw || !bad_count || B <= 0 || M <= 0 || (box_dim != 4 && box_dim != 5)) return SOD_EARG;
  const int g = nblk((long long)B * M, 4096);
  if (box_dim == 4) SOD_LAUNCH(rpn_clip_filter_kernel<4>, dim3(g), dim3(256), 0, (hipStream_t)stream, boxes, scores, image_hw, B, M, min_size, bad_count);
  else SOD_LAUNCH(rpn_clip_filter_kernel<5>, dim3(g), dim3(256), 0, (hipStream_t)stream, boxes, scores, image_hw, B, M, min_size, bad_count);
  SOD_CHECK_LAUNCH();
  return SOD_OK;
}
