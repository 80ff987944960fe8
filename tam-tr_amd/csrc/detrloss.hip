// detrloss.hip - the per-layer terms of the RT-DETR loss (ultralytics/models/utils/loss.py:85-166,282-326: varifocal class loss on the matched
// IoU, L1 and RIOU box losses; RIOU = ultralytics/utils/metrics.py:91-130) and the Hungarian matcher's cost matrix
// (ultralytics/models/utils/ops.py:84-112) as a handful of kernels, gfx950.
//
// As torch ops the 12 terms of a step are ~380 elementwise / gather / scatter / reduce launches forward and ~400 backward (profiles/
// r04_step_phases.txt: 1.07 + ~1.5 ms of GPU time, 8 ms of host issue time) on tensors of a few thousand elements.  Here, per call of
// DETRLoss._layers (all decoder layers stacked: pb [Lr, B, nq, 4], ps [Lr, B, nq, nc], the matched pairs as flat index lists):
//   detr_pairs_fwd   thread per matched pair: plain IoU -> the class target / IoU-score tables [Lr, B, nq]; L1 and 1 - RIOU of the pair
//   detr_class_fwd   thread per query row: the varifocal terms of its nc classes, workgroup partial sums per layer
//   detr_reduce      one workgroup per layer adds the partials and the pair terms in index order -> (class, bbox, giou) [Lr]
//   detr_class_bwd / detr_pairs_bwd   the analytic gradients (RIOU's alpha is a constant, as under torch.no_grad in the reference;
//                    max / min ties split the gradient in halves and |.|' (0) = 0, as torch's autograd does)
//   detr_match_cost  thread per (layer, image, query, box): 2 focal + 5 L1 + 2 (1 - RIOU), non-finite -> 0 (ops.py:112)
// fp32 throughout (the reference computes the loss with autocast disabled); every sum in a fixed order.
#include "common.h"

namespace {

constexpr float DL_EPS = 1e-7f;
constexpr float DL_PI = 3.14159265358979323846f;
constexpr int DL_THREADS = 256;

struct Box { float x, y, w, h; };

__device__ __forceinline__ Box ldbox(const float* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  return Box{v.x, v.y, v.z, v.w};
}

// plain IoU of xywh boxes (metrics.py:93-110)
__device__ __forceinline__ float iou_xywh(const Box& a, const Box& b, float& inter_out, float& union_out) {
  const float ax1 = a.x - a.w / 2, ax2 = a.x + a.w / 2, ay1 = a.y - a.h / 2, ay2 = a.y + a.h / 2;
  const float bx1 = b.x - b.w / 2, bx2 = b.x + b.w / 2, by1 = b.y - b.h / 2, by2 = b.y + b.h / 2;
  const float iw = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f), ih = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
  inter_out = iw * ih;
  union_out = a.w * a.h + b.w * b.h - inter_out + DL_EPS;
  return inter_out / union_out;
}

// RIOU (metrics.py:112-130) of prediction a against box b; alpha returned for the backward
__device__ __forceinline__ float riou_xywh(const Box& a, const Box& b, float* iou_out = nullptr) {
  float inter, uni;
  const float iou = iou_xywh(a, b, inter, uni);
  const float ax1 = a.x - a.w / 2, ax2 = a.x + a.w / 2, ay1 = a.y - a.h / 2, ay2 = a.y + a.h / 2;
  const float bx1 = b.x - b.w / 2, bx2 = b.x + b.w / 2, by1 = b.y - b.h / 2, by2 = b.y + b.h / 2;
  const float sx = bx1 + bx2 - ax1 - ax2, sy = by1 + by2 - ay1 - ay2;
  const float rho2 = (sx * sx + sy * sy) / 4;
  const float cs = fmaxf(a.w, a.h) + fmaxf(b.w, b.h) + sqrtf(rho2) + DL_EPS;
  const float da = atanf(b.w / b.h) - atanf(a.w / a.h);
  const float v = (4.f / (DL_PI * DL_PI)) * da * da;
  const float alpha = v / (v - iou + (1.f + DL_EPS));
  if (iou_out) *iou_out = iou;
  return iou - (rho2 / (cs * cs) + v * alpha);
}

// d(1 - RIOU) / d(a) for a = (x, y, w, h), alpha held constant (metrics.py:127-129)
__device__ __forceinline__ void riou_loss_grad(const Box& a, const Box& b, float (&g)[4]) {
  const float ax1 = a.x - a.w / 2, ax2 = a.x + a.w / 2, ay1 = a.y - a.h / 2, ay2 = a.y + a.h / 2;
  const float bx1 = b.x - b.w / 2, bx2 = b.x + b.w / 2, by1 = b.y - b.h / 2, by2 = b.y + b.h / 2;
  // iw = clamp(min(ax2, bx2) - max(ax1, bx1), 0): d/d(ax2) = [ax2 < bx2] (1/2 at a tie), d/d(ax1) = -[ax1 > bx1] (1/2 at a tie), times [iw >= 0]
  const float ixr = fminf(ax2, bx2) - fmaxf(ax1, bx1), iyr = fminf(ay2, by2) - fmaxf(ay1, by1);
  const float iw = fmaxf(ixr, 0.f), ih = fmaxf(iyr, 0.f);
  const float cx = ixr >= 0.f ? 1.f : 0.f, cy = iyr >= 0.f ? 1.f : 0.f;
  const float d_ax2 = (ax2 < bx2 ? 1.f : ax2 == bx2 ? 0.5f : 0.f) * cx, d_ax1 = -(ax1 > bx1 ? 1.f : ax1 == bx1 ? 0.5f : 0.f) * cx;
  const float d_ay2 = (ay2 < by2 ? 1.f : ay2 == by2 ? 0.5f : 0.f) * cy, d_ay1 = -(ay1 > by1 ? 1.f : ay1 == by1 ? 0.5f : 0.f) * cy;
  // ax1 = x - w/2, ax2 = x + w/2
  const float diw[4] = {d_ax1 + d_ax2, 0.f, 0.5f * (d_ax2 - d_ax1), 0.f};
  const float dih[4] = {0.f, d_ay1 + d_ay2, 0.f, 0.5f * (d_ay2 - d_ay1)};
  const float inter = iw * ih, uni = a.w * a.h + b.w * b.h - inter + DL_EPS, iou = inter / uni;
  const float sx = bx1 + bx2 - ax1 - ax2, sy = by1 + by2 - ay1 - ay2;   // = 2 (bx - ax), 2 (by - ay)
  const float rho2 = (sx * sx + sy * sy) / 4, s = sqrtf(rho2);
  const float m1 = fmaxf(a.w, a.h), cs = m1 + fmaxf(b.w, b.h) + s + DL_EPS, c2 = cs * cs;
  const float r1 = a.w / a.h, da = atanf(b.w / b.h) - atanf(r1);
  const float kv = 4.f / (DL_PI * DL_PI), v = kv * da * da, alpha = v / (v - iou + (1.f + DL_EPS));
  const float drho2[4] = {-sx, -sy, 0.f, 0.f};                         // d(rho2)/d(ax1 + ax2) = -2 sx / 4 each; ax1 + ax2 = 2 x
  const float dm1[4] = {0.f, 0.f, a.w > a.h ? 1.f : a.w == a.h ? 0.5f : 0.f, a.h > a.w ? 1.f : a.w == a.h ? 0.5f : 0.f};
  const float inv = 1.f / (a.h * (1.f + r1 * r1));                      // d(atan(w/h)): d/dw = 1 / (h (1 + r^2)), d/dh = -r / (h (1 + r^2))
  const float dat[4] = {0.f, 0.f, inv, -r1 * inv};
  const float dwh[4] = {0.f, 0.f, a.h, a.w};                            // d(w h)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float dinter = diw[k] * ih + iw * dih[k];
    const float diou = (dinter * uni - inter * (dwh[k] - dinter)) / (uni * uni);
    const float ds = 0.5f / s * drho2[k];                               // (inf * 0 = NaN when the centres coincide, as in the reference)
    const float dcs = dm1[k] + ds;
    const float dterm = (drho2[k] * c2 - rho2 * 2.f * cs * dcs) / (c2 * c2);
    const float dv = 2.f * kv * da * (-dat[k]);
    g[k] = -(diou - dterm - alpha * dv);
  }
}

__device__ __forceinline__ float block_sum256(float v, float* red) {   // fixed-order tree; valid in thread 0
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, WAVE);
  if (threadIdx.x % WAVE == 0) red[threadIdx.x / WAVE] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0)
    for (int w = 0; w < DL_THREADS / WAVE; ++w) t += red[w];
  __syncthreads();
  return t;
}

// pair k: (layer li, image bi, query si) <-> box gi.  tables [Lr, B, nq]: tgt (class, preset to nc = no object), score (preset to 0)
__global__ __launch_bounds__(DL_THREADS) void detr_pairs_fwd_kernel(const float* __restrict__ pb, const float* __restrict__ gtb, const long long* __restrict__ gtc,
                                                                    const long long* __restrict__ li, const long long* __restrict__ bi,
                                                                    const long long* __restrict__ si, const long long* __restrict__ gi, int npairs, int B, int nq,
                                                                    long long* __restrict__ tgt, float* __restrict__ score, float* __restrict__ pair_l1,
                                                                    float* __restrict__ pair_riou) {
  const int k = blockIdx.x * DL_THREADS + threadIdx.x;
  if (k >= npairs) return;
  const size_t row = ((size_t)li[k] * B + bi[k]) * nq + si[k];
  const Box a = ldbox(pb + row * 4), b = ldbox(gtb + (size_t)gi[k] * 4);
  float iou;
  const float r = riou_xywh(a, b, &iou);
  tgt[row] = gtc[gi[k]];
  score[row] = iou;
  pair_l1[k] = fabsf(a.x - b.x) + fabsf(a.y - b.y) + fabsf(a.w - b.w) + fabsf(a.h - b.h);
  pair_riou[k] = 1.f - r;
}

__device__ __forceinline__ float bce_logits(float x, float t) { return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))); }

// varifocal terms (loss.py:118-127, utils/loss.py:146-153) of one query row; partial[l][blk] = sum over the workgroup's rows
__global__ __launch_bounds__(DL_THREADS) void detr_class_fwd_kernel(const float* __restrict__ ps, const long long* __restrict__ tgt, const float* __restrict__ score,
                                                                    int rows_per_layer, int nc, float* __restrict__ partial) {
  __shared__ float red[DL_THREADS / WAVE];
  const int l = blockIdx.y, r = blockIdx.x * DL_THREADS + threadIdx.x;
  float acc = 0.f;
  if (r < rows_per_layer) {
    const size_t row = (size_t)l * rows_per_layer + r;
    const long long t = tgt[row];
    const float s = score[row];
    const float* x = ps + row * nc;
    for (int c = 0; c < nc; ++c) {
      const float xv = x[c], p = 1.f / (1.f + expf(-xv));
      const bool pos = c == t;
      const float tv = pos ? s : 0.f, w = pos ? s : 0.75f * p * p;
      acc += bce_logits(xv, tv) * w;
    }
  }
  const float tot = block_sum256(acc, red);
  if (threadIdx.x == 0) partial[(size_t)l * gridDim.x + blockIdx.x] = tot;
}

// one workgroup per layer: class partials, pair L1 terms, pair RIOU terms -> out[l] = (class, bbox, giou), scaled
__global__ __launch_bounds__(DL_THREADS) void detr_reduce_kernel(const float* __restrict__ partial, int nblk, const float* __restrict__ pair_l1,
                                                                 const float* __restrict__ pair_riou, int n, float g_class, float g_bbox, float g_giou,
                                                                 float* __restrict__ out, int Lr) {
  __shared__ float red[DL_THREADS / WAVE];
  const int l = blockIdx.x;
  float a = 0.f, b = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < nblk; i += DL_THREADS) a += partial[(size_t)l * nblk + i];
  for (int i = threadIdx.x; i < n; i += DL_THREADS) { b += pair_l1[(size_t)l * n + i]; c += pair_riou[(size_t)l * n + i]; }
  const float sa = block_sum256(a, red), sb = block_sum256(b, red), sc = block_sum256(c, red);
  if (threadIdx.x == 0) {
    const float dn = (float)(n > 1 ? n : 1);
    out[l] = sa / dn * g_class;
    out[Lr + l] = n ? g_bbox * sb / (float)n : 0.f;
    out[2 * Lr + l] = n ? g_giou * sc / (float)n : 0.f;
  }
}

// d(ps): dense.  up[l] = upstream gradient of the layer's class term
__global__ __launch_bounds__(DL_THREADS) void detr_class_bwd_kernel(const float* __restrict__ ps, const long long* __restrict__ tgt, const float* __restrict__ score,
                                                                    const float* __restrict__ up, int rows_per_layer, int nc, float scale, float* __restrict__ gps) {
  const int l = blockIdx.y, r = blockIdx.x * DL_THREADS + threadIdx.x;
  if (r >= rows_per_layer) return;
  const size_t row = (size_t)l * rows_per_layer + r;
  const long long t = tgt[row];
  const float s = score[row], u = up[l] * scale;
  const float* x = ps + row * nc;
  float* g = gps + row * nc;
  for (int c = 0; c < nc; ++c) {
    const float xv = x[c], p = 1.f / (1.f + expf(-xv));
    const bool pos = c == t;
    const float tv = pos ? s : 0.f, w = pos ? s : 0.75f * p * p;
    const float dw = pos ? 0.f : 1.5f * p * p * (1.f - p);            // d(0.75 sigma(x)^2) / dx
    g[c] = u * (w * (p - tv) + bce_logits(xv, tv) * dw);
  }
}

// d(pb): rows of matched queries only (the caller zero-fills the rest)
__global__ __launch_bounds__(DL_THREADS) void detr_pairs_bwd_kernel(const float* __restrict__ pb, const float* __restrict__ gtb, const long long* __restrict__ li,
                                                                    const long long* __restrict__ bi, const long long* __restrict__ si,
                                                                    const long long* __restrict__ gi, int npairs, int B, int nq, const float* __restrict__ up_box,
                                                                    const float* __restrict__ up_iou, float s_box, float s_iou, float* __restrict__ gpb) {
  const int k = blockIdx.x * DL_THREADS + threadIdx.x;
  if (k >= npairs) return;
  const int l = (int)li[k];
  const size_t row = ((size_t)l * B + bi[k]) * nq + si[k];
  const Box a = ldbox(pb + row * 4), b = ldbox(gtb + (size_t)gi[k] * 4);
  float gr[4];
  riou_loss_grad(a, b, gr);
  const float d[4] = {a.x - b.x, a.y - b.y, a.w - b.w, a.h - b.h};
  const float ub = up_box[l] * s_box, ui = up_iou[l] * s_iou;
  float o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = ub * (d[j] > 0.f ? 1.f : d[j] < 0.f ? -1.f : 0.f) + ui * gr[j];
  *reinterpret_cast<float4*>(gpb + row * 4) = make_float4(o[0], o[1], o[2], o[3]);
}

// matcher cost (ops.py:84-112), all layers at once: C[l, b, q, g] over the flattened box list (the solver reads each image's own columns)
__global__ __launch_bounds__(DL_THREADS) void detr_match_cost_kernel(const float* __restrict__ ps, const float* __restrict__ pb, const float* __restrict__ gtb,
                                                                     const long long* __restrict__ gtc, long long rows, int nc, int G, float g_class,
                                                                     float g_bbox, float g_giou, float alpha, float gamma, float* __restrict__ C) {
  const long long e = (long long)blockIdx.x * DL_THREADS + threadIdx.x;
  if (e >= rows * G) return;
  const long long row = e / G;
  const int g = (int)(e - row * G);
  const float p = 1.f / (1.f + expf(-ps[row * nc + gtc[g]]));
  const float neg = (1.f - alpha) * powf(p, gamma) * (-logf(1.f - p + 1e-8f));
  const float pos = alpha * powf(1.f - p, gamma) * (-logf(p + 1e-8f));
  const Box a = ldbox(pb + row * 4), b = ldbox(gtb + (size_t)g * 4);
  const float l1 = fabsf(a.x - b.x) + fabsf(a.y - b.y) + fabsf(a.w - b.w) + fabsf(a.h - b.h);
  const float c = g_class * (pos - neg) + g_bbox * l1 + g_giou * (1.f - riou_xywh(a, b));
  C[e] = isfinite(c) ? c : 0.f;
}

// ---- iterative box refinement of the decoder (nn/modules/transformer.py:881-887: `refined = sigmoid(bbox_head(x) + inverse_sigmoid(ref))`,
// nn/modules/utils.py:46-52 inverse_sigmoid with eps 1e-5): as torch ops a clamp pair, two clamps, a division, a log, an add and a sigmoid
// on [B, Q, 4] - eight launches forward and as many backward, five times per step.  One kernel each way, the same formulas in fp32.
constexpr float REF_EPS = 1e-5f;
__global__ void box_refine_fwd_kernel(const float* __restrict__ d, const float* __restrict__ r, float* __restrict__ y, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = fminf(fmaxf(r[i], 0.f), 1.f);
  const float z = d[i] + logf(fmaxf(x, REF_EPS) / fmaxf(1.f - x, REF_EPS));
  y[i] = 1.f / (1.f + expf(-z));
}
// gd = gy y (1 - y);  gr = gd * d(inverse_sigmoid)/dr, with torch's clamp gradients (passed where the bound holds, bounds included)
__global__ void box_refine_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y, const float* __restrict__ r, float* __restrict__ gd,
                                      float* __restrict__ gr, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float yy = y[i], g = gy[i] * yy * (1.f - yy);
  gd[i] = g;
  if (gr) {
    const float rr = r[i], x = fminf(fmaxf(rr, 0.f), 1.f), om = 1.f - x;
    const float di = (x >= REF_EPS ? 1.f / x : 0.f) + (om >= REF_EPS ? 1.f / om : 0.f);
    gr[i] = (rr >= 0.f && rr <= 1.f) ? g * di : 0.f;
  }
}

}  // namespace

extern "C" int tamtr_detr_blocks(int rows_per_layer) { return rows_per_layer > 0 ? (rows_per_layer + DL_THREADS - 1) / DL_THREADS : 0; }

/* see include/tamtr_hip.h */
extern "C" int tamtr_detr_layers_fwd(const float* pb, const float* ps, const float* gt_bboxes, const long long* gt_cls, const long long* li,
                                     const long long* bi, const long long* si, const long long* gi, int Lr, int B, int nq, int nc, int n,
                                     long long* tgt, float* score, float* pair_l1, float* pair_riou, float* partial, float g_class, float g_bbox,
                                     float g_giou, float* out, void* stream) {
  if (!pb || !ps || !gt_bboxes || !gt_cls || !li || !bi || !si || !gi || !tgt || !score || !pair_l1 || !pair_riou || !partial || !out) return TAMTR_EINVAL;
  if (Lr <= 0 || B <= 0 || nq <= 0 || nc <= 0 || n <= 0 || Lr > 65535) return TAMTR_EINVAL;
  if (((uintptr_t)pb | (uintptr_t)gt_bboxes) % 16) return TAMTR_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
  const int np = Lr * n, rpl = B * nq, nblk = tamtr_detr_blocks(rpl);
  hipLaunchKernelGGL(detr_pairs_fwd_kernel, dim3((np + DL_THREADS - 1) / DL_THREADS), dim3(DL_THREADS), 0, s, pb, gt_bboxes, gt_cls, li, bi, si, gi, np, B, nq, tgt,
                     score, pair_l1, pair_riou);
  hipLaunchKernelGGL(detr_class_fwd_kernel, dim3(nblk, Lr), dim3(DL_THREADS), 0, s, ps, tgt, score, rpl, nc, partial);
  hipLaunchKernelGGL(detr_reduce_kernel, dim3(Lr), dim3(DL_THREADS), 0, s, partial, nblk, pair_l1, pair_riou, n, g_class, g_bbox, g_giou, out, Lr);
  return tamtr_launch_status();
}

extern "C" int tamtr_detr_layers_bwd(const float* pb, const float* ps, const float* gt_bboxes, const long long* li, const long long* bi,
                                     const long long* si, const long long* gi, const long long* tgt, const float* score, const float* up, int Lr, int B,
                                     int nq, int nc, int n, float g_class, float g_bbox, float g_giou, float* gpb, float* gps, void* stream) {
  if (!pb || !ps || !gt_bboxes || !li || !bi || !si || !gi || !tgt || !score || !up || !gpb || !gps) return TAMTR_EINVAL;
  if (Lr <= 0 || B <= 0 || nq <= 0 || nc <= 0 || n <= 0 || Lr > 65535) return TAMTR_EINVAL;
  if (((uintptr_t)pb | (uintptr_t)gt_bboxes | (uintptr_t)gpb) % 16) return TAMTR_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
  const int np = Lr * n, rpl = B * nq, nblk = tamtr_detr_blocks(rpl);
  // up = upstream gradients [3, Lr] of (class, bbox, giou)
  hipLaunchKernelGGL(detr_class_bwd_kernel, dim3(nblk, Lr), dim3(DL_THREADS), 0, s, ps, tgt, score, up, rpl, nc, g_class / (float)n, gps);
  hipLaunchKernelGGL(detr_pairs_bwd_kernel, dim3((np + DL_THREADS - 1) / DL_THREADS), dim3(DL_THREADS), 0, s, pb, gt_bboxes, li, bi, si, gi, np, B, nq, up + Lr,
                     up + 2 * Lr, g_bbox / (float)n, g_giou / (float)n, gpb);
  return tamtr_launch_status();
}

extern "C" int tamtr_detr_match_cost(const float* ps, const float* pb, const float* gt_bboxes, const long long* gt_cls, long long rows, int nc, int G,
                                     float g_class, float g_bbox, float g_giou, float alpha, float gamma, float* C, void* stream) {
  if (!ps || !pb || !gt_bboxes || !gt_cls || !C || rows <= 0 || nc <= 0 || G <= 0) return TAMTR_EINVAL;
  if (((uintptr_t)pb | (uintptr_t)gt_bboxes) % 16) return TAMTR_EUNSUP;
  const long long total = rows * G, blocks = (total + DL_THREADS - 1) / DL_THREADS;
  if (blocks > 0x7fffffffLL) return TAMTR_EUNSUP;
  hipLaunchKernelGGL(detr_match_cost_kernel, dim3((unsigned)blocks), dim3(DL_THREADS), 0, (hipStream_t)stream, ps, pb, gt_bboxes, gt_cls, rows, nc, G, g_class,
                     g_bbox, g_giou, alpha, gamma, C);
  return tamtr_launch_status();
}

/* see include/tamtr_hip.h */
extern "C" int tamtr_box_refine_fwd(const float* delta, const float* ref, float* out, long long n, void* stream) {
  if (!delta || !ref || !out || n <= 0) return TAMTR_EINVAL;
  if ((n + 255) / 256 > 0x7fffffffLL) return TAMTR_EUNSUP;
  hipLaunchKernelGGL(box_refine_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, delta, ref, out, n);
  return tamtr_launch_status();
}

extern "C" int tamtr_box_refine_bwd(const float* gout, const float* out, const float* ref, float* gdelta, float* gref, long long n, void* stream) {
  if (!gout || !out || !ref || !gdelta || n <= 0) return TAMTR_EINVAL;
  if ((n + 255) / 256 > 0x7fffffffLL) return TAMTR_EUNSUP;
  hipLaunchKernelGGL(box_refine_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gout, out, ref, gdelta, gref, n);
  return tamtr_launch_status();
}
