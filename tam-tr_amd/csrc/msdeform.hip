// msdeform.hip - fused multi-scale deformable-attention core (MEH decoder cross-attention), forward + backward, gfx950.
//
// Replaces multi_scale_deformable_attn_pytorch (reference: ultralytics/nn/modules/utils.py:42-89): per level a
// reshape/transpose of the whole value tensor, F.grid_sample, a stack and a weighted sum.  Here one gather kernel:
//   value [B, L, M, D] is kept in the layout the value projection writes it (token-major, head, channel), so one
//   sampled corner of one head is ONE contiguous row of D elements (256 B f32 / 128 B bf16 at D = 64);
//   forward : a group of D/VEC lanes owns one (b, q, head); each lane carries VEC channels (16-B loads), walks the
//             nl*P samples x 4 bilinear corners and accumulates in registers -> gather-bound (L2 / Infinity-Cache / HBM
//             row fetches), no LDS, no atomics;
//   backward: one lane per channel so that every float-atomic wave-instruction on grad_value covers whole contiguous
//             256-B (D=64) / 128-B (D=32) row segments - the full-rate atomic shape on gfx950 - while d/d(loc) and
//             d/d(weight) are reduced across the group with xor-shuffles (no LDS).
// Bilinear convention == grid_sample(align_corners=False, padding zeros): pixel coords x = loc_x*W - 0.5; a corner
// outside the map contributes 0 to the value and to every gradient.
#include "common.h"

namespace {

constexpr int MAXL = 8;
struct Levels {
  int H[MAXL], W[MAXL], start[MAXL];
};

constexpr int MSDA_THREADS = 256;

template <typename ET, int VEC>
struct VecLd;
template <>
struct VecLd<float, 4> {
  static __device__ __forceinline__ void ld(const float* p, float (&o)[4]) { Elt<float>::ld4(p, o); }
  static __device__ __forceinline__ void st(float* p, const float (&v)[4]) { Elt<float>::st4(p, v); }
};
template <>
struct VecLd<bf16_t, 8> {
  static __device__ __forceinline__ void ld(const bf16_t* p, float (&o)[8]) {
    uint4 t = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float (&v)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f2bf(v[2 * i]) | ((uint32_t)f2bf(v[2 * i + 1]) << 16);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};

// LPG lanes per (b,q,m) item, each lane VEC channels; D <= LPG*VEC (tail lanes idle).
template <typename ET, int VEC, int LPG>
__global__ __launch_bounds__(MSDA_THREADS) void msda_fwd_kernel(const ET* __restrict__ value, const float* __restrict__ loc,
                                                                 const float* __restrict__ aw, ET* __restrict__ out, Levels lv,
                                                                 int n_items, int L, int M, int D, int Q, int nl, int P) {
  const int gid = (blockIdx.x * MSDA_THREADS + threadIdx.x) / LPG;
  const int gl = threadIdx.x % LPG;
  if (gid >= n_items) return;
  const int d0 = gl * VEC;
  const bool act = d0 < D;
  const int m = gid % M;
  const int bq = gid / M;
  const int b = bq / Q;
  const float* lp = loc + (size_t)gid * nl * P * 2;
  const float* ap = aw + (size_t)gid * nl * P;
  const ET* vb = value + ((size_t)b * L * M + m) * D + (act ? d0 : 0);
  const size_t tok = (size_t)M * D;
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  for (int l = 0; l < nl; ++l) {
    const int H = lv.H[l], W = lv.W[l];
    const ET* vl = vb + (size_t)lv.start[l] * tok;
#pragma unroll 4
    for (int p = 0; p < P; ++p) {
      const float2 xy = *reinterpret_cast<const float2*>(lp + (l * P + p) * 2);
      const float a = ap[l * P + p];
      const float x = xy.x * W - 0.5f, y = xy.y * H - 0.5f;
      const float xf = floorf(x), yf = floorf(y);
      const float fx = x - xf, fy = y - yf;
      const int x0 = (int)xf, y0 = (int)yf;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int xi = x0 + (c & 1), yi = y0 + (c >> 1);
        const bool ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H);
        const float w = ((c & 1) ? fx : 1.f - fx) * ((c >> 1) ? fy : 1.f - fy) * a;
        if (ok) {  // group-uniform
          float vv[VEC];
          VecLd<ET, VEC>::ld(vl + (size_t)(yi * W + xi) * tok, vv);
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] = fmaf(w, vv[i], acc[i]);
        }
      }
    }
  }
  if (act) VecLd<ET, VEC>::st(out + (size_t)gid * D + d0, acc);
}

// one lane per channel (LPG = min(64, pow2ceil(D)) lanes per item; D > 64 loops)
template <typename ET, int LPG>
__global__ __launch_bounds__(MSDA_THREADS) void msda_bwd_kernel(const ET* __restrict__ gout, const ET* __restrict__ value,
                                                                 const float* __restrict__ loc, const float* __restrict__ aw,
                                                                 float* __restrict__ gvalue, float* __restrict__ gloc,
                                                                 float* __restrict__ gaw, Levels lv, int n_items, int L, int M,
                                                                 int D, int Q, int nl, int P) {
  const int gid = (blockIdx.x * MSDA_THREADS + threadIdx.x) / LPG;
  const int gl = threadIdx.x % LPG;
  if (gid >= n_items) return;  // whole groups leave together (LPG divides the block)
  const int m = gid % M;
  const int b = (gid / M) / Q;
  const float* lp = loc + (size_t)gid * nl * P * 2;
  const float* ap = aw + (size_t)gid * nl * P;
  const size_t tok = (size_t)M * D;
  const size_t hb = ((size_t)b * L * M + m) * D;
  for (int l = 0; l < nl; ++l) {
    const int H = lv.H[l], W = lv.W[l];
    const size_t lb = hb + (size_t)lv.start[l] * tok;
    for (int p = 0; p < P; ++p) {
      const float2 xy = *reinterpret_cast<const float2*>(lp + (l * P + p) * 2);
      const float a = ap[l * P + p];
      const float x = xy.x * W - 0.5f, y = xy.y * H - 0.5f;
      const float xf = floorf(x), yf = floorf(y);
      const float fx = x - xf, fy = y - yf;
      const int x0 = (int)xf, y0 = (int)yf;
      float s_aw = 0.f, s_x = 0.f, s_y = 0.f;
      for (int d = gl; d < D; d += LPG) {
        const float g = Elt<ET>::ld(gout + (size_t)gid * D + d);
        float vc[4];
        size_t oc[4];
        bool okc[4];
        // the four corner loads go out together (clamped addresses, zeroed afterwards): a load inside `if (ok)` is followed by
        // s_waitcnt vmcnt(0), i.e. four serial round trips per sampling point
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int xi = x0 + (c & 1), yi = y0 + (c >> 1);
          okc[c] = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H);
          oc[c] = lb + (size_t)(min(max(yi, 0), H - 1) * W + min(max(xi, 0), W - 1)) * tok + d;
          vc[c] = Elt<ET>::ld(value + oc[c]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (okc[c]) {  // group-uniform
            const float w = ((c & 1) ? fx : 1.f - fx) * ((c >> 1) ? fy : 1.f - fy);
            atomicAdd(gvalue + oc[c], w * a * g);
          } else {
            vc[c] = 0.f;
          }
        }
        // v(x,y) = (1-fy)((1-fx) v00 + fx v01) + fy((1-fx) v10 + fx v11);  vc = {v00, v01, v10, v11}
        s_aw = fmaf(g, (1.f - fy) * ((1.f - fx) * vc[0] + fx * vc[1]) + fy * ((1.f - fx) * vc[2] + fx * vc[3]), s_aw);
        s_x = fmaf(g, (1.f - fy) * (vc[1] - vc[0]) + fy * (vc[3] - vc[2]), s_x);
        s_y = fmaf(g, (1.f - fx) * (vc[2] - vc[0]) + fx * (vc[3] - vc[1]), s_y);
      }
      s_aw = group_sum<LPG>(s_aw);
      s_x = group_sum<LPG>(s_x);
      s_y = group_sum<LPG>(s_y);
      if (gl == 0) {
        gaw[(size_t)gid * nl * P + l * P + p] = s_aw;
        *reinterpret_cast<float2*>(gloc + ((size_t)gid * nl * P + l * P + p) * 2) = make_float2(s_x * a * W, s_y * a * H);
      }
    }
  }
}

bool make_levels(const int32_t* shapes, int nl, int L, Levels& lv) {
  if (nl < 1 || nl > MAXL) return false;
  int s = 0;
  for (int l = 0; l < nl; ++l) {
    lv.H[l] = shapes[2 * l];
    lv.W[l] = shapes[2 * l + 1];
    lv.start[l] = s;
    if (lv.H[l] <= 0 || lv.W[l] <= 0) return false;
    s += lv.H[l] * lv.W[l];
  }
  return s == L;  // the reference asserts the same (transformer.py:271)
}

}  // namespace

extern "C" int tamtr_msdeform_attn_fwd(const void* value, const int32_t* shapes_host, const float* loc, const float* aw,
                                       void* out, int B, int L, int M, int D, int Q, int nl, int P, int dtype, void* stream) {
  if (!value || !shapes_host || !loc || !aw || !out || B <= 0 || L <= 0 || M <= 0 || D <= 0 || Q <= 0 || P <= 0)
    return TAMTR_EINVAL;
  Levels lv;
  if (!make_levels(shapes_host, nl, L, lv)) return TAMTR_EINVAL;
  const long long n_items = (long long)B * Q * M;
  if (n_items > (1ll << 30)) return TAMTR_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
#define GO(ET, VEC, LPG)                                                                                                  \
  {                                                                                                                       \
    const int per_blk = MSDA_THREADS / LPG;                                                                               \
    dim3 grid((unsigned)((n_items + per_blk - 1) / per_blk));                                                             \
    hipLaunchKernelGGL((msda_fwd_kernel<ET, VEC, LPG>), grid, dim3(MSDA_THREADS), 0, s, (const ET*)value, loc, aw, (ET*)out, \
                       lv, (int)n_items, L, M, D, Q, nl, P);                                                              \
  }
  if (dtype == TAMTR_F32) {
    if (D % 4 || D > 256) return TAMTR_EUNSUP;
    const int lanes = D / 4;
    if (lanes <= 4) GO(float, 4, 4) else if (lanes <= 8) GO(float, 4, 8) else if (lanes <= 16) GO(float, 4, 16)
    else if (lanes <= 32) GO(float, 4, 32) else GO(float, 4, 64)
  } else if (dtype == TAMTR_BF16) {
    if (D % 8 || D > 256) return TAMTR_EUNSUP;
    const int lanes = D / 8;
    if (lanes <= 4) GO(bf16_t, 8, 4) else if (lanes <= 8) GO(bf16_t, 8, 8) else if (lanes <= 16) GO(bf16_t, 8, 16)
    else GO(bf16_t, 8, 32)
  } else {
    return TAMTR_EINVAL;
  }
#undef GO
  return tamtr_launch_status();
}

extern "C" int tamtr_msdeform_attn_bwd(const void* gout, const void* value, const int32_t* shapes_host, const float* loc,
                                       const float* aw, float* gvalue, float* gloc, float* gaw, int B, int L, int M, int D, int Q,
                                       int nl, int P, int dtype, void* stream) {
  if (!gout || !value || !shapes_host || !loc || !aw || !gvalue || !gloc || !gaw || B <= 0 || L <= 0 || M <= 0 || D <= 0 ||
      Q <= 0 || P <= 0)
    return TAMTR_EINVAL;
  Levels lv;
  if (!make_levels(shapes_host, nl, L, lv)) return TAMTR_EINVAL;
  const long long n_items = (long long)B * Q * M;
  if (n_items > (1ll << 30) || D > 256) return TAMTR_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
#define GO(ET, LPG)                                                                                                       \
  {                                                                                                                       \
    const int per_blk = MSDA_THREADS / LPG;                                                                               \
    dim3 grid((unsigned)((n_items + per_blk - 1) / per_blk));                                                             \
    hipLaunchKernelGGL((msda_bwd_kernel<ET, LPG>), grid, dim3(MSDA_THREADS), 0, s, (const ET*)gout, (const ET*)value, loc, aw, \
                       gvalue, gloc, gaw, lv, (int)n_items, L, M, D, Q, nl, P);                                           \
  }
#define PICK(ET)                                                                                           \
  if (D <= 8) GO(ET, 8) else if (D <= 16) GO(ET, 16) else if (D <= 32) GO(ET, 32) else GO(ET, 64)
  if (dtype == TAMTR_F32) { PICK(float) }
  else if (dtype == TAMTR_BF16) { PICK(bf16_t) }
  else return TAMTR_EINVAL;
#undef PICK
#undef GO
  return tamtr_launch_status();
}
