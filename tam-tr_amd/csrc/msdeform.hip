// msdeform.hip - fused multi-scale deformable-attention core (MEH decoder cross-attention), forward + backward, gfx950.
//
// Replaces multi_scale_deformable_attn_pytorch (reference: ultralytics/nn/modules/utils.py:42-89): per level a
// reshape/transpose of the whole value tensor, F.grid_sample, a stack and a weighted sum.  Here one gather kernel:
//   value [B, L, M, D] is kept in the layout the value projection writes it (token-major, head, channel), so one
//   sampled corner of one head is ONE contiguous row of D elements (256 B f32 / 128 B bf16 at D = 64);
//   forward : a group of D/VEC lanes owns one (b, q, head); each lane carries VEC channels (16-B loads), walks the
//             nl*P samples x 4 bilinear corners and accumulates in registers -> gather-bound (L2 / Infinity-Cache / HBM
//             row fetches), no LDS, no atomics;
//   backward (round 3, tamtr_msdeform_attn_bwd_sorted): no atomics - d/d(loc), d/d(weight) in the forward's lane layout with
//             xor-shuffle sums (msda_bwd_locaw_kernel); d/d(value) as a sorted segmented sum written once per element in the value's
//             dtype (msda_gvalue_sorted_kernel: keys sorted in registers / LDS per (image, head, level), every row sums its run);
//   backward (rounds 1-2, tamtr_msdeform_attn_bwd, kept as the reference the tests compare with): one lane per channel so that
//             every float-atomic wave-instruction on grad_value covers whole contiguous 256-B (D=64) / 128-B (D=32) row segments
//             - the full-rate atomic shape on gfx950 - d/d(loc) and d/d(weight) reduced across the group with xor-shuffles.
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

// d/d(loc), d/d(weight) alone (the backward's gather half once d/d(value) has its own kernel): the forward's lane layout - LPG lanes per
// (b, q, head) item, VEC channels and 16-byte loads each - instead of one lane per channel, which was the shape the float atomics wanted.
template <typename ET, int VEC, int LPG>
__global__ __launch_bounds__(MSDA_THREADS) void msda_bwd_locaw_kernel(const ET* __restrict__ gout, const ET* __restrict__ value,
                                                                       const float* __restrict__ loc, const float* __restrict__ aw,
                                                                       float* __restrict__ gloc, float* __restrict__ gaw,
                                                                       float* __restrict__ colw, Levels lv,
                                                                       int n_items, int L, int M, int D, int Q, int nl, int P) {
  const int gid = (blockIdx.x * MSDA_THREADS + threadIdx.x) / LPG;
  const int gl = threadIdx.x % LPG;
  if (gid >= n_items) return;  // whole groups leave together
  const int d0 = gl * VEC;
  const bool act = d0 < D;
  const int m = gid % M;
  const int b = (gid / M) / Q;
  const float* lp = loc + (size_t)gid * nl * P * 2;
  const float* ap = aw + (size_t)gid * nl * P;
  const ET* vb = value + ((size_t)b * L * M + m) * D + (act ? d0 : 0);
  const size_t tok = (size_t)M * D;
  float g[VEC];
  VecLd<ET, VEC>::ld(gout + (size_t)gid * D + (act ? d0 : 0), g);
#pragma unroll
  for (int i = 0; i < VEC; ++i) g[i] = act ? g[i] : 0.f;
  float cw = 0.f;   // the item's total weight on the map: sum over (level, point, corner on the map) of a * bilinear weight
  for (int l = 0; l < nl; ++l) {
    const int H = lv.H[l], W = lv.W[l];
    const ET* vl = vb + (size_t)lv.start[l] * tok;
    for (int p = 0; p < P; ++p) {
      const float2 xy = *reinterpret_cast<const float2*>(lp + (l * P + p) * 2);
      const float a = ap[l * P + p];
      const float x = xy.x * W - 0.5f, y = xy.y * H - 0.5f;
      const float xf = floorf(x), yf = floorf(y);
      const float fx = x - xf, fy = y - yf;
      const int x0 = (int)xf, y0 = (int)yf;
      // the four corners' dot products with gout go out together (clamped addresses, zero factor for corners off the map)
      float dc[4];
      float vv[4][VEC];
      bool okc[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int xi = x0 + (c & 1), yi = y0 + (c >> 1);
        okc[c] = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H);
        VecLd<ET, VEC>::ld(vl + (size_t)(min(max(yi, 0), H - 1) * W + min(max(xi, 0), W - 1)) * tok, vv[c]);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) d = fmaf(g[i], vv[c][i], d);
        dc[c] = okc[c] ? d : 0.f;
      }
      cw = fmaf(a, (okc[0] ? (1.f - fx) * (1.f - fy) : 0.f) + (okc[1] ? fx * (1.f - fy) : 0.f) + (okc[2] ? (1.f - fx) * fy : 0.f) +
                       (okc[3] ? fx * fy : 0.f), cw);
      // v(x,y) = (1-fy)((1-fx) v00 + fx v01) + fy((1-fx) v10 + fx v11);  dc = <gout, {v00, v01, v10, v11}>
      float s_aw = (1.f - fy) * ((1.f - fx) * dc[0] + fx * dc[1]) + fy * ((1.f - fx) * dc[2] + fx * dc[3]);
      float s_x = (1.f - fy) * (dc[1] - dc[0]) + fy * (dc[3] - dc[2]);
      float s_y = (1.f - fx) * (dc[2] - dc[0]) + fx * (dc[3] - dc[1]);
      s_aw = group_sum<LPG>(s_aw);
      s_x = group_sum<LPG>(s_x);
      s_y = group_sum<LPG>(s_y);
      if (gl == 0) {
        gaw[(size_t)gid * nl * P + l * P + p] = s_aw;
        *reinterpret_cast<float2*>(gloc + ((size_t)gid * nl * P + l * P + p) * 2) = make_float2(s_x * a * W, s_y * a * H);
      }
    }
  }
  if (colw && gl == 0) colw[gid] = cw;
}

// ---------------------------------------------------------------------------------------------------------------------------
// d/d(value) WITHOUT atomics (and without a zero fill, an fp32 accumulator or a cast): every row of the gradient is written exactly
// once, in the value's own dtype, as an ordered sum.
//   One workgroup per (image, head, row slice of one level).  The contributions of that (image, head) to the level are the
//   Q*P*4 bilinear corners (<= 8192): the workgroup computes them into LDS as 32-bit keys (row inside the level << 13 | corner
//   index; corners off the map get the all-ones key), sorts the keys (bitonic, in LDS), finds every row's run of keys by binary
//   search, and then walks its rows: a group of D/8 lanes owns a row, adds weight * gout[b, q, m, :] over the row's run IN KEY
//   ORDER (= a fixed order: bitwise reproducible) and stores the row - zeros for the (majority of) rows nobody sampled.
//   Row slices (<= MSDA_SLICE_ROWS rows) only spread the emission over more workgroups; each re-sorts its level's keys.
constexpr int MSDA_SORT_THREADS = 1024;
constexpr int MSDA_MAX_KEYS = 8192;      // corners per (image, head, level): Q * P * 4
constexpr int MSDA_IDX_BITS = 13;
#ifndef MSDA_SLICE
#define MSDA_SLICE 12800   // measured at the MEH shape (tools/build_variant.sh msdeform): 3200: 482, 6400: 399, 12800: 383, 25600: 416 us fwd+bwd
#endif
constexpr int MSDA_SLICE_ROWS = MSDA_SLICE;

struct SliceTab {
  int first[MAXL + 1];  // prefix sums of the slices per level; first[nl] = slices in total
};

// ---- sort of exactly 8192 keys by 1024 threads, 8 keys per thread (thread t holds positions 8 t .. 8 t + 7).  The same bitonic network as
// the plain LDS loop in the kernel, but only its strides >= 512 (partner in another wave: 10 of the 91 passes) go through LDS and a
// barrier; strides 8 .. 256 are lane exchanges inside the wave (ds_bpermute, no barrier) and strides 4, 2, 1 stay inside the thread.
// Measured on the MEH shape: the LDS loop was 158 us of the kernel's ~355 us (tools/build_variant.sh msdeform ablations).
template <int J>
__device__ __forceinline__ void cx_in_thread(uint32_t (&k)[8], bool up_t, int kk) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if ((i & J) == 0) {
      const bool up = kk >= 8 ? up_t : ((i & kk) == 0);
      const uint32_t a = k[i], b = k[i | J];
      const uint32_t lo = min(a, b), hi = max(a, b);
      k[i] = up ? lo : hi;
      k[i | J] = up ? hi : lo;
    }
  }
}

__device__ __forceinline__ void bitonic_sort_8192(uint32_t* __restrict__ keys, int tid) {
  uint32_t k[8];
  {
    const uint4 a = *reinterpret_cast<const uint4*>(keys + tid * 8), b = *reinterpret_cast<const uint4*>(keys + tid * 8 + 4);
    k[0] = a.x; k[1] = a.y; k[2] = a.z; k[3] = a.w; k[4] = b.x; k[5] = b.y; k[6] = b.z; k[7] = b.w;
  }
  cx_in_thread<1>(k, false, 2);
  cx_in_thread<2>(k, false, 4);
  cx_in_thread<1>(k, false, 4);
  for (int kk = 8; kk <= 8192; kk <<= 1) {
    const bool up_t = (tid & (kk >> 3)) == 0;   // direction of this thread's 8 positions in stage kk
    for (int j = kk >> 1; j >= 8; j >>= 1) {
      const int dt = j >> 3;                    // partner thread = tid ^ dt
      uint32_t pv[8];
      if (dt >= WAVE) {                         // another wave: through LDS (wave-uniform branch)
        __syncthreads();
        *reinterpret_cast<uint4*>(keys + tid * 8) = make_uint4(k[0], k[1], k[2], k[3]);
        *reinterpret_cast<uint4*>(keys + tid * 8 + 4) = make_uint4(k[4], k[5], k[6], k[7]);
        __syncthreads();
        const uint4 a = *reinterpret_cast<const uint4*>(keys + (tid ^ dt) * 8), b = *reinterpret_cast<const uint4*>(keys + (tid ^ dt) * 8 + 4);
        pv[0] = a.x; pv[1] = a.y; pv[2] = a.z; pv[3] = a.w; pv[4] = b.x; pv[5] = b.y; pv[6] = b.z; pv[7] = b.w;
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) pv[i] = (uint32_t)__shfl_xor((int)k[i], dt, WAVE);
      }
      const bool keep_min = ((tid & dt) == 0) == up_t;   // the lower position of a pair keeps the minimum in an ascending block
#pragma unroll
      for (int i = 0; i < 8; ++i) k[i] = keep_min ? min(k[i], pv[i]) : max(k[i], pv[i]);
    }
    cx_in_thread<4>(k, up_t, kk);
    cx_in_thread<2>(k, up_t, kk);
    cx_in_thread<1>(k, up_t, kk);
  }
  __syncthreads();
  *reinterpret_cast<uint4*>(keys + tid * 8) = make_uint4(k[0], k[1], k[2], k[3]);
  *reinterpret_cast<uint4*>(keys + tid * 8 + 4) = make_uint4(k[4], k[5], k[6], k[7]);
  __syncthreads();
}

// STAGE: the Q x D block of gout that this (image, head) reads is copied into LDS first (37 KB at Q = 292, D = 64, bf16).  Without it
// every corner of a row's run costs a dependent global load (key -> corner -> gout row: ~1.5 us each, one after the other in the row's
// 8 lanes): at 36 corners per lane group on the two coarse levels the kernel spent most of its 355 us waiting for those, not storing.
constexpr int MSDA_FIXED_LDS = MSDA_MAX_KEYS * 8 + ((MSDA_SLICE_ROWS + 2) * 2 + 15) / 16 * 16;   // keys | weights | first-key table

template <typename ET, int LPR, bool STAGE>  // LPR lanes per row, 8 channels each
__global__ __launch_bounds__(MSDA_SORT_THREADS) void msda_gvalue_sorted_kernel(const ET* __restrict__ gout, const float* __restrict__ loc,
                                                                                const float* __restrict__ aw, ET* __restrict__ gvalue,
                                                                                Levels lv, SliceTab tab, int B, int L, int M, int D, int Q,
                                                                                int nl, int P, int NS, long long ldg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char msda_lds[];
  uint32_t* keys = reinterpret_cast<uint32_t*>(msda_lds);
  float* wts = reinterpret_cast<float*>(msda_lds + MSDA_MAX_KEYS * 4);
  uint16_t* first_key = reinterpret_cast<uint16_t*>(msda_lds + MSDA_MAX_KEYS * 8);
  ET* sg = reinterpret_cast<ET*>(msda_lds + MSDA_FIXED_LDS);   // [Q][D] (STAGE only)
  const int tid = threadIdx.x;
  // workgroup id -> (image, slice, head), XCD-aware: the hardware deals consecutive workgroup ids round-robin over the 8 XCDs, so
  // id = xcd + 8 * (m + M * t) puts the M heads of one (image, slice) on ONE XCD, back to back in dispatch order.  Their stores are the M
  // 128-byte pieces of the same 1 KB token rows: meeting in one L2 they leave for HBM as whole rows (with the heads spread over the
  // XCDs every 128-byte line went out alone: 1.5 TB/s).
  const int nslices = tab.first[nl];
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int m = q % M, bs = (q / M) * 8 + xcd;   // bs = b * nslices + slice
  if (bs >= B * nslices) return;                  // (grid padded to a multiple of 8 (image, slice) pairs; whole workgroups leave)
  const int b = bs / nslices, sl = bs - b * nslices;
  int l = 0;
  while (l + 1 < nl && sl >= tab.first[l + 1]) ++l;
  const int H = lv.H[l], W = lv.W[l];
  const int rows = H * W;
  const int nsl = tab.first[l + 1] - tab.first[l];
  const int per = (rows + nsl - 1) / nsl;
  const int lo = (sl - tab.first[l]) * per;
  const int hi = min(lo + per, rows);
  if (STAGE) {  // gout[b, :, m, :] -> LDS, 16 bytes per thread and trip (consumed only after the barriers of the sort)
    const int vpr = D * (int)sizeof(ET) / 16;   // 16-byte pieces per query row
    for (int i = tid; i < Q * vpr; i += MSDA_SORT_THREADS) {
      const int q = i / vpr, v = i - q * vpr;
      reinterpret_cast<uint4*>(sg)[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(gout + (((size_t)b * Q + q) * M + m) * D) + v * 16);
    }
  }
  // ---- the level's corners of this (image, head)
  const int n_pts = Q * P;
  for (int e = tid; e < n_pts; e += MSDA_SORT_THREADS) {
    const int q = e / P, p = e - q * P;
    const size_t at = ((((size_t)b * Q + q) * M + m) * nl + l) * P + p;
    const float2 xy = *reinterpret_cast<const float2*>(loc + at * 2);
    const float a = aw[at];
    const float x = xy.x * W - 0.5f, y = xy.y * H - 0.5f;
    const float xf = floorf(x), yf = floorf(y);
    const float fx = x - xf, fy = y - yf;
    // floorf of a huge / non-finite coordinate: clamp before the int conversion (any value outside the map is "off")
    const int x0 = (int)fminf(fmaxf(xf, -2.f), (float)W + 1.f), y0 = (int)fminf(fmaxf(yf, -2.f), (float)H + 1.f);
    const bool fin = (x == x) & (y == y);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int xi = x0 + (c & 1), yi = y0 + (c >> 1);
      const bool ok = fin & (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H);
      const int idx = e * 4 + c;
      keys[idx] = ok ? (((uint32_t)(yi * W + xi) << MSDA_IDX_BITS) | (uint32_t)idx) : 0xffffffffu;
      wts[idx] = ((c & 1) ? fx : 1.f - fx) * ((c >> 1) ? fy : 1.f - fy) * a;
    }
  }
  for (int i = n_pts * 4 + tid; i < NS; i += MSDA_SORT_THREADS) keys[i] = 0xffffffffu;
  __syncthreads();
  // ---- bitonic sort of NS keys, ascending   (MSDA_ABL_*: timing-only builds of tools/build_variant.sh msdeform, results are wrong)
#ifndef MSDA_ABL_NOSORT
  if (NS == MSDA_MAX_KEYS) {
    bitonic_sort_8192(keys, tid);
  } else
  for (int k = 2; k <= NS; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (NS >> 1); t += MSDA_SORT_THREADS) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // the lower index of pair t at distance j
        const int ix = i | j;
        const uint32_t ka = keys[i], kb = keys[ix];
        const bool up = (i & k) == 0;
        if ((ka > kb) == up) { keys[i] = kb; keys[ix] = ka; }
      }
      __syncthreads();
    }
  }
#endif
  // ---- first key of every row of the slice (and of the row after it)
  for (int r = lo + tid; r <= hi; r += MSDA_SORT_THREADS) {
    const uint32_t want = (uint32_t)r << MSDA_IDX_BITS;
    int a0 = 0, n = NS;           // lower_bound(keys, want)
    while (n > 0) {
      const int h = n >> 1;
      if (keys[a0 + h] < want) { a0 += h + 1; n -= h + 1; } else n = h;
    }
    first_key[r - lo] = (uint16_t)a0;
  }
  __syncthreads();
  // ---- the rows
  const int g = tid / LPR, ln = tid % LPR;
  const int d0 = ln * 8;
  const bool act = d0 < D;
  const size_t gb = (size_t)b * Q * M * D + (size_t)m * D + (act ? d0 : 0);
  ET* const ob = gvalue + ((size_t)b * L + lv.start[l]) * ldg + (size_t)m * D + (act ? d0 : 0);
#ifdef MSDA_ABL_NOEMIT
  if (lo >= 0) return;
#endif
  for (int r = lo + g; r < hi; r += MSDA_SORT_THREADS / LPR) {
#ifdef MSDA_ABL_NORUNS
    const int s = 0, e = 0;
#else
    const int s = first_key[r - lo], e = first_key[r - lo + 1];
#endif
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int i = s; i < e; ++i) {
      const int idx = (int)(keys[i] & (MSDA_MAX_KEYS - 1));
      const float w = wts[idx];
      const int q = idx / (4 * P);
      float gv[8];
      if (sizeof(ET) == 2) {
        VecLd<bf16_t, 8>::ld(STAGE ? reinterpret_cast<const bf16_t*>(sg) + q * D + (act ? d0 : 0)
                                   : reinterpret_cast<const bf16_t*>(gout) + gb + (size_t)q * M * D, gv);
      } else {
        const float* gp = STAGE ? reinterpret_cast<const float*>(sg) + q * D + (act ? d0 : 0)
                                : reinterpret_cast<const float*>(gout) + gb + (size_t)q * M * D;
        float lo4[4], hi4[4];
        Elt<float>::ld4(gp, lo4);
        Elt<float>::ld4(gp + 4, hi4);
#pragma unroll
        for (int t = 0; t < 4; ++t) { gv[t] = lo4[t]; gv[4 + t] = hi4[t]; }
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = fmaf(w, gv[t], acc[t]);
    }
    if (act) {
      if (sizeof(ET) == 2) {
        VecLd<bf16_t, 8>::st(reinterpret_cast<bf16_t*>(ob) + (size_t)r * ldg, acc);
      } else {
        float* op = reinterpret_cast<float*>(ob) + (size_t)r * ldg;
        const float lo4[4] = {acc[0], acc[1], acc[2], acc[3]}, hi4[4] = {acc[4], acc[5], acc[6], acc[7]};
        Elt<float>::st4(op, lo4);
        Elt<float>::st4(op + 4, hi4);
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

// Deterministic backward: gloc / gaw as above (no scatter), gvalue by the sorted segmented sum - written ONCE per element in the
// value's dtype (no zero fill by the caller, no float atomics, bitwise reproducible).  ldg: token pitch of gvalue in elements
// (M*D when it is its own [B,L,M,D] tensor; larger when it is a column block of a wider [B*L, ldg] matrix).
extern "C" int tamtr_msdeform_attn_bwd_sorted(const void* gout, const void* value, const int32_t* shapes_host, const float* loc,
                                              const float* aw, void* gvalue, float* gloc, float* gaw, float* colw, int B, int L, int M,
                                              int D, int Q, int nl, int P, long long ldg, int dtype, void* stream) {
  if (!gout || !value || !shapes_host || !loc || !aw || !gvalue || !gloc || !gaw || B <= 0 || L <= 0 || M <= 0 || D <= 0 ||
      Q <= 0 || P <= 0 || ldg < (long long)M * D)
    return TAMTR_EINVAL;
  Levels lv;
  if (!make_levels(shapes_host, nl, L, lv)) return TAMTR_EINVAL;
  const long long n_items = (long long)B * Q * M;
  if (n_items > (1ll << 30) || D > 256 || D % 8 || (long long)Q * P * 4 > MSDA_MAX_KEYS || M > 65535 || B > 65535 ||
      (dtype != TAMTR_F32 && dtype != TAMTR_BF16) || (ldg % (dtype == TAMTR_BF16 ? 8 : 4)))
    return TAMTR_EUNSUP;
  SliceTab tab;
  tab.first[0] = 0;
  for (int l = 0; l < nl; ++l) {
    const long long rows = (long long)lv.H[l] * lv.W[l];
    if (rows >= (1ll << (32 - MSDA_IDX_BITS)) - 1) return TAMTR_EUNSUP;
    tab.first[l + 1] = tab.first[l] + (int)((rows + MSDA_SLICE_ROWS - 1) / MSDA_SLICE_ROWS);
  }
  int NS = 64;
  while (NS < Q * P * 4) NS <<= 1;
  hipStream_t s = (hipStream_t)stream;
  {  // d/d(loc), d/d(weight): the gather half, in the forward's vector layout
#define GOL(ET, VEC, LPG)                                                                                                 \
  {                                                                                                                       \
    const int per_blk = MSDA_THREADS / LPG;                                                                               \
    dim3 g1((unsigned)((n_items + per_blk - 1) / per_blk));                                                               \
    hipLaunchKernelGGL((msda_bwd_locaw_kernel<ET, VEC, LPG>), g1, dim3(MSDA_THREADS), 0, s, (const ET*)gout, (const ET*)value, loc, aw, \
                       gloc, gaw, colw, lv, (int)n_items, L, M, D, Q, nl, P);                                             \
  }
    if (dtype == TAMTR_F32) {
      const int lanes = D / 4;
      if (lanes <= 4) GOL(float, 4, 4) else if (lanes <= 8) GOL(float, 4, 8) else if (lanes <= 16) GOL(float, 4, 16)
      else if (lanes <= 32) GOL(float, 4, 32) else GOL(float, 4, 64)
    } else {
      const int lanes = D / 8;
      if (lanes <= 4) GOL(bf16_t, 8, 4) else if (lanes <= 8) GOL(bf16_t, 8, 8) else if (lanes <= 16) GOL(bf16_t, 8, 16)
      else GOL(bf16_t, 8, 32)
    }
#undef GOL
  }
  const long long pairs = ((long long)B * tab.first[nl] + 7) / 8 * 8;   // (image, slice) pairs, padded to the 8 XCDs
  if (pairs * M > 0x7fffffffLL) return TAMTR_EUNSUP;
  dim3 grid((unsigned)(pairs * M));
  // gout of one (image, head) staged in LDS when it fits next to the sort (one workgroup per CU then: 16 waves)
  const size_t stage_bytes = (size_t)Q * D * (dtype == TAMTR_BF16 ? 2 : 4);
  const bool stage = MSDA_FIXED_LDS + stage_bytes <= 150 * 1024 && (D * (dtype == TAMTR_BF16 ? 2 : 4)) % 16 == 0;
  const size_t lds = MSDA_FIXED_LDS + (stage ? stage_bytes : 0);
#define GO1(ET, LPR, ST)                                                                                                     \
  {                                                                                                                          \
    if (lds > 64 * 1024 &&                                                                                                   \
        hipFuncSetAttribute((const void*)msda_gvalue_sorted_kernel<ET, LPR, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return TAMTR_ELAUNCH;                                                                                                  \
    hipLaunchKernelGGL((msda_gvalue_sorted_kernel<ET, LPR, ST>), grid, dim3(MSDA_SORT_THREADS), lds, s, (const ET*)gout, loc, aw, (ET*)gvalue, \
                       lv, tab, B, L, M, D, Q, nl, P, NS, ldg);                                                              \
  }
#define GO(ET, LPR) { if (stage) GO1(ET, LPR, true) else GO1(ET, LPR, false) }
#define PICK(ET)                                                                                                             \
  if (D <= 8) GO(ET, 1) else if (D <= 16) GO(ET, 2) else if (D <= 32) GO(ET, 4) else if (D <= 64) GO(ET, 8)                  \
  else if (D <= 128) GO(ET, 16) else GO(ET, 32)
  if (dtype == TAMTR_F32) { PICK(float) } else { PICK(bf16_t) }
#undef PICK
#undef GO
#undef GO1
  return tamtr_launch_status();
}
