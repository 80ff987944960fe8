// selfattn.hip - masked multi-head self-attention over the <= ~300 decoder queries (MEH), forward + backward, gfx950.
//
// Replaces the attention core of nn.MultiheadAttention as called at ultralytics/nn/modules/transformer.py:546 (which,
// with need_weights=True, materialises [B*nh, Q, Q] probabilities and averages them over heads for a result the
// reference throws away).  The problem is tiny and latency-bound (Q ~ 292, dh = 64: 2.9 GFLOP per layer), so the design
// goal is ONE launch with everything on chip, not MFMA throughput:
//   * one lane (dh = 32) or a lane pair (dh = 64) owns one query row: q row and the output accumulator live in registers, flash-style online softmax,
//     so max / sum / rescale never leave the lane (no cross-lane reductions at all);
//   * K/V are streamed through LDS in 32-key tiles and read back as wave-uniform broadcasts (conflict-free);
//   * the boolean mask arrives bit-packed (one u32 per query row per 32-key tile);
//   * backward = two launches, both atomics-free and deterministic: dQ with a lane per query row, dK/dV with a lane
//     per key row (the query tile is the broadcast operand), each recomputing P from the saved log-sum-exp.
#include "common.h"

namespace {

constexpr int KT = 32;  // keys (or queries, in the dK/dV kernel) per LDS tile

template <typename ET>
__device__ __forceinline__ void load_row(const ET* p, float* o, int n);
template <>
__device__ __forceinline__ void load_row<float>(const float* p, float* o, int n) {
  for (int i = 0; i < n; i += 4) {
    const float4 t = *reinterpret_cast<const float4*>(p + i);
    o[i] = t.x; o[i + 1] = t.y; o[i + 2] = t.z; o[i + 3] = t.w;
  }
}
template <>
__device__ __forceinline__ void load_row<bf16_t>(const bf16_t* p, float* o, int n) {
  for (int i = 0; i < n; i += 4) {
    const uint2 t = *reinterpret_cast<const uint2*>(p + i);
    o[i] = __uint_as_float(t.x << 16); o[i + 1] = __uint_as_float(t.x & 0xffff0000u);
    o[i + 2] = __uint_as_float(t.y << 16); o[i + 3] = __uint_as_float(t.y & 0xffff0000u);
  }
}
template <typename ET>
__device__ __forceinline__ void store_row(ET* p, const float* v, int n);
template <>
__device__ __forceinline__ void store_row<float>(float* p, const float* v, int n) {
  for (int i = 0; i < n; i += 4) *reinterpret_cast<float4*>(p + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
}
template <>
__device__ __forceinline__ void store_row<bf16_t>(bf16_t* p, const float* v, int n) {
  for (int i = 0; i < n; i += 4) {
    uint2 t;
    t.x = (uint32_t)f2bf(v[i]) | ((uint32_t)f2bf(v[i + 1]) << 16);
    t.y = (uint32_t)f2bf(v[i + 2]) | ((uint32_t)f2bf(v[i + 3]) << 16);
    *reinterpret_cast<uint2*>(p + i) = t;
  }
}

// stage rows [r0, r0+KT) of a [*, ld] matrix (head slice of DH columns) into s[KT][DH] as f32; rows >= R are zeroed
template <typename ET, int DH>
__device__ __forceinline__ void stage_tile(const ET* __restrict__ base, int ld, int r0, int R, float (*s)[DH]) {
  constexpr int PER_ROW = DH / 4;
  for (int i = threadIdx.x; i < KT * PER_ROW; i += WAVE) {
    const int r = i / PER_ROW, c = (i % PER_ROW) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r0 + r < R) load_row<ET>(base + (size_t)(r0 + r) * ld + c, v, 4);
    *reinterpret_cast<float4*>(&s[r][c]) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

constexpr int CH = 32;  // channels held by one lane; a row of DH channels is spread over LPR = DH/CH adjacent lanes

// <reg[0..CH), s_row[c0..c0+CH)> summed over the LPR lanes of the row
template <int LPR>
__device__ __forceinline__ float dot_lds(const float* reg, const float* s_row) {
  float acc = 0.f;
#pragma unroll
  for (int d = 0; d < CH; d += 4) {
    const float4 t = *reinterpret_cast<const float4*>(s_row + d);
    acc = fmaf(reg[d], t.x, acc); acc = fmaf(reg[d + 1], t.y, acc);
    acc = fmaf(reg[d + 2], t.z, acc); acc = fmaf(reg[d + 3], t.w, acc);
  }
  if (LPR == 2) acc += __shfl_xor(acc, 1, WAVE);
  return acc;
}
__device__ __forceinline__ void axpy_lds(float a, const float* s_row, float* acc) {
#pragma unroll
  for (int d = 0; d < CH; d += 4) {
    const float4 t = *reinterpret_cast<const float4*>(s_row + d);
    acc[d] = fmaf(a, t.x, acc[d]); acc[d + 1] = fmaf(a, t.y, acc[d + 1]);
    acc[d + 2] = fmaf(a, t.z, acc[d + 2]); acc[d + 3] = fmaf(a, t.w, acc[d + 3]);
  }
}

template <typename ET, int DH>
__global__ __launch_bounds__(WAVE) void selfattn_fwd_kernel(const ET* __restrict__ q, const ET* __restrict__ k,
                                                             const ET* __restrict__ v, const uint32_t* __restrict__ mbits,
                                                             ET* __restrict__ o, float* __restrict__ lse, int Q, int nh, int ldq,
                                                             int ldk, int ldv, int mwords) {
  constexpr int LPR = DH / CH, RPW = WAVE / LPR;  // lanes per row, rows per wave
  __shared__ float sK[KT][DH];
  __shared__ float sV[KT][DH];
  const int h = blockIdx.y, b = blockIdx.z;
  const int qi = blockIdx.x * RPW + threadIdx.x / LPR;
  const int c0 = (threadIdx.x % LPR) * CH;
  const bool live = qi < Q;
  const float scale = rsqrtf((float)DH);
  float qr[CH], acc[CH];
  if (live) load_row<ET>(q + ((size_t)b * Q + qi) * ldq + h * DH + c0, qr, CH);
#pragma unroll
  for (int d = 0; d < CH; ++d) { qr[d] = live ? qr[d] * scale : 0.f; acc[d] = 0.f; }
  float m = -INFINITY, l = 0.f;
  const ET* kb = k + (size_t)b * Q * ldk + h * DH;
  const ET* vb = v + (size_t)b * Q * ldv + h * DH;
  for (int j0 = 0; j0 < Q; j0 += KT) {
    __syncthreads();
    stage_tile<ET, DH>(kb, ldk, j0, Q, sK);
    stage_tile<ET, DH>(vb, ldv, j0, Q, sV);
    __syncthreads();
    uint32_t blocked = (mbits && live) ? mbits[(size_t)qi * mwords + j0 / KT] : 0u;
    const int jn = min(KT, Q - j0);
    if (jn < KT) blocked |= ~0u << jn;
    float s[KT];
    float tm = -INFINITY;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      s[j] = dot_lds<LPR>(qr, &sK[j][c0]);
      if ((blocked >> j) & 1u) s[j] = -INFINITY;
      tm = fmaxf(tm, s[j]);
    }
    const float mn = fmaxf(m, tm);
    const float msafe = (mn == -INFINITY) ? 0.f : mn;
    const float corr = __expf(m - msafe);  // m = -inf -> 0
    l *= corr;
#pragma unroll
    for (int d = 0; d < CH; ++d) acc[d] *= corr;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      const float p = __expf(s[j] - msafe);  // masked -> exp(-inf) = 0
      l += p;
      axpy_lds(p, &sV[j][c0], acc);
    }
    m = mn;
  }
  if (live) {
    const float inv = 1.f / l;  // a fully masked row gives NaN exactly like the reference softmax
#pragma unroll
    for (int d = 0; d < CH; ++d) acc[d] *= inv;
    store_row<ET>(o + ((size_t)b * Q + qi) * (nh * DH) + h * DH + c0, acc, CH);
    if (c0 == 0) lse[((size_t)b * nh + h) * Q + qi] = m + __logf(l);
  }
}

// dQ: (part of) a lane per query row; also writes delta_i = <dO_i, O_i> for the dK/dV kernel
template <typename ET, int DH>
__global__ __launch_bounds__(WAVE) void selfattn_bwd_dq_kernel(const ET* __restrict__ go, const ET* __restrict__ q,
                                                                const ET* __restrict__ k, const ET* __restrict__ v,
                                                                const ET* __restrict__ o, const float* __restrict__ lse,
                                                                const uint32_t* __restrict__ mbits, ET* __restrict__ gq,
                                                                float* __restrict__ delta, int Q, int nh, int ldq, int ldk, int ldv,
                                                                int mwords) {
  constexpr int LPR = DH / CH, RPW = WAVE / LPR;
  __shared__ float sK[KT][DH];
  __shared__ float sV[KT][DH];
  const int h = blockIdx.y, b = blockIdx.z;
  const int qi = blockIdx.x * RPW + threadIdx.x / LPR;
  const int c0 = (threadIdx.x % LPR) * CH;
  const bool live = qi < Q;
  const float scale = rsqrtf((float)DH);
  float qr[CH], gor[CH], acc[CH];
  float dl = 0.f, ls = 0.f;
  if (live) {
    const size_t ro = ((size_t)b * Q + qi) * (nh * DH) + h * DH + c0;
    load_row<ET>(q + ((size_t)b * Q + qi) * ldq + h * DH + c0, qr, CH);
    load_row<ET>(go + ro, gor, CH);
    load_row<ET>(o + ro, acc, CH);
#pragma unroll
    for (int d = 0; d < CH; ++d) dl = fmaf(gor[d], acc[d], dl);
    ls = lse[((size_t)b * nh + h) * Q + qi];
  }
  if (LPR == 2) dl += __shfl_xor(dl, 1, WAVE);
  if (live && c0 == 0) delta[((size_t)b * nh + h) * Q + qi] = dl;
#pragma unroll
  for (int d = 0; d < CH; ++d) { if (!live) { qr[d] = 0.f; gor[d] = 0.f; } acc[d] = 0.f; }
  const ET* kb = k + (size_t)b * Q * ldk + h * DH;
  const ET* vb = v + (size_t)b * Q * ldv + h * DH;
  for (int j0 = 0; j0 < Q; j0 += KT) {
    __syncthreads();
    stage_tile<ET, DH>(kb, ldk, j0, Q, sK);
    stage_tile<ET, DH>(vb, ldv, j0, Q, sV);
    __syncthreads();
    uint32_t blocked = (mbits && live) ? mbits[(size_t)qi * mwords + j0 / KT] : 0u;
    const int jn = min(KT, Q - j0);
    if (jn < KT) blocked |= ~0u << jn;
#pragma unroll 4
    for (int j = 0; j < KT; ++j) {
      const float s = dot_lds<LPR>(qr, &sK[j][c0]) * scale;
      const float p = ((blocked >> j) & 1u) ? 0.f : __expf(s - ls);
      const float dp = dot_lds<LPR>(gor, &sV[j][c0]);
      axpy_lds(p * (dp - dl) * scale, &sK[j][c0], acc);
    }
  }
  if (live) store_row<ET>(gq + ((size_t)b * Q + qi) * (nh * DH) + h * DH + c0, acc, CH);
}

// dK, dV: (part of) a lane per key row; queries stream through LDS
template <typename ET, int DH>
__global__ __launch_bounds__(WAVE) void selfattn_bwd_dkv_kernel(const ET* __restrict__ go, const ET* __restrict__ q,
                                                                 const ET* __restrict__ k, const ET* __restrict__ v,
                                                                 const float* __restrict__ lse, const float* __restrict__ delta,
                                                                 const uint32_t* __restrict__ mbits, ET* __restrict__ gk,
                                                                 ET* __restrict__ gv, int Q, int nh, int ldq, int ldk, int ldv,
                                                                 int mwords) {
  constexpr int LPR = DH / CH, RPW = WAVE / LPR;
  __shared__ float sQ[KT][DH];
  __shared__ float sG[KT][DH];
  __shared__ float sL[KT], sD[KT];
  __shared__ uint32_t sM[KT][2];
  const int h = blockIdx.y, b = blockIdx.z;
  const int kloc = threadIdx.x / LPR;            // key index inside this block's span of RPW keys
  const int kj = blockIdx.x * RPW + kloc;
  const int c0 = (threadIdx.x % LPR) * CH;
  const bool live = kj < Q;
  const float scale = rsqrtf((float)DH);
  float kr[CH], vr[CH], dk[CH], dv[CH];
  if (live) {
    load_row<ET>(k + ((size_t)b * Q + kj) * ldk + h * DH + c0, kr, CH);
    load_row<ET>(v + ((size_t)b * Q + kj) * ldv + h * DH + c0, vr, CH);
  }
#pragma unroll
  for (int d = 0; d < CH; ++d) { if (!live) { kr[d] = 0.f; vr[d] = 0.f; } dk[d] = 0.f; dv[d] = 0.f; }
  const ET* qb = q + (size_t)b * Q * ldq + h * DH;
  const ET* gb = go + (size_t)b * Q * (nh * DH) + h * DH;
  const int w0 = (blockIdx.x * RPW) / 32;  // first mask word covering this block's keys (RPW = 32 or 64: word aligned)
  for (int i0 = 0; i0 < Q; i0 += KT) {
    __syncthreads();
    stage_tile<ET, DH>(qb, ldq, i0, Q, sQ);
    stage_tile<ET, DH>(gb, nh * DH, i0, Q, sG);
    if (threadIdx.x < KT) {
      const int i = i0 + threadIdx.x;
      const bool ok = i < Q;
      sL[threadIdx.x] = ok ? lse[((size_t)b * nh + h) * Q + i] : INFINITY;  // +inf -> p = 0 for padded queries
      sD[threadIdx.x] = ok ? delta[((size_t)b * nh + h) * Q + i] : 0.f;
      sM[threadIdx.x][0] = (mbits && ok) ? mbits[(size_t)i * mwords + w0] : 0u;
      sM[threadIdx.x][1] = (mbits && ok && w0 + 1 < mwords) ? mbits[(size_t)i * mwords + w0 + 1] : 0u;
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < KT; ++i) {
      const float s = dot_lds<LPR>(kr, &sQ[i][c0]) * scale;
      const bool blk = (sM[i][kloc >> 5] >> (kloc & 31)) & 1u;
      const float p = blk ? 0.f : __expf(s - sL[i]);
      const float dp = dot_lds<LPR>(vr, &sG[i][c0]);
      axpy_lds(p * (dp - sD[i]) * scale, &sQ[i][c0], dk);
      axpy_lds(p, &sG[i][c0], dv);
    }
  }
  if (live) {
    const size_t ro = ((size_t)b * Q + kj) * (nh * DH) + h * DH + c0;
    store_row<ET>(gk + ro, dk, CH);
    store_row<ET>(gv + ro, dv, CH);
  }
}

inline bool sa_ok(int B, int Q, int nh, int dh) { return B > 0 && Q > 0 && nh > 0 && B <= 65535 && nh <= 65535 && dh > 0; }

}  // namespace

extern "C" int tamtr_selfattn_fwd(const void* q, const void* k, const void* v, const uint32_t* mask_bits, void* o, float* lse,
                                  int B, int Q, int nh, int dh, int ldq, int ldk, int ldv, int dtype, void* stream) {
  if (!q || !k || !v || !o || !lse || !sa_ok(B, Q, nh, dh)) return TAMTR_EINVAL;
  if ((dh != 32 && dh != 64) || Q > 4096 || (ldq | ldk | ldv) % 4) return TAMTR_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
  const int rpw = WAVE / (dh / 32);
  dim3 grid((Q + rpw - 1) / rpw, nh, B);
  const int mw = (Q + 31) / 32;
#define GO(ET, DH) \
  hipLaunchKernelGGL((selfattn_fwd_kernel<ET, DH>), grid, dim3(WAVE), 0, s, (const ET*)q, (const ET*)k, (const ET*)v, mask_bits, \
                     (ET*)o, lse, Q, nh, ldq, ldk, ldv, mw)
  if (dtype == TAMTR_F32) { if (dh == 32) GO(float, 32); else GO(float, 64); }
  else if (dtype == TAMTR_BF16) { if (dh == 32) GO(bf16_t, 32); else GO(bf16_t, 64); }
  else return TAMTR_EINVAL;
#undef GO
  return tamtr_launch_status();
}

extern "C" int tamtr_selfattn_bwd(const void* go, const void* q, const void* k, const void* v, const void* o, const float* lse,
                                  const uint32_t* mask_bits, void* gq, void* gk, void* gv, float* delta_ws, int B, int Q, int nh,
                                  int dh, int ldq, int ldk, int ldv, int dtype, void* stream) {
  if (!go || !q || !k || !v || !o || !lse || !gq || !gk || !gv || !delta_ws || !sa_ok(B, Q, nh, dh)) return TAMTR_EINVAL;
  if ((dh != 32 && dh != 64) || Q > 4096 || (ldq | ldk | ldv) % 4) return TAMTR_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
  const int rpw = WAVE / (dh / 32);
  dim3 grid((Q + rpw - 1) / rpw, nh, B);
  const int mw = (Q + 31) / 32;
#define GO(ET, DH)                                                                                                        \
  {                                                                                                                       \
    hipLaunchKernelGGL((selfattn_bwd_dq_kernel<ET, DH>), grid, dim3(WAVE), 0, s, (const ET*)go, (const ET*)q, (const ET*)k, \
                       (const ET*)v, (const ET*)o, lse, mask_bits, (ET*)gq, delta_ws, Q, nh, ldq, ldk, ldv, mw);          \
    hipLaunchKernelGGL((selfattn_bwd_dkv_kernel<ET, DH>), grid, dim3(WAVE), 0, s, (const ET*)go, (const ET*)q, (const ET*)k, \
                       (const ET*)v, lse, delta_ws, mask_bits, (ET*)gk, (ET*)gv, Q, nh, ldq, ldk, ldv, mw);               \
  }
  if (dtype == TAMTR_F32) { if (dh == 32) GO(float, 32) else GO(float, 64) }
  else if (dtype == TAMTR_BF16) { if (dh == 32) GO(bf16_t, 32) else GO(bf16_t, 64) }
  else return TAMTR_EINVAL;
#undef GO
  return tamtr_launch_status();
}
