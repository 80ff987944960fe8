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
#include <cstdlib>

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

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// MFMA path: bf16 operands, dh = 64 (the MEH decoder: 8 heads of 64).  One wave owns 32 query rows (forward, dQ) or 32 key rows
// (dK/dV) of one (image, head) and walks the other axis in tiles of 32; QK^T and PV run on v_mfma_f32_32x32x16_bf16.
//
// Operand layouts of that instruction (lane l, lr = l % 32, lh = l / 32): A[m = lr][k = 8 lh + 0..7], B[k = 8 lh + 0..7][n = lr],
// C[m = 8 j + 4 lh + i][n = lr] in register 4 j + i.  The orientation is chosen so that the softmax axis never crosses lanes:
//   forward / dQ:  S^T = K Q^T  (C: row = key, lane = query)  -> row max / sum are in-lane over 16 registers + one exchange with
//                  the other half-wave; P^T is then the B operand of O^T = V^T P^T as it stands: the contraction slot (lh, t) of
//                  K-step s2 is DEFINED as key 16 s2 + 8 (t / 4) + 4 lh + t % 4, i.e. registers 8 s2 .. 8 s2 + 7 of the lane, and the
//                  A operand (V^T, gathered from an LDS copy of the V tile) uses the same slot order;
//   dK/dV:         S = Q K^T    (C: row = query, lane = key) -> P and dS are the B operands of dV^T = dO^T P and dK^T = Q^T dS.
// K / Q / dO / V fragments whose k index is the head channel are 16-byte row reads straight from global memory (L2 resident:
// 75 KB per (image, head)); only the "transposed" operands go through LDS.  Mask: bit-packed words; LSE as in the scalar path.
namespace {

constexpr int MT = 32;          // rows per wave and per tile
constexpr int MP = 64 + 8;      // LDS row pitch (bf16 elements): 16-byte aligned rows

__device__ __forceinline__ s16x8 row_frag(const bf16_t* row, int s, int lh) {   // row[16 s + 8 lh .. + 7]
  return *reinterpret_cast<const s16x8*>(row + 16 * s + 8 * lh);
}
// A operand with m = channel (32 dt + lr), k = the slot order above, from a row-major LDS tile t[row][channel]
__device__ __forceinline__ s16x8 slot_frag(const bf16_t (*t)[MP], int s2, int lh, int col) {
  s16x8 f;
#pragma unroll
  for (int u = 0; u < 8; ++u) f[u] = (short)t[16 * s2 + 8 * (u / 4) + 4 * lh + (u % 4)][col];
  return f;
}
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ s16x8 pack_frag(const float* p) {   // 8 consecutive C registers -> one B operand
  const uint32_t w[4] = {pk_bf16(p[0], p[1]), pk_bf16(p[2], p[3]), pk_bf16(p[4], p[5]), pk_bf16(p[6], p[7])};
  s16x8 f;
  __builtin_memcpy(&f, w, 16);
  return f;
}
// stage rows [r0, r0 + 32) (clamped to R - 1) x 64 channels of a [*, ld] matrix into t: lane = (row l / 2, half l % 2), 4 x 16 B
__device__ __forceinline__ void stage32(const bf16_t* __restrict__ base, int ld, int r0, int R, bf16_t (*t)[MP]) {
  const int r = threadIdx.x / 2, c = (threadIdx.x % 2) * 32;
  const bf16_t* src = base + (size_t)min(r0 + r, R - 1) * ld + c;
#pragma unroll
  for (int u = 0; u < 4; ++u) *reinterpret_cast<uint4*>(&t[r][c + 8 * u]) = *reinterpret_cast<const uint4*>(src + 8 * u);
}
// C tile (m = channel 8 j + 4 lh + i of d-tile dt, n = row lr) -> out[row][32 dt + ...] as bf16
__device__ __forceinline__ void store_ct(bf16_t* __restrict__ row, const f32x16 (&acc)[2], int lh, float mul) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint2 w;
      w.x = pk_bf16(acc[dt][4 * j] * mul, acc[dt][4 * j + 1] * mul);
      w.y = pk_bf16(acc[dt][4 * j + 2] * mul, acc[dt][4 * j + 3] * mul);
      *reinterpret_cast<uint2*>(row + 32 * dt + 8 * j + 4 * lh) = w;
    }
}

__global__ __launch_bounds__(WAVE) void selfattn_mfma_fwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                                  const bf16_t* __restrict__ v, const uint32_t* __restrict__ mbits,
                                                                  bf16_t* __restrict__ o, float* __restrict__ lse, int Q, int nh, int ldq,
                                                                  int ldk, int ldv, int mwords) {
  __shared__ bf16_t sV[MT][MP];
  const int h = blockIdx.y, b = blockIdx.z, lr = threadIdx.x % 32, lh = threadIdx.x / 32;
  const int qi = blockIdx.x * MT + lr, qc = min(qi, Q - 1);
  const float scale = 0.125f;  // 1 / sqrt(64)
  const bf16_t* qrow = q + ((size_t)b * Q + qc) * ldq + h * 64;
  const bf16_t* kb = k + (size_t)b * Q * ldk + h * 64;
  const bf16_t* vb = v + (size_t)b * Q * ldv + h * 64;
  s16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = row_frag(qrow, s, lh);
  f32x16 acc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
  float m = -INFINITY, l = 0.f;
  const int nt = (Q + MT - 1) / MT;
  for (int t = 0; t < nt; ++t) {
    const int j0 = t * MT;
    const bf16_t* krow = kb + (size_t)min(j0 + lr, Q - 1) * ldk;
    s16x8 kf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) kf[s] = row_frag(krow, s, lh);
    __syncthreads();
    stage32(vb, ldv, j0, Q, sV);
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s], qf[s], st, 0, 0, 0);
    uint32_t blocked = mbits ? mbits[(size_t)qc * mwords + t] : 0u;
    if (Q - j0 < MT) blocked |= ~0u << (Q - j0);
    float sv[16], tm = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 8 * (r / 4) + 4 * lh + (r % 4);
      sv[r] = ((blocked >> key) & 1u) ? -INFINITY : st[r] * scale;
      tm = fmaxf(tm, sv[r]);
    }
    tm = fmaxf(tm, __shfl_xor(tm, 32, WAVE));
    const float mn = fmaxf(m, tm);
    const float msafe = (mn == -INFINITY) ? 0.f : mn;
    const float corr = __expf(m - msafe);
    l *= corr;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] *= corr; acc[1][r] *= corr; }
#pragma unroll
    for (int r = 0; r < 16; ++r) { sv[r] = __expf(sv[r] - msafe); l += sv[r]; }
    const s16x8 pf[2] = {pack_frag(sv), pack_frag(sv + 8)};
    __syncthreads();
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(slot_frag(sV, s2, lh, 32 * dt + lr), pf[s2], acc[dt], 0, 0, 0);
    m = mn;
  }
  l += __shfl_xor(l, 32, WAVE);
  if (qi < Q) {
    store_ct(o + ((size_t)b * Q + qi) * (nh * 64) + h * 64, acc, lh, 1.f / l);  // a fully masked row gives NaN like the reference softmax
    if (lh == 0) lse[((size_t)b * nh + h) * Q + qi] = m + __logf(l);
  }
}

// dQ (and delta_i = <dO_i, O_i>): the forward's orientation; dQ^T = K^T dS^T
__global__ __launch_bounds__(WAVE) void selfattn_mfma_dq_kernel(const bf16_t* __restrict__ go, const bf16_t* __restrict__ q,
                                                                 const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                                 const bf16_t* __restrict__ o, const float* __restrict__ lse,
                                                                 const uint32_t* __restrict__ mbits, bf16_t* __restrict__ gq,
                                                                 float* __restrict__ delta, int Q, int nh, int ldq, int ldk, int ldv,
                                                                 int mwords) {
  __shared__ bf16_t sK[MT][MP];
  const int h = blockIdx.y, b = blockIdx.z, lr = threadIdx.x % 32, lh = threadIdx.x / 32;
  const int qi = blockIdx.x * MT + lr, qc = min(qi, Q - 1);
  const float scale = 0.125f;
  const size_t orow = ((size_t)b * Q + qc) * (nh * 64) + h * 64;
  const bf16_t* qrow = q + ((size_t)b * Q + qc) * ldq + h * 64;
  const bf16_t* kb = k + (size_t)b * Q * ldk + h * 64;
  const bf16_t* vb = v + (size_t)b * Q * ldv + h * 64;
  s16x8 qf[4], gf[4];
  float dl = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qf[s] = row_frag(qrow, s, lh);
    gf[s] = row_frag(go + orow, s, lh);
    const s16x8 of = row_frag(o + orow, s, lh);
#pragma unroll
    for (int u = 0; u < 8; ++u) dl = fmaf(bf2f((bf16_t)gf[s][u]), bf2f((bf16_t)of[u]), dl);
  }
  dl += __shfl_xor(dl, 32, WAVE);
  const float ls = lse[((size_t)b * nh + h) * Q + qc];
  if (qi < Q && lh == 0) delta[((size_t)b * nh + h) * Q + qi] = dl;
  f32x16 acc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
  const int nt = (Q + MT - 1) / MT;
  for (int t = 0; t < nt; ++t) {
    const int j0 = t * MT;
    const bf16_t* krow = kb + (size_t)min(j0 + lr, Q - 1) * ldk;
    const bf16_t* vrow = vb + (size_t)min(j0 + lr, Q - 1) * ldv;
    s16x8 kf[4], vf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { kf[s] = row_frag(krow, s, lh); vf[s] = row_frag(vrow, s, lh); }
    __syncthreads();
    stage32(kb, ldk, j0, Q, sK);
    f32x16 st, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s], qf[s], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s], gf[s], dp, 0, 0, 0);
    }
    uint32_t blocked = mbits ? mbits[(size_t)qc * mwords + t] : 0u;
    if (Q - j0 < MT) blocked |= ~0u << (Q - j0);
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 8 * (r / 4) + 4 * lh + (r % 4);
      const float p = ((blocked >> key) & 1u) ? 0.f : __expf(st[r] * scale - ls);
      ds[r] = p * (dp[r] - dl) * scale;
    }
    const s16x8 df[2] = {pack_frag(ds), pack_frag(ds + 8)};
    __syncthreads();
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(slot_frag(sK, s2, lh, 32 * dt + lr), df[s2], acc[dt], 0, 0, 0);
  }
  if (qi < Q) store_ct(gq + ((size_t)b * Q + qi) * (nh * 64) + h * 64, acc, lh, 1.f);
}

// dK, dV: a wave owns 32 keys (lane = key); queries stream by in tiles of 32
__global__ __launch_bounds__(WAVE) void selfattn_mfma_dkv_kernel(const bf16_t* __restrict__ go, const bf16_t* __restrict__ q,
                                                                  const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                                  const float* __restrict__ lse, const float* __restrict__ delta,
                                                                  const uint32_t* __restrict__ mbits, bf16_t* __restrict__ gk,
                                                                  bf16_t* __restrict__ gv, int Q, int nh, int ldq, int ldk, int ldv,
                                                                  int mwords) {
  __shared__ bf16_t sQ[MT][MP];
  __shared__ bf16_t sG[MT][MP];
  __shared__ float sL[MT], sD[MT];
  __shared__ uint32_t sM[MT];
  const int h = blockIdx.y, b = blockIdx.z, lr = threadIdx.x % 32, lh = threadIdx.x / 32;
  const int kj = blockIdx.x * MT + lr, kc = min(kj, Q - 1);
  const float scale = 0.125f;
  const bf16_t* qb = q + (size_t)b * Q * ldq + h * 64;
  const bf16_t* gb = go + (size_t)b * Q * (nh * 64) + h * 64;
  const bf16_t* krow = k + ((size_t)b * Q + kc) * ldk + h * 64;
  const bf16_t* vrow = v + ((size_t)b * Q + kc) * ldv + h * 64;
  s16x8 kf[4], vf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) { kf[s] = row_frag(krow, s, lh); vf[s] = row_frag(vrow, s, lh); }
  f32x16 ak[2], av[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { ak[0][r] = 0.f; ak[1][r] = 0.f; av[0][r] = 0.f; av[1][r] = 0.f; }
  const int nt = (Q + MT - 1) / MT;
  for (int t = 0; t < nt; ++t) {
    const int i0 = t * MT;
    const bf16_t* qrow = qb + (size_t)min(i0 + lr, Q - 1) * ldq;
    const bf16_t* grow = gb + (size_t)min(i0 + lr, Q - 1) * (nh * 64);
    s16x8 qf[4], gf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { qf[s] = row_frag(qrow, s, lh); gf[s] = row_frag(grow, s, lh); }
    __syncthreads();
    stage32(qb, ldq, i0, Q, sQ);
    stage32(gb, nh * 64, i0, Q, sG);
    if (threadIdx.x < MT) {
      const int i = i0 + threadIdx.x;
      const bool ok = i < Q;
      sL[threadIdx.x] = ok ? lse[((size_t)b * nh + h) * Q + i] : INFINITY;   // +inf -> p = 0 for padded queries
      sD[threadIdx.x] = ok ? delta[((size_t)b * nh + h) * Q + i] : 0.f;
      sM[threadIdx.x] = (mbits && ok) ? mbits[(size_t)i * mwords + blockIdx.x] : 0u;   // the word of this wave's 32 keys
    }
    f32x16 st, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[s], kf[s], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf[s], vf[s], dp, 0, 0, 0);
    }
    __syncthreads();
    float pp[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qr = 8 * (r / 4) + 4 * lh + (r % 4);
      const float p = ((sM[qr] >> lr) & 1u) ? 0.f : __expf(st[r] * scale - sL[qr]);
      pp[r] = p;
      ds[r] = p * (dp[r] - sD[qr]) * scale;
    }
    const s16x8 pf[2] = {pack_frag(pp), pack_frag(pp + 8)}, df[2] = {pack_frag(ds), pack_frag(ds + 8)};
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        av[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(slot_frag(sG, s2, lh, 32 * dt + lr), pf[s2], av[dt], 0, 0, 0);
        ak[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(slot_frag(sQ, s2, lh, 32 * dt + lr), df[s2], ak[dt], 0, 0, 0);
      }
  }
  if (kj < Q) {
    const size_t ro = ((size_t)b * Q + kj) * (nh * 64) + h * 64;
    store_ct(gk + ro, ak, lh, 1.f);
    store_ct(gv + ro, av, lh, 1.f);
  }
}

inline bool mfma_ok(const void* a, const void* b, const void* c, int dh, int ldq, int ldk, int ldv, int nh, int dtype) {
  static const bool scalar_only = getenv("TAMTR_SELFATTN_SCALAR") != nullptr;   // A/B switch, read once when first used (not per call)
  return dtype == TAMTR_BF16 && dh == 64 && ((ldq | ldk | ldv | (nh * 64)) % 8) == 0 &&
         (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) % 16) == 0 && !scalar_only;
}

inline bool sa_ok(int B, int Q, int nh, int dh) { return B > 0 && Q > 0 && nh > 0 && B <= 65535 && nh <= 65535 && dh > 0; }

}  // namespace

extern "C" int tamtr_selfattn_fwd(const void* q, const void* k, const void* v, const uint32_t* mask_bits, void* o, float* lse,
                                  int B, int Q, int nh, int dh, int ldq, int ldk, int ldv, int dtype, void* stream) {
  if (!q || !k || !v || !o || !lse || !sa_ok(B, Q, nh, dh)) return TAMTR_EINVAL;
  if ((dh != 32 && dh != 64) || Q > 4096 || (ldq | ldk | ldv) % 4) return TAMTR_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
  const int mw = (Q + 31) / 32;
  if (mfma_ok(q, k, v, dh, ldq, ldk, ldv, nh, dtype) && (uintptr_t)o % 16 == 0) {
    hipLaunchKernelGGL(selfattn_mfma_fwd_kernel, dim3((Q + MT - 1) / MT, nh, B), dim3(WAVE), 0, s, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, mask_bits, (bf16_t*)o, lse, Q, nh, ldq, ldk, ldv, mw);
    return tamtr_launch_status();
  }
  const int rpw = WAVE / (dh / 32);
  dim3 grid((Q + rpw - 1) / rpw, nh, B);
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
  const int mw = (Q + 31) / 32;
  if (mfma_ok(q, k, v, dh, ldq, ldk, ldv, nh, dtype) && (((uintptr_t)go | (uintptr_t)o | (uintptr_t)gq | (uintptr_t)gk | (uintptr_t)gv) % 16) == 0) {
    const dim3 g((Q + MT - 1) / MT, nh, B);
    hipLaunchKernelGGL(selfattn_mfma_dq_kernel, g, dim3(WAVE), 0, s, (const bf16_t*)go, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                       (const bf16_t*)o, lse, mask_bits, (bf16_t*)gq, delta_ws, Q, nh, ldq, ldk, ldv, mw);
    hipLaunchKernelGGL(selfattn_mfma_dkv_kernel, g, dim3(WAVE), 0, s, (const bf16_t*)go, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                       lse, delta_ws, mask_bits, (bf16_t*)gk, (bf16_t*)gv, Q, nh, ldq, ldk, ldv, mw);
    return tamtr_launch_status();
  }
  const int rpw = WAVE / (dh / 32);
  dim3 grid((Q + rpw - 1) / rpw, nh, B);
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
