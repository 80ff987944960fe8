// contrastive.hip - fused L2-normalise + text-contrastive logits (MEH classification head), forward + backward, gfx950.
//
// Replaces ContrastiveHeadMLP.forward (reference: ultralytics/nn/modules/block.py:534-541): permute, two F.normalize
// passes, einsum, scale+bias, permute.  HBM-bound (reads x once, writes K logits per row):
//   * the text tile w[b] ([K, C] f32, 20 KB at K=10, C=512) is staged once per workgroup in LDS together with its
//     reciprocal row norms; a workgroup serves ROWS query rows of the SAME image so the tile is reused;
//   * one wave per query row: lanes stride the channel axis with 16-B loads, keep their slice of x in registers,
//     and the |x|^2 and K dot-product reductions are xor-shuffle butterflies (no LDS round trip, no atomics).
#include "common.h"

namespace {

constexpr int CT_THREADS = 256;
constexpr int CT_WAVES = CT_THREADS / WAVE;
constexpr int CT_ROWS = 16;  // query rows per workgroup (forward)
constexpr int CT_BROWS = 80; // backward: few workgroups per image: each stores its d(what) partial, the caller adds the few partials
constexpr int CT_KR = 16;    // text rows whose d(what) partial sums a wave keeps in registers (larger vocabularies: LDS atomics)

template <typename ET>
struct CV;  // per-lane vector width (elements per 16 bytes)
template <>
struct CV<float> { static constexpr int V = 4; };
template <>
struct CV<bf16_t> { static constexpr int V = 8; };

template <typename ET>
__device__ __forceinline__ void ldv(const ET* p, float* o);
template <>
__device__ __forceinline__ void ldv<float>(const float* p, float* o) {
  float4 t = *reinterpret_cast<const float4*>(p);
  o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
}
template <>
__device__ __forceinline__ void ldv<bf16_t>(const bf16_t* p, float* o) {
  uint4 t = *reinterpret_cast<const uint4*>(p);
  const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
template <typename ET>
__device__ __forceinline__ void stv(ET* p, const float* v);
template <>
__device__ __forceinline__ void stv<float>(float* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <>
__device__ __forceinline__ void stv<bf16_t>(bf16_t* p, const float* v) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f2bf(v[2 * i]) | ((uint32_t)f2bf(v[2 * i + 1]) << 16);
  *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}

// stage w[b] -> s_w[K][C], reciprocal norms -> s_winv[K]
__device__ __forceinline__ void stage_text(const float* __restrict__ wb, float* s_w, float* s_winv, int K, int C) {
  for (int i = threadIdx.x * 4; i < K * C; i += CT_THREADS * 4)
    *reinterpret_cast<float4*>(s_w + i) = *reinterpret_cast<const float4*>(wb + i);
  __syncthreads();
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  for (int k = wave; k < K; k += CT_WAVES) {
    float ss = 0.f;
    for (int c = lane; c < C; c += WAVE) ss = fmaf(s_w[k * C + c], s_w[k * C + c], ss);
    ss = group_sum<WAVE>(ss);
    if (lane == 0) s_winv[k] = 1.f / fmaxf(sqrtf(ss), 1e-12f);
  }
  __syncthreads();
}

// NP = passes over the channel axis per lane (C <= NP * 64 * V)
template <typename ET, int NP>
__global__ __launch_bounds__(CT_THREADS) void contrastive_fwd_kernel(const ET* __restrict__ x, const float* __restrict__ w,
                                                                      const float* __restrict__ logit_scale,
                                                                      const float* __restrict__ bias, float* __restrict__ logits,
                                                                      float* __restrict__ xinv, float* __restrict__ winv, int Q,
                                                                      int K, int C) {
  constexpr int V = CV<ET>::V;
  extern __shared__ float smem[];
  float* s_w = smem;
  float* s_winv = smem + (size_t)K * C;
  const int b = blockIdx.y;
  stage_text(w + (size_t)b * K * C, s_w, s_winv, K, C);
  if (blockIdx.x == 0)
    for (int k = threadIdx.x; k < K; k += CT_THREADS) winv[(size_t)b * K + k] = s_winv[k];
  const float sc = __expf(*logit_scale), bi = *bias;
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  for (int r = wave; r < CT_ROWS; r += CT_WAVES) {
    const int q = blockIdx.x * CT_ROWS + r;
    if (q >= Q) break;  // wave-uniform
    const ET* xr = x + ((size_t)b * Q + q) * C;
    float xv[NP][V];
    float ss = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int c = (p * WAVE + lane) * V;
      if (c < C) ldv<ET>(xr + c, xv[p]);
      else
#pragma unroll
        for (int i = 0; i < V; ++i) xv[p][i] = 0.f;
#pragma unroll
      for (int i = 0; i < V; ++i) ss = fmaf(xv[p][i], xv[p][i], ss);
    }
    ss = group_sum<WAVE>(ss);
    const float xi = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    if (lane == 0) xinv[(size_t)b * Q + q] = xi;
    for (int k0 = 0; k0 < K; k0 += WAVE) {
      float mine = 0.f;
      const int kn = min(WAVE, K - k0);
      for (int kk = 0; kk < kn; ++kk) {
        const float* wr = s_w + (size_t)(k0 + kk) * C;
        float d = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int c = (p * WAVE + lane) * V;
          if (c < C) {
#pragma unroll
            for (int i = 0; i < V; i += 4) {
              const float4 t = *reinterpret_cast<const float4*>(wr + c + i);
              d = fmaf(xv[p][i], t.x, d); d = fmaf(xv[p][i + 1], t.y, d);
              d = fmaf(xv[p][i + 2], t.z, d); d = fmaf(xv[p][i + 3], t.w, d);
            }
          }
        }
        d = group_sum<WAVE>(d);
        if (lane == kk) mine = d * xi * s_winv[k0 + kk] * sc + bi;
      }
      if (lane < kn) logits[((size_t)b * Q + q) * K + k0 + lane] = mine;
    }
  }
}

template <typename ET, int NP, bool REGS>
__global__ __launch_bounds__(CT_THREADS) void contrastive_bwd_kernel(const float* __restrict__ g, const ET* __restrict__ x,
                                                                      const float* __restrict__ w,
                                                                      const float* __restrict__ logit_scale,
                                                                      const float* __restrict__ xinv, ET* __restrict__ dx,
                                                                      float* __restrict__ dwhat, int Q, int K, int C) {
  constexpr int V = CV<ET>::V;
  extern __shared__ float smem[];
  float* s_w = smem;                      // [K][C] raw text rows
  float* s_acc = smem + (size_t)K * C;    // [K][C] sum_q s*g*xhat (this workgroup's rows)
  float* s_winv = s_acc + (size_t)K * C;  // [K]
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < K * C; i += CT_THREADS) s_acc[i] = 0.f;
  stage_text(w + (size_t)b * K * C, s_w, s_winv, K, C);
  const float sc = __expf(*logit_scale);
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  float wacc[REGS ? CT_KR : 1][NP][V];   // this wave's sum over its rows of s g xhat, per text row (K <= CT_KR)
#pragma unroll
  for (int k = 0; k < (REGS ? CT_KR : 1); ++k)
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int i = 0; i < V; ++i) wacc[k][p][i] = 0.f;
  for (int r = wave; r < CT_BROWS; r += CT_WAVES) {
    const int q = blockIdx.x * CT_BROWS + r;
    if (q >= Q) break;
    const size_t row = (size_t)b * Q + q;
    const float xi = xinv[row];
    float xh[NP][V], dh[NP][V];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int c = (p * WAVE + lane) * V;
      if (c < C) ldv<ET>(x + row * C + c, xh[p]);
#pragma unroll
      for (int i = 0; i < V; ++i) { xh[p][i] = (c < C) ? xh[p][i] * xi : 0.f; dh[p][i] = 0.f; }
    }
    if (REGS) {
#pragma unroll
      for (int k = 0; k < CT_KR; ++k) {
        if (k < K) {
          const float gs = g[row * K + k] * sc;  // wave-uniform
          const float wi = s_winv[k];
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            const int c = (p * WAVE + lane) * V;
            if (c < C) {
#pragma unroll
              for (int i = 0; i < V; ++i) {
                dh[p][i] = fmaf(gs * wi, s_w[(size_t)k * C + c + i], dh[p][i]);   // d xhat += s g what
                wacc[k][p][i] = fmaf(gs, xh[p][i], wacc[k][p][i]);                 // d what += s g xhat
              }
            }
          }
        }
      }
    } else {
      for (int k = 0; k < K; ++k) {
        const float gs = g[row * K + k] * sc;  // wave-uniform
        const float wi = s_winv[k];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int c = (p * WAVE + lane) * V;
          if (c < C) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
              dh[p][i] = fmaf(gs * wi, s_w[(size_t)k * C + c + i], dh[p][i]);   // d xhat += s g what
              atomicAdd(&s_acc[(size_t)k * C + c + i], gs * xh[p][i]);           // d what += s g xhat
            }
          }
        }
      }
    }
    float dot = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int i = 0; i < V; ++i) dot = fmaf(dh[p][i], xh[p][i], dot);
    dot = group_sum<WAVE>(dot);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int c = (p * WAVE + lane) * V;
      if (c < C) {
        float o[V];
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = xi * (dh[p][i] - dot * xh[p][i]);
        stv<ET>(dx + row * C + c, o);
      }
    }
  }
  if (REGS) {   // the four waves' register sums meet in LDS once, one wave after the other: a fixed order of additions
    for (int turn = 0; turn < CT_WAVES; ++turn) {
      __syncthreads();
      if (wave == turn) {
#pragma unroll
        for (int k = 0; k < CT_KR; ++k)
          if (k < K)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
              const int c = (p * WAVE + lane) * V;
              if (c < C)
#pragma unroll
                for (int i = 0; i < V; ++i) s_acc[(size_t)k * C + c + i] += wacc[k][p][i];
            }
      }
    }
  }
  __syncthreads();
  // this workgroup's partial of d(what): plain stores into its own slab [b][blockIdx.x][K][C] (no global atomics; the caller sums the
  // ceil(Q / CT_BROWS) slabs)
  float* slab = dwhat + ((size_t)b * gridDim.x + blockIdx.x) * K * C;
  for (int i = threadIdx.x; i < K * C; i += CT_THREADS) slab[i] = s_acc[i];
}

inline int pick_np(int C, int V) {
  const int per = WAVE * V;
  const int n = (C + per - 1) / per;
  return n <= 1 ? 1 : n <= 2 ? 2 : n <= 4 ? 4 : n <= 8 ? 8 : 0;
}

}  // namespace

extern "C" int tamtr_contrastive_logits_fwd(const void* x, const float* w, const float* logit_scale, const float* bias,
                                            float* logits, float* xinv, float* winv, int B, int Q, int K, int C, int dtype,
                                            void* stream) {
  if (!x || !w || !logit_scale || !bias || !logits || !xinv || !winv || B <= 0 || Q <= 0 || K <= 0 || C <= 0)
    return TAMTR_EINVAL;
  if (dtype != TAMTR_F32 && dtype != TAMTR_BF16) return TAMTR_EINVAL;
  const int V = dtype == TAMTR_F32 ? 4 : 8;
  const int np = pick_np(C, V);
  const size_t lds = ((size_t)K * C + K) * sizeof(float);
  if (C % 8 || np == 0 || lds > 96 * 1024 || K > 128 || B > 65535) return TAMTR_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((Q + CT_ROWS - 1) / CT_ROWS, B);
#define GO(ET, NP) \
  hipLaunchKernelGGL((contrastive_fwd_kernel<ET, NP>), grid, dim3(CT_THREADS), lds, s, (const ET*)x, w, logit_scale, bias, \
                     logits, xinv, winv, Q, K, C)
#define PICK(ET) \
  switch (np) { case 1: GO(ET, 1); break; case 2: GO(ET, 2); break; case 4: GO(ET, 4); break; default: GO(ET, 8); }
  if (dtype == TAMTR_F32) { PICK(float) } else { PICK(bf16_t) }
#undef PICK
#undef GO
#undef GO1
  return tamtr_launch_status();
}

extern "C" int tamtr_contrastive_bwd_slabs(int Q) { return Q > 0 ? (Q + CT_BROWS - 1) / CT_BROWS : 0; }

extern "C" int tamtr_contrastive_logits_bwd(const float* g, const void* x, const float* w, const float* logit_scale,
                                            const float* xinv, const float* winv, void* dx, float* dwhat, int B, int Q, int K,
                                            int C, int dtype, void* stream) {
  (void)winv;
  if (!g || !x || !w || !logit_scale || !xinv || !dx || !dwhat || B <= 0 || Q <= 0 || K <= 0 || C <= 0) return TAMTR_EINVAL;
  if (dtype != TAMTR_F32 && dtype != TAMTR_BF16) return TAMTR_EINVAL;
  const int V = dtype == TAMTR_F32 ? 4 : 8;
  const int np = pick_np(C, V);
  const size_t lds = ((size_t)2 * K * C + K) * sizeof(float);
  if (C % 8 || np == 0 || lds > 128 * 1024 || K > 128 || B > 65535) return TAMTR_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((Q + CT_BROWS - 1) / CT_BROWS, B);
#define GO1(ET, NP, REGS)                                                                                       \
  {                                                                                                             \
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)contrastive_bwd_kernel<ET, NP, REGS>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               (int)lds) != hipSuccess)                                        \
      return TAMTR_ELAUNCH;                                                                                     \
    hipLaunchKernelGGL((contrastive_bwd_kernel<ET, NP, REGS>), grid, dim3(CT_THREADS), lds, s, g, (const ET*)x, w, logit_scale, xinv, \
                       (ET*)dx, dwhat, Q, K, C);                                                                \
  }
#define GO(ET, NP) { if (K <= CT_KR) GO1(ET, NP, true) else GO1(ET, NP, false) }   /* register sums: up to 2 column groups per lane */
#define PICK(ET) \
  switch (np) { case 1: GO(ET, 1) break; case 2: GO(ET, 2) break; case 4: GO1(ET, 4, false) break; default: GO1(ET, 8, false) }
  if (dtype == TAMTR_F32) { PICK(float) } else { PICK(bf16_t) }
#undef PICK
#undef GO
#undef GO1
  return tamtr_launch_status();
}
