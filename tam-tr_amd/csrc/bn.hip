// bn.hip - training-mode BatchNorm2d (+ optional SiLU) on NCHW maps, forward and backward, gfx950.
//
// Reference: `self.act(self.bn(self.conv(x)))`, ultralytics/nn/modules/conv.py:36-40 (Conv), with the BatchNorm settings the
// reference rewrites after construction (eps 1e-3, momentum 0.03, utils/torch_utils.py:303-313); also the bare BatchNorms of
// RepConvN branches and of the MEH input projection.  MIOpen's spatial BatchNorm gives every channel to ONE workgroup: the
// [16, 64, 320, 320] map of the first layer ran on 64 workgroups of a 256-CU chip (954 us for 630 MB of traffic, 0.66 TB/s),
// and SiLU was a further full pass each way.  Here a channel is split over B x ceil(HW / 4096) workgroups:
//   bn_stats     : per-slice (count, mean, M2) with the slice held in registers (exact two-pass inside the slice)
//   bn_apply     : every workgroup Chan-combines the channel's slices -> batch mean / biased var -> rstd (the first slice also
//                  publishes them and updates the running stats with the unbiased var), then
//                  y = act(gamma * (x - mean) * rstd + beta), act = identity | SiLU, 16-B loads and stores
//   bn_finalize  : the same combination as its own one-workgroup-per-channel kernel (channels-last variant)
//   bn_bwd_reduce: per-slice sums of dz and dz * xhat, dz = gy * act'(z) with z recomputed from x
//   bn_bwd_apply : gx = gamma * rstd * (dz - mean(dz) - xhat * mean(dz * xhat)); the slice sums of the channel are re-added
//                  by every workgroup (<= a few hundred floats), d(gamma) / d(beta) are written by the first slice
// Traffic: 3 passes forward, 5 backward (MIOpen + SiLU: 5 and 8).
#include "common.h"

namespace {

constexpr int BN_THREADS = 256;
constexpr int BN_EPT = 16;                      // elements per thread per slice
constexpr int BN_SLICE = BN_THREADS * BN_EPT;   // 4096 elements of one (image, channel) plane per workgroup

template <typename T>
__device__ __forceinline__ void ld_slice(const T* __restrict__ p, int n, bool vec, float (&v)[BN_EPT], int (&cnt)) {
  // thread t holds elements [t*EPT, t*EPT + EPT) of the slice (n valid elements)
  const int base = threadIdx.x * BN_EPT;
  cnt = max(0, min(BN_EPT, n - base));
  if (vec && cnt == BN_EPT) {
#pragma unroll
    for (int q = 0; q < BN_EPT / 4; ++q) {
      float o[4];
      Elt<T>::ld4(p + base + 4 * q, o);
      v[4 * q] = o[0]; v[4 * q + 1] = o[1]; v[4 * q + 2] = o[2]; v[4 * q + 3] = o[3];
    }
  } else {
#pragma unroll
    for (int i = 0; i < BN_EPT; ++i) v[i] = i < cnt ? Elt<T>::ld(p + base + i) : 0.f;
  }
}

__device__ __forceinline__ float block_sum(float v, float* s_red) {
  v = group_sum<WAVE>(v);
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < BN_THREADS / WAVE; ++w) t += s_red[w];
  return t;
}

__device__ __forceinline__ float act_fwd(float z, int act) { return act ? z / (1.f + __expf(-z)) : z; }
__device__ __forceinline__ float act_bwd(float z, int act) {
  if (!act) return 1.f;
  const float s = 1.f / (1.f + __expf(-z));
  return s * (1.f + z * (1.f - s));
}

// grid: (slices_hw, C, B); part[c][b * slices_hw + s][3] = count, mean, M2
template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(const T* __restrict__ x, float* __restrict__ part, int C, int HW, int vec) {
  __shared__ float s_red[BN_THREADS / WAVE];
  const int s = blockIdx.x, c = blockIdx.y, b = blockIdx.z, S = gridDim.x * gridDim.z;
  const int n = min(BN_SLICE, HW - s * BN_SLICE);
  float v[BN_EPT];
  int cnt;
  ld_slice<T>(x + ((size_t)b * C + c) * HW + (size_t)s * BN_SLICE, n, vec, v, cnt);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < BN_EPT; ++i) sum += v[i];
  const float mean = block_sum(sum, s_red) / n;
  float m2 = 0.f;
#pragma unroll
  for (int i = 0; i < BN_EPT; ++i) { const float d = i < cnt ? v[i] - mean : 0.f; m2 = fmaf(d, d, m2); }
  m2 = block_sum(m2, s_red);
  if (threadIdx.x == 0) {
    float* o = part + ((size_t)c * S + (size_t)b * gridDim.x + s) * 3;
    o[0] = (float)n; o[1] = mean; o[2] = m2;
  }
}

// Chan-combine the S slice partials of channel c inside a workgroup (every apply workgroup redoes it: S <= a few hundred
// triples, against a 4096-element slice of real work; saves the separate one-wave-per-channel finalize launch)
__device__ __forceinline__ void combine_slices(const float* __restrict__ part, int c, int S, float* s_tri, float& mean_o, float& m2_o, float& n_o) {
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int s = threadIdx.x; s < S; s += BN_THREADS) {
    const float* p = part + ((size_t)c * S + s) * 3;
    const float nb = p[0], mb = p[1], m2b = p[2];
    const float nt = n + nb, d = mb - mean;
    mean += d * (nb / nt);
    m2 += m2b + d * d * (n * nb / nt);
    n = nt;
  }
#pragma unroll
  for (int o = WAVE / 2; o > 0; o >>= 1) {
    const float nb = __shfl_xor(n, o, WAVE), mb = __shfl_xor(mean, o, WAVE), m2b = __shfl_xor(m2, o, WAVE);
    const float nt = n + nb;
    if (nt > 0.f) {
      const float d = mb - mean;
      mean += d * (nb / nt);
      m2 += m2b + d * d * (n * nb / nt);
    }
    n = nt;
  }
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  if (lane == 0) { s_tri[wave * 3] = n; s_tri[wave * 3 + 1] = mean; s_tri[wave * 3 + 2] = m2; }
  __syncthreads();
  n = 0.f; mean = 0.f; m2 = 0.f;
#pragma unroll
  for (int w = 0; w < BN_THREADS / WAVE; ++w) {
    const float nb = s_tri[w * 3], mb = s_tri[w * 3 + 1], m2b = s_tri[w * 3 + 2];
    const float nt = n + nb;
    if (nt > 0.f) {
      const float d = mb - mean;
      mean += d * (nb / nt);
      m2 += m2b + d * d * (n * nb / nt);
    }
    n = nt;
  }
  mean_o = mean; m2_o = m2; n_o = n;
}

// one workgroup per channel (channels-last variant: S can be ~1000 chunks)
__global__ __launch_bounds__(BN_THREADS) void bn_finalize_kernel(const float* __restrict__ part, float* __restrict__ mean_rstd,
                                                                  float* __restrict__ running_mean, float* __restrict__ running_var, int S,
                                                                  float eps, float momentum) {
  __shared__ float s_tri[3 * BN_THREADS / WAVE];
  const int c = blockIdx.x;
  float mean, m2, n;
  combine_slices(part, c, S, s_tri, mean, m2, n);
  if (threadIdx.x == 0) {
    const float var = m2 / n;
    mean_rstd[2 * c] = mean;
    mean_rstd[2 * c + 1] = rsqrtf(var + eps);
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (n > 1.f ? m2 / (n - 1.f) : var);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ part,
                                                               float* __restrict__ mean_rstd, float* __restrict__ running_mean,
                                                               float* __restrict__ running_var, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, T* __restrict__ y, int C, int HW, int vec,
                                                               int act, float eps, float momentum) {
  __shared__ float s_tri[3 * BN_THREADS / WAVE];
  const int s = blockIdx.x, c = blockIdx.y, b = blockIdx.z, S = gridDim.x * gridDim.z;
  float mean, m2, cnt_all;
  combine_slices(part, c, S, s_tri, mean, m2, cnt_all);
  const float var = m2 / cnt_all, rstd = rsqrtf(var + eps);
  if (s == 0 && b == 0 && threadIdx.x == 0) {  // the first slice of the channel publishes the statistics
    mean_rstd[2 * c] = mean;
    mean_rstd[2 * c + 1] = rstd;
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (cnt_all > 1.f ? m2 / (cnt_all - 1.f) : var);
    }
  }
  const int n = min(BN_SLICE, HW - s * BN_SLICE);
  const size_t off = ((size_t)b * C + c) * HW + (size_t)s * BN_SLICE;
  const float a = rstd * gamma[c], be = beta[c];
  const int base = threadIdx.x * BN_EPT;
  const int cnt = max(0, min(BN_EPT, n - base));
  if (vec && cnt == BN_EPT) {
#pragma unroll
    for (int q = 0; q < BN_EPT / 4; ++q) {
      float v[4];
      Elt<T>::ld4(x + off + base + 4 * q, v);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = act_fwd(fmaf(v[i] - mean, a, be), act);
      Elt<T>::st4(y + off + base + 4 * q, v);
    }
  } else {
    for (int i = 0; i < cnt; ++i) Elt<T>::st(y + off + base + i, act_fwd(fmaf(Elt<T>::ld(x + off + base + i) - mean, a, be), act));
  }
}

// part[c][slice][2] = sum dz, sum dz * xhat
template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_reduce_kernel(const T* __restrict__ gy, const T* __restrict__ x,
                                                                    const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, float* __restrict__ part, int C,
                                                                    int HW, int vec, int act) {
  __shared__ float s_red[BN_THREADS / WAVE];
  const int s = blockIdx.x, c = blockIdx.y, b = blockIdx.z, S = gridDim.x * gridDim.z;
  const int n = min(BN_SLICE, HW - s * BN_SLICE);
  const size_t off = ((size_t)b * C + c) * HW + (size_t)s * BN_SLICE;
  const float mean = mean_rstd[2 * c], rstd = mean_rstd[2 * c + 1], g = gamma[c], be = beta[c];
  float xv[BN_EPT], gv[BN_EPT];
  int cnt;
  ld_slice<T>(x + off, n, vec, xv, cnt);
  ld_slice<T>(gy + off, n, vec, gv, cnt);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < BN_EPT; ++i) {
    if (i < cnt) {
      const float xh = (xv[i] - mean) * rstd;
      const float dz = gv[i] * act_bwd(fmaf(xh, g, be), act);
      s1 += dz;
      s2 = fmaf(dz, xh, s2);
    }
  }
  s1 = block_sum(s1, s_red);
  s2 = block_sum(s2, s_red);
  if (threadIdx.x == 0) {
    float* o = part + ((size_t)c * S + (size_t)b * gridDim.x + s) * 2;
    o[0] = s1; o[1] = s2;
  }
}

template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(const T* __restrict__ gy, const T* __restrict__ x,
                                                                   const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, const float* __restrict__ part,
                                                                   T* __restrict__ gx, float* __restrict__ ggamma, float* __restrict__ gbeta,
                                                                   int C, int HW, int vec, int act, float inv_count) {
  __shared__ float s_red[BN_THREADS / WAVE];
  const int s = blockIdx.x, c = blockIdx.y, b = blockIdx.z, S = gridDim.x * gridDim.z;
  float p1 = 0.f, p2 = 0.f;
  for (int i = threadIdx.x; i < S; i += BN_THREADS) { p1 += part[((size_t)c * S + i) * 2]; p2 += part[((size_t)c * S + i) * 2 + 1]; }
  const float sum_dz = block_sum(p1, s_red), sum_dzx = block_sum(p2, s_red);
  if (s == 0 && b == 0 && threadIdx.x == 0) { ggamma[c] = sum_dzx; gbeta[c] = sum_dz; }
  const int n = min(BN_SLICE, HW - s * BN_SLICE);
  const size_t off = ((size_t)b * C + c) * HW + (size_t)s * BN_SLICE;
  const float mean = mean_rstd[2 * c], rstd = mean_rstd[2 * c + 1], g = gamma[c], be = beta[c];
  const float k1 = sum_dz * inv_count, k2 = sum_dzx * inv_count, gr = g * rstd;
  const int base = threadIdx.x * BN_EPT;
  const int cnt = max(0, min(BN_EPT, n - base));
  if (vec && cnt == BN_EPT) {
#pragma unroll
    for (int q = 0; q < BN_EPT / 4; ++q) {
      float xv[4], gv[4], o[4];
      Elt<T>::ld4(x + off + base + 4 * q, xv);
      Elt<T>::ld4(gy + off + base + 4 * q, gv);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xh = (xv[i] - mean) * rstd;
        const float dz = gv[i] * act_bwd(fmaf(xh, g, be), act);
        o[i] = gr * (dz - k1 - xh * k2);
      }
      Elt<T>::st4(gx + off + base + 4 * q, o);
    }
  } else {
    for (int i = 0; i < cnt; ++i) {
      const float xh = (Elt<T>::ld(x + off + base + i) - mean) * rstd;
      const float dz = Elt<T>::ld(gy + off + base + i) * act_bwd(fmaf(xh, g, be), act);
      Elt<T>::st(gx + off + base + i, gr * (dz - k1 - xh * k2));
    }
  }
}

// ---- channels-last variant: x [N, C] with C contiguous (token-major maps: the MEH input projection output, and the trunk when it
// runs NHWC).  The map is cut into contiguous chunks of iters x 256 x V elements, V = the elements of one 16-byte load (8 bf16, 4 fp32);
// one workgroup streams one chunk, thread t takes the t-th 16 bytes of every 256 x V piece.  256 x V is a multiple of C, so a
// thread meets the same V columns in every piece: its per-column constants stay in registers, and a chunk starts on a row.
template <typename T, int V>
struct Vec;
template <>
struct Vec<float, 4> {
  static __device__ __forceinline__ void ld(const float* p, float (&o)[4]) { Elt<float>::ld4(p, o); }
  static __device__ __forceinline__ void st(float* p, const float (&v)[4]) { Elt<float>::st4(p, v); }
};
template <>
struct Vec<bf16_t, 4> {
  static __device__ __forceinline__ void ld(const bf16_t* p, float (&o)[4]) { Elt<bf16_t>::ld4(p, o); }
  static __device__ __forceinline__ void st(bf16_t* p, const float (&v)[4]) { Elt<bf16_t>::st4(p, v); }
};
template <>
struct Vec<bf16_t, 8> {
  static __device__ __forceinline__ void ld(const bf16_t* p, float (&o)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
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

constexpr int CL_UNROLL = 4;   // 16-byte loads a thread keeps in flight; iters is a multiple of it

// Per-column sums of the workgroup's 256 threads: s_acc[t][j] (j < W values per thread, thread t owns column group t % cg) ->
// s_col[g * W + j] = the sum over the 256 / cg threads of group g.  Every thread adds 256 / cg values for W * cg / 256 outputs.
template <int W>
__device__ __forceinline__ void fold_row_lanes(const float (&mine)[W], float* s_acc, float* s_col, int cg) {
#pragma unroll
  for (int j = 0; j < W; ++j) s_acc[threadIdx.x * W + j] = mine[j];
  __syncthreads();
  const int rl = BN_THREADS / cg;
  for (int o = threadIdx.x; o < cg * W; o += BN_THREADS) {
    const int g = o / W, j = o % W;
    float a = 0.f;
    for (int q = 0; q < rl; ++q) a += s_acc[(q * cg + g) * W + j];
    s_col[o] = a;
  }
  __syncthreads();
}

// part[c][chunk][3] = rows, mean, M2 of the chunk's rows in column c.  Sums are taken of (x - k), k = the chunk's first row (well
// conditioned, and a load past the end of the map is redirected to that row: it adds 0).
// A [N, C] map that lives as B image segments inside a wider buffer (level i of the MEH token memory [B, L_0 + L_1 + L_2, C], head.py:1202-1219:
// row n = image n / rows, position n % rows): element offset of flat element e (C = 1 << cshift).  rows == 0: not segmented, row pitch ld.
struct Seg { unsigned rows; size_t pitch; };
__device__ __forceinline__ size_t seg_addr(size_t e, int C, int cshift, size_t ld, Seg sg) {
  const size_t row = e >> cshift, col = e & (size_t)(C - 1);
  if (!sg.rows) return row * ld + col;
  const unsigned img = (unsigned)row / sg.rows;
  return (size_t)img * sg.pitch + ((unsigned)row - img * sg.rows) * ld + col;
}

template <typename T, int V>
__global__ __launch_bounds__(BN_THREADS) void bncl_stats_kernel(const T* __restrict__ x, float* __restrict__ part, size_t total, int C,
                                                                 int iters) {
  __shared__ float s_acc[BN_THREADS * 2 * V];
  __shared__ float s_col[2 * 1024];
  const size_t piece = (size_t)BN_THREADS * V, start = (size_t)blockIdx.x * iters * piece, end = min(total, start + iters * piece);
  const int cg = C / V, g = threadIdx.x % cg, S = gridDim.x;
  const T* home = x + start + (size_t)g * V;
  float k[V], acc[2 * V];
  Vec<T, V>::ld(home, k);
#pragma unroll
  for (int j = 0; j < 2 * V; ++j) acc[j] = 0.f;
  const size_t e0 = start + (size_t)threadIdx.x * V;
  for (int it = 0; it < iters; it += CL_UNROLL) {
    float v[CL_UNROLL][V];
#pragma unroll
    for (int u = 0; u < CL_UNROLL; ++u) {
      const size_t e = e0 + (size_t)(it + u) * piece;
      Vec<T, V>::ld(e < end ? x + e : home, v[u]);
    }
#pragma unroll
    for (int u = 0; u < CL_UNROLL; ++u)
#pragma unroll
      for (int j = 0; j < V; ++j) { const float d = v[u][j] - k[j]; acc[j] += d; acc[V + j] = fmaf(d, d, acc[V + j]); }
  }
  fold_row_lanes<2 * V>(acc, s_acc, s_col, cg);
  const float n = (float)((end - start) / C);
  for (int c = threadIdx.x; c < C; c += BN_THREADS) {
    const float s1 = s_col[(c / V) * 2 * V + c % V], s2 = s_col[(c / V) * 2 * V + V + c % V], m = s1 / n;
    float* o = part + ((size_t)c * S + blockIdx.x) * 3;
    o[0] = n; o[1] = Elt<T>::ld(x + start + c) + m; o[2] = s2 - s1 * m;
  }
}

template <typename T, int V>
__global__ __launch_bounds__(BN_THREADS) void bncl_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean_rstd,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const T* __restrict__ res, T* __restrict__ y, size_t total, int C, int iters,
                                                                 int act, Seg ysg = Seg{0, 0}, int cshift = 0) {
  const size_t piece = (size_t)BN_THREADS * V, start = (size_t)blockIdx.x * iters * piece, end = min(total, start + iters * piece);
  const int c0 = (threadIdx.x % (C / V)) * V;
  const size_t home = start + c0;   // first row of the chunk: where a load past the end is redirected
  float mean[V], a[V], be[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { mean[j] = mean_rstd[2 * (c0 + j)]; a[j] = mean_rstd[2 * (c0 + j) + 1] * gamma[c0 + j]; be[j] = beta[c0 + j]; }
  const size_t e0 = start + (size_t)threadIdx.x * V;
  for (int it = 0; it < iters; it += CL_UNROLL) {
    float v[CL_UNROLL][V];
#pragma unroll
    for (int u = 0; u < CL_UNROLL; ++u) {
      const size_t e = e0 + (size_t)(it + u) * piece;
      Vec<T, V>::ld(x + (e < end ? e : home), v[u]);
    }
#pragma unroll
    for (int u = 0; u < CL_UNROLL; ++u) {
      const size_t e = e0 + (size_t)(it + u) * piece;
#pragma unroll
      for (int j = 0; j < V; ++j) v[u][j] = act_fwd(fmaf(v[u][j] - mean[j], a[j], be[j]), act);
      if (res && e < end) {   // x + Conv(...)(x) of a bottleneck: the shortcut joins here instead of in an add kernel
        float r[V];
        Vec<T, V>::ld(res + e, r);
#pragma unroll
        for (int j = 0; j < V; ++j) v[u][j] += r[j];
      }
      if (e < end) Vec<T, V>::st(y + (ysg.rows ? seg_addr(e, C, cshift, (size_t)C, ysg) : e), v[u]);
    }
  }
}

// ---- two BatchNorms into one activation: y = act(bn1(x1) + bn2(x2)) - RepConvN's training form (3x3 branch + 1x1 branch,
// extra_modules/block.py:66-69).  As separate ops that was two applies, an add and a SiLU forward (11 map passes) and a SiLU backward
// plus two BatchNorm backwards (13); here 3 + 8.  Same chunking and register-resident column constants as the single kernels.
template <typename T, int V>
__global__ __launch_bounds__(BN_THREADS) void bncl2_apply_kernel(const T* __restrict__ x1, const T* __restrict__ x2,
                                                                  const float* __restrict__ mr1, const float* __restrict__ mr2,
                                                                  const float* __restrict__ g1, const float* __restrict__ b1,
                                                                  const float* __restrict__ g2, const float* __restrict__ b2,
                                                                  T* __restrict__ y, size_t total, int C, int iters, int act) {
  const size_t piece = (size_t)BN_THREADS * V, start = (size_t)blockIdx.x * iters * piece, end = min(total, start + iters * piece);
  const int c0 = (threadIdx.x % (C / V)) * V;
  const size_t home = start + c0;
  float m1[V], a1[V], m2[V], a2[V], be[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    m1[j] = mr1[2 * (c0 + j)]; a1[j] = mr1[2 * (c0 + j) + 1] * g1[c0 + j];
    m2[j] = mr2[2 * (c0 + j)]; a2[j] = mr2[2 * (c0 + j) + 1] * g2[c0 + j];
    be[j] = b1[c0 + j] + b2[c0 + j];
  }
  const size_t e0 = start + (size_t)threadIdx.x * V;
  for (int it = 0; it < iters; it += CL_UNROLL) {
    float v[CL_UNROLL][V], w[CL_UNROLL][V];
#pragma unroll
    for (int u = 0; u < CL_UNROLL; ++u) {
      const size_t e = e0 + (size_t)(it + u) * piece, ee = e < end ? e : home;
      Vec<T, V>::ld(x1 + ee, v[u]);
      Vec<T, V>::ld(x2 + ee, w[u]);
    }
#pragma unroll
    for (int u = 0; u < CL_UNROLL; ++u) {
      const size_t e = e0 + (size_t)(it + u) * piece;
#pragma unroll
      for (int j = 0; j < V; ++j) v[u][j] = act_fwd(fmaf(w[u][j] - m2[j], a2[j], fmaf(v[u][j] - m1[j], a1[j], be[j])), act);
      if (e < end) Vec<T, V>::st(y + e, v[u]);
    }
  }
}

// part[c][chunk][3] = sum dz, sum dz * xhat1, sum dz * xhat2 with dz = gy * act'(z1 + z2)
template <typename T, int V>
__global__ __launch_bounds__(BN_THREADS) void bncl2_bwd_reduce_kernel(const T* __restrict__ gy, const T* __restrict__ x1, const T* __restrict__ x2,
                                                                       const float* __restrict__ mr1, const float* __restrict__ mr2,
                                                                       const float* __restrict__ g1, const float* __restrict__ b1,
                                                                       const float* __restrict__ g2, const float* __restrict__ b2,
                                                                       float* __restrict__ part, size_t total, int C, int iters, int act,
                                                                       size_t ldgy, int cshift) {
  __shared__ float s_acc[BN_THREADS * 3 * V];
  __shared__ float s_col[3 * 1024];
  const size_t piece = (size_t)BN_THREADS * V, start = (size_t)blockIdx.x * iters * piece, end = min(total, start + iters * piece);
  const int cg = C / V, c0 = (threadIdx.x % cg) * V, S = gridDim.x;
  const size_t home = start + c0;
  float m1[V], r1[V], m2[V], r2[V], ga1[V], ga2[V], be[V], acc[3 * V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    m1[j] = mr1[2 * (c0 + j)]; r1[j] = mr1[2 * (c0 + j) + 1]; m2[j] = mr2[2 * (c0 + j)]; r2[j] = mr2[2 * (c0 + j) + 1];
    ga1[j] = g1[c0 + j]; ga2[j] = g2[c0 + j]; be[j] = b1[c0 + j] + b2[c0 + j];
    acc[j] = 0.f; acc[V + j] = 0.f; acc[2 * V + j] = 0.f;
  }
  const size_t e0 = start + (size_t)threadIdx.x * V;
  for (int it = 0; it < iters; it += CL_UNROLL) {
    float xa[CL_UNROLL][V], xb[CL_UNROLL][V], gv[CL_UNROLL][V];
#pragma unroll
    for (int u = 0; u < CL_UNROLL; ++u) {
      const size_t e = e0 + (size_t)(it + u) * piece, ee = e < end ? e : home;
      Vec<T, V>::ld(x1 + ee, xa[u]);
      Vec<T, V>::ld(x2 + ee, xb[u]);
      Vec<T, V>::ld(gy + (ee >> cshift) * ldgy + (ee & (size_t)(C - 1)), gv[u]);
    }
#pragma unroll
    for (int u = 0; u < CL_UNROLL; ++u) {
      const float live = e0 + (size_t)(it + u) * piece < end ? 1.f : 0.f;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float h1 = (xa[u][j] - m1[j]) * r1[j], h2 = (xb[u][j] - m2[j]) * r2[j];
        const float dz = live * gv[u][j] * act_bwd(fmaf(h2, ga2[j], fmaf(h1, ga1[j], be[j])), act);
        acc[j] += dz; acc[V + j] = fmaf(dz, h1, acc[V + j]); acc[2 * V + j] = fmaf(dz, h2, acc[2 * V + j]);
      }
    }
  }
  fold_row_lanes<3 * V>(acc, s_acc, s_col, cg);
  for (int c = threadIdx.x; c < C; c += BN_THREADS) {
    float* o = part + ((size_t)c * S + blockIdx.x) * 3;
    o[0] = s_col[(c / V) * 3 * V + c % V]; o[1] = s_col[(c / V) * 3 * V + V + c % V]; o[2] = s_col[(c / V) * 3 * V + 2 * V + c % V];
  }
}

// sums[c][3] over the S chunks; d(beta) of both BatchNorms = sum dz, d(gamma_i) = sum dz * xhat_i
__global__ __launch_bounds__(BN_THREADS) void bncl2_sum_kernel(const float* __restrict__ part, float* __restrict__ sums, float* __restrict__ gg1,
                                                                float* __restrict__ gb1, float* __restrict__ gg2, float* __restrict__ gb2, int S) {
  __shared__ float s_red[BN_THREADS / WAVE];
  const int c = blockIdx.x;
  float a = 0.f, b = 0.f, d = 0.f;
  for (int s = threadIdx.x; s < S; s += BN_THREADS) {
    const float* p = part + ((size_t)c * S + s) * 3;
    a += p[0]; b += p[1]; d += p[2];
  }
  a = block_sum(a, s_red); b = block_sum(b, s_red); d = block_sum(d, s_red);
  if (threadIdx.x == 0) { sums[3 * c] = a; sums[3 * c + 1] = b; sums[3 * c + 2] = d; gb1[c] = a; gb2[c] = a; gg1[c] = b; gg2[c] = d; }
}

template <typename T, int V>
__global__ __launch_bounds__(BN_THREADS) void bncl2_bwd_apply_kernel(const T* __restrict__ gy, const T* __restrict__ x1, const T* __restrict__ x2,
                                                                      const float* __restrict__ mr1, const float* __restrict__ mr2,
                                                                      const float* __restrict__ g1, const float* __restrict__ b1,
                                                                      const float* __restrict__ g2, const float* __restrict__ b2,
                                                                      const float* __restrict__ sums, T* __restrict__ gx1, T* __restrict__ gx2,
                                                                      size_t total, int C, int iters, int act, float inv_count, size_t ldgy,
                                                                      int cshift) {
  const size_t piece = (size_t)BN_THREADS * V, start = (size_t)blockIdx.x * iters * piece, end = min(total, start + iters * piece);
  const int c0 = (threadIdx.x % (C / V)) * V;
  const size_t home = start + c0;
  float m1[V], r1[V], m2[V], r2[V], ga1[V], ga2[V], be[V], k1[V], k2[V], k3[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    m1[j] = mr1[2 * (c0 + j)]; r1[j] = mr1[2 * (c0 + j) + 1]; m2[j] = mr2[2 * (c0 + j)]; r2[j] = mr2[2 * (c0 + j) + 1];
    ga1[j] = g1[c0 + j]; ga2[j] = g2[c0 + j]; be[j] = b1[c0 + j] + b2[c0 + j];
    k1[j] = sums[3 * (c0 + j)] * inv_count; k2[j] = sums[3 * (c0 + j) + 1] * inv_count; k3[j] = sums[3 * (c0 + j) + 2] * inv_count;
  }
  const size_t e0 = start + (size_t)threadIdx.x * V;
  for (int it = 0; it < iters; it += CL_UNROLL / 2) {   // three streams in, two out: half the unroll keeps the registers in bounds
    float xa[CL_UNROLL / 2][V], xb[CL_UNROLL / 2][V], gv[CL_UNROLL / 2][V];
#pragma unroll
    for (int u = 0; u < CL_UNROLL / 2; ++u) {
      const size_t e = e0 + (size_t)(it + u) * piece, ee = e < end ? e : home;
      Vec<T, V>::ld(x1 + ee, xa[u]);
      Vec<T, V>::ld(x2 + ee, xb[u]);
      Vec<T, V>::ld(gy + (ee >> cshift) * ldgy + (ee & (size_t)(C - 1)), gv[u]);
    }
#pragma unroll
    for (int u = 0; u < CL_UNROLL / 2; ++u) {
      const size_t e = e0 + (size_t)(it + u) * piece;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float h1 = (xa[u][j] - m1[j]) * r1[j], h2 = (xb[u][j] - m2[j]) * r2[j];
        const float dz = gv[u][j] * act_bwd(fmaf(h2, ga2[j], fmaf(h1, ga1[j], be[j])), act);
        xa[u][j] = ga1[j] * r1[j] * (dz - k1[j] - h1 * k2[j]);
        xb[u][j] = ga2[j] * r2[j] * (dz - k1[j] - h2 * k3[j]);
      }
      if (e < end) { Vec<T, V>::st(gx1 + e, xa[u]); Vec<T, V>::st(gx2 + e, xb[u]); }
    }
  }
}

// part[c][chunk][2] = sum dz, sum dz * xhat over the chunk's rows
template <typename T, int V>
__global__ __launch_bounds__(BN_THREADS) void bncl_bwd_reduce_kernel(const T* __restrict__ gy, const T* __restrict__ x,
                                                                      const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                                      const float* __restrict__ beta, float* __restrict__ part, size_t total,
                                                                      int C, int iters, int act, size_t ldgy, int cshift, Seg gsg = Seg{0, 0}) {
  __shared__ float s_acc[BN_THREADS * 2 * V];
  __shared__ float s_col[2 * 1024];
  const size_t piece = (size_t)BN_THREADS * V, start = (size_t)blockIdx.x * iters * piece, end = min(total, start + iters * piece);
  const int cg = C / V, c0 = (threadIdx.x % cg) * V, S = gridDim.x;
  const size_t home = start + c0;
  float mean[V], rstd[V], gm[V], be[V], acc[2 * V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    mean[j] = mean_rstd[2 * (c0 + j)]; rstd[j] = mean_rstd[2 * (c0 + j) + 1]; gm[j] = gamma[c0 + j]; be[j] = beta[c0 + j];
    acc[j] = 0.f; acc[V + j] = 0.f;
  }
  const size_t e0 = start + (size_t)threadIdx.x * V;
  for (int it = 0; it < iters; it += CL_UNROLL) {
    float xv[CL_UNROLL][V], gv[CL_UNROLL][V];
#pragma unroll
    for (int u = 0; u < CL_UNROLL; ++u) {
      const size_t e = e0 + (size_t)(it + u) * piece, ee = e < end ? e : home;
      Vec<T, V>::ld(x + ee, xv[u]);
      Vec<T, V>::ld(gy + seg_addr(ee, C, cshift, ldgy, gsg), gv[u]);   // gy may be a channel slice of a wider map
    }
#pragma unroll
    for (int u = 0; u < CL_UNROLL; ++u) {
      const float live = e0 + (size_t)(it + u) * piece < end ? 1.f : 0.f;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float xh = (xv[u][j] - mean[j]) * rstd[j];
        const float dz = live * gv[u][j] * act_bwd(fmaf(xh, gm[j], be[j]), act);
        acc[j] += dz; acc[V + j] = fmaf(dz, xh, acc[V + j]);
      }
    }
  }
  fold_row_lanes<2 * V>(acc, s_acc, s_col, cg);
  for (int c = threadIdx.x; c < C; c += BN_THREADS) {
    float* o = part + ((size_t)c * S + blockIdx.x) * 2;
    o[0] = s_col[(c / V) * 2 * V + c % V]; o[1] = s_col[(c / V) * 2 * V + V + c % V];
  }
}

// sums[c][2] = sum over the S chunks (one workgroup per channel)
__global__ __launch_bounds__(BN_THREADS) void bncl_sum_kernel(const float* __restrict__ part, float* __restrict__ sums, float* __restrict__ ggamma,
                                                               float* __restrict__ gbeta, int S) {
  __shared__ float s_red[BN_THREADS / WAVE];
  const int c = blockIdx.x;
  float a = 0.f, b = 0.f;
  for (int s = threadIdx.x; s < S; s += BN_THREADS) { a += part[((size_t)c * S + s) * 2]; b += part[((size_t)c * S + s) * 2 + 1]; }
  a = block_sum(a, s_red); b = block_sum(b, s_red);
  if (threadIdx.x == 0) { sums[2 * c] = a; sums[2 * c + 1] = b; gbeta[c] = a; ggamma[c] = b; }
}

template <typename T, int V>
__global__ __launch_bounds__(BN_THREADS) void bncl_bwd_apply_kernel(const T* __restrict__ gy, const T* __restrict__ x,
                                                                     const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, const float* __restrict__ sums,
                                                                     T* __restrict__ gx, size_t total, int C, int iters, int act,
                                                                     float inv_count, size_t ldgy, int cshift, Seg gsg = Seg{0, 0}) {
  const size_t piece = (size_t)BN_THREADS * V, start = (size_t)blockIdx.x * iters * piece, end = min(total, start + iters * piece);
  const int c0 = (threadIdx.x % (C / V)) * V;
  const size_t home = start + c0;   // first row of the chunk: where a load past the end is redirected
  float mean[V], rstd[V], gm[V], be[V], k1[V], k2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    mean[j] = mean_rstd[2 * (c0 + j)]; rstd[j] = mean_rstd[2 * (c0 + j) + 1]; gm[j] = gamma[c0 + j]; be[j] = beta[c0 + j];
    k1[j] = sums[2 * (c0 + j)] * inv_count; k2[j] = sums[2 * (c0 + j) + 1] * inv_count;
  }
  const size_t e0 = start + (size_t)threadIdx.x * V;
  for (int it = 0; it < iters; it += CL_UNROLL) {
    float xv[CL_UNROLL][V], gv[CL_UNROLL][V];
#pragma unroll
    for (int u = 0; u < CL_UNROLL; ++u) {
      const size_t e = e0 + (size_t)(it + u) * piece, ee = e < end ? e : home;
      Vec<T, V>::ld(x + ee, xv[u]);
      Vec<T, V>::ld(gy + seg_addr(ee, C, cshift, ldgy, gsg), gv[u]);   // gy may be a channel slice of a wider map
    }
#pragma unroll
    for (int u = 0; u < CL_UNROLL; ++u) {
      const size_t e = e0 + (size_t)(it + u) * piece;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float xh = (xv[u][j] - mean[j]) * rstd[j];
        const float dz = gv[u][j] * act_bwd(fmaf(xh, gm[j], be[j]), act);
        gv[u][j] = gm[j] * rstd[j] * (dz - k1[j] - xh * k2[j]);
      }
      if (e < end) Vec<T, V>::st(gx + e, gv[u]);
    }
  }
}

}  // namespace

extern "C" int tamtr_bn_slices(int B, int HW) { return B * ((HW + BN_SLICE - 1) / BN_SLICE); }

static int bn_check(const void* a, const void* b, int B, int C, int HW, int dtype, int act) {
  if (!a || !b || B <= 0 || C <= 0 || HW <= 0) return TAMTR_EINVAL;
  if ((dtype != TAMTR_F32 && dtype != TAMTR_BF16) || (act != 0 && act != 1)) return TAMTR_EINVAL;
  if (B > 65535 || C > 65535) return TAMTR_EUNSUP;
  return TAMTR_OK;
}
static int bn_vec(const void* p, const void* q, int HW, int dtype) {
  const int e = dtype == TAMTR_F32 ? 4 : 2;
  return HW % 4 == 0 && ((uintptr_t)p % (4 * e)) == 0 && (!q || ((uintptr_t)q % (4 * e)) == 0);
}

extern "C" int tamtr_bn_act_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, void* y,
                                float* mean_rstd, float* partials, int B, int C, int HW, float eps, float momentum, int act, int dtype,
                                void* stream) {
  const int rc = bn_check(x, y, B, C, HW, dtype, act);
  if (rc) return rc;
  if (!gamma || !beta || !mean_rstd || !partials) return TAMTR_EINVAL;
  const int sh = (HW + BN_SLICE - 1) / BN_SLICE, vec = bn_vec(x, y, HW, dtype);
  const dim3 grid(sh, C, B);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TAMTR_F32) hipLaunchKernelGGL(bn_stats_kernel<float>, grid, dim3(BN_THREADS), 0, s, (const float*)x, partials, C, HW, vec);
  else hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, grid, dim3(BN_THREADS), 0, s, (const bf16_t*)x, partials, C, HW, vec);
  if (dtype == TAMTR_F32)
    hipLaunchKernelGGL(bn_apply_kernel<float>, grid, dim3(BN_THREADS), 0, s, (const float*)x, partials, mean_rstd, running_mean, running_var,
                       gamma, beta, (float*)y, C, HW, vec, act, eps, momentum);
  else
    hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, grid, dim3(BN_THREADS), 0, s, (const bf16_t*)x, partials, mean_rstd, running_mean,
                       running_var, gamma, beta, (bf16_t*)y, C, HW, vec, act, eps, momentum);
  return tamtr_launch_status();
}

extern "C" int tamtr_bn_act_bwd(const void* gy, const void* x, const float* gamma, const float* beta, const float* mean_rstd, void* gx,
                                float* ggamma, float* gbeta, float* partials, int B, int C, int HW, int act, int dtype, void* stream) {
  const int rc = bn_check(gy, x, B, C, HW, dtype, act);
  if (rc) return rc;
  if (!gamma || !beta || !mean_rstd || !gx || !ggamma || !gbeta || !partials) return TAMTR_EINVAL;
  const int sh = (HW + BN_SLICE - 1) / BN_SLICE, vec = bn_vec(x, gy, HW, dtype) && bn_vec(gx, nullptr, HW, dtype);
  const dim3 grid(sh, C, B);
  hipStream_t s = (hipStream_t)stream;
  const float inv = 1.f / ((float)B * (float)HW);
  if (dtype == TAMTR_F32) {
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, grid, dim3(BN_THREADS), 0, s, (const float*)gy, (const float*)x, mean_rstd, gamma, beta,
                       partials, C, HW, vec, act);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, grid, dim3(BN_THREADS), 0, s, (const float*)gy, (const float*)x, mean_rstd, gamma, beta,
                       partials, (float*)gx, ggamma, gbeta, C, HW, vec, act, inv);
  } else {
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, grid, dim3(BN_THREADS), 0, s, (const bf16_t*)gy, (const bf16_t*)x, mean_rstd, gamma,
                       beta, partials, C, HW, vec, act);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, grid, dim3(BN_THREADS), 0, s, (const bf16_t*)gy, (const bf16_t*)x, mean_rstd, gamma,
                       beta, partials, (bf16_t*)gx, ggamma, gbeta, C, HW, vec, act, inv);
  }
  return tamtr_launch_status();
}

// ---- channels-last entry points: x [N, C], C contiguous
static int bncl_vec(int C, int dtype) { return dtype == TAMTR_BF16 && C % 8 == 0 ? 8 : 4; }
static int bncl_iters(long long N, int C, int dtype) {   // chunk length: about 1024 workgroups, 16 KB .. 256 KB each
  const long long piece = (long long)BN_THREADS * bncl_vec(C, dtype), total = N * C;
  const long long it = (total / piece / 1024 + CL_UNROLL - 1) / CL_UNROLL * CL_UNROLL;
  return (int)(it < 4 ? 4 : it > 64 ? 64 : it);
}
extern "C" int tamtr_bncl_blocks(long long N, int C, int dtype) {
  if (N <= 0 || C <= 0) return 0;
  const long long chunk = (long long)bncl_iters(N, C, dtype) * BN_THREADS * bncl_vec(C, dtype);
  return (int)((N * C + chunk - 1) / chunk);
}

static int bncl_check(const void* a, const void* b, long long N, int C, int dtype, int act) {
  if (!a || !b || N <= 0 || C <= 0) return TAMTR_EINVAL;
  if ((dtype != TAMTR_F32 && dtype != TAMTR_BF16) || (act != 0 && act != 1)) return TAMTR_EINVAL;
  const int cg = C / bncl_vec(C, dtype);
  // C in {4, 8, ..., 1024} with 256 % (C / V) == 0, 16-byte (8-byte when V = 4 on bf16) aligned maps
  if (C % 4 || C > 1024 || cg > BN_THREADS || (BN_THREADS % cg) || N > 2000000000LL / C) return TAMTR_EUNSUP;
  if (((uintptr_t)a | (uintptr_t)b) % 16) return TAMTR_EUNSUP;
  return TAMTR_OK;
}

static int bncl_act_fwd_impl(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                             const void* residual, void* y, float* mean_rstd, float* partials, long long N, int C, float eps,
                             float momentum, int act, int dtype, void* stream, Seg ysg, int cshift) {
  const int rc = bncl_check(x, y, N, C, dtype, act);
  if (rc) return rc;
  if (!gamma || !beta || !mean_rstd || !partials) return TAMTR_EINVAL;
  if (residual && (uintptr_t)residual % 16) return TAMTR_EUNSUP;
  const int S = tamtr_bncl_blocks(N, C, dtype), V = bncl_vec(C, dtype), iters = bncl_iters(N, C, dtype);
  const size_t total = (size_t)N * C;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TAMTR_F32) hipLaunchKernelGGL((bncl_stats_kernel<float, 4>), dim3(S), dim3(BN_THREADS), 0, s, (const float*)x, partials, total, C, iters);
  else if (V == 8) hipLaunchKernelGGL((bncl_stats_kernel<bf16_t, 8>), dim3(S), dim3(BN_THREADS), 0, s, (const bf16_t*)x, partials, total, C, iters);
  else hipLaunchKernelGGL((bncl_stats_kernel<bf16_t, 4>), dim3(S), dim3(BN_THREADS), 0, s, (const bf16_t*)x, partials, total, C, iters);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(BN_THREADS), 0, s, partials, mean_rstd, running_mean, running_var, S, eps, momentum);
  if (dtype == TAMTR_F32)
    hipLaunchKernelGGL((bncl_apply_kernel<float, 4>), dim3(S), dim3(BN_THREADS), 0, s, (const float*)x, mean_rstd, gamma, beta, (const float*)residual, (float*)y, total, C, iters, act, ysg, cshift);
  else if (V == 8)
    hipLaunchKernelGGL((bncl_apply_kernel<bf16_t, 8>), dim3(S), dim3(BN_THREADS), 0, s, (const bf16_t*)x, mean_rstd, gamma, beta, (const bf16_t*)residual, (bf16_t*)y, total, C, iters, act, ysg, cshift);
  else
    hipLaunchKernelGGL((bncl_apply_kernel<bf16_t, 4>), dim3(S), dim3(BN_THREADS), 0, s, (const bf16_t*)x, mean_rstd, gamma, beta, (const bf16_t*)residual, (bf16_t*)y, total, C, iters, act, ysg, cshift);
  return tamtr_launch_status();
}

extern "C" int tamtr_bncl_act_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                  const void* residual, void* y, float* mean_rstd, float* partials, long long N, int C, float eps,
                                  float momentum, int act, int dtype, void* stream) {
  return bncl_act_fwd_impl(x, gamma, beta, running_mean, running_var, residual, y, mean_rstd, partials, N, C, eps, momentum, act, dtype, stream,
                           Seg{0, 0}, 0);
}

/* the same with the OUTPUT written as image segments of a wider buffer: y points at the first row of image 0's segment, image b's rows
   start seg_pitch elements further each; N = images * seg_rows rows; C a power of two (see include/tamtr_hip.h) */
extern "C" int tamtr_bncl_act_seg_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, void* y,
                                      long long seg_rows, long long seg_pitch, float* mean_rstd, float* partials, long long N, int C, float eps,
                                      float momentum, int act, int dtype, void* stream) {
  if (seg_rows <= 0 || N % seg_rows || seg_pitch < seg_rows * C || seg_pitch % 8 || (C & (C - 1)) || seg_rows > 0x7fffffffLL) return TAMTR_EUNSUP;
  int cshift = 0;
  while ((1 << cshift) < C) ++cshift;
  return bncl_act_fwd_impl(x, gamma, beta, running_mean, running_var, nullptr, y, mean_rstd, partials, N, C, eps, momentum, act, dtype, stream,
                           Seg{(unsigned)seg_rows, (size_t)seg_pitch}, cshift);
}

static int bncl_act_bwd_impl(const void* gy, long long ldgy, const void* x, const float* gamma, const float* beta,
                             const float* mean_rstd, void* gx, float* ggamma, float* gbeta, float* partials, long long N, int C, int act,
                             int dtype, void* stream, Seg gsg) {
  const int rc = bncl_check(gy, x, N, C, dtype, act);
  if (rc) return rc;
  if (ldgy < C || ldgy % bncl_vec(C, dtype) || (C & (C - 1))) return TAMTR_EUNSUP;   // row pitch of gy in elements (C: packed)
  int cshift = 0;
  while ((1 << cshift) < C) ++cshift;
  if (!gamma || !beta || !mean_rstd || !gx || !ggamma || !gbeta || !partials || (uintptr_t)gx % 16) return TAMTR_EINVAL;
  const int S = tamtr_bncl_blocks(N, C, dtype), V = bncl_vec(C, dtype), iters = bncl_iters(N, C, dtype);
  const size_t total = (size_t)N * C;
  hipStream_t s = (hipStream_t)stream;
  float* sums = partials + (size_t)C * S * 2;  // [C][2] after the per-chunk partials
  const float inv = 1.f / (float)N;
  if (dtype == TAMTR_F32)
    hipLaunchKernelGGL((bncl_bwd_reduce_kernel<float, 4>), dim3(S), dim3(BN_THREADS), 0, s, (const float*)gy, (const float*)x, mean_rstd, gamma, beta, partials, total, C, iters, act, (size_t)ldgy, cshift, gsg);
  else if (V == 8)
    hipLaunchKernelGGL((bncl_bwd_reduce_kernel<bf16_t, 8>), dim3(S), dim3(BN_THREADS), 0, s, (const bf16_t*)gy, (const bf16_t*)x, mean_rstd, gamma, beta, partials, total, C, iters, act, (size_t)ldgy, cshift, gsg);
  else
    hipLaunchKernelGGL((bncl_bwd_reduce_kernel<bf16_t, 4>), dim3(S), dim3(BN_THREADS), 0, s, (const bf16_t*)gy, (const bf16_t*)x, mean_rstd, gamma, beta, partials, total, C, iters, act, (size_t)ldgy, cshift, gsg);
  hipLaunchKernelGGL(bncl_sum_kernel, dim3(C), dim3(BN_THREADS), 0, s, partials, sums, ggamma, gbeta, S);
  if (dtype == TAMTR_F32)
    hipLaunchKernelGGL((bncl_bwd_apply_kernel<float, 4>), dim3(S), dim3(BN_THREADS), 0, s, (const float*)gy, (const float*)x, mean_rstd, gamma, beta, sums, (float*)gx, total, C, iters, act, inv, (size_t)ldgy, cshift, gsg);
  else if (V == 8)
    hipLaunchKernelGGL((bncl_bwd_apply_kernel<bf16_t, 8>), dim3(S), dim3(BN_THREADS), 0, s, (const bf16_t*)gy, (const bf16_t*)x, mean_rstd, gamma, beta, sums, (bf16_t*)gx, total, C, iters, act, inv, (size_t)ldgy, cshift, gsg);
  else
    hipLaunchKernelGGL((bncl_bwd_apply_kernel<bf16_t, 4>), dim3(S), dim3(BN_THREADS), 0, s, (const bf16_t*)gy, (const bf16_t*)x, mean_rstd, gamma, beta, sums, (bf16_t*)gx, total, C, iters, act, inv, (size_t)ldgy, cshift, gsg);
  return tamtr_launch_status();
}

extern "C" int tamtr_bncl_act_bwd(const void* gy, long long ldgy, const void* x, const float* gamma, const float* beta,
                                  const float* mean_rstd, void* gx, float* ggamma, float* gbeta, float* partials, long long N, int C, int act,
                                  int dtype, void* stream) {
  return bncl_act_bwd_impl(gy, ldgy, x, gamma, beta, mean_rstd, gx, ggamma, gbeta, partials, N, C, act, dtype, stream, Seg{0, 0});
}

/* the same with the incoming gradient read as image segments of a wider buffer (the gradient of tamtr_bncl_act_seg_fwd's output) */
extern "C" int tamtr_bncl_act_seg_bwd(const void* gy, long long seg_rows, long long seg_pitch, const void* x, const float* gamma, const float* beta,
                                      const float* mean_rstd, void* gx, float* ggamma, float* gbeta, float* partials, long long N, int C, int act,
                                      int dtype, void* stream) {
  if (seg_rows <= 0 || N % seg_rows || seg_pitch < seg_rows * C || seg_pitch % 8 || seg_rows > 0x7fffffffLL) return TAMTR_EUNSUP;
  return bncl_act_bwd_impl(gy, C, x, gamma, beta, mean_rstd, gx, ggamma, gbeta, partials, N, C, act, dtype, stream,
                           Seg{(unsigned)seg_rows, (size_t)seg_pitch});
}

// ---- y = act(bn1(x1) + bn2(x2)): both BatchNorms in training mode over the same [N, C] shape (RepConvN)
extern "C" int tamtr_bncl2_act_fwd(const void* x1, const float* gamma1, const float* beta1, float* running_mean1, float* running_var1,
                                   const void* x2, const float* gamma2, const float* beta2, float* running_mean2, float* running_var2,
                                   void* y, float* mean_rstd, float* partials, long long N, int C, float eps, float momentum, int act,
                                   int dtype, void* stream) {
  int rc = bncl_check(x1, y, N, C, dtype, act);
  if (!rc) rc = bncl_check(x2, y, N, C, dtype, act);
  if (rc) return rc;
  if (!gamma1 || !beta1 || !gamma2 || !beta2 || !mean_rstd || !partials) return TAMTR_EINVAL;
  const int S = tamtr_bncl_blocks(N, C, dtype), V = bncl_vec(C, dtype), iters = bncl_iters(N, C, dtype);
  const size_t total = (size_t)N * C;
  hipStream_t s = (hipStream_t)stream;
  float* mr2 = mean_rstd + 2 * (size_t)C;           // [2][C][2]
  float* part2 = partials + (size_t)C * S * 3;      // [2][C][S][3]
#define STATS(T, VV, X, P) hipLaunchKernelGGL((bncl_stats_kernel<T, VV>), dim3(S), dim3(BN_THREADS), 0, s, (const T*)X, P, total, C, iters)
#define APPLY(T, VV)                                                                                                                   \
  hipLaunchKernelGGL((bncl2_apply_kernel<T, VV>), dim3(S), dim3(BN_THREADS), 0, s, (const T*)x1, (const T*)x2, mean_rstd, mr2, gamma1, beta1, \
                     gamma2, beta2, (T*)y, total, C, iters, act)
  if (dtype == TAMTR_F32) { STATS(float, 4, x1, partials); STATS(float, 4, x2, part2); }
  else if (V == 8) { STATS(bf16_t, 8, x1, partials); STATS(bf16_t, 8, x2, part2); }
  else { STATS(bf16_t, 4, x1, partials); STATS(bf16_t, 4, x2, part2); }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(BN_THREADS), 0, s, partials, mean_rstd, running_mean1, running_var1, S, eps, momentum);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(BN_THREADS), 0, s, part2, mr2, running_mean2, running_var2, S, eps, momentum);
  if (dtype == TAMTR_F32) APPLY(float, 4); else if (V == 8) APPLY(bf16_t, 8); else APPLY(bf16_t, 4);
#undef STATS
#undef APPLY
  return tamtr_launch_status();
}

extern "C" int tamtr_bncl2_act_bwd(const void* gy, long long ldgy, const void* x1, const void* x2, const float* gamma1, const float* beta1,
                                   const float* gamma2, const float* beta2, const float* mean_rstd, void* gx1, void* gx2, float* ggamma1,
                                   float* gbeta1, float* ggamma2, float* gbeta2, float* partials, long long N, int C, int act, int dtype,
                                   void* stream) {
  int rc = bncl_check(gy, x1, N, C, dtype, act);
  if (!rc) rc = bncl_check(x2, gx1, N, C, dtype, act);
  if (rc) return rc;
  if (!gamma1 || !beta1 || !gamma2 || !beta2 || !mean_rstd || !gx2 || !ggamma1 || !gbeta1 || !ggamma2 || !gbeta2 || !partials ||
      (uintptr_t)gx2 % 16)
    return TAMTR_EINVAL;
  if (ldgy < C || ldgy % bncl_vec(C, dtype) || (C & (C - 1))) return TAMTR_EUNSUP;
  int cshift = 0;
  while ((1 << cshift) < C) ++cshift;
  const int S = tamtr_bncl_blocks(N, C, dtype), V = bncl_vec(C, dtype), iters = bncl_iters(N, C, dtype);
  const size_t total = (size_t)N * C;
  hipStream_t s = (hipStream_t)stream;
  const float* mr2 = mean_rstd + 2 * (size_t)C;
  float* sums = partials + (size_t)C * S * 3;  // [C][3] after the per-chunk partials
  const float inv = 1.f / (float)N;
#define RED(T, VV)                                                                                                                       \
  hipLaunchKernelGGL((bncl2_bwd_reduce_kernel<T, VV>), dim3(S), dim3(BN_THREADS), 0, s, (const T*)gy, (const T*)x1, (const T*)x2, mean_rstd, mr2, \
                     gamma1, beta1, gamma2, beta2, partials, total, C, iters, act, (size_t)ldgy, cshift)
#define APP(T, VV)                                                                                                                       \
  hipLaunchKernelGGL((bncl2_bwd_apply_kernel<T, VV>), dim3(S), dim3(BN_THREADS), 0, s, (const T*)gy, (const T*)x1, (const T*)x2, mean_rstd, mr2,  \
                     gamma1, beta1, gamma2, beta2, sums, (T*)gx1, (T*)gx2, total, C, iters, act, inv, (size_t)ldgy, cshift)
  if (dtype == TAMTR_F32) RED(float, 4); else if (V == 8) RED(bf16_t, 8); else RED(bf16_t, 4);
  hipLaunchKernelGGL(bncl2_sum_kernel, dim3(C), dim3(BN_THREADS), 0, s, partials, sums, ggamma1, gbeta1, ggamma2, gbeta2, S);
  if (dtype == TAMTR_F32) APP(float, 4); else if (V == 8) APP(bf16_t, 8); else APP(bf16_t, 4);
#undef RED
#undef APP
  return tamtr_launch_status();
}

// ---- per-channel combine alone, for a producer that leaves (count, mean, M2) partials itself (conv3x3.hip's epilogue): partials f32
// [C][S][3] -> mean_rstd f32 [C][2] + the running-statistics update
extern "C" int tamtr_bn_finalize(const float* partials, float* mean_rstd, float* running_mean, float* running_var, int C, int S, float eps,
                                 float momentum, void* stream) {
  if (!partials || !mean_rstd || C <= 0 || S <= 0 || (!running_mean) != (!running_var)) return TAMTR_EINVAL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(BN_THREADS), 0, (hipStream_t)stream, partials, mean_rstd, running_mean, running_var, S, eps,
                     momentum);
  return tamtr_launch_status();
}

// ---- batch statistics only (for a consumer that applies the normalisation itself: the channels-last gate, gate.hip):
// mean_rstd f32 [C][2] of x [N, C] plus the running-statistics update; partials as for tamtr_bncl_act_fwd
extern "C" int tamtr_bncl_stats(const void* x, float* running_mean, float* running_var, float* mean_rstd, float* partials, long long N, int C,
                                float eps, float momentum, int dtype, void* stream) {
  const int rc = bncl_check(x, mean_rstd, N, C, dtype, 0);
  if (rc) return rc;
  if (!partials) return TAMTR_EINVAL;
  const int S = tamtr_bncl_blocks(N, C, dtype), V = bncl_vec(C, dtype), iters = bncl_iters(N, C, dtype);
  const size_t total = (size_t)N * C;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TAMTR_F32) hipLaunchKernelGGL((bncl_stats_kernel<float, 4>), dim3(S), dim3(BN_THREADS), 0, s, (const float*)x, partials, total, C, iters);
  else if (V == 8) hipLaunchKernelGGL((bncl_stats_kernel<bf16_t, 8>), dim3(S), dim3(BN_THREADS), 0, s, (const bf16_t*)x, partials, total, C, iters);
  else hipLaunchKernelGGL((bncl_stats_kernel<bf16_t, 4>), dim3(S), dim3(BN_THREADS), 0, s, (const bf16_t*)x, partials, total, C, iters);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(BN_THREADS), 0, s, partials, mean_rstd, running_mean, running_var, S, eps, momentum);
  return tamtr_launch_status();
}
