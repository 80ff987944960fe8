// bn.hip - training-mode BatchNorm2d (+ optional SiLU) on NCHW maps, forward and backward, gfx950.
//
// Reference: `self.act(self.bn(self.conv(x)))`, ultralytics/nn/modules/conv.py:36-40 (Conv), with the BatchNorm settings the
// reference rewrites after construction (eps 1e-3, momentum 0.03, utils/torch_utils.py:303-313); also the bare BatchNorms of
// RepConvN branches and of the MEH input projection.  MIOpen's spatial BatchNorm gives every channel to ONE workgroup: the
// [16, 64, 320, 320] map of the first layer ran on 64 workgroups of a 256-CU chip (954 us for 630 MB of traffic, 0.66 TB/s),
// and SiLU was a further full pass each way.  Here a channel is split over B x ceil(HW / 4096) workgroups:
//   bn_stats     : per-slice (count, mean, M2) with the slice held in registers (exact two-pass inside the slice)
//   bn_finalize  : Chan-combines the slices of a channel -> batch mean / biased var -> rstd; running stats (unbiased var)
//   bn_apply     : y = act(gamma * (x - mean) * rstd + beta), act = identity | SiLU, 16-B loads and stores
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

// one workgroup (one wave) per channel
__global__ __launch_bounds__(WAVE) void bn_finalize_kernel(const float* __restrict__ part, float* __restrict__ mean_rstd,
                                                            float* __restrict__ running_mean, float* __restrict__ running_var, int S,
                                                            float eps, float momentum) {
  const int c = blockIdx.x, lane = threadIdx.x;
  // Chan's parallel combination, lane-strided then butterfly (combination is associative)
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int s = lane; s < S; s += WAVE) {
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
  if (lane == 0) {
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
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean_rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               T* __restrict__ y, int C, int HW, int vec, int act) {
  const int s = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
  const int n = min(BN_SLICE, HW - s * BN_SLICE);
  const size_t off = ((size_t)b * C + c) * HW + (size_t)s * BN_SLICE;
  const float mean = mean_rstd[2 * c], a = mean_rstd[2 * c + 1] * gamma[c], be = beta[c];
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
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(WAVE), 0, s, partials, mean_rstd, running_mean, running_var, sh * B, eps, momentum);
  if (dtype == TAMTR_F32)
    hipLaunchKernelGGL(bn_apply_kernel<float>, grid, dim3(BN_THREADS), 0, s, (const float*)x, mean_rstd, gamma, beta, (float*)y, C, HW, vec, act);
  else
    hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, grid, dim3(BN_THREADS), 0, s, (const bf16_t*)x, mean_rstd, gamma, beta, (bf16_t*)y, C, HW, vec,
                       act);
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
