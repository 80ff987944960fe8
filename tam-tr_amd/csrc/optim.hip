// optim.hip - the reference's optimizer_step (ultralytics/engine/trainer.py:471-479: clip_grad_norm_(max_norm) -> optimizer.step() ->
// ema.update(model)) as three table-driven multi-tensor kernels, gfx950.
//
// PyTorch runs this as clip_grad_norm_ (group tensors, _foreach_norm, stack, norm, clamp, _foreach_mul_), fused AdamW (group tensors,
// 19 multi-tensor launches) and the EMA's two _foreach ops over ~1 200 state_dict entries: 9.4 ms of HOST time per step in Python list
// handling (profiles/r04_host_phases.txt: 3.9 + 1.5 + 4.0 ms) for 1.5 ms of GPU work, on a step whose host side is the critical path.
// Everything about the ~750 tensors except this step's gradient addresses is fixed, so it lives in a device table built once:
//   pass 1  sqnorm     one workgroup per 8 192-element chunk: sum of g^2 -> partial[chunk]                       (reads g)
//   pass 2  finalize   one workgroup adds the partials in index order: norm = sqrt(sum), coef = min(1, max_norm / (norm + 1e-6))
//   pass 3  update     per chunk: g *= coef (written back: the clipped gradient is what clip_grad_norm_ leaves in .grad), decoupled
//                      weight decay, Adam moments, bias-corrected step (torch's FusedAdamKernel arithmetic, fp32), and the EMA
//                      v = d v + (1 - d) p of the NEW parameter value in the same pass (p is not re-read)       (g p m v e: 5 reads, 5 writes)
//                      + optionally the bf16 SHADOW of the new value: the copy autocast would cast out of the fp32 master at every use of the
//                      weight in the next step (~400 cast launches per step on this graph) is written here, where the value is in registers
// The norm never visits the host.  A tensor without a gradient this step (grad pointer 0: the 30 discarded-gate parameters, SURVEY D2;
// denoising_class_embed on a batch without boxes) skips Adam exactly like torch (its step count does not advance) but still takes part
// in the EMA; EMA-only entries (BatchNorm running statistics) have no moment buffers at all.  All sums are in a fixed order.
#include "common.h"

namespace {

constexpr int OPT_CHUNK = 8192;      // elements per workgroup
constexpr int OPT_THREADS = 256;

struct OptTable {                    // device arrays, one entry per tensor (built once by the host side, engine.FusedOptimStep)
  float* const* p;                   // parameter (or buffer) values
  float* const* m;                   // exp_avg        (0: EMA-only entry)
  float* const* v;                   // exp_avg_sq
  float* const* e;                   // EMA copy       (0: no EMA)
  unsigned short* const* sh;         // bf16 shadow of the parameter (array may be null; entry 0: none): the compute copy bf16 mode casts per use
  float* step;                       // per-tensor Adam step counts (float, as torch keeps them)
  const long long* numel;
  const unsigned char* group;        // parameter group (lr / weight decay) of the tensor
  const int* chunk_tensor;           // chunk -> tensor index
  const long long* chunk_off;        // chunk -> first element
};

__device__ __forceinline__ float block_sum(float v, float* red) {   // fixed-order tree over the 256 threads
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, WAVE);
  const int lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0)
    for (int w = 0; w < OPT_THREADS / WAVE; ++w) t += red[w];
  return t;   // valid in thread 0
}

__global__ __launch_bounds__(OPT_THREADS) void opt_sqnorm_kernel(OptTable T, const float* const* __restrict__ grads, float* __restrict__ partial) {
  __shared__ float red[OPT_THREADS / WAVE];
  const int c = blockIdx.x, ti = T.chunk_tensor[c];
  const float* g = grads[ti];
  float acc = 0.f;
  if (g) {
    const long long off = T.chunk_off[c], n = min((long long)OPT_CHUNK, T.numel[ti] - off);
    g += off;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
      const long long n4 = n / 4;
      for (long long i = threadIdx.x; i < n4; i += OPT_THREADS) {
        const float4 x = reinterpret_cast<const float4*>(g)[i];
        acc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
      }
      for (long long i = n4 * 4 + threadIdx.x; i < n; i += OPT_THREADS) acc += g[i] * g[i];
    } else {
      for (long long i = threadIdx.x; i < n; i += OPT_THREADS) acc += g[i] * g[i];
    }
  }
  const float t = block_sum(acc, red);
  if (threadIdx.x == 0) partial[c] = t;
}

__global__ __launch_bounds__(1024) void opt_finalize_kernel(const float* __restrict__ partial, int nchunks, float max_norm, float* __restrict__ out) {
  __shared__ double red[1024];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nchunks; i += 1024) acc += (double)partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]);
    out[0] = norm;                                               // total gradient norm (what clip_grad_norm_ returns)
    out[1] = max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;   // clip coefficient, clamped to 1 (torch/nn/utils/clip_grad.py)
  }
}

struct OptScalars {
  float lr[4], wd[4];                // per parameter group
  float beta1, beta2, eps, ema_decay;
};

template <bool VEC>
__device__ __forceinline__ void opt_update_span(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                float* __restrict__ e, unsigned short* __restrict__ sh, long long n, float coef, float lr, float wd,
                                                float b1, float b2, float eps, float step_size, float bc2_sqrt, float d) {
  constexpr int W = VEC ? 4 : 1;
  for (long long i = (long long)threadIdx.x * W; i + W <= n; i += (long long)OPT_THREADS * W) {
    float pv[W], ev[W];
    if (g) {
      float gv[W], mv[W], vv[W];
      if constexpr (VEC) {
        const float4 a = *reinterpret_cast<const float4*>(p + i), b = *reinterpret_cast<const float4*>(g + i), c = *reinterpret_cast<const float4*>(m + i),
                     dd = *reinterpret_cast<const float4*>(v + i);
        pv[0] = a.x; pv[1] = a.y; pv[2] = a.z; pv[3] = a.w; gv[0] = b.x; gv[1] = b.y; gv[2] = b.z; gv[3] = b.w;
        mv[0] = c.x; mv[1] = c.y; mv[2] = c.z; mv[3] = c.w; vv[0] = dd.x; vv[1] = dd.y; vv[2] = dd.z; vv[3] = dd.w;
      } else { pv[0] = p[i]; gv[0] = g[i]; mv[0] = m[i]; vv[0] = v[i]; }
#pragma unroll
      for (int k = 0; k < W; ++k) {
        gv[k] *= coef;
        pv[k] -= lr * wd * pv[k];
        mv[k] = mv[k] + (1.f - b1) * (gv[k] - mv[k]);                 // lerp(exp_avg, grad, 1 - beta1)
        vv[k] = b2 * vv[k] + (1.f - b2) * gv[k] * gv[k];
        pv[k] -= step_size * mv[k] / (sqrtf(vv[k]) / bc2_sqrt + eps);
      }
      if constexpr (VEC) {
        *reinterpret_cast<float4*>(p + i) = make_float4(pv[0], pv[1], pv[2], pv[3]);
        *reinterpret_cast<float4*>(g + i) = make_float4(gv[0], gv[1], gv[2], gv[3]);
        *reinterpret_cast<float4*>(m + i) = make_float4(mv[0], mv[1], mv[2], mv[3]);
        *reinterpret_cast<float4*>(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        if (sh) *reinterpret_cast<uint2*>(sh + i) = make_uint2((uint32_t)f2bf(pv[0]) | ((uint32_t)f2bf(pv[1]) << 16), (uint32_t)f2bf(pv[2]) | ((uint32_t)f2bf(pv[3]) << 16));
      } else { p[i] = pv[0]; g[i] = gv[0]; m[i] = mv[0]; v[i] = vv[0]; if (sh) sh[i] = f2bf(pv[0]); }
    } else if (e) {
      if constexpr (VEC) { const float4 a = *reinterpret_cast<const float4*>(p + i); pv[0] = a.x; pv[1] = a.y; pv[2] = a.z; pv[3] = a.w; }
      else pv[0] = p[i];
    }
    if (e) {
      if constexpr (VEC) { const float4 a = *reinterpret_cast<const float4*>(e + i); ev[0] = a.x; ev[1] = a.y; ev[2] = a.z; ev[3] = a.w; }
      else ev[0] = e[i];
#pragma unroll
      for (int k = 0; k < W; ++k) ev[k] = ev[k] * d + (1.f - d) * pv[k];   // v *= d; v += (1 - d) * m  (utils/torch_utils.py:410-415)
      if constexpr (VEC) *reinterpret_cast<float4*>(e + i) = make_float4(ev[0], ev[1], ev[2], ev[3]);
      else e[i] = ev[0];
    }
  }
}

__global__ __launch_bounds__(OPT_THREADS) void opt_update_kernel(OptTable T, float* const* __restrict__ grads, const float* __restrict__ normcoef,
                                                                 OptScalars S, int do_ema) {
  const int c = blockIdx.x, ti = T.chunk_tensor[c];
  const long long off = T.chunk_off[c], n = min((long long)OPT_CHUNK, T.numel[ti] - off);
  float* p = T.p[ti] + off;
  float* m = T.m[ti];
  float* g = m ? grads[ti] : nullptr;
  float* e = do_ema ? T.e[ti] : nullptr;
  unsigned short* sh = (g && T.sh) ? T.sh[ti] : nullptr;   // (a tensor without a gradient keeps its value: its shadow stays valid)
  if (!g && !e) return;
  float* v = nullptr;
  float step_size = 0.f, bc2_sqrt = 1.f;
  const int grp = T.group[ti];
  if (g) {
    g += off; m += off; v = T.v[ti] + off;
    const float step = T.step[ti] + 1.f;                           // every chunk of the tensor sees the old count; chunk 0 stores the new one below
    step_size = S.lr[grp] / (1.f - powf(S.beta1, step));
    bc2_sqrt = sqrtf(1.f - powf(S.beta2, step));
  }
  if (e) e += off;
  if (sh) sh += off;
  const uintptr_t al = reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
                       reinterpret_cast<uintptr_t>(e) | (reinterpret_cast<uintptr_t>(sh) << 1);   // (8-byte aligned bf16 quads)
  const float coef = normcoef[1];
  if ((al & 15) == 0) {
    const long long n4 = n / 4 * 4;
    opt_update_span<true>(p, g, m, v, e, sh, n4, coef, S.lr[grp], S.wd[grp], S.beta1, S.beta2, S.eps, step_size, bc2_sqrt, S.ema_decay);
    if (n4 < n)
      opt_update_span<false>(p + n4, g ? g + n4 : g, m ? m + n4 : m, v ? v + n4 : v, e ? e + n4 : e, sh ? sh + n4 : sh, n - n4, coef, S.lr[grp], S.wd[grp], S.beta1, S.beta2,
                             S.eps, step_size, bc2_sqrt, S.ema_decay);
  } else {
    opt_update_span<false>(p, g, m, v, e, sh, n, coef, S.lr[grp], S.wd[grp], S.beta1, S.beta2, S.eps, step_size, bc2_sqrt, S.ema_decay);
  }
}

// the step counts advance in a kernel of their own, AFTER every chunk of the update has read the old value
__global__ void opt_step_count_kernel(OptTable T, const float* const* __restrict__ grads, int ntensors) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ntensors && T.m[i] && grads[i]) T.step[i] += 1.f;
}

}  // namespace

extern "C" int tamtr_optim_chunk(void) { return OPT_CHUNK; }

/* see include/tamtr_hip.h */
extern "C" int tamtr_optim_step(const void* const* p, const void* const* m, const void* const* v, const void* const* e, const void* const* shadow,
                                float* step,
                                const long long* numel, const unsigned char* group, const int* chunk_tensor, const long long* chunk_off,
                                const void* const* grads, int ntensors, int nchunks, float* partial, float* normcoef, const float* lr,
                                const float* wd, int ngroups, float beta1, float beta2, float eps, float max_norm, float ema_decay, int do_ema,
                                void* stream) {
  if (!p || !m || !v || !e || !step || !numel || !group || !chunk_tensor || !chunk_off || !grads || !partial || !normcoef || !lr || !wd)
    return TAMTR_EINVAL;
  if (ntensors <= 0 || nchunks <= 0 || ngroups <= 0) return TAMTR_EINVAL;
  if (ngroups > 4) return TAMTR_EUNSUP;
  OptTable T{(float* const*)p, (float* const*)m, (float* const*)v, (float* const*)e, (unsigned short* const*)shadow, step, numel, group, chunk_tensor, chunk_off};
  OptScalars S;
  for (int i = 0; i < 4; ++i) { S.lr[i] = lr[i < ngroups ? i : 0]; S.wd[i] = wd[i < ngroups ? i : 0]; }
  S.beta1 = beta1; S.beta2 = beta2; S.eps = eps; S.ema_decay = ema_decay;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(opt_sqnorm_kernel, dim3(nchunks), dim3(OPT_THREADS), 0, s, T, (const float* const*)grads, partial);
  hipLaunchKernelGGL(opt_finalize_kernel, dim3(1), dim3(1024), 0, s, partial, nchunks, max_norm, normcoef);
  hipLaunchKernelGGL(opt_update_kernel, dim3(nchunks), dim3(OPT_THREADS), 0, s, T, (float* const*)grads, normcoef, S, do_ema);
  hipLaunchKernelGGL(opt_step_count_kernel, dim3((ntensors + 255) / 256), dim3(256), 0, s, T, (const float* const*)grads, ntensors);
  return tamtr_launch_status();
}
