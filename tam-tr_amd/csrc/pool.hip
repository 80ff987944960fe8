// pool.hip - max pooling (k x k, stride s, padding p, floor mode) forward and backward, NCHW or NHWC maps, gfx950.
//
// Reference call sites: SPPELAN's three chained 5x5 / stride 1 / pad 2 pools (ultralytics/nn/extra_modules/block.py:255-268, as
// nn.MaxPool2d) and the 3x3 / stride 2 / pad 1 pool of CPAM's channel gate (block.py:274).  torch's backward kernels scatter with
// atomics into a zero-filled map and carry int64 indices: max_pool_backward_nhwc took 0.32 ms for a [16, 256, 20, 20] map, the NCHW
// one 0.2 ms per CPAM site.  Here the forward stores which element of its window won as ONE byte (position inside the
// unclipped k x k window, row-major), and the backward is a gather: every input element looks at the <= ceil(k/s)^2 windows
// that contain it and adds the output gradients of those that chose it.  No atomics, no zero fill, deterministic.
// Ties: the first maximum in row-major window order wins, NaN wins over everything (torch's `val > maxval || isnan(val)`).
#include "common.h"

namespace {

constexpr int P_THREADS = 256;

struct PoolGeom {
  int C, H, W, Ho, Wo, k, s, p, nhwc;
};

// element (b, c, h, w) of a map in either layout
__device__ __forceinline__ size_t at(const PoolGeom& g, int b, int c, int h, int w, int HH, int WW) {
  return g.nhwc ? (((size_t)b * HH + h) * WW + w) * g.C + c : (((size_t)b * g.C + c) * HH + h) * WW + w;
}
// linear thread index -> (b, c, h, w) with the layout's contiguous axis fastest
__device__ __forceinline__ void unflat(const PoolGeom& g, size_t i, int HH, int WW, int& b, int& c, int& h, int& w) {
  if (g.nhwc) { c = (int)(i % g.C); i /= g.C; w = (int)(i % WW); i /= WW; h = (int)(i % HH); b = (int)(i / HH); }
  else { w = (int)(i % WW); i /= WW; h = (int)(i % HH); i /= HH; c = (int)(i % g.C); b = (int)(i / g.C); }
}

template <typename T>
__global__ __launch_bounds__(P_THREADS) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ code,
                                                                PoolGeom g, size_t n_out) {
  const size_t i = (size_t)blockIdx.x * P_THREADS + threadIdx.x;
  if (i >= n_out) return;
  int b, c, oh, ow;
  unflat(g, i, g.Ho, g.Wo, b, c, oh, ow);
  const int h0 = oh * g.s - g.p, w0 = ow * g.s - g.p;
  float best = -INFINITY;
  int win = max(0, -h0) * g.k + max(0, -w0);   // torch starts from the window's first real element
  for (int dh = 0; dh < g.k; ++dh) {
    const int h = h0 + dh;
    if (h < 0 || h >= g.H) continue;
    for (int dw = 0; dw < g.k; ++dw) {
      const int w = w0 + dw;
      if (w < 0 || w >= g.W) continue;
      const float v = Elt<T>::ld(x + at(g, b, c, h, w, g.H, g.W));
      if (v > best || v != v) { best = v; win = dh * g.k + dw; }
    }
  }
  Elt<T>::st(y + i, best);
  code[i] = (uint8_t)win;
}

template <typename T>
__global__ __launch_bounds__(P_THREADS) void maxpool_bwd_kernel(const T* __restrict__ gy, const uint8_t* __restrict__ code,
                                                                const T* __restrict__ addend, T* __restrict__ gx, PoolGeom g, size_t n_in) {
  const size_t i = (size_t)blockIdx.x * P_THREADS + threadIdx.x;
  if (i >= n_in) return;
  int b, c, h, w;
  unflat(g, i, g.H, g.W, b, c, h, w);
  // windows oh with oh * s - p <= h <= oh * s - p + k - 1
  const int oh_lo = max(0, (h + g.p - g.k + g.s) / g.s), oh_hi = min(g.Ho - 1, (h + g.p) / g.s);
  const int ow_lo = max(0, (w + g.p - g.k + g.s) / g.s), ow_hi = min(g.Wo - 1, (w + g.p) / g.s);
  float acc = addend ? Elt<T>::ld(addend + i) : 0.f;
  for (int oh = oh_lo; oh <= oh_hi; ++oh)
    for (int ow = ow_lo; ow <= ow_hi; ++ow) {
      const size_t o = at(g, b, c, oh, ow, g.Ho, g.Wo);
      const int mine = (h - (oh * g.s - g.p)) * g.k + (w - (ow * g.s - g.p));
      if (code[o] == mine) acc += Elt<T>::ld(gy + o);
    }
  Elt<T>::st(gx + i, acc);
}

int pool_geom(PoolGeom& g, int B, int C, int H, int W, int k, int s, int p, int nhwc) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0 || s <= 0 || p < 0) return TAMTR_EINVAL;
  if (k > 15 || 2 * p > k || H + 2 * p < k || W + 2 * p < k) return TAMTR_EUNSUP;   // code fits a byte; every window holds a real element
  g = PoolGeom{C, H, W, (H + 2 * p - k) / s + 1, (W + 2 * p - k) / s + 1, k, s, p, nhwc ? 1 : 0};
  return TAMTR_OK;
}

}  // namespace

extern "C" int tamtr_maxpool_out(int n, int k, int s, int p) { return (n + 2 * p - k) / s + 1; }

extern "C" int tamtr_maxpool_fwd(const void* x, void* y, uint8_t* code, int B, int C, int H, int W, int k, int s, int p, int nhwc,
                                 int dtype, void* stream) {
  PoolGeom g;
  const int rc = pool_geom(g, B, C, H, W, k, s, p, nhwc);
  if (rc) return rc;
  if (!x || !y || !code || (dtype != TAMTR_F32 && dtype != TAMTR_BF16)) return TAMTR_EINVAL;
  const size_t n = (size_t)B * C * g.Ho * g.Wo;
  const unsigned blocks = (unsigned)((n + P_THREADS - 1) / P_THREADS);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TAMTR_F32) hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(blocks), dim3(P_THREADS), 0, st, (const float*)x, (float*)y, code, g, n);
  else hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(blocks), dim3(P_THREADS), 0, st, (const bf16_t*)x, (bf16_t*)y, code, g, n);
  return tamtr_launch_status();
}

extern "C" int tamtr_maxpool_bwd(const void* gy, const uint8_t* code, const void* addend, void* gx, int B, int C, int H, int W, int k, int s, int p, int nhwc,
                                 int dtype, void* stream) {
  PoolGeom g;
  const int rc = pool_geom(g, B, C, H, W, k, s, p, nhwc);
  if (rc) return rc;
  if (!gy || !gx || !code || (dtype != TAMTR_F32 && dtype != TAMTR_BF16)) return TAMTR_EINVAL;
  const size_t n = (size_t)B * C * H * W;
  const unsigned blocks = (unsigned)((n + P_THREADS - 1) / P_THREADS);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TAMTR_F32) hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(blocks), dim3(P_THREADS), 0, st, (const float*)gy, code, (const float*)addend, (float*)gx, g, n);
  else hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(blocks), dim3(P_THREADS), 0, st, (const bf16_t*)gy, code, (const bf16_t*)addend, (bf16_t*)gx, g, n);
  return tamtr_launch_status();
}
