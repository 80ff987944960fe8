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

// Thread -> element: NCHW: blockIdx.y = plane (b * C + c), the x axis runs over the plane's pixels (w fastest);
//                    NHWC: blockIdx.y = pixel row (b * HH + h), the x axis runs over (w, c) with c fastest.  32-bit index math only.
struct Where {
  int c, h, w;
  int unit;   // NCHW: plane index b * C + c; NHWC: image index b
};
__device__ __forceinline__ bool locate(const PoolGeom& g, int HH, int WW, Where& q) {
  const int i = blockIdx.x * P_THREADS + threadIdx.x;
  if (g.nhwc) {
    if (i >= WW * g.C) return false;
    q.c = i % g.C; q.w = i / g.C; q.h = blockIdx.y % HH;
    q.unit = blockIdx.y / HH;
  } else {
    if (i >= HH * WW) return false;
    q.w = i % WW; q.h = i / WW; q.c = 0;
    q.unit = blockIdx.y;
  }
  return true;
}
__device__ __forceinline__ size_t off(const PoolGeom& g, int WW, int c, int h, int w) {   // inside the plane (NCHW) / image (NHWC)
  return g.nhwc ? ((size_t)h * WW + w) * g.C + c : (size_t)h * WW + w;
}
__device__ __forceinline__ size_t base(const PoolGeom& g, int unit, int HH, int WW) {      // of the plane (NCHW) / image (NHWC)
  return (size_t)unit * HH * WW * (g.nhwc ? g.C : 1);
}

// KS = 16 * k + s known at compile time for the two shapes of the graph (integer divisions by s become shifts), 0 = run-time values
template <typename T, int KS>
__global__ __launch_bounds__(P_THREADS) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ code,
                                                                PoolGeom g) {
  if (KS) { g.k = KS / 16; g.s = KS % 16; g.p = g.k / 2; }
  Where q;
  if (!locate(g, g.Ho, g.Wo, q)) return;
  const T* xp = x + base(g, q.unit, g.H, g.W);
  const int h0 = q.h * g.s - g.p, w0 = q.w * g.s - g.p;
  float best = -INFINITY;
  int win = max(0, -h0) * g.k + max(0, -w0);   // torch starts from the window's first real element
  for (int dh = 0; dh < g.k; ++dh) {
    const int h = h0 + dh;
    if (h < 0 || h >= g.H) continue;
    for (int dw = 0; dw < g.k; ++dw) {
      const int w = w0 + dw;
      if (w < 0 || w >= g.W) continue;
      const float v = Elt<T>::ld(xp + off(g, g.W, q.c, h, w));
      if (v > best || v != v) { best = v; win = dh * g.k + dw; }
    }
  }
  const size_t o = base(g, q.unit, g.Ho, g.Wo) + off(g, g.Wo, q.c, q.h, q.w);
  Elt<T>::st(y + o, best);
  code[o] = (uint8_t)win;
}

template <typename T, int KS>
__global__ __launch_bounds__(P_THREADS) void maxpool_bwd_kernel(const T* __restrict__ gy, const uint8_t* __restrict__ code,
                                                                const T* __restrict__ addend, T* __restrict__ gx, PoolGeom g) {
  if (KS) { g.k = KS / 16; g.s = KS % 16; g.p = g.k / 2; }
  Where q;
  if (!locate(g, g.H, g.W, q)) return;
  const size_t out_base = base(g, q.unit, g.Ho, g.Wo);
  const T* gyp = gy + out_base;
  const uint8_t* cp = code + out_base;
  const int h = q.h, w = q.w;
  // windows oh with oh * s - p <= h <= oh * s - p + k - 1
  const int oh_lo = max(0, (h + g.p - g.k + g.s) / g.s), oh_hi = min(g.Ho - 1, (h + g.p) / g.s);
  const int ow_lo = max(0, (w + g.p - g.k + g.s) / g.s), ow_hi = min(g.Wo - 1, (w + g.p) / g.s);
  const size_t i = base(g, q.unit, g.H, g.W) + off(g, g.W, q.c, h, w);
  float acc = addend ? Elt<T>::ld(addend + i) : 0.f;
  for (int oh = oh_lo; oh <= oh_hi; ++oh)
    for (int ow = ow_lo; ow <= ow_hi; ++ow) {
      const size_t o = off(g, g.Wo, q.c, oh, ow);
      const int mine = (h - (oh * g.s - g.p)) * g.k + (w - (ow * g.s - g.p));
      if (cp[o] == mine) acc += Elt<T>::ld(gyp + o);
    }
  Elt<T>::st(gx + i, acc);
}

// NCHW, W % 4 == 0: a thread owns 4 consecutive pixels of a row (one 8-byte (bf16) / 16-byte (f32) load of the addend and store of gx
// instead of four 2-byte ones; the window codes and output gradients around them come from the same few cache lines)
template <typename T, int KS>
__global__ __launch_bounds__(P_THREADS) void maxpool_bwd4_kernel(const T* __restrict__ gy, const uint8_t* __restrict__ code,
                                                                 const T* __restrict__ addend, T* __restrict__ gx, PoolGeom g) {
  if (KS) { g.k = KS / 16; g.s = KS % 16; g.p = g.k / 2; }
  const int i4 = blockIdx.x * P_THREADS + threadIdx.x, w4 = g.W / 4;
  if (i4 >= g.H * w4) return;
  const int h = i4 / w4, wb = (i4 - h * w4) * 4;
  const size_t in_base = (size_t)blockIdx.y * g.H * g.W, out_base = (size_t)blockIdx.y * g.Ho * g.Wo;
  const T* gyp = gy + out_base;
  const uint8_t* cp = code + out_base;
  const int oh_lo = max(0, (h + g.p - g.k + g.s) / g.s), oh_hi = min(g.Ho - 1, (h + g.p) / g.s);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (addend) Elt<T>::ld4(addend + in_base + (size_t)h * g.W + wb, acc);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int w = wb + j;
    const int ow_lo = max(0, (w + g.p - g.k + g.s) / g.s), ow_hi = min(g.Wo - 1, (w + g.p) / g.s);
    for (int oh = oh_lo; oh <= oh_hi; ++oh)
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const size_t o = (size_t)oh * g.Wo + ow;
        const int mine = (h - (oh * g.s - g.p)) * g.k + (w - (ow * g.s - g.p));
        if (cp[o] == mine) acc[j] += Elt<T>::ld(gyp + o);
      }
  }
  Elt<T>::st4(gx + in_base + (size_t)h * g.W + wb, acc);
}

int pool_geom(PoolGeom& g, int B, int C, int H, int W, int k, int s, int p, int nhwc) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0 || s <= 0 || p < 0) return TAMTR_EINVAL;
  if (k > 15 || 2 * p > k || H + 2 * p < k || W + 2 * p < k) return TAMTR_EUNSUP;   // code fits a byte; every window holds a real element
  g = PoolGeom{C, H, W, (H + 2 * p - k) / s + 1, (W + 2 * p - k) / s + 1, k, s, p, nhwc ? 1 : 0};
  return TAMTR_OK;
}

// grid for a map of HH x WW elements per (image, channel): see locate()
dim3 pool_grid(const PoolGeom& g, int B, int HH, int WW) {
  const long long per = g.nhwc ? (long long)WW * g.C : (long long)HH * WW, rows = g.nhwc ? (long long)B * HH : (long long)B * g.C;
  if (per > 0x7fffffffLL - P_THREADS || rows > 65535LL * 65535LL) return dim3(0);
  // blockIdx.y is limited to 65535 on this launch path: fold the excess into z is not needed for the maps of this graph
  if (rows > 65535) return dim3(0);
  return dim3((unsigned)((per + P_THREADS - 1) / P_THREADS), (unsigned)rows);
}

}  // namespace

extern "C" int tamtr_maxpool_out(int n, int k, int s, int p) { return (n + 2 * p - k) / s + 1; }

extern "C" int tamtr_maxpool_fwd(const void* x, void* y, uint8_t* code, int B, int C, int H, int W, int k, int s, int p, int nhwc,
                                 int dtype, void* stream) {
  PoolGeom g;
  const int rc = pool_geom(g, B, C, H, W, k, s, p, nhwc);
  if (rc) return rc;
  if (!x || !y || !code || (dtype != TAMTR_F32 && dtype != TAMTR_BF16)) return TAMTR_EINVAL;
  const dim3 grid = pool_grid(g, B, g.Ho, g.Wo);
  if (!grid.x) return TAMTR_EUNSUP;
  hipStream_t st = (hipStream_t)stream;
  const int ks = (p == k / 2 && ((k == 3 && s == 2) || (k == 5 && s == 1))) ? 16 * k + s : 0;
#define FWD(T, KS) hipLaunchKernelGGL((maxpool_fwd_kernel<T, KS>), grid, dim3(P_THREADS), 0, st, (const T*)x, (T*)y, code, g)
  if (dtype == TAMTR_F32) { if (ks == 50) FWD(float, 50); else if (ks == 81) FWD(float, 81); else FWD(float, 0); }
  else { if (ks == 50) FWD(bf16_t, 50); else if (ks == 81) FWD(bf16_t, 81); else FWD(bf16_t, 0); }
#undef FWD
  return tamtr_launch_status();
}

extern "C" int tamtr_maxpool_bwd(const void* gy, const uint8_t* code, const void* addend, void* gx, int B, int C, int H, int W, int k, int s, int p, int nhwc,
                                 int dtype, void* stream) {
  PoolGeom g;
  const int rc = pool_geom(g, B, C, H, W, k, s, p, nhwc);
  if (rc) return rc;
  if (!gy || !gx || !code || (dtype != TAMTR_F32 && dtype != TAMTR_BF16)) return TAMTR_EINVAL;
  const dim3 grid = pool_grid(g, B, H, W);
  if (!grid.x) return TAMTR_EUNSUP;
  hipStream_t st = (hipStream_t)stream;
  const int ks = (p == k / 2 && ((k == 3 && s == 2) || (k == 5 && s == 1))) ? 16 * k + s : 0;
  const int al = dtype == TAMTR_F32 ? 16 : 8;
  if (!g.nhwc && W % 4 == 0 && ((uintptr_t)gx % al) == 0 && (!addend || ((uintptr_t)addend % al) == 0) && (long long)B * C <= 65535) {
    const dim3 g4((unsigned)(((long long)H * (W / 4) + P_THREADS - 1) / P_THREADS), (unsigned)(B * C));
#define BWD4(T, KS) hipLaunchKernelGGL((maxpool_bwd4_kernel<T, KS>), g4, dim3(P_THREADS), 0, st, (const T*)gy, code, (const T*)addend, (T*)gx, g)
    if (dtype == TAMTR_F32) { if (ks == 50) BWD4(float, 50); else if (ks == 81) BWD4(float, 81); else BWD4(float, 0); }
    else { if (ks == 50) BWD4(bf16_t, 50); else if (ks == 81) BWD4(bf16_t, 81); else BWD4(bf16_t, 0); }
#undef BWD4
    return tamtr_launch_status();
  }
#define BWD(T, KS) hipLaunchKernelGGL((maxpool_bwd_kernel<T, KS>), grid, dim3(P_THREADS), 0, st, (const T*)gy, code, (const T*)addend, (T*)gx, g)
  if (dtype == TAMTR_F32) { if (ks == 50) BWD(float, 50); else if (ks == 81) BWD(float, 81); else BWD(float, 0); }
  else { if (ks == 50) BWD(bf16_t, 50); else if (ks == 81) BWD(bf16_t, 81); else BWD(bf16_t, 0); }
#undef BWD
  return tamtr_launch_status();
}
