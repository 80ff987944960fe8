// imgaug.hip - the pixel half of the training transforms on the device (include/tamtr_hip.h):
// RandomPerspective's affine warp -> RandomHSV -> RandomFlip -> Format -> img.float() / 255
// (reference: ultralytics/data/augment.py:415-420,590-609,636-666,920-926 + models/yolo/detect/train.py:56), one kernel per batch.
//
// Same arithmetic as the host kernels in csrc/host/imgproc.c (include/tamtr_host.h), bit for bit: the inverse map in 10-bit fixed
// point from doubles, positions quantised to 1/32 pixel, 15-bit tap weights, integer RGB->HSV with 12-bit reciprocal tables, float
// HSV->RGB.  This file is compiled with -ffp-contract=off (Makefile), as the host code is: HIP's default -ffp-contract=fast lets
// the backend fuse a multiply and an add into an FMA - also across the __fmul_rn / __fsub_rn intrinsics, which are plain operators,
// and also under `#pragma clang fp contract(off)` - and with FMAs one pixel in ~10 000 came out one level off at a rounding tie
// (measured).  With the flag the ISA holds no f32 FMA at all and f64 FMAs only inside the two correctly-rounded divisions.
// HBM-bound and tiny: per image 1.2 MB of uint8 in, 4.9 MB of fp32 out; one thread per output pixel, three coalesced plane stores.
#include "common.h"

namespace {

constexpr int IA_THREADS = 256;

__global__ __launch_bounds__(IA_THREADS) void img_augment_kernel(const uint8_t* __restrict__ src, const double* __restrict__ inv,
                                                                 const uint8_t* __restrict__ luts, const int32_t* __restrict__ flags,
                                                                 float* __restrict__ out, int SH, int SW, int H, int W, int border) {
  __shared__ int s_w[32 * 32][4];
  __shared__ int s_sdiv[256], s_hdiv[256];
  __shared__ uint8_t s_lut[3 * 256];
  const int b = blockIdx.z, y = blockIdx.y, tid = threadIdx.x;
  for (int e = tid; e < 32 * 32; e += IA_THREADS) {
    const float ty = __fmul_rn((float)(e >> 5), 0.03125f), tx = __fmul_rn((float)(e & 31), 0.03125f);   // exact: / 32
    const float wy[2] = {__fsub_rn(1.f, ty), ty}, wx[2] = {__fsub_rn(1.f, tx), tx};
    int w[4], sum = 0, big = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int v = __float2int_rn(__fmul_rn(__fmul_rn(wy[k >> 1], wx[k & 1]), 32768.f));
      w[k] = v > 32767 ? 32767 : v;
      sum += w[k];
    }
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (w[k] > w[big]) big = k;
#pragma unroll
    for (int k = 0; k < 4; ++k) s_w[e][k] = w[k] + (k == big ? 32768 - sum : 0);
  }
  for (int e = tid; e < 256; e += IA_THREADS) {
    s_sdiv[e] = e ? (int)__double2ll_rn(__ddiv_rn((double)(255 << 12), (double)e)) : 0;
    s_hdiv[e] = e ? (int)__double2ll_rn(__ddiv_rn((double)(180 << 12), __dmul_rn(6.0, (double)e))) : 0;
  }
  for (int e = tid; e < 3 * 256; e += IA_THREADS) s_lut[e] = luts[(size_t)b * 768 + e];
  __syncthreads();
  const int x = blockIdx.x * IA_THREADS + tid;
  if (x >= W) return;
  const int fl = flags[b];
  const int ya = (fl & 1) ? H - 1 - y : y, xa = (fl & 2) ? W - 1 - x : x;   // where this output pixel sits in the warped image
  const double* m = inv + (size_t)b * 6;                                     // m0 m1 b1 / m3 m4 b2: destination -> source
  const long long X0 = __double2ll_rn(__dmul_rn(__dadd_rn(__dmul_rn(m[1], (double)ya), m[2]), 1024.0)) + 16;
  const long long Y0 = __double2ll_rn(__dmul_rn(__dadd_rn(__dmul_rn(m[4], (double)ya), m[5]), 1024.0)) + 16;
  const long long X = (X0 + __double2ll_rn(__dmul_rn(__dmul_rn(m[0], (double)xa), 1024.0))) >> 5;
  const long long Y = (Y0 + __double2ll_rn(__dmul_rn(__dmul_rn(m[3], (double)xa), 1024.0))) >> 5;
  const long long px = X >> 5, py = Y >> 5;
  const int* w = s_w[(int)(Y & 31) * 32 + (int)(X & 31)];
  const uint8_t* img = src + (size_t)b * SH * SW * 3;
  int acc[3] = {0, 0, 0};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const long long qx = px + (t & 1), qy = py + (t >> 1);
    if (qx >= 0 && qx < SW && qy >= 0 && qy < SH) {
      const uint8_t* p = img + ((size_t)qy * SW + (size_t)qx) * 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) acc[k] += p[k] * w[t];
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) acc[k] += border * w[t];
    }
  }
  int c[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int v = (acc[k] + (1 << 14)) >> 15;
    c[k] = v < 0 ? 0 : (v > 255 ? 255 : v);
  }
  const size_t plane = (size_t)H * W;
  float* dst = out + (size_t)b * 3 * plane + (size_t)y * W + x;
  if (fl & 4) {  // no HSV step in this sample's transform list (all gains zero)
#pragma unroll
    for (int k = 0; k < 3; ++k) dst[k * plane] = __fmul_rn((float)c[k], (float)(1.0 / 255.0));
    return;
  }
  // ---- RGB -> HSV (integer), look-up, HSV -> RGB (float)
  const int r = c[0], g = c[1], bl = c[2];
  const int v = max(r, max(g, bl)), diff = v - min(r, min(g, bl));
  const int s = (diff * s_sdiv[v] + (1 << 11)) >> 12;
  int h = v == r ? g - bl : (v == g ? bl - r + 2 * diff : r - g + 4 * diff);
  h = (h * s_hdiv[diff] + (1 << 11)) >> 12;
  if (h < 0) h += 180;
  const int Hq = s_lut[h & 255], Sq = s_lut[256 + s], Vq = s_lut[512 + v];
  const float vf = __fmul_rn((float)Vq, (float)(1 / 255.0)), sf = __fmul_rn((float)Sq, (float)(1 / 255.0));
  float o[3];
  if (Sq == 0) {
    o[0] = o[1] = o[2] = vf;
  } else {
    float hf = __fmul_rn((float)Hq, (float)(6.0 / 180.0));
    if (hf >= 6.f) hf = __fsub_rn(hf, 6.f);
    int sector = (int)floorf(hf);
    float fr = __fsub_rn(hf, (float)sector);
    if (sector >= 6) { sector = 0; fr = 0.f; }
    const float tab[4] = {vf, __fmul_rn(vf, __fsub_rn(1.f, sf)), __fmul_rn(vf, __fsub_rn(1.f, __fmul_rn(sf, fr))),
                          __fmul_rn(vf, __fsub_rn(1.f, __fmul_rn(sf, __fsub_rn(1.f, fr))))};
    const int pick[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};   // (b, g, r) per sextant
    o[0] = tab[pick[sector][2]];
    o[1] = tab[pick[sector][1]];
    o[2] = tab[pick[sector][0]];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float q = rintf(__fmul_rn(o[k], 255.f));
    q = q < 0.f ? 0.f : (q > 255.f ? 255.f : q);
    // Format's uint8 pixel, then the trainer's img.float() / 255 - which torch evaluates on the device as a multiplication by
    // float(1.0 / 255.0) (division by a host scalar, ATen BinaryDivTrueKernel); same here so that both paths agree bit for bit
    dst[k * plane] = __fmul_rn(q, (float)(1.0 / 255.0));
  }
}

}  // namespace

extern "C" int tamtr_img_augment_u8(const uint8_t* src, const double* inv_affine, const uint8_t* luts, const int32_t* flags, float* out,
                                    int B, int SH, int SW, int H, int W, int border, void* stream) {
  if (!src || !inv_affine || !luts || !flags || !out || B <= 0 || SH <= 0 || SW <= 0 || H <= 0 || W <= 0) return TAMTR_EINVAL;
  if (border < 0 || border > 255) return TAMTR_EINVAL;
  if (B > 65535 || H > 65535) return TAMTR_EUNSUP;
  const dim3 grid((unsigned)((W + IA_THREADS - 1) / IA_THREADS), (unsigned)H, (unsigned)B);
  hipLaunchKernelGGL(img_augment_kernel, grid, dim3(IA_THREADS), 0, (hipStream_t)stream, src, inv_affine, luts, flags, out, SH, SW, H, W, border);
  return tamtr_launch_status();
}
