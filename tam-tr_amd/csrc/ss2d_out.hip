// ss2d_out.hip - SS2D back end: CrossMerge into token-major layout, and out_norm (LayerNorm) x SiLU(z) gate, gfx950.
//
// Reference: CrossMerge (ultralytics/nn/extra_modules/VManba/csms6s.py:26-34) on the four scan outputs, then
// `y = out_norm(y); y = y * act(z); out_proj(y)` (VManba/vmamba.py:1005-1008,1029-1036).  Through PyTorch the scan outputs
// [B, 4, D, L] became [B, L, D] in seven strided passes (two adds, a transposing add, a transposing copy for LayerNorm, LayerNorm,
// the gate, a cast), and the backward undid them with as many again plus LayerNorm's two backward kernels.  Here:
//   cross_merge_fwd : y4 [B,4,D,L] -> ymT [B,L,D]   (16x16 pixel x 32 channel tiles through LDS; both flattenings are read
//                     along their own contiguous axis, the token-major result is written in 128-B channel runs)
//   cross_merge_bwd : d(ymT) -> the gradient in the scan's pair layout [B,2,D,L] (what tamtr_selective_scan_*_bwd reads, xmode 3)
//   ln_gate_fwd/bwd : one wave per token: LayerNorm over D in registers, SiLU(z) read from the channels-last in_proj output where
//                     it lies (second half of each pixel's 2*D row), bf16/fp32 output; backward also yields d(z) in place,
//                     per-workgroup partial d(gamma) / d(beta) (no atomics).
#include "common.h"

namespace {

constexpr int TS = 16, CB = 32, MT_THREADS = 256, PITCH = TS + 1;

// one element of a scan plane (f32, or bf16 in the bf16 mode: include/tamtr_hip.h "bf16 PLANES")
__device__ __forceinline__ float pl_ld(const float* p) { return *p; }
__device__ __forceinline__ float pl_ld(const bf16_t* p) { return bf2f(*p); }
__device__ __forceinline__ void pl_st(float* p, float v) { *p = v; }
__device__ __forceinline__ void pl_st(bf16_t* p, float v) { *p = f2bf(v); }

template <typename PT>
__global__ __launch_bounds__(MT_THREADS) void cross_merge_fwd_kernel(const PT* __restrict__ y4, float* __restrict__ ymT, int D, int H,
                                                                      int W, int tiles_w) {
  __shared__ float s[CB][TS][PITCH];
  const int th = blockIdx.x / tiles_w, tw = blockIdx.x - th * tiles_w;
  const int d0 = blockIdx.y * CB, b = blockIdx.z, h0 = th * TS, w0 = tw * TS;
  const size_t L = (size_t)H * W;
  const PT* yb = y4 + (size_t)b * 4 * D * L;
  // (round 4: the 2 x 32 row loads of a thread used to sit one by one behind `in image ? load : 0` - a queue drain per channel, 2.1 TB/s;
  // now clamped addresses, eight channels = 16 loads in flight, and only the final store is predicated)
  {  // directions 0 and 2: row-major flattening, lanes along x
    const int ty = threadIdx.x / TS, tx = threadIdx.x % TS;
    const size_t p = (size_t)min(h0 + ty, H - 1) * W + min(w0 + tx, W - 1);
#pragma unroll
    for (int c0 = 0; c0 < CB; c0 += 8) {
      float a[8], c2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[j] = pl_ld(yb + (size_t)(d0 + c0 + j) * L + p); c2[j] = pl_ld(yb + ((size_t)2 * D + d0 + c0 + j) * L + p); }
#pragma unroll
      for (int j = 0; j < 8; ++j) s[c0 + j][ty][tx] = a[j] + c2[j];
    }
  }
  __syncthreads();
  {  // directions 1 and 3: column-major flattening, lanes along y
    const int ty = threadIdx.x % TS, tx = threadIdx.x / TS;
    const size_t p = (size_t)min(w0 + tx, W - 1) * H + min(h0 + ty, H - 1);
#pragma unroll
    for (int c0 = 0; c0 < CB; c0 += 8) {
      float a[8], c2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[j] = pl_ld(yb + ((size_t)D + d0 + c0 + j) * L + p); c2[j] = pl_ld(yb + ((size_t)3 * D + d0 + c0 + j) * L + p); }
#pragma unroll
      for (int j = 0; j < 8; ++j) s[c0 + j][ty][tx] += a[j] + c2[j];
    }
  }
  __syncthreads();
  for (int it = threadIdx.x; it < TS * TS * (CB / 4); it += MT_THREADS) {
    const int pix = it / (CB / 4), g = it - pix * (CB / 4), py = pix / TS, px = pix - py * TS;
    const int h = h0 + py, w = w0 + px;
    if (h < H && w < W)
      *reinterpret_cast<float4*>(ymT + ((size_t)b * L + (size_t)h * W + w) * D + d0 + g * 4) =
          make_float4(s[g * 4][py][px], s[g * 4 + 1][py][px], s[g * 4 + 2][py][px], s[g * 4 + 3][py][px]);
  }
}

template <typename PT>
__global__ __launch_bounds__(MT_THREADS) void cross_merge_bwd_kernel(const float* __restrict__ gT, PT* __restrict__ g2, int D, int H,
                                                                      int W, int tiles_w) {
  __shared__ float s[CB][TS][PITCH];
  const int th = blockIdx.x / tiles_w, tw = blockIdx.x - th * tiles_w;
  const int d0 = blockIdx.y * CB, b = blockIdx.z, h0 = th * TS, w0 = tw * TS;
  const size_t L = (size_t)H * W;
  {  // all of a thread's 16-byte loads first (clamped addresses): a load inside `if (in image)` ends in s_waitcnt vmcnt(0), i.e. one
     // round trip per iteration; tiles outside the image keep their (unused) clamped values
    constexpr int N_IT = TS * TS * (CB / 4) / MT_THREADS;
    static_assert(TS * TS * (CB / 4) % MT_THREADS == 0, "whole iterations");
    float4 v[N_IT];
#pragma unroll
    for (int k = 0; k < N_IT; ++k) {
      const int it = threadIdx.x + k * MT_THREADS;
      const int pix = it / (CB / 4), g = it - pix * (CB / 4), py = pix / TS, px = pix - py * TS;
      const int h = min(h0 + py, H - 1), w = min(w0 + px, W - 1);
      v[k] = *reinterpret_cast<const float4*>(gT + ((size_t)b * L + (size_t)h * W + w) * D + d0 + g * 4);
    }
#pragma unroll
    for (int k = 0; k < N_IT; ++k) {
      const int it = threadIdx.x + k * MT_THREADS;
      const int pix = it / (CB / 4), g = it - pix * (CB / 4), py = pix / TS, px = pix - py * TS;
      s[g * 4][py][px] = v[k].x; s[g * 4 + 1][py][px] = v[k].y; s[g * 4 + 2][py][px] = v[k].z; s[g * 4 + 3][py][px] = v[k].w;
    }
  }
  __syncthreads();
  PT* gb = g2 + (size_t)b * 2 * D * L;
  {
    const int ty = threadIdx.x / TS, tx = threadIdx.x % TS, h = h0 + ty, w = w0 + tx;
    if (h < H && w < W)
      for (int c = 0; c < CB; ++c) pl_st(gb + (size_t)(d0 + c) * L + (size_t)h * W + w, s[c][ty][tx]);
  }
  {
    const int ty = threadIdx.x % TS, tx = threadIdx.x / TS, h = h0 + ty, w = w0 + tx;
    if (h < H && w < W)
      for (int c = 0; c < CB; ++c) pl_st(gb + ((size_t)D + d0 + c) * L + (size_t)w * H + h, s[c][ty][tx]);
  }
}

// ---- bf16 planes (bf16 mode), H and W even: 32 x 32 pixel x 16 channel tiles.  A 16-pixel run of a bf16 plane is 32 bytes - half a
// 64-byte request - and the 16 x 16 tiles above moved bf16 planes no faster than fp32 ones (profiles/r04_ss2d_bf16_planes.txt: 0.91x / 1.12x
// at level 0).  Here a lane moves a pixel PAIR (one dword), 16 lanes a 64-byte run along the plane's own contiguous axis, and the token-major
// side is 16 channels = 64 bytes of fp32 per pixel.  LDS: [16][32][34] floats + 2 per channel plane (70 KB: two workgroups per CU).
constexpr int T2 = 32, C2B = 16, P2 = T2 + 2, PL2 = T2 * P2 + 2;

__global__ __launch_bounds__(MT_THREADS) void cross_merge_fwd16_kernel(const bf16_t* __restrict__ y4, float* __restrict__ ymT, int D, int H,
                                                                        int W, int tiles_w) {
  extern __shared__ __attribute__((aligned(16))) float s2[];   // [C2B][PL2]
  const int th = blockIdx.x / tiles_w, tw = blockIdx.x - th * tiles_w;
  const int d0 = blockIdx.y * C2B, b = blockIdx.z, h0 = th * T2, w0 = tw * T2;
  const size_t L = (size_t)H * W;
  const bf16_t* yb = y4 + (size_t)b * 4 * D * L;
  const int lo = threadIdx.x % 16, hi = threadIdx.x / 16;   // lo: the pixel pair along the plane's contiguous axis; hi, hi + 16: the lines
  {  // directions 0 and 2: row-major flattening (pairs along w)
    const int w = min(w0 + 2 * lo, W - 2);
    size_t p[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) p[r] = (size_t)min(h0 + hi + 16 * r, H - 1) * W + w;
#pragma unroll
    for (int c0 = 0; c0 < C2B; c0 += 4) {
      uint32_t a[4][2], c2[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          a[j][r] = *reinterpret_cast<const uint32_t*>(yb + (size_t)(d0 + c0 + j) * L + p[r]);
          c2[j][r] = *reinterpret_cast<const uint32_t*>(yb + ((size_t)2 * D + d0 + c0 + j) * L + p[r]);
        }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 2; ++r)
          *reinterpret_cast<float2*>(&s2[(c0 + j) * PL2 + (hi + 16 * r) * P2 + 2 * lo]) =
              make_float2(__uint_as_float(a[j][r] << 16) + __uint_as_float(c2[j][r] << 16),
                          __uint_as_float(a[j][r] & 0xffff0000u) + __uint_as_float(c2[j][r] & 0xffff0000u));
    }
  }
  __syncthreads();
  {  // directions 1 and 3: column-major flattening (pairs along h)
    const int h = min(h0 + 2 * lo, H - 2);
    size_t p[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) p[r] = (size_t)min(w0 + hi + 16 * r, W - 1) * H + h;
#pragma unroll
    for (int c0 = 0; c0 < C2B; c0 += 4) {
      uint32_t a[4][2], c2[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          a[j][r] = *reinterpret_cast<const uint32_t*>(yb + ((size_t)D + d0 + c0 + j) * L + p[r]);
          c2[j][r] = *reinterpret_cast<const uint32_t*>(yb + ((size_t)3 * D + d0 + c0 + j) * L + p[r]);
        }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          float* q = &s2[(c0 + j) * PL2 + (2 * lo) * P2 + hi + 16 * r];
          q[0] += __uint_as_float(a[j][r] << 16) + __uint_as_float(c2[j][r] << 16);
          q[P2] += __uint_as_float(a[j][r] & 0xffff0000u) + __uint_as_float(c2[j][r] & 0xffff0000u);
        }
    }
  }
  __syncthreads();
  for (int it = threadIdx.x; it < T2 * T2 * (C2B / 4); it += MT_THREADS) {
    const int pix = it / (C2B / 4), g = it - pix * (C2B / 4), py = pix / T2, px = pix - py * T2;
    const int h = h0 + py, w = w0 + px;
    const float* q = &s2[(g * 4) * PL2 + py * P2 + px];
    if (h < H && w < W)
      *reinterpret_cast<float4*>(ymT + ((size_t)b * L + (size_t)h * W + w) * D + d0 + g * 4) = make_float4(q[0], q[PL2], q[2 * PL2], q[3 * PL2]);
  }
}

__global__ __launch_bounds__(MT_THREADS) void cross_merge_bwd16_kernel(const float* __restrict__ gT, bf16_t* __restrict__ g2, int D, int H,
                                                                        int W, int tiles_w) {
  extern __shared__ __attribute__((aligned(16))) float s2[];   // [C2B][PL2]
  const int th = blockIdx.x / tiles_w, tw = blockIdx.x - th * tiles_w;
  const int d0 = blockIdx.y * C2B, b = blockIdx.z, h0 = th * T2, w0 = tw * T2;
  const size_t L = (size_t)H * W;
  {
    constexpr int N_IT = T2 * T2 * (C2B / 4) / MT_THREADS;   // 16 float4 per thread, eight in flight at a time
#pragma unroll
    for (int k0 = 0; k0 < N_IT; k0 += 8) {
      float4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int it = threadIdx.x + (k0 + k) * MT_THREADS;
        const int pix = it / (C2B / 4), g = it - pix * (C2B / 4), py = pix / T2, px = pix - py * T2;
        const int h = min(h0 + py, H - 1), w = min(w0 + px, W - 1);
        v[k] = *reinterpret_cast<const float4*>(gT + ((size_t)b * L + (size_t)h * W + w) * D + d0 + g * 4);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int it = threadIdx.x + (k0 + k) * MT_THREADS;
        const int pix = it / (C2B / 4), g = it - pix * (C2B / 4), py = pix / T2, px = pix - py * T2;
        float* q = &s2[(g * 4) * PL2 + py * P2 + px];
        q[0] = v[k].x; q[PL2] = v[k].y; q[2 * PL2] = v[k].z; q[3 * PL2] = v[k].w;
      }
    }
  }
  __syncthreads();
  bf16_t* gb = g2 + (size_t)b * 2 * D * L;
  const int lo = threadIdx.x % 16, hi = threadIdx.x / 16;
  auto pk = [](float a, float c) { return (uint32_t)f2bf(a) | ((uint32_t)f2bf(c) << 16); };
#pragma unroll
  for (int r = 0; r < 2; ++r) {  // row-major plane: pixel pairs along w
    const int h = h0 + hi + 16 * r, w = w0 + 2 * lo;
    if (h < H && w < W)
      for (int c = 0; c < C2B; ++c) {
        const float2 v = *reinterpret_cast<const float2*>(&s2[c * PL2 + (hi + 16 * r) * P2 + 2 * lo]);
        *reinterpret_cast<uint32_t*>(gb + (size_t)(d0 + c) * L + (size_t)h * W + w) = pk(v.x, v.y);
      }
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {  // column-major plane: pixel pairs along h
    const int w = w0 + hi + 16 * r, h = h0 + 2 * lo;
    if (h < H && w < W)
      for (int c = 0; c < C2B; ++c) {
        const float* q = &s2[c * PL2 + (2 * lo) * P2 + hi + 16 * r];
        *reinterpret_cast<uint32_t*>(gb + ((size_t)D + d0 + c) * L + (size_t)w * H + h) = pk(q[0], q[P2]);
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
constexpr int LG_WAVES = 4;       // tokens in flight per workgroup (one wave each)
constexpr int LG_TOK_BWD = 16;    // tokens walked by each wave of the backward kernel (d(gamma)/d(beta) accumulate in registers)

template <typename T>
__device__ __forceinline__ void ld4t(const T* p, float (&o)[4]) { Elt<T>::ld4(p, o); }

__device__ __forceinline__ float wsum(float v) { return group_sum<WAVE>(v); }
struct Vec8 {   // eight consecutive bf16 = 16 bytes
  static __device__ __forceinline__ void ld(const bf16_t* p, float (&o)[8]) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float (&v)[8]) {
    uint4 o;
    o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16); o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    o.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16); o.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
    *reinterpret_cast<uint4*>(p) = o;
  }
};

template <typename T, int D>
__global__ __launch_bounds__(LG_WAVES* WAVE) void ln_gate_fwd_kernel(const float* __restrict__ x, const T* __restrict__ xz, size_t zs,
                                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                      T* __restrict__ out, float* __restrict__ stats, size_t ntok,
                                                                      float eps) {
  constexpr int NJ = (D + 255) / 256;  // float4 pieces per lane; lanes beyond D / 4 idle when D < 256
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  const bool act = lane * 4 < D;
  const size_t tok = (size_t)blockIdx.x * LG_WAVES + wave;
  if (tok >= ntok) return;
  float v[NJ][4];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0.f;
    if (act) Elt<float>::ld4(x + tok * D + j * 256 + lane * 4, v[j]);
    sum += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
  }
  const float mean = wsum(sum) * (1.f / D);
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float c = act ? v[j][i] - mean : 0.f; sq = fmaf(c, c, sq); }
  const float rstd = rsqrtf(wsum(sq) * (1.f / D) + eps);
  if (lane == 0) { stats[2 * tok] = mean; stats[2 * tok + 1] = rstd; }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int d = j * 256 + lane * 4;
    if (!act) continue;
    float z[4], g[4], be[4], o[4];
    ld4t<T>(xz + tok * zs + D + d, z);
    Elt<float>::ld4(gamma + d, g);
    Elt<float>::ld4(beta + d, be);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float y = fmaf((v[j][i] - mean) * rstd, g[i], be[i]);
      o[i] = y * (z[i] / (1.f + __expf(-z[i])));
    }
    Elt<T>::st4(out + tok * D + d, o);
  }
}

template <typename T, int D>
__global__ __launch_bounds__(LG_WAVES* WAVE) void ln_gate_bwd_kernel(const T* __restrict__ gout, const float* __restrict__ x,
                                                                      const T* __restrict__ xz, size_t zs, const float* __restrict__ gamma,
                                                                      const float* __restrict__ beta, const float* __restrict__ stats,
                                                                      float* __restrict__ gx, T* __restrict__ gxz, float* __restrict__ part,
                                                                      size_t ntok) {
  constexpr int NJ = (D + 255) / 256;
  __shared__ float s_red[LG_WAVES][2][NJ * 256];
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  const bool act = lane * 4 < D;
  float dgam[NJ][4], dbet[NJ][4], gm[NJ][4], bt[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { dgam[j][i] = 0.f; dbet[j][i] = 0.f; gm[j][i] = 0.f; bt[j][i] = 0.f; }
    if (act) {
      Elt<float>::ld4(gamma + j * 256 + lane * 4, gm[j]);
      Elt<float>::ld4(beta + j * 256 + lane * 4, bt[j]);
    }
  }
  const size_t tok0 = ((size_t)blockIdx.x * LG_WAVES + wave) * LG_TOK_BWD;
  for (int t = 0; t < LG_TOK_BWD; ++t) {
    const size_t tok = tok0 + t;
    if (tok >= ntok) break;  // wave-uniform
    const float mean = stats[2 * tok], rstd = stats[2 * tok + 1];
    float xh[NJ][4], gxh[NJ][4];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int d = j * 256 + lane * 4;
      float xv[4] = {mean, mean, mean, mean}, z[4] = {0.f, 0.f, 0.f, 0.f}, g[4] = {0.f, 0.f, 0.f, 0.f}, gz[4];
      if (act) {
        Elt<float>::ld4(x + tok * D + d, xv);
        ld4t<T>(xz + tok * zs + D + d, z);
        ld4t<T>(gout + tok * D + d, g);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xh[j][i] = (xv[i] - mean) * rstd;
        const float y = fmaf(xh[j][i], gm[j][i], bt[j][i]);
        const float sg = 1.f / (1.f + __expf(-z[i]));
        const float gy = g[i] * (z[i] * sg);                      // through the gate
        gz[i] = g[i] * y * (sg * (1.f + z[i] * (1.f - sg)));      // d SiLU(z)
        dgam[j][i] = fmaf(gy, xh[j][i], dgam[j][i]);
        dbet[j][i] += gy;
        gxh[j][i] = gy * gm[j][i];
        c1 += gxh[j][i];
        c2 = fmaf(gxh[j][i], xh[j][i], c2);
      }
      if (act) Elt<T>::st4(gxz + tok * zs + D + d, gz);
    }
    c1 = wsum(c1) * (1.f / D);
    c2 = wsum(c2) * (1.f / D);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = rstd * (gxh[j][i] - c1 - xh[j][i] * c2);
      if (act) Elt<float>::st4(gx + tok * D + j * 256 + lane * 4, o);
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) { s_red[wave][0][j * 256 + lane * 4 + i] = dgam[j][i]; s_red[wave][1][j * 256 + lane * 4 + i] = dbet[j][i]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += LG_WAVES * WAVE) {
    const int which = i / D, d = i - which * D;  // (lanes with lane * 4 >= D wrote zeros beyond D: never read)
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < LG_WAVES; ++w) a += s_red[w][which][d];
    part[((size_t)blockIdx.x * 2 + which) * D + d] = a;
  }
}

// plain LayerNorm over the channel axis of a token-major map (VSSBlock.norm / norm2, vmamba.py:1190,1222): same wave-per-token
// scheme; reads and writes the activation dtype directly (autocast otherwise wraps the fp32 LayerNorm kernel in two casts)
template <typename T, int D>
__global__ __launch_bounds__(LG_WAVES* WAVE) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, T* __restrict__ out,
                                                                 float* __restrict__ stats, size_t ntok, float eps) {
  constexpr int NJ = (D + 255) / 256;
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  const bool act = lane * 4 < D;
  const size_t tok = (size_t)blockIdx.x * LG_WAVES + wave;
  if (tok >= ntok) return;
  float v[NJ][4];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0.f;
    if (act) ld4t<T>(x + tok * D + j * 256 + lane * 4, v[j]);
    sum += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
  }
  const float mean = wsum(sum) * (1.f / D);
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float c = act ? v[j][i] - mean : 0.f; sq = fmaf(c, c, sq); }
  const float rstd = rsqrtf(wsum(sq) * (1.f / D) + eps);
  if (lane == 0) { stats[2 * tok] = mean; stats[2 * tok + 1] = rstd; }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int d = j * 256 + lane * 4;
    if (!act) continue;
    float g[4], be[4], o[4];
    Elt<float>::ld4(gamma + d, g);
    Elt<float>::ld4(beta + d, be);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = fmaf((v[j][i] - mean) * rstd, g[i], be[i]);
    Elt<T>::st4(out + tok * D + d, o);
  }
}

template <typename T, int D>
__global__ __launch_bounds__(LG_WAVES* WAVE) void ln_bwd_kernel(const T* __restrict__ gout, const T* __restrict__ x,
                                                                 const float* __restrict__ gamma, const float* __restrict__ stats,
                                                                 T* __restrict__ gx, float* __restrict__ part, size_t ntok) {
  constexpr int NJ = (D + 255) / 256;
  __shared__ float s_red[LG_WAVES][2][NJ * 256];
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  const bool act = lane * 4 < D;
  float dgam[NJ][4], dbet[NJ][4], gm[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { dgam[j][i] = 0.f; dbet[j][i] = 0.f; gm[j][i] = 0.f; }
    if (act) Elt<float>::ld4(gamma + j * 256 + lane * 4, gm[j]);
  }
  const size_t tok0 = ((size_t)blockIdx.x * LG_WAVES + wave) * LG_TOK_BWD;
  for (int t = 0; t < LG_TOK_BWD; ++t) {
    const size_t tok = tok0 + t;
    if (tok >= ntok) break;  // wave-uniform
    const float mean = stats[2 * tok], rstd = stats[2 * tok + 1];
    float xh[NJ][4], gxh[NJ][4];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int d = j * 256 + lane * 4;
      float xv[4] = {mean, mean, mean, mean}, g[4] = {0.f, 0.f, 0.f, 0.f};
      if (act) {
        ld4t<T>(x + tok * D + d, xv);
        ld4t<T>(gout + tok * D + d, g);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xh[j][i] = (xv[i] - mean) * rstd;
        dgam[j][i] = fmaf(g[i], xh[j][i], dgam[j][i]);
        dbet[j][i] += g[i];
        gxh[j][i] = g[i] * gm[j][i];
        c1 += gxh[j][i];
        c2 = fmaf(gxh[j][i], xh[j][i], c2);
      }
    }
    c1 = wsum(c1) * (1.f / D);
    c2 = wsum(c2) * (1.f / D);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = rstd * (gxh[j][i] - c1 - xh[j][i] * c2);
      if (act) Elt<T>::st4(gx + tok * D + j * 256 + lane * 4, o);
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) { s_red[wave][0][j * 256 + lane * 4 + i] = dgam[j][i]; s_red[wave][1][j * 256 + lane * 4 + i] = dbet[j][i]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += LG_WAVES * WAVE) {
    const int which = i / D, d = i - which * D;
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < LG_WAVES; ++w) a += s_red[w][which][d];
    part[((size_t)blockIdx.x * 2 + which) * D + d] = a;
  }
}

// ---- LayerNorm over SHORT rows of bf16 (D = 64 | 128: VSSBlock.norm / norm2 at MEH level 0, hidden 128 - 409 600 tokens per pass at 640 px).
// One wave per token leaves half the lanes idle at D = 128 and moves 8 bytes per lane: 105 us forward / 107 us backward for 210 MB (2 TB/s).
// Here a token is LPT = D / 8 lanes x 16 bytes and a wave holds TPW = 64 / LPT tokens at a time; sums are butterflies over the LPT lanes.
// Same partial-sum layout as ln_bwd_kernel (64 tokens per workgroup), so the callers do not change.
template <int D>
__global__ __launch_bounds__(LG_WAVES* WAVE) void ln_fwd_narrow_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                                        const float* __restrict__ beta, bf16_t* __restrict__ out,
                                                                        float* __restrict__ stats, size_t ntok, float eps) {
  constexpr int LPT = D / 8, TPW = WAVE / LPT;
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE, slot = lane / LPT, l = lane % LPT;
  const size_t tok = ((size_t)blockIdx.x * LG_WAVES + wave) * TPW + slot;
  const size_t tc = tok < ntok ? tok : ntok - 1;     // (clamped: lanes past the end take part in the butterflies, store nothing)
  float v[8], g[8], be[8];
  Vec8::ld(x + tc * D + l * 8, v);
  Elt<float>::ld4(gamma + l * 8, *reinterpret_cast<float(*)[4]>(g)); Elt<float>::ld4(gamma + l * 8 + 4, *reinterpret_cast<float(*)[4]>(g + 4));
  Elt<float>::ld4(beta + l * 8, *reinterpret_cast<float(*)[4]>(be)); Elt<float>::ld4(beta + l * 8 + 4, *reinterpret_cast<float(*)[4]>(be + 4));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) sum += v[i];
  const float mean = group_sum<LPT>(sum) * (1.f / D);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { const float c = v[i] - mean; sq = fmaf(c, c, sq); }
  const float rstd = rsqrtf(group_sum<LPT>(sq) * (1.f / D) + eps);
  if (tok >= ntok) return;
  if (l == 0) { stats[2 * tok] = mean; stats[2 * tok + 1] = rstd; }
  float o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = fmaf((v[i] - mean) * rstd, g[i], be[i]);
  Vec8::st(out + tok * D + l * 8, o);
}

template <int D>
__global__ __launch_bounds__(LG_WAVES* WAVE) void ln_bwd_narrow_kernel(const bf16_t* __restrict__ gout, const bf16_t* __restrict__ x,
                                                                        const float* __restrict__ gamma, const float* __restrict__ stats,
                                                                        bf16_t* __restrict__ gx, float* __restrict__ part, size_t ntok) {
  constexpr int LPT = D / 8, TPW = WAVE / LPT;
  __shared__ float s_red[LG_WAVES][2][D];
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE, slot = lane / LPT, l = lane % LPT;
  float gm[8], dgam[8], dbet[8];
  Elt<float>::ld4(gamma + l * 8, *reinterpret_cast<float(*)[4]>(gm)); Elt<float>::ld4(gamma + l * 8 + 4, *reinterpret_cast<float(*)[4]>(gm + 4));
#pragma unroll
  for (int i = 0; i < 8; ++i) { dgam[i] = 0.f; dbet[i] = 0.f; }
  const size_t tok0 = ((size_t)blockIdx.x * LG_WAVES + wave) * LG_TOK_BWD;
  static_assert(LG_TOK_BWD % TPW == 0, "whole groups of tokens per wave");
#pragma unroll 2
  for (int t = 0; t < LG_TOK_BWD; t += TPW) {
    const size_t tok = tok0 + t + slot;
    const bool live = tok < ntok;
    const size_t tc = live ? tok : ntok - 1;
    const float mean = stats[2 * tc], rstd = stats[2 * tc + 1];
    float xv[8], g[8], xh[8], gxh[8];
    Vec8::ld(x + tc * D + l * 8, xv);
    Vec8::ld(gout + tc * D + l * 8, g);
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (!live) g[i] = 0.f;
      xh[i] = (xv[i] - mean) * rstd;
      dgam[i] = fmaf(g[i], xh[i], dgam[i]);
      dbet[i] += g[i];
      gxh[i] = g[i] * gm[i];
      c1 += gxh[i];
      c2 = fmaf(gxh[i], xh[i], c2);
    }
    c1 = group_sum<LPT>(c1) * (1.f / D);
    c2 = group_sum<LPT>(c2) * (1.f / D);
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = rstd * (gxh[i] - c1 - xh[i] * c2);
    if (live) Vec8::st(gx + tok * D + l * 8, o);
  }
  // the TPW token slots of the wave hold sums for the same channels: add them (fixed order), slot 0 publishes
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int o = LPT; o < WAVE; o <<= 1) { dgam[i] += __shfl_xor(dgam[i], o, WAVE); dbet[i] += __shfl_xor(dbet[i], o, WAVE); }
  if (slot == 0)
#pragma unroll
    for (int i = 0; i < 8; ++i) { s_red[wave][0][l * 8 + i] = dgam[i]; s_red[wave][1][l * 8 + i] = dbet[i]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += LG_WAVES * WAVE) {
    const int which = i / D, d = i - which * D;
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < LG_WAVES; ++w) a += s_red[w][which][d];
    part[((size_t)blockIdx.x * 2 + which) * D + d] = a;
  }
}

}  // namespace

extern "C" int tamtr_cross_merge_fwd(const void* y4, float* ymT, int B, int D, int H, int W, int plane_dtype, void* stream) {
  if (!y4 || !ymT || B <= 0 || D <= 0 || H <= 0 || W <= 0 || (plane_dtype != TAMTR_F32 && plane_dtype != TAMTR_BF16)) return TAMTR_EINVAL;
  if (D % CB || B > 65535 || D / CB > 65535) return TAMTR_EUNSUP;
  const int tiles_w = (W + TS - 1) / TS, tiles_h = (H + TS - 1) / TS;
  const dim3 grid(tiles_w * tiles_h, D / CB, B);
  if (plane_dtype == TAMTR_F32)
    hipLaunchKernelGGL(cross_merge_fwd_kernel<float>, grid, dim3(MT_THREADS), 0, (hipStream_t)stream, (const float*)y4, ymT, D, H, W, tiles_w);
  else if (H % 2 == 0 && W % 2 == 0 && (uintptr_t)y4 % 4 == 0 && (uintptr_t)ymT % 16 == 0) {
    const int t2w = (W + T2 - 1) / T2, t2h = (H + T2 - 1) / T2;
    const size_t dyn = (size_t)C2B * PL2 * sizeof(float);
    if (hipFuncSetAttribute((const void*)cross_merge_fwd16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) return TAMTR_ELAUNCH;
    hipLaunchKernelGGL(cross_merge_fwd16_kernel, dim3(t2w * t2h, D / C2B, B), dim3(MT_THREADS), dyn, (hipStream_t)stream, (const bf16_t*)y4, ymT, D, H,
                       W, t2w);
  } else
    hipLaunchKernelGGL(cross_merge_fwd_kernel<bf16_t>, grid, dim3(MT_THREADS), 0, (hipStream_t)stream, (const bf16_t*)y4, ymT, D, H, W, tiles_w);
  return tamtr_launch_status();
}

extern "C" int tamtr_cross_merge_bwd(const float* gymT, void* g2, int B, int D, int H, int W, int plane_dtype, void* stream) {
  if (!gymT || !g2 || B <= 0 || D <= 0 || H <= 0 || W <= 0 || (plane_dtype != TAMTR_F32 && plane_dtype != TAMTR_BF16)) return TAMTR_EINVAL;
  if (D % CB || B > 65535 || D / CB > 65535) return TAMTR_EUNSUP;
  const int tiles_w = (W + TS - 1) / TS, tiles_h = (H + TS - 1) / TS;
  const dim3 grid(tiles_w * tiles_h, D / CB, B);
  if (plane_dtype == TAMTR_F32)
    hipLaunchKernelGGL(cross_merge_bwd_kernel<float>, grid, dim3(MT_THREADS), 0, (hipStream_t)stream, gymT, (float*)g2, D, H, W, tiles_w);
  else if (H % 2 == 0 && W % 2 == 0 && (uintptr_t)g2 % 4 == 0 && (uintptr_t)gymT % 16 == 0 &&
           (long long)((W + T2 - 1) / T2) * ((H + T2 - 1) / T2) * T2 * T2 * 4 <= (long long)H * W * 5) {
    // (the 32 x 32 tiles where they cover the map with at most 25 % overhang: 160^2 189 against 338 us; 80^2 = 2.5 tiles a side 148 / 152 us;
    // 40^2 95 against 64 us - profiles/r04_ss2d_bf16_planes.txt)
    const int t2w = (W + T2 - 1) / T2, t2h = (H + T2 - 1) / T2;
    const size_t dyn = (size_t)C2B * PL2 * sizeof(float);
    if (hipFuncSetAttribute((const void*)cross_merge_bwd16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) return TAMTR_ELAUNCH;
    hipLaunchKernelGGL(cross_merge_bwd16_kernel, dim3(t2w * t2h, D / C2B, B), dim3(MT_THREADS), dyn, (hipStream_t)stream, gymT, (bf16_t*)g2, D, H, W,
                       t2w);
  } else
    hipLaunchKernelGGL(cross_merge_bwd_kernel<bf16_t>, grid, dim3(MT_THREADS), 0, (hipStream_t)stream, gymT, (bf16_t*)g2, D, H, W, tiles_w);
  return tamtr_launch_status();
}

static int ln_gate_check(const void* a, const void* b, const void* c, long long ntok, int D, long long zs, int dtype) {
  if (!a || !b || !c || ntok <= 0 || D <= 0 || zs < 2LL * D) return TAMTR_EINVAL;
  if (dtype != TAMTR_F32 && dtype != TAMTR_BF16) return TAMTR_EINVAL;
  if ((D != 64 && D != 128 && D != 256 && D != 512 && D != 1024) || zs % 4) return TAMTR_EUNSUP;
  return TAMTR_OK;
}

extern "C" int tamtr_ln_gate_blocks(long long ntok) { return (int)((ntok + LG_WAVES * LG_TOK_BWD - 1) / (LG_WAVES * LG_TOK_BWD)); }

#define LG_DISPATCH(KERNEL, T, ...)                                                                      \
  switch (D) {                                                                                           \
    case 64: hipLaunchKernelGGL((KERNEL<T, 64>), grid, dim3(LG_WAVES * WAVE), 0, s, __VA_ARGS__); break;   \
    case 128: hipLaunchKernelGGL((KERNEL<T, 128>), grid, dim3(LG_WAVES * WAVE), 0, s, __VA_ARGS__); break; \
    case 256: hipLaunchKernelGGL((KERNEL<T, 256>), grid, dim3(LG_WAVES * WAVE), 0, s, __VA_ARGS__); break; \
    case 512: hipLaunchKernelGGL((KERNEL<T, 512>), grid, dim3(LG_WAVES * WAVE), 0, s, __VA_ARGS__); break; \
    default: hipLaunchKernelGGL((KERNEL<T, 1024>), grid, dim3(LG_WAVES * WAVE), 0, s, __VA_ARGS__); break; \
  }

extern "C" int tamtr_ln_gate_fwd(const float* x, const void* xz, long long xz_token_stride, const float* gamma, const float* beta,
                                 void* out, float* stats, long long ntok, int D, float eps, int dtype, void* stream) {
  const int rc = ln_gate_check(x, xz, out, ntok, D, xz_token_stride, dtype);
  if (rc) return rc;
  if (!gamma || !beta || !stats) return TAMTR_EINVAL;
  const dim3 grid((unsigned)((ntok + LG_WAVES - 1) / LG_WAVES));
  hipStream_t s = (hipStream_t)stream;
  const size_t zs = (size_t)xz_token_stride, nt = (size_t)ntok;
  if (dtype == TAMTR_F32) { LG_DISPATCH(ln_gate_fwd_kernel, float, x, (const float*)xz, zs, gamma, beta, (float*)out, stats, nt, eps) }
  else { LG_DISPATCH(ln_gate_fwd_kernel, bf16_t, x, (const bf16_t*)xz, zs, gamma, beta, (bf16_t*)out, stats, nt, eps) }
  return tamtr_launch_status();
}

extern "C" int tamtr_ln_gate_bwd(const void* gout, const float* x, const void* xz, long long xz_token_stride, const float* gamma,
                                 const float* beta, const float* stats, float* gx, void* gxz, float* partials, long long ntok, int D,
                                 int dtype, void* stream) {
  const int rc = ln_gate_check(gout, x, xz, ntok, D, xz_token_stride, dtype);
  if (rc) return rc;
  if (!gamma || !beta || !stats || !gx || !gxz || !partials) return TAMTR_EINVAL;
  const dim3 grid((unsigned)tamtr_ln_gate_blocks(ntok));
  hipStream_t s = (hipStream_t)stream;
  const size_t zs = (size_t)xz_token_stride, nt = (size_t)ntok;
  if (dtype == TAMTR_F32) {
    LG_DISPATCH(ln_gate_bwd_kernel, float, (const float*)gout, x, (const float*)xz, zs, gamma, beta, stats, gx, (float*)gxz, partials, nt)
  } else {
    LG_DISPATCH(ln_gate_bwd_kernel, bf16_t, (const bf16_t*)gout, x, (const bf16_t*)xz, zs, gamma, beta, stats, gx, (bf16_t*)gxz, partials, nt)
  }
  return tamtr_launch_status();
}

#define LN_DISPATCH(KERNEL, T, ...)                                                                      \
  switch (D) {                                                                                           \
    case 32: hipLaunchKernelGGL((KERNEL<T, 32>), grid, dim3(LG_WAVES * WAVE), 0, s, __VA_ARGS__); break;   \
    case 64: hipLaunchKernelGGL((KERNEL<T, 64>), grid, dim3(LG_WAVES * WAVE), 0, s, __VA_ARGS__); break;   \
    case 128: hipLaunchKernelGGL((KERNEL<T, 128>), grid, dim3(LG_WAVES * WAVE), 0, s, __VA_ARGS__); break; \
    case 256: hipLaunchKernelGGL((KERNEL<T, 256>), grid, dim3(LG_WAVES * WAVE), 0, s, __VA_ARGS__); break; \
    case 512: hipLaunchKernelGGL((KERNEL<T, 512>), grid, dim3(LG_WAVES * WAVE), 0, s, __VA_ARGS__); break; \
    default: hipLaunchKernelGGL((KERNEL<T, 1024>), grid, dim3(LG_WAVES * WAVE), 0, s, __VA_ARGS__); break; \
  }

static int ln_check(const void* a, const void* b, const void* c, long long ntok, int D, int dtype) {
  if (!a || !b || !c || ntok <= 0 || D <= 0) return TAMTR_EINVAL;
  if (dtype != TAMTR_F32 && dtype != TAMTR_BF16) return TAMTR_EINVAL;
  if (D != 32 && D != 64 && D != 128 && D != 256 && D != 512 && D != 1024) return TAMTR_EUNSUP;
  return TAMTR_OK;
}

extern "C" int tamtr_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* out, float* stats, long long ntok, int D,
                                   float eps, int dtype, void* stream) {
  const int rc = ln_check(x, gamma, out, ntok, D, dtype);
  if (rc) return rc;
  if (!beta || !stats) return TAMTR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const size_t nt = (size_t)ntok;
  if (dtype == TAMTR_BF16 && (D == 64 || D == 128) && ((uintptr_t)x | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) % 16 == 0) {   // short rows: several tokens per wave
    const int tpw = WAVE / (D / 8);
    const dim3 g2((unsigned)((ntok + LG_WAVES * tpw - 1) / (LG_WAVES * tpw)));
    if (D == 64) hipLaunchKernelGGL(ln_fwd_narrow_kernel<64>, g2, dim3(LG_WAVES * WAVE), 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)out, stats, nt, eps);
    else hipLaunchKernelGGL(ln_fwd_narrow_kernel<128>, g2, dim3(LG_WAVES * WAVE), 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)out, stats, nt, eps);
    return tamtr_launch_status();
  }
  const dim3 grid((unsigned)((ntok + LG_WAVES - 1) / LG_WAVES));
  if (dtype == TAMTR_F32) { LN_DISPATCH(ln_fwd_kernel, float, (const float*)x, gamma, beta, (float*)out, stats, nt, eps) }
  else { LN_DISPATCH(ln_fwd_kernel, bf16_t, (const bf16_t*)x, gamma, beta, (bf16_t*)out, stats, nt, eps) }
  return tamtr_launch_status();
}

extern "C" int tamtr_layernorm_bwd(const void* gout, const void* x, const float* gamma, const float* stats, void* gx, float* partials,
                                   long long ntok, int D, int dtype, void* stream) {
  const int rc = ln_check(gout, x, gamma, ntok, D, dtype);
  if (rc) return rc;
  if (!stats || !gx || !partials) return TAMTR_EINVAL;
  const dim3 grid((unsigned)tamtr_ln_gate_blocks(ntok));
  hipStream_t s = (hipStream_t)stream;
  const size_t nt = (size_t)ntok;
  if (dtype == TAMTR_BF16 && (D == 64 || D == 128) && ((uintptr_t)x | (uintptr_t)gout | (uintptr_t)gx | (uintptr_t)gamma) % 16 == 0) {
    if (D == 64) hipLaunchKernelGGL(ln_bwd_narrow_kernel<64>, grid, dim3(LG_WAVES * WAVE), 0, s, (const bf16_t*)gout, (const bf16_t*)x, gamma, stats, (bf16_t*)gx, partials, nt);
    else hipLaunchKernelGGL(ln_bwd_narrow_kernel<128>, grid, dim3(LG_WAVES * WAVE), 0, s, (const bf16_t*)gout, (const bf16_t*)x, gamma, stats, (bf16_t*)gx, partials, nt);
    return tamtr_launch_status();
  }
  if (dtype == TAMTR_F32) { LN_DISPATCH(ln_bwd_kernel, float, (const float*)gout, (const float*)x, gamma, stats, (float*)gx, partials, nt) }
  else { LN_DISPATCH(ln_bwd_kernel, bf16_t, (const bf16_t*)gout, (const bf16_t*)x, gamma, stats, (bf16_t*)gx, partials, nt) }
  return tamtr_launch_status();
}
