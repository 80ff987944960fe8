// cpam.hip - fused CPAM gates (BTA-PAN), forward + backward, gfx950.
//
// Reference: CPAM.forward, ultralytics/nn/extra_modules/block.py:271-308
//     c   = sigmoid( bilinear_x2( maxpool3x3/s2/p1(x) ) ) * x                      (ChannelAttentionModule, per channel)
//     out = cat_g( sigmoid( max_{ch in chunk g} c ) * c_g ),  8 channel chunks      (SpatialAttentionModule per chunk)
// which PyTorch runs as 7 kernels forward and ~12 backward, each a full pass over the feature map (the fp32 bilinear
// kernel alone took 1.5 ms per site).  The op is parameter-free and purely HBM-bound, so the whole chain after the max-pool is
// ONE kernel here: x is read once (twice when a chunk has more than 16 channels: the second read is a cache hit), out is
// written once, the 1/4-size pooled map p is the only intermediate that exists in memory.
//
// Mapping: a thread owns a 2x2 pixel block (2k..2k+1, 2l..2l+1) of one (image, channel chunk) - exactly the pixels whose
// bilinear taps are the 3x3 pooled cells around (k, l) - and walks the chunk's channels; lanes run along l so x / out rows
// are contiguous 4-B (bf16) or 8-B (f32) per-lane streams.  The channel max / argmax never leaves the lane.
// Backward: kernel 1 re-derives c, writes the direct part of dx and du = dL/d(upsampled map); kernel 2 gathers du into
// dL/dp (each pooled cell is touched by a 4x4 pixel patch with weights {.25,.75,.75,.25}, clamped at the borders as
// upsample_bilinear2d(align_corners=False) does); the max-pool scatter stays with the caller (it owns the pool indices).
#include "common.h"

namespace {

constexpr int CPAM_THREADS = 256;
constexpr int CHUNKS = 8;  // c.chunk(8, 1) in the reference

template <typename T>
__device__ __forceinline__ void ld2(const T* p, float& a, float& b);
template <>
__device__ __forceinline__ void ld2<float>(const float* p, float& a, float& b) {
  const float2 v = *reinterpret_cast<const float2*>(p);
  a = v.x; b = v.y;
}
template <>
__device__ __forceinline__ void ld2<bf16_t>(const bf16_t* p, float& a, float& b) {
  const uint32_t v = *reinterpret_cast<const uint32_t*>(p);
  a = __uint_as_float(v << 16); b = __uint_as_float(v & 0xffff0000u);
}
template <typename T>
__device__ __forceinline__ void st2(T* p, float a, float b);
template <>
__device__ __forceinline__ void st2<float>(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }
template <>
__device__ __forceinline__ void st2<bf16_t>(bf16_t* p, float a, float b) {
  *reinterpret_cast<uint32_t*>(p) = (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
}

__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + __expf(-v)); }

// geometry of one 2x2 pixel block: pooled rows / columns it taps and the bilinear weights of its first row / column
struct Taps {
  int rA, rB, rC, cA, cB, cC;
  float wt0, wt1, wl0, wl1;  // pixel row 2k: wt0 * p[rA] + wt1 * p[rB]; pixel column 2l: wl0 * p[.][cA] + wl1 * p[.][cB]
};
__device__ __forceinline__ Taps make_taps(int k, int l, int Hp, int Wp) {
  Taps t;
  t.rA = max(k - 1, 0); t.rB = k; t.rC = min(k + 1, Hp - 1);
  t.cA = max(l - 1, 0); t.cB = l; t.cC = min(l + 1, Wp - 1);
  // source coordinate of pixel 2k is k - 0.25, clamped to 0 for k == 0 (upsample_bilinear2d, align_corners=False)
  t.wt0 = k == 0 ? 0.f : 0.25f; t.wt1 = k == 0 ? 1.f : 0.75f;
  t.wl0 = l == 0 ? 0.f : 0.25f; t.wl1 = l == 0 ? 1.f : 0.75f;
  return t;
}

// upsampled values of the block's 4 pixels for one channel (pc = that channel's pooled map), PyTorch's association order
template <typename T>
__device__ __forceinline__ void upsample4(const T* __restrict__ pc, const Taps& t, int Wp, float (&u)[4]) {
  const T* ra = pc + (size_t)t.rA * Wp;
  const T* rb = pc + (size_t)t.rB * Wp;
  const T* rc = pc + (size_t)t.rC * Wp;
  const float aA = Elt<T>::ld(ra + t.cA), aB = Elt<T>::ld(ra + t.cB), aC = Elt<T>::ld(ra + t.cC);
  const float bA = Elt<T>::ld(rb + t.cA), bB = Elt<T>::ld(rb + t.cB), bC = Elt<T>::ld(rb + t.cC);
  const float cA = Elt<T>::ld(rc + t.cA), cB = Elt<T>::ld(rc + t.cB), cC = Elt<T>::ld(rc + t.cC);
  const float hla = t.wl0 * aA + t.wl1 * aB, hra = 0.75f * aB + 0.25f * aC;
  const float hlb = t.wl0 * bA + t.wl1 * bB, hrb = 0.75f * bB + 0.25f * bC;
  const float hlc = t.wl0 * cA + t.wl1 * cB, hrc = 0.75f * cB + 0.25f * cC;
  u[0] = t.wt0 * hla + t.wt1 * hlb; u[1] = t.wt0 * hra + t.wt1 * hrb;
  u[2] = 0.75f * hlb + 0.25f * hlc; u[3] = 0.75f * hrb + 0.25f * hrc;
}

// KEEP > 0: chunks of at most KEEP channels keep their gated values in registers (single pass over x)
template <typename T, int KEEP>
__global__ __launch_bounds__(CPAM_THREADS) void cpam_fwd_kernel(const T* __restrict__ x, const T* __restrict__ p, T* __restrict__ out,
                                                                 float* __restrict__ s2_out, int32_t* __restrict__ arg_out, int C,
                                                                 int H, int W) {
  const int Hp = H / 2, Wp = W / 2, Cg = C / CHUNKS;
  const int q = blockIdx.x * CPAM_THREADS + threadIdx.x;
  if (q >= Hp * Wp) return;
  const int k = q / Wp, l = q - k * Wp, g = blockIdx.y, b = blockIdx.z;
  const Taps t = make_taps(k, l, Hp, Wp);
  const size_t c0 = (size_t)b * C + (size_t)g * Cg;
  const T* xb = x + c0 * H * W + (size_t)(2 * k) * W + 2 * l;
  T* ob = out + c0 * H * W + (size_t)(2 * k) * W + 2 * l;
  const T* pb = p + c0 * Hp * Wp;
  float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int am[4] = {0, 0, 0, 0};
  float keep[KEEP > 0 ? KEEP : 1][4];

  auto gated = [&](int c, float (&cv)[4]) {
    float u[4], xv[4];
    upsample4<T>(pb + (size_t)c * Hp * Wp, t, Wp, u);
    ld2<T>(xb + (size_t)c * H * W, xv[0], xv[1]);
    ld2<T>(xb + (size_t)c * H * W + W, xv[2], xv[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) cv[i] = sigm(u[i]) * xv[i];
  };

  if constexpr (KEEP > 0) {
#pragma unroll
    for (int c = 0; c < KEEP; ++c) {
      if (c < Cg) {
        gated(c, keep[c]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (keep[c][i] > m[i]) { m[i] = keep[c][i]; am[i] = c; }
      }
    }
  } else {
    for (int c = 0; c < Cg; ++c) {
      float cv[4];
      gated(c, cv);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (cv[i] > m[i]) { m[i] = cv[i]; am[i] = c; }
    }
  }
  float s2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) s2[i] = sigm(m[i]);
  const size_t sp = ((size_t)b * CHUNKS + g) * H * W + (size_t)(2 * k) * W + 2 * l;
  *reinterpret_cast<float2*>(s2_out + sp) = make_float2(s2[0], s2[1]);
  *reinterpret_cast<float2*>(s2_out + sp + W) = make_float2(s2[2], s2[3]);
  *reinterpret_cast<int2*>(arg_out + sp) = make_int2(am[0], am[1]);
  *reinterpret_cast<int2*>(arg_out + sp + W) = make_int2(am[2], am[3]);
  if constexpr (KEEP > 0) {
#pragma unroll
    for (int c = 0; c < KEEP; ++c) {
      if (c < Cg) {
        st2<T>(ob + (size_t)c * H * W, s2[0] * keep[c][0], s2[1] * keep[c][1]);
        st2<T>(ob + (size_t)c * H * W + W, s2[2] * keep[c][2], s2[3] * keep[c][3]);
      }
    }
  } else {
    for (int c = 0; c < Cg; ++c) {
      float cv[4];
      gated(c, cv);
      st2<T>(ob + (size_t)c * H * W, s2[0] * cv[0], s2[1] * cv[1]);
      st2<T>(ob + (size_t)c * H * W + W, s2[2] * cv[2], s2[3] * cv[3]);
    }
  }
}

// backward 1: dxd = dL/dx through the product c = s1 * x (the pooled path is added by the caller), du = dL/du
template <typename T>
__global__ __launch_bounds__(CPAM_THREADS) void cpam_bwd_kernel(const T* __restrict__ gout, const T* __restrict__ x,
                                                                 const T* __restrict__ p, const float* __restrict__ s2_in,
                                                                 const int32_t* __restrict__ arg_in, T* __restrict__ dxd,
                                                                 T* __restrict__ du, int C, int H, int W) {
  const int Hp = H / 2, Wp = W / 2, Cg = C / CHUNKS;
  const int q = blockIdx.x * CPAM_THREADS + threadIdx.x;
  if (q >= Hp * Wp) return;
  const int k = q / Wp, l = q - k * Wp, g = blockIdx.y, b = blockIdx.z;
  const Taps t = make_taps(k, l, Hp, Wp);
  const size_t c0 = (size_t)b * C + (size_t)g * Cg;
  const size_t pix = (size_t)(2 * k) * W + 2 * l;
  const T* xb = x + c0 * H * W + pix;
  const T* gb = gout + c0 * H * W + pix;
  const T* pb = p + c0 * Hp * Wp;
  const size_t sp = ((size_t)b * CHUNKS + g) * H * W + pix;
  float s2[4];
  int am[4];
  {
    const float2 a = *reinterpret_cast<const float2*>(s2_in + sp), bb = *reinterpret_cast<const float2*>(s2_in + sp + W);
    const int2 ia = *reinterpret_cast<const int2*>(arg_in + sp), ib = *reinterpret_cast<const int2*>(arg_in + sp + W);
    s2[0] = a.x; s2[1] = a.y; s2[2] = bb.x; s2[3] = bb.y;
    am[0] = ia.x; am[1] = ia.y; am[2] = ib.x; am[3] = ib.y;
  }
  // S = sum_c gout_c * c_c  (what flows into the chunk max through sigmoid(max))
  float S[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < Cg; ++c) {
    float u[4], xv[4], go[4];
    upsample4<T>(pb + (size_t)c * Hp * Wp, t, Wp, u);
    ld2<T>(xb + (size_t)c * H * W, xv[0], xv[1]);
    ld2<T>(xb + (size_t)c * H * W + W, xv[2], xv[3]);
    ld2<T>(gb + (size_t)c * H * W, go[0], go[1]);
    ld2<T>(gb + (size_t)c * H * W + W, go[2], go[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) S[i] = fmaf(go[i], sigm(u[i]) * xv[i], S[i]);
  }
  float dm[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dm[i] = s2[i] * (1.f - s2[i]) * S[i];
  for (int c = 0; c < Cg; ++c) {
    float u[4], xv[4], go[4], dx[4], dd[4];
    upsample4<T>(pb + (size_t)c * Hp * Wp, t, Wp, u);
    ld2<T>(xb + (size_t)c * H * W, xv[0], xv[1]);
    ld2<T>(xb + (size_t)c * H * W + W, xv[2], xv[3]);
    ld2<T>(gb + (size_t)c * H * W, go[0], go[1]);
    ld2<T>(gb + (size_t)c * H * W + W, go[2], go[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float s1 = sigm(u[i]);
      const float dc = go[i] * s2[i] + (am[i] == c ? dm[i] : 0.f);
      dx[i] = dc * s1;
      dd[i] = dc * xv[i] * s1 * (1.f - s1);
    }
    const size_t o = (c0 + c) * H * W + pix;
    st2<T>(dxd + o, dx[0], dx[1]);
    st2<T>(dxd + o + W, dx[2], dx[3]);
    st2<T>(du + o, dd[0], dd[1]);
    st2<T>(du + o + W, dd[2], dd[3]);
  }
}

// backward 2: dp[k, l] = sum over the 4x4 pixel patch (2k-1..2k+2) x (2l-1..2l+2) of wy * wx * du
template <typename T>
__global__ __launch_bounds__(CPAM_THREADS) void cpam_dp_kernel(const T* __restrict__ du, T* __restrict__ dp, int H, int W) {
  const int Hp = H / 2, Wp = W / 2;
  const int q = blockIdx.x * CPAM_THREADS + threadIdx.x;
  if (q >= Hp * Wp) return;
  const int k = q / Wp, l = q - k * Wp;
  const size_t bc = blockIdx.y;  // flattened (image, channel)
  const T* d = du + bc * H * W;
  const float wy[4] = {k >= 1 ? 0.25f : 0.f, k == 0 ? 1.f : 0.75f, k == Hp - 1 ? 1.f : 0.75f, k < Hp - 1 ? 0.25f : 0.f};
  const float wx0 = l >= 1 ? 0.25f : 0.f, wx1 = l == 0 ? 1.f : 0.75f, wx2 = l == Wp - 1 ? 1.f : 0.75f, wx3 = l < Wp - 1 ? 0.25f : 0.f;
  const int xl = max(2 * l - 1, 0), xr = min(2 * l + 2, W - 1);
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = min(max(2 * k - 1 + i, 0), H - 1);  // clamped rows carry weight 0
    const T* r = d + (size_t)y * W;
    float m0, m1;
    ld2<T>(r + 2 * l, m0, m1);
    const float row = wx0 * Elt<T>::ld(r + xl) + wx1 * m0 + wx2 * m1 + wx3 * Elt<T>::ld(r + xr);
    acc = fmaf(wy[i], row, acc);
  }
  Elt<T>::st(dp + bc * Hp * Wp + q, acc);
}

// ------------------------------------------------------------------------------------------------ channels-last forms
// The trunk runs channels-last (model.py); the kernels above are NCHW and needed a transposing copy on the way in and out (16 launches,
// 0.39 ms per step at the four sites) and walk a chunk's channels one dependent round trip at a time (65 - 75 us per site whatever its
// size).  Here x / p / out are [B][H][W][C]: a lane owns a 2 x 2 pixel block x V consecutive channels (16 bytes: V = 8 bf16 | 4 f32), the
// C / V lanes of a block sit side by side, a chunk (C / 8 channels) is LPC = C / (8 V) neighbouring lanes, and the chunk max / argmax /
// sum is a butterfly over those lanes.  Thirteen 16-byte loads per lane (9 pooled taps + 4 pixels), all issued up front.
// s2 / arg: f32 / i32 [B][H][W][8] (chunk innermost).
template <typename T> struct VecT;
template <> struct VecT<float> { static constexpr int V = 4; };
template <> struct VecT<bf16_t> { static constexpr int V = 8; };

template <typename T, int V>
__device__ __forceinline__ void ldv(const T* p, float (&o)[V]) {
  if constexpr (V == 8) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  } else {
    const float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
}
template <typename T, int V>
__device__ __forceinline__ void stv(T* p, const float (&v)[V]) {
  if constexpr (V == 8) {
    uint4 o;
    o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16); o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    o.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16); o.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
    *reinterpret_cast<uint4*>(p) = o;
  } else {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// the block's geometry and its 13 vectors: pooled taps pt[3][3][V] (rows rA rB rC x columns cA cB cC) and pixels xv[4][V]
// (pixel order: (2k, 2l), (2k, 2l + 1), (2k + 1, 2l), (2k + 1, 2l + 1) - the order of upsample4 / the NCHW kernels)
template <typename T, int V>
__device__ __forceinline__ void cl_upsample(const float (&pt)[3][3][V], const Taps& t, float (&u)[4][V]) {
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const float hla = t.wl0 * pt[0][0][j] + t.wl1 * pt[0][1][j], hra = 0.75f * pt[0][1][j] + 0.25f * pt[0][2][j];
    const float hlb = t.wl0 * pt[1][0][j] + t.wl1 * pt[1][1][j], hrb = 0.75f * pt[1][1][j] + 0.25f * pt[1][2][j];
    const float hlc = t.wl0 * pt[2][0][j] + t.wl1 * pt[2][1][j], hrc = 0.75f * pt[2][1][j] + 0.25f * pt[2][2][j];
    u[0][j] = t.wt0 * hla + t.wt1 * hlb; u[1][j] = t.wt0 * hra + t.wt1 * hrb;
    u[2][j] = 0.75f * hlb + 0.25f * hlc; u[3][j] = 0.75f * hrb + 0.25f * hrc;
  }
}

template <typename T>
__global__ __launch_bounds__(CPAM_THREADS) void cpam_cl_fwd_kernel(const T* __restrict__ x, const T* __restrict__ p, T* __restrict__ out,
                                                                    float* __restrict__ s2_out, int32_t* __restrict__ arg_out, int B, int C,
                                                                    int H, int W) {
  constexpr int V = VecT<T>::V;
  const int Hp = H / 2, Wp = W / 2, LPB = C / V, LPC = LPB / CHUNKS;   // lanes per block / per chunk
  const int lb = threadIdx.x % LPB;                                    // this lane's channel vector inside the block
  const long long blk = (long long)blockIdx.x * (CPAM_THREADS / LPB) + threadIdx.x / LPB;
  const long long nblk = (long long)B * Hp * Wp;
  const long long q = blk < nblk ? blk : nblk - 1;                      // (clamped: a lane past the end computes the last block again, stores nothing)
  const int b = (int)(q / (Hp * Wp)), r = (int)(q - (long long)b * Hp * Wp), k = r / Wp, l = r - k * Wp;
  const Taps t = make_taps(k, l, Hp, Wp);
  const int ch0 = lb * V;
  const T* pb = p + (size_t)b * Hp * Wp * C + ch0;
  const T* xb = x + (((size_t)b * H + 2 * k) * W + 2 * l) * C + ch0;
  const int rows[3] = {t.rA, t.rB, t.rC}, cols[3] = {t.cA, t.cB, t.cC};
  float pt[3][3][V], xv[4][V];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) ldv<T, V>(pb + ((size_t)rows[a] * Wp + cols[c]) * C, pt[a][c]);
  ldv<T, V>(xb, xv[0]); ldv<T, V>(xb + C, xv[1]); ldv<T, V>(xb + (size_t)W * C, xv[2]); ldv<T, V>(xb + (size_t)W * C + C, xv[3]);
  float u[4][V];
  cl_upsample<T, V>(pt, t, u);
  float m[4];
  int am[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    m[i] = -INFINITY; am[i] = 0;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      u[i][j] = sigm(u[i][j]) * xv[i][j];                      // the channel-gated value c
      if (u[i][j] > m[i]) { m[i] = u[i][j]; am[i] = (lb % LPC) * V + j; }
    }
    for (int o = 1; o < LPC; o <<= 1) {                        // chunk max over its LPC lanes; ties: the lower channel (the NCHW kernel's first maximum)
      const float om = __shfl_xor(m[i], o, WAVE);
      const int oa = __shfl_xor(am[i], o, WAVE);
      if (om > m[i] || (om == m[i] && oa < am[i])) { m[i] = om; am[i] = oa; }
    }
  }
  if (blk >= nblk) return;
  float s2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) s2[i] = sigm(m[i]);
  T* ob = out + (((size_t)b * H + 2 * k) * W + 2 * l) * C + ch0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float o[V];
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = s2[i] * u[i][j];
    stv<T, V>(ob + ((size_t)(i >> 1) * W + (i & 1)) * C, o);
  }
  if (lb % LPC == 0) {
    const int g = lb / LPC;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t sp = (((size_t)b * H + 2 * k + (i >> 1)) * W + 2 * l + (i & 1)) * CHUNKS + g;
      s2_out[sp] = s2[i];
      arg_out[sp] = am[i];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(CPAM_THREADS) void cpam_cl_bwd_kernel(const T* __restrict__ gout, const T* __restrict__ x, const T* __restrict__ p,
                                                                    const float* __restrict__ s2_in, const int32_t* __restrict__ arg_in,
                                                                    T* __restrict__ dxd, T* __restrict__ du, int B, int C, int H, int W) {
  constexpr int V = VecT<T>::V;
  const int Hp = H / 2, Wp = W / 2, LPB = C / V, LPC = LPB / CHUNKS;
  const int lb = threadIdx.x % LPB;
  const long long blk = (long long)blockIdx.x * (CPAM_THREADS / LPB) + threadIdx.x / LPB;
  const long long nblk = (long long)B * Hp * Wp;
  const long long q = blk < nblk ? blk : nblk - 1;
  const int b = (int)(q / (Hp * Wp)), r = (int)(q - (long long)b * Hp * Wp), k = r / Wp, l = r - k * Wp;
  const Taps t = make_taps(k, l, Hp, Wp);
  const int ch0 = lb * V, g = lb / LPC;
  const T* pb = p + (size_t)b * Hp * Wp * C + ch0;
  const size_t pix = (((size_t)b * H + 2 * k) * W + 2 * l) * C + ch0;
  const int rows[3] = {t.rA, t.rB, t.rC}, cols[3] = {t.cA, t.cB, t.cC};
  float pt[3][3][V], xv[4][V], go[4][V], s2[4];
  int am[4];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) ldv<T, V>(pb + ((size_t)rows[a] * Wp + cols[c]) * C, pt[a][c]);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const size_t o = ((size_t)(i >> 1) * W + (i & 1)) * C;
    ldv<T, V>(x + pix + o, xv[i]);
    ldv<T, V>(gout + pix + o, go[i]);
    const size_t sp = (((size_t)b * H + 2 * k + (i >> 1)) * W + 2 * l + (i & 1)) * CHUNKS + g;
    s2[i] = s2_in[sp];
    am[i] = arg_in[sp];
  }
  float u[4][V];
  cl_upsample<T, V>(pt, t, u);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float S = 0.f;                                              // sum over the chunk of gout * c (what flows into the chunk max)
#pragma unroll
    for (int j = 0; j < V; ++j) {
      u[i][j] = sigm(u[i][j]);                                  // s1
      S = fmaf(go[i][j], u[i][j] * xv[i][j], S);
    }
    for (int o = 1; o < LPC; o <<= 1) S += __shfl_xor(S, o, WAVE);
    const float dm = s2[i] * (1.f - s2[i]) * S;
    float dx[V], dd[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float s1 = u[i][j];
      const float dc = go[i][j] * s2[i] + (am[i] == (lb % LPC) * V + j ? dm : 0.f);
      dx[j] = dc * s1;
      dd[j] = dc * xv[i][j] * s1 * (1.f - s1);
    }
    if (blk < nblk) {
      const size_t o = pix + ((size_t)(i >> 1) * W + (i & 1)) * C;
      stv<T, V>(dxd + o, dx);
      stv<T, V>(du + o, dd);
    }
  }
}

// dp[b, k, l, :] = sum over the 4 x 4 pixel patch (2k-1..2k+2) x (2l-1..2l+2) of wy * wx * du[b, y, x, :]
template <typename T>
__global__ __launch_bounds__(CPAM_THREADS) void cpam_cl_dp_kernel(const T* __restrict__ du, T* __restrict__ dp, int B, int C, int H, int W) {
  constexpr int V = VecT<T>::V;
  const int Hp = H / 2, Wp = W / 2, LPB = C / V;
  const long long id = (long long)blockIdx.x * CPAM_THREADS + threadIdx.x, total = (long long)B * Hp * Wp * LPB;
  if (id >= total) return;
  const int lb = (int)(id % LPB);
  const long long q = id / LPB;
  const int b = (int)(q / (Hp * Wp)), r = (int)(q - (long long)b * Hp * Wp), k = r / Wp, l = r - k * Wp;
  const float wy[4] = {k >= 1 ? 0.25f : 0.f, k == 0 ? 1.f : 0.75f, k == Hp - 1 ? 1.f : 0.75f, k < Hp - 1 ? 0.25f : 0.f};
  const float wx[4] = {l >= 1 ? 0.25f : 0.f, l == 0 ? 1.f : 0.75f, l == Wp - 1 ? 1.f : 0.75f, l < Wp - 1 ? 0.25f : 0.f};
  const int xs[4] = {max(2 * l - 1, 0), 2 * l, 2 * l + 1, min(2 * l + 2, W - 1)};
  const T* d = du + (size_t)b * H * W * C + lb * V;
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = min(max(2 * k - 1 + i, 0), H - 1);            // clamped rows / columns carry weight 0
    float v[4][V];
#pragma unroll
    for (int c = 0; c < 4; ++c) ldv<T, V>(d + ((size_t)y * W + xs[c]) * C, v[c]);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float row = wx[0] * v[0][j] + wx[1] * v[1][j] + wx[2] * v[2][j] + wx[3] * v[3][j];   // (the NCHW kernel's association)
      acc[j] = fmaf(wy[i], row, acc[j]);
    }
  }
  stv<T, V>(dp + (((size_t)b * Hp + k) * Wp + l) * C + lb * V, acc);
}

// ---- CPAM's max-pool (3 x 3, stride 2, padding 1; block.py:274) on channels-last maps, 16-byte channel vectors per lane.  Same rules as
// csrc/pool.hip (whose NHWC form handles one element per thread): the first maximum in row-major window order wins, NaN wins over everything;
// the winner's position inside the unclipped window is kept as one byte per element and the backward is a gather (no atomics).
template <typename T>
__global__ __launch_bounds__(CPAM_THREADS) void pool3s2_cl_fwd_kernel(const T* __restrict__ x, T* __restrict__ p, uint8_t* __restrict__ code, int B, int C,
                                                                       int H, int W) {
  constexpr int V = VecT<T>::V;
  const int Hp = H / 2, Wp = W / 2, LPB = C / V;
  const long long id = (long long)blockIdx.x * CPAM_THREADS + threadIdx.x, total = (long long)B * Hp * Wp * LPB;
  if (id >= total) return;
  const int lb = (int)(id % LPB);
  const long long q = id / LPB;
  const int b = (int)(q / (Hp * Wp)), r = (int)(q - (long long)b * Hp * Wp), k = r / Wp, l = r - k * Wp;
  const T* xb = x + (size_t)b * H * W * C + lb * V;
  float v[9][V];
#pragma unroll
  for (int dh = 0; dh < 3; ++dh)
#pragma unroll
    for (int dw = 0; dw < 3; ++dw)   // (clamped addresses: every load is issued; taps outside the image are skipped below.  H, W even: only row / column -1 can be outside)
      ldv<T, V>(xb + ((size_t)max(2 * k - 1 + dh, 0) * W + max(2 * l - 1 + dw, 0)) * C, v[dh * 3 + dw]);
  float best[V];
  int win[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { best[j] = -INFINITY; win[j] = (k == 0 ? 3 : 0) + (l == 0 ? 1 : 0); }
#pragma unroll
  for (int dh = 0; dh < 3; ++dh)
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) {
      const bool inside = (dh > 0 || k > 0) && (dw > 0 || l > 0);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float t = v[dh * 3 + dw][j];
        if (inside && (t > best[j] || t != t)) { best[j] = t; win[j] = dh * 3 + dw; }
      }
    }
  const size_t o = (((size_t)b * Hp + k) * Wp + l) * C + lb * V;
  stv<T, V>(p + o, best);
  if constexpr (V == 8) {
    *reinterpret_cast<uint2*>(code + o) = make_uint2((uint32_t)win[0] | (win[1] << 8) | (win[2] << 16) | (win[3] << 24),
                                                     (uint32_t)win[4] | (win[5] << 8) | (win[6] << 16) | (win[7] << 24));
  } else {
    *reinterpret_cast<uint32_t*>(code + o) = (uint32_t)win[0] | (win[1] << 8) | (win[2] << 16) | (win[3] << 24);
  }
}

// gx[b, y, x, :] = addend + sum over the <= 2 x 2 windows that contain (y, x) of [code == my position] * gy
template <typename T>
__global__ __launch_bounds__(CPAM_THREADS) void pool3s2_cl_bwd_kernel(const T* __restrict__ gy, const uint8_t* __restrict__ code, const T* __restrict__ addend,
                                                                       T* __restrict__ gx, int B, int C, int H, int W) {
  constexpr int V = VecT<T>::V;
  const int Hp = H / 2, Wp = W / 2, LPB = C / V;
  const long long id = (long long)blockIdx.x * CPAM_THREADS + threadIdx.x, total = (long long)B * H * W * LPB;
  if (id >= total) return;
  const int lb = (int)(id % LPB);
  const long long q = id / LPB;
  const int b = (int)(q / (H * W)), r = (int)(q - (long long)b * H * W), y = r / W, xx = r - y * W;
  // window rows containing y: y even -> {y / 2}; y odd -> {(y - 1) / 2, (y + 1) / 2} (the second only if it exists)
  const int oh0 = y >> 1, oh1 = (y & 1) ? oh0 + 1 : -1, ow0 = xx >> 1, ow1 = (xx & 1) ? ow0 + 1 : -1;
  const int ohs[2] = {oh0, oh1}, ows[2] = {ow0, ow1};
  float acc[V], g[4][V];
  ldv<T, V>(addend + (size_t)q * C + lb * V, acc);
  uint32_t cd[4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const size_t o = (((size_t)b * Hp + min(max(ohs[a], 0), Hp - 1)) * Wp + min(max(ows[c], 0), Wp - 1)) * C + lb * V;
      ldv<T, V>(gy + o, g[a * 2 + c]);
      if constexpr (V == 8) { const uint2 t = *reinterpret_cast<const uint2*>(code + o); cd[a * 2 + c][0] = t.x; cd[a * 2 + c][1] = t.y; }
      else { cd[a * 2 + c][0] = *reinterpret_cast<const uint32_t*>(code + o); cd[a * 2 + c][1] = 0; }
    }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const bool live = ohs[a] >= 0 && ohs[a] < Hp && ows[c] >= 0 && ows[c] < Wp;
      const int mine = (y - (2 * ohs[a] - 1)) * 3 + (xx - (2 * ows[c] - 1));
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const int cj = (cd[a * 2 + c][j >> 2] >> (8 * (j & 3))) & 0xff;
        acc[j] += (live && cj == mine) ? g[a * 2 + c][j] : 0.f;
      }
    }
  stv<T, V>(gx + (size_t)q * C + lb * V, acc);
}

}  // namespace

static int cpam_check(const void* a, const void* b, const void* c, int B, int C, int H, int W, int dtype) {
  if (!a || !b || !c || B <= 0 || C <= 0 || H <= 0 || W <= 0) return TAMTR_EINVAL;
  if (dtype != TAMTR_F32 && dtype != TAMTR_BF16) return TAMTR_EINVAL;
  // odd maps make the reference itself fail (pool -> x2 upsample no longer matches x); 8 equal channel chunks as chunk(8, 1)
  if ((H & 1) || (W & 1) || C % CHUNKS || B > 65535) return TAMTR_EUNSUP;
  return TAMTR_OK;
}

extern "C" int tamtr_cpam_fwd(const void* x, const void* p, void* out, float* s2, int32_t* arg, int B, int C, int H, int W, int dtype,
                              void* stream) {
  const int rc = cpam_check(x, p, out, B, C, H, W, dtype);
  if (rc) return rc;
  if (!s2 || !arg) return TAMTR_EINVAL;
  const dim3 grid(((H / 2) * (W / 2) + CPAM_THREADS - 1) / CPAM_THREADS, CHUNKS, B);
  hipStream_t s = (hipStream_t)stream;
  const bool keep = C / CHUNKS <= 16;
#define CPAM_FWD(T, KEEP) \
  hipLaunchKernelGGL((cpam_fwd_kernel<T, KEEP>), grid, dim3(CPAM_THREADS), 0, s, (const T*)x, (const T*)p, (T*)out, s2, arg, C, H, W)
  if (dtype == TAMTR_F32) { if (keep) CPAM_FWD(float, 16); else CPAM_FWD(float, 0); }
  else { if (keep) CPAM_FWD(bf16_t, 16); else CPAM_FWD(bf16_t, 0); }
#undef CPAM_FWD
  return tamtr_launch_status();
}

extern "C" int tamtr_cpam_bwd(const void* gout, const void* x, const void* p, const float* s2, const int32_t* arg, void* dx_direct,
                              void* du_ws, void* dp, int B, int C, int H, int W, int dtype, void* stream) {
  const int rc = cpam_check(gout, x, p, B, C, H, W, dtype);
  if (rc) return rc;
  if (!s2 || !arg || !dx_direct || !du_ws || !dp) return TAMTR_EINVAL;
  if ((long long)B * C > 65535) return TAMTR_EUNSUP;  // grid.y of the dp kernel
  const int nb = ((H / 2) * (W / 2) + CPAM_THREADS - 1) / CPAM_THREADS;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TAMTR_F32) {
    hipLaunchKernelGGL(cpam_bwd_kernel<float>, dim3(nb, CHUNKS, B), dim3(CPAM_THREADS), 0, s, (const float*)gout, (const float*)x,
                       (const float*)p, s2, arg, (float*)dx_direct, (float*)du_ws, C, H, W);
    hipLaunchKernelGGL(cpam_dp_kernel<float>, dim3(nb, B * C), dim3(CPAM_THREADS), 0, s, (const float*)du_ws, (float*)dp, H, W);
  } else {
    hipLaunchKernelGGL(cpam_bwd_kernel<bf16_t>, dim3(nb, CHUNKS, B), dim3(CPAM_THREADS), 0, s, (const bf16_t*)gout, (const bf16_t*)x,
                       (const bf16_t*)p, s2, arg, (bf16_t*)dx_direct, (bf16_t*)du_ws, C, H, W);
    hipLaunchKernelGGL(cpam_dp_kernel<bf16_t>, dim3(nb, B * C), dim3(CPAM_THREADS), 0, s, (const bf16_t*)du_ws, (bf16_t*)dp, H, W);
  }
  return tamtr_launch_status();
}

// channels-last forms: x, p, out, gout, dx_direct, du_ws, dp (T) [B][H][W][C] / [B][H/2][W/2][C]; s2 f32, arg i32 [B][H][W][8]
static int cpam_cl_check(const void* a, const void* b, const void* c, int B, int C, int H, int W, int dtype) {
  const int rc = cpam_check(a, b, c, B, C, H, W, dtype);
  if (rc) return rc;
  const int V = dtype == TAMTR_F32 ? 4 : 8, LPB = C / V;
  // a chunk = a power-of-two number of whole lanes, a block's lanes inside one wave-aligned group of the 256 threads
  if (C % (CHUNKS * V) || LPB > CPAM_THREADS || CPAM_THREADS % LPB || ((LPB / CHUNKS) & (LPB / CHUNKS - 1)) || LPB / CHUNKS > WAVE) return TAMTR_EUNSUP;
  if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) % 16) return TAMTR_EUNSUP;
  return TAMTR_OK;
}

extern "C" int tamtr_cpam_cl_fwd(const void* x, void* p, uint8_t* code, void* out, float* s2, int32_t* arg, int B, int C, int H, int W, int dtype,
                                 void* stream) {
  const int rc = cpam_cl_check(x, p, out, B, C, H, W, dtype);
  if (rc) return rc;
  if (!s2 || !arg || !code || (uintptr_t)code % 8) return TAMTR_EINVAL;
  const int V = dtype == TAMTR_F32 ? 4 : 8, LPB = C / V, bpw = CPAM_THREADS / LPB;
  const long long nblk = (long long)B * (H / 2) * (W / 2), grid = (nblk + bpw - 1) / bpw, g1 = (nblk * LPB + CPAM_THREADS - 1) / CPAM_THREADS;
  if (grid > 0x7fffffffLL || g1 > 0x7fffffffLL) return TAMTR_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TAMTR_F32) {
    hipLaunchKernelGGL(pool3s2_cl_fwd_kernel<float>, dim3((unsigned)g1), dim3(CPAM_THREADS), 0, s, (const float*)x, (float*)p, code, B, C, H, W);
    hipLaunchKernelGGL(cpam_cl_fwd_kernel<float>, dim3((unsigned)grid), dim3(CPAM_THREADS), 0, s, (const float*)x, (const float*)p, (float*)out, s2, arg, B, C, H, W);
  } else {
    hipLaunchKernelGGL(pool3s2_cl_fwd_kernel<bf16_t>, dim3((unsigned)g1), dim3(CPAM_THREADS), 0, s, (const bf16_t*)x, (bf16_t*)p, code, B, C, H, W);
    hipLaunchKernelGGL(cpam_cl_fwd_kernel<bf16_t>, dim3((unsigned)grid), dim3(CPAM_THREADS), 0, s, (const bf16_t*)x, (const bf16_t*)p, (bf16_t*)out, s2, arg, B, C,
                       H, W);
  }
  return tamtr_launch_status();
}

extern "C" int tamtr_cpam_cl_bwd(const void* gout, const void* x, const void* p, const uint8_t* code, const float* s2, const int32_t* arg, void* dxd_ws,
                                 void* du_ws, void* dp_ws, void* dx, int B, int C, int H, int W, int dtype, void* stream) {
  const int rc = cpam_cl_check(gout, x, p, B, C, H, W, dtype);
  if (rc) return rc;
  if (!s2 || !arg || !code || !dxd_ws || !du_ws || !dp_ws || !dx) return TAMTR_EINVAL;
  if (((uintptr_t)dxd_ws | (uintptr_t)du_ws | (uintptr_t)dp_ws | (uintptr_t)dx) % 16 || (uintptr_t)code % 8) return TAMTR_EUNSUP;
  const int V = dtype == TAMTR_F32 ? 4 : 8, LPB = C / V, bpw = CPAM_THREADS / LPB;
  const long long nblk = (long long)B * (H / 2) * (W / 2), grid = (nblk + bpw - 1) / bpw, g2 = (nblk * LPB + CPAM_THREADS - 1) / CPAM_THREADS,
                  g3 = (4 * nblk * LPB + CPAM_THREADS - 1) / CPAM_THREADS;
  if (grid > 0x7fffffffLL || g3 > 0x7fffffffLL) return TAMTR_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
#define CPAM_CL_BWD(T)                                                                                                                              \
  hipLaunchKernelGGL(cpam_cl_bwd_kernel<T>, dim3((unsigned)grid), dim3(CPAM_THREADS), 0, s, (const T*)gout, (const T*)x, (const T*)p, s2, arg,     \
                     (T*)dxd_ws, (T*)du_ws, B, C, H, W);                                                                                            \
  hipLaunchKernelGGL(cpam_cl_dp_kernel<T>, dim3((unsigned)g2), dim3(CPAM_THREADS), 0, s, (const T*)du_ws, (T*)dp_ws, B, C, H, W);                  \
  hipLaunchKernelGGL(pool3s2_cl_bwd_kernel<T>, dim3((unsigned)g3), dim3(CPAM_THREADS), 0, s, (const T*)dp_ws, code, (const T*)dxd_ws, (T*)dx, B, C, H, W)
  if (dtype == TAMTR_F32) { CPAM_CL_BWD(float); } else { CPAM_CL_BWD(bf16_t); }
#undef CPAM_CL_BWD
  return tamtr_launch_status();
}
