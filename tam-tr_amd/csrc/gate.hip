// gate.hip - fused max-sigmoid text gate (BTA-PAN), forward + backward, gfx950.
//
// Replaces MaxSigmoidAttnBlock.forward's einsum/max/div/add/sigmoid/mul chain
// (reference: ultralytics/nn/extra_modules/block.py:217-226), which materialises aw[B,nh,H,W,T] and makes >= 8 passes
// over HBM.  Here: ONE pass.  HBM-bound by construction (AI ~ 5 flop/B), so the design rules are the byte ones:
//   * NCHW is read as it lies: a workgroup owns (b, head m, a run of PXB pixels); lanes run along the pixel axis so
//     every channel row is a coalesced 16-B-per-lane (f32) / 8-B-per-lane (bf16) stream;
//   * the text tile gk[b, :, m, :]  ([T, hc] f32, <= 20 KB) is staged ONCE per workgroup in LDS and read back as
//     wave-uniform broadcasts (no bank conflicts);
//   * the T dot products of a pixel live in registers (TC accumulators x 4 pixels per lane), max/argmax over the
//     text axis never leaves the lane, so no cross-lane traffic at all;
//   * x is read once (T <= TC; once per TC-chunk otherwise), v once, out written once.
#include "common.h"

namespace {

constexpr int GATE_THREADS = 256;
constexpr int TC = 16;  // text rows accumulated per pass

template <typename ET, int PX>
__global__ __launch_bounds__(GATE_THREADS) void gate_fwd_kernel(const ET* __restrict__ x, const float* __restrict__ gk,
                                                                 const float* __restrict__ bias, const ET* __restrict__ v,
                                                                 ET* __restrict__ out, float* __restrict__ aw_out,
                                                                 int32_t* __restrict__ arg_out, int nh, int hc, int HW, int T,
                                                                 float scale) {
  extern __shared__ float s_gk[];  // [T][hc]
  const int m = blockIdx.y, b = blockIdx.z;
  const int C = nh * hc;
  for (int i = threadIdx.x; i < T * hc; i += GATE_THREADS) {
    int n = i / hc, c = i - n * hc;
    s_gk[i] = gk[((size_t)b * T + n) * C + m * hc + c];
  }
  __syncthreads();
  const int p0 = (blockIdx.x * GATE_THREADS + threadIdx.x) * PX;
  if (p0 >= HW) return;
  const size_t base = ((size_t)b * C + (size_t)m * hc) * HW + p0;
  const ET* xp = x + base;

  float best[PX];
  int besti[PX];
#pragma unroll
  for (int j = 0; j < PX; ++j) { best[j] = -INFINITY; besti[j] = 0; }

  for (int n0 = 0; n0 < T; n0 += TC) {
    float acc[TC][PX];
#pragma unroll
    for (int n = 0; n < TC; ++n)
#pragma unroll
      for (int j = 0; j < PX; ++j) acc[n][j] = 0.f;
    for (int c = 0; c < hc; ++c) {
      float xv[PX];
      if constexpr (PX == 4) Elt<ET>::ld4(xp + (size_t)c * HW, xv);
      else xv[0] = Elt<ET>::ld(xp + (size_t)c * HW);
#pragma unroll
      for (int n = 0; n < TC; ++n) {
        // rows beyond T read row T-1 again (in bounds) and are ignored by the max below
        float g = s_gk[min(n0 + n, T - 1) * hc + c];
#pragma unroll
        for (int j = 0; j < PX; ++j) acc[n][j] = fmaf(xv[j], g, acc[n][j]);
      }
    }
#pragma unroll
    for (int n = 0; n < TC; ++n) {
      if (n0 + n < T) {
#pragma unroll
        for (int j = 0; j < PX; ++j)
          if (acc[n][j] > best[j]) { best[j] = acc[n][j]; besti[j] = n0 + n; }
      }
    }
  }

  const float rs = sqrtf((float)hc), bm = bias[m];
  float a[PX];
#pragma unroll
  for (int j = 0; j < PX; ++j) a[j] = 1.f / (1.f + __expf(-(best[j] / rs + bm)));
  const size_t pa = ((size_t)b * nh + m) * HW + p0;
  if constexpr (PX == 4) {
    *reinterpret_cast<float4*>(aw_out + pa) = make_float4(a[0], a[1], a[2], a[3]);
    *reinterpret_cast<int4*>(arg_out + pa) = make_int4(besti[0], besti[1], besti[2], besti[3]);
  } else {
    aw_out[pa] = a[0];
    arg_out[pa] = besti[0];
  }
  const ET* vp = v + base;
  ET* op = out + base;
  for (int c = 0; c < hc; ++c) {
    float vv[PX];
    if constexpr (PX == 4) {
      Elt<ET>::ld4(vp + (size_t)c * HW, vv);
#pragma unroll
      for (int j = 0; j < PX; ++j) vv[j] *= a[j] * scale;
      Elt<ET>::st4(op + (size_t)c * HW, vv);
    } else {
      Elt<ET>::st(op + (size_t)c * HW, Elt<ET>::ld(vp + (size_t)c * HW) * a[0] * scale);
    }
  }
}

template <typename ET, int PX>
__global__ __launch_bounds__(GATE_THREADS) void gate_bwd_kernel(const ET* __restrict__ dout, const float* __restrict__ gk,
                                                                 const ET* __restrict__ v, const float* __restrict__ aw,
                                                                 const int32_t* __restrict__ arg, ET* __restrict__ dx,
                                                                 ET* __restrict__ dv, float* __restrict__ dlogit, int nh, int hc,
                                                                 int HW, int T, float scale) {
  extern __shared__ float s_gk[];  // [T][hc+1] (padded: per-lane row gather in the last loop)
  const int m = blockIdx.y, b = blockIdx.z;
  const int C = nh * hc, ld = hc + 1;
  for (int i = threadIdx.x; i < T * hc; i += GATE_THREADS) {
    int n = i / hc, c = i - n * hc;
    s_gk[n * ld + c] = gk[((size_t)b * T + n) * C + m * hc + c];
  }
  __syncthreads();
  const int p0 = (blockIdx.x * GATE_THREADS + threadIdx.x) * PX;
  if (p0 >= HW) return;
  const size_t base = ((size_t)b * C + (size_t)m * hc) * HW + p0;
  const size_t pa = ((size_t)b * nh + m) * HW + p0;
  float a[PX], daw[PX];
  int n_[PX];
  if constexpr (PX == 4) {
    float4 t = *reinterpret_cast<const float4*>(aw + pa);
    int4 q = *reinterpret_cast<const int4*>(arg + pa);
    a[0] = t.x; a[1] = t.y; a[2] = t.z; a[3] = t.w;
    n_[0] = q.x; n_[1] = q.y; n_[2] = q.z; n_[3] = q.w;
  } else {
    a[0] = aw[pa];
    n_[0] = arg[pa];
  }
#pragma unroll
  for (int j = 0; j < PX; ++j) daw[j] = 0.f;
  for (int c = 0; c < hc; ++c) {
    float g[PX], vv[PX];
    if constexpr (PX == 4) {
      Elt<ET>::ld4(dout + base + (size_t)c * HW, g);
      Elt<ET>::ld4(v + base + (size_t)c * HW, vv);
    } else {
      g[0] = Elt<ET>::ld(dout + base + (size_t)c * HW);
      vv[0] = Elt<ET>::ld(v + base + (size_t)c * HW);
    }
#pragma unroll
    for (int j = 0; j < PX; ++j) { daw[j] = fmaf(g[j], vv[j], daw[j]); g[j] *= a[j] * scale; }
    if constexpr (PX == 4) Elt<ET>::st4(dv + base + (size_t)c * HW, g);
    else Elt<ET>::st(dv + base + (size_t)c * HW, g[0]);
  }
  const float rs = sqrtf((float)hc);
  float dl[PX];
#pragma unroll
  for (int j = 0; j < PX; ++j) dl[j] = daw[j] * scale * a[j] * (1.f - a[j]) / rs;
  if constexpr (PX == 4) *reinterpret_cast<float4*>(dlogit + pa) = make_float4(dl[0], dl[1], dl[2], dl[3]);
  else dlogit[pa] = dl[0];
  for (int c = 0; c < hc; ++c) {
    float o[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) o[j] = dl[j] * s_gk[n_[j] * ld + c];
    if constexpr (PX == 4) Elt<ET>::st4(dx + base + (size_t)c * HW, o);
    else Elt<ET>::st(dx + base + (size_t)c * HW, o[0]);
  }
}

inline bool gate_args_ok(int B, int nh, int hc, int HW, int T) {
  return B > 0 && nh > 0 && hc > 0 && HW > 0 && T > 0 && nh <= 65535 && B <= 65535;
}


// ---- channels-last forward with the value branch's BatchNorm folded into the load (second half).
// Reference: `self.proj_conv(x)` = Conv3x3 + BatchNorm (block.py:205,223) feeding `x * aw` (block.py:224-226); the embed branch
// (`self.ec`, block.py:203) likewise when it exists (it is None in every TAM-TR instance: embed = x).  With the NHWC trunk both
// operands arrive pixel-major; as separate ops the value branch cost a BatchNorm apply pass and each operand a transpose to the NCHW
// planes of gate_fwd_kernel.  Here: e and the RAW convolution output v are read once ([B*HW, C], e possibly a channel slice of a wider
// map through its row pitch), each lane owns 8 channels and applies its per-channel affine (a = rstd * gamma, b = beta - mean * a,
// built from the batch statistics) in registers; a head's hc channels sit in hc / 8 adjacent lanes, so the T text dot products are
// 8 FMAs per lane plus a butterfly over those lanes; max / sigmoid / scale and one 16-byte store.  Text tile [T, C] in LDS.
//
// Round 4: the round-3 form walked its 128 rows one at a time - one 16-byte load in flight per lane, every row a
// full memory round trip behind the previous one's stores: 54 - 68 us per site = 0.7 - 2.3 TB/s, slower than the NCHW kernel it
// replaced, with 208 workgroups at the 40 x 40 sites.  Now a lane takes U = 4 rows: their 8 loads (e and v) are issued together
// before anything is computed, the text row is read from LDS once per t for all four, the butterfly is two DPP quad permutes
// (hc = 32: four lanes per head) instead of ds_bpermute round trips, and a workgroup covers exactly 4 waves x (64 / lanes-per-row)
// x U rows - 32 / 64 / 128 rows at C = 256 / 128 / 64, i.e. 800 / 1 600 / 3 200 workgroups at the three site shapes (16 images).
template <typename ET>
__device__ __forceinline__ void ld8(const ET* p, float (&o)[8]);
template <>
__device__ __forceinline__ void ld8<float>(const float* p, float (&o)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
template <>
__device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float (&o)[8]) {
  const uint4 t = *reinterpret_cast<const uint4*>(p);
  const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
template <typename ET>
__device__ __forceinline__ void st8(ET* p, const float (&v)[8]);
template <>
__device__ __forceinline__ void st8<float>(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <>
__device__ __forceinline__ void st8<bf16_t>(bf16_t* p, const float (&v)[8]) {
  uint4 t;
  t.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
  t.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
  t.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
  t.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
  *reinterpret_cast<uint4*>(p) = t;
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {   // quad permute of v (DPP: no LDS round trip)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

constexpr int GATE_CL_U = 4;   // rows per lane, all loads issued up front

template <typename ET, bool VBN>
__global__ __launch_bounds__(GATE_THREADS) void gate_cl_fwd_kernel(const ET* __restrict__ e, size_t ld_e, const float* __restrict__ mr_e,
                                                                    const float* __restrict__ ga_e, const float* __restrict__ be_e,
                                                                    const ET* __restrict__ v, const float* __restrict__ mr_v,
                                                                    const float* __restrict__ ga_v, const float* __restrict__ be_v,
                                                                    const float* __restrict__ gk, const float* __restrict__ bias,
                                                                    ET* __restrict__ out, float* __restrict__ aw_out,
                                                                    int32_t* __restrict__ arg_out, int C, int hc, int HW, int T, float scale) {
  constexpr int U = GATE_CL_U;
  extern __shared__ __attribute__((aligned(16))) float s_gk[];  // [T][C]
  const int b = blockIdx.y, nh = C / hc;
  const int lpr = C / 8, rpw = WAVE / lpr;           // lanes per pixel row, rows per wave
  const int lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE;
  const int c0 = (lane % lpr) * 8, rl = lane / lpr, head = c0 / hc, lph = hc / 8;
  const int stride = (GATE_THREADS / WAVE) * rpw;    // rows between a lane's consecutive rows
  const int r0 = blockIdx.x * (stride * U) + wave * rpw + rl;
  // Everything this thread reads from global memory is requested here, unconditionally and in the order it is needed (a load inside a
  // conditional block costs a drain of the whole queue at the block's end): text tile pieces, the value branch's BatchNorm constants,
  // then the U rows of e and v.
  const int n4 = T * C / 4;                           // float4 pieces of the text tile; <= 3 per thread at T * C <= 3072
  const float4* gk4 = reinterpret_cast<const float4*>(gk + (size_t)b * T * C);
  float4 tile[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) tile[k] = gk4[min((int)threadIdx.x + k * GATE_THREADS, n4 - 1)];
  float4 mrv[4], gav[2], bev[2];
  if constexpr (VBN) {
#pragma unroll
    for (int k = 0; k < 4; ++k) mrv[k] = reinterpret_cast<const float4*>(mr_v + 2 * c0)[k];
#pragma unroll
    for (int k = 0; k < 2; ++k) { gav[k] = reinterpret_cast<const float4*>(ga_v + c0)[k]; bev[k] = reinterpret_cast<const float4*>(be_v + c0)[k]; }
  }
  const float bm = bias[head];
  float ev[U][8], vv[U][8];
  size_t rowi[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    rowi[u] = (size_t)b * HW + min(r0 + u * stride, HW - 1);
    ld8<ET>(e + rowi[u] * ld_e + c0, ev[u]);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) ld8<ET>(v + rowi[u] * C + c0, vv[u]);
#pragma unroll
  for (int k = 0; k < 3; ++k)   // (clamped like the loads: a conditional store would pull its load into the branch, behind a queue drain)
    reinterpret_cast<float4*>(s_gk)[min((int)threadIdx.x + k * GATE_THREADS, n4 - 1)] = tile[k];
  for (int i = threadIdx.x + 3 * GATE_THREADS; i < n4; i += GATE_THREADS) reinterpret_cast<float4*>(s_gk)[i] = gk4[i];   // (T * C > 3072 only)
  float av[8], bv[8];
  if constexpr (VBN) {
    const float mr[16] = {mrv[0].x, mrv[0].y, mrv[0].z, mrv[0].w, mrv[1].x, mrv[1].y, mrv[1].z, mrv[1].w,
                          mrv[2].x, mrv[2].y, mrv[2].z, mrv[2].w, mrv[3].x, mrv[3].y, mrv[3].z, mrv[3].w};
    const float ga[8] = {gav[0].x, gav[0].y, gav[0].z, gav[0].w, gav[1].x, gav[1].y, gav[1].z, gav[1].w};
    const float be[8] = {bev[0].x, bev[0].y, bev[0].z, bev[0].w, bev[1].x, bev[1].y, bev[1].z, bev[1].w};
#pragma unroll
    for (int j = 0; j < 8; ++j) { av[j] = mr[2 * j + 1] * ga[j]; bv[j] = be[j] - mr[2 * j] * av[j]; }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) { av[j] = 1.f; bv[j] = 0.f; }
  }
  if (mr_e) {   // (never in TAM-TR: ec is None, the embed operand has no BatchNorm)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = mr_e[2 * (c0 + j) + 1] * ga_e[c0 + j], bb = be_e[c0 + j] - mr_e[2 * (c0 + j)] * a;
#pragma unroll
      for (int u = 0; u < U; ++u) ev[u][j] = fmaf(ev[u][j], a, bb);
    }
  }
  const float rs = rsqrtf((float)hc);
  __syncthreads();
  float best[U];
  int besti[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { best[u] = -INFINITY; besti[u] = 0; }
  for (int t = 0; t < T; ++t) {
    const float4* gt = reinterpret_cast<const float4*>(s_gk) + ((t * C + c0) >> 2);   // (float4 indexing: provably 16-byte aligned -> ds_read_b128)
    const float4 g0 = gt[0], g1 = gt[1];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float d = ev[u][0] * g0.x;
      d = fmaf(ev[u][1], g0.y, d); d = fmaf(ev[u][2], g0.z, d); d = fmaf(ev[u][3], g0.w, d);
      d = fmaf(ev[u][4], g1.x, d); d = fmaf(ev[u][5], g1.y, d); d = fmaf(ev[u][6], g1.z, d); d = fmaf(ev[u][7], g1.w, d);
      if (lph >= 2) d += dpp_f<0xB1>(d);             // lane ^ 1
      if (lph >= 4) d += dpp_f<0x4E>(d);             // lane ^ 2
      for (int o = 4; o < lph; o <<= 1) d += __shfl_xor(d, o, WAVE);
      if (d > best[u]) { best[u] = d; besti[u] = t; }
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int r = r0 + u * stride;
    const float a = 1.f / (1.f + __expf(-(best[u] * rs + bm)));
#pragma unroll
    for (int j = 0; j < 8; ++j) vv[u][j] = fmaf(vv[u][j], av[j], bv[j]) * (a * scale);
    if (r < HW) {
      st8<ET>(out + rowi[u] * C + c0, vv[u]);
      if (aw_out && c0 % hc == 0) {
        aw_out[((size_t)b * nh + head) * HW + r] = a;
        if (arg_out) arg_out[((size_t)b * nh + head) * HW + r] = besti[u];
      }
    }
  }
}

}  // namespace

extern "C" int tamtr_maxsigmoid_gate_fwd(const void* x, const float* gk, const float* bias, const void* v, void* out,
                                         float* aw, int32_t* arg, int B, int nh, int hc, int HW, int T, float scale, int dtype,
                                         void* stream) {
  if (!x || !gk || !bias || !v || !out || !aw || !arg || !gate_args_ok(B, nh, hc, HW, T)) return TAMTR_EINVAL;
  if ((size_t)T * hc * sizeof(float) > 60 * 1024) return TAMTR_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)T * hc * sizeof(float);
  const bool vec = (HW % 4) == 0;
  const int px = vec ? 4 : 1;
  dim3 grid((HW + GATE_THREADS * px - 1) / (GATE_THREADS * px), nh, B);
#define GO(ET, PX) \
  hipLaunchKernelGGL((gate_fwd_kernel<ET, PX>), grid, dim3(GATE_THREADS), lds, s, (const ET*)x, gk, bias, (const ET*)v, \
                     (ET*)out, aw, arg, nh, hc, HW, T, scale)
  if (dtype == TAMTR_F32) { if (vec) GO(float, 4); else GO(float, 1); }
  else if (dtype == TAMTR_BF16) { if (vec) GO(bf16_t, 4); else GO(bf16_t, 1); }
  else return TAMTR_EINVAL;
#undef GO
  return tamtr_launch_status();
}

extern "C" int tamtr_maxsigmoid_gate_bwd(const void* dout, const void* x, const float* gk, const void* v, const float* aw,
                                         const int32_t* arg, void* dx, void* dv, float* dlogit, int B, int nh, int hc, int HW,
                                         int T, float scale, int dtype, void* stream) {
  (void)x;
  if (!dout || !gk || !v || !aw || !arg || !dx || !dv || !dlogit || !gate_args_ok(B, nh, hc, HW, T)) return TAMTR_EINVAL;
  if ((size_t)T * (hc + 1) * sizeof(float) > 60 * 1024) return TAMTR_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)T * (hc + 1) * sizeof(float);
  const bool vec = (HW % 4) == 0;
  const int px = vec ? 4 : 1;
  dim3 grid((HW + GATE_THREADS * px - 1) / (GATE_THREADS * px), nh, B);
#define GO(ET, PX) \
  hipLaunchKernelGGL((gate_bwd_kernel<ET, PX>), grid, dim3(GATE_THREADS), lds, s, (const ET*)dout, gk, (const ET*)v, aw, arg, \
                     (ET*)dx, (ET*)dv, dlogit, nh, hc, HW, T, scale)
  if (dtype == TAMTR_F32) { if (vec) GO(float, 4); else GO(float, 1); }
  else if (dtype == TAMTR_BF16) { if (vec) GO(bf16_t, 4); else GO(bf16_t, 1); }
  else return TAMTR_EINVAL;
#undef GO
  return tamtr_launch_status();
}

/* channels-last forward, per-channel affines (BatchNorm of the value / embed branch) applied in the load: see gate_cl_fwd_kernel */
extern "C" int tamtr_maxsigmoid_gate_cl_fwd(const void* e, long long ld_e, const float* mean_rstd_e, const float* gamma_e, const float* beta_e,
                                            const void* v, const float* mean_rstd_v, const float* gamma_v, const float* beta_v,
                                            const float* gk, const float* bias, void* out, float* aw, int32_t* arg, int B, int nh, int hc,
                                            int HW, int T, float scale, int dtype, void* stream) {
  if (!e || !v || !gk || !bias || !out || !gate_args_ok(B, nh, hc, HW, T)) return TAMTR_EINVAL;
  if ((mean_rstd_e && (!gamma_e || !beta_e)) || (mean_rstd_v && (!gamma_v || !beta_v)) || (arg && !aw)) return TAMTR_EINVAL;
  if (dtype != TAMTR_F32 && dtype != TAMTR_BF16) return TAMTR_EINVAL;
  const int C = nh * hc, lpr = C / 8, al = 16;          // a lane moves 8 channels: 16-byte (bf16) / 2 x 16-byte (f32) accesses
  if (C % 8 || hc % 8 || lpr > WAVE || (lpr & (lpr - 1)) || ((hc / 8) & (hc / 8 - 1)) || ld_e < C || ld_e % (dtype == TAMTR_F32 ? 4 : 8) ||
      ((uintptr_t)e | (uintptr_t)v | (uintptr_t)out | (uintptr_t)gk | (uintptr_t)mean_rstd_v | (uintptr_t)gamma_v | (uintptr_t)beta_v) % al || (size_t)T * C * sizeof(float) > 60 * 1024 || B > 65535)
    return TAMTR_EUNSUP;
  const int rows_per_wg = (GATE_THREADS / WAVE) * (WAVE / lpr) * GATE_CL_U;   // 32 / 64 / 128 rows at C = 256 / 128 / 64
  const dim3 grid((HW + rows_per_wg - 1) / rows_per_wg, B);
  const size_t lds = (size_t)T * C * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
#define GO(ET, VBN)                                                                                                                       \
  hipLaunchKernelGGL((gate_cl_fwd_kernel<ET, VBN>), grid, dim3(GATE_THREADS), lds, s, (const ET*)e, (size_t)ld_e, mean_rstd_e, gamma_e, beta_e, \
                     (const ET*)v, mean_rstd_v, gamma_v, beta_v, gk, bias, (ET*)out, aw, arg, C, hc, HW, T, scale)
  if (dtype == TAMTR_F32) { if (mean_rstd_v) GO(float, true); else GO(float, false); }
  else { if (mean_rstd_v) GO(bf16_t, true); else GO(bf16_t, false); }
#undef GO
  return tamtr_launch_status();
}
