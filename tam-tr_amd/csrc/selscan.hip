// selscan.hip - selective scan (S6) forward + backward for the MEH VSSBlocks, gfx950, fp32.
//
// Replaces the external CUDA extension `selective_scan_cuda_core` the reference calls at
// ultralytics/nn/extra_modules/VManba/csms6s.py:257,267 (contract: vmamba.py:962-990).  Written from the recurrence,
// not from that extension (whose source is not in the reference tree):
//     dt = softplus(delta + dbias);  h_t = exp(dt_t A_n) h_{t-1} + dt_t B_{n,t} u_t;  y_t = sum_n C_{n,t} h_{n,t} + D u_t
//
// Mapping (wave64): ONE WAVE PER ROW (b, k*Dk+d); lanes run along TIME, 4 consecutive steps per lane, so a wave eats
// a CHUNK of 256 steps per iteration with perfectly coalesced 16-B/lane loads of u, delta, B_n, C_n and stores of y.
// Inside a chunk the linear recurrence is an associative scan on pairs (a, b): 3 sequential steps inside the lane,
// then a 6-level Hillis-Steele scan across the 64 lanes with wave shuffles, then the carry h_{n} (wave-uniform) of the
// previous chunk is folded in.  The 16 states are looped, the y dot product stays inside the lane (no cross-lane
// reduction).  Chunk-boundary states are written out for the backward pass, which walks the chunks in reverse,
// recomputes h inside the chunk and runs the mirrored (suffix) scan for dL/dh.
// dB/dC are shared by the Dk rows of a (b, k) group: the ROWS waves of a workgroup (all in one group) first reduce
// them in LDS, then issue full-row (256-B contiguous) float atomics - the full-rate atomic shape on gfx950.
#include "common.h"

namespace {

constexpr int NS = 16;        // d_state
constexpr int ITEMS = 4;      // time steps per lane
constexpr int CHUNK = WAVE * ITEMS;
constexpr int FWD_ROWS = 4;   // waves (rows) per workgroup, forward
constexpr int BWD_ROWS = 8;   // waves (rows) per workgroup, backward (shares the dB/dC LDS reduction)

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(__expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }

template <bool VEC>
__device__ __forceinline__ void load4(const float* __restrict__ p, int t, int L, float (&o)[ITEMS], float fill) {
  if (VEC) {
    if (t < L) {  // L % 4 == 0 and t % 4 == 0: all-or-nothing
      const float4 v = *reinterpret_cast<const float4*>(p + t);
      o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    } else {
      o[0] = o[1] = o[2] = o[3] = fill;
    }
  } else {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) o[i] = (t + i < L) ? p[t + i] : fill;
  }
}
template <bool VEC>
__device__ __forceinline__ void store4(float* __restrict__ p, int t, int L, const float (&v)[ITEMS]) {
  if (VEC) {
    if (t < L) *reinterpret_cast<float4*>(p + t) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
      if (t + i < L) p[t + i] = v[i];
  }
}

// inclusive prefix scan over lanes of the affine maps h -> A*h + B (composition: earlier map applied first)
__device__ __forceinline__ void wave_scan_prefix(float& A, float& Bv, int lane) {
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    const float pa = __shfl_up(A, o, WAVE), pb = __shfl_up(Bv, o, WAVE);
    if (lane >= o) { Bv = fmaf(A, pb, Bv); A *= pa; }
  }
}
// inclusive suffix scan: g -> A*g + B, later map applied first (mirror image)
__device__ __forceinline__ void wave_scan_suffix(float& A, float& Bv, int lane) {
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    const float pa = __shfl_down(A, o, WAVE), pb = __shfl_down(Bv, o, WAVE);
    if (lane + o < WAVE) { Bv = fmaf(A, pb, Bv); A *= pa; }
  }
}

template <bool VEC>
__global__ __launch_bounds__(FWD_ROWS* WAVE) void selscan_fwd_kernel(const float* __restrict__ u, const float* __restrict__ delta,
                                                                      const float* __restrict__ Am, const float* __restrict__ Bm,
                                                                      const float* __restrict__ Cm, const float* __restrict__ Dv,
                                                                      const float* __restrict__ dbias, float* __restrict__ y,
                                                                      float* __restrict__ hstate, int n_rows, int K, int Dk, int L,
                                                                      int nchunk) {
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  const int row = blockIdx.x * FWD_ROWS + wave;  // row = b*K*Dk + k*Dk + d
  if (row >= n_rows) return;
  const int kd = row % (K * Dk);
  const int b = row / (K * Dk);
  const int k = kd / Dk;
  const float* up = u + (size_t)row * L;
  const float* dp = delta + (size_t)row * L;
  const float* Bp = Bm + ((size_t)b * K + k) * NS * L;
  const float* Cp = Cm + ((size_t)b * K + k) * NS * L;
  float* yp = y + (size_t)row * L;
  // per-state wave-uniform values (A_n, carried h_n) live in LDS so that the state loop can stay rolled: only this
  // wave touches its slots and a wave's LDS operations execute in order, so no barrier is involved
  __shared__ float s_A[FWD_ROWS][NS], s_h[FWD_ROWS][NS];
  float* An = s_A[wave];
  float* h = s_h[wave];
  if (lane < NS) { An[lane] = Am[(size_t)kd * NS + lane]; h[lane] = 0.f; }
  const float Dd = Dv[kd], bias = dbias[kd];

  for (int c = 0; c < nchunk; ++c) {
    const int t = c * CHUNK + lane * ITEMS;
    float uu[ITEMS], dt[ITEMS], yy[ITEMS];
    load4<VEC>(up, t, L, uu, 0.f);
    load4<VEC>(dp, t, L, dt, 0.f);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      // steps beyond L become the identity map (a = 1, b = 0)
      dt[i] = (t + i < L) ? softplus_f(dt[i] + bias) : 0.f;
      yy[i] = Dd * uu[i];
    }
#pragma unroll 4
    for (int n = 0; n < NS; ++n) {
      float bb[ITEMS], cc[ITEMS], a[ITEMS];
      load4<VEC>(Bp + (size_t)n * L, t, L, bb, 0.f);
      load4<VEC>(Cp + (size_t)n * L, t, L, cc, 0.f);
      const float An_n = An[n];
      float PA = 1.f, PB = 0.f;
#pragma unroll
      for (int i = 0; i < ITEMS; ++i) {
        a[i] = __expf(dt[i] * An_n);
        bb[i] = dt[i] * uu[i] * bb[i];
        PB = fmaf(a[i], PB, bb[i]);
        PA *= a[i];
      }
      wave_scan_prefix(PA, PB, lane);
      float EA = __shfl_up(PA, 1, WAVE), EB = __shfl_up(PB, 1, WAVE);
      if (lane == 0) { EA = 1.f; EB = 0.f; }
      float hh = fmaf(EA, h[n], EB);  // state entering this lane's first step
#pragma unroll
      for (int i = 0; i < ITEMS; ++i) {
        hh = fmaf(a[i], hh, bb[i]);
        yy[i] = fmaf(cc[i], hh, yy[i]);
      }
      if (lane == WAVE - 1) h[n] = hh;  // state after the chunk
    }
    store4<VEC>(yp, t, L, yy);
    if (lane < NS) hstate[((size_t)row * nchunk + c) * NS + lane] = h[lane];
  }
}

template <bool VEC>
__global__ __launch_bounds__(BWD_ROWS* WAVE) void selscan_bwd_kernel(
    const float* __restrict__ gy, const float* __restrict__ u, const float* __restrict__ delta, const float* __restrict__ Am,
    const float* __restrict__ Bm, const float* __restrict__ Cm, const float* __restrict__ Dv, const float* __restrict__ dbias,
    const float* __restrict__ hstate, float* __restrict__ gu, float* __restrict__ gdelta, float* __restrict__ gA,
    float* __restrict__ gB, float* __restrict__ gC, float* __restrict__ gD, float* __restrict__ gdbias, int K, int Dk, int L,
    int nchunk) {
  __shared__ float s_dB[NS][CHUNK];
  __shared__ float s_dC[NS][CHUNK];
  __shared__ float s_A[BWD_ROWS][NS], s_carry[BWD_ROWS][NS];  // wave-private, wave-uniform per-state values
  __shared__ float s_dA[BWD_ROWS][NS][WAVE];                   // per-lane dA partial sums
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  // grid: x = row-block inside the (b,k) group, y = b*K + k  -> all waves of a workgroup share B/C
  const int d = blockIdx.x * BWD_ROWS + wave;
  const bool live = d < Dk;
  const int bk = blockIdx.y;
  const int k = bk % K;
  const int kd = k * Dk + (live ? d : 0);
  const size_t row = (size_t)(bk / K) * K * Dk + kd;
  const float* up = u + row * L;
  const float* dp = delta + row * L;
  const float* gyp = gy + row * L;
  const float* Bp = Bm + (size_t)bk * NS * L;
  const float* Cp = Cm + (size_t)bk * NS * L;
  float* An = s_A[wave];
  float* carry = s_carry[wave];
  if (lane < NS) { An[lane] = Am[(size_t)kd * NS + lane]; carry[lane] = 0.f; }
  for (int n = 0; n < NS; ++n) s_dA[wave][n][lane] = 0.f;
  const float Dd = Dv[kd], bias = dbias[kd];
  float dD = 0.f, dbs = 0.f;

  for (int c = nchunk - 1; c >= 0; --c) {
    for (int i = threadIdx.x; i < NS * CHUNK; i += BWD_ROWS * WAVE) { (&s_dB[0][0])[i] = 0.f; (&s_dC[0][0])[i] = 0.f; }
    __syncthreads();
    const int t = c * CHUNK + lane * ITEMS;
    if (live) {
      float uu[ITEMS], dl[ITEMS], dt[ITEMS], g[ITEMS], ddt[ITEMS], du[ITEMS];
      load4<VEC>(up, t, L, uu, 0.f);
      load4<VEC>(dp, t, L, dl, 0.f);
      load4<VEC>(gyp, t, L, g, 0.f);
#pragma unroll
      for (int i = 0; i < ITEMS; ++i) {
        dt[i] = (t + i < L) ? softplus_f(dl[i] + bias) : 0.f;
        ddt[i] = 0.f;
        du[i] = Dd * g[i];
        dD = fmaf(g[i], uu[i], dD);
      }
#pragma unroll 2
      for (int n = 0; n < NS; ++n) {
        float bb[ITEMS], cc[ITEMS], a[ITEMS], hh[ITEMS], bu[ITEMS];
        load4<VEC>(Bp + (size_t)n * L, t, L, bb, 0.f);
        load4<VEC>(Cp + (size_t)n * L, t, L, cc, 0.f);
        const float An_n = An[n];
        float dA_n = 0.f;
        // ---- recompute h inside the chunk (same arithmetic as the forward)
        float PA = 1.f, PB = 0.f;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
          a[i] = __expf(dt[i] * An_n);
          bu[i] = dt[i] * uu[i] * bb[i];
          PB = fmaf(a[i], PB, bu[i]);
          PA *= a[i];
        }
        wave_scan_prefix(PA, PB, lane);
        float EA = __shfl_up(PA, 1, WAVE), EB = __shfl_up(PB, 1, WAVE);
        if (lane == 0) { EA = 1.f; EB = 0.f; }
        const float h0 = (c == 0) ? 0.f : hstate[(row * nchunk + (c - 1)) * NS + n];
        const float hin = fmaf(EA, h0, EB);  // h_{t-1} of this lane's first step
        float hp = hin;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) { hp = fmaf(a[i], hp, bu[i]); hh[i] = hp; }
        // ---- dL/dh suffix scan:  gh_i = cc_i*g_i + a_{i+1} * gh_{i+1}
        float a_next = __shfl_down(a[0], 1, WAVE);  // a of the next lane's first step
        if (lane == WAVE - 1) a_next = 1.f;          // the chunk's last step takes `carry` (= a*gh of the next chunk)
        float al[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS - 1; ++i) al[i] = a[i + 1];
        al[ITEMS - 1] = a_next;
        float SA = 1.f, SB = 0.f;
#pragma unroll
        for (int i = ITEMS - 1; i >= 0; --i) { SB = fmaf(al[i], SB, cc[i] * g[i]); SA *= al[i]; }
        wave_scan_suffix(SA, SB, lane);
        float XA = __shfl_down(SA, 1, WAVE), XB = __shfl_down(SB, 1, WAVE);
        if (lane == WAVE - 1) { XA = 1.f; XB = 0.f; }
        float gh = fmaf(XA, carry[n], XB);  // gh of the step right after this lane's last one (already times its a)
        float dBv[ITEMS], dCv[ITEMS];
#pragma unroll
        for (int i = ITEMS - 1; i >= 0; --i) {
          gh = fmaf(al[i], gh, cc[i] * g[i]);  // dL/dh_t
          const float hprev = (i == 0) ? hin : hh[i - 1];
          const float da = gh * hprev * a[i];  // dL/d(dt*A) through a = exp(dt*A)
          dA_n = fmaf(da, dt[i], dA_n);
          ddt[i] = fmaf(da, An_n, ddt[i]);
          ddt[i] = fmaf(gh * uu[i], bb[i], ddt[i]);
          du[i] = fmaf(gh * dt[i], bb[i], du[i]);
          dBv[i] = gh * dt[i] * uu[i];
          dCv[i] = g[i] * hh[i];
        }
        if (lane == 0) carry[n] = a[0] * gh;  // a_t * gh_t of this chunk's first step, for the previous chunk
        s_dA[wave][n][lane] += dA_n;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
          atomicAdd(&s_dB[n][lane * ITEMS + i], dBv[i]);
          atomicAdd(&s_dC[n][lane * ITEMS + i], dCv[i]);
        }
      }
      float gd[ITEMS];
#pragma unroll
      for (int i = 0; i < ITEMS; ++i) {
        gd[i] = (t + i < L) ? ddt[i] * sigmoid_f(dl[i] + bias) : 0.f;
        dbs += gd[i];
      }
      store4<VEC>(gu + row * L, t, L, du);
      store4<VEC>(gdelta + row * L, t, L, gd);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NS * CHUNK; i += BWD_ROWS * WAVE) {
      const int n = i / CHUNK, tt = c * CHUNK + (i % CHUNK);
      if (tt < L) {
        atomicAdd(gB + ((size_t)bk * NS + n) * L + tt, (&s_dB[0][0])[i]);
        atomicAdd(gC + ((size_t)bk * NS + n) * L + tt, (&s_dC[0][0])[i]);
      }
    }
    __syncthreads();
  }
  if (live) {
    for (int n = 0; n < NS; ++n) {
      const float s = group_sum<WAVE>(s_dA[wave][n][lane]);
      if (lane == 0) atomicAdd(gA + (size_t)kd * NS + n, s);
    }
    dD = group_sum<WAVE>(dD);
    dbs = group_sum<WAVE>(dbs);
    if (lane == 0) { atomicAdd(gD + kd, dD); atomicAdd(gdbias + kd, dbs); }
  }
}

}  // namespace

extern "C" int tamtr_selective_scan_chunk(void) { return CHUNK; }

extern "C" int tamtr_selective_scan_fwd(const float* u, const float* delta, const float* A, const float* Bm, const float* Cm,
                                        const float* D, const float* dbias, float* y, float* hstate, int B, int K, int Dk, int N,
                                        int L, void* stream) {
  if (!u || !delta || !A || !Bm || !Cm || !D || !dbias || !y || !hstate || B <= 0 || K <= 0 || Dk <= 0 || L <= 0)
    return TAMTR_EINVAL;
  if (N != NS) return TAMTR_EUNSUP;
  const long long rows = (long long)B * K * Dk;
  if (rows > (1ll << 30)) return TAMTR_EUNSUP;
  const int nchunk = (L + CHUNK - 1) / CHUNK;
  dim3 grid((unsigned)((rows + FWD_ROWS - 1) / FWD_ROWS));
  hipStream_t s = (hipStream_t)stream;
  if (L % 4 == 0)
    hipLaunchKernelGGL(selscan_fwd_kernel<true>, grid, dim3(FWD_ROWS * WAVE), 0, s, u, delta, A, Bm, Cm, D, dbias, y, hstate,
                       (int)rows, K, Dk, L, nchunk);
  else
    hipLaunchKernelGGL(selscan_fwd_kernel<false>, grid, dim3(FWD_ROWS * WAVE), 0, s, u, delta, A, Bm, Cm, D, dbias, y, hstate,
                       (int)rows, K, Dk, L, nchunk);
  return tamtr_launch_status();
}

extern "C" int tamtr_selective_scan_bwd(const float* gy, const float* u, const float* delta, const float* A, const float* Bm,
                                        const float* Cm, const float* D, const float* dbias, const float* hstate, float* gu,
                                        float* gdelta, float* gA, float* gB, float* gC, float* gD, float* gdbias, int B, int K,
                                        int Dk, int N, int L, void* stream) {
  if (!gy || !u || !delta || !A || !Bm || !Cm || !D || !dbias || !hstate || !gu || !gdelta || !gA || !gB || !gC || !gD ||
      !gdbias || B <= 0 || K <= 0 || Dk <= 0 || L <= 0)
    return TAMTR_EINVAL;
  if (N != NS || (long long)B * K > 65535) return TAMTR_EUNSUP;
  const int nchunk = (L + CHUNK - 1) / CHUNK;
  dim3 grid((Dk + BWD_ROWS - 1) / BWD_ROWS, B * K);
  hipStream_t s = (hipStream_t)stream;
  if (L % 4 == 0)
    hipLaunchKernelGGL(selscan_bwd_kernel<true>, grid, dim3(BWD_ROWS * WAVE), 0, s, gy, u, delta, A, Bm, Cm, D, dbias, hstate, gu,
                       gdelta, gA, gB, gC, gD, gdbias, K, Dk, L, nchunk);
  else
    hipLaunchKernelGGL(selscan_bwd_kernel<false>, grid, dim3(BWD_ROWS * WAVE), 0, s, gy, u, delta, A, Bm, Cm, D, dbias, hstate, gu,
                       gdelta, gA, gB, gC, gD, gdbias, K, Dk, L, nchunk);
  return tamtr_launch_status();
}
