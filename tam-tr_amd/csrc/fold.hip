// fold.hip - gradient of the two stored cross-scan copies, SS2D backward, gfx950.
//
// The scan backward returns d/d(u) per direction, un-reversed: g4 [B, 4, D, L].  Directions k and k + 2 read the same stored copy
// (csms6s.py:4-14), and the x_proj products read it too, so d/d(u2)[b, i] = g4[b, i] + g4[b, i + 2] + m_i[b] with m_i = W_i^T gx_i
// (vmamba.py:962-970 backward), i = 0, 1.  As separate torch ops that was a slice add, a bf16 -> fp32 cast of m and autograd's
// accumulation: 5 reads + 3 writes of an 839 MB plane pair at level 0; here 3 reads + 1 write, 16 bytes per lane.
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void fold_add_kernel(const float* __restrict__ g4, const T* __restrict__ m0, const T* __restrict__ m1,
                                                       float* __restrict__ out, size_t plane4, int B) {
  // plane4 = D * L / 4 float4 groups of one [D, L] plane; grid.y = 2 * B
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= plane4) return;
  const int b = blockIdx.y / 2, c = blockIdx.y % 2;
  const float4 a = reinterpret_cast<const float4*>(g4)[((size_t)b * 4 + c) * plane4 + i];
  const float4 d = reinterpret_cast<const float4*>(g4)[((size_t)b * 4 + c + 2) * plane4 + i];
  float mv[4];
  Elt<T>::ld4((c ? m1 : m0) + ((size_t)b * plane4 + i) * 4, mv);
  reinterpret_cast<float4*>(out)[((size_t)b * 2 + c) * plane4 + i] =
      make_float4(a.x + d.x + mv[0], a.y + d.y + mv[1], a.z + d.z + mv[2], a.w + d.w + mv[3]);
}

// out = a0 + a1 + ... + a(n-1), n <= 8 tensors of one shape (the gradients that reach a tensor with several consumers: autograd adds
// them pairwise, n - 1 kernels of 2 reads + 1 write each; here n reads + 1 write)
struct SumArgs {
  const void* p[8];
};
// N is a template parameter: all N loads of a thread are unconditional and in flight together (a load under `if (k < n)` is followed by
// s_waitcnt vmcnt(0): the first form of this kernel moved the token memory's four 550 MB gradients at 5.7 TB/s, this one at 6.0);
// two groups of 4 elements per thread.
template <typename T, int N>
__global__ __launch_bounds__(256) void sum_n_kernel(SumArgs a, T* __restrict__ out, size_t n4) {
  const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (i0 >= n4) return;
  const size_t i1 = i0 + 1 < n4 ? i0 + 1 : i0;   // (odd tail: the second group repeats the first; the duplicate store writes the same value)
  float v[N][2][4];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    Elt<T>::ld4(reinterpret_cast<const T*>(a.p[k]) + i0 * 4, v[k][0]);
    Elt<T>::ld4(reinterpret_cast<const T*>(a.p[k]) + i1 * 4, v[k][1]);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < N; ++k) { acc[0] += v[k][h][0]; acc[1] += v[k][h][1]; acc[2] += v[k][h][2]; acc[3] += v[k][h][3]; }
    Elt<T>::st4(out + (h ? i1 : i0) * 4, acc);
  }
}

// Column sums of a tall bf16 matrix X [M, N] (the bias gradient of a token-wise Linear: db = sum over the B*L = 537 600 tokens of dY,
// transformer.py:273 / head.py:1118 backward): partial[blk][n] = sum over the block's rows, fp32.  A thread owns 8 adjacent columns (16 B
// per row), a wave one full 1 KB row at N = 512, the 256 threads of a workgroup N / 8 column groups x 2048 / N rows at a time, four rows in
// flight per thread.  torch's generic reduce_kernel runs this shape at 2.9 TB/s (190 us for 550 MB); the caller adds the <= 2 048 partials.
constexpr int CS_MAXBLK = 2048;

__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ X, float* __restrict__ partial, long long M, int N,
                                                          int rows_per_blk) {
  __shared__ float red[256][9];
  const int cgs = N / 8;              // column groups (<= 256)
  const int lanes = 256 / cgs;        // rows handled at a time
  const int cg = threadIdx.x % cgs, rl = threadIdx.x / cgs;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const long long r0 = (long long)blockIdx.x * rows_per_blk, r1 = min(M, r0 + rows_per_blk);
  if (rl < lanes) {
    long long r = r0 + rl;
    for (; r + 3 * lanes < r1; r += 4 * lanes) {
      uint4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const uint4*>(X + (r + (long long)j * lanes) * N + cg * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[2 * i] += __uint_as_float(w[i] << 16); acc[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u); }
      }
    }
    for (; r < r1; r += lanes) {
      const uint4 v = *reinterpret_cast<const uint4*>(X + r * N + cg * 8);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc[2 * i] += __uint_as_float(w[i] << 16); acc[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u); }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[threadIdx.x][i] = acc[i];
  __syncthreads();
  if (threadIdx.x < cgs) {   // the row lanes of a column group, added in lane order (a fixed order: reproducible)
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = 0.f;
    for (int l = 0; l < lanes; ++l)
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] += red[l * cgs + threadIdx.x][i];
    float* o = partial + (size_t)blockIdx.x * N + threadIdx.x * 8;
    *reinterpret_cast<float4*>(o) = make_float4(t[0], t[1], t[2], t[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(t[4], t[5], t[6], t[7]);
  }
}


// out[c] = sum over r of in[r][c], r = 0 .. R-1: the LAST stage of every two-stage reduction of this package (per-workgroup partial
// rows of d(gamma) / d(beta) / d(weight), per-image rows of the scan's d(A) / d(D), per-slice products of a split-K weight gradient),
// in one launch and in a FIXED order (a thread adds its rows r = rg, rg + RG, ... in sequence, then the RG row groups are added in
// index order through LDS): bitwise reproducible, no atomics, no semaphore.  Replaces torch's `partials.sum(0)`, whose multi-workgroup
// form counts arrivals in a semaphore buffer that it zeroes with a memset node before every launch - the one node kind that does not
// survive a HIP-graph replay under the runtime's AQL packet capture (profiles/r04_packet_capture_bisect.txt).
// Two shapes of one kernel: R <= 32 (many columns, few rows: split-K slabs, per-image rows): RG = 1, a thread streams R rows of its
// 4 columns; larger R (few columns, up to a few thousand partial rows): 1024 threads = CG column groups x RG row groups.
template <typename T>
__global__ __launch_bounds__(1024) void slab_sum_rows_kernel(const T* __restrict__ in, float* __restrict__ out, int R, long long C, int CG, int RG) {
  __shared__ float4 red[1024];
  const int cg = threadIdx.x % CG, rg = threadIdx.x / CG;
  const long long c4 = (long long)blockIdx.x * CG + cg;      // group of 4 columns
  const bool live = c4 * 4 < C;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    const T* p = in + c4 * 4;
    int r = rg;
    for (; r + 3 * RG < R; r += 4 * RG) {   // four rows in flight
      float v[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) Elt<T>::ld4(p + (long long)(r + j * RG) * C, v[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc.x += v[j][0]; acc.y += v[j][1]; acc.z += v[j][2]; acc.w += v[j][3]; }
    }
    for (; r < R; r += RG) {
      float v[4];
      Elt<T>::ld4(p + (long long)r * C, v);
      acc.x += v[0]; acc.y += v[1]; acc.z += v[2]; acc.w += v[3];
    }
  }
  if (RG == 1) {
    if (live) *reinterpret_cast<float4*>(out + c4 * 4) = acc;
    return;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (rg == 0 && live) {
    float4 t = red[cg];
    for (int g = 1; g < RG; ++g) {
      const float4 u = red[g * CG + cg];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    *reinterpret_cast<float4*>(out + c4 * 4) = t;
  }
}

}  // namespace

extern "C" int tamtr_colsum_blocks(long long M) {
  if (M <= 0) return 0;
  const long long b = (M + 63) / 64;
  return (int)(b < CS_MAXBLK ? b : CS_MAXBLK);
}

// X bf16 [M, N] row-major -> partial f32 [tamtr_colsum_blocks(M), N]: per-block column sums (the caller adds the blocks);
// N % 8 == 0, 8 <= N <= 2048, N a power of two times ... any N with 256 % (N / 8) == 0
extern "C" int tamtr_colsum_bf16(const void* X, float* partial, long long M, int N, void* stream) {
  if (!X || !partial || M <= 0 || N <= 0) return TAMTR_EINVAL;
  if (N % 8 || N > 2048 || 256 % (N / 8) || (uintptr_t)X % 16 || (uintptr_t)partial % 16) return TAMTR_EUNSUP;
  const int nblk = tamtr_colsum_blocks(M);
  const int rpb = (int)((M + nblk - 1) / nblk);
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)X, partial, M, N, rpb);
  return tamtr_launch_status();
}

// in (T) [R, C] row-major -> out f32 [C] = the sum of the R rows, added in a fixed order (see slab_sum_rows_kernel); C % 4 == 0
extern "C" int tamtr_slab_sum_rows(const void* in, float* out, int R, long long C, int dtype, void* stream) {
  if (!in || !out || R <= 0 || C <= 0) return TAMTR_EINVAL;
  if (dtype != TAMTR_F32 && dtype != TAMTR_BF16) return TAMTR_EINVAL;
  if (C % 4 || (uintptr_t)in % (dtype == TAMTR_F32 ? 16 : 8) || (uintptr_t)out % 16) return TAMTR_EUNSUP;
  const long long c4 = C / 4;
  int CG, RG;
  if (R <= 32) { CG = 256; RG = 1; }
  else {
    CG = 64;
    while (CG > 1 && CG / 2 >= c4) CG /= 2;     // fewer column groups than 64: more row groups per workgroup
    RG = 1024 / CG;
  }
  const long long blocks = (c4 + CG - 1) / CG;
  if (blocks > 0x7fffffffLL) return TAMTR_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TAMTR_F32)
    hipLaunchKernelGGL(slab_sum_rows_kernel<float>, dim3((unsigned)blocks), dim3(CG * RG), 0, s, (const float*)in, out, R, C, CG, RG);
  else
    hipLaunchKernelGGL(slab_sum_rows_kernel<bf16_t>, dim3((unsigned)blocks), dim3(CG * RG), 0, s, (const bf16_t*)in, out, R, C, CG, RG);
  return tamtr_launch_status();
}

// out (T) [n_elems] = sum of the n <= 8 tensors src[k] (T, same length; src is a HOST array of device pointers); n_elems % 4 == 0
extern "C" int tamtr_sum_n(const void* const* src, int n, void* out, long long n_elems, int dtype, void* stream) {
  if (!src || !out || n < 1 || n > 8 || n_elems <= 0) return TAMTR_EINVAL;
  if (dtype != TAMTR_F32 && dtype != TAMTR_BF16) return TAMTR_EINVAL;
  const int al = dtype == TAMTR_F32 ? 16 : 8;
  if (n_elems % 4 || (uintptr_t)out % al) return TAMTR_EUNSUP;
  SumArgs a;
  for (int k = 0; k < 8; ++k) {
    a.p[k] = src[k < n ? k : 0];
    if (k < n && (!src[k] || (uintptr_t)src[k] % al)) return src[k] ? TAMTR_EUNSUP : TAMTR_EINVAL;
  }
  const size_t n4 = (size_t)n_elems / 4;
  const dim3 grid((unsigned)((n4 + 511) / 512));
  hipStream_t s = (hipStream_t)stream;
#define SUM_N(N)                                                                                                           \
  case N:                                                                                                                  \
    if (dtype == TAMTR_F32) hipLaunchKernelGGL((sum_n_kernel<float, N>), grid, dim3(256), 0, s, a, (float*)out, n4);       \
    else hipLaunchKernelGGL((sum_n_kernel<bf16_t, N>), grid, dim3(256), 0, s, a, (bf16_t*)out, n4);                        \
    break;
  switch (n) { SUM_N(1) SUM_N(2) SUM_N(3) SUM_N(4) SUM_N(5) SUM_N(6) SUM_N(7) default: SUM_N(8) }
#undef SUM_N
  return tamtr_launch_status();
}

// g4 f32 [B, 4, D*L]; m0, m1 (T) [B, D*L]; out f32 [B, 2, D*L];  n = D*L, n % 4 == 0
extern "C" int tamtr_fold_add(const float* g4, const void* m0, const void* m1, float* out, int B, long long n, int dtype, void* stream) {
  if (!g4 || !m0 || !m1 || !out || B <= 0 || n <= 0) return TAMTR_EINVAL;
  if (dtype != TAMTR_F32 && dtype != TAMTR_BF16) return TAMTR_EINVAL;
  if (n % 4 || 2 * B > 65535 || ((uintptr_t)m0 | (uintptr_t)m1) % 8) return TAMTR_EUNSUP;
  const size_t plane4 = (size_t)n / 4;
  const dim3 grid((unsigned)((plane4 + 255) / 256), 2 * B);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TAMTR_F32) hipLaunchKernelGGL(fold_add_kernel<float>, grid, dim3(256), 0, s, g4, (const float*)m0, (const float*)m1, out, plane4, B);
  else hipLaunchKernelGGL(fold_add_kernel<bf16_t>, grid, dim3(256), 0, s, g4, (const bf16_t*)m0, (const bf16_t*)m1, out, plane4, B);
  return tamtr_launch_status();
}
