// layout.hip - NCHW <-> NHWC repacking of a feature map, gfx950.
//
// The trunk runs channels-last (MIOpen's bf16 convolutions are NHWC kernels); the gate and CPAM kernels (gate.hip, cpam.hip) are
// written for NCHW planes.  torch's generic strided copy moved the [16, 128, 160, 160] map of CPAM site 33 at 0.4 TB/s (262 us per
// direction, four times per step).  This is the plain tiled transpose of the per-image [C, HW] matrix: a 64 x 64 tile goes through
// LDS, global reads run along the source's contiguous axis and global writes along the destination's, 8 bytes (bf16) / 16 bytes
// (fp32) per lane.  HBM-bound: 2 x map bytes.
#include "common.h"

namespace {

constexpr int LT = 64;          // tile edge
constexpr int L_THREADS = 256;

// src: [B][R][S] (S contiguous, row pitch lds)  ->  dst: [B][S][R] (R contiguous, row pitch ldd).  grid: (ceil(S / 64), ceil(R / 64), B)
template <typename E>
__global__ __launch_bounds__(L_THREADS) void transpose_tiles_kernel(const E* __restrict__ src, E* __restrict__ dst, int R, int S, int lds,
                                                                    int ldd, int vec) {
  __shared__ E tile[LT][LT + 4 / sizeof(E) + 1];
  const int s0 = blockIdx.x * LT, r0 = blockIdx.y * LT;
  const size_t splane = (size_t)blockIdx.z * R * lds, dplane = (size_t)blockIdx.z * S * ldd;
  const int q = (threadIdx.x % 16) * 4, row = threadIdx.x / 16;
  // read: 16 lanes x 4 elements along S per source row, 16 rows per pass
#pragma unroll
  for (int k = 0; k < LT; k += 16) {
    const int r = r0 + row + k, s = s0 + q;
    if (r < R) {
      const E* p = src + splane + (size_t)r * lds + s;
      if (vec && s + 3 < S) {
        E v[4];
        __builtin_memcpy(v, __builtin_assume_aligned(p, 4 * sizeof(E)), 4 * sizeof(E));
#pragma unroll
        for (int i = 0; i < 4; ++i) tile[row + k][q + i] = v[i];
      } else {
        for (int i = 0; i < 4; ++i) if (s + i < S) tile[row + k][q + i] = p[i];
      }
    }
  }
  __syncthreads();
  // write: 16 lanes x 4 elements along R per destination row
#pragma unroll
  for (int k = 0; k < LT; k += 16) {
    const int s = s0 + row + k, r = r0 + q;
    if (s < S) {
      E* p = dst + dplane + (size_t)s * ldd + r;
      if (vec && r + 3 < R) {
        E v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = tile[q + i][row + k];
        __builtin_memcpy(__builtin_assume_aligned(p, 4 * sizeof(E)), v, 4 * sizeof(E));
      } else {
        for (int i = 0; i < 4; ++i) if (r + i < R) p[i] = tile[q + i][row + k];
      }
    }
  }
}

// dst[r][0..C) = src[r][0..C) for N rows with row pitches lds / ldd (elements): the channel-slice copies around torch.cat / chunk on
// channels-last maps.  One 16-byte vector (or one element when C or the pitches do not allow it) per thread.
template <typename V>
__global__ __launch_bounds__(L_THREADS) void copy_rows_kernel(const V* __restrict__ src, V* __restrict__ dst, size_t n, int per_row, size_t lds,
                                                              size_t ldd) {
  const size_t i = (size_t)blockIdx.x * L_THREADS + threadIdx.x;
  if (i >= n) return;
  const size_t r = i / per_row, c = i % per_row;
  dst[r * ldd + c] = src[r * lds + c];
}

// nearest-neighbour x2 / x0.5 resampling of a packed NHWC map (nn.Upsample(scale_factor=2.0 | 0.5, mode='nearest'), TAMTR.yaml layers
// 11/14/19/22/27/30), forward and backward, one 16-byte channel vector per thread:
//   mode 0  up, forward    dst [B,2H,2W,C] <- src [B,H,W,C]:    dst[2h+i][2w+j] = src[h][w]
//   mode 1  up, backward   dst [B,H,W,C]   <- src [B,2H,2W,C]:  dst[h][w] = sum of the four
//   mode 2  down, forward  dst [B,H/2,W/2,C] <- src [B,H,W,C]:  dst[h][w] = src[2h][2w]
//   mode 3  down, backward dst [B,H,W,C]   <- src [B,H/2,W/2,C]: dst[2h][2w] = src[h][w], zero elsewhere (one pass, no fill + scatter)
// H, W are those of the map written [B,H,W,C] in the line of the mode.
template <typename T, int V>
__global__ __launch_bounds__(L_THREADS) void resample2_kernel(const T* __restrict__ src, T* __restrict__ dst, int H, int W, int cv, int mode,
                                                              size_t n) {
  const size_t i = (size_t)blockIdx.x * L_THREADS + threadIdx.x;   // one vector of the destination
  if (i >= n) return;
  const int c = (int)(i % cv);
  size_t r = i / cv;
  // destination geometry
  const int dw_ = (mode == 0) ? 2 * W : (mode == 2) ? W / 2 : W, dh_ = (mode == 0) ? 2 * H : (mode == 2) ? H / 2 : H;
  const int w = (int)(r % dw_); r /= dw_;
  const int h = (int)(r % dh_);
  const size_t b = r / dh_;
  typedef T vec_t __attribute__((ext_vector_type(V)));
  const vec_t* s = reinterpret_cast<const vec_t*>(src);
  vec_t* d = reinterpret_cast<vec_t*>(dst);
  if (mode == 0) {
    d[i] = s[((b * H + h / 2) * W + w / 2) * cv + c];
  } else if (mode == 2) {
    d[i] = s[((b * H + 2 * h) * W + 2 * w) * cv + c];
  } else if (mode == 3) {
    vec_t z;
#pragma unroll
    for (int u = 0; u < V; ++u) z[u] = 0;
    d[i] = ((h | w) & 1) ? z : s[((b * (H / 2) + h / 2) * (W / 2) + w / 2) * cv + c];
  } else {
    float acc[V];
#pragma unroll
    for (int u = 0; u < V; ++u) acc[u] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const vec_t v = s[((b * 2 * H + 2 * h + q / 2) * (2 * W) + 2 * w + q % 2) * cv + c];
#pragma unroll
      for (int u = 0; u < V; ++u) acc[u] += sizeof(T) == 2 ? bf2f((bf16_t)v[u]) : __builtin_bit_cast(float, (uint32_t)v[u]);
    }
    vec_t o;
#pragma unroll
    for (int u = 0; u < V; ++u) o[u] = sizeof(T) == 2 ? (T)f2bf(acc[u]) : (T)__builtin_bit_cast(uint32_t, acc[u]);
    d[i] = o;
  }
}

}  // namespace

// see resample2_kernel; [B,H,W,C] is the map the mode's comment names that way.  C % 8 == 0 (bf16) / C % 4 == 0 (f32), 16-byte aligned.
extern "C" int tamtr_resample2(const void* src, void* dst, int B, int H, int W, int C, int mode, int dtype, void* stream) {
  if (!src || !dst || B <= 0 || H <= 0 || W <= 0 || C <= 0 || mode < 0 || mode > 3) return TAMTR_EINVAL;
  if (dtype != TAMTR_F32 && dtype != TAMTR_BF16) return TAMTR_EINVAL;
  const int v = dtype == TAMTR_F32 ? 4 : 8;
  if (C % v || ((uintptr_t)src | (uintptr_t)dst) % 16 || ((mode == 2 || mode == 3) && ((H | W) & 1))) return TAMTR_EUNSUP;
  const size_t rows = mode == 0 ? (size_t)B * 2 * H * 2 * W : mode == 2 ? (size_t)B * (H / 2) * (W / 2) : (size_t)B * H * W;
  const size_t n = rows * (C / v);
  const unsigned blocks = (unsigned)((n + L_THREADS - 1) / L_THREADS);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TAMTR_F32)
    hipLaunchKernelGGL((resample2_kernel<uint32_t, 4>), dim3(blocks), dim3(L_THREADS), 0, s, (const uint32_t*)src, (uint32_t*)dst, H, W, C / v, mode, n);
  else
    hipLaunchKernelGGL((resample2_kernel<uint16_t, 8>), dim3(blocks), dim3(L_THREADS), 0, s, (const uint16_t*)src, (uint16_t*)dst, H, W, C / v, mode, n);
  return tamtr_launch_status();
}

namespace {

// up to four row-copies into column slices of one destination in one launch (blockIdx.y = which): the inputs of a torch.cat(..., 1)
struct CatArgs {
  const void* src[4];
  unsigned long long lds[4];   // source row pitch, in 16-byte vectors
  unsigned int per_row[4];     // vectors per row
  unsigned int off[4];         // first destination vector of the slice
};
__global__ __launch_bounds__(L_THREADS) void cat_rows_kernel(CatArgs a, uint4* __restrict__ dst, size_t ldd, size_t N) {
  const int k = blockIdx.y;
  const size_t i = (size_t)blockIdx.x * L_THREADS + threadIdx.x, per = a.per_row[k];
  if (i >= N * per) return;
  const size_t r = i / per, c = i - r * per;
  dst[r * ldd + a.off[k] + c] = reinterpret_cast<const uint4*>(a.src[k])[r * a.lds[k] + c];
}

}  // namespace

// dst[r][off_k .. off_k + C_k) = src_k[r][0 .. C_k) for k < n <= 4 inputs: N rows, dst row pitch ldd elements, source pitches lds[k];
// all widths, pitches and offsets multiples of 16 bytes, 16-byte aligned pointers (else TAMTR_EUNSUP: use tamtr_copy_rows per input)
extern "C" int tamtr_cat_rows(const void* const* src, const long long* lds, const int* C, int n, void* dst, long long ldd, long long N, int dtype,
                              void* stream) {
  if (!src || !lds || !C || !dst || n < 1 || n > 4 || N <= 0) return TAMTR_EINVAL;
  if (dtype != TAMTR_F32 && dtype != TAMTR_BF16) return TAMTR_EINVAL;
  const int v = dtype == TAMTR_F32 ? 4 : 8;
  CatArgs a;
  unsigned off = 0, widest = 0;
  if (ldd % v || (uintptr_t)dst % 16) return TAMTR_EUNSUP;
  for (int k = 0; k < 4; ++k) {
    const int kk = k < n ? k : 0;
    if (k < n && (!src[k] || C[k] <= 0 || lds[k] < C[k])) return TAMTR_EINVAL;
    if (k < n && (C[k] % v || lds[k] % v || (uintptr_t)src[k] % 16)) return TAMTR_EUNSUP;
    a.src[k] = src[kk]; a.lds[k] = (unsigned long long)(lds[kk] / v); a.per_row[k] = k < n ? (unsigned)(C[k] / v) : 0u; a.off[k] = off;
    if (k < n) { off += (unsigned)(C[k] / v); widest = a.per_row[k] > widest ? a.per_row[k] : widest; }
  }
  if ((long long)off * v > ldd) return TAMTR_EINVAL;
  const size_t items = (size_t)N * widest;
  hipLaunchKernelGGL(cat_rows_kernel, dim3((unsigned)((items + L_THREADS - 1) / L_THREADS), n), dim3(L_THREADS), 0, (hipStream_t)stream, a,
                     (uint4*)dst, (size_t)ldd / v, (size_t)N);
  return tamtr_launch_status();
}

// N rows of C elements, source / destination row pitch lds / ldd elements (>= C).  T = f32 | bf16 (any 2- or 4-byte element).
extern "C" int tamtr_copy_rows(const void* src, long long lds, void* dst, long long ldd, long long N, int C, int dtype, void* stream) {
  if (!src || !dst || N <= 0 || C <= 0 || lds < C || ldd < C) return TAMTR_EINVAL;
  if (dtype != TAMTR_F32 && dtype != TAMTR_BF16) return TAMTR_EINVAL;
  const int e = dtype == TAMTR_F32 ? 4 : 2, v = 16 / e;
  hipStream_t s = (hipStream_t)stream;
  const bool vec = C % v == 0 && lds % v == 0 && ldd % v == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0;
  if (vec) {
    const size_t n = (size_t)N * (C / v);
    hipLaunchKernelGGL(copy_rows_kernel<uint4>, dim3((unsigned)((n + L_THREADS - 1) / L_THREADS)), dim3(L_THREADS), 0, s, (const uint4*)src,
                       (uint4*)dst, n, C / v, (size_t)lds / v, (size_t)ldd / v);
  } else if (e == 4) {
    const size_t n = (size_t)N * C;
    hipLaunchKernelGGL(copy_rows_kernel<uint32_t>, dim3((unsigned)((n + L_THREADS - 1) / L_THREADS)), dim3(L_THREADS), 0, s, (const uint32_t*)src,
                       (uint32_t*)dst, n, C, (size_t)lds, (size_t)ldd);
  } else {
    const size_t n = (size_t)N * C;
    hipLaunchKernelGGL(copy_rows_kernel<uint16_t>, dim3((unsigned)((n + L_THREADS - 1) / L_THREADS)), dim3(L_THREADS), 0, s, (const uint16_t*)src,
                       (uint16_t*)dst, n, C, (size_t)lds, (size_t)ldd);
  }
  return tamtr_launch_status();
}

// to_nhwc != 0: src [B, C, HW] -> dst [B, HW, C];  to_nhwc == 0: src [B, HW, C] -> dst [B, C, HW].  The NHWC side may be a channel
// slice of a wider map: its pixel pitch is ld >= C elements (ld == C: packed).
extern "C" int tamtr_relayout(const void* src, void* dst, int B, int C, int HW, int ld, int to_nhwc, int dtype, void* stream) {
  if (!src || !dst || B <= 0 || C <= 0 || HW <= 0 || ld < C) return TAMTR_EINVAL;
  if (dtype != TAMTR_F32 && dtype != TAMTR_BF16) return TAMTR_EINVAL;
  const int R = to_nhwc ? C : HW, S = to_nhwc ? HW : C;
  const unsigned gx = (S + LT - 1) / LT, gy = (R + LT - 1) / LT;
  if (B > 65535 || gy > 65535) return TAMTR_EUNSUP;
  const int e = dtype == TAMTR_F32 ? 4 : 2;
  const int lds = to_nhwc ? HW : ld, ldd = to_nhwc ? ld : HW;
  const int vec = R % 4 == 0 && S % 4 == 0 && ld % 4 == 0 && ((uintptr_t)src % (4 * e)) == 0 && ((uintptr_t)dst % (4 * e)) == 0;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TAMTR_F32)
    hipLaunchKernelGGL(transpose_tiles_kernel<uint32_t>, dim3(gx, gy, B), dim3(L_THREADS), 0, s, (const uint32_t*)src, (uint32_t*)dst, R, S, lds, ldd, vec);
  else
    hipLaunchKernelGGL(transpose_tiles_kernel<uint16_t>, dim3(gx, gy, B), dim3(L_THREADS), 0, s, (const uint16_t*)src, (uint16_t*)dst, R, S, lds, ldd, vec);
  return tamtr_launch_status();
}
