// dwconv.hip - SS2D front end: depthwise 3x3 conv + bias + SiLU + cross-scan layout, forward and backward, gfx950.
//
// Reference: SS2D.forward_corev2 input path, ultralytics/nn/extra_modules/VManba/vmamba.py:949-952 (`x = self.act(self.conv2d(x))`
// on the NCHW permutation of the in_proj output) followed by CrossScan (csms6s.py:4-14).  Through PyTorch this is a
// permute+contiguous copy, MIOpen's *naive* depthwise kernels (0.94 ms forward, 1.59 + 0.94 ms backward at the 160x160
// level), SiLU, a dtype cast and two flattening copies - eight passes over the map forward, about as many backward.
// Here the channels-last in_proj output is read where it lies (xi = the first d_inner channels of each pixel's 2*d_inner row)
// and the two fp32 flattenings the scan kernels consume (row-major and column-major, [B, 2, D, H*W]) are written directly.
//
// Mapping: a workgroup owns a 16x16 pixel tile x CB channels.  The halo tile is staged in LDS channel-major
// ([c][y][x], row pitch odd) so that lanes running along x OR along y both read conflict-free; the conv is evaluated twice,
// once with lanes along x (row-major plane: 64-B row segments) and once with lanes along y (column-major plane), instead of
// transposing the result.  Backward: the two gradient planes are summed into an LDS tile (each read along its own
// contiguous axis), multiplied by SiLU'(conv) recomputed from the staged input, and then used three ways from LDS:
// transposed 3x3 for d(input) (written back channels-last), per-(channel, tap) sums for d(weight), per-channel sums for d(bias)
// (partials per tile, summed by the caller: no atomics).
#include "common.h"

namespace {

constexpr int TS = 16;            // tile side (pixels)
constexpr int DW_THREADS = TS * TS;
#ifndef DW_CB_BWD
#define DW_CB_BWD 16
#endif
constexpr int CB_FWD = 32, CB_BWD = DW_CB_BWD;

__device__ __forceinline__ float silu_f(float v) { return v / (1.f + __expf(-v)); }

// 16 bytes of consecutive channels of one pixel -> floats
template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void ld(const float* p, float (&o)[4]) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <>
struct Vec16<bf16_t> {
  static constexpr int N = 8;
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

// one element of a cross-scan plane (u2 / d(u2): f32, or bf16 in the bf16 mode - see include/tamtr_hip.h, "bf16 PLANES")
__device__ __forceinline__ void plane_st(float* p, float v) { *p = v; }
__device__ __forceinline__ void plane_st(bf16_t* p, float v) { *p = f2bf(v); }
__device__ __forceinline__ float plane_ld(const float* p) { return *p; }
__device__ __forceinline__ float plane_ld(const bf16_t* p) { return bf2f(*p); }
// four consecutive elements (16- / 8-byte aligned)
__device__ __forceinline__ float4 plane_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 plane_ld4(const bf16_t* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}

// stage the (TS + 2*HALO)^2 x CB halo tile of x (channels-last, pixel stride xs elements) into s[c][y][x], zero outside the image
template <typename T, int CB, int HALO, int PITCH>
__device__ __forceinline__ void stage_tile(const T* __restrict__ x, size_t xs, int b, int d0, int h0, int w0, int H, int W,
                                           float (*s)[TS + 2 * HALO][PITCH]) {
  constexpr int SIDE = TS + 2 * HALO, VN = Vec16<T>::N, GROUPS = CB / VN, TOTAL = SIDE * SIDE * GROUPS;
  constexpr int N_IT = (TOTAL + DW_THREADS - 1) / DW_THREADS;
  // all of a thread's 16-byte loads first (clamped addresses; pixels outside the image are zeroed afterwards): a load inside a
  // conditional block is followed by s_waitcnt vmcnt(0), i.e. one memory round trip per iteration
  float v[N_IT][VN];
#pragma unroll
  for (int k = 0; k < N_IT; ++k) {
    const int it = min((int)threadIdx.x + k * DW_THREADS, TOTAL - 1);
    const int pix = it / GROUPS, gq = it - pix * GROUPS;
    const int py = pix / SIDE, px = pix - py * SIDE;
    const int h = min(max(h0 + py - HALO, 0), H - 1), w = min(max(w0 + px - HALO, 0), W - 1);
    Vec16<T>::ld(x + ((size_t)(b * H + h) * W + w) * xs + d0 + gq * VN, v[k]);
  }
#pragma unroll
  for (int k = 0; k < N_IT; ++k) {
    const int it = threadIdx.x + k * DW_THREADS;
    if (it < TOTAL) {
      const int pix = it / GROUPS, gq = it - pix * GROUPS;
      const int py = pix / SIDE, px = pix - py * SIDE;
      const int h = h0 + py - HALO, w = w0 + px - HALO;
      const float keep = (h >= 0 && h < H && w >= 0 && w < W) ? 1.f : 0.f;
#pragma unroll
      for (int j = 0; j < VN; ++j) s[gq * VN + j][py][px] = v[k][j] * keep;
    }
  }
}

template <typename T, typename PT>
__global__ __launch_bounds__(DW_THREADS) void dwconv_cross_fwd_kernel(const T* __restrict__ x, size_t xs, const float* __restrict__ wgt,
                                                                       const float* __restrict__ bias, PT* __restrict__ u2, int D,
                                                                       int H, int W, int tiles_w) {
  constexpr int CB = CB_FWD, SIDE = TS + 2, PITCH = SIDE + 1;
  __shared__ float s_in[CB][SIDE][PITCH];
  __shared__ float s_w[CB][10];
  const int th = blockIdx.x / tiles_w, tw = blockIdx.x - th * tiles_w;
  const int d0 = blockIdx.y * CB, b = blockIdx.z, h0 = th * TS, w0 = tw * TS;
  const size_t L = (size_t)H * W;
  for (int i = threadIdx.x; i < CB * 10; i += DW_THREADS) {
    const int c = i / 10, k = i - c * 10;
    s_w[c][k] = k < 9 ? wgt[(size_t)(d0 + c) * 9 + k] : (bias ? bias[d0 + c] : 0.f);
  }
  stage_tile<T, CB, 1, PITCH>(x, xs, b, d0, h0, w0, H, W, s_in);
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    // pass 0: lanes along x -> row-major plane; pass 1: lanes along y -> column-major plane
    const int ty = pass == 0 ? threadIdx.x / TS : threadIdx.x % TS;
    const int tx = pass == 0 ? threadIdx.x % TS : threadIdx.x / TS;
    const int h = h0 + ty, w = w0 + tx;
    if (h >= H || w >= W) continue;
    PT* out = u2 + ((size_t)(b * 2 + pass) * D + d0) * L + (pass == 0 ? (size_t)h * W + w : (size_t)w * H + h);
    for (int c = 0; c < CB; ++c) {
      float acc = s_w[c][9];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc = fmaf(s_w[c][ky * 3 + kx], s_in[c][ty + ky][tx + kx], acc);
      plane_st(out + (size_t)c * L, silu_f(acc));
    }
  }
}

template <typename T, typename PT>
__global__ __launch_bounds__(DW_THREADS) void dwconv_cross_bwd_kernel(const PT* __restrict__ g2, const T* __restrict__ x, size_t xs,
                                                                       const float* __restrict__ wgt, const float* __restrict__ bias,
                                                                       T* __restrict__ gx, size_t gxs, float* __restrict__ ws, int D,
                                                                       int H, int W, int tiles_w, int tiles) {
  constexpr int CB = CB_BWD, XS = TS + 4, XP = XS + 1, GS = TS + 2, GP = GS + 1;
  __shared__ float s_x[CB][XS][XP];  // input, halo 2; reused as the d(input) staging tile at the end
  __shared__ float s_g[CB][GS][GP];  // dL/d(conv), halo 1
  __shared__ float s_w[CB][10];
  const int th = blockIdx.x / tiles_w, tw = blockIdx.x - th * tiles_w;
  const int d0 = blockIdx.y * CB, b = blockIdx.z, h0 = th * TS, w0 = tw * TS;
  const size_t L = (size_t)H * W;
  for (int i = threadIdx.x; i < CB * 10; i += DW_THREADS) {
    const int c = i / 10, k = i - c * 10;
    s_w[c][k] = k < 9 ? wgt[(size_t)(d0 + c) * 9 + k] : (bias ? bias[d0 + c] : 0.f);
  }
  // (DW_ABL_*: timing-only builds of tools/build_variant.sh dwconv; their results are wrong)
#ifndef DW_ABL_NOX
  stage_tile<T, CB, 2, XP>(x, xs, b, d0, h0, w0, H, W, s_x);
#endif
  // gradient tile = row-major plane (lanes along x) + column-major plane (lanes along y: its contiguous axis).  Loads go out in
  // batches of 7 with clamped addresses and a 0/1 factor instead of a branch: hipcc ends a conditional block that contains a load with
  // s_waitcnt vmcnt(0), which made a thread's 21 loads per plane 21 serial round trips (71 % of the wave cycles were waits, round 2 PMC).
#ifndef DW_ABL_NOG
  if ((W & 3) == 0 && (H & 3) == 0) {
    // Aligned 16-byte loads: a tile row of the gradient (18 values starting one pixel left of the tile) is covered by the 6 aligned float4
    // at [w0 - 4, w0 + 20); the scalar version below issued 21 dependent-free but narrow loads per thread and plane and moved the two planes
    // at 2 TB/s (420 us of the kernel's ~990 at level 0, measured by ablation: profiles/r03_dwconv_backward.txt).  7 loads per thread and
    // plane, all in flight together; values outside the image or outside the 18-wide window are dropped when they are written to LDS.
    constexpr int NV = 6, ITEMS_V = CB * GS * NV, N_ITV = (ITEMS_V + DW_THREADS - 1) / DW_THREADS;
#pragma unroll
    for (int plane = 0; plane < 2; ++plane) {
      const PT* gp = g2 + ((size_t)(b * 2 + plane) * D + d0) * L;
      float4 v[N_ITV];
#pragma unroll
      for (int k = 0; k < N_ITV; ++k) {
        const int it = min((int)threadIdx.x + k * DW_THREADS, ITEMS_V - 1);
        const int c = it / (GS * NV), r = it - c * (GS * NV), line = r / NV, q = r - line * NV;
        // plane 0: line = tile row (h), vectors along w;  plane 1 (column-major map): line = tile column (w), vectors along h
        const int fixed = (plane == 0 ? h0 : w0) + line - 1, base = (plane == 0 ? w0 : h0) - 4 + 4 * q;
        const int nfix = plane == 0 ? H : W, nrun = plane == 0 ? W : H;
        const bool ok = fixed >= 0 && fixed < nfix && base >= 0 && base + 3 < nrun;
        const float4 t = plane_ld4(gp + (size_t)c * L + (size_t)min(max(fixed, 0), nfix - 1) * nrun + min(max(base, 0), nrun - 4));
        const float f = ok ? 1.f : 0.f;
        v[k] = make_float4(t.x * f, t.y * f, t.z * f, t.w * f);
      }
      if (plane == 1) __syncthreads();   // plane 0 is complete in LDS before plane 1 is added onto it
#pragma unroll
      for (int k = 0; k < N_ITV; ++k) {
        const int it = threadIdx.x + k * DW_THREADS;
        if (it < ITEMS_V) {
          const int c = it / (GS * NV), r = it - c * (GS * NV), line = r / NV, q = r - line * NV;
          const float e[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int pos = 4 * q + j - 3;   // position inside the 18-wide window
            if (pos >= 0 && pos < GS) {
              if (plane == 0) s_g[c][line][pos] = e[j];
              else s_g[c][pos][line] += e[j];
            }
          }
        }
      }
    }
  } else {
  {
    constexpr int GB = 7, N_IT = ((CB * GS * GS + DW_THREADS - 1) / DW_THREADS + GB - 1) / GB * GB;   // whole batches (extra trips load a clamped item and write nothing)
    const PT* gp = g2 + ((size_t)(b * 2 + 0) * D + d0) * L;
#pragma unroll 1
    for (int k0 = 0; k0 < N_IT; k0 += GB) {
      float v[GB];
#pragma unroll
      for (int k = 0; k < GB; ++k) {
        const int it = min((int)threadIdx.x + (k0 + k) * DW_THREADS, CB * GS * GS - 1);
        const int c = it / (GS * GS), r = it - c * (GS * GS), py = r / GS, px = r - py * GS;
        const int h = h0 + py - 1, w = w0 + px - 1;
        const bool ok = h >= 0 && h < H && w >= 0 && w < W;
        v[k] = plane_ld(gp + (size_t)c * L + (size_t)min(max(h, 0), H - 1) * W + min(max(w, 0), W - 1)) * (ok ? 1.f : 0.f);
      }
#pragma unroll
      for (int k = 0; k < GB; ++k) {
        const int it = threadIdx.x + (k0 + k) * DW_THREADS;
        if (it < CB * GS * GS) {
          const int c = it / (GS * GS), r = it - c * (GS * GS), py = r / GS, px = r - py * GS;
          s_g[c][py][px] = v[k];
        }
      }
    }
  }
  __syncthreads();
  {
    constexpr int GB = 7, N_IT = ((CB * GS * GS + DW_THREADS - 1) / DW_THREADS + GB - 1) / GB * GB;   // whole batches (extra trips load a clamped item and write nothing)
    const PT* gp = g2 + ((size_t)(b * 2 + 1) * D + d0) * L;
#pragma unroll 1
    for (int k0 = 0; k0 < N_IT; k0 += GB) {
      float v[GB];
#pragma unroll
      for (int k = 0; k < GB; ++k) {
        const int it = min((int)threadIdx.x + (k0 + k) * DW_THREADS, CB * GS * GS - 1);
        const int c = it / (GS * GS), r = it - c * (GS * GS), px = r / GS, py = r - px * GS;
        const int h = h0 + py - 1, w = w0 + px - 1;
        const bool ok = h >= 0 && h < H && w >= 0 && w < W;
        v[k] = plane_ld(gp + (size_t)c * L + (size_t)min(max(w, 0), W - 1) * H + min(max(h, 0), H - 1)) * (ok ? 1.f : 0.f);
      }
#pragma unroll
      for (int k = 0; k < GB; ++k) {
        const int it = threadIdx.x + (k0 + k) * DW_THREADS;
        if (it < CB * GS * GS) {
          const int c = it / (GS * GS), r = it - c * (GS * GS), px = r / GS, py = r - px * GS;
          s_g[c][py][px] += v[k];
        }
      }
    }
  }
  }
#endif
  __syncthreads();
  // times SiLU'(conv) with the conv recomputed from the staged input.
  // 16 threads per channel walk its 18 x 18 halo pixels: the nine inputs loaded for the conv are exactly the factors of
  // d(weight) for that pixel, so the per-(channel, tap) sums ride along in registers (core pixels only) and are combined over
  // the 16 threads at the end - no separate pass over the tile.
#ifndef DW_ABL_NOSILU
  {
    constexpr int TPC = DW_THREADS / CB;   // threads per channel (16 at CB = 16)
    static_assert(DW_THREADS % CB == 0 && TPC <= WAVE && (TPC & (TPC - 1)) == 0, "a power-of-two lane group per channel");
    const int c = threadIdx.x / TPC, sub = threadIdx.x % TPC;
    float wk[9], dwk[10];
#pragma unroll
    for (int k9 = 0; k9 < 9; ++k9) { wk[k9] = s_w[c][k9]; dwk[k9] = 0.f; }
    dwk[9] = 0.f;
    const float bc = s_w[c][9];
    for (int r = sub; r < GS * GS; r += TPC) {
      const int px = r / GS, py = r - px * GS;
      const int h = h0 + py - 1, w = w0 + px - 1;
      float g = 0.f;
      if (h >= 0 && h < H && w >= 0 && w < W) {
        g = s_g[c][py][px];
        float xin[9], acc = bc;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) { xin[ky * 3 + kx] = s_x[c][py + ky][px + kx]; acc = fmaf(wk[ky * 3 + kx], xin[ky * 3 + kx], acc); }
        const float sg = 1.f / (1.f + __expf(-acc));
        g *= sg * (1.f + acc * (1.f - sg));
        if (py >= 1 && py <= TS && px >= 1 && px <= TS) {  // core pixel of this tile: counts once over all tiles
#pragma unroll
          for (int k9 = 0; k9 < 9; ++k9) dwk[k9] = fmaf(g, xin[k9], dwk[k9]);
          dwk[9] += g;
        }
      }
      s_g[c][py][px] = g;
    }
#pragma unroll
    for (int k9 = 0; k9 < 10; ++k9) dwk[k9] = group_sum<TPC>(dwk[k9]);
    if (sub == 0) {
      float* o = ws + (((size_t)b * tiles + blockIdx.x) * D + d0 + c) * 10;
#pragma unroll
      for (int k9 = 0; k9 < 10; ++k9) o[k9] = dwk[k9];
    }
  }
#endif
  __syncthreads();
  // d(input) on the core: transposed 3x3 of dL/d(conv)
  const int ty = threadIdx.x / TS, tx = threadIdx.x % TS;
  float dxv[CB];
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc = fmaf(s_w[c][ky * 3 + kx], s_g[c][ty + 2 - ky][tx + 2 - kx], acc);
    dxv[c] = acc;
  }
  __syncthreads();  // every thread is done with s_x (weight partials) before it becomes the output staging tile
#pragma unroll
  for (int c = 0; c < CB; ++c) s_x[c][ty][tx] = dxv[c];
  __syncthreads();
  constexpr int VN = Vec16<T>::N, GROUPS = CB / VN;
  for (int it = threadIdx.x; it < TS * TS * GROUPS; it += DW_THREADS) {
    const int pix = it / GROUPS, gq = it - pix * GROUPS, py = pix / TS, px = pix - py * TS;
    const int h = h0 + py, w = w0 + px;
    if (h < H && w < W) {
      float v[VN];
#pragma unroll
      for (int j = 0; j < VN; ++j) v[j] = s_x[gq * VN + j][py][px];
      Vec16<T>::st(gx + ((size_t)(b * H + h) * W + w) * gxs + d0 + gq * VN, v);
    }
  }
}

}  // namespace

static int dwconv_check(const void* a, const void* b, const void* c, int B, int D, int H, int W, long long xs, int dtype) {
  if (!a || !b || !c || B <= 0 || D <= 0 || H <= 0 || W <= 0 || xs < D) return TAMTR_EINVAL;
  if (dtype != TAMTR_F32 && dtype != TAMTR_BF16) return TAMTR_EINVAL;
  const int vn = dtype == TAMTR_F32 ? 4 : 8;
  if (D % CB_FWD || xs % vn || B > 65535 || D / CB_BWD > 65535) return TAMTR_EUNSUP;
  return TAMTR_OK;
}

extern "C" int tamtr_dwconv_silu_cross_fwd(const void* x, long long x_pixel_stride, const float* weight, const float* bias, void* u2,
                                           int B, int D, int H, int W, int dtype, int plane_dtype, void* stream) {
  const int rc = dwconv_check(x, weight, u2, B, D, H, W, x_pixel_stride, dtype);
  if (rc) return rc;
  if (plane_dtype != TAMTR_F32 && !(plane_dtype == TAMTR_BF16 && dtype == TAMTR_BF16)) return TAMTR_EINVAL;
  const int tiles_w = (W + TS - 1) / TS, tiles_h = (H + TS - 1) / TS;
  const dim3 grid(tiles_w * tiles_h, D / CB_FWD, B);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TAMTR_F32)
    hipLaunchKernelGGL((dwconv_cross_fwd_kernel<float, float>), grid, dim3(DW_THREADS), 0, s, (const float*)x, (size_t)x_pixel_stride, weight, bias,
                       (float*)u2, D, H, W, tiles_w);
  else if (plane_dtype == TAMTR_F32)
    hipLaunchKernelGGL((dwconv_cross_fwd_kernel<bf16_t, float>), grid, dim3(DW_THREADS), 0, s, (const bf16_t*)x, (size_t)x_pixel_stride, weight,
                       bias, (float*)u2, D, H, W, tiles_w);
  else   // (a 32 x 32-tile form with dword stores was built and measured for the bf16 planes: 310 against 319 us at level 0, slower at the
         // 80^2 / 40^2 levels - this kernel is bound by its LDS reads and the SiLU, not by its stores; profiles/r04_ss2d_bf16_planes.txt)
    hipLaunchKernelGGL((dwconv_cross_fwd_kernel<bf16_t, bf16_t>), grid, dim3(DW_THREADS), 0, s, (const bf16_t*)x, (size_t)x_pixel_stride, weight,
                       bias, (bf16_t*)u2, D, H, W, tiles_w);
  return tamtr_launch_status();
}

extern "C" int tamtr_dwconv_tiles(int H, int W) { return ((W + TS - 1) / TS) * ((H + TS - 1) / TS); }

extern "C" int tamtr_dwconv_silu_cross_bwd(const void* g2, const void* x, long long x_pixel_stride, const float* weight, const float* bias,
                                           void* gx, long long gx_pixel_stride, float* ws, int B, int D, int H, int W, int dtype,
                                           int plane_dtype, void* stream) {
  const int rc = dwconv_check(x, weight, g2, B, D, H, W, x_pixel_stride, dtype);
  if (rc) return rc;
  if (!gx || !ws || gx_pixel_stride < D || gx_pixel_stride % (dtype == TAMTR_F32 ? 4 : 8)) return TAMTR_EINVAL;
  if (plane_dtype != TAMTR_F32 && !(plane_dtype == TAMTR_BF16 && dtype == TAMTR_BF16)) return TAMTR_EINVAL;
  if ((uintptr_t)g2 % 16) return TAMTR_EUNSUP;
  const int tiles_w = (W + TS - 1) / TS, tiles_h = (H + TS - 1) / TS;
  const dim3 grid(tiles_w * tiles_h, D / CB_BWD, B);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TAMTR_F32)
    hipLaunchKernelGGL((dwconv_cross_bwd_kernel<float, float>), grid, dim3(DW_THREADS), 0, s, (const float*)g2, (const float*)x, (size_t)x_pixel_stride,
                       weight, bias, (float*)gx, (size_t)gx_pixel_stride, ws, D, H, W, tiles_w, tiles_w * tiles_h);
  else if (plane_dtype == TAMTR_F32)
    hipLaunchKernelGGL((dwconv_cross_bwd_kernel<bf16_t, float>), grid, dim3(DW_THREADS), 0, s, (const float*)g2, (const bf16_t*)x,
                       (size_t)x_pixel_stride, weight, bias, (bf16_t*)gx, (size_t)gx_pixel_stride, ws, D, H, W, tiles_w, tiles_w * tiles_h);
  else
    hipLaunchKernelGGL((dwconv_cross_bwd_kernel<bf16_t, bf16_t>), grid, dim3(DW_THREADS), 0, s, (const bf16_t*)g2, (const bf16_t*)x,
                       (size_t)x_pixel_stride, weight, bias, (bf16_t*)gx, (size_t)gx_pixel_stride, ws, D, H, W, tiles_w, tiles_w * tiles_h);
  return tamtr_launch_status();
}
