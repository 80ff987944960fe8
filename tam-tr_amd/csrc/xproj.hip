// xproj.hip - the x_proj of SS2D (ultralytics/nn/extra_modules/VManba/vmamba.py:962-975: x_dbl = einsum("b k d l, k c d -> b k c l", xs,
// x_proj_weight); dts, Bs, Cs = split(x_dbl, [R, N, N])) on the cross-scan pair layout, forward and backward, gfx950.
//
// Operands as the scan kernels keep them (csrc/selscan.hip): u2 f32 [B, 2, D, L] - SiLU(dwconv(x)) in its row-major and column-major
// flattening; directions k and k + 2 read copy k & 1 (2, 3 back to front, which a per-position product does not see).  So per image and
// copy i it is ONE skinny product OUT_i [2C, L] = Wcat_i [2C, D] U_i [D, L], Wcat_i = [W_i ; W_(i+2)], C = R + 2N = 40 / 48 / 64 rows,
// D = 256 / 512 / 1024, L = 25 600 / 6 400 / 1 600 at 640 px - 17 - 33 GFLOP per level against 0.4 - 0.8 GB of operands: HBM-bound.
// As torch ops it was: a cast of u2 to bf16 (1.26 GB of traffic at level 0), two weight concatenations, two batched library GEMMs with
// M = 80 rows (CK batched_gemm: 83 us each), three stack + float kernels for the split, and in the backward the concatenation + cast of
// the six gradient planes, a transposed copy of u2 for the batched dW product, two more skinny GEMMs and the fold of d/d(u2):
// 2.6 ms per step in ~60 launches (profiles/r04_step_phases.txt).  Here three kernels, all on v_mfma_f32_32x32x16_bf16 with the pixel axis
// on the lanes, every operand fragment built from global memory in registers (the reduction index of each product is either the
// lane-strided row index - 8 coalesced dword loads per fragment - or contiguous in memory - two 16-byte loads), no LDS, no atomics:
//   xproj_fwd     U read once as f32 (rounded to bf16 in registers, as torch's cast did), the three outputs written in their final
//                 [B, 4, R | N | N, L] f32 layout (values rounded to bf16 first: what the bf16 GEMM returned);
//   xproj_bwd_dx  d/d(u2)[b, i] = gu[b, i] + gu[b, i + 2] + Wcat_i^T G_i in ONE pass: the scan's four d/d(u) planes are folded while the
//                 product's accumulator tile is still in registers (replaces fold.hip's fold_add and the [B, D, L] product round trip);
//   xproj_bwd_dw  dWcat_i [2C, D] = sum over images and pixels of G_i U_i^T as per-(image, 1 024-pixel slice) partial tiles, added in slice
//                 order by tamtr_slab_sum_rows (bitwise reproducible).
// G_i = the gradient rows [gdtr_i ; gB_i ; gC_i ; gdtr_(i+2) ; gB_(i+2) ; gC_(i+2)] read where the scan backward left them.
#include "common.h"

namespace {

constexpr int XP_N = 16;             // d_state: rows of B and of C per direction
constexpr int XP_SLICE = 1024;       // pixels per partial tile of the weight gradient

__device__ __forceinline__ uint32_t xp_pk(float a, float b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ s16x8 xp_pack(const float (&p)[8]) {
  const uint32_t w[4] = {xp_pk(p[0], p[1]), xp_pk(p[2], p[3]), xp_pk(p[4], p[5]), xp_pk(p[6], p[7])};
  s16x8 f;
  __builtin_memcpy(&f, w, 16);
  return f;
}
__device__ __forceinline__ float xp_round(float x) { return bf2f(f2bf(x)); }   // what a bf16 GEMM hands back, widened again

// row m (0 <= m < 2C) of copy i's product / gradient block -> its row in dtr / Bs / Cs ([B, 4, R | 16 | 16, L]); nullptr for padding rows
template <typename T>
__device__ __forceinline__ T* xp_row(T* dtr, T* Bs, T* Cs, int b, int i, int m, int R, int C, size_t L) {
  if (m >= 2 * C) return nullptr;
  const int dir = m >= C ? i + 2 : i, c = m >= C ? m - C : m;
  if (c < R) return dtr + (((size_t)b * 4 + dir) * R + c) * L;
  if (c < R + XP_N) return Bs + (((size_t)b * 4 + dir) * XP_N + (c - R)) * L;
  return Cs + (((size_t)b * 4 + dir) * XP_N + (c - R - XP_N)) * L;
}

// ---- forward: a wave owns 32 pixels of (image b, copy i) and all MB * 32 >= 2C output rows
template <int MB>
__global__ __launch_bounds__(256) void xproj_fwd_kernel(const float* __restrict__ u2, const bf16_t* __restrict__ wcat, float* __restrict__ dtr,
                                                        float* __restrict__ Bs, float* __restrict__ Cs, int D, int L, int R) {
  const int i = blockIdx.y, b = blockIdx.z, C = R + 2 * XP_N;
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE, lr = lane & 31, lh = lane >> 5;
  const int p0 = (blockIdx.x * 4 + wave) * 32;
  if (p0 >= L) return;                                    // (wave-uniform)
  const int p = min(p0 + lr, L - 1);
  const float* U = u2 + ((size_t)b * 2 + i) * D * (size_t)L + p + (size_t)(8 * lh) * L;   // U[k][p], k = k0 + 8 lh + j
  const bf16_t* W = wcat + ((size_t)i * MB * 32 + lr) * D + 8 * lh;                        // Wcat_i[32 mb + lr][k0 + 8 lh ..]
  f32x16 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

  for (int k0 = 0; k0 < D; k0 += 16) {
    float uv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) uv[j] = U[(size_t)(k0 + j) * L];
    s16x8 af[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) af[mb] = *reinterpret_cast<const s16x8*>(W + (size_t)mb * 32 * D + k0);
    const s16x8 bf = xp_pack(uv);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mb], bf, acc[mb], 0, 0, 0);
  }
  if (p0 + lr >= L) return;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * lh;     // C layout of the 32x32 tile: row on the register, pixel on the lane
      float* dst = xp_row(dtr, Bs, Cs, b, i, m, R, C, (size_t)L);
      if (dst) dst[p0 + lr] = xp_round(acc[mb][r]);
    }
}

// ---- forward on a bf16 plane (bf16 mode: include/tamtr_hip.h "bf16 PLANES"): a lane's dword holds two neighbouring pixels, so a wave owns
// 64 pixels - the even ones in one B fragment, the odd ones in the other (two v_perm per register pair instead of eight conversions) -
// and every output row is written as 256 contiguous bytes (float2 per lane)
__device__ __forceinline__ s16x8 xp_halves(const uint32_t (&d)[8], bool odd) {
  uint32_t w[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) w[q] = odd ? __builtin_amdgcn_perm(d[2 * q + 1], d[2 * q], 0x07060302u) : __builtin_amdgcn_perm(d[2 * q + 1], d[2 * q], 0x05040100u);
  s16x8 f;
  __builtin_memcpy(&f, w, 16);
  return f;
}
template <int MB>
__global__ __launch_bounds__(256) void xproj_fwd16_kernel(const bf16_t* __restrict__ u2, const bf16_t* __restrict__ wcat, float* __restrict__ dtr,
                                                          float* __restrict__ Bs, float* __restrict__ Cs, int D, int L, int R) {
  const int i = blockIdx.y, b = blockIdx.z, C = R + 2 * XP_N;
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE, lr = lane & 31, lh = lane >> 5;
  const int p0 = (blockIdx.x * 4 + wave) * 64;
  if (p0 >= L) return;                                    // (wave-uniform)
  const int p = min(p0 + 2 * lr, L - 2);                  // L % 2 == 0: a pixel pair is inside or outside as a whole
  const uint32_t* U = reinterpret_cast<const uint32_t*>(u2 + ((size_t)b * 2 + i) * D * (size_t)L + p + (size_t)(8 * lh) * L);
  const size_t Ld = (size_t)L / 2;                        // row pitch in dwords
  const bf16_t* W = wcat + ((size_t)i * MB * 32 + lr) * D + 8 * lh;
  f32x16 acc[2][MB];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[e][mb][r] = 0.f;

  for (int k0 = 0; k0 < D; k0 += 16) {
    uint32_t uv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) uv[j] = U[(size_t)(k0 + j) * Ld];
    s16x8 af[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) af[mb] = *reinterpret_cast<const s16x8*>(W + (size_t)mb * 32 * D + k0);
    const s16x8 be = xp_halves(uv, false), bo = xp_halves(uv, true);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      acc[0][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mb], be, acc[0][mb], 0, 0, 0);
      acc[1][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mb], bo, acc[1][mb], 0, 0, 0);
    }
  }
  if (p0 + 2 * lr >= L) return;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * lh;
      float* dst = xp_row(dtr, Bs, Cs, b, i, m, R, C, (size_t)L);
      if (dst) *reinterpret_cast<float2*>(dst + p0 + 2 * lr) = make_float2(xp_round(acc[0][mb][r]), xp_round(acc[1][mb][r]));
    }
}

// ---- backward, d/d(u2): a wave owns 32 pixels of (b, i); its B operand = the 2C gradient rows of those pixels, kept for all D / 32 row blocks
template <int KS>   // k-steps of 16 gradient rows: KS * 16 >= 2C
__global__ __launch_bounds__(256) void xproj_bwd_dx_kernel(const float* __restrict__ gu, const float* __restrict__ gdtr, const float* __restrict__ gB,
                                                           const float* __restrict__ gC, const bf16_t* __restrict__ wT, float* __restrict__ gu2,
                                                           int D, int L, int R) {
  const int i = blockIdx.y, b = blockIdx.z, C = R + 2 * XP_N;
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE, lr = lane & 31, lh = lane >> 5;
  const int p0 = (blockIdx.x * 4 + wave) * 32;
  if (p0 >= L) return;
  const int p = min(p0 + lr, L - 1);
  s16x8 gf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    float gv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float* row = xp_row(gdtr, gB, gC, b, i, 16 * ks + 8 * lh + j, R, C, (size_t)L);
      gv[j] = row ? row[p] : 0.f;
    }
    gf[ks] = xp_pack(gv);
  }
  const bool live = p0 + lr < L;
  const size_t plane = (size_t)D * L;
  const float* g0 = gu + ((size_t)b * 4 + i) * plane + p0 + lr;       // direction i; direction i + 2 sits two planes further
  float* out = gu2 + ((size_t)b * 2 + i) * plane + p0 + lr;
  const bf16_t* W = wT + ((size_t)i * D + lr) * (KS * 16) + 8 * lh;   // Wcat_i^T[32 db + lr][16 ks + 8 lh ..]
  for (int db = 0; db < D / 32; ++db) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const s16x8 af = *reinterpret_cast<const s16x8*>(W + (size_t)db * 32 * (KS * 16) + 16 * ks);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, gf[ks], acc, 0, 0, 0);
    }
    if (live) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const size_t o = (size_t)(32 * db + (r & 3) + 8 * (r >> 2) + 4 * lh) * L;
        out[o] = g0[o] + g0[o + 2 * plane] + xp_round(acc[r]);
      }
    }
  }
}

// ---- the same on bf16 planes: gu [B, 4, D, L] and the result [B, 2, D, L] are bf16; a wave owns 64 pixels (a dword = a pixel pair per lane)
template <int KS>
__global__ __launch_bounds__(256) void xproj_bwd_dx16_kernel(const bf16_t* __restrict__ gu, const float* __restrict__ gdtr, const float* __restrict__ gB,
                                                             const float* __restrict__ gC, const bf16_t* __restrict__ wT, bf16_t* __restrict__ gu2,
                                                             int D, int L, int R) {
  const int i = blockIdx.y, b = blockIdx.z, C = R + 2 * XP_N;
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE, lr = lane & 31, lh = lane >> 5;
  const int p0 = (blockIdx.x * 4 + wave) * 64;
  if (p0 >= L) return;
  const int p = min(p0 + 2 * lr, L - 2);
  s16x8 ge[KS], go[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    float ve[8], vo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float* row = xp_row(gdtr, gB, gC, b, i, 16 * ks + 8 * lh + j, R, C, (size_t)L);
      const float2 v = row ? *reinterpret_cast<const float2*>(row + p) : make_float2(0.f, 0.f);
      ve[j] = v.x; vo[j] = v.y;
    }
    ge[ks] = xp_pack(ve); go[ks] = xp_pack(vo);
  }
  const bool live = p0 + 2 * lr < L;
  const size_t plane = (size_t)D * L;
  const bf16_t* g0 = gu + ((size_t)b * 4 + i) * plane + p;            // direction i; direction i + 2 sits two planes further
  bf16_t* out = gu2 + ((size_t)b * 2 + i) * plane + p;
  const bf16_t* W = wT + ((size_t)i * D + lr) * (KS * 16) + 8 * lh;
  for (int db = 0; db < D / 32; ++db) {
    f32x16 ae, ao;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ae[r] = 0.f; ao[r] = 0.f; }
    uint32_t a0[16], a2[16];   // the folded planes' pixel pairs: requested before the products
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const size_t o = (size_t)(32 * db + (r & 3) + 8 * (r >> 2) + 4 * lh) * L;
      a0[r] = *reinterpret_cast<const uint32_t*>(g0 + o);
      a2[r] = *reinterpret_cast<const uint32_t*>(g0 + o + 2 * plane);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const s16x8 af = *reinterpret_cast<const s16x8*>(W + (size_t)db * 32 * (KS * 16) + 16 * ks);
      ae = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, ge[ks], ae, 0, 0, 0);
      ao = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, go[ks], ao, 0, 0, 0);
    }
    if (live) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const size_t o = (size_t)(32 * db + (r & 3) + 8 * (r >> 2) + 4 * lh) * L;
        const float se = __uint_as_float(a0[r] << 16) + __uint_as_float(a2[r] << 16) + xp_round(ae[r]);
        const float so = __uint_as_float(a0[r] & 0xffff0000u) + __uint_as_float(a2[r] & 0xffff0000u) + xp_round(ao[r]);
        *reinterpret_cast<uint32_t*>(out + o) = xp_pk(se, so);
      }
    }
  }
}

// ---- backward, weight gradient: a workgroup = (1 024-pixel slice, 256-channel group, b, i); wave w: channels 64 w .. 64 w + 63 of the group x all rows
template <int MB, bool P16 = false>   // P16: u2 is a bf16 plane - eight pixels of a row are one 16-byte load and already the MFMA fragment
__global__ __launch_bounds__(256) void xproj_bwd_dw_kernel(const float* __restrict__ u2, const float* __restrict__ gdtr, const float* __restrict__ gB,
                                                           const float* __restrict__ gC, float* __restrict__ part, int D, int L, int R, int nslice) {
  const int C = R + 2 * XP_N, ngrp = D / 256;
  const int sl = blockIdx.x / ngrp, grp = blockIdx.x % ngrp, i = blockIdx.y, b = blockIdx.z;
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE, lr = lane & 31, lh = lane >> 5;
  const int pa = sl * XP_SLICE, pb = min(pa + XP_SLICE, L);
  const float* grow[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) grow[mb] = xp_row(gdtr, gB, gC, b, i, 32 * mb + lr, R, C, (size_t)L);
  const int d0 = grp * 256 + wave * 64;
  const float* urow[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const size_t e = (((size_t)b * 2 + i) * D + d0 + 32 * q + lr) * (size_t)L;
    urow[q] = P16 ? reinterpret_cast<const float*>(reinterpret_cast<const bf16_t*>(u2) + e) : u2 + e;
  }
  f32x16 acc[2][MB];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][mb][r] = 0.f;

  // The k index of an MFMA is only a summation index: both operands may assign ANY pixels to its 16 slots as long as they agree.  Two MFMA
  // steps per iteration over 32 pixels; lane half lh takes the 16 CONSECUTIVE pixels [k0 + 16 lh, + 16) - the first step sums its first
  // eight, the second its last eight - so a lane reads 32 (bf16 row) / 64 (f32 row) contiguous bytes per row instead of two 16 / 32-byte
  // pieces 32 pixels apart, and twice as many loads are in flight (the first form ran at 2.1 TB/s of algorithmic bytes).
  for (int k0 = pa; k0 < pb; k0 += 32) {
    const int pk = k0 + 16 * lh;                      // this lane's 16 pixels (L % 8 == 0: each half of them all inside or all outside)
    const bool in0 = pk < pb, in1 = pk + 8 < pb;
    const int pc = in0 ? pk : pa, pc1 = in1 ? pk + 8 : pa;   // (clamped: the loads are issued either way, their values zeroed)
    s16x8 af[2][MB], bf[2][2];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      // (a padding row - m >= 2C, only in the last row block - reads a real row and is zeroed: no load sits under a branch, hipcc would end
      // the block with s_waitcnt vmcnt(0))
      const float* gr = grow[mb] ? grow[mb] : gdtr;
      const float4 x0 = *reinterpret_cast<const float4*>(gr + pc), x1 = *reinterpret_cast<const float4*>(gr + pc + 4);
      const float4 y0 = *reinterpret_cast<const float4*>(gr + pc1), y1 = *reinterpret_cast<const float4*>(gr + pc1 + 4);
      const float z0 = (grow[mb] && in0) ? 1.f : 0.f, z1 = (grow[mb] && in1) ? 1.f : 0.f;
      const float gv[2][8] = {{x0.x * z0, x0.y * z0, x0.z * z0, x0.w * z0, x1.x * z0, x1.y * z0, x1.z * z0, x1.w * z0},
                              {y0.x * z1, y0.y * z1, y0.z * z1, y0.w * z1, y1.x * z1, y1.y * z1, y1.z * z1, y1.w * z1}};
      af[0][mb] = xp_pack(gv[0]);
      af[1][mb] = xp_pack(gv[1]);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if constexpr (P16) {
        uint4 x = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(urow[q]) + pc);
        uint4 y = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(urow[q]) + pc1);
        if (!in0) x = make_uint4(0u, 0u, 0u, 0u);
        if (!in1) y = make_uint4(0u, 0u, 0u, 0u);
        __builtin_memcpy(&bf[0][q], &x, 16);
        __builtin_memcpy(&bf[1][q], &y, 16);
      } else {
        const float4 x0 = *reinterpret_cast<const float4*>(urow[q] + pc), x1 = *reinterpret_cast<const float4*>(urow[q] + pc + 4);
        const float4 y0 = *reinterpret_cast<const float4*>(urow[q] + pc1), y1 = *reinterpret_cast<const float4*>(urow[q] + pc1 + 4);
        float uv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w}, uw[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) { if (!in0) uv[j] = 0.f; if (!in1) uw[j] = 0.f; }
        bf[0][q] = xp_pack(uv);
        bf[1][q] = xp_pack(uw);
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[q][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[h][mb], bf[h][q], acc[q][mb], 0, 0, 0);
  }
  // partial tile: part[(b * nslice + sl)][i][m][d], m < 2C
  float* dst = part + (((size_t)b * nslice + sl) * 2 + i) * (size_t)(2 * C) * D;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < 2 * C) dst[(size_t)m * D + d0 + 32 * q + lr] = acc[q][mb][r];
      }
}

inline bool xp_args_ok(int B, int D, int L, int R) { return B > 0 && B <= 65535 && D > 0 && L > 0 && R >= 1 && R <= 32; }

}  // namespace

extern "C" int tamtr_xproj_dw_slices(int L) { return L > 0 ? (L + XP_SLICE - 1) / XP_SLICE : 0; }

/* see include/tamtr_hip.h */
extern "C" int tamtr_xproj_fwd(const void* u2, const void* wcat, float* dtr, float* Bs, float* Cs, int B, int D, int L, int R, int plane_dtype,
                               void* stream) {
  if (!u2 || !wcat || !dtr || !Bs || !Cs || !xp_args_ok(B, D, L, R) || (plane_dtype != TAMTR_F32 && plane_dtype != TAMTR_BF16)) return TAMTR_EINVAL;
  const int MB = (2 * (R + 2 * XP_N) + 31) / 32;
  if (D % 16 || (uintptr_t)wcat % 16 || MB < 3 || MB > 4) return TAMTR_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
  if (plane_dtype == TAMTR_BF16) {
    if (L % 2 || ((uintptr_t)u2 | (uintptr_t)dtr | (uintptr_t)Bs | (uintptr_t)Cs) % 8) return TAMTR_EUNSUP;
    const dim3 grid((L + 255) / 256, 2, B);
    if (MB == 3) hipLaunchKernelGGL(xproj_fwd16_kernel<3>, grid, dim3(256), 0, s, (const bf16_t*)u2, (const bf16_t*)wcat, dtr, Bs, Cs, D, L, R);
    else hipLaunchKernelGGL(xproj_fwd16_kernel<4>, grid, dim3(256), 0, s, (const bf16_t*)u2, (const bf16_t*)wcat, dtr, Bs, Cs, D, L, R);
    return tamtr_launch_status();
  }
  const dim3 grid((L + 127) / 128, 2, B);
  if (MB == 3) hipLaunchKernelGGL(xproj_fwd_kernel<3>, grid, dim3(256), 0, s, (const float*)u2, (const bf16_t*)wcat, dtr, Bs, Cs, D, L, R);
  else hipLaunchKernelGGL(xproj_fwd_kernel<4>, grid, dim3(256), 0, s, (const float*)u2, (const bf16_t*)wcat, dtr, Bs, Cs, D, L, R);
  return tamtr_launch_status();
}

extern "C" int tamtr_xproj_bwd_dx(const void* gu, const float* gdtr, const float* gB, const float* gC, const void* wT, void* gu2, int B, int D,
                                  int L, int R, int plane_dtype, void* stream) {
  if (!gu || !gdtr || !gB || !gC || !wT || !gu2 || !xp_args_ok(B, D, L, R) || (plane_dtype != TAMTR_F32 && plane_dtype != TAMTR_BF16)) return TAMTR_EINVAL;
  const int KS = (2 * (R + 2 * XP_N) + 15) / 16;
  if (D % 32 || (uintptr_t)wT % 16 || KS < 5 || KS > 8) return TAMTR_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
  if (plane_dtype == TAMTR_BF16) {
    if (L % 2 || ((uintptr_t)gu | (uintptr_t)gu2) % 4 || ((uintptr_t)gdtr | (uintptr_t)gB | (uintptr_t)gC) % 8) return TAMTR_EUNSUP;
    const dim3 grid((L + 255) / 256, 2, B);
#define GO(K) hipLaunchKernelGGL(xproj_bwd_dx16_kernel<K>, grid, dim3(256), 0, s, (const bf16_t*)gu, gdtr, gB, gC, (const bf16_t*)wT, (bf16_t*)gu2, D, L, R)
    switch (KS) { case 5: GO(5); break; case 6: GO(6); break; case 7: GO(7); break; default: GO(8); break; }
#undef GO
    return tamtr_launch_status();
  }
  const dim3 grid((L + 127) / 128, 2, B);
#define GO(K) hipLaunchKernelGGL(xproj_bwd_dx_kernel<K>, grid, dim3(256), 0, s, (const float*)gu, gdtr, gB, gC, (const bf16_t*)wT, (float*)gu2, D, L, R)
  switch (KS) { case 5: GO(5); break; case 6: GO(6); break; case 7: GO(7); break; default: GO(8); break; }
#undef GO
  return tamtr_launch_status();
}

extern "C" int tamtr_xproj_bwd_dw(const void* u2, const float* gdtr, const float* gB, const float* gC, float* part, int B, int D, int L, int R,
                                  int plane_dtype, void* stream) {
  if (!u2 || !gdtr || !gB || !gC || !part || !xp_args_ok(B, D, L, R) || (plane_dtype != TAMTR_F32 && plane_dtype != TAMTR_BF16)) return TAMTR_EINVAL;
  const int MB = (2 * (R + 2 * XP_N) + 31) / 32;
  if (D % 256 || L % 8 || MB < 3 || MB > 4 || ((uintptr_t)u2 | (uintptr_t)gdtr | (uintptr_t)gB | (uintptr_t)gC) % 16) return TAMTR_EUNSUP;
  const int nslice = tamtr_xproj_dw_slices(L);
  const dim3 grid(nslice * (D / 256), 2, B);
  hipStream_t s = (hipStream_t)stream;
  const float* u = (const float*)u2;
  if (plane_dtype == TAMTR_BF16) {
    if (MB == 3) hipLaunchKernelGGL((xproj_bwd_dw_kernel<3, true>), grid, dim3(256), 0, s, u, gdtr, gB, gC, part, D, L, R, nslice);
    else hipLaunchKernelGGL((xproj_bwd_dw_kernel<4, true>), grid, dim3(256), 0, s, u, gdtr, gB, gC, part, D, L, R, nslice);
  } else {
    if (MB == 3) hipLaunchKernelGGL((xproj_bwd_dw_kernel<3, false>), grid, dim3(256), 0, s, u, gdtr, gB, gC, part, D, L, R, nslice);
    else hipLaunchKernelGGL((xproj_bwd_dw_kernel<4, false>), grid, dim3(256), 0, s, u, gdtr, gB, gC, part, D, L, R, nslice);
  }
  return tamtr_launch_status();
}
