// gemm_bf16.hip - Y[M,N] = X[M,K] @ W[N,K]^T + bias, bf16 in / fp32 MFMA accumulate / bf16 out, gfx950.
//
// The MEH value projection (reference: `value = self.value_proj(value)`, ultralytics/nn/modules/transformer.py:273):
// M = B*L = 537 600 rows at 640^2 bs 16, N = K = 512, three times per step - the dominant dense contraction of the
// hot path and the kernel bench.py prices against the bf16 MFMA roof.  Shape facts that drive the design:
//   * both operands are K-contiguous ("NT" GEMM): a 16-byte row chunk IS an MFMA fragment for A and for B;
//   * K = 512 is short (8 steps of 64) and W (512 KB) is L2-resident, so per output byte the kernel moves
//     2*K/N bytes of X: arithmetic intensity 256 flop/B sits right at the chip's 2.5 PF / 8 TB/s balance point -
//     X must be read from HBM ONCE.  All N/128 column tiles of one 128-row panel are therefore placed on the SAME
//     XCD back to back (blockIdx -> (xcd, slot) remap) so that three of the four panel reads hit that XCD's L2;
//   * operands are swapped (A-operand = W rows, B-operand = X rows) so that each lane's accumulator holds 4 consecutive
//     n for one m: the epilogue packs 8-byte pieces, transposes through LDS and stores whole 128-byte row segments.
// Tile 128(M) x 128(N) x 64(K), 4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16 tiles; global->register->
// LDS staging with the next tile's loads in flight under the MFMAs, XOR-swizzled 128-B rows (chunk ^ (row & 7)) so
// ds_read_b128 fragment reads spread over all 16-B slots; two LDS stages, one barrier per K-step.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int GT = 256;                      // threads
constexpr int ROW_BYTES = BK * 2;            // 128 B per staged row
constexpr int TILE_BYTES = BM * ROW_BYTES;   // 16 KB per operand tile
constexpr int EP_PITCH = 64 + 8;             // epilogue LDS row pitch (bf16 elements)

__device__ __forceinline__ uint32_t swz(int row, int chunk) { return (uint32_t)row * ROW_BYTES + (uint32_t)((chunk ^ (row & 7)) << 4); }

__global__ __launch_bounds__(GT, 2) void linear_bf16_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                             const float* __restrict__ bias, bf16_t* __restrict__ Y, int M, int N,
                                                             int K, int m_blocks, int n_blocks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2 stages][X tile | W tile]
  // ---- XCD-aware tile assignment: the n_blocks column tiles of a row panel run consecutively on one XCD
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int mb = (slot / n_blocks) * 8 + xcd, nb = slot % n_blocks;
  if (mb >= m_blocks) return;
  const int m0 = mb * BM, n0 = nb * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;  // wave's 64x64 sub-tile
  const int lr = lane & 31, lh = lane >> 5;

  // staging map: chunk id = tid + GT*i -> row = id/8, chunk = id%8 (8 threads cover one 128-B row: coalesced)
  uint4 rx0, rx1, rx2, rx3, rw0, rw1, rw2, rw3;
  const int srow = tid >> 3, sc = tid & 7;  // + 32 rows per i
  const bf16_t* xg = X + (size_t)(m0 + srow) * K + sc * 8;
  const bf16_t* wg = W + (size_t)(n0 + srow) * K + sc * 8;
  const size_t rstep = (size_t)32 * K;
  const bool ok0 = m0 + srow < M, ok1 = m0 + srow + 32 < M, ok2 = m0 + srow + 64 < M, ok3 = m0 + srow + 96 < M;
  const uint4 z4 = make_uint4(0, 0, 0, 0);
#define G_LOAD(k0)                                                                 \
  {                                                                                \
    rx0 = ok0 ? *reinterpret_cast<const uint4*>(xg + (k0)) : z4;                   \
    rx1 = ok1 ? *reinterpret_cast<const uint4*>(xg + rstep + (k0)) : z4;           \
    rx2 = ok2 ? *reinterpret_cast<const uint4*>(xg + 2 * rstep + (k0)) : z4;       \
    rx3 = ok3 ? *reinterpret_cast<const uint4*>(xg + 3 * rstep + (k0)) : z4;       \
    rw0 = *reinterpret_cast<const uint4*>(wg + (k0));                              \
    rw1 = *reinterpret_cast<const uint4*>(wg + rstep + (k0));                      \
    rw2 = *reinterpret_cast<const uint4*>(wg + 2 * rstep + (k0));                  \
    rw3 = *reinterpret_cast<const uint4*>(wg + 3 * rstep + (k0));                  \
  }
  // (srow + 32*i) & 7 == srow & 7: the swizzled chunk offset is the same for the four rows of a thread
  const uint32_t soff = swz(srow, sc);
#define S_STORE(stage)                                                             \
  {                                                                                \
    unsigned char* sx_ = smem + (stage) * 2 * TILE_BYTES + soff;                   \
    unsigned char* sw_ = sx_ + TILE_BYTES;                                         \
    *reinterpret_cast<uint4*>(sx_) = rx0;                                          \
    *reinterpret_cast<uint4*>(sx_ + 32 * ROW_BYTES) = rx1;                         \
    *reinterpret_cast<uint4*>(sx_ + 64 * ROW_BYTES) = rx2;                         \
    *reinterpret_cast<uint4*>(sx_ + 96 * ROW_BYTES) = rx3;                         \
    *reinterpret_cast<uint4*>(sw_) = rw0;                                          \
    *reinterpret_cast<uint4*>(sw_ + 32 * ROW_BYTES) = rw1;                         \
    *reinterpret_cast<uint4*>(sw_ + 64 * ROW_BYTES) = rw2;                         \
    *reinterpret_cast<uint4*>(sw_ + 96 * ROW_BYTES) = rw3;                         \
  }

  f32x16 acc[2][2];  // [ni][mi]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nk = K / BK;
  G_LOAD(0)
  S_STORE(0)
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) G_LOAD((kt + 1) * BK)  // in flight under the MFMAs below
    const unsigned char* sx = smem + cur * 2 * TILE_BYTES;
    const unsigned char* sw = sx + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int c = ks * 2 + lh;
      s16x8 fw[2], fx[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fw[i] = *reinterpret_cast<const s16x8*>(sw + swz(wn * 64 + i * 32 + lr, c));
        fx[i] = *reinterpret_cast<const s16x8*>(sx + swz(wm * 64 + i * 32 + lr, c));
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ni], fx[mi], acc[ni][mi], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      S_STORE(cur ^ 1)  // the other stage was last read before the previous barrier
      __syncthreads();
    }
  }

#undef G_LOAD
#undef S_STORE
  // ---- epilogue: D[i][j] = Y[m0 + j][n0 + i]; lane: j = lr, i = (r&3) + 8*(r>>2) + 4*lh
  __syncthreads();  // all waves done reading the staging tiles
  bf16_t* ep = reinterpret_cast<bf16_t*>(smem) + wave * 64 * EP_PITCH;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nn = ni * 32 + 8 * g + 4 * lh;  // first of 4 consecutive n (within the wave's 64)
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bias) bv = *reinterpret_cast<const float4*>(bias + n0 + wn * 64 + nn);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const f32x16& a = acc[ni][mi];
        uint2 pk;
        pk.x = (uint32_t)f2bf(a[4 * g] + bv.x) | ((uint32_t)f2bf(a[4 * g + 1] + bv.y) << 16);
        pk.y = (uint32_t)f2bf(a[4 * g + 2] + bv.z) | ((uint32_t)f2bf(a[4 * g + 3] + bv.w) << 16);
        *reinterpret_cast<uint2*>(ep + (mi * 32 + lr) * EP_PITCH + nn) = pk;
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed (same-wave readback)
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = i * 8 + (lane >> 3), cc = (lane & 7) * 8;
    const int gm = m0 + wm * 64 + row;
    const uint4 vv = *reinterpret_cast<const uint4*>(ep + row * EP_PITCH + cc);
    if (gm < M) *reinterpret_cast<uint4*>(Y + (size_t)gm * N + n0 + wn * 64 + cc) = vv;
  }
}

// =====================================================================================================================
// N-tile = 512 variant (the value projection itself: N = K = 512).  A workgroup owns 128 rows x ALL 512 columns, so X is
// read from HBM exactly once by construction and the operand that is re-read per workgroup is the small one (W, 512 KB,
// always L2-resident).  8 waves (2 along M x 4 along N), each 64 x 128 outputs = 4x2 MFMA tiles (128 accumulator
// registers); K runs in 32-deep steps through a 3-stage LDS ring filled by global_load_lds (16 B per lane, 1 KiB per
// wave-instruction, no staging registers); the XOR swizzle (chunk ^ ((row >> 2) & 3) on 64-B rows, conflict-free for
// ds_read_b128's 16-lane groups) is applied on the SOURCE address because LDS-DMA writes lane-linear; loads of step k+2
// are in flight across the barrier of step k+1 (counted vmcnt, raw s_barrier).
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gl_void;

constexpr int TM = 128, TN = 512, TK = 32, NST = 4, T5 = 512;
constexpr int ROWB = TK * 2;                       // 64 B per staged row
constexpr int W_BYTES = TN * ROWB;                 // 32 KB
constexpr int ST_BYTES = (TN + TM) * ROWB;         // 40 KB per stage
constexpr int GROUPS = (TN + TM) / 16;             // 1-KiB row groups per stage = 40
constexpr int GPW = GROUPS / (T5 / 64);            // groups per wave = 5

__device__ __forceinline__ uint32_t swz64(int row, int chunk) { return (uint32_t)row * ROWB + (uint32_t)((chunk ^ ((row >> 2) & 3)) << 4); }

constexpr int EPQ = 32 + 8;                        // epilogue quarter pitch (bf16 elements): 32 columns + pad
constexpr int EP_BYTES = 8 * 64 * EPQ * 2;         // 40 KB, outside the ring so that the next tile's loads keep flowing

// PERSISTENT: one workgroup per CU walks tiles t = blockIdx.x, += gridDim.x.  The k-step stream is continuous across
// tiles (step s+2 is always in flight, also over a tile boundary), so only the very first tile pays a cold prologue;
// the epilogue goes through its own LDS region while the next tile's first stages are already landing in the ring.
__global__ __launch_bounds__(T5, 1) void linear_bf16_n512_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                                  const float* __restrict__ bias, bf16_t* __restrict__ Y, int M,
                                                                  int N, int K, int n_blocks, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [NST stages][W tile | X tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 31, lh = lane >> 5;
  const int nk = K / TK;
  const int my_tiles = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_tiles * nk;

  // ---- LDS-DMA assignment: wave w fills row groups w*5 .. w*5+4 of every stage; lane -> (row lane>>2, physical chunk lane&3)
  const int srow = lane >> 2;
  const int slog = (lane & 3) ^ ((lane >> 4) & 3);  // logical 16-B chunk this lane must fetch (row & 15 == lane >> 2)
  auto issue = [&](int s) {
    const int tile = (int)blockIdx.x + (s / nk) * (int)gridDim.x;
    const int m0 = (tile / n_blocks) * TM, n0 = (tile % n_blocks) * TN;
    const int k0 = (s % nk) * TK;
    unsigned char* stage = smem + (s % NST) * ST_BYTES;
#pragma unroll
    for (int j = 0; j < GPW; ++j) {
      const int g = wave * GPW + j;
      const bf16_t* src;
      if (g < TN / 16) src = W + (size_t)(n0 + g * 16 + srow) * K + slog * 8 + k0;
      else src = X + (size_t)min(m0 + (g - TN / 16) * 16 + srow, M - 1) * K + slog * 8 + k0;  // rows >= M: re-read row M-1
#ifdef GEMM_ABL_NO_W
      if (g < TN / 16) continue;
#endif
#ifdef GEMM_ABL_NO_X
      if (g >= TN / 16) src = X + (size_t)srow * K + slog * 8;
#endif
      __builtin_amdgcn_global_load_lds((gl_void*)src, (lds_void*)(stage + g * 1024), 16, 0, 0);
    }
  };

  if (total > 0) issue(0);
  if (total > 1) issue(1);
  if (total > 2) issue(2);
  int s = 0;
  for (int ti = 0; ti < my_tiles; ++ti) {
    const int tile = (int)blockIdx.x + ti * (int)gridDim.x;
    const int m0 = (tile / n_blocks) * TM, n0 = (tile % n_blocks) * TN;
    // accumulators start at the bias (rows of D = n): no global load is left in the epilogue, where its s_waitcnt vmcnt(0)
    // would also wait for every store issued before it (measured: 16 serialised L2 round trips per tile, ~55 % of the kernel)
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bv = *reinterpret_cast<const float4*>(bias + n0 + wn * 128 + i * 32 + 8 * g + 4 * lh);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j][4 * g] = bv.x; acc[i][j][4 * g + 1] = bv.y; acc[i][j][4 * g + 2] = bv.z; acc[i][j][4 * g + 3] = bv.w;
        }
      }

    for (int kt = 0; kt < nk; ++kt, ++s) {
      // step s has landed for this wave once at most the two newer groups (steps s+1, s+2) are outstanding: the ring is 4
      // deep so that 3 steps = 120 KB are in flight per CU (ingest rate = bytes in flight / latency: with 80 KB in flight
      // the LDS-DMA stream alone took 248 us for this GEMM)
      if (s + 2 < total) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else if (s + 1 < total) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave's part of step s is in LDS; everyone is done reading step s-1
      const unsigned char* sw = smem + (s % NST) * ST_BYTES;
      const unsigned char* sx = sw + W_BYTES;
      // all 12 fragments of the step first (48 VGPRs), then the LDS-DMA issue for step s+2 (its ~60-100 cycles per piece
      // run while the reads are in flight), then 16 MFMAs back to back
      s16x8 fw[2][4], fx[2][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int c = ks * 2 + lh;
#pragma unroll
        for (int i = 0; i < 4; ++i) fw[ks][i] = *reinterpret_cast<const s16x8*>(sw + swz64(wn * 128 + i * 32 + lr, c));
#pragma unroll
        for (int j = 0; j < 2; ++j) fx[ks][j] = *reinterpret_cast<const s16x8*>(sx + swz64(wm * 64 + j * 32 + lr, c));
      }
      if (s + 3 < total) issue(s + 3);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ks][i], fx[ks][j], acc[i][j], 0, 0, 0);
    }

    // ---- epilogue (wave-private LDS image): D[i][j]: n = wn*128 + i*32 + (r&3) + 8*(r>>2) + 4*lh, m = wm*64 + j*32 + lr
    // image aliased onto the stage the tile's last step consumed (free until issue(s+3) at the next step's barrier)
    __builtin_amdgcn_s_barrier();  // all waves done reading that stage
    bf16_t* ep = reinterpret_cast<bf16_t*>(smem + ((s - 1) % NST) * ST_BYTES) + wave * 64 * EPQ;
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // one 32-column quarter of the wave's 128 columns at a time
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nn = 8 * g + 4 * lh;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          uint2 pk;
          pk.x = (uint32_t)f2bf(acc[i][j][4 * g]) | ((uint32_t)f2bf(acc[i][j][4 * g + 1]) << 16);
          pk.y = (uint32_t)f2bf(acc[i][j][4 * g + 2]) | ((uint32_t)f2bf(acc[i][j][4 * g + 3]) << 16);
          *reinterpret_cast<uint2*>(ep + (j * 32 + lr) * EPQ + nn) = pk;
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 4; ++it) {  // 64 rows x 64 B: 4 lanes per row, 16 rows per instruction
        const int row = it * 16 + (lane >> 2), cc = (lane & 3) * 8;
        const int gm = m0 + wm * 64 + row;
        const uint4 vv = *reinterpret_cast<const uint4*>(ep + row * EPQ + cc);
#ifdef GEMM_ABL_NO_STORE
        if (gm < M && vv.x == 0x12345678u)
#else
        if (gm < M)
#endif
          *reinterpret_cast<uint4*>(Y + (size_t)gm * N + n0 + wn * 128 + i * 32 + cc) = vv;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// TK = 64 variant: 128-B row pieces (whole cache lines per LDS-DMA request), 2-stage ring of 80 KB, epilogue image aliased
// onto the stage that was just consumed.
constexpr int K6 = 64, ROW6 = 128, W6_BYTES = TN * ROW6, ST6 = (TN + TM) * ROW6, G6 = ST6 / 1024, GPW6 = G6 / 8;
__device__ __forceinline__ uint32_t swz128(int row, int chunk) { return (uint32_t)row * ROW6 + (uint32_t)((chunk ^ ((row >> 1) & 7)) << 4); }

__global__ __launch_bounds__(T5, 1) void linear_bf16_n512_k64_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                                      const float* __restrict__ bias, bf16_t* __restrict__ Y, int M,
                                                                      int N, int K, int n_blocks, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2 stages][W tile | X tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 31, lh = lane >> 5;
  const int nk = K / K6;
  const int my_tiles = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_tiles * nk;
  const int srow = lane >> 3;  // 8 rows of 128 B per 1-KiB piece
  // ---- LDS-DMA addressing, all hoisted: per lane a 32-bit byte offset per piece (relative to the tile's W / X row 0),
  // per wave a running step cursor (tile index, k offset) advanced by addition only.  (A first version recomputed
  // tile = s / nk, m0, min(row, M-1) and 64-bit products inside every step: in-kernel stamps put 35 % of the wave time there.)
  uint32_t poff[GPW6];
  bool px[GPW6];
#pragma unroll
  for (int j = 0; j < GPW6; ++j) {
    const int g = wave * GPW6 + j;
    const int slog = (lane & 7) ^ ((4 * g + (lane >> 4)) & 7);  // logical chunk: row = 8g + lane>>3, swizzle (row>>1)&7
    px[j] = g >= W6_BYTES / 1024;
    const int row = (px[j] ? g - W6_BYTES / 1024 : g) * 8 + srow;
    poff[j] = ((uint32_t)row * (uint32_t)K + (uint32_t)slog * 8u) * 2u;
  }
  int is_tile = (int)blockIdx.x, is_k = 0, is_s = 0;  // issue cursor
  const uint32_t xlast = ((uint32_t)(M - 1) * (uint32_t)K) * 2u;  // byte offset of the last valid X row
  // a step's GPW6 pieces are emitted a few at a time between the MFMA groups (issue_begin / issue_piece x GPW6 /
  // issue_end): all 80 pieces of a CU fired back to back right after the barrier queue up in the CU's one vector-memory
  // pipe (measured 266 cycles of issue stall per piece, 35-44 % of the wave time) and nothing overlaps them
  const unsigned char *is_wb = nullptr, *is_xb = nullptr;
  unsigned char* is_stage = nullptr;
  bool is_tail = false;
  uint32_t is_lim = 0;
  auto issue_begin = [&]() {
    const int m0 = (is_tile / n_blocks) * TM, n0 = (is_tile % n_blocks) * TN;  // n_blocks == 1 for N = 512: folds away
    is_wb = reinterpret_cast<const unsigned char*>(W) + ((size_t)n0 * K + is_k) * 2;
    is_xb = reinterpret_cast<const unsigned char*>(X) + ((size_t)m0 * K + is_k) * 2;
    is_tail = m0 + TM > M;
    is_lim = xlast - (uint32_t)m0 * (uint32_t)K * 2u;
    is_stage = smem + (is_s & 1) * ST6 + wave * GPW6 * 1024;
  };
  auto issue_piece = [&](int j) {
    const unsigned char* src;
    if (!px[j]) src = is_wb + poff[j];
    else if (!is_tail) src = is_xb + poff[j];
    else {  // rows >= M re-read row M-1 (never stored)
      const uint32_t rowoff = poff[j] - (poff[j] % ((uint32_t)K * 2u));
      src = is_xb + (rowoff > is_lim ? is_lim + (poff[j] - rowoff) : poff[j]);
    }
#ifdef GEMM_ABL_NO_LOAD
    if (is_s < 2)  // timing-only build: only the first two steps are actually loaded
#endif
    __builtin_amdgcn_global_load_lds((gl_void*)src, (lds_void*)(is_stage + j * 1024), 16, 0, 0);
  };
  auto issue_end = [&]() {
    ++is_s;
    is_k += K6;
    if (is_k == K) { is_k = 0; is_tile += (int)gridDim.x; }
  };
  auto issue = [&]() {
    issue_begin();
#pragma unroll
    for (int j = 0; j < GPW6; ++j) issue_piece(j);
    issue_end();
  };
  if (total > 0) issue();
  int s = 0;
#ifdef GEMM_STAMP
  unsigned long long sw_ = 0, si_ = 0, sc_ = 0, se_ = 0, tstart;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tstart)::"memory");
#endif
  for (int ti = 0; ti < my_tiles; ++ti) {
    const int tile = (int)blockIdx.x + ti * (int)gridDim.x;
    const int m0 = (tile / n_blocks) * TM, n0 = (tile % n_blocks) * TN;
    // accumulators start at the bias (rows of D = n): no global load is left in the epilogue, where its s_waitcnt vmcnt(0)
    // would also wait for every store issued before it (measured: 16 serialised L2 round trips per tile, ~55 % of the kernel)
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bv = *reinterpret_cast<const float4*>(bias + n0 + wn * 128 + i * 32 + 8 * g + 4 * lh);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j][4 * g] = bv.x; acc[i][j][4 * g + 1] = bv.y; acc[i][j][4 * g + 2] = bv.z; acc[i][j][4 * g + 3] = bv.w;
        }
      }
    s16x8 fw[2][4], fx[2][2];
    for (int kt = 0; kt < nk; ++kt, ++s) {
#ifdef GEMM_STAMP
      unsigned long long st0, st1, st2, st3;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st0)::"memory");
      __builtin_amdgcn_sched_barrier(0);
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of step s have landed
      __builtin_amdgcn_s_barrier();                      // ... and everybody's; everyone is done with step s-1's stage
#ifdef GEMM_STAMP
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st1)::"memory");
      __builtin_amdgcn_sched_barrier(0);
#endif
      // HALF-STEP STAGGER between the two wave halves (each SIMD hosts one wave of each): waves 0-3 fire their LDS-DMA pieces
      // for step s+1 right after the barrier and then compute; waves 4-7 compute first and fire afterwards.  While one half is
      // parked in the vector-memory pipe the other half owns the matrix pipe - no extra synchronisation involved.
      const bool more = s + 1 < total;
      const bool late = wave >= 4;
      if (more && !late) issue();
#ifdef GEMM_STAMP
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st2)::"memory");
      __builtin_amdgcn_sched_barrier(0);
#endif
      const unsigned char* sw = smem + (s & 1) * ST6;
      const unsigned char* sx = sw + W6_BYTES;
      // software-pipelined fragments: the reads of k16-step q+1 are issued BEFORE the 8 MFMAs of step q (two register sets,
      // pinned with sched_barrier: left alone, hipcc re-serialises this into read -> lgkmcnt(0) -> MFMA groups and
      // every group pays the full LDS latency - 55 % of the wave time was spent parked on those waits)
      // ---- software pipeline ACROSS the barrier: the fragments of the step's last k16 group are read but their 8 MFMAs are
      // deferred to the top of the next step, where they cover the LDS latency of the new stage's first fragment reads
      // (without this, both waves of a SIMD sit in lgkmcnt waits right after every barrier: the compute loop alone, no
      // global traffic at all, ran at 2.2x its MFMA time).  Two register sets, order pinned with sched_barrier.
#define LDFRAG(buf, q)                                                                                            \
  {                                                                                                               \
    const int c_ = (q) * 2 + lh;                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
        fw[buf][i] = *reinterpret_cast<const s16x8*>(sw + swz128(wn * 128 + i * 32 + lr, c_));                    \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                 \
        fx[buf][j] = *reinterpret_cast<const s16x8*>(sx + swz128(wm * 64 + j * 32 + lr, c_));                     \
  }
#define MMA8(buf)                                                                                                 \
  {                                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                   \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[buf][i], fx[buf][j], acc[i][j], 0, 0, 0);          \
  }
      LDFRAG(0, 0)
      __builtin_amdgcn_sched_barrier(0);
      if (kt > 0) MMA8(1)  // deferred group of the previous step (wave-uniform)
      __builtin_amdgcn_sched_barrier(0);
      LDFRAG(1, 1)
      __builtin_amdgcn_sched_barrier(0);
      MMA8(0)
      __builtin_amdgcn_sched_barrier(0);
      LDFRAG(0, 2)
      __builtin_amdgcn_sched_barrier(0);
      MMA8(1)
      __builtin_amdgcn_sched_barrier(0);
      LDFRAG(1, 3)
      __builtin_amdgcn_sched_barrier(0);
      MMA8(0)
      __builtin_amdgcn_sched_barrier(0);
      if (kt == nk - 1) MMA8(1)  // last step of the tile: nothing to hide behind, flush
      if (more && late) issue();
#undef LDFRAG
#undef MMA8
#ifdef GEMM_STAMP
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st3)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      sw_ += st1 - st0; si_ += st2 - st1; sc_ += st3 - st2;
#endif
    }
    // ---- epilogue: image aliased onto the stage consumed by the tile's last step (free until issue(s+1) of the next step)
#ifdef GEMM_STAMP
    unsigned long long se0, se1;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(se0)::"memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
    __builtin_amdgcn_s_barrier();  // all waves done reading that stage
    bf16_t* ep = reinterpret_cast<bf16_t*>(smem + ((s - 1) & 1) * ST6) + wave * 64 * EPQ;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nn = 8 * g + 4 * lh;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          uint2 pk;
          pk.x = (uint32_t)f2bf(acc[i][j][4 * g]) | ((uint32_t)f2bf(acc[i][j][4 * g + 1]) << 16);
          pk.y = (uint32_t)f2bf(acc[i][j][4 * g + 2]) | ((uint32_t)f2bf(acc[i][j][4 * g + 3]) << 16);
          *reinterpret_cast<uint2*>(ep + (j * 32 + lr) * EPQ + nn) = pk;
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 16 + (lane >> 2), cc = (lane & 3) * 8;
        const int gm = m0 + wm * 64 + row;
        const uint4 vv = *reinterpret_cast<const uint4*>(ep + row * EPQ + cc);
#ifdef GEMM_ABL_NO_STORE
        if (gm < M && vv.x == 0x12345678u)
#else
        if (gm < M)
#endif
          *reinterpret_cast<uint4*>(Y + (size_t)gm * N + n0 + wn * 128 + i * 32 + cc) = vv;
      }
      __builtin_amdgcn_wave_barrier();
    }
#ifdef GEMM_STAMP
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(se1)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    se_ += se1 - se0;
#endif
  }
#ifdef GEMM_STAMP
  unsigned long long tend;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tend)::"memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (lane == 0) {  // diagnostic build only: the stamp sums overwrite the head of Y (its outputs are not used)
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(Y) + ((size_t)blockIdx.x * 8 + wave) * 8;
    dbg[0] = sw_; dbg[1] = si_; dbg[2] = sc_; dbg[3] = se_; dbg[4] = tend - tstart; dbg[5] = (unsigned long long)total;
  }
#endif
}

// =====================================================================================================================
// W-STATIONARY variant (K <= 512, N % 256 == 0): the weights never move again after the prologue.
//
// What held the full-row kernel above at 27 % of the MFMA roof: every 128-row tile re-streams all of W (512 KB) through
// LDS-DMA - 80 one-KiB pieces per K-step and CU, each blocking its wave for ~260 cycles in the CU's single vector-memory
// pipe - and pays a barrier per 64-deep K-step.  Here a wave keeps its 32 output columns x K of W in REGISTERS as ready-made
// MFMA A-fragments (K/16 fragments x 4 VGPRs = 128 VGPRs at K = 512) for the whole kernel; a workgroup (8 waves) owns 256
// columns and streams 32-row blocks of X (32 KB at K = 512: 32 one-KiB pieces per block and CU instead of 640 per 128 rows)
// through a 4-slot LDS ring.  Per block a wave runs ONE dependent chain of K/16 + 1 v_mfma_f32_32x32x16_bf16 on a single 32x32
// accumulator tile (a single accumulation chain of this instruction runs at the issue rate), each fed by one ds_read_b128
// of the X block; one barrier per block.  N / 256 workgroups share a row block; they sit on the same XCD (blockIdx -> (xcd, slot)
// map) so that the second reader of an X block hits that XCD's L2 and HBM sees X once.
//   * Bias: one more MFMA per block whose A-fragment holds the bias split into two bf16 (hi + lo: 16 mantissa bits) against a
//     constant B-fragment of two ones - no bias registers, loads or adds in the epilogue.
//   * Epilogue without LDS: the accumulator (lane = one row, four groups of 4 consecutive columns) is packed with v_cvt_pk_bf16_f32,
//     v_permlane32_swap joins the two half-waves' pieces into 16 bytes per lane, two stores of 32 rows x 32 bytes.  (Through a
//     wave-private LDS image the three dependent LDS round trips cost ~1700 cycles per block next to the other waves' fragment reads.)
//   * Stagger: behind the block's barrier waves 0-3 go straight to the matrix pipe and issue their LDS-DMA pieces (for block b + 3)
//     after the chain; waves 4-7 issue their pieces first and then run the chain.  The two waves of a SIMD (w, w + 4) thus run
//     about half a block apart - one on the matrix pipe while the other issues memory operations and converts - and every wave's
//     stores come last, far behind the DMA burst (issued right behind it they queued in the CU's memory pipe for ~2000 cycles).
// LDS image of a block: row r (K*2 bytes) | 16-byte chunk c stored at position c ^ (r & 15): conflict-free for the 16-lane groups of
// ds_read_b128 (32 different rows per wave read); LDS-DMA writes lane-linear, so the XOR is applied to the SOURCE chunk of every lane.
constexpr int WS_ROWS = 32, WS_SLOTS = 4, WS_T = 512, WS_COLS = 256;
#ifndef WS_PF
#define WS_PF 8   // X fragments in flight per wave
#endif

// 16 bytes per lane from global memory straight into LDS at (wave-uniform byte address lds_dst) + 16 * lane.  Inline asm, not the
// builtin: hipcc treats the builtin as a store to LDS that any later LDS access may alias and drains it (s_waitcnt vmcnt(0)) in
// front of the next ds_read / ds_write - which is every block here.  Hidden in asm the transfer is ordered only by the counted
// s_waitcnt vmcnt(N) + barrier below.  M0 carries the LDS address; it is compiler-reserved, so it is saved and restored inside the
// statement (s_nop: the SALU-writes-M0 -> LDS-DMA wait state).
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

template <int KK>
__global__ __launch_bounds__(WS_T, 2) void linear_bf16_wstat_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                                     const float* __restrict__ bias, bf16_t* __restrict__ Y, int M,
                                                                     int N, int ncol, int n_workers, int n_blocks) {
  constexpr int RB = KK * 2;                   // bytes per X row
  constexpr int BLK = WS_ROWS * RB;            // bytes per block
  constexpr int CPR = RB / 16;                 // 16-byte chunks per row
  constexpr int RPP = 1024 / RB;               // rows per 1-KiB LDS-DMA piece (1, 2 or 4)
  constexpr int PPL = BLK / 1024 / 8;          // LDS-DMA pieces per wave and block (4, 2 or 1)
  constexpr int NKS = KK / 16;                 // MFMAs per block and wave (+ 1 for the bias)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [WS_SLOTS blocks]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int colgrp = slot % ncol, worker = (slot / ncol) * 8 + xcd;
  const int n0w = colgrp * WS_COLS + wave * 32;  // this wave's 32 output columns
  const int my_blocks = worker < n_workers ? (n_blocks - worker + n_workers - 1) / n_workers : 0;

  // ---- prologue: this wave's W rows as A-fragments (lane: row n0w + lr, k = 16 ks + 8 lh .. + 8) and the bias fragment
  // (k = 0: bf16(b), k = 1: bf16(b - bf16(b)), rest zero; lanes of the upper half (k = 8 ..) all zero)
  s16x8 wf[NKS + 1];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) wf[ks] = *reinterpret_cast<const s16x8*>(W + (size_t)(n0w + lr) * KK + ks * 16 + lh * 8);
  {
    const float bv = (bias && lh == 0) ? bias[n0w + lr] : 0.f;
    const bf16_t hi = f2bf(bv), lo = f2bf(bv - bf2f(hi));
#pragma unroll
    for (int j = 0; j < 8; ++j) wf[NKS][j] = 0;
    wf[NKS][0] = (short)hi;
    wf[NKS][1] = (short)lo;
  }
  s16x8 ones;  // B-fragment of the bias MFMA: X "row" with ones at k = 0, 1
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = 0;
  if (lh == 0) { ones[0] = (short)0x3f80; ones[1] = (short)0x3f80; }
  // (the compiler waits for these loads where they are first USED: make that here, in front of the first LDS-DMA - its wait is a
  // vmcnt(0), and placed at the first MFMA it would also drain the three blocks of X requested in between)
#pragma unroll
  for (int ks = 0; ks <= NKS; ++ks) asm volatile("" : "+v"(wf[ks]));

  // ---- LDS-DMA: a block is 8 * PPL pieces of 1 KiB; wave w issues pieces w * PPL .. + PPL.  Lane -> (row in piece, position); the
  // lane fetches the chunk that belongs at its position: pos ^ (row & 15)
  const int prow = lane / CPR, ppos = lane % CPR;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_void*)smem;  // byte address of the ring inside the workgroup's LDS allocation
  auto issue = [&](int bi) {  // bi: this worker's block index (0 .. my_blocks)
    const int m0 = (worker + bi * n_workers) * WS_ROWS;
    const uint32_t dst = lds_base + (uint32_t)((bi % WS_SLOTS) * BLK + wave * PPL * 1024);
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      const int row = (wave * PPL + j) * RPP + prow;
      const int chunk = ppos ^ (row & 15);
      const bf16_t* src = X + (size_t)min(m0 + row, M - 1) * KK + chunk * 8;  // rows past M re-read the last row (never stored)
#ifdef WS_ABL_NO_LOAD
      if (bi < WS_SLOTS)   // timing-only build: only the first ring-full of blocks is really loaded
#endif
      glds16(src, dst + j * 1024);
    }
  };
  const bool late = wave >= 4;
#pragma unroll
  for (int i = 0; i < WS_SLOTS - 1; ++i)
    if (i < my_blocks) issue(i);

  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  auto pack2 = [](float a, float b) -> uint32_t {  // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
  };
#ifdef WS_STAMP
  unsigned long long sta[6] = {0, 0, 0, 0, 0, 0}, stl;
#define WSTAMP(i) { unsigned long long n_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n_)::"memory"); __builtin_amdgcn_sched_barrier(0); sta[i] += n_ - stl; stl = n_; }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stl)::"memory");
#else
#define WSTAMP(i)
#endif
  for (int bi = 0; bi < my_blocks; ++bi) {
    // This wave's pieces of block bi have landed once only the operations issued AFTER them are outstanding (vmcnt counts loads,
    // LDS-DMA and stores together, in issue order).  Per iteration j every wave issues [PPL pieces of block j+3][2 stores of block j]
    // (waves 4-7: pieces, chain, stores; waves 0-3: chain, pieces, stores); behind the pieces of block bi therefore sit the stores of
    // the iterations bi-3 .. bi-1 that exist and the pieces of the blocks bi+1, bi+2 that exist.
    {
      const int ahead = min(my_blocks - 1 - bi, WS_SLOTS - 2), st = min(bi, 3);
#define VMW(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
      if (ahead == 2) { if (st == 0) VMW(2 * PPL); else if (st == 1) VMW(2 * PPL + 2); else if (st == 2) VMW(2 * PPL + 4); else VMW(2 * PPL + 6); }
      else if (ahead == 1) { if (st == 0) VMW(PPL); else if (st == 1) VMW(PPL + 2); else if (st == 2) VMW(PPL + 4); else VMW(PPL + 6); }
      else VMW(0);
#undef VMW
    }
    WSTAMP(0)
    __builtin_amdgcn_s_barrier();  // the pieces of block bi are in LDS; everyone is done reading block bi - 1
    WSTAMP(1)
    const bool more = bi + WS_SLOTS - 1 < my_blocks;
    if (late && more) issue(bi + WS_SLOTS - 1);  // into the slot block bi - 1 has just left
    WSTAMP(2)
    const unsigned char* xb = smem + (bi % WS_SLOTS) * BLK + lr * RB;
    int sw = lr & 15;              // swizzle key through an opaque copy: the NKS read offsets are rebuilt per block (one VALU op in the
    asm volatile("" : "+v"(sw));   // shadow of each MFMA) instead of being hoisted out of the loop into NKS registers - and spilled
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    // fragment reads run PF MFMAs ahead of their use (left to itself hipcc issues each read right in front of its MFMA and every
    // pair of MFMAs waits out an LDS round trip); the order is pinned with sched_group_barrier
    constexpr int PF = NKS < WS_PF ? NKS : WS_PF;
    s16x8 xf[PF];
#pragma unroll
    for (int ks = 0; ks < PF; ++ks) xf[ks] = *reinterpret_cast<const s16x8*>(xb + (((ks * 2 + lh) ^ sw) << 4));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[NKS], ones, acc, 0, 0, 0);  // bias
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], xf[ks % PF], acc, 0, 0, 0);
      if (ks + PF < NKS) xf[ks % PF] = *reinterpret_cast<const s16x8*>(xb + ((((ks + PF) * 2 + lh) ^ sw) << 4));
    }
    __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);  // DS_READ x PF
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);      // MFMA (bias)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);    // MFMA
      if (ks + PF < NKS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS_READ
    }
    WSTAMP(3)
    if (!late && more) issue(bi + WS_SLOTS - 1);  // waves 0-3: behind their chain (the matrix pipe is the partner wave's now)
    WSTAMP(5)
    // ---- epilogue: D[i][j], i = n = (q & 3) + 8 (q >> 2) + 4 lh, j = m = lr.  Per group g = q >> 2 a lane holds 4 consecutive columns
    // (8 bytes at column 8 g + 4 lh); v_permlane32_swap hands the lower half-wave both halves of the even groups and the upper one
    // both halves of the odd groups: 16 contiguous bytes per lane, columns 16 (g >> 1) + 8 lh .. + 8
    const int m0 = (worker + bi * n_workers) * WS_ROWS;
    uint32_t pk[8];
#pragma unroll
    for (int g = 0; g < 4; ++g) { pk[2 * g] = pack2(acc[4 * g], acc[4 * g + 1]); pk[2 * g + 1] = pack2(acc[4 * g + 2], acc[4 * g + 3]); }
#pragma unroll
    for (int p = 0; p < 2; ++p) {  // groups (2p, 2p + 1)
      const auto r0 = __builtin_amdgcn_permlane32_swap(pk[4 * p], pk[4 * p + 2], false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(pk[4 * p + 1], pk[4 * p + 3], false, false);
      // lower half: [own even-group half | partner's even-group half]; upper half: [partner's odd-group half | own odd-group half]
      const uint4 v = make_uint4(r0[0], r1[0], r0[1], r1[1]);
      const int gm = m0 + lr;
#ifdef WS_ABL_NO_STORE
      if (gm < M && v.x == 0x12345678u)
#else
      if (gm < M)   // (always two store instructions per block and wave, which the vmcnt counts above rely on)
#endif
        *reinterpret_cast<uint4*>(Y + (size_t)gm * N + n0w + 16 * p + 8 * lh) = v;
    }
    WSTAMP(4)
  }
#ifdef WS_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4)) {  // diagnostic build only: the sums overwrite the head of Y
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(Y) + (wave ? 8 : 0);
    for (int i = 0; i < 6; ++i) dbg[i] = sta[i];
    dbg[6] = (unsigned long long)my_blocks;
  }
#endif
#undef WSTAMP
}

}  // namespace

extern "C" int tamtr_linear_bf16(const void* X, const void* W, const float* bias, void* Y, int M, int N, int K, void* stream) {
  if (!X || !W || !Y || M <= 0 || N <= 0 || K <= 0) return TAMTR_EINVAL;
  if (K % BK || N % BN) return TAMTR_EUNSUP;
  // kernel choice (A/B switch TAMTR_GEMM = ws1 | tile), read once when the library is loaded - not per call
  static const int choice = [] { const char* e = getenv("TAMTR_GEMM"); return (e && e[0] == 't') ? 0 : 1; }();
  if (choice >= 1 && (K == 512 || K == 256 || K == 128) && N % WS_COLS == 0 && 32 % (N / WS_COLS) == 0) {
    // W-stationary: 256 persistent workgroups, N / 256 of them (same XCD) per row block
    const int ncol = N / WS_COLS, n_workers = (32 / ncol) * 8, n_blocks = (M + WS_ROWS - 1) / WS_ROWS;
    const size_t lds = (size_t)WS_SLOTS * WS_ROWS * K * 2;
#define LAUNCH_WS(KK)                                                                                                        \
  {                                                                                                                          \
    if (hipFuncSetAttribute((const void*)linear_bf16_wstat_kernel<KK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return TAMTR_ELAUNCH;   /* per call: the attribute is per DEVICE (a process-wide "done" flag would skip the second GPU) */ \
    hipLaunchKernelGGL(linear_bf16_wstat_kernel<KK>, dim3(256), dim3(WS_T), lds, (hipStream_t)stream, (const bf16_t*)X,      \
                       (const bf16_t*)W, bias, (bf16_t*)Y, M, N, ncol, n_workers, n_blocks);                                 \
  }
    if (K == 512) LAUNCH_WS(512) else if (K == 256) LAUNCH_WS(256) else LAUNCH_WS(128)
#undef LAUNCH_WS
    return tamtr_launch_status();
  }
  if (N % TN == 0 && K % TK == 0) {  // full-row tiles: X read once (the value projection shape)
    const int mbl = (M + TM - 1) / TM, nbl = N / TN;
    if ((long long)mbl * nbl > 0x7fffffffLL) return TAMTR_EUNSUP;
    const size_t lds5 = (size_t)NST * ST_BYTES;  // 4 x 40 KB ring = all 160 KB of a CU (epilogue image aliases a stage)
    static_assert(NST * ST_BYTES <= 160 * 1024 && EP_BYTES <= ST_BYTES, "LDS budget");
    if (hipFuncSetAttribute((const void*)linear_bf16_n512_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5) != hipSuccess) return TAMTR_ELAUNCH;
    const int tiles = mbl * nbl;
    const int grid = tiles < 256 ? tiles : 256;  // one persistent workgroup per CU
    if (K % K6 == 0) {  // 128-B pieces, 2-stage ring (measured 5 % faster than the 64-B / 4-stage ring below)
      if (hipFuncSetAttribute((const void*)linear_bf16_n512_k64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ST6) != hipSuccess) return TAMTR_ELAUNCH;
      hipLaunchKernelGGL(linear_bf16_n512_k64_kernel, dim3((unsigned)grid), dim3(T5), 2 * ST6, (hipStream_t)stream,
                         (const bf16_t*)X, (const bf16_t*)W, bias, (bf16_t*)Y, M, N, K, nbl, tiles);
      return tamtr_launch_status();
    }
    hipLaunchKernelGGL(linear_bf16_n512_kernel, dim3((unsigned)grid), dim3(T5), lds5, (hipStream_t)stream, (const bf16_t*)X,
                       (const bf16_t*)W, bias, (bf16_t*)Y, M, N, K, nbl, tiles);
    return tamtr_launch_status();
  }
  const int m_blocks = (M + BM - 1) / BM, n_blocks = N / BN;
  const long long blocks = (long long)((m_blocks + 7) / 8) * 8 * n_blocks;
  if (blocks > 0x7fffffffLL) return TAMTR_EUNSUP;
  const size_t lds = 2 * 2 * TILE_BYTES;  // 64 KB (>= the 4 x 64 x EP_PITCH x 2 B epilogue image)
  static_assert(4 * 64 * EP_PITCH * 2 <= 2 * 2 * TILE_BYTES, "epilogue image must fit the staging LDS");
  hipLaunchKernelGGL(linear_bf16_kernel, dim3((unsigned)blocks), dim3(GT), lds, (hipStream_t)stream, (const bf16_t*)X,
                     (const bf16_t*)W, bias, (bf16_t*)Y, M, N, K, m_blocks, n_blocks);
  return tamtr_launch_status();
}
