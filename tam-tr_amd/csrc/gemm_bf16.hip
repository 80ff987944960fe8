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

}  // namespace

extern "C" int tamtr_linear_bf16(const void* X, const void* W, const float* bias, void* Y, int M, int N, int K, void* stream) {
  if (!X || !W || !Y || M <= 0 || N <= 0 || K <= 0) return TAMTR_EINVAL;
  if (K % BK || N % BN) return TAMTR_EUNSUP;
  const int m_blocks = (M + BM - 1) / BM, n_blocks = N / BN;
  const long long blocks = (long long)((m_blocks + 7) / 8) * 8 * n_blocks;
  if (blocks > 0x7fffffffLL) return TAMTR_EUNSUP;
  const size_t lds = 2 * 2 * TILE_BYTES;  // 64 KB (>= the 4 x 64 x EP_PITCH x 2 B epilogue image)
  static_assert(4 * 64 * EP_PITCH * 2 <= 2 * 2 * TILE_BYTES, "epilogue image must fit the staging LDS");
  hipLaunchKernelGGL(linear_bf16_kernel, dim3((unsigned)blocks), dim3(GT), lds, (hipStream_t)stream, (const bf16_t*)X,
                     (const bf16_t*)W, bias, (bf16_t*)Y, M, N, K, m_blocks, n_blocks);
  return tamtr_launch_status();
}
