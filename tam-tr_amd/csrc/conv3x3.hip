// conv3x3.hip - the value branch of the text gate: proj_conv = Conv2d(c, c, 3, stride 1, pad 1, no bias) followed by BatchNorm
// (ultralytics/nn/extra_modules/block.py:205,223) as an implicit-GEMM MFMA kernel over channels-last bf16 maps,
// with the BatchNorm's batch statistics taken in the epilogue.  The gate kernel (gate.hip: gate_cl_fwd_kernel) applies the affine
// in its own load, so the branch is: this kernel -> one per-channel combine -> the gate.  Forward only: every TAM-TR instance of the
// gate is the discarded evaluation of TIAGELAN, whose only live result is the running statistics.
//
// GEMM view: out[p][co] = sum over (tap, ci) of x[p + tap][ci] * w[co][ci][tap]; M = B*H*W pixels, N = C2, K = 9*C1.
// A workgroup owns an 8 x 16 pixel tile x 64 output channels.  K is walked in chunks of 32 input channels: per chunk the 10 x 18
// HALO tile of the input (12.8 KB at LDS row pitch 20) and the chunk's 9 x 64 x 32 weights (36.9 KB) are staged in LDS once and serve
// all nine taps - the input map is read once per 64 output channels, never nine times.  Both LDS tiles hold 64-byte rows (32 bf16)
// whose four 16-byte pieces are XOR-swizzled with bits 2-3 of the row index, which makes every MFMA fragment read conflict-free
// without padding (sw64 below).  The next chunk's global loads are issued before the chunk's 36 MFMAs per wave and written to LDS
// after them; inside a chunk the six fragment reads of a tap run one tap ahead of its four MFMAs.
// Measured (MI355X, 16 images, profiles/r03_proj_conv.txt): 46 - 58 us per site = 30.2 GFLOP at 520 - 650 TFLOP/s; with the weight
// repacking and the per-channel combine 66 - 78 us against 83 - 107 us for the library convolution + the statistics kernel.
// MFMA orientation: A = weights (m = output channel), B = pixels (n = pixel), so a lane ends up with 4 consecutive channels of
// one pixel per accumulator group: 8-byte pieces of a [pixel][channel] LDS tile, from which the workgroup stores whole 128-byte
// channel runs and sums the per-channel statistics of the values AS STORED (bf16-rounded, what a separate statistics pass over
// the stored map would see).
#include "common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int CV_TH = 8, CV_TW = 16;             // pixel tile
constexpr int CV_HW = CV_TW + 2, CV_HH = CV_TH + 2;
constexpr int CV_HALO = CV_HH * CV_HW;           // 180 halo pixels
constexpr int CV_HP = 20;                        // their row pitch in LDS (pixel (hy, hx) is LDS row hy * 20 + hx: see sw64)
constexpr int CV_NT = 64;                        // output channels per workgroup
constexpr int CV_CK = 32;                        // input channels per LDS stage
constexpr int CV_THREADS = 256;
constexpr int CV_XI = (CV_HALO * 4 + CV_THREADS - 1) / CV_THREADS;  // 16-byte halo pieces per thread (3)
constexpr int CV_WI = 9 * CV_NT * 4 / CV_THREADS;                   // 16-byte weight pieces per thread (9)
constexpr int CV_SX_BYTES = CV_HH * CV_HP * 64;   // 12 800
constexpr int CV_SW_BYTES = 9 * CV_NT * 64;      // 36 864

// byte offset of 16-byte piece c (0..3) of 64-byte row r.  ds_read_b128 is served in four groups of 16 lanes - {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31} and the same + 32 - over 64 banks (MI355X_MICROARCH.md, LDS): with the piece index XOR-ed by bits 2-3 of the
// row, 32 consecutive rows (the weight fragments) and an 8 x 4 pixel block at row pitch 20 (the pixel fragments, for every tap) both put
// each group's 16 pieces on 16 different 16-byte bank sets (checked by enumeration: 4 LDS cycles per read, the minimum; bits 1-2 of
// the row and a 2 x 16 block at pitch 18 took 8)
__device__ __forceinline__ int sw64(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

// w [C2][C1][3][3] (f32 or bf16) -> wpk bf16 [C1/32][9][C2][32]: a stage's rows (one tap, 64 channels) are contiguous 64-byte rows
template <typename T>
__global__ void conv3x3_pack_kernel(const T* __restrict__ w, bf16_t* __restrict__ wpk, int C1, int C2) {
  const int total = 9 * C1 * C2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int ci_l = i % CV_CK, co = (i / CV_CK) % C2, tap = (i / (CV_CK * C2)) % 9, cc = i / (CV_CK * C2 * 9);
    const int ci = cc * CV_CK + ci_l;
    wpk[i] = f2bf(Elt<T>::ld(w + ((size_t)co * C1 + ci) * 9 + tap));
  }
}

__global__ __launch_bounds__(CV_THREADS) void conv3x3_cl_kernel(const bf16_t* __restrict__ x, size_t ld_x, const bf16_t* __restrict__ wpk,
                                                                 bf16_t* __restrict__ y, float* __restrict__ part, int H, int W, int C1,
                                                                 int C2, int tiles_x, int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cv_smem[];
  unsigned char* s_x = cv_smem;
  unsigned char* s_w = cv_smem + CV_SX_BYTES;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE), lane = tid % WAVE, lr = lane & 31, lh = lane >> 5;
  const int tile = blockIdx.x, n0 = blockIdx.y * CV_NT;
  const int S = gridDim.x;
  const int b = tile / (tiles_x * tiles_y), ty0 = ((tile / tiles_x) % tiles_y) * CV_TH, tx0 = (tile % tiles_x) * CV_TW;
  const int nchunk = C1 / CV_CK;

  // ---- staging plan of this thread (the same for every chunk): source element offsets (or -1: zero) and LDS byte offsets
  long long xsrc[CV_XI];
  int xdst[CV_XI];
  bool xok[CV_XI];
#pragma unroll
  for (int j = 0; j < CV_XI; ++j) {
    const int i = tid + j * CV_THREADS;
    const int hp = min(i, CV_HALO * 4 - 1) >> 2, c = i & 3;
    const int hy = hp / CV_HW, hx = hp % CV_HW;
    const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
    const bool in = i < CV_HALO * 4 && gy >= 0 && gy < H && gx >= 0 && gx < W;
    // (a piece outside the image is loaded from the map's first pixel and replaced by zeros when it is written to LDS: no load sits
    // under a branch, and nothing waits for a load where it is issued)
    xsrc[j] = in ? (long long)(((size_t)b * H + gy) * W + gx) * (long long)ld_x + c * 8 : 0;
    xok[j] = in;
    xdst[j] = i < CV_HALO * 4 ? sw64(hy * CV_HP + hx, c) : -1;
  }
  // (staging written out as macros: arrays captured by reference in a lambda end up in scratch memory with this compiler)
  u32x4 xr[CV_XI], wr[CV_WI];
#define CV_FETCH(cc)                                                                                                         \
  {                                                                                                                          \
    _Pragma("unroll") for (int j = 0; j < CV_XI; ++j) {                                                                      \
      xr[j] = *reinterpret_cast<const u32x4*>(x + xsrc[j] + (cc) * CV_CK);                                                   \
    }                                                                                                                        \
    _Pragma("unroll") for (int j = 0; j < CV_WI; ++j) {                                                                      \
      const int i = tid + j * CV_THREADS, row = i >> 2, c = i & 3; /* row = tap * 64 + co */                                 \
      wr[j] = *reinterpret_cast<const u32x4*>(wpk + (((size_t)(cc) * 9 + (row >> 6)) * C2 + n0 + (row & 63)) * CV_CK + c * 8); \
    }                                                                                                                        \
  }
#define CV_COMMIT()                                                                                                          \
  {                                                                                                                          \
    _Pragma("unroll") for (int j = 0; j < CV_XI; ++j)                                                                        \
      if (xdst[j] >= 0) *reinterpret_cast<u32x4*>(s_x + xdst[j]) = xok[j] ? xr[j] : u32x4{0u, 0u, 0u, 0u};                   \
    _Pragma("unroll") for (int j = 0; j < CV_WI; ++j) {                                                                      \
      const int i = tid + j * CV_THREADS;                                                                                    \
      *reinterpret_cast<u32x4*>(s_w + sw64(i >> 2, i & 3)) = wr[j];                                                          \
    }                                                                                                                        \
  }

  f32x16 acc[2];
#pragma unroll
  for (int q = 0; q < 16; ++q) { acc[0][q] = 0.f; acc[1][q] = 0.f; }
  // this lane's pixel in the halo tile (tap (0, 0)): wave w owns the tile's columns 4 w .. 4 w + 3, all 8 rows (lane = row * 4 + column)
  const int hp0 = (lr >> 2) * CV_HP + 4 * wave + (lr & 3);
  const int wsw = (lr >> 2) & 3;  // swizzle key of the weight rows this lane reads (row = tap * 64 + 32 nb + lr)

  CV_FETCH(0)
  for (int cc = 0; cc < nchunk; ++cc) {
    __syncthreads();  // previous chunk's fragments all read
    CV_COMMIT()
    __syncthreads();
    if (cc + 1 < nchunk) CV_FETCH(cc + 1)
    // the six fragments of a tap (pixels, weights of both channel blocks; two k-steps) are requested one tap ahead of their four
    // MFMAs: left to itself the compiler keeps about one k-step in flight and every MFMA waits out an LDS round trip
    s16x8 fr[2][6];
#define CV_LOAD_TAP(buf, tap)                                                                                                \
  {                                                                                                                          \
    const int hp = hp0 + ((tap) / 3) * CV_HP + (tap) % 3;                                                                    \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                                       \
      const int c = 2 * ks + lh;                                                                                             \
      const unsigned char* wrow = s_w + ((tap) * CV_NT + lr) * 64 + ((c ^ wsw) << 4);                                        \
      fr[buf][3 * ks] = *reinterpret_cast<const s16x8*>(s_x + sw64(hp, c));                                                  \
      fr[buf][3 * ks + 1] = *reinterpret_cast<const s16x8*>(wrow);                                                           \
      fr[buf][3 * ks + 2] = *reinterpret_cast<const s16x8*>(wrow + 32 * 64);                                                 \
    }                                                                                                                        \
  }
    CV_LOAD_TAP(0, 0)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) CV_LOAD_TAP((tap + 1) & 1, tap + 1)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[tap & 1][3 * ks + 1], fr[tap & 1][3 * ks], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[tap & 1][3 * ks + 2], fr[tap & 1][3 * ks], acc[1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef CV_LOAD_TAP
  }

  // ---- epilogue.  D[i][j]: i = channel (q & 3) + 8 (q >> 2) + 4 lh (+ 32 nb), j = pixel lr of the wave's 8 x 4 block.
  // LDS tile [128 pixels][64 channels] bf16 (128-byte rows) over the weight stage, 16-byte pieces XOR-ed with cv_okey(pixel).
  // Round 4 (`SQ_LDS_BANK_CONFLICT` 8.3e5 = 108 cycles per wave against a layout called conflict-free): enumerated
  // per instruction with the guide's lane groups - the staging ds_write_b128 (8 lanes = two 64-byte rows r, r + 1 or r, r + 3: 4 r and
  // 4 (r + 1) fall into different 64-byte halves of the 32-bank window, the XOR only permutes inside a half) and the fragment
  // ds_read_b128 are conflict-free; the conflicts were HERE: a 16-lane group of these ds_write_b64 is 4 tile rows x 4 columns, and with
  // the key `pixel & 7` = 4 (wave & 1) + column the four rows of a column shared one 16-byte piece: 4-way, 8 stores x 12 extra cycles = 96
  // per wave.  The key now also takes the row's low bit, two rows per piece: 2-way, the minimum (the 8-byte half of a piece is fixed by
  // the accumulator layout: 16 lanes of one half-wave can reach only 8 of the 16 8-byte slots of a 128-byte window).
  __syncthreads();
  unsigned char* s_o = s_w;
  auto cv_okey = [](int p) { return (p & 3) | ((((p >> 2) ^ (p >> 4)) & 1) << 2); };
  {
    const int p = (lr >> 2) * CV_TW + 4 * wave + (lr & 3);  // tile pixel: row * 16 + column
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = 32 * nb + 8 * g + 4 * lh;
        uint2 v;
        v.x = (uint32_t)f2bf(acc[nb][4 * g]) | ((uint32_t)f2bf(acc[nb][4 * g + 1]) << 16);
        v.y = (uint32_t)f2bf(acc[nb][4 * g + 2]) | ((uint32_t)f2bf(acc[nb][4 * g + 3]) << 16);
        *reinterpret_cast<uint2*>(s_o + p * 128 + (((co >> 3) ^ cv_okey(p)) << 4) + (co & 7) * 2) = v;
      }
  }
  __syncthreads();
  // thread = (channel group j of 8, pixel t / 8 + 32 i): whole 128-byte channel runs per pixel; statistics of what is stored
  const int j8 = tid & 7;
  float sm[8], sq[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { sm[k] = 0.f; sq[k] = 0.f; }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = (tid >> 3) + 32 * i;  // tile pixel: row p / 16, column p % 16
    const int gy = ty0 + (p >> 4), gx = tx0 + (p & 15);
    const uint4 v = *reinterpret_cast<const uint4*>(s_o + p * 128 + ((j8 ^ cv_okey(p)) << 4));
    if (gy < H && gx < W) {
      *reinterpret_cast<uint4*>(y + (((size_t)b * H + gy) * W + gx) * C2 + n0 + 8 * j8) = v;
      const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a = __uint_as_float(u[k] << 16), c = __uint_as_float(u[k] & 0xffff0000u);
        sm[2 * k] += a; sq[2 * k] = fmaf(a, a, sq[2 * k]);
        sm[2 * k + 1] += c; sq[2 * k + 1] = fmaf(c, c, sq[2 * k + 1]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int o = 8; o < WAVE; o <<= 1) { sm[k] += __shfl_xor(sm[k], o, WAVE); sq[k] += __shfl_xor(sq[k], o, WAVE); }
  float* s_st = reinterpret_cast<float*>(s_x);  // [4 waves][64 channels][2]
  if (lane < 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s_st[(wave * CV_NT + 8 * j8 + k) * 2] = sm[k];
      s_st[(wave * CV_NT + 8 * j8 + k) * 2 + 1] = sq[k];
    }
  }
  __syncthreads();
  if (tid < CV_NT) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int w = 0; w < CV_THREADS / WAVE; ++w) { s += s_st[(w * CV_NT + tid) * 2]; q += s_st[(w * CV_NT + tid) * 2 + 1]; }
    const float n = (float)(min(CV_TH, H - ty0) * min(CV_TW, W - tx0));
    const float mean = s / n;
    float* o = part + ((size_t)(n0 + tid) * S + tile) * 3;  // (count, mean, M2) of the tile, bn.hip's partial layout
    o[0] = n; o[1] = mean; o[2] = fmaxf(q - s * mean, 0.f);
  }
}

#undef CV_FETCH
#undef CV_COMMIT

}  // namespace

extern "C" int tamtr_conv3x3_tiles(int B, int H, int W) { return B * ((H + CV_TH - 1) / CV_TH) * ((W + CV_TW - 1) / CV_TW); }

extern "C" int tamtr_conv3x3_pack_weight(const void* w, void* wpk, int C1, int C2, int dtype, void* stream) {
  if (!w || !wpk || C1 <= 0 || C2 <= 0) return TAMTR_EINVAL;
  if (C1 % CV_CK || C2 % CV_NT) return TAMTR_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
  const int total = 9 * C1 * C2, blocks = (total + 255) / 256;
  if (dtype == TAMTR_F32) hipLaunchKernelGGL(conv3x3_pack_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)w, (bf16_t*)wpk, C1, C2);
  else if (dtype == TAMTR_BF16) hipLaunchKernelGGL(conv3x3_pack_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)w, (bf16_t*)wpk, C1, C2);
  else return TAMTR_EINVAL;
  return tamtr_launch_status();
}

extern "C" int tamtr_conv3x3_cl_stats_fwd(const void* x, long long ld_x, const void* wpk, void* y, float* running_mean, float* running_var,
                                          float* mean_rstd, float* partials, int B, int H, int W, int C1, int C2, float eps, float momentum,
                                          void* stream) {
  if (!x || !wpk || !y || !partials || B <= 0 || H <= 0 || W <= 0 || ld_x < C1) return TAMTR_EINVAL;
  if (C1 % CV_CK || C2 % CV_NT || ld_x % 8 || ((uintptr_t)x | (uintptr_t)wpk | (uintptr_t)y) % 16) return TAMTR_EUNSUP;
  const int tiles_y = (H + CV_TH - 1) / CV_TH, tiles_x = (W + CV_TW - 1) / CV_TW;
  const long long S = (long long)B * tiles_x * tiles_y;
  if (S > 0x7fffffffLL || C2 / CV_NT > 65535) return TAMTR_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(conv3x3_cl_kernel, dim3((unsigned)S, C2 / CV_NT), dim3(CV_THREADS), CV_SX_BYTES + CV_SW_BYTES, s, (const bf16_t*)x,
                     (size_t)ld_x, (const bf16_t*)wpk, (bf16_t*)y, partials, H, W, C1, C2, tiles_x, tiles_y);
  if (hipGetLastError() != hipSuccess) return TAMTR_ELAUNCH;
  if (!mean_rstd) return TAMTR_OK;
  return tamtr_bn_finalize(partials, mean_rstd, running_mean, running_var, C2, (int)S, eps, momentum, stream);
}
