// selscan.hip - selective scan (S6) forward + backward for the MEH VSSBlocks, gfx950, fp32.
//
// Replaces the external CUDA extension `selective_scan_cuda_core` the reference calls at
// ultralytics/nn/extra_modules/VManba/csms6s.py:257,267 (contract: vmamba.py:962-990).  Written from the recurrence,
// not from that extension (whose source is not in the reference tree):
//     dt = softplus(delta + dbias);  h_t = exp(dt_t A_n) h_{t-1} + dt_t B_{n,t} u_t;  y_t = sum_n C_{n,t} h_{n,t} + D u_t
//
// Mapping (wave64): ONE WAVE PER ROW (b, k*Dk+d); lanes run along TIME, 4 consecutive steps per lane, so a wave eats
// a CHUNK of 256 steps per iteration with perfectly coalesced 16-B/lane loads of u (and gy) and stores of y.
// Inside a chunk the linear recurrence is an associative scan on affine maps h -> a*h + b: 3 sequential steps inside
// the lane, then a 6-level scan across the 64 lanes with DPP row shifts / row broadcasts (VALU-rate cross-lane moves, no
// LDS round trip), then the carry h_n (wave-uniform) of the previous chunk is folded in.  The 16 states are looped; the
// y dot product stays inside the lane.  Chunk-boundary states are written out for the backward pass, which walks the
// chunks in reverse, recomputes h inside the chunk and runs the mirrored (suffix) scan for dL/dh.
// B/C (and dB/dC) are shared by the Dk rows of a (b, k) group: a workgroup holds rows of ONE group and stages the chunk's
// B/C tiles once in LDS.
//
// What the instruction stream looks like matters as much as its length here (both kernels are VALU-issue bound, one
// instruction per ~4 cycles per wave):
//   * no load sits under a branch: hipcc waits vmcnt(0) at the end of every conditional block that holds a load, which
//     turned the "prefetches" of the first version into synchronous round trips (in-kernel stamps: 22 % of the backward in
//     the row prologue, 10 % in the tile staging).  Addresses are clamped and results selected instead;
//   * hipcc never folds a DPP move into a float multiply or fma (it does for adds), so the scans are inline asm with
//     v_fmac_f32_dpp / v_mul_f32_dpp, two independent chains interleaved so that no DPP hazard nop is needed;
//   * per-row scalars (A, D, bias, chunk-entry / carried states) travel in ONE VGPR each (lane n holds entry n) and are
//     read per state with v_readlane, new carries are assembled with v_writelane: no LDS round trips, no branches.
#include "common.h"

constexpr int CHUNK_DEFAULT = 64 * 4;

namespace {

constexpr int NS = 16;         // d_state
constexpr int ITEMS = 4;       // time steps per lane
constexpr int CHUNK = WAVE * ITEMS;
// Checkpoints the forward leaves for the backward: the state after every 16-lane DPP row of the backward (64 steps), hstate
// [row][block][16], 1 byte per step and row.  The backward rebuilds the states entering its lanes from there with four in-row scan levels.
// Two denser forms were built and measured in round 3 and removed (profiles/r03_scan_checkpoints.txt): a checkpoint per LANE, read instead
// of rebuilt (13 % fewer vector instructions in the backward, 2 % less time, forward +13 % for 6.7 GB of stores at level 0), and one per
// QUAD of lanes (two scan levels instead of four: four DPP instructions fewer per state, the same time to the microsecond).
constexpr int BLK = 16 * ITEMS;   // checkpoint interval in steps
constexpr int FWD_ROWS = 8;    // waves (= rows of one (b,k) group) per workgroup, forward
constexpr int FWD_RPW = 4;     // rows per wave and chunk, forward (they share the staged tiles)
constexpr int BWD_WAVES = 4;   // waves per workgroup, backward
#ifndef SCAN_BWD_RPW
#define SCAN_BWD_RPW 8
#endif
constexpr int BWD_RPW = SCAN_BWD_RPW;  // rows handled one after the other by each wave per chunk
constexpr int BWD_ROWS = BWD_WAVES * BWD_RPW;  // rows of one (b,k) group per workgroup = one dB/dC slab
constexpr int RMAX = 32;       // largest dt rank built (d_model 512 / 16)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// softplus with torch's threshold (20); log(1+e^x) through the hardware exp2/log2 (abs err ~1e-7, the e^x branch keeps
// the relative accuracy for very negative x where 1 + e^x rounds to 1)
__device__ __forceinline__ float softplus_f(float x) {
  const float e = __builtin_amdgcn_exp2f(x * LOG2E);
  const float sp = __builtin_amdgcn_logf(1.f + e) * LN2;
  return x > 20.f ? x : (x < -10.f ? e : sp);
}

// scan step t lives at memory position t (rev = false) or L-1-t (rev = true: the two reversed scan directions of the
// cross-scan read and write the SAME buffers as the forward ones, back to front - still 16-B coalesced, descending).
// Row streams go through raw buffer loads / stores with the row as the buffer (base = the row's wave-uniform pointer,
// num_records = 4 L): a lane past the end of the row gets zeros / stores nothing by the hardware's range check, so neither
// a branch nor a select surrounds the memory instruction - hipcc can then count them (a store under `if (t < L)` makes it wait
// vmcnt(0) for the loads issued before it).  Two halves, so that a request can be issued a whole row before its use:
// `load4_issue` only loads, `load4_take` puts the four values into step order where they are consumed.
struct Raw4 { float v[ITEMS]; };
constexpr unsigned OOB = 0x7ffffff0u;  // byte offset no row reaches: the access is dropped by the range check
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const float* rowp, int L) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rowp), 0, L * 4, 0x00020000);
}
// P16: the row is a row of bf16 (the u / y / d(y) / d(u) planes of the bf16 mode: the recurrence itself stays fp32, only what crosses
// HBM is narrowed); 4 steps = 8 bytes per lane, vector path only.  `rowp` is then a bf16 pointer in disguise (plane_row below).
template <bool P16>
__device__ __forceinline__ const float* plane_row(const float* base, size_t elems) {
  return P16 ? reinterpret_cast<const float*>(reinterpret_cast<const bf16_t*>(base) + elems) : base + elems;
}
template <bool P16>
__device__ __forceinline__ float* plane_row(float* base, size_t elems) {
  return P16 ? reinterpret_cast<float*>(reinterpret_cast<bf16_t*>(base) + elems) : base + elems;
}
template <bool VEC, bool P16 = false>
__device__ __forceinline__ Raw4 load4_issue(const float* __restrict__ rowp, int t, int L, bool rev) {
  Raw4 r;
  if constexpr (P16) {
    static_assert(VEC, "bf16 planes: L % 4 == 0 only");
    const __amdgpu_buffer_rsrc_t rs16 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rowp), 0, L * 2, 0x00020000);
    const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs16, t < L ? (rev ? L - 4 - t : t) * 2 : OOB, 0, 0);
    static_assert(sizeof(v) == 8, "64-bit buffer load");
    __builtin_memcpy(r.v, &v, 8);
    r.v[2] = r.v[3] = 0.f;
    return r;
  }
  const __amdgpu_buffer_rsrc_t rs = row_rsrc(rowp, L);
  if (VEC) {  // L % 4 == 0 and t % 4 == 0: all-or-nothing
    // (the builtin's 128-bit result type is whatever this clang gives it: moved by memcpy, not converted - a conversion to a
    // 4 x u32 vector type compiled to a ONE-dword load splat over the four elements)
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, t < L ? (rev ? L - 4 - t : t) * 4 : OOB, 0, 0);
    static_assert(sizeof(v) == sizeof(r.v), "128-bit buffer load");
    __builtin_memcpy(r.v, &v, sizeof(r.v));
  } else {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
      r.v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, t + i < L ? (rev ? L - 1 - t - i : t + i) * 4 : OOB, 0, 0));
  }
  return r;
}
template <bool VEC, bool P16 = false>
__device__ __forceinline__ void load4_take(const Raw4& r, float (&o)[ITEMS], bool rev) {
  if constexpr (P16) {
    const unsigned w0 = __builtin_bit_cast(unsigned, r.v[0]), w1 = __builtin_bit_cast(unsigned, r.v[1]);
    const float q[ITEMS] = {__builtin_bit_cast(float, w0 << 16), __builtin_bit_cast(float, w0 & 0xffff0000u),
                            __builtin_bit_cast(float, w1 << 16), __builtin_bit_cast(float, w1 & 0xffff0000u)};
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) o[i] = rev ? q[ITEMS - 1 - i] : q[i];
    return;
  }
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) o[i] = VEC ? (rev ? r.v[ITEMS - 1 - i] : r.v[i]) : r.v[i];
}
__device__ __forceinline__ void store4_bf16(bf16_t* __restrict__ rowp, int t, int L, const float (&v)[ITEMS], bool rev);
template <bool VEC, bool P16 = false>
__device__ __forceinline__ void store4(float* __restrict__ rowp, int t, int L, const float (&v)[ITEMS], bool rev) {
  if constexpr (P16) { store4_bf16(reinterpret_cast<bf16_t*>(rowp), t, L, v, rev); return; }
  const __amdgpu_buffer_rsrc_t rs = row_rsrc(rowp, L);
  if (VEC) {
    float o[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) o[i] = rev ? v[ITEMS - 1 - i] : v[i];
    decltype(__builtin_amdgcn_raw_buffer_load_b128(rs, 0, 0, 0)) w;
    __builtin_memcpy(&w, o, sizeof(o));
    __builtin_amdgcn_raw_buffer_store_b128(w, rs, t < L ? (rev ? L - 4 - t : t) * 4 : OOB, 0, 0);
  } else {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[i]), rs, t + i < L ? (rev ? L - 1 - t - i : t + i) * 4 : OOB, 0, 0);
  }
}

// the same for a row of bf16 (the d(delta) workspace of the bf16 mode): 4 values = 8 bytes per lane; L % 4 == 0 only
__device__ __forceinline__ void store4_bf16(bf16_t* __restrict__ rowp, int t, int L, const float (&v)[ITEMS], bool rev) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(rowp, 0, L * 2, 0x00020000);
  float o[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) o[i] = rev ? v[ITEMS - 1 - i] : v[i];
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 w = {(uint32_t)f2bf(o[0]) | ((uint32_t)f2bf(o[1]) << 16), (uint32_t)f2bf(o[2]) | ((uint32_t)f2bf(o[3]) << 16)};
  __builtin_amdgcn_raw_buffer_store_b64(w, rs, t < L ? (rev ? L - 4 - t : t) * 2 : OOB, 0, 0);
}

// DPP move: lanes without a valid source (row edge / masked row) keep `old`
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL,
                                                               ROWMASK, 0xf, false));
}
// value of the previous lane (lane 0 keeps `old`): wave_shr:1
__device__ __forceinline__ float prev_lane(float old, float v) { return dpp<0x138, 0xf>(old, v); }
// value of the previous lane, zero in lane 0 (bound_ctrl: no register has to be preset for the lane without a source)
__device__ __forceinline__ float prev_lane0(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
// value of the next lane (lane 63 keeps `old`): wave_shl:1
__device__ __forceinline__ float next_lane(float old, float v) { return dpp<0x130, 0xf>(old, v); }
__device__ __forceinline__ float rdlane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
// v[lane n] = s (n < 16; the lane select is an inline constant: a second SGPR would exceed the constant bus of a gfx9 VALU op)
template <int N>
__device__ __forceinline__ void write_lane_c(float& v, float s) { asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(s), "n"(N)); }
__device__ __forceinline__ void write_lane(float& v, float s, int n) {  // n is a constant after unrolling: the switch folds
  switch (n) {
#define WL(N) case N: write_lane_c<N>(v, s); break;
    WL(0) WL(1) WL(2) WL(3) WL(4) WL(5) WL(6) WL(7) WL(8) WL(9) WL(10) WL(11) WL(12) WL(13) WL(14) WL(15)
#undef WL
  }
}

// sum over the 64 lanes, valid in lane 63 (row_shr 1,2,4,8 + row_bcast 15,31; hipcc folds these DPP moves into the adds)
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp<0x111, 0xf>(0.f, v); v += dpp<0x112, 0xf>(0.f, v); v += dpp<0x114, 0xf>(0.f, v); v += dpp<0x118, 0xf>(0.f, v);
  v += dpp<0x142, 0xa>(0.f, v); v += dpp<0x143, 0xc>(0.f, v);
  return v;
}

// ---- fused DPP scan steps.  (A, B) <- one level of the inclusive scan of the affine maps h -> A*h + B:
//      B += A * B[src lane];  A *= A[src lane];  lanes without a source keep their values (DPP write disable).
// Inside a statement every DPP read is at least two instructions behind the write of its source register (the gfx9
// VALU-write -> DPP-read hazard); the s_nop at both ends covers the compiler's instructions around the statement, which it
// cannot see into.
#define LVL(b, a, c) "v_fmac_f32_dpp %" #b ", %" #b ", %" #a " " c "\n\tv_mul_f32_dpp %" #a ", %" #a ", %" #a " " c "\n\t"
#define FULL "row_mask:0xf bank_mask:0xf"
// two independent inclusive PREFIX scans over the 64 lanes (earlier map applied first): row_shr 1,2,4,8, row_bcast 15 / 31
__device__ __forceinline__ void prefix_scan_x2(float& A0, float& B0, float& A1, float& B1) {
  asm volatile("s_nop 1\n\t"
               LVL(1, 0, "row_shr:1 " FULL) LVL(3, 2, "row_shr:1 " FULL) LVL(1, 0, "row_shr:2 " FULL) LVL(3, 2, "row_shr:2 " FULL)
               LVL(1, 0, "row_shr:4 " FULL) LVL(3, 2, "row_shr:4 " FULL) LVL(1, 0, "row_shr:8 " FULL) LVL(3, 2, "row_shr:8 " FULL)
               LVL(1, 0, "row_bcast:15 row_mask:0xa bank_mask:0xf") LVL(3, 2, "row_bcast:15 row_mask:0xa bank_mask:0xf")
               LVL(1, 0, "row_bcast:31 row_mask:0xc bank_mask:0xf") LVL(3, 2, "row_bcast:31 row_mask:0xc bank_mask:0xf")
               "s_nop 1"
               : "+v"(A0), "+v"(B0), "+v"(A1), "+v"(B1));
}
// the same when only the B parts are used afterwards (the entry state has been folded in at lane 0): the last level leaves A alone
#define LVLB(b, a, c) "v_fmac_f32_dpp %" #b ", %" #b ", %" #a " " c "\n\t"
__device__ __forceinline__ void prefix_scan_x2_b(float& A0, float& B0, float& A1, float& B1) {
  asm volatile("s_nop 1\n\t"
               LVL(1, 0, "row_shr:1 " FULL) LVL(3, 2, "row_shr:1 " FULL) LVL(1, 0, "row_shr:2 " FULL) LVL(3, 2, "row_shr:2 " FULL)
               LVL(1, 0, "row_shr:4 " FULL) LVL(3, 2, "row_shr:4 " FULL) LVL(1, 0, "row_shr:8 " FULL) LVL(3, 2, "row_shr:8 " FULL)
               LVL(1, 0, "row_bcast:15 row_mask:0xa bank_mask:0xf") LVL(3, 2, "row_bcast:15 row_mask:0xa bank_mask:0xf")
               LVLB(1, 0, "row_bcast:31 row_mask:0xc bank_mask:0xf") LVLB(3, 2, "row_bcast:31 row_mask:0xc bank_mask:0xf")
               "s_nop 1"
               : "+v"(A0), "+v"(B0), "+v"(A1), "+v"(B1));
}
#undef LVLB
// (pA, pB): inclusive PREFIX scan over the wave; (sA, sB): the four in-row levels (row_shl 1,2,4,8) of the inclusive SUFFIX scan
// (later map applied first) - its two cross-row levels have no DPP mode and are done by the caller
__device__ __forceinline__ void prefix_and_row_suffix_scan(float& pA, float& pB, float& sA, float& sB) {
  asm volatile("s_nop 1\n\t"
               LVL(1, 0, "row_shr:1 " FULL) LVL(3, 2, "row_shl:1 " FULL) LVL(1, 0, "row_shr:2 " FULL) LVL(3, 2, "row_shl:2 " FULL)
               LVL(1, 0, "row_shr:4 " FULL) LVL(3, 2, "row_shl:4 " FULL) LVL(1, 0, "row_shr:8 " FULL) LVL(3, 2, "row_shl:8 " FULL)
               LVL(1, 0, "row_bcast:15 row_mask:0xa bank_mask:0xf") "s_nop 0\n\t"
               LVL(1, 0, "row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1"
               : "+v"(pA), "+v"(pB), "+v"(sA), "+v"(sB));
}
// the backward's pair: (pA, pB) inclusive PREFIX scan over the wave, (sA, sB) the four in-row SUFFIX levels - and only the B parts are
// used afterwards (both chains have had their entry value folded in at their first lane), so the last level of each chain leaves its
// A part alone
#define LVLB(b, a, c) "v_fmac_f32_dpp %" #b ", %" #b ", %" #a " " c "\n\t"
__device__ __forceinline__ void prefix_and_row_suffix_scan_b(float& pA, float& pB, float& sA, float& sB) {
  asm volatile("s_nop 1\n\t"
               LVL(1, 0, "row_shr:1 " FULL) LVL(3, 2, "row_shl:1 " FULL) LVL(1, 0, "row_shr:2 " FULL) LVL(3, 2, "row_shl:2 " FULL)
               LVL(1, 0, "row_shr:4 " FULL) LVL(3, 2, "row_shl:4 " FULL) LVL(1, 0, "row_shr:8 " FULL) LVLB(3, 2, "row_shl:8 " FULL)
               LVL(1, 0, "row_bcast:15 row_mask:0xa bank_mask:0xf") "s_nop 0\n\t"
               LVLB(1, 0, "row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1"
               : "+v"(pA), "+v"(pB), "+v"(sA), "+v"(sB));
}
#undef LVLB
#undef LVL
#undef FULL

// ---- sums over the 64 lanes of two sets of 16 per-lane values at once: a reduce-scatter - every step halves both the number
// of values a lane carries and the group of lanes that share them - instead of 32 separate 6-level reductions:
//   lane bit 5: v_permlane32_swap of (value j, value j+8), the lower half keeps j, the upper half j+8          16 -> 8
//   lane bit 4: v_permlane16_swap of (j, j+4), even 16-lane rows keep j, odd rows j+4                          8 -> 4
//   lane bit 3 / bit 2: two DPP adds with complementary bank masks per kept value (row_shl / row_shr by 8 / 4)  4 -> 2 -> 1
//   lane bits 1, 0: butterfly over the quad.
// On return lane l holds in x[0] / y[0] the total of value (l >> 2) & 15 of its set (the four lanes of a quad agree).
__device__ __forceinline__ float f_of(unsigned v) { return __builtin_bit_cast(float, v); }
__device__ __forceinline__ unsigned u_of(float v) { return __builtin_bit_cast(unsigned, v); }
#define DA(d, s, c) "v_add_f32_dpp %" #d ", %" #s ", %" #s " " c "\n\t"
// first step for one pair of values: the lower half-wave ends up with a + (a from lane + 32), the upper with b + (b from lane - 32)
__device__ __forceinline__ float pair_sum32(float a, float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(u_of(a), u_of(b), false, false);
  return f_of(r[0]) + f_of(r[1]);
}
// the remaining steps on x[j] = pair_sum32(value j, value j + 8), j < 8 (callers do the first step as the values appear,
// so that at most 8 + 8 registers are alive)
__device__ __forceinline__ void reduce8x2(float (&x)[8], float (&y)[8]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const auto rx = __builtin_amdgcn_permlane16_swap(u_of(x[j]), u_of(x[j + 4]), false, false);
    const auto ry = __builtin_amdgcn_permlane16_swap(u_of(y[j]), u_of(y[j + 4]), false, false);
    x[j] = f_of(rx[0]) + f_of(rx[1]);
    y[j] = f_of(ry[0]) + f_of(ry[1]);
  }
  asm volatile("s_nop 1\n\t"
               DA(0, 0, "row_shl:8 row_mask:0xf bank_mask:0x3") DA(4, 4, "row_shl:8 row_mask:0xf bank_mask:0x3")
               DA(1, 1, "row_shl:8 row_mask:0xf bank_mask:0x3") DA(5, 5, "row_shl:8 row_mask:0xf bank_mask:0x3")
               DA(0, 2, "row_shr:8 row_mask:0xf bank_mask:0xc") DA(4, 6, "row_shr:8 row_mask:0xf bank_mask:0xc")
               DA(1, 3, "row_shr:8 row_mask:0xf bank_mask:0xc") DA(5, 7, "row_shr:8 row_mask:0xf bank_mask:0xc")
               DA(0, 0, "row_shl:4 row_mask:0xf bank_mask:0x5") DA(4, 4, "row_shl:4 row_mask:0xf bank_mask:0x5")
               DA(0, 1, "row_shr:4 row_mask:0xf bank_mask:0xa") DA(4, 5, "row_shr:4 row_mask:0xf bank_mask:0xa")
               "s_nop 0\n\t"
               DA(0, 0, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") DA(4, 4, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
               "s_nop 0\n\t"
               DA(0, 0, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf") DA(4, 4, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
               "s_nop 1"
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
}
#undef DA

// ---- cooperative load of `rows` time-indexed rows ([rows][L] in global memory) of the chunk starting at step t0 into LDS
// ([rows][CHUNK], step order, zero beyond L).  PER float4s per thread are requested together and only then written: the
// loads' latency is paid once per call, not once per float4.
template <int THREADS, int PER, int CH = CHUNK>
__device__ __forceinline__ void stage_rows_vec(const float* __restrict__ P, int rows, int t0, int L, float (*s)[CH], bool rev, int tid) {
  constexpr int CHUNK = CH;  // (the chunk length of the caller: 256 steps, or 512 in the forward with 8 steps per lane)
  const int total = rows * (CHUNK / 4);
  for (int base = 0; base < total; base += THREADS * PER) {
    float4 v[PER];
    int dst[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = base + j * THREADS + tid;
      const int r = min(i, total - 1) / (CHUNK / 4), tt = (i % (CHUNK / 4)) * 4;
      const bool ok = i < total && t0 + tt < L;
      v[j] = *reinterpret_cast<const float4*>(P + (ok ? (size_t)r * L + (rev ? L - 4 - t0 - tt : t0 + tt) : 0));
      if (!ok) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rev) v[j] = make_float4(v[j].w, v[j].z, v[j].y, v[j].x);
      // (512-step tiles, 8 steps per lane: the lane's two 16-byte pieces are kept 1 KB apart - piece h of lane l at (h * 64 + l) * 4 -
      // so that a wave's fragment read is one contiguous kilobyte; side by side the lanes' 32-byte stride made every ds_read_b128 2-way
      // conflicted)
      const int tp = CH == 2 * CHUNK_DEFAULT ? ((tt >> 2) & 1) * (CH / 2) + (tt >> 3) * 4 : tt;
      dst[j] = i < total ? r * CHUNK + tp : -1;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j)
      if (dst[j] >= 0) *reinterpret_cast<float4*>(&s[0][0] + dst[j]) = v[j];
  }
}
template <int THREADS>
__device__ __forceinline__ void stage_rows_scalar(const float* __restrict__ P, int rows, int t0, int L, float (*s)[CHUNK], bool rev, int tid) {
  for (int i = tid; i < rows * CHUNK; i += THREADS) {  // L % 4 != 0: odd test shapes only
    const int r = i / CHUNK, tt = i % CHUNK;
    const bool ok = t0 + tt < L;
    const float v = P[ok ? (size_t)r * L + (rev ? L - 1 - t0 - tt : t0 + tt) : 0];
    s[r][tt] = ok ? v : 0.f;
  }
}
template <int THREADS, bool VEC, int CH = CHUNK>
__device__ __forceinline__ void stage_tiles(const float* __restrict__ Bp, const float* __restrict__ Cp, const float* __restrict__ Rp,
                                            int R, int t0, int L, float (*sB)[CH], float (*sC)[CH], float (*s_dtr)[CH], bool rev) {
  constexpr int CHUNK = CH;
  // the thread index through an opaque copy: everything derived from it (source offsets, LDS addresses) is recomputed per call
  // instead of being hoisted out of the chunk loop and held - or spilled - across the row loop
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  if (VEC) {
    constexpr int PER = NS * (CHUNK / 4) / THREADS;  // 4 (256 threads) or 2 (512)
    stage_rows_vec<THREADS, PER, CH>(Bp, NS, t0, L, sB, rev, tid);
    stage_rows_vec<THREADS, PER, CH>(Cp, NS, t0, L, sC, rev, tid);
    if (Rp) stage_rows_vec<THREADS, PER, CH>(Rp, R, t0, L, s_dtr, rev, tid);
  } else if constexpr (CH == ::CHUNK_DEFAULT) {
    stage_rows_scalar<THREADS>(Bp, NS, t0, L, sB, rev, tid);
    stage_rows_scalar<THREADS>(Cp, NS, t0, L, sC, rev, tid);
    if (Rp) stage_rows_scalar<THREADS>(Rp, R, t0, L, s_dtr, rev, tid);
  }
}

// acc[i] += sum_q W[q] * s_dtr[q][4*lane + i]: the factors are walked in blocks of 8 so that the 8 tile reads (and the two
// broadcast reads of the weights) of a block are in flight together; a runtime-R loop of single reads paid one LDS round trip
// per factor (~300 cycles each with every wave of the CU doing the same)
// the same for a lane that holds 8 steps (the forward with 512-step chunks): two 16-byte reads per factor
__device__ __forceinline__ void dtproj_row8(const float* __restrict__ Wrow, const float (*s_dtr)[2 * CHUNK], int R, int lane, float (&acc)[8]) {
  int q = 0;
  for (; q + 4 <= R; q += 4) {
    const float4 w0 = *reinterpret_cast<const float4*>(Wrow + q);
    const float w[4] = {w0.x, w0.y, w0.z, w0.w};
    float4 f[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f[j][0] = *reinterpret_cast<const float4*>(&s_dtr[q + j][lane * 4]);          // (tile layout: stage_rows_vec)
      f[j][1] = *reinterpret_cast<const float4*>(&s_dtr[q + j][CHUNK + lane * 4]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[0] = fmaf(w[j], f[j][0].x, acc[0]); acc[1] = fmaf(w[j], f[j][0].y, acc[1]);
      acc[2] = fmaf(w[j], f[j][0].z, acc[2]); acc[3] = fmaf(w[j], f[j][0].w, acc[3]);
      acc[4] = fmaf(w[j], f[j][1].x, acc[4]); acc[5] = fmaf(w[j], f[j][1].y, acc[5]);
      acc[6] = fmaf(w[j], f[j][1].z, acc[6]); acc[7] = fmaf(w[j], f[j][1].w, acc[7]);
    }
  }
  for (; q < R; ++q) {
    const float w = Wrow[q];
    const float4 f0 = *reinterpret_cast<const float4*>(&s_dtr[q][lane * 4]), f1 = *reinterpret_cast<const float4*>(&s_dtr[q][CHUNK + lane * 4]);
    acc[0] = fmaf(w, f0.x, acc[0]); acc[1] = fmaf(w, f0.y, acc[1]); acc[2] = fmaf(w, f0.z, acc[2]); acc[3] = fmaf(w, f0.w, acc[3]);
    acc[4] = fmaf(w, f1.x, acc[4]); acc[5] = fmaf(w, f1.y, acc[5]); acc[6] = fmaf(w, f1.z, acc[6]); acc[7] = fmaf(w, f1.w, acc[7]);
  }
}
__device__ __forceinline__ void dtproj_row(const float* __restrict__ Wrow, const float (*s_dtr)[CHUNK], int R, int lane, float (&acc)[ITEMS]) {
  int q = 0;
  for (; q + 8 <= R; q += 8) {
    const float4 w0 = *reinterpret_cast<const float4*>(Wrow + q), w1 = *reinterpret_cast<const float4*>(Wrow + q + 4);
    const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    float4 f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = *reinterpret_cast<const float4*>(&s_dtr[q + j][lane * ITEMS]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc[0] = fmaf(w[j], f[j].x, acc[0]); acc[1] = fmaf(w[j], f[j].y, acc[1]);
      acc[2] = fmaf(w[j], f[j].z, acc[2]); acc[3] = fmaf(w[j], f[j].w, acc[3]);
    }
  }
  for (; q < R; ++q) {
    const float w = Wrow[q];
    const float4 f = *reinterpret_cast<const float4*>(&s_dtr[q][lane * ITEMS]);
    acc[0] = fmaf(w, f.x, acc[0]); acc[1] = fmaf(w, f.y, acc[1]); acc[2] = fmaf(w, f.z, acc[2]); acc[3] = fmaf(w, f.w, acc[3]);
  }
}

// ------------------------------------------------------------------------------------------------ forward
// NW waves x FWD_RPW rows each = rows of ONE (b, k) group per workgroup.  Per row and chunk the 16 states are walked in
// PAIRS: the two prefix scans of a pair are independent and interleave in one fused-DPP statement.  Row constants (A * log2 e,
// D, bias) and the carried state h live in two LDS words per lane-slot that a row reads into one VGPR each at its start.
// IT = steps per lane.  The eleven DPP instructions of a state's scan cost a wave the same whether its lanes hold 4 or 8 steps, and they
// are 40 % of the forward's issue time at 4: the product runs IT = 8 (512-step chunks, 64 KB of B/C tiles: one workgroup of NW = 16
// waves per CU instead of three of 8 - the same four waves per SIMD).  IT = 4 / NW = 8 remains for sequence lengths that are not a
// multiple of 4.
template <bool VEC, bool DTR, int IT, int NW, bool P16 = false>  // DTR: fused dt projection (delta formed in-kernel from dtr and Wdt) vs a materialised delta; P16: u and y are bf16 planes
__global__ __launch_bounds__(NW* WAVE) void selscan_fwd_kernel(const float* __restrict__ u, const float* __restrict__ delta,
                                                               const float* __restrict__ Am, const float* __restrict__ Bm,
                                                               const float* __restrict__ Cm, const float* __restrict__ Dv,
                                                               const float* __restrict__ dbias, float* __restrict__ y,
                                                               float* __restrict__ hstate, int K, int Dk, int L, int nchunk,
                                                               int xmode, const float* __restrict__ dtr,
                                                               const float* __restrict__ Wdt, int R) {
  static_assert(IT == ITEMS || (IT == 2 * ITEMS && VEC), "4 steps per lane, or 8 on the vector path");
  constexpr int CH = WAVE * IT, H = IT / 4;   // chunk length; 16-byte groups per lane
  constexpr int LPB = BLK / IT;               // lanes per checkpoint block
  const int nblk = (L + BLK - 1) / BLK;       // checkpoints per row: hstate is [rows][nblk][NS]
  extern __shared__ __attribute__((aligned(16))) float smem[];  // B tile | C tile | dt factors | Wdt rows | row constants | carried h
  float(*sB)[CH] = reinterpret_cast<float(*)[CH]>(smem);
  float(*sC)[CH] = reinterpret_cast<float(*)[CH]>(smem + NS * CH);
  float(*s_dtr)[CH] = reinterpret_cast<float(*)[CH]>(smem + 2 * NS * CH);
  float* s_W = smem + 2 * NS * CH + (size_t)R * CH;  // [NW * FWD_RPW][RMAX]
  float* s_par = s_W + NW * FWD_RPW * RMAX;          // [NW * FWD_RPW][32]: A * log2(e) | D | bias
  float* s_h = s_par + NW * FWD_RPW * 32;            // [NW * FWD_RPW][NS]
  // (wave index as a provably uniform value: row pointers then live in SGPRs and the loads take SGPR base + lane offset)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), lane = threadIdx.x % WAVE;
  const int d0 = (blockIdx.x * NW + wave) * FWD_RPW;
  const int bk = blockIdx.y, k = bk % K, b = bk / K;
  // cross-scan layout (xmode): u is [B, 2, Dk, L] (k & 1 picks the row-major / column-major copy) and directions k >= 2
  // walk every time-indexed buffer back to front
  const bool rev = xmode && k >= 2;
  const float* Bp = Bm + (size_t)bk * NS * L;
  const float* Cp = Cm + (size_t)bk * NS * L;
  const float* Rp = DTR ? dtr + (size_t)bk * R * L : nullptr;
  for (int r = 0; r < FWD_RPW; ++r) {
    const int wr = wave * FWD_RPW + r;
    const int kd = k * Dk + min(d0 + r, Dk - 1);
    if (lane < 32) s_par[wr * 32 + lane] = lane < NS ? Am[(size_t)kd * NS + lane] * LOG2E : lane == 16 ? Dv[kd] : lane == 17 ? dbias[kd] : 0.f;
    if (lane < NS) s_h[wr * NS + lane] = 0.f;
    if (DTR && lane < RMAX) s_W[wr * RMAX + lane] = lane < R ? Wdt[(size_t)kd * R + lane] : 0.f;
  }
  const int nrow = min(FWD_RPW, Dk - d0);  // rows this wave really has (<= 0: none)
  // row d's operand rows (d clamped: a row past the end is loaded like the last one and never used)
  auto urow = [&](int d) { d = min(d, Dk - 1); return plane_row<P16>(u, xmode ? (((size_t)b * 2 + (k & 1)) * Dk + d) * L : (((size_t)b * K + k) * Dk + d) * L); };
  auto drow = [&](int d) { return delta + (((size_t)b * K + k) * Dk + min(d, Dk - 1)) * L; };

#ifndef SCAN_FPRIO
#define SCAN_FPRIO 1   // the same issue-fairness trick as in the backward (below); forward: 4 waves per SIMD, two per workgroup, so the
#endif                 // parity is taken from the slot PAIR.  Measured: off 5.04 ms, slot parity 5.03, slot-pair parity 4.89 (three levels)
#if SCAN_FPRIO
  unsigned fhwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(fhwid));   // wave slot on its SIMD
#endif
  for (int c = 0; c < nchunk; ++c) {
#if SCAN_FPRIO
    if (((fhwid >> 1) + c) & 1) asm volatile("s_setprio 2"); else asm volatile("s_setprio 0");
#endif
    __syncthreads();  // previous chunk's tiles fully consumed
    const int t = c * CH + lane * IT;
    // first row's streams: requested before the staging, consumed after it
    Raw4 n_uu[H], n_dl[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      n_uu[h] = load4_issue<VEC, P16>(urow(d0), t + 4 * h, L, rev);
      n_dl[h] = n_uu[h];
      if (!DTR) n_dl[h] = load4_issue<VEC>(drow(d0), t + 4 * h, L, rev);
    }
    stage_tiles<NW * WAVE, VEC, CH>(Bp, Cp, Rp, R, c * CH, L, sB, sC, s_dtr, rev);
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < nrow; ++r) {
      const int d = d0 + r, wr = wave * FWD_RPW + r;
      const size_t row = ((size_t)b * K + k) * Dk + d;
      float uu[IT], dt[IT], dtu[IT], yy[IT];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float q4[ITEMS];
        load4_take<VEC, P16>(n_uu[h], q4, rev);
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) uu[4 * h + i] = q4[i];
        if (!DTR) load4_take<VEC>(n_dl[h], q4, rev);
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) dt[4 * h + i] = DTR ? 0.f : q4[i];
        n_uu[h] = load4_issue<VEC, P16>(urow(d + 1), t + 4 * h, L, rev);  // next row's streams, in flight behind this row's arithmetic
        if (!DTR) n_dl[h] = load4_issue<VEC>(drow(d + 1), t + 4 * h, L, rev);
      }
      const float par = s_par[wr * 32 + (lane & 31)];
      const float hc = s_h[wr * NS + (lane & (NS - 1))];  // state entering the chunk, lane n holds state n
      float nh = 0.f;
      // delta_t = <Wdt[kd, :], dtr[:, t]>: the [B, 4*d_inner, L] delta tensor of the reference is never materialised
      if constexpr (DTR && IT == ITEMS) dtproj_row(s_W + wr * RMAX, s_dtr, R, lane, dt);
      if constexpr (DTR && IT != ITEMS) dtproj_row8(s_W + wr * RMAX, s_dtr, R, lane, dt);
      const float Dd = rdlane(par, 16), bias = rdlane(par, 17);
      const float m0 = lane == 0 ? 1.f : 0.f;
      // checkpoints for the backward (BLK above): what the last lane of each BLK-step block holds after its last step, stored by
      // that lane itself, a state pair at a time (a block that starts beyond L, and every other lane, is dropped by the buffer's range
      // check)
      const __amdgpu_buffer_rsrc_t hs_rs = row_rsrc(hstate + row * (size_t)nblk * NS, nblk * NS);
      const int blk = c * (CH / BLK) + lane / LPB;
      const unsigned hs_off = (lane % LPB == LPB - 1 && blk < nblk) ? (unsigned)blk * NS * 4 : OOB;
      float sdt = 0.f;   // sum of the lane's dt: the lane's composed decay is exp2(A2 * sdt) - one exp instead of IT - 1 multiplies per state
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        dt[i] = softplus_f((t + i < L) ? dt[i] + bias : -1e30f);  // steps beyond L become the identity map (dt = 0: a = 1, b = 0)
        dtu[i] = dt[i] * uu[i];
        yy[i] = Dd * uu[i];
        sdt += dt[i];
      }
#pragma unroll
      for (int n = 0; n < NS; n += 2) {
        float a[2][IT], bb[2][IT], cc[2][IT], A[2], Bv[2], hend[2], hin[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int h = 0; h < H; ++h) {
            const float4 b4 = *reinterpret_cast<const float4*>(&sB[n + j][h * (CH / H) + lane * 4]);  // (IT = 8: piece h of a lane lies 1 KB
            const float4 c4 = *reinterpret_cast<const float4*>(&sC[n + j][h * (CH / H) + lane * 4]);  //  after piece 0, see stage_rows_vec)
            bb[j][4 * h] = b4.x; bb[j][4 * h + 1] = b4.y; bb[j][4 * h + 2] = b4.z; bb[j][4 * h + 3] = b4.w;
            cc[j][4 * h] = c4.x; cc[j][4 * h + 1] = c4.y; cc[j][4 * h + 2] = c4.z; cc[j][4 * h + 3] = c4.w;
          }
          const float A2 = rdlane(par, n + j);
#pragma unroll
          for (int i = 0; i < IT; ++i) {
            a[j][i] = __builtin_amdgcn_exp2f(dt[i] * A2);
            bb[j][i] *= dtu[i];
          }
#ifndef SCAN_FWD_SUMDT
#define SCAN_FWD_SUMDT 1   // (A/B build: 0 = the lane's decay as the product of its steps' decays)
#endif
          constexpr bool SUMDT = SCAN_FWD_SUMDT && IT > ITEMS;
          A[j] = SUMDT ? __builtin_amdgcn_exp2f(sdt * A2) : a[j][0];
          Bv[j] = bb[j][0];
#pragma unroll
          for (int i = 1; i < IT; ++i) { Bv[j] = fmaf(a[j][i], Bv[j], bb[j][i]); if (!SUMDT) A[j] *= a[j][i]; }
          hin[j] = rdlane(hc, n + j);               // state entering the chunk: joins at lane 0, from there on Bv is the state itself
          Bv[j] = fmaf(A[j] * m0, hin[j], Bv[j]);
        }
        prefix_scan_x2_b(A[0], Bv[0], A[1], Bv[1]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float hh = fmaf(m0, hin[j], prev_lane0(Bv[j]));  // state entering this lane's first step (lane 0: the chunk's)
#pragma unroll
          for (int i = 0; i < IT; ++i) {
            hh = fmaf(a[j][i], hh, bb[j][i]);
            yy[i] = fmaf(cc[j][i], hh, yy[i]);
          }
          write_lane(nh, rdlane(hh, WAVE - 1), n + j);  // state after the chunk
          hend[j] = hh;
        }
        {  // the pair's two checkpoint values are neighbours in memory: one 8-byte store
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          const u32x2 w = {__builtin_bit_cast(unsigned, hend[0]), __builtin_bit_cast(unsigned, hend[1])};
          __builtin_amdgcn_raw_buffer_store_b64(w, hs_rs, hs_off, n * 4, 0);
        }
        if constexpr (IT == ITEMS) asm volatile("" : "+v"(yy[0]), "+v"(yy[1]), "+v"(yy[2]), "+v"(yy[3]), "+v"(nh));
        else asm volatile("" : "+v"(yy[0]), "+v"(yy[1]), "+v"(yy[2]), "+v"(yy[3]), "+v"(yy[IT - 4]), "+v"(yy[IT - 3]), "+v"(yy[IT - 2]), "+v"(yy[IT - 1]), "+v"(nh));
        __builtin_amdgcn_sched_barrier(0);  // one pair's temporaries at a time
      }
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const float q4[ITEMS] = {yy[4 * h], yy[4 * h + 1], yy[4 * h + 2], yy[4 * h + 3]};
        store4<VEC, P16>(plane_row<P16>(y, row * L), t + 4 * h, L, q4, rev);
      }
      if (lane < NS) s_h[wr * NS + lane] = nh;
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
// One pass per (row, chunk) over all 16 states.  The first version walked the states in groups of 4 (register tile of dB/dC =
// 2 x 4 x ITEMS) and therefore repeated the row prologue (dt projection, softplus, u / gy loads) four times and carried the
// per-row partial du / d(delta) sums between the groups through HBM (33 GB of traffic per level-0 launch, 16.5 ms).  Here:
//   * the dB/dC tile of ALL states stays in registers (2 x 16 x ITEMS = 128 VGPRs) across the BWD_RPW rows of a wave; the loop
//     over the states is unrolled, in ONE basic block, with scheduling fences between the states so that only one state's
//     temporaries are live (256 VGPRs, two waves per SIMD).  (With a branch inside the loop LLVM sinks the accumulator updates
//     to the end of the row and spills their operands: 400 VGPRs of scratch.)
//   * sum_n gh_n B_n is accumulated once per step (S) and turned into du and the B-part of d(delta) after the states, instead of
//     four FMAs per (state, step);
//   * the prefix scan of h and the in-row part of the suffix scan of dL/dh interleave in one fused-DPP statement;
//   * the 16 per-state dA sums and the rank-R d(Wdt) sums of a row are reduced over the wave together by a reduce-scatter;
//   * the four waves fold their register tiles into the LDS tile in parallel, each wave a different quarter of the states per round.
// per-row LDS accumulators, ACC floats per row: [0, 16) dA | [16, 16 + R) d(Wdt) | [48] dD, [49] d(bias)
constexpr int ACC = 64;

// u, gy of one row for one chunk
template <bool VEC, bool P16 = false>
__device__ __forceinline__ void bwd_fetch_row(const float* __restrict__ u, const float* __restrict__ gy, int b, int K, int k, int Dk,
                                              int d, int L, int t, bool rev, int xmode, Raw4& uu, Raw4& g) {
  const size_t row = ((size_t)b * K + k) * Dk + d;
  const size_t prow = ((size_t)b * 2 + (k & 1)) * Dk + d;
  uu = load4_issue<VEC, P16>(plane_row<P16>(u, (xmode ? prow : row) * L), t, L, rev);
  g = load4_issue<VEC, P16>(plane_row<P16>(gy, ((xmode & 2) ? prow : row) * L), t, L, rev);
}
// the row's small operands in ONE register: lane n < 16: A[kd][n]; lane 16: D[kd]; lane 17: delta bias.  Read back per state with
// v_readlane (wave-uniform operands of the VALU ops).
__device__ __forceinline__ float bwd_fetch_param(const float* __restrict__ Am, const float* __restrict__ Dv, const float* __restrict__ dbias,
                                                 int kd, int lane) {
  const float* p = Am + (size_t)kd * NS + (lane & (NS - 1));
  if (lane == 16) p = Dv + kd;
  if (lane == 17) p = dbias + kd;
  return *p;
}
// the four checkpoints a (row, chunk) needs, ONE coalesced 256-byte load: lane (r, j) gets state j ENTERING the 64-step block that DPP
// row r works on (time runs against the lanes in the backward: row r holds block 3 - r of the chunk), i.e. the forward's checkpoint
// after block c * 4 + (3 - r) - 1.  Block 0 of the sequence starts from zero (selected by the reader, address clamped here), a block
// that starts beyond L is never looked at with a non-zero weight (clamped too).
__device__ __forceinline__ float bwd_fetch_ck(const float* __restrict__ hstate, size_t row, int nblk, int c, int lane) {
  const int g = c * (CHUNK / BLK) + (3 - (lane >> 4)) - 1;
  return hstate[(row * nblk + min(max(g, 0), nblk - 1)) * NS + (lane & (NS - 1))];
}
// v[lane - k] inside each 16-lane row (k is a constant after unrolling: the switch folds); lanes without a source get `old`
template <int K>
__device__ __forceinline__ float row_shr_k(float old, float v) { return dpp<0x110 + K, 0xf>(old, v); }
__device__ __forceinline__ float row_shr(float old, float v, int k) {
  switch (k) {
#define RS(K) case K: return row_shr_k<K>(old, v);
    RS(1) RS(2) RS(3) RS(4) RS(5) RS(6) RS(7) RS(8) RS(9) RS(10) RS(11) RS(12) RS(13) RS(14) RS(15)
#undef RS
  }
  return v;  // k == 0
}

// In-kernel phase timing (diagnostic build only: -DSCAN_STAMP through tools/build_scan_variant.sh; tools/scan_stamps.py reads the
// sums that wave 0 of workgroup (0, 0) leaves in the first floats of gu): s_memtime deltas accumulated per phase.
#ifdef SCAN_STAMP
#define STAMP_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#define STAMP(i) { unsigned long long st_now; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_now)::"memory"); __builtin_amdgcn_sched_barrier(0); st_acc[i] += st_now - st_last; st_last = st_now; }
#else
#define STAMP_DECL
#define STAMP(i)
#endif

// one wave's share of one fold round: states [4q, 4q + 4) of its register tile into the LDS tile (first round: plain stores).
// All eight tile reads are issued before the first add: written as read-add-write per state they run as eight LDS round trips.
#define FOLD_QUARTER(q, add)                                                                                          \
  {                                                                                                                   \
    float4 ob[4], oc[4];                                                                                              \
    _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                                                \
      ob[jj] = add ? *reinterpret_cast<const float4*>(&s_dB[4 * q + jj][tl * ITEMS]) : make_float4(0.f, 0.f, 0.f, 0.f); \
      oc[jj] = add ? *reinterpret_cast<const float4*>(&s_dC[4 * q + jj][tl * ITEMS]) : make_float4(0.f, 0.f, 0.f, 0.f); \
    }                                                                                                                 \
    _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                                                \
      ob[jj].x += accB[4 * q + jj][0]; ob[jj].y += accB[4 * q + jj][1]; ob[jj].z += accB[4 * q + jj][2]; ob[jj].w += accB[4 * q + jj][3]; \
      oc[jj].x += accC[4 * q + jj][0]; oc[jj].y += accC[4 * q + jj][1]; oc[jj].z += accC[4 * q + jj][2]; oc[jj].w += accC[4 * q + jj][3]; \
    }                                                                                                                 \
    _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                                                \
      *reinterpret_cast<float4*>(&s_dB[4 * q + jj][tl * ITEMS]) = ob[jj];                                           \
      *reinterpret_cast<float4*>(&s_dC[4 * q + jj][tl * ITEMS]) = oc[jj];                                           \
    }                                                                                                                 \
  }

// Backward: BWD_WAVES waves x BWD_RPW rows each = BWD_ROWS rows of one (b, k) group per workgroup.
// TIME RUNS AGAINST THE LANES here (lane l holds steps 4 (63 - l) .. of the chunk): the scan that has to cross the whole wave - dL/dh,
// which needs everything later in time - is then a PREFIX scan over the lanes, for which DPP has all six levels (row_shr 1..8,
// row_bcast 15 / 31), and the recomputation of h only has to cover one 16-lane row, because the forward left a checkpoint every 64
// steps: four row_shl levels, the checkpoint entering at the row's last lane.  (Until round 3 time ran with the lanes, h was a
// six-level prefix scan from the chunk's entry state and dL/dh a suffix scan whose two cross-row levels have no DPP mode: 6
// v_readlane, 4 scalar compositions and 6 selects per state, ~20 of the ~155 vector instructions per state and row-chunk.)
// SETS: 16-value sets of d(Wdt) factors per row (0: materialised delta, no dt projection; 1: rank <= 16; 2: rank <= 32)
// GD16: d(delta) is written as bf16 (dt-projection variant in bf16 mode: it is only the operand of gdtr = Wdt^T gdelta, whose result
// the caller rounds to bf16 anyway; halves the 5.9 GB per step that this workspace is written and read)
// P16: gy, u and gu are bf16 planes (bf16 mode, see load4_issue)
template <bool VEC, int SETS, bool GD16 = false, bool P16 = false>
__global__ __launch_bounds__(BWD_WAVES* WAVE, 2) void selscan_bwd_kernel(
    const float* __restrict__ gy, const float* __restrict__ u, const float* __restrict__ delta, const float* __restrict__ Am,
    const float* __restrict__ Bm, const float* __restrict__ Cm, const float* __restrict__ Dv, const float* __restrict__ dbias,
    const float* __restrict__ hstate, float* __restrict__ gu, float* __restrict__ gdelta, float* __restrict__ grow,
    float* __restrict__ wsB, float* __restrict__ wsC, int K, int Dk, int L,
    int nchunk, size_t slab_elems, int xmode, const float* __restrict__ dtr, const float* __restrict__ Wdt, int R) {
  constexpr bool DTR = SETS > 0;
  const int nblk = (L + BLK - 1) / BLK;  // hstate is [rows][nblk][NS]: the state after every 64-step block
  // one dynamic LDS array: B tile | C tile (the dB/dC fold tile aliases them) | rank-R dt factors | Wdt rows | per-row sums | carry
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float(*sB)[CHUNK] = reinterpret_cast<float(*)[CHUNK]>(smem);
  float(*sC)[CHUNK] = reinterpret_cast<float(*)[CHUNK]>(smem + NS * CHUNK);
  float(*s_dtr)[CHUNK] = reinterpret_cast<float(*)[CHUNK]>(smem + 2 * NS * CHUNK);
  float* s_W = smem + 2 * NS * CHUNK + (size_t)R * CHUNK;  // [BWD_ROWS][RMAX]
  float* s_acc = s_W + BWD_ROWS * RMAX;                    // [BWD_ROWS][ACC]    dA | d(Wdt) | dD, d(bias) sums
  float* s_carry = s_acc + BWD_ROWS * ACC;                 // [BWD_ROWS][NS]     a_t * dL/dh_t entering from the next chunk
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), lane0 = threadIdx.x % WAVE;  // wave: provably uniform
  int lane = lane0;
  const int bk = blockIdx.y, k = bk % K, b = bk / K;
  const float* Bp = Bm + (size_t)bk * NS * L;
  const float* Cp = Cm + (size_t)bk * NS * L;
  float* slabB = wsB + (size_t)blockIdx.x * slab_elems + (size_t)bk * NS * L;
  float* slabC = wsC + (size_t)blockIdx.x * slab_elems + (size_t)bk * NS * L;
  const int d0 = blockIdx.x * BWD_ROWS + wave * BWD_RPW;
  const bool rev = xmode && k >= 2;
  for (int r = 0; r < BWD_RPW; ++r) {
    const int wr = wave * BWD_RPW + r;
    const int kd = k * Dk + min(d0 + r, Dk - 1);
    if (lane < NS) s_carry[wr * NS + lane] = 0.f;
    s_acc[wr * ACC + lane] = 0.f;
    if (DTR && lane < RMAX) s_W[wr * RMAX + lane] = lane < R ? Wdt[(size_t)kd * R + lane] : 0.f;
  }
  const float* Rp = DTR ? dtr + (size_t)bk * R * L : nullptr;
  const int nrow = min(BWD_RPW, Dk - d0);  // rows this wave really has (<= 0: none)
  const int dlast = Dk - 1;

  STAMP_DECL
#ifdef SCAN_STAMP
  const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifndef SCAN_PRIO
#define SCAN_PRIO 1   // measured (profiles/r03_scan_priority.txt; three levels, backward op): off 15.89 ms, on 15.32; priority 3 instead of 2, per-row or
                      // every-second-chunk alternation were no better (15.37 / 15.52 / 15.48)
#endif
#if SCAN_PRIO
  // Issue fairness between the two waves of a SIMD (they belong to two workgroups and never meet at a barrier): the hardware arbitrates
  // oldest-first, so one wave ran at nearly full speed and the other took what was left and finished alone, slowly (round 2's stamps:
  // workgroups ending at 5.6 and at 8.2 ms at level 0).  The two alternate a raised priority chunk by chunk (wave slot parity + chunk).
  unsigned hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(hwid));   // wave slot on its SIMD
#endif
  for (int c = nchunk - 1; c >= 0; --c) {
#if SCAN_PRIO
    if ((hwid + c) & 1) asm volatile("s_setprio 2"); else asm volatile("s_setprio 0");
#endif
    __syncthreads();  // previous chunk's tiles fully consumed / flushed
    STAMP(0)  // 0: waiting at the chunk-top barrier
    // (lane through an opaque copy at every level of the loop nest: per-lane addresses and predicates are cheap to rebuild, and
    // hoisted out of the loops they do not fit next to the 128 accumulators - hipcc then spills them, and scratch reloads share
    // vmcnt with the prefetches)
    lane = lane0;
    asm volatile("" : "+v"(lane));
    int tl = WAVE - 1 - lane;  // this lane's place in time: steps 4 tl .. 4 tl + 3 of the chunk
    int t = c * CHUNK + tl * ITEMS;
    // first row's streams: requested before the staging, consumed after it (row index clamped: always issued, never under a branch)
    Raw4 n_uu, n_g, n_dl;
    float n_par, n_ck;
    {
      const int dd = min(d0, dlast);
      bwd_fetch_row<VEC, P16>(u, gy, b, K, k, Dk, dd, L, t, rev, xmode, n_uu, n_g);
      n_par = bwd_fetch_param(Am, Dv, dbias, k * Dk + dd, lane);
      n_ck = bwd_fetch_ck(hstate, ((size_t)b * K + k) * Dk + dd, nblk, c, lane);
      n_dl = n_uu;
      if (!DTR) n_dl = load4_issue<VEC>(delta + (((size_t)b * K + k) * Dk + dd) * L, t, L, rev);
    }
    stage_tiles<BWD_WAVES * WAVE, VEC>(Bp, Cp, Rp, R, c * CHUNK, L, sB, sC, s_dtr, rev);
    __syncthreads();
    STAMP(1)  // 1: staging of the B/C/dt tiles + barrier
    float accB[NS][ITEMS], accC[NS][ITEMS];  // this wave's rows' dB/dC for the chunk, summed in registers
#pragma unroll
    for (int n = 0; n < NS; ++n)
#pragma unroll
      for (int i = 0; i < ITEMS; ++i) { accB[n][i] = 0.f; accC[n][i] = 0.f; }

#pragma unroll 1
    for (int r = 0; r < nrow; ++r) {
      const int d = d0 + r, wr = wave * BWD_RPW + r;
      const size_t row = ((size_t)b * K + k) * Dk + d;
      lane = lane0;
      asm volatile("" : "+v"(lane));
      tl = WAVE - 1 - lane;
      t = c * CHUNK + tl * ITEMS;
      float uu[ITEMS], g[ITEMS], dt[ITEMS], dtu[ITEMS], S[ITEMS], ddtA[ITEMS];
      const float par = n_par;
      // states entering this lane's 64-step block, lane (r, j): state j (zero for the first block of the sequence)
      // states entering this lane's 64-step block, lane (r, j): state j (zero for the first block of the sequence)
      const float ck = (c > 0 || lane < 48) ? n_ck : 0.f;
      const float m15 = (lane & 15) == 15 ? 1.f : 0.f;  // the lane of each row that comes first in time
      const float m0 = lane == 0 ? 1.f : 0.f;           // the lane that comes last in time: where the next chunk's carry enters
      {
        float dl[ITEMS];
        load4_take<VEC, P16>(n_uu, uu, rev);
        load4_take<VEC, P16>(n_g, g, rev);
        if (!DTR) load4_take<VEC>(n_dl, dl, rev);
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) { S[i] = 0.f; ddtA[i] = 0.f; if (DTR) dl[i] = 0.f; }
        {  // next row's streams, in flight behind this row's arithmetic
          const int dn = min(d + 1, dlast);
          bwd_fetch_row<VEC, P16>(u, gy, b, K, k, Dk, dn, L, t, rev, xmode, n_uu, n_g);
          n_par = bwd_fetch_param(Am, Dv, dbias, k * Dk + dn, lane);
          n_ck = bwd_fetch_ck(hstate, ((size_t)b * K + k) * Dk + dn, nblk, c, lane);
          if (!DTR) n_dl = load4_issue<VEC>(delta + (((size_t)b * K + k) * Dk + dn) * L, t, L, rev);
        }
        if (DTR) dtproj_row(s_W + wr * RMAX, s_dtr, R, tl, dl);
        const float bias = rdlane(par, 17);
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
          dt[i] = softplus_f((t + i < L) ? dl[i] + bias : -1e30f);  // steps beyond L: dt = 0, the identity map, and no gradient
          dtu[i] = dt[i] * uu[i];
        }
      }
      const float Dd = rdlane(par, 16);
      const float par2 = par * LOG2E;  // lanes < 16: A * log2(e), the exponent scale of exp2; d(dt) below is summed in that scale
      STAMP(2)  // 2: row prologue: next row's requests, dt projection, softplus
      float dAv[8];  // dA_n + dA_{n+8} after the first reduction step (pair_sum32)
      float4 nb4 = *reinterpret_cast<const float4*>(&sB[0][tl * ITEMS]), nc4 = *reinterpret_cast<const float4*>(&sC[0][tl * ITEMS]);
      // a_t * dL/dh_t entering from the next chunk, lane n holds state n's; the new one is assembled lane by lane (v_writelane)
      const float cry = s_carry[wr * NS + (lane & (NS - 1))];
      float ncry = 0.f;
#pragma unroll
      for (int n = 0; n < NS; ++n) {
        const float4 b4 = nb4, c4 = nc4;  // requested one state ahead: at two waves per SIMD the LDS latency is not covered otherwise
        if (n + 1 < NS) {
          nb4 = *reinterpret_cast<const float4*>(&sB[n + 1][tl * ITEMS]);
          nc4 = *reinterpret_cast<const float4*>(&sC[n + 1][tl * ITEMS]);
        }
        const float bb[ITEMS] = {b4.x, b4.y, b4.z, b4.w}, cc[ITEMS] = {c4.x, c4.y, c4.z, c4.w};
        float a[ITEMS], hh[ITEMS], bu[ITEMS], cg[ITEMS];
        const float A2 = rdlane(par2, n);  // A[kd][n] * log2(e)
        // ---- h inside the chunk (same arithmetic as the forward) and the in-lane part of the dL/dh recurrence
        //      gh_i = cc_i g_i + a_{i+1} gh_{i+1}
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
          a[i] = __builtin_amdgcn_exp2f(dt[i] * A2);
          bu[i] = dtu[i] * bb[i];
          cg[i] = cc[i] * g[i];
        }
        const float P = a[1] * a[2] * a[3];
        // the next chunk's carry (= a * gh of its first step) joins at the lane that comes last in time (lane 0), as part of that
        // lane's last step; nothing else enters there, so its `alast` - a of the first step of the lane that follows in time
        // (lane - 1) - may be the zero that a lane without a DPP source gets for free
        cg[ITEMS - 1] = fmaf(m0, rdlane(cry, n), cg[ITEMS - 1]);
        const float alast = prev_lane0(a[0]);
        float SA = alast * P, SB = cg[ITEMS - 1];  // the lane's map gh -> SA gh + SB
#pragma unroll
        for (int i = ITEMS - 2; i >= 0; --i) SB = fmaf(a[i + 1], SB, cg[i]);
        // h: the lane's map h -> A h + Bv (A = a0 P); the block's entry state joins at the lane that comes first in time (lane 15 of the
        // row): from there on Bv is the state itself; four in-row suffix levels, fused with the six prefix levels of dL/dh
        float A = a[0] * P, Bv = bu[0];
#pragma unroll
        for (int i = 1; i < ITEMS; ++i) Bv = fmaf(a[i], Bv, bu[i]);
        const float h0 = row_shr(0.f, ck, 15 - n);  // lane 15 of each row: state n of the row's checkpoint (elsewhere: finite, times 0)
        Bv = fmaf(A * m15, h0, Bv);
        prefix_and_row_suffix_scan_b(SA, SB, A, Bv);
        const float hin = dpp<0x101, 0xf>(h0, Bv);  // h_{t-1} of this lane's first step: the state after lane + 1 (row_shl:1), the checkpoint at lane 15
        hh[0] = fmaf(a[0], hin, bu[0]);
#pragma unroll
        for (int i = 1; i < ITEMS; ++i) hh[i] = fmaf(a[i], hh[i - 1], bu[i]);
        float gh = prev_lane0(SB);  // gh of the step right after this lane's last one (lane 0: nothing, its alast is 0)
        float dA_n = 0.f;
#pragma unroll
        for (int i = ITEMS - 1; i >= 0; --i) {
          gh = fmaf(i == ITEMS - 1 ? alast : a[i + 1], gh, cg[i]);  // dL/dh_t
          const float da = gh * (i == 0 ? hin : hh[i - 1]) * a[i];   // dL/d(dt*A) through a = exp(dt*A)
          dA_n = fmaf(da, dt[i], dA_n);
          ddtA[i] = fmaf(da, A2, ddtA[i]);   // (x log2(e): undone once per step after the states)
          S[i] = fmaf(gh, bb[i], S[i]);
          accB[n][i] = fmaf(gh, dtu[i], accB[n][i]);
          accC[n][i] = fmaf(g[i], hh[i], accC[n][i]);
        }
        // a_t * gh_t of this chunk's first step (lane 63), for the previous chunk
        write_lane(ncry, rdlane(a[0] * gh, WAVE - 1), n);
        if (n < 8) dAv[n] = dA_n; else dAv[n - 8] = pair_sum32(dAv[n - 8], dA_n);
        // the state's updates happen HERE (the asm makes the accumulators opaque at this point), and nothing crosses the fence
        asm volatile("" : "+v"(accB[n][0]), "+v"(accB[n][1]), "+v"(accB[n][2]), "+v"(accB[n][3]), "+v"(accC[n][0]), "+v"(accC[n][1]),
                          "+v"(accC[n][2]), "+v"(accC[n][3]));
        asm volatile("" : "+v"(S[0]), "+v"(S[1]), "+v"(S[2]), "+v"(S[3]), "+v"(ddtA[0]), "+v"(ddtA[1]), "+v"(ddtA[2]), "+v"(ddtA[3]),
                          "+v"(dAv[n & 7]), "+v"(ncry));
        __builtin_amdgcn_sched_barrier(0);
      }
      STAMP(3)  // 3: the 16 states
      if (lane < NS) s_carry[wr * NS + lane] = ncry;
      // ---- after the states: du = D gy + dt S;  d(dt) = sum_n da_n A_n + u S;  d(delta) = d(dt) * sigmoid(delta + bias)
      float du[ITEMS], gd[ITEMS], dD = 0.f, dbs = 0.f;
#pragma unroll
      for (int i = 0; i < ITEMS; ++i) {
        du[i] = fmaf(dt[i], S[i], Dd * g[i]);
        const float ddt = fmaf(uu[i], S[i], ddtA[i] * LN2);
        // sigmoid(x) = 1 - exp(-softplus(x)); for x < -10 (dt = e^x < 4.6e-5) the difference cancels and sigmoid(x) = dt to 5e-5
        const float sg = dt[i] < 4.6e-5f ? dt[i] : 1.f - __builtin_amdgcn_exp2f(-dt[i] * LOG2E);
        gd[i] = ddt * sg;  // dt = 0 beyond L, hence gd = 0 there
        dbs += gd[i];
        dD = fmaf(g[i], uu[i], dD);
      }
      store4<VEC, P16>(plane_row<P16>(gu, row * L), t, L, du, rev);  // gu is always [B, K*Dk, L]; the host folds direction pairs in xmode
      if (GD16) store4_bf16(reinterpret_cast<bf16_t*>(gdelta) + row * L, t, L, gd, rev);
      else store4<VEC>(gdelta + row * L, t, L, gd, rev);
      // ---- per-row sums over the chunk's steps by reduce-scatter, two sets of 16 values at a time: the 16 dA_n travel with the
      // first 16 d(Wdt) factors (gWdt[kd, q] += sum_t gdelta_t * dtr[q, t]); at ranks > 16 the other factors with (dD, d(bias))
      float* acc = s_acc + wr * ACC;
      const int slot = (lane >> 2) & 15;
      const bool writer = (lane & 3) == 0;
      auto fac = [&](int q) -> float {  // (branch-free: a factor index past the rank reads row 0 and is replaced by zero)
        const float4 f = *reinterpret_cast<const float4*>(&s_dtr[q < R ? q : 0][tl * ITEMS]);
        const float v = fmaf(gd[0], f.x, fmaf(gd[1], f.y, fmaf(gd[2], f.z, gd[3] * f.w)));
        return q < R ? v : 0.f;
      };
      float m[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        m[j] = SETS ? pair_sum32(fac(j), fac(j + 8)) : 0.f;
        if (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);  // 8 tile reads in flight at a time (unfenced: all 16, 64 registers)
      }
      reduce8x2(dAv, m);
      if (writer) {
        const float o0 = acc[slot], o1 = acc[16 + slot];
        acc[slot] = o0 + dAv[0];
        acc[16 + slot] = o1 + m[0];
      }
      if (SETS == 2) {
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          m[j] = pair_sum32(fac(16 + j), fac(24 + j));
          e[j] = j == 0 ? pair_sum32(dD, 0.f) : j == 1 ? pair_sum32(dbs, 0.f) : 0.f;
          if (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);
        }
        reduce8x2(m, e);
        if (writer) {
          const float o0 = acc[32 + slot], o1 = acc[48 + slot];
          acc[32 + slot] = o0 + m[0];
          acc[48 + slot] = o1 + e[0];
        }
      } else {
        dD = wave_sum_dpp(dD);
        dbs = wave_sum_dpp(dbs);
        if (lane == WAVE - 1) {
          const float o0 = acc[48], o1 = acc[49];
          acc[48] = o0 + dD;
          acc[49] = o1 + dbs;
        }
      }
      STAMP(4)  // 4: row epilogue: du / d(delta) stores, per-row sums
    }
    // ---- fold the BWD_WAVES register tiles into the LDS tile (it aliases the B/C tiles, dead once every wave is past its rows):
    // four rounds, in round j wave w handles the quarter (w + j) & 3 of the states, so the waves never touch the same rows
    float(*s_dB)[CHUNK] = sB;
    float(*s_dC)[CHUNK] = sC;
    lane = lane0;
    asm volatile("" : "+v"(lane));
    tl = WAVE - 1 - lane;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < BWD_WAVES; ++j) {
      switch ((wave + j) & 3) {  // wave-uniform
        case 0: if (j == 0) { FOLD_QUARTER(0, false) } else { FOLD_QUARTER(0, true) } break;
        case 1: if (j == 0) { FOLD_QUARTER(1, false) } else { FOLD_QUARTER(1, true) } break;
        case 2: if (j == 0) { FOLD_QUARTER(2, false) } else { FOLD_QUARTER(2, true) } break;
        default: if (j == 0) { FOLD_QUARTER(3, false) } else { FOLD_QUARTER(3, true) } break;
      }
      __syncthreads();
    }
    STAMP(5)  // 5: fold of the register tiles (barriers included)
    // ---- plain, coalesced stores of this workgroup's partial dB/dC tile into its slab
    const int tid = wave * WAVE + lane;
    if (VEC) {
      for (int i = tid; i < NS * CHUNK / 4; i += BWD_WAVES * WAVE) {
        const int n = i / (CHUNK / 4), tt = (i % (CHUNK / 4)) * 4, tg = c * CHUNK + tt;
        if (tg < L) {
          float4 vb = *reinterpret_cast<const float4*>(&s_dB[n][tt]), vc = *reinterpret_cast<const float4*>(&s_dC[n][tt]);
          if (rev) { vb = make_float4(vb.w, vb.z, vb.y, vb.x); vc = make_float4(vc.w, vc.z, vc.y, vc.x); }
          const int pos = rev ? L - 4 - tg : tg;
          *reinterpret_cast<float4*>(slabB + (size_t)n * L + pos) = vb;
          *reinterpret_cast<float4*>(slabC + (size_t)n * L + pos) = vc;
        }
      }
    } else {
      for (int i = tid; i < NS * CHUNK; i += BWD_WAVES * WAVE) {
        const int n = i / CHUNK, tg = c * CHUNK + (i % CHUNK);
        const int pos = rev ? L - 1 - tg : tg;
        if (tg < L) { slabB[(size_t)n * L + pos] = s_dB[n][i % CHUNK]; slabC[(size_t)n * L + pos] = s_dC[n][i % CHUNK]; }
      }
    }
    STAMP(6)  // 6: slab stores
  }
  lane = lane0;
#ifdef SCAN_STAMP
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    for (int i = 0; i < 8; ++i) gu[i] = (float)(st_acc[i] >> 6);  // units of 64 ticks
  }
  if (threadIdx.x == 0) {  // per workgroup: start / end on the 100 MHz real-time counter, XCC id
    const unsigned long long wg_t1 = __builtin_amdgcn_s_memrealtime();
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    gu[16 + 4 * wg] = (float)(wg_t0 & 0xffffff);
    gu[17 + 4 * wg] = (float)(wg_t1 & 0xffffff);
    gu[18 + 4 * wg] = (float)(xcc & 0xf);
  }
#endif
  // the rows' sums over time - dA | d(Wdt) | dD, d(bias) - as they lie in LDS: one 256-byte row per (image, row), plain stores (every
  // (b, kd) has exactly one writer); the caller adds the images in a fixed order (round 2 added them here with float atomics, whose
  // order - and last bits - changed from run to run)
  for (int r = 0; r < BWD_RPW; ++r) {
    const int d = d0 + r, wr = wave * BWD_RPW + r;
    if (d < Dk) grow[(((size_t)b * K + k) * Dk + d) * ACC + lane] = s_acc[wr * ACC + lane];
  }
}

// out[i] = sum_s ws[s][i]
__global__ void slab_sum_kernel(const float* __restrict__ wsB, const float* __restrict__ wsC, float* __restrict__ gB,
                                float* __restrict__ gC, size_t n4, int nslab) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 sb = make_float4(0.f, 0.f, 0.f, 0.f), sc = sb;
    for (int s = 0; s < nslab; ++s) {
      const float4 b = reinterpret_cast<const float4*>(wsB)[(size_t)s * n4 + i];
      const float4 c = reinterpret_cast<const float4*>(wsC)[(size_t)s * n4 + i];
      sb.x += b.x; sb.y += b.y; sb.z += b.z; sb.w += b.w;
      sc.x += c.x; sc.y += c.y; sc.z += c.z; sc.w += c.w;
    }
    reinterpret_cast<float4*>(gB)[i] = sb;
    reinterpret_cast<float4*>(gC)[i] = sc;
  }
}

// out[i] = sum_s ws[s][i], one array
__global__ void slab_sum1_kernel(const float* __restrict__ ws, float* __restrict__ out, size_t n4, int nslab) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < nslab; ++s) {
      const float4 b = reinterpret_cast<const float4*>(ws)[(size_t)s * n4 + i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(out)[i] = a;
  }
}

// gdtr[b,k,r,l] = sum_d Wdt[k,d,r] * gdelta[b,k,d,l]: one pass over gdelta, lanes along l (16 B per lane), the Dk x R factor
// matrix streamed through LDS in 64-row tiles, R x 4 accumulators per lane.  Everything is in un-reversed position space.
// scalar variant for L % 4 != 0 (odd test shapes only)
__global__ __launch_bounds__(256) void dtproj_gdtr_scalar_kernel(const float* __restrict__ gdelta, const float* __restrict__ Wdt,
                                                                 float* __restrict__ gdtr, int K, int Dk, int R, int L) {
  const int bk = blockIdx.y, k = bk % K;
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= L) return;
  for (int r = 0; r < R; ++r) {
    float acc = 0.f;
    for (int d = 0; d < Dk; ++d) acc = fmaf(Wdt[((size_t)k * Dk + d) * R + r], gdelta[((size_t)bk * Dk + d) * L + l], acc);
    gdtr[((size_t)bk * R + r) * L + l] = acc;
  }
}

template <int RT, bool G16>
__global__ __launch_bounds__(256) void dtproj_gdtr_kernel(const float* __restrict__ gdelta, const float* __restrict__ Wdt,
                                                          float* __restrict__ gdtr, int K, int Dk, int R, int L, size_t zstride) {
  __shared__ float sW[64][RT];
  const int bk = blockIdx.y, k = bk % K;
  const int l0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const bool live = l0 < L;
  float acc[RT][4];
#pragma unroll
  for (int r = 0; r < RT; ++r) acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.f;
  const float* gp = gdelta + (size_t)bk * Dk * L;
  const bf16_t* gp16 = reinterpret_cast<const bf16_t*>(gdelta) + (size_t)bk * Dk * L;
  // gridDim.z > 1: the Dk rows are split over z (short sequences alone do not fill the chip: L = 1600 gives 2 x 64 workgroups);
  // slice z then writes its partial into slab z of `gdtr` (a workspace of gridDim.z slabs of zstride floats), summed afterwards in
  // slab order by slab_sum1_kernel (round 2: float atomics into a zeroed gdtr)
  const int dper = ((Dk + (int)gridDim.z - 1) / (int)gridDim.z + 63) / 64 * 64;
  const int dbeg = blockIdx.z * dper, dend = min(Dk, dbeg + dper);
  for (int d0 = dbeg; d0 < dend; d0 += 64) {
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * RT; i += 256) {
      const int dd = i / RT, r = i % RT;
      sW[dd][r] = (d0 + dd < Dk && r < R) ? Wdt[((size_t)k * Dk + d0 + dd) * R + r] : 0.f;
    }
    __syncthreads();
    if (live) {
      // ranks 16 / 32: eight rows' loads are issued together (row index clamped, the surplus rows' values replaced by zero).  Written as
      // load - use per row the loop keeps ONE 8-byte load in flight per thread and waits out a memory round trip per row; with R x 4
      // accumulators per thread there are too few waves per SIMD to hide that (level 1: 241 -> 175 us, level 2: 212 -> 130 us); at rank 8 the
      // plain loop already runs at the rate of its bytes and the extra registers cost waves (302 -> 332 us): it stays
      constexpr int UB = RT >= 16 ? 8 : 1;
      const int dn = min(64, dend - d0);
      for (int dd = 0; dd < dn; dd += UB) {
        float4 g[UB];
#pragma unroll
        for (int j = 0; j < UB; ++j) {
          const size_t ro = (size_t)(d0 + min(dd + j, dn - 1)) * L + l0;
          if (G16) {
            const uint2 h = *reinterpret_cast<const uint2*>(gp16 + ro);
            g[j] = make_float4(__uint_as_float(h.x << 16), __uint_as_float(h.x & 0xffff0000u), __uint_as_float(h.y << 16), __uint_as_float(h.y & 0xffff0000u));
          } else {
            g[j] = *reinterpret_cast<const float4*>(gp + ro);
          }
        }
#pragma unroll
        for (int j = 0; j < UB; ++j) {
          const bool ok = dd + j < dn;
          const float4 gj = ok ? g[j] : make_float4(0.f, 0.f, 0.f, 0.f);
          const int row = min(dd + j, 63);
#pragma unroll
          for (int r = 0; r < RT; ++r) {
            const float w = sW[row][r];
            acc[r][0] = fmaf(w, gj.x, acc[r][0]); acc[r][1] = fmaf(w, gj.y, acc[r][1]);
            acc[r][2] = fmaf(w, gj.z, acc[r][2]); acc[r][3] = fmaf(w, gj.w, acc[r][3]);
          }
        }
      }
    }
  }
  if (live) {
#pragma unroll
    for (int r = 0; r < RT; ++r)
      if (r < R) {
        float* o = gdtr + (size_t)blockIdx.z * zstride + ((size_t)bk * R + r) * L + l0;
        *reinterpret_cast<float4*>(o) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);   // (an empty slice stores zeros)
      }
  }
}

}  // namespace

extern "C" int tamtr_selective_scan_chunk(void) { return BLK; }  // the checkpoint interval: hstate is [B, KD, ceil(L / this), N]
extern "C" int tamtr_selective_scan_bwd_slabs(int Dk) { return Dk > 0 ? (Dk + BWD_ROWS - 1) / BWD_ROWS : 0; }

static int scan_fwd_launch(const float* u, const float* delta, const float* dtr, const float* Wdt, int R, const float* A,
                           const float* Bm, const float* Cm, const float* D, const float* dbias, float* y, float* hstate, int B,
                           int K, int Dk, int N, int L, int xmode, void* stream, int p16 = 0) {
  if (!u || (!delta && !dtr) || (dtr && !Wdt) || !A || !Bm || !Cm || !D || !dbias || !y || !hstate || B <= 0 || K <= 0 || Dk <= 0 ||
      L <= 0)
    return TAMTR_EINVAL;
  if (xmode != 0 && xmode != 1) return TAMTR_EINVAL;
  if (p16 && (!dtr || L % 4 || ((uintptr_t)u | (uintptr_t)y) % 8)) return TAMTR_EUNSUP;   // bf16 planes: the dt-projection form on the vector path
  if (N != NS || (long long)B * K > 65535 || (xmode && K != 4) || (dtr && (R < 1 || R > RMAX))) return TAMTR_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
  if (!dtr) R = 0;
  // 8 steps per lane (512-step chunks, 16 waves) on the vector path; 4 steps / 8 waves for L % 4 != 0 and TAMTR_SCAN_FWD4=1
  static const bool fwd4 = [] { const char* e = getenv("TAMTR_SCAN_FWD4"); return e && e[0] == '1'; }();  // read once (A/B switch)
  // ... and only where 512-step chunks do not pad the sequence much more than 256-step ones (L = 1600: 2048 against 1792 steps, measured
  // 0.87 against 0.78 ms; L = 25600 / 6400: 2.46 against 2.80 ms, 1.26 against 1.34 ms - profiles/r03_scan_forward_it8.txt)
  const long long pad8 = (L + 2 * CHUNK - 1) / (2 * CHUNK) * (2LL * CHUNK), pad4 = (L + CHUNK - 1) / CHUNK * (long long)CHUNK;
  const bool it8 = L % 4 == 0 && !fwd4 && pad8 * 16 <= pad4 * 17;
  // 16 waves per workgroup keep four waves on every SIMD beside the 64 KB of tiles - if the grid still covers the chip (256 CUs on MI355X:
  // the only target); a small grid (1280 px / 8 images, level 0: 128 such workgroups) keeps 8-wave workgroups, one per CU
#ifdef SCAN_FWD_NARROW   // (A/B build: 8-wave workgroups always)
  const bool wide = false;
#else
  const bool wide = it8 && (long long)((Dk + 2 * FWD_ROWS * FWD_RPW - 1) / (2 * FWD_ROWS * FWD_RPW)) * B * K >= 256;
#endif
  const int ch = it8 ? 2 * CHUNK : CHUNK, nw = wide ? 2 * FWD_ROWS : FWD_ROWS;
  const int nchunk = (L + ch - 1) / ch;
  dim3 grid((Dk + nw * FWD_RPW - 1) / (nw * FWD_RPW), B * K);
  const size_t dyn = ((size_t)2 * NS * ch + (size_t)R * ch + (size_t)nw * FWD_RPW * (RMAX + 32 + NS)) * sizeof(float);
#define LAUNCH_FWD(VEC, DTR, IT, NW) LAUNCH_FWD_(VEC, DTR, IT, NW, false)
#define LAUNCH_FWD_(VEC, DTR, IT, NW, P16)                                                                                             \
  {                                                                                                                                    \
    auto kern = selscan_fwd_kernel<VEC, DTR, IT, NW, P16>;                                                                             \
    if (dyn > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess)  \
      return TAMTR_ELAUNCH; /* (per call: the attribute belongs to the current device) */                                              \
    hipLaunchKernelGGL(kern, grid, dim3(NW * WAVE), dyn, s, u, delta, A, Bm, Cm, D, dbias, y, hstate, K, Dk, L, nchunk, xmode, dtr, Wdt, R); \
  }
  if (p16) {
    if (it8 && wide) LAUNCH_FWD_(true, true, 2 * ITEMS, 2 * FWD_ROWS, true)
    else if (it8) LAUNCH_FWD_(true, true, 2 * ITEMS, FWD_ROWS, true)
    else LAUNCH_FWD_(true, true, ITEMS, FWD_ROWS, true)
    return tamtr_launch_status();
  }
  if (it8 && wide) { if (dtr) LAUNCH_FWD(true, true, 2 * ITEMS, 2 * FWD_ROWS) else LAUNCH_FWD(true, false, 2 * ITEMS, 2 * FWD_ROWS) }
  if (it8 && !wide) { if (dtr) LAUNCH_FWD(true, true, 2 * ITEMS, FWD_ROWS) else LAUNCH_FWD(true, false, 2 * ITEMS, FWD_ROWS) }
  if (!it8) {
    if (L % 4 == 0) { if (dtr) LAUNCH_FWD(true, true, ITEMS, FWD_ROWS) else LAUNCH_FWD(true, false, ITEMS, FWD_ROWS) }
    else { if (dtr) LAUNCH_FWD(false, true, ITEMS, FWD_ROWS) else LAUNCH_FWD(false, false, ITEMS, FWD_ROWS) }
  }
#undef LAUNCH_FWD
#undef LAUNCH_FWD_
  return tamtr_launch_status();
}

extern "C" int tamtr_selective_scan_fwd(const float* u, const float* delta, const float* A, const float* Bm, const float* Cm,
                                        const float* D, const float* dbias, float* y, float* hstate, int B, int K, int Dk, int N,
                                        int L, int xmode, void* stream) {
  return scan_fwd_launch(u, delta, nullptr, nullptr, 0, A, Bm, Cm, D, dbias, y, hstate, B, K, Dk, N, L, xmode, stream);
}

extern "C" int tamtr_selective_scan_dtproj_fwd(const void* u, const float* dtr, const float* Wdt, const float* A, const float* Bm,
                                               const float* Cm, const float* D, const float* dbias, void* y, float* hstate, int B,
                                               int K, int Dk, int N, int R, int L, int xmode, int planes_bf16, void* stream) {
  if (!dtr || (planes_bf16 != 0 && planes_bf16 != 1)) return TAMTR_EINVAL;
  return scan_fwd_launch((const float*)u, nullptr, dtr, Wdt, R, A, Bm, Cm, D, dbias, (float*)y, hstate, B, K, Dk, N, L, xmode, stream, planes_bf16);
}

static int scan_bwd_launch(const float* gy, const float* u, const float* delta, const float* dtr, const float* Wdt, int R,
                           const float* A, const float* Bm, const float* Cm, const float* D, const float* dbias, const float* hstate,
                           float* gu, float* gdelta, float* gdtr, float* grow, float* gB, float* gC, float* ws, int B, int K, int Dk,
                           int N, int L, int xmode, int gd16, void* stream);

extern "C" int tamtr_selective_scan_row_sums(void) { return ACC; }

extern "C" int tamtr_selective_scan_bwd(const float* gy, const float* u, const float* delta, const float* A, const float* Bm,
                                        const float* Cm, const float* D, const float* dbias, const float* hstate, float* gu,
                                        float* gdelta, float* grow, float* gB, float* gC, float* ws, int B, int K, int Dk, int N, int L,
                                        int xmode, void* stream) {
  if (!delta) return TAMTR_EINVAL;
  return scan_bwd_launch(gy, u, delta, nullptr, nullptr, 0, A, Bm, Cm, D, dbias, hstate, gu, gdelta, nullptr, grow, gB, gC, ws, B, K, Dk,
                         N, L, xmode, 0, stream);
}

extern "C" int tamtr_selective_scan_dtproj_bwd(const void* gy, const void* u, const float* dtr, const float* Wdt, const float* A,
                                               const float* Bm, const float* Cm, const float* D, const float* dbias,
                                               const float* hstate, void* gu, void* gdelta_ws, float* gdtr, float* grow, float* gB,
                                               float* gC, float* ws, int B, int K, int Dk, int N, int R, int L, int xmode, int bf16_flags,
                                               void* stream) {
  // bf16_flags: bit 0 - the d(delta) workspace is bf16; bit 1 - gy, u and gu are bf16 planes (needs bit 0)
  if (!dtr || !Wdt || !gdtr || bf16_flags < 0 || bf16_flags > 3 || bf16_flags == 2) return TAMTR_EINVAL;
  if (bf16_flags && L % 4) return TAMTR_EUNSUP;
  if ((bf16_flags & 2) && ((uintptr_t)gy | (uintptr_t)u | (uintptr_t)gu) % 8) return TAMTR_EUNSUP;
  return scan_bwd_launch((const float*)gy, (const float*)u, nullptr, dtr, Wdt, R, A, Bm, Cm, D, dbias, hstate, (float*)gu, (float*)gdelta_ws, gdtr,
                         grow, gB, gC, ws, B, K, Dk, N, L, xmode, bf16_flags, stream);
}

static int scan_bwd_launch(const float* gy, const float* u, const float* delta, const float* dtr, const float* Wdt, int R,
                           const float* A, const float* Bm, const float* Cm, const float* D, const float* dbias, const float* hstate,
                           float* gu, float* gdelta, float* gdtr, float* grow, float* gB, float* gC, float* ws, int B, int K, int Dk,
                           int N, int L, int xmode, int gd16, void* stream) {
  if (!gy || !u || (!delta && !dtr) || !A || !Bm || !Cm || !D || !dbias || !hstate || !gu || !gdelta || !grow || !gB || !gC || !ws ||
      B <= 0 || K <= 0 || Dk <= 0 || L <= 0)
    return TAMTR_EINVAL;
  if (xmode != 0 && xmode != 1 && xmode != 3) return TAMTR_EINVAL;
  if (N != NS || (long long)B * K > 65535 || (xmode && K != 4) || (dtr && (R < 1 || R > RMAX))) return TAMTR_EUNSUP;
  const int nchunk = (L + CHUNK - 1) / CHUNK;
  const int nslab = (Dk + BWD_ROWS - 1) / BWD_ROWS;
  const size_t slab = (size_t)B * K * NS * L;
  float* wsB = ws;
  float* wsC = ws + (size_t)nslab * slab;
  dim3 grid(nslab, B * K);
  hipStream_t s = (hipStream_t)stream;
  if (!dtr) R = 0;
  // B/C tiles | rank-R dt factors | Wdt rows | per-row sums | carry: 78 KB at rank 32, two workgroups per CU
  const size_t dyn = ((size_t)2 * NS * CHUNK + (size_t)R * CHUNK + (size_t)BWD_ROWS * (RMAX + ACC + NS)) * sizeof(float);
#define LAUNCH_BWD(VEC, SETS)                                                                                                          \
  hipLaunchKernelGGL((selscan_bwd_kernel<VEC, SETS, false>), grid, dim3(BWD_WAVES * WAVE), dyn, s, gy, u, delta, A, Bm, Cm, D, dbias, hstate, gu, \
                     gdelta, grow, wsB, wsC, K, Dk, L, nchunk, slab, xmode, dtr, Wdt, R)
#define LAUNCH_BWD16(SETS)                                                                                                             \
  hipLaunchKernelGGL((selscan_bwd_kernel<true, SETS, true>), grid, dim3(BWD_WAVES * WAVE), dyn, s, gy, u, delta, A, Bm, Cm, D, dbias, hstate, gu, \
                     gdelta, grow, wsB, wsC, K, Dk, L, nchunk, slab, xmode, dtr, Wdt, R)
#define LAUNCH_BWD16P(SETS)                                                                                                            \
  hipLaunchKernelGGL((selscan_bwd_kernel<true, SETS, true, true>), grid, dim3(BWD_WAVES * WAVE), dyn, s, gy, u, delta, A, Bm, Cm, D, dbias, hstate, \
                     gu, gdelta, grow, wsB, wsC, K, Dk, L, nchunk, slab, xmode, dtr, Wdt, R)
  if ((gd16 & 2) && dtr) {
    if (R <= 16) LAUNCH_BWD16P(1); else LAUNCH_BWD16P(2);
    gd16 = 1;
  } else if (gd16 && dtr) {
    if (R <= 16) LAUNCH_BWD16(1); else LAUNCH_BWD16(2);
  } else if (L % 4 == 0) {
    if (R == 0) LAUNCH_BWD(true, 0); else if (R <= 16) LAUNCH_BWD(true, 1); else LAUNCH_BWD(true, 2);
  } else {
    if (R == 0) LAUNCH_BWD(false, 0); else if (R <= 16) LAUNCH_BWD(false, 1); else LAUNCH_BWD(false, 2);
  }
#undef LAUNCH_BWD
#undef LAUNCH_BWD16
#undef LAUNCH_BWD16P
  // dB / dC: the workgroups' slabs added in slab order (this also frees the workspace for the split gdtr product below)
  const size_t n4 = slab / 4;  // N = 16 makes slab a multiple of 4
  const unsigned blocks = (unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(slab_sum_kernel, dim3(blocks), dim3(256), 0, s, wsB, wsC, gB, gC, n4, nslab);
  if (dtr && L % 4) {
    hipLaunchKernelGGL(dtproj_gdtr_scalar_kernel, dim3((L + 255) / 256, B * K), dim3(256), 0, s, gdelta, Wdt, gdtr, K, Dk, R, L);
  } else if (dtr) {  // gdtr = Wdt^T gdelta (position space, un-reversed)
    dim3 g2((L / 4 + 255) / 256, B * K);
    const int wgs = (int)g2.x * B * K;
    const size_t gsz = (size_t)B * K * R * L;
    float* out = gdtr;
    if (wgs < 768 && Dk >= 128) {  // not enough workgroups for 256 CUs: split the rows over z, one partial slab per slice in the
      // (now free) dB/dC workspace - z * R <= (Dk / 64) * 32 <= 2 * nslab * 16 floats per position, so it fits
      g2.z = (unsigned)min(Dk / 64, (1024 + wgs - 1) / wgs);
      if (g2.z > 1) out = ws;
    }
#define GDTR(RT)                                                                                                       \
  {                                                                                                                    \
    if (gd16) hipLaunchKernelGGL((dtproj_gdtr_kernel<RT, true>), g2, dim3(256), 0, s, gdelta, Wdt, out, K, Dk, R, L, gsz); \
    else hipLaunchKernelGGL((dtproj_gdtr_kernel<RT, false>), g2, dim3(256), 0, s, gdelta, Wdt, out, K, Dk, R, L, gsz);   \
  }
    if (R <= 8) GDTR(8) else if (R <= 16) GDTR(16) else GDTR(32)
#undef GDTR
    if (out != gdtr) {
      const size_t m4 = gsz / 4;  // L % 4 == 0 here
      const unsigned bl = (unsigned)((m4 + 255) / 256 < 4096 ? (m4 + 255) / 256 : 4096);
      hipLaunchKernelGGL(slab_sum1_kernel, dim3(bl), dim3(256), 0, s, ws, gdtr, m4, (int)g2.z);
    }
  }
  return tamtr_launch_status();
}
