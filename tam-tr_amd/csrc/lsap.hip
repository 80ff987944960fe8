// lsap.hip - device-side Hungarian assignment for the RT-DETR matcher.
//
// Replaces `C.cpu()` + scipy.optimize.linear_sum_assignment per image in HungarianMatcher.forward
// (ultralytics/models/utils/ops.py:98-119): the four device->host syncs per training step disappear and the host thread
// can run ahead of the GPU across steps.
//
// Algorithm: the shortest-augmenting-path solver scipy ships (Crouse, "On implementing 2D rectangular assignment
// algorithms", 2016; scipy 1.15.3 is the version in this image) restated for one wavefront per image: float64 duals
// and path costs exactly as scipy (it promotes the float32 cost matrix to float64), the same transposition rule
// (solve with rows = the smaller side), the same scan order over the `remaining` column list and the same tie rule
// (strictly smaller wins; an equal cost wins only if its column is still unassigned), so the assignment is identical to
// scipy's even on tied costs (non-finite costs are zeroed by the caller, ops.py:112, which makes exact ties real).
// The scan over the remaining columns is what the 64 lanes share: each lane walks its strided slice in scan order and
// the lane winners are merged with a comparator that encodes the sequential rule (among equal minima: the LAST
// unassigned column in scan order, else the FIRST column).
//
// Work is tiny and latency-bound (nq = 100 queries x <= a few hundred boxes): one launch for the whole batch, LDS holds
// duals / paths / the image's cost block when it fits.
#include "common.h"

#define LSAP_MAX_IMAGES 256

struct LsapGroups {
  int off[LSAP_MAX_IMAGES + 1];  // prefix sums of the per-image box counts (kernel argument, no device buffer needed)
};

struct Cand {
  double val;
  int it;   // position in `remaining`
  int un;   // 1 = column not assigned yet
};

__device__ __forceinline__ bool cand_wins(const Cand& a, const Cand& b) {
  if (a.val != b.val) return a.val < b.val;
  if (a.un != b.un) return a.un > b.un;
  return a.un ? a.it > b.it : a.it < b.it;
}

__global__ __launch_bounds__(WAVE) void lsap_kernel(const float* __restrict__ cost, LsapGroups grp, int nq, int G,
                                                    int stage_cost, int64_t* __restrict__ bi, int64_t* __restrict__ si,
                                                    int64_t* __restrict__ gi) {
  extern __shared__ __align__(16) unsigned char lds[];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int goff = grp.off[b], n = grp.off[b + 1] - goff;
  if (n <= 0) return;
  int out0 = 0;
  for (int k = 0; k < b; ++k) out0 += min(nq, grp.off[k + 1] - grp.off[k]);
  const bool tr = nq > n;  // scipy transposes when there are more rows than columns
  const int R = tr ? n : nq, Cn = tr ? nq : n;
  const float* cb = cost + (size_t)b * nq * G + goff;  // element (query q, box t) at cb[q * G + t]

  double* u = reinterpret_cast<double*>(lds);
  double* v = u + R;
  double* sp = v + Cn;
  int* path = reinterpret_cast<int*>(sp + Cn);
  int* row4col = path + Cn;
  int* remaining = row4col + Cn;
  int* col4row = remaining + Cn;
  unsigned char* SR = reinterpret_cast<unsigned char*>(col4row + R);
  unsigned char* SC = SR + R;
  float* sC = reinterpret_cast<float*>(lds + ((8 * (R + 2 * Cn) + 4 * (3 * Cn + R) + R + Cn + 15) & ~15));

  for (int i = lane; i < R; i += WAVE) { u[i] = 0.0; col4row[i] = -1; }
  for (int j = lane; j < Cn; j += WAVE) { v[j] = 0.0; row4col[j] = -1; path[j] = -1; }
  if (stage_cost)
    for (int e = lane; e < R * Cn; e += WAVE) {
      int i = e / Cn, j = e - i * Cn;
      sC[e] = tr ? cb[(size_t)j * G + i] : cb[(size_t)i * G + j];
    }
  __syncthreads();

  const double INF = __longlong_as_double(0x7ff0000000000000ll);
  bool failed = false;
  for (int cur = 0; cur < R && !failed; ++cur) {
    // ---- shortest augmenting path from row `cur`
    for (int i = lane; i < R; i += WAVE) SR[i] = 0;
    for (int j = lane; j < Cn; j += WAVE) { SC[j] = 0; sp[j] = INF; remaining[j] = Cn - j - 1; }
    __syncthreads();
    int num_remaining = Cn, i = cur, sink = -1;
    double minVal = 0.0;
    while (sink == -1 && num_remaining > 0) {
      if (lane == 0) SR[i] = 1;
      const double ui = u[i];
      Cand best{INF, 0x7fffffff, 0};
      for (int it = lane; it < num_remaining; it += WAVE) {
        const int j = remaining[it];
        const float c = stage_cost ? sC[i * Cn + j] : (tr ? cb[(size_t)j * G + i] : cb[(size_t)i * G + j]);
        const double r = ((minVal + (double)c) - ui) - v[j];
        double s = sp[j];
        if (r < s) { path[j] = i; sp[j] = r; s = r; }
        Cand cnd{s, it, row4col[j] == -1 ? 1 : 0};
        if (cand_wins(cnd, best)) best = cnd;
      }
#pragma unroll
      for (int o = WAVE / 2; o > 0; o >>= 1) {
        Cand other{__shfl_xor(best.val, o, WAVE), __shfl_xor(best.it, o, WAVE), __shfl_xor(best.un, o, WAVE)};
        if (cand_wins(other, best)) best = other;
      }
      if (!(best.val < INF) || best.it >= num_remaining) { failed = true; break; }  // infeasible / NaN costs
      minVal = best.val;
      const int j = remaining[best.it];
      const int r4c = row4col[j];
      if (r4c == -1) sink = j; else i = r4c;
      __syncthreads();  // every lane has read remaining[] before it is edited
      --num_remaining;
      if (lane == 0) { SC[j] = 1; remaining[best.it] = remaining[num_remaining]; }
      __syncthreads();
    }
    if (failed || sink == -1) { failed = true; break; }
    // ---- dual update (reads sp/col4row, writes u/v: disjoint arrays)
    for (int r = lane; r < R; r += WAVE) {
      if (r == cur) u[r] += minVal;
      else if (SR[r]) u[r] += minVal - sp[col4row[r]];
    }
    for (int j = lane; j < Cn; j += WAVE)
      if (SC[j]) v[j] -= minVal - sp[j];
    __syncthreads();
    // ---- augment along the path
    if (lane == 0) {
      int j = sink;
      for (int guard = 0; guard <= R; ++guard) {
        const int r = path[j];
        row4col[j] = r;
        const int t = col4row[r];
        col4row[r] = j;
        j = t;
        if (r == cur) break;
      }
    }
    __syncthreads();
  }

  // ---- emit (query, global box) pairs sorted by query index, as scipy returns them
  const int m = min(nq, n);
  if (failed) {
    for (int k = lane; k < m; k += WAVE) { bi[out0 + k] = b; si[out0 + k] = -1; gi[out0 + k] = -1; }
    return;
  }
  if (tr) {
    for (int g = lane; g < n; g += WAVE) {
      const int q = col4row[g];
      int rank = 0;
      for (int h = 0; h < n; ++h) rank += col4row[h] < q;
      bi[out0 + rank] = b; si[out0 + rank] = q; gi[out0 + rank] = goff + g;
    }
  } else {
    for (int q = lane; q < nq; q += WAVE) { bi[out0 + q] = b; si[out0 + q] = q; gi[out0 + q] = goff + col4row[q]; }
  }
}

extern "C" int tamtr_lsap_assign(const float* cost, const int32_t* group_sizes_host, int bs, int nq, int G,
                                 int64_t* batch_idx, int64_t* query_idx, int64_t* gt_idx, void* stream) {
  if (!cost || !group_sizes_host || !batch_idx || !query_idx || !gt_idx || bs <= 0 || nq <= 0 || G < 0)
    return TAMTR_EINVAL;
  if (bs > LSAP_MAX_IMAGES) return TAMTR_EUNSUP;
  LsapGroups grp;
  grp.off[0] = 0;
  size_t lds = 0, lds_staged = 0;
  for (int b = 0; b < bs; ++b) {
    const int n = group_sizes_host[b];
    if (n < 0) return TAMTR_EINVAL;
    grp.off[b + 1] = grp.off[b] + n;
    const size_t R = n < nq ? n : nq, Cn = n < nq ? nq : n;
    const size_t base = (8 * (R + 2 * Cn) + 4 * (3 * Cn + R) + R + Cn + 15) & ~(size_t)15;
    if (base > lds) lds = base;
    if (base + 4 * R * Cn > lds_staged) lds_staged = base + 4 * R * Cn;
  }
  for (int b = bs; b < LSAP_MAX_IMAGES; ++b) grp.off[b + 1] = grp.off[bs];
  if (grp.off[bs] != G) return TAMTR_EINVAL;
  if (G == 0) return TAMTR_OK;
  const size_t LIMIT = 64 * 1024;
  if (lds > LIMIT) return TAMTR_EUNSUP;
  const int stage = lds_staged <= LIMIT;
  hipLaunchKernelGGL(lsap_kernel, dim3(bs), dim3(WAVE), stage ? lds_staged : lds, (hipStream_t)stream, cost, grp, nq, G,
                     stage, batch_idx, query_idx, gt_idx);
  return tamtr_launch_status();
}
