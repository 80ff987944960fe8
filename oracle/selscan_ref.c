/* selscan_ref.c - CPU twin of the selective-scan oracle (oracle/tamtr_oracle.py: selective_scan) - TEST INFRASTRUCTURE.
 *
 * Plain sequential S6 recurrence per the call contract at ultralytics/nn/extra_modules/VManba/vmamba.py:962-990 /
 * csms6s.py:252-270 (the reference's own scan is an external CUDA extension that is not in its tree: PARITY UNPINNED,
 * this file is cross-checked against the pure-torch loop in tamtr_oracle.py only):
 *     dt = softplus(delta + bias);  h_t = exp(dt_t A_n) h_{t-1} + dt_t B_{n,t} u_t;  y_t = sum_n C_{n,t} h_{n,t} + D u_t
 * One row (b, kd) at a time, OpenMP over rows; the backward stores the row's states [L][N] and walks back.
 * Used only as the checker at sizes the torch loop cannot reach and by bench.py's cpu_baseline leg.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }

void selscan_ref_fwd(const float* u, const float* delta, const float* A, const float* Bm, const float* Cm, const float* D,
                     const float* bias, float* y, int Bn, int K, int Dk, int N, int L) {
  const long rows = (long)Bn * K * Dk;
#pragma omp parallel for schedule(static)
  for (long r = 0; r < rows; ++r) {
    const int kd = (int)(r % ((long)K * Dk)), b = (int)(r / ((long)K * Dk)), k = kd / Dk;
    const float* ur = u + r * L;
    const float* dr = delta + r * L;
    const float* Bp = Bm + ((long)b * K + k) * N * L;
    const float* Cp = Cm + ((long)b * K + k) * N * L;
    float h[64];
    for (int n = 0; n < N; ++n) h[n] = 0.f;
    for (int t = 0; t < L; ++t) {
      const float dt = softplusf(dr[t] + bias[kd]);
      float acc = D[kd] * ur[t];
      for (int n = 0; n < N; ++n) {
        h[n] = expf(dt * A[(long)kd * N + n]) * h[n] + dt * Bp[(long)n * L + t] * ur[t];
        acc += Cp[(long)n * L + t] * h[n];
      }
      y[r * L + t] = acc;
    }
  }
}

/* gB, gC, gA, gD, gbias must be zeroed by the caller; they are accumulated under omp critical sections per row. */
void selscan_ref_bwd(const float* gy, const float* u, const float* delta, const float* A, const float* Bm, const float* Cm,
                     const float* D, const float* bias, float* gu, float* gdelta, float* gA, float* gB, float* gC, float* gD,
                     float* gbias, int Bn, int K, int Dk, int N, int L) {
  const long rows = (long)Bn * K * Dk;
#pragma omp parallel
  {
    float* hs = (float*)malloc(sizeof(float) * (size_t)(L + 1) * N);   /* hs[t+1][n] = h_t, hs[0] = 0 */
    float* lB = (float*)malloc(sizeof(float) * (size_t)L * N);
    float* lC = (float*)malloc(sizeof(float) * (size_t)L * N);
#pragma omp for schedule(static)
    for (long r = 0; r < rows; ++r) {
      const int kd = (int)(r % ((long)K * Dk)), b = (int)(r / ((long)K * Dk)), k = kd / Dk;
      const float* ur = u + r * L;
      const float* dr = delta + r * L;
      const float* gr = gy + r * L;
      const float* Bp = Bm + ((long)b * K + k) * N * L;
      const float* Cp = Cm + ((long)b * K + k) * N * L;
      const float* Ar = A + (long)kd * N;
      for (int n = 0; n < N; ++n) hs[n] = 0.f;
      for (int t = 0; t < L; ++t) {
        const float dt = softplusf(dr[t] + bias[kd]);
        for (int n = 0; n < N; ++n)
          hs[(long)(t + 1) * N + n] = expf(dt * Ar[n]) * hs[(long)t * N + n] + dt * Bp[(long)n * L + t] * ur[t];
      }
      float gh[64], lA[64];
      for (int n = 0; n < N; ++n) { gh[n] = 0.f; lA[n] = 0.f; }
      float lD = 0.f, lb = 0.f;
      for (int t = L - 1; t >= 0; --t) {
        const float x = dr[t] + bias[kd];
        const float dt = softplusf(x);
        float ddt = 0.f, du = D[kd] * gr[t];
        lD += gr[t] * ur[t];
        for (int n = 0; n < N; ++n) {
          const float a = expf(dt * Ar[n]);
          const float g = Cp[(long)n * L + t] * gr[t] + gh[n];   /* dL/dh_t (gh carries a_{t+1} * g_{t+1}) */
          lC[(long)t * N + n] = gr[t] * hs[(long)(t + 1) * N + n];
          const float da = g * hs[(long)t * N + n] * a;
          lA[n] += da * dt;
          ddt += da * Ar[n] + g * ur[t] * Bp[(long)n * L + t];
          du += g * dt * Bp[(long)n * L + t];
          lB[(long)t * N + n] = g * dt * ur[t];
          gh[n] = a * g;
        }
        const float gd = ddt / (1.f + expf(-x));
        gu[r * L + t] = du;
        gdelta[r * L + t] = gd;
        lb += gd;
      }
#pragma omp critical
      {
        for (int n = 0; n < N; ++n) gA[(long)kd * N + n] += lA[n];
        gD[kd] += lD;
        gbias[kd] += lb;
        float* gBp = gB + ((long)b * K + k) * N * L;
        float* gCp = gC + ((long)b * K + k) * N * L;
        for (int t = 0; t < L; ++t)
          for (int n = 0; n < N; ++n) { gBp[(long)n * L + t] += lB[(long)t * N + n]; gCp[(long)n * L + t] += lC[(long)t * N + n]; }
      }
    }
    free(hs); free(lB); free(lC);
  }
}
