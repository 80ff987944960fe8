// VALU issue-rate probe for gfx950: wave64 v_fma_f32 vs v_pk_fma_f32, 1 / 2 / 4 waves per SIMD, independent accumulators.
// hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters, float s) {
  float a[16];
  f2 p[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
  const f2 s2 = {s, s * 0.5f};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(s2));
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
    }
  }
  float r = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += a[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE>
double run(int waves_per_simd, int iters, float* d) {
  const int threads = 64 * 4 * waves_per_simd;  // one block per CU, 256 CUs
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 10, 1.0001f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters, 1.0001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  float* d;
  hipMalloc(&d, 256 * 1024 * sizeof(float));
  const int iters = 20000;
  const char* names[4] = {"v_fma_f32 x16", "v_pk_fma_f32 x8 (16 fma)", "v_mul_f32_dpp x16", "v_exp_f32 x16"};
  for (int mode = 0; mode < 4; ++mode)
    for (int w : {1, 2, 4}) {
      double ms = mode == 0 ? run<0>(w, iters, d) : mode == 1 ? run<1>(w, iters, d) : mode == 2 ? run<2>(w, iters, d) : run<3>(w, iters, d);
      const double ninst = (double)iters * (mode == 1 ? 8 : 16);  // per wave
      // cycles per instruction per wave assuming 2.4 GHz; and per SIMD (divide by waves)
      printf("%-28s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instruction, %.2f ns per SIMD-instruction\n", names[mode], w, ms, ms * 1e6 / ninst,
             ms * 1e6 / ninst / w);
    }
  return 0;
}
