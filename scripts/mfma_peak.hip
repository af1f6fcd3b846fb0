// Micro-benchmark: sustained rate of v_mfma_f32_32x32x2_f32 for 1 and 2 waves per SIMD (tuning aid).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, const char* name) {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double mfmas = (double)blocks * 4 * iters * 16 * NACC;
  double tf = mfmas * 4096.0 / (ms * 1e-3) / 1e12;
  printf("%s: blocks %d NACC %d: %.3f ms, %.1f TF, %.1f cycles/MFMA/SIMD @2.4GHz\n", name, blocks, NACC, ms, tf,
         ms * 1e-3 * 2.4e9 / (mfmas / 1024.0));
}
int main() {
  run<4>(256, "1 wave/SIMD");
  run<4>(512, "2 waves/SIMD");
  run<2>(512, "2 waves/SIMD");
  run<1>(512, "2 waves/SIMD");
  run<4>(1024, "4 waves/SIMD");
  return 0;
}
