// Stand-alone probe of an fp32 MFMA GEMM core for gfx950 (not part of the product): how busy can the matrix pipe be
// kept when a wave owns a 64 x 64 output tile (four independent accumulator blocks), reads the NEXT quarter's fragments
// under the current quarter's MFMAs, and the workgroup synchronises once per KSTEP (= 1 or 2 LDS-DMA'd K tiles of 32)?
//   C[M][N] = A[M][K] * B[N][K]^T, fp32, v_mfma_f32_32x32x2_f32, operands by global_load_lds_dwordx4 into an
//   XOR-swizzled LDS ring (same layout as wino_gemm_kernel), 256 threads = 2 x 2 waves, workgroup tile 128 x 128.
// Build / run (on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/gemm_core_probe.hip -o /tmp/gemm_probe && /tmp/gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_t;

__device__ __forceinline__ void glds16(const float* src, lds_t dst) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
#endif
}

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

// NST stages of (BM + BN) rows x 32 floats; one barrier per K tile.
template <int NST, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_probe_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                               float* __restrict__ C, int M, int N, int K, int tiles_n,
                                                               int ntiles) {
  constexpr int BM = 128, BN = 128;
  constexpr int STAGE = (BM + BN) * 32;  // floats
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm0 = (wid >> 1) * 64, wn0 = (wid & 1) * 64;
  const int l31 = lane & 31, lh = lane >> 5;
  const int KT = K >> 5;
  const int prow = lane >> 3;
  constexpr int PA = BM / 8 / 4, PB = BN / 8 / 4;  // 8-row DMA pieces per wave and tile: 4 + 4
  const int x7 = (l31 >> 1) & 7;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int mt = t / tiles_n, nt = t - mt * tiles_n;
    const int m0 = mt * BM, n0 = nt * BN;
    unsigned aoff[PA], boff[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int row = (wid * PA + i) * 8 + prow;
      aoff[i] = (unsigned)((m0 + row) * K + (((lane & 7) ^ ((row >> 1) & 7)) << 2));
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int row = (wid * PB + i) * 8 + prow;
      boff[i] = (unsigned)((n0 + row) * K + (((lane & 7) ^ ((row >> 1) & 7)) << 2));
    }
    auto issue = [&](int kt, int st) {
      const float* a = A + (kt << 5);
      const float* b = B + (kt << 5);
      float* base = smem + st * STAGE;
#pragma unroll
      for (int i = 0; i < PA; ++i) glds16(a + aoff[i], (lds_t)(base + (wid * PA + i) * 256));
#pragma unroll
      for (int i = 0; i < PB; ++i) glds16(b + boff[i], (lds_t)(base + BM * 32 + (wid * PB + i) * 256));
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    __syncthreads();  // the previous tile's readers are done with the ring
    // prologue: NST - 1 tiles in flight
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
      if (s < KT) issue(s, s);
    if (NST == 3) {
      if (KT > 1)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int kt = 0; kt < KT; ++kt) {
      const bool more = kt + NST - 1 < KT;
      int stn = cur + NST - 1;
      if (stn >= NST) stn -= NST;
      const float* cA = smem + cur * STAGE + (wm0 + l31) * 32;
      const float* cB = smem + cur * STAGE + BM * 32 + (wn0 + l31) * 32;
      f32x4 fa[2][2], fb[2][2];
      {
        const int o = (lh ^ x7) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(cA + i * 32 * 32 + o);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[0][j] = *reinterpret_cast<const f32x4*>(cB + j * 32 * 32 + o);
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int cb = jj & 1, nb = cb ^ 1;
        if (jj < 3) {
          const int o = (((jj + 1) * 2 + lh) ^ x7) * 4;
#pragma unroll
          for (int i = 0; i < 2; ++i) fa[nb][i] = *reinterpret_cast<const f32x4*>(cA + i * 32 * 32 + o);
#pragma unroll
          for (int j = 0; j < 2; ++j) fb[nb][j] = *reinterpret_cast<const f32x4*>(cB + j * 32 * 32 + o);
        }
        if (jj == 0 && more) issue(kt + NST - 1, stn);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cb][i][e], fb[cb][j][e], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
      // tile kt + 1 must have landed (this wave's share); with 3 stages tile kt + 2 stays in flight
      if (NST == 3 && more)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur = cur + 1 == NST ? 0 : cur + 1;
    }
    // epilogue: straight from the accumulators (register r of a block = row (r&3) + 8 (r>>2) + 4 lh, column l31)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int col = n0 + wn0 + j * 32 + l31;
          C[(size_t)row * N + col] = acc[i][j][r];
        }
  }
}

template <int NST, int OCC>
static double run(const float* dA, const float* dB, float* dC, int M, int N, int K, int reps) {
  const int tiles_n = N / 128, ntiles = (M / 128) * tiles_n;
  const int lds = NST * (128 + 128) * 32 * 4;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_probe_kernel<NST, OCC>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int grid = 256 * OCC;
  if (grid > ntiles) grid = ntiles;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((gemm_probe_kernel<NST, OCC>), dim3(grid), dim3(256), lds, 0, dA, dB, dC, M, N, K, tiles_n, ntiles);
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((gemm_probe_kernel<NST, OCC>), dim3(grid), dim3(256), lds, 0, dA, dB, dC, M, N, K, tiles_n, ntiles);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return 2.0 * M * N * K * reps / (ms * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 1024;
  std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
  unsigned s = 12345;
  for (auto& v : hA) v = ((s = s * 1664525u + 1013904223u) >> 8) * (1.0f / 16777216.0f) - 0.5f;
  for (auto& v : hB) v = ((s = s * 1664525u + 1013904223u) >> 8) * (1.0f / 16777216.0f) - 0.5f;
  float *dA, *dB, *dC;
  CHECK(hipMalloc(&dA, hA.size() * 4));
  CHECK(hipMalloc(&dB, hB.size() * 4));
  CHECK(hipMalloc(&dC, (size_t)M * N * 4));
  CHECK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
  const double t31 = run<3, 1>(dA, dB, dC, M, N, K, 10);
  // correctness on a few entries (fp64 reference)
  std::vector<float> hC((size_t)M * N);
  CHECK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int q = 0; q < 64; ++q) {
    const int r = (q * 977) % M, c = (q * 3331) % N;
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * hB[(size_t)c * K + k];
    worst = fmax(worst, fabs(ref - hC[(size_t)r * N + c]));
  }
  const double t21 = run<2, 1>(dA, dB, dC, M, N, K, 10);
  const double t22 = run<2, 2>(dA, dB, dC, M, N, K, 10);
  printf("M %d N %d K %d  max abs err (64 samples) %.3e\n", M, N, K, worst);
  printf("3 stages, 1 workgroup / CU : %.1f TFLOP/s = %.3f of 157.3\n", t31, t31 / 157.3);
  printf("2 stages, 1 workgroup / CU : %.1f TFLOP/s = %.3f\n", t21, t21 / 157.3);
  printf("2 stages, 2 workgroups / CU: %.1f TFLOP/s = %.3f\n", t22, t22 / 157.3);
  return 0;
}
