// Prototype: fp32 GEMM C[M][N] = A[M][K] * W[N][K]^T on the bf16 matrix cores with a 3-way bf16 split of both
// operands (x = hi + mid + lo exactly; 6 of the 9 cross products kept, fp32 accumulate).  Measures the main-loop
// structure planned for the conv kernels: 128x128 tile, 4 waves, K step 16, LDS double buffer, A split at
// LDS-store time, W pre-split in HBM as [N][K/16][3][16] bf16.
//   hipcc --offload-arch=gfx950 -O3 scripts/bf16x3_gemm.hip -o /tmp/bf16x3 && /tmp/bf16x3 [M N K]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  unsigned u = __float_as_uint(x);
  h = u & 0xffff0000u;
  float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xffff0000u;
  float r2 = r1 - __uint_as_float(m);
  l = __float_as_uint(r2);
}
// two fp32 -> packed bf16 pair of each plane (element 0 in the low half)
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  unsigned h0, m0, l0, h1, m1, l1;
  split3(x0, h0, m0, l0);
  split3(x1, h1, m1, l1);
  h = (h0 >> 16) | h1;
  m = (m0 >> 16) | m1;
  l = (l0 >> 16) | (l1 & 0xffff0000u);
}

__global__ void presplit_w(const float* __restrict__ w, unsigned short* __restrict__ out, int N, int K) {
  size_t total = (size_t)N * K;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int n = (int)(i / K), k = (int)(i % K);
    unsigned h, m, l;
    split3(w[i], h, m, l);
    size_t base = ((size_t)n * (K / 16) + k / 16) * 48 + (k % 16);
    out[base] = (unsigned short)(h >> 16);
    out[base + 16] = (unsigned short)(m >> 16);
    out[base + 32] = (unsigned short)(l >> 16);
  }
}

constexpr int BM = 128, BN = 128, KS = 16, ROWB = 112;  // LDS row: 3 planes x 32 B + 16 B pad

template <int NPROD>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3(const float* __restrict__ A, const unsigned char* __restrict__ Wp,
                                                      float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;                      // [2][BM][ROWB]
  unsigned char* sB = smem + 2 * BM * ROWB;      // [2][BN][ROWB]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ntn = N / BN;
  const int KT = K / KS;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm0 = (wid >> 1) * 64, wn0 = (wid & 1) * 64;
  // A gather: thread -> (row r0 + 64 i, k quad kq)
  const int kq = tid & 3, r0 = tid >> 2;
  for (int tile = blockIdx.x; tile < (M / BM) * ntn; tile += gridDim.x) {
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const float* a_ptr[2];
    a_ptr[0] = A + (size_t)(m0 + r0) * K + kq * 4;
    a_ptr[1] = A + (size_t)(m0 + r0 + 64) * K + kq * 4;
    // B copy: piece f = tid + 256 i -> row f/6, piece f%6
    const unsigned char* b_ptr[3];
    int b_lds[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      int f = tid + 256 * i;
      int row = f / 6, piece = f - row * 6;
      b_ptr[i] = Wp + ((size_t)(n0 + row) * KT) * 96 + piece * 16;
      b_lds[i] = row * ROWB + piece * 16;
    }
    f32x4 ra[2];
    u32x4 rb[3];
    auto gload = [&](int kt) {
#pragma unroll
      for (int i = 0; i < 2; ++i) ra[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + kt * KS);
#pragma unroll
      for (int i = 0; i < 3; ++i) rb[i] = *reinterpret_cast<const u32x4*>(b_ptr[i] + (size_t)kt * 96);
    };
    auto lstore = [&](int buf) {
      unsigned char* dA = sA + buf * BM * ROWB;
      unsigned char* dB = sB + buf * BN * ROWB;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        unsigned h0, m0, l0, h1, m1, l1;
        split3_pair(ra[i][0], ra[i][1], h0, m0, l0);
        split3_pair(ra[i][2], ra[i][3], h1, m1, l1);
        u32x2 h = {h0, h1}, m = {m0, m1}, l = {l0, l1};
        unsigned char* p = dA + (r0 + 64 * i) * ROWB + kq * 8;
        *reinterpret_cast<u32x2*>(p) = h;
        *reinterpret_cast<u32x2*>(p + 32) = m;
        *reinterpret_cast<u32x2*>(p + 64) = l;
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x4*>(dB + b_lds[i]) = rb[i];
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    __syncthreads();
    gload(0);
    lstore(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < KT; ++kt) {
      const bool last = kt + 1 == KT;
      if (!last) gload(kt + 1);
      const unsigned char* cA = sA + cur * BM * ROWB + (wm0 + l31) * ROWB + lh * 16;
      const unsigned char* cB = sB + cur * BN * ROWB + (wn0 + l31) * ROWB + lh * 16;
      bf16x8 fa[2][3], fb[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          fa[i][p] = *reinterpret_cast<const bf16x8*>(cA + i * 32 * ROWB + p * 32);
          fb[i][p] = *reinterpret_cast<const bf16x8*>(cB + i * 32 * ROWB + p * 32);
        }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // small terms first
          if (NPROD >= 6) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], acc[i][j], 0, 0, 0);
          }
          if (NPROD >= 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
        }
      __builtin_amdgcn_s_setprio(0);
      if (!last) lstore(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
    // epilogue: direct (uncoalesced) stores -- prototype only
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          int col = n0 + wn0 + j * 32 + l31;
          C[(size_t)row * N + col] = acc[i][j][r];
        }
  }
}

int main(int argc, char** argv) {
  int M = argc > 3 ? atoi(argv[1]) : 44544, N = argc > 3 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 2304;
  printf("M %d N %d K %d\n", M, N, K);
  std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f + ((s >> 24) / 256.0f) * 1e-3f; };
  for (auto& v : hA) v = rnd();
  for (auto& v : hW) v = rnd();
  float *dA, *dW, *dC;
  unsigned char* dWp;
  CK(hipMalloc(&dA, hA.size() * 4));
  CK(hipMalloc(&dW, hW.size() * 4));
  CK(hipMalloc(&dC, (size_t)M * N * 4));
  CK(hipMalloc(&dWp, (size_t)N * K * 6));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(presplit_w, dim3(1024), dim3(256), 0, 0, dW, (unsigned short*)dWp, N, K);
  CK(hipDeviceSynchronize());
  size_t lds = 2 * (BM + BN) * ROWB;
  int grid = 512;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int nprod : {6, 3, 1}) {
    auto launch = [&]() {
      if (nprod == 6) hipLaunchKernelGGL(gemm_bf16x3<6>, dim3(grid), dim3(256), lds, 0, dA, dWp, dC, M, N, K);
      if (nprod == 3) hipLaunchKernelGGL(gemm_bf16x3<3>, dim3(grid), dim3(256), lds, 0, dA, dWp, dC, M, N, K);
      if (nprod == 1) hipLaunchKernelGGL(gemm_bf16x3<1>, dim3(grid), dim3(256), lds, 0, dA, dWp, dC, M, N, K);
    };
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int it = 0; it < 10; ++it) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 10;
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    // accuracy on a sample vs fp64, with the plain fp32 dot product's error as the yardstick
    double e_gpu = 0, e_f32 = 0, ref_n = 0;
    for (int t = 0; t < 4000; ++t) {
      int m = (int)((t * 7919LL) % M), n = (int)((t * 104729LL) % N);
      double r = 0;
      float f = 0.f;
      for (int k = 0; k < K; ++k) {
        r += (double)hA[(size_t)m * K + k] * hW[(size_t)n * K + k];
        f = fmaf(hA[(size_t)m * K + k], hW[(size_t)n * K + k], f);
      }
      e_gpu += (hC[(size_t)m * N + n] - r) * (hC[(size_t)m * N + n] - r);
      e_f32 += (f - r) * (f - r);
      ref_n += r * r;
    }
    printf("nprod %d: %.3f ms  %.1f TF(fp32-equivalent)  rel-L2 err vs fp64: gpu %.3e  fp32-fma-chain %.3e\n", nprod, ms,
           2.0 * M * N * K / (ms * 1e-3) / 1e12, sqrt(e_gpu / ref_n), sqrt(e_f32 / ref_n));
  }
  return 0;
}
