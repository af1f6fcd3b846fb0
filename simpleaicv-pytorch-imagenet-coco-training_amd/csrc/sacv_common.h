// Shared helpers for libsacv_hip.so (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define SACV_OK 0
#define SACV_ERR_ARG -1
#define SACV_ERR_LAUNCH -2
#define SACV_ERR_UNSUPPORTED -3

// The library is built with -fvisibility=hidden: only the C ABI of include/sacv_hip.h is exported.  Every entry point
// is defined between SACV_API_BEGIN and SACV_API_END (C linkage, default visibility); helpers shared between the
// translation units (sacv_internal_*, sacv_set_error) stay internal to the shared object.
#define SACV_API_BEGIN extern "C" { _Pragma("GCC visibility push(default)")
#define SACV_API_END _Pragma("GCC visibility pop") }
#define SACV_API extern "C" __attribute__((visibility("default")))

void sacv_set_error(const char* fmt, ...);

// Launch plan shared by the persistent kernels (errors.hip; set through sacv_launch_share): how many workgroup slots per CU
// a launch may plan for and how many CUs it leaves alone -- the caller runs a second queue (weight gradients, RCCL)
// beside the launch stream and the two have to FIT a CU's LDS / registers together.
struct SacvShare {
  int nt64_wg_per_cu;    // conv_nt_kernel<128,64>   (3 alone)
  int nt128_wg_per_cu;   // conv_nt_kernel<128,128>  (2 alone)
  int wino_wg_per_cu;    // wino_gemm_kernel, narrow items (3 alone; wide items: min(2, this))
  int wgrad_wg_per_cu;   // wino_wgrad_kernel plan (2 alone); conv_tn's split count is the caller's `chunk`
  int cu_reserve;        // CUs every persistent grid leaves free
  int split_policy;      // remainder tiles of conv_nt: 3 = cost model (uniform K split / stream-K / whole tiles), 0 = whole tiles only
  int wino_big;          // split-operand wino_gemm: 1 = cost model chooses how many rows go to 128 x 128 items, 0 = 64 x 64
                         // items only, 2 = every whole 128-row tile to the big items (tests)
  int nt_split;          // conv_nt: 1 = operands split into three bf16 terms at the LDS store, products on the bf16 pipe
};
extern SacvShare sacv_share;  // process-wide (errors.hip)

#define SACV_CHECK_ARG(cond, ...)          \
  do {                                     \
    if (!(cond)) {                         \
      sacv_set_error(__VA_ARGS__);         \
      return SACV_ERR_ARG;                 \
    }                                      \
  } while (0)

#define SACV_CHECK_LAUNCH(name)                                            \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      sacv_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return SACV_ERR_LAUNCH;                                              \
    }                                                                      \
  } while (0)

// internal (not part of the C ABI): fixed-order sum of the weight-gradient slabs part[chunks][Cout][KH*KW*Cin_pad]
// into dw (conv_gemm.hip); shared by the fp32 and the bf16 weight-gradient kernels
int sacv_internal_wgrad_reduce(const float* part, float* dw, int chunks, int Cout, int Cin, int Cin_pad, int KH,
                               int KW, int accumulate, int dw_taps, int dw_layout, void* stream);

// internal: plain bf16 NT GEMM on the LDS-DMA pipelined kernel (gemm_bf16.hip), used by sacv_conv2d_gemm_bf16 for
// stride-1 1x1 problems / Linear layers; flags = the CF_* bits of the bf16 convolution
// SacvBnBwdH (optional): the BatchNorm layer whose upstream gradient the launch stores (sacv_conv2d_gemm_bf16_bnbwd) --
// the epilogue then writes that layer's backward first-stage sums into `stats` instead of the column sums
struct SacvBnBwdH {
  const void* y;        // the layer's pre-normalisation input [M][N] bf16
  const void* z;        // ReLU mask source (the stored layer output, bf16) or null: mask recomputed from y
  const float* mean;    // [N]
  const float* invstd;  // [N]
  const float* gamma;   // [N] (mask recomputation)
  const float* beta;    // [N] (mask recomputation; null together with z: no ReLU, no mask)
};
int sacv_internal_gemm_nt_bf16_supported(long long M, int N, int K);
int sacv_internal_gemm_nt_bf16(const void* A, const void* B, void* C, const float* bias, const float* residual,
                               float* stats, long long M, int N, int K, int flags, void* stream,
                               const SacvBnBwdH* bn = nullptr);

// internal: halo-resident bf16 3x3 / stride 1 / pad 1 convolution with 64 output channels (conv3x3_halo_bf16.hip),
// used by sacv_conv2d_gemm_bf16 for the layer1-class problems; mode 0 forward, 1 data gradient; flags: CF_STATS only
int sacv_internal_conv3x3_halo_bf16_supported(int Nb, int H, int W, int Cs, int N);
int sacv_internal_conv3x3_halo_bf16_stats_rows(int Nb, int H, int W);
int sacv_internal_conv3x3_halo_bf16(const void* src, void* dst, int Nb, int H, int W, const void* w, int Cs, int N,
                                    int mode, int flags, float* stats, void* stream);

static inline int sacv_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 16-byte vector I/O on activations stored as fp32 (4 elements) or bf16 (8 elements); arithmetic is always fp32
typedef __bf16 sacv_bf16;
typedef __bf16 sacv_bf16x8 __attribute__((ext_vector_type(8)));
template <typename T>
struct VecIO;
template <>
struct VecIO<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = t[e];
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  }
};
template <>
struct VecIO<sacv_bf16> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const sacv_bf16* p, float (&v)[8]) {
    const sacv_bf16x8 t = *reinterpret_cast<const sacv_bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
  }
  static __device__ __forceinline__ void store(sacv_bf16* p, const float (&v)[8]) {
    sacv_bf16x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (sacv_bf16)v[e];
    *reinterpret_cast<sacv_bf16x8*>(p) = t;
  }
};
// V consecutive fp32 per-channel parameters (V = 4 or 8; 16-byte aligned)
template <int V>
__device__ __forceinline__ void ld_param(const float* p, float (&v)[V]) {
#pragma unroll
  for (int q = 0; q < V / 4; ++q) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * q + e] = t[e];
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
