// bf16-storage versions of the trunk's HBM-bound helpers (the autocast classification recipe):
// nn.MaxPool2d(3,2,1) (simpleAICV/classification/backbones/resnet.py:189, :230) and nn.AdaptiveAvgPool2d((1,1))
// (:203, :241) on NHWC bf16 activations.  8 channels (16 bytes) per thread and access; comparisons and sums in fp32.
#include "sacv_common.h"

typedef sacv_bf16 bf16;
typedef sacv_bf16x8 bf16x8;

static int grid_for_h(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b > 65536 ? 65536 : (b < 1 ? 1 : b));
}

__global__ __launch_bounds__(256) void maxpool_fwd_bf16_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                                unsigned char* __restrict__ idx, int Nb, int H, int W,
                                                                int C, int Ho, int Wo, int k, int s, int p) {
  const int cg = C >> 3;
  const size_t total = (size_t)Nb * Ho * Wo * cg;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % cg);
    size_t q = i / cg;
    const int wo = (int)(q % Wo);
    q /= Wo;
    const int ho = (int)(q % Ho);
    const int n = (int)(q / Ho);
    float best[8];
    int bi[8];
    bool first[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      best[e] = -INFINITY;
      bi[e] = 0;
      first[e] = true;
    }
    for (int kh = 0; kh < k; ++kh) {
      const int hi = ho * s - p + kh;
      if ((unsigned)hi >= (unsigned)H) continue;
      for (int kw = 0; kw < k; ++kw) {
        const int wi = wo * s - p + kw;
        if ((unsigned)wi >= (unsigned)W) continue;
        float v[8];
        VecIO<bf16>::load(x + ((size_t)(n * H + hi) * W + wi) * C + c8 * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          // first maximum in window scan order wins (torch max_pool2d); NaN propagates
          if (first[e] || v[e] > best[e] || v[e] != v[e]) {
            best[e] = v[e];
            bi[e] = kh * k + kw;
            first[e] = false;
          }
        }
      }
    }
    const size_t o = ((size_t)(n * Ho + ho) * Wo + wo) * C + c8 * 8;
    VecIO<bf16>::store(y + o, best);
    if (idx) {
      unsigned long long packed = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) packed |= (unsigned long long)(bi[e] & 255) << (8 * e);
      *reinterpret_cast<unsigned long long*>(idx + o) = packed;
    }
  }
}

// gather form of the backward: each input pixel sums the windows that selected it (fp32 sum, one rounding)
__global__ __launch_bounds__(256) void maxpool_bwd_bf16_kernel(const bf16* __restrict__ dy,
                                                                const unsigned char* __restrict__ idx,
                                                                bf16* __restrict__ dx, int Nb, int H, int W, int C,
                                                                int Ho, int Wo, int k, int s, int p) {
  const int cg = C >> 3;
  const size_t total = (size_t)Nb * H * W * cg;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % cg);
    size_t q = i / cg;
    const int wi = (int)(q % W);
    q /= W;
    const int hi = (int)(q % H);
    const int n = (int)(q / H);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    int ho_lo = (hi + p - (k - 1) + s - 1) / s;
    if (hi + p - (k - 1) < 0) ho_lo = 0;
    const int ho_hi = min(Ho - 1, (hi + p) / s);
    int wo_lo = (wi + p - (k - 1) + s - 1) / s;
    if (wi + p - (k - 1) < 0) wo_lo = 0;
    const int wo_hi = min(Wo - 1, (wi + p) / s);
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
      const int kh = hi - (ho * s - p);
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        const int kw = wi - (wo * s - p);
        const int code = kh * k + kw;
        const size_t o = ((size_t)(n * Ho + ho) * Wo + wo) * C + c8 * 8;
        const unsigned long long b = *reinterpret_cast<const unsigned long long*>(idx + o);
        float g[8];
        VecIO<bf16>::load(dy + o, g);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if ((int)((b >> (8 * e)) & 255) == code) acc[e] += g[e];
      }
    }
    VecIO<bf16>::store(dx + i * 8, acc);
  }
}

// out[n][c] = mean_hw x[n][hw][c]: one block per (image, 64-channel slab), fixed-order fp32 sums
__global__ __launch_bounds__(256) void avgpool_fwd_bf16_kernel(const bf16* __restrict__ x, bf16* __restrict__ out,
                                                                int N, int HW, int C) {
  __shared__ float red[32][65];
  const int slabs = (C + 63) / 64;
  const int n = blockIdx.x / slabs, sl = blockIdx.x - n * slabs;
  const int c8 = threadIdx.x & 7, rl = threadIdx.x >> 3;  // 8 chunks x 32 row lanes
  const int c = sl * 64 + c8 * 8;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (c < C) {
    const bf16* p = x + (size_t)n * HW * C + c;
    for (int r = rl; r < HW; r += 32) {
      float v[8];
      VecIO<bf16>::load(p + (size_t)r * C, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][c8 * 8 + e] = s[e];
  __syncthreads();
  if (threadIdx.x < 64 && sl * 64 + threadIdx.x < C) {
    float t = 0.f;
    for (int l = 0; l < 32; ++l) t += red[l][threadIdx.x];
    out[(size_t)n * C + sl * 64 + threadIdx.x] = (bf16)(t / (float)HW);
  }
}

// dx[n][hw][c] = dy[n][c] / HW
__global__ __launch_bounds__(256) void avgpool_bwd_bf16_kernel(const bf16* __restrict__ dy, bf16* __restrict__ dx,
                                                                int N, int HW, int C) {
  const int cg = C >> 3;
  const size_t total = (size_t)N * HW * cg;
  const float inv = (float)HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % cg);
    const int n = (int)(i / ((size_t)HW * cg));
    float g[8];
    VecIO<bf16>::load(dy + (size_t)n * C + c8 * 8, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = g[e] / inv;
    VecIO<bf16>::store(dx + i * 8, g);
  }
}

SACV_API_BEGIN

int sacv_maxpool_fwd_bf16(const void* x, void* y, unsigned char* idx, int Nb, int H, int W, int C, int Ho, int Wo,
                          int k, int s, int p, void* stream) {
  SACV_CHECK_ARG(x && y, "maxpool_fwd_bf16: null pointer");
  SACV_CHECK_ARG(C % 8 == 0 && k * k <= 255, "maxpool_fwd_bf16: bad C/k");
  SACV_CHECK_ARG(Ho == (H + 2 * p - k) / s + 1 && Wo == (W + 2 * p - k) / s + 1, "maxpool_fwd_bf16: output dims");
  const size_t total = (size_t)Nb * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(maxpool_fwd_bf16_kernel, dim3(grid_for_h(total)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)x, (bf16*)y, idx, Nb, H, W, C, Ho, Wo, k, s, p);
  SACV_CHECK_LAUNCH("maxpool_fwd_bf16");
  return SACV_OK;
}

int sacv_maxpool_bwd_bf16(const void* dy, const unsigned char* idx, void* dx, int Nb, int H, int W, int C, int Ho,
                          int Wo, int k, int s, int p, void* stream) {
  SACV_CHECK_ARG(dy && idx && dx, "maxpool_bwd_bf16: null pointer");
  SACV_CHECK_ARG(C % 8 == 0, "maxpool_bwd_bf16: C %% 8");
  const size_t total = (size_t)Nb * H * W * (C / 8);
  hipLaunchKernelGGL(maxpool_bwd_bf16_kernel, dim3(grid_for_h(total)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)dy, idx, (bf16*)dx, Nb, H, W, C, Ho, Wo, k, s, p);
  SACV_CHECK_LAUNCH("maxpool_bwd_bf16");
  return SACV_OK;
}

int sacv_avgpool_fwd_bf16(const void* x, void* out, int N, int HW, int C, void* stream) {
  SACV_CHECK_ARG(x && out && N > 0 && HW > 0 && C % 8 == 0, "avgpool_fwd_bf16: bad args");
  hipLaunchKernelGGL(avgpool_fwd_bf16_kernel, dim3(N * ((C + 63) / 64)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)x, (bf16*)out, N, HW, C);
  SACV_CHECK_LAUNCH("avgpool_fwd_bf16");
  return SACV_OK;
}

int sacv_avgpool_bwd_bf16(const void* dy, void* dx, int N, int HW, int C, void* stream) {
  SACV_CHECK_ARG(dy && dx && N > 0 && HW > 0 && C % 8 == 0, "avgpool_bwd_bf16: bad args");
  const size_t total = (size_t)N * HW * (C / 8);
  hipLaunchKernelGGL(avgpool_bwd_bf16_kernel, dim3(grid_for_h(total)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)dy, (bf16*)dx, N, HW, C);
  SACV_CHECK_LAUNCH("avgpool_bwd_bf16");
  return SACV_OK;
}

SACV_API_END
