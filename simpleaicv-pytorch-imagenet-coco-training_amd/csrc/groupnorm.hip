// GroupNorm(32, C) + ReLU for the FCOS head towers (simpleAICV/detection/models/head.py:110,125:
// nn.GroupNorm(32, inplanes) after every tower conv), NHWC activations [N][HW][C], eps 1e-5.
// Statistics are per (image, group) over HW x (C/32) values.  HBM-bound float4 streams; two-stage
// fixed-order reductions in fp64 (deterministic).  The affine parameters are shared by the five pyramid
// levels, so the parameter gradients accumulate across calls (`accumulate`).
#include "sacv_common.h"

#define GN_THREADS 256

// partial[(n*nb + b)][0][c] = sum_r v0, [1][c] = sum_r v1 over the block's rows of image n, where
//   fwd : v0 = x,              v1 = x*x
//   bwd : v0 = g = dz*(z>0),   v1 = g*xhat
template <int BWD>
__global__ __launch_bounds__(GN_THREADS) void gn_partial_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ dz,
                                                                 const float* __restrict__ zmask,
                                                                 const float* __restrict__ stats, int HW, int C,
                                                                 int G, int rows_per_block,
                                                                 float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int cg = C >> 2;            // float4 groups per row (<= 256 required)
  const int rl = GN_THREADS / cg;   // row lanes
  const int tid = threadIdx.x;
  const int c4 = tid % cg, rlane = tid / cg;
  const int n = blockIdx.y;
  const int r_begin = blockIdx.x * rows_per_block;
  const int r_end = min(HW, r_begin + rows_per_block);
  const int cpg = C / G;
  f32x4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
  if (rlane < rl) {
    float mu = 0.f, is = 0.f;
    if (BWD) {
      int g = (c4 * 4) / cpg;  // cpg % 4 == 0: a float4 never straddles groups
      mu = stats[((size_t)n * G + g) * 2];
      is = stats[((size_t)n * G + g) * 2 + 1];
    }
    for (int r = r_begin + rlane; r < r_end; r += rl) {
      size_t o = ((size_t)n * HW + r) * C + c4 * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(x + o);
      if (BWD) {
        f32x4 g = *reinterpret_cast<const f32x4*>(dz + o);
        if (zmask) {
          f32x4 zz = *reinterpret_cast<const f32x4*>(zmask + o);
#pragma unroll
          for (int e = 0; e < 4; ++e) g[e] = zz[e] > 0.f ? g[e] : 0.f;
        }
        s += g;
        q += g * ((v - mu) * is);
      } else {
        s += v;
        q += v * v;
      }
    }
    *reinterpret_cast<f32x4*>(sm + (size_t)rlane * 2 * C + c4 * 4) = s;
    *reinterpret_cast<f32x4*>(sm + (size_t)rlane * 2 * C + C + c4 * 4) = q;
  }
  __syncthreads();
  float* out = partial + ((size_t)n * gridDim.x + blockIdx.x) * 2 * C;
  for (int i = tid; i < 2 * C; i += GN_THREADS) {
    float acc = 0.f;
    for (int l = 0; l < rl; ++l) acc += sm[(size_t)l * 2 * C + i];
    out[i] = acc;
  }
}

// forward second stage: stats[n][g] = (mean, invstd).  One wave per (image, group): lane l sums the elements
// l, l+64, ... of the nb x (C/G) partials in fp64, then a fixed shuffle tree (deterministic).
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ partial, int nb, int N, int HW,
                                                          int C, int G, float eps, float* __restrict__ stats) {
  const int i = blockIdx.x;  // (n, g)
  const int n = i / G, g = i - n * G;
  const int cpg = C / G;
  const int total = nb * cpg;
  double s = 0.0, q = 0.0;
  for (int e = threadIdx.x; e < total; e += 64) {
    const int b = e / cpg, c = g * cpg + (e - b * cpg);
    const float* p = partial + ((size_t)n * nb + b) * 2 * C;
    s += (double)p[c];
    q += (double)p[C + c];
  }
  s = wave_sum_d(s);
  q = wave_sum_d(q);
  if (threadIdx.x == 0) {
    double m = (double)HW * cpg;
    double mean = s / m;
    double var = q / m - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[(size_t)i * 2] = (float)mean;
    stats[(size_t)i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ z,
                                                        int N, int HW, int C, int G, int relu) {
  const int cg = C >> 2;
  const int cpg = C / G;
  size_t total = (size_t)N * HW * cg;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c4 = (int)(i % cg) * 4;
    int n = (int)(i / ((size_t)HW * cg));
    int g = c4 / cpg;
    float mu = stats[((size_t)n * G + g) * 2], is = stats[((size_t)n * G + g) * 2 + 1];
    f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
    f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c4);
    f32x4 be = *reinterpret_cast<const f32x4*>(beta + c4);
    f32x4 o = (v - mu) * is * ga + be;
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
    }
    *reinterpret_cast<f32x4*>(z + i * 4) = o;
  }
}

// backward second stage.  Per (n,c): A = sum_hw g, B = sum_hw g*xhat.
//   dbeta[c] (+)= sum_n A, dgamma[c] (+)= sum_n B
//   coef[n][g] = (sum_{c in g} gamma[c]*A / m, sum_{c in g} gamma[c]*B / m),  m = HW * C/G
// One block per group g: the N x C/G (image, channel) pairs of the group share the block's 256 threads, `lanes`
// threads per pair summing every lanes-th partial row in fp64; fixed-order combines through LDS (deterministic).
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ partial, int nb, int N, int HW,
                                                               int C, int G, const float* __restrict__ gamma,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ coef, int accumulate) {
  __shared__ double shA[256], shB[256];
  const int g = blockIdx.x, tid = threadIdx.x;
  const int cpg = C / G;
  const int pairs = N * cpg;  // <= 256 (checked on the host)
  const int lanes = 256 / pairs;
  const int pair = tid / lanes, lane = tid - pair * lanes;
  double a = 0.0, b = 0.0;
  if (pair < pairs) {
    const int n = pair / cpg, c = g * cpg + (pair - n * cpg);
    for (int k = lane; k < nb; k += lanes) {
      const float* p = partial + ((size_t)n * nb + k) * 2 * C;
      a += (double)p[c];
      b += (double)p[C + c];
    }
  }
  shA[tid] = a;
  shB[tid] = b;
  __syncthreads();
  if (pair < pairs && lane == 0) {
    for (int l = 1; l < lanes; ++l) {
      a += shA[tid + l];
      b += shB[tid + l];
    }
  }
  __syncthreads();
  if (pair < pairs && lane == 0) {  // compact: pair p at index p
    shA[pair] = a;
    shB[pair] = b;
  }
  __syncthreads();
  if (tid < cpg) {  // parameter gradients of channel c: sum over the images
    const int c = g * cpg + tid;
    double sa = 0.0, sb = 0.0;
    for (int n = 0; n < N; ++n) {
      sa += shA[n * cpg + tid];
      sb += shB[n * cpg + tid];
    }
    if (accumulate) {
      dbeta[c] += (float)sa;
      dgamma[c] += (float)sb;
    } else {
      dbeta[c] = (float)sa;
      dgamma[c] = (float)sb;
    }
  }
  if (tid >= 64 && tid < 64 + N) {  // coefficients of (image n, group g)
    const int n = tid - 64;
    const double m = (double)HW * cpg;
    double s1 = 0.0, s2 = 0.0;
    for (int c = 0; c < cpg; ++c) {
      s1 += (double)gamma[g * cpg + c] * shA[n * cpg + c];
      s2 += (double)gamma[g * cpg + c] * shB[n * cpg + c];
    }
    coef[((size_t)n * G + g) * 2] = (float)(s1 / m);
    coef[((size_t)n * G + g) * 2 + 1] = (float)(s2 / m);
  }
}

// dx = invstd * (g*gamma - c1 - xhat*c2)
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                            const float* __restrict__ zmask,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ coef, float* __restrict__ dx,
                                                            int N, int HW, int C, int G) {
  const int cg = C >> 2;
  const int cpg = C / G;
  size_t total = (size_t)N * HW * cg;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c4 = (int)(i % cg) * 4;
    int n = (int)(i / ((size_t)HW * cg));
    int gi = n * G + c4 / cpg;
    float mu = stats[(size_t)gi * 2], is = stats[(size_t)gi * 2 + 1];
    float c1 = coef[(size_t)gi * 2], c2 = coef[(size_t)gi * 2 + 1];
    f32x4 g = *reinterpret_cast<const f32x4*>(dz + i * 4);
    if (zmask) {
      f32x4 zz = *reinterpret_cast<const f32x4*>(zmask + i * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = zz[e] > 0.f ? g[e] : 0.f;
    }
    f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
    f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c4);
    f32x4 xhat = (v - mu) * is;
    f32x4 o = (g * ga - c1 - xhat * c2) * is;
    *reinterpret_cast<f32x4*>(dx + i * 4) = o;
  }
}

static int gn_blocks(int HW, int* rpb) {
  int nb = HW < 2048 ? (HW + 31) / 32 : 128;
  if (nb < 1) nb = 1;
  if (nb > 128) nb = 128;
  int r = (HW + nb - 1) / nb;
  nb = (HW + r - 1) / r;
  *rpb = r;
  return nb;
}

SACV_API_BEGIN

long long sacv_gn_workspace_floats(int N, int HW, int C) {
  int rpb;
  int nb = gn_blocks(HW, &rpb);
  return (long long)N * nb * 2 * C + 2LL * N * C;
}

// x: [N][HW][C]; stats out: [N][G][2] (mean, invstd); z = relu?(gn(x))
int sacv_groupnorm_fwd(const float* x, const float* gamma, const float* beta, float* z, float* stats, int N, int HW,
                       int C, int G, float eps, int relu, float* workspace, void* stream) {
  SACV_CHECK_ARG(x && gamma && beta && z && stats && workspace, "groupnorm_fwd: null pointer");
  SACV_CHECK_ARG(C % G == 0 && (C / G) % 4 == 0 && C / 4 <= GN_THREADS && N > 0 && HW > 0,
                 "groupnorm_fwd: need C %% G == 0, (C/G) %% 4 == 0, C <= 1024");
  int rpb;
  int nb = gn_blocks(HW, &rpb);
  hipStream_t st = (hipStream_t)stream;
  int rl = GN_THREADS / (C / 4);
  size_t lds = (size_t)rl * 2 * C * sizeof(float);
  hipLaunchKernelGGL((gn_partial_kernel<0>), dim3(nb, N), dim3(GN_THREADS), lds, st, x, nullptr, nullptr, nullptr,
                     HW, C, G, rpb, workspace);
  SACV_CHECK_LAUNCH("gn_partial");
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(N * G), dim3(64), 0, st, workspace, nb, N, HW, C, G, eps, stats);
  SACV_CHECK_LAUNCH("gn_finalize");
  size_t total = (size_t)N * HW * (C / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gn_apply_kernel, dim3(blocks), dim3(256), 0, st, x, stats, gamma, beta, z, N, HW, C, G, relu);
  SACV_CHECK_LAUNCH("gn_apply");
  return SACV_OK;
}

// dz: grad w.r.t. the (ReLU'd) output; zmask = that output (or NULL when no ReLU); dx out.
int sacv_groupnorm_bwd(const float* x, const float* dz, const float* zmask, const float* stats, const float* gamma,
                       float* dx, float* dgamma, float* dbeta, int N, int HW, int C, int G, int accumulate,
                       float* workspace, void* stream) {
  SACV_CHECK_ARG(x && dz && stats && gamma && dx && dgamma && dbeta && workspace, "groupnorm_bwd: null pointer");
  SACV_CHECK_ARG(C % G == 0 && (C / G) % 4 == 0 && C / 4 <= GN_THREADS, "groupnorm_bwd: bad C/G");
  SACV_CHECK_ARG(N * (C / G) <= 256 && N <= 192, "groupnorm_bwd: N * C/G must be <= 256");
  int rpb;
  int nb = gn_blocks(HW, &rpb);
  hipStream_t st = (hipStream_t)stream;
  int rl = GN_THREADS / (C / 4);
  size_t lds = (size_t)rl * 2 * C * sizeof(float);
  float* coef = workspace + (size_t)N * nb * 2 * C;
  hipLaunchKernelGGL((gn_partial_kernel<1>), dim3(nb, N), dim3(GN_THREADS), lds, st, x, dz, zmask, stats, HW, C, G,
                     rpb, workspace);
  SACV_CHECK_LAUNCH("gn_bwd_partial");
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(G), dim3(256), 0, st, workspace, nb, N, HW, C, G, gamma, dgamma,
                     dbeta, coef, accumulate);
  SACV_CHECK_LAUNCH("gn_bwd_finalize");
  size_t total = (size_t)N * HW * (C / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, st, x, dz, zmask, stats, gamma, coef, dx, N, HW,
                     C, G);
  SACV_CHECK_LAUNCH("gn_bwd_apply");
  return SACV_OK;
}

SACV_API_END
