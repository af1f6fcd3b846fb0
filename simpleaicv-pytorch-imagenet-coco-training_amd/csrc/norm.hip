// BatchNorm2d (training-mode batch statistics and eval-mode running statistics) for NHWC
// activations viewed as [R = N*H*W][C].  HBM-bound: every kernel streams float4 along C.
// Reference: nn.BatchNorm2d inside ConvBnActBlock (simpleAICV/classification/backbones/
// resnet.py:41), eps 1e-5, momentum 0.1, biased variance for normalisation, unbiased for the
// running estimate; ReLU (resnet.py:42) and the bottleneck residual add + ReLU
// (resnet.py:150-153) are fused into the apply pass.
//
// All cross-block reductions are two-stage (per-block partials in a workspace, then a fixed
// order fp64 sum) so results are bit-reproducible run to run -- no float atomics.
#include "sacv_common.h"

#define BN_THREADS 256

// partial[b][0][c] = sum_r x, partial[b][1][c] = sum_r x^2 over the block's rows
__global__ __launch_bounds__(BN_THREADS) void bn_stats_partial_kernel(const float* __restrict__ x,
                                                                       float* __restrict__ partial, int R, int C,
                                                                       int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int cg = C >> 2;
  const int tid = threadIdx.x;
  const int r_begin = blockIdx.x * rows_per_block;
  const int r_end = min(R, r_begin + rows_per_block);
  float* out = partial + (size_t)blockIdx.x * 2 * C;
  if (cg >= BN_THREADS) {
    for (int c4 = tid; c4 < cg; c4 += BN_THREADS) {
      f32x4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
      for (int r = r_begin; r < r_end; ++r) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)r * C + c4 * 4);
        s += v;
        q += v * v;
      }
      *reinterpret_cast<f32x4*>(out + c4 * 4) = s;
      *reinterpret_cast<f32x4*>(out + C + c4 * 4) = q;
    }
    return;
  }
  const int rl = BN_THREADS / cg;  // row lanes
  const int c4 = tid % cg, rlane = tid / cg;
  f32x4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
  if (rlane < rl) {
    for (int r = r_begin + rlane; r < r_end; r += rl) {
      f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)r * C + c4 * 4);
      s += v;
      q += v * v;
    }
  }
  // LDS reduce across row lanes: sm[rl][2][C]
  if (rlane < rl) {
    *reinterpret_cast<f32x4*>(sm + (size_t)rlane * 2 * C + c4 * 4) = s;
    *reinterpret_cast<f32x4*>(sm + (size_t)rlane * 2 * C + C + c4 * 4) = q;
  }
  __syncthreads();
  for (int i = tid; i < 2 * C; i += BN_THREADS) {
    float acc = 0.f;
    for (int l = 0; l < rl; ++l) acc += sm[(size_t)l * 2 * C + i];
    out[i] = acc;
  }
}

// Second stage: 16 channels per 256-thread block, 16 threads per channel each summing a fixed strided
// subset of the per-block partials (coalesced 64-byte rows), then a fixed-order fp64 combine in LDS.
#define FIN_CH 4
#define FIN_PARTS 64
__device__ __forceinline__ void finalize_sums(const float* __restrict__ partial, int nblocks, int C, int c, int part,
                                              bool valid, double* red, double& s, double& q) {
  double ls = 0.0, lq = 0.0;
  if (valid) {
    for (int b = part; b < nblocks; b += FIN_PARTS) {
      ls += (double)partial[(size_t)b * 2 * C + c];
      lq += (double)partial[(size_t)b * 2 * C + C + c];
    }
  }
  const int cl = threadIdx.x % FIN_CH;
  red[(part * FIN_CH + cl) * 2 + 0] = ls;
  red[(part * FIN_CH + cl) * 2 + 1] = lq;
  __syncthreads();
  s = 0.0;
  q = 0.0;
  if (part == 0) {
    for (int k = 0; k < FIN_PARTS; ++k) {
      s += red[(k * FIN_CH + cl) * 2 + 0];
      q += red[(k * FIN_CH + cl) * 2 + 1];
    }
  }
}

__global__ __launch_bounds__(FIN_CH* FIN_PARTS) void bn_stats_finalize_kernel(
    const float* __restrict__ partial, int nblocks, int R, int C, float eps, float momentum, float* __restrict__ mean,
    float* __restrict__ invstd, float* __restrict__ running_mean, float* __restrict__ running_var,
    long long* __restrict__ num_batches_tracked) {
  __shared__ double red[FIN_CH * FIN_PARTS * 2];
  const int cl = threadIdx.x % FIN_CH, part = threadIdx.x / FIN_CH;
  const int c = blockIdx.x * FIN_CH + cl;
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += 1;
  double s, q;
  finalize_sums(partial, nblocks, C, c, part, c < C, red, s, q);
  if (part != 0 || c >= C) return;
  double m = s / R;
  double var = q / R - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    double unbiased = (R > 1) ? var * ((double)R / (double)(R - 1)) : var;
    running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * m);
    running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
  }
}

// z = act( (y - mean) * invstd * gamma + beta [+ residual] );  stat_is_var: invstd holds a variance
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        const float* __restrict__ residual, float* __restrict__ z,
                                                        size_t n4, int C, int relu, int stat_is_var, float eps) {
  const int cg = C >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    int c4 = (int)(i % cg) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(y + i * 4);
    f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4);
    f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c4);
    f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c4);
    f32x4 b = *reinterpret_cast<const f32x4*>(beta + c4);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float inv = stat_is_var ? (1.0f / sqrtf(is[k] + eps)) : is[k];
      o[k] = (v[k] - mu[k]) * inv * g[k] + b[k];
    }
    if (residual) o += *reinterpret_cast<const f32x4*>(residual + i * 4);
    if (relu) {
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = fmaxf(o[k], 0.f);
    }
    *reinterpret_cast<f32x4*>(z + i * 4) = o;
  }
}

// backward partials: partial[b][0][c] = sum g, partial[b][1][c] = sum g*xhat,  g = dz * (z > 0).
// The ReLU mask comes from `zmask` (the stored block output) or, when rc_beta is given, is recomputed from y
// with the forward's own expression ((y - mean) * invstd) * gamma + beta (bit-identical: -ffp-contract=off),
// which saves reading z (layers without a residual input).
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_partial_kernel(const float* __restrict__ dz,
                                                                     const float* __restrict__ zmask,
                                                                     const float* __restrict__ y,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd,
                                                                     const float* __restrict__ rc_gamma,
                                                                     const float* __restrict__ rc_beta,
                                                                     float* __restrict__ partial, int R, int C,
                                                                     int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int cg = C >> 2;
  const int tid = threadIdx.x;
  const int r_begin = blockIdx.x * rows_per_block;
  const int r_end = min(R, r_begin + rows_per_block);
  float* out = partial + (size_t)blockIdx.x * 2 * C;
  auto accum = [&](int r, int c4, const f32x4& mu, const f32x4& is, f32x4& s, f32x4& q) {
    size_t o = (size_t)r * C + c4 * 4;
    f32x4 g = *reinterpret_cast<const f32x4*>(dz + o);
    f32x4 v = *reinterpret_cast<const f32x4*>(y + o);
    const f32x4 xhat = (v - mu) * is;
    if (zmask) {
      f32x4 zz = *reinterpret_cast<const f32x4*>(zmask + o);
#pragma unroll
      for (int k = 0; k < 4; ++k) g[k] = zz[k] > 0.f ? g[k] : 0.f;
    } else if (rc_beta) {
      const f32x4 ga = *reinterpret_cast<const f32x4*>(rc_gamma + c4 * 4);
      const f32x4 be = *reinterpret_cast<const f32x4*>(rc_beta + c4 * 4);
      const f32x4 zz = xhat * ga + be;
#pragma unroll
      for (int k = 0; k < 4; ++k) g[k] = zz[k] > 0.f ? g[k] : 0.f;
    }
    s += g;
    q += g * xhat;
  };
  if (cg >= BN_THREADS) {
    for (int c4 = tid; c4 < cg; c4 += BN_THREADS) {
      f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4 * 4);
      f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c4 * 4);
      f32x4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
      for (int r = r_begin; r < r_end; ++r) accum(r, c4, mu, is, s, q);
      *reinterpret_cast<f32x4*>(out + c4 * 4) = s;
      *reinterpret_cast<f32x4*>(out + C + c4 * 4) = q;
    }
    return;
  }
  const int rl = BN_THREADS / cg;
  const int c4 = tid % cg, rlane = tid / cg;
  f32x4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
  if (rlane < rl) {
    f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4 * 4);
    f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c4 * 4);
    for (int r = r_begin + rlane; r < r_end; r += rl) accum(r, c4, mu, is, s, q);
    *reinterpret_cast<f32x4*>(sm + (size_t)rlane * 2 * C + c4 * 4) = s;
    *reinterpret_cast<f32x4*>(sm + (size_t)rlane * 2 * C + C + c4 * 4) = q;
  }
  __syncthreads();
  for (int i = tid; i < 2 * C; i += BN_THREADS) {
    float acc = 0.f;
    for (int l = 0; l < rl; ++l) acc += sm[(size_t)l * 2 * C + i];
    out[i] = acc;
  }
}

// dgamma/dbeta (assign or accumulate) + per-channel means kept in `coef` for the apply pass:
// coef[0][c] = sum_g / R, coef[1][c] = sum_gx / R
__global__ __launch_bounds__(FIN_CH* FIN_PARTS) void bn_bwd_finalize_kernel(const float* __restrict__ partial,
                                                                              int nblocks, int R, int C,
                                                                              float* __restrict__ dgamma,
                                                                              float* __restrict__ dbeta,
                                                                              float* __restrict__ coef,
                                                                              int accumulate) {
  __shared__ double red[FIN_CH * FIN_PARTS * 2];
  const int cl = threadIdx.x % FIN_CH, part = threadIdx.x / FIN_CH;
  const int c = blockIdx.x * FIN_CH + cl;
  double s, q;
  finalize_sums(partial, nblocks, C, c, part, c < C, red, s, q);
  if (part != 0 || c >= C) return;
  if (accumulate) {
    dbeta[c] += (float)s;
    dgamma[c] += (float)q;
  } else {
    dbeta[c] = (float)s;
    dgamma[c] = (float)q;
  }
  coef[c] = (float)(s / R);
  coef[C + c] = (float)(q / R);
}

// dy = gamma*invstd * (g - mean_g - xhat*mean_gx);  optional g_out = g (masked upstream grad)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dz,
                                                            const float* __restrict__ zmask,
                                                            const float* __restrict__ y,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ rc_beta,
                                                            const float* __restrict__ coef, float* __restrict__ dy,
                                                            float* __restrict__ g_out, size_t n4, int C) {
  const int cg = C >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    int c4 = (int)(i % cg) * 4;
    f32x4 g = *reinterpret_cast<const f32x4*>(dz + i * 4);
    f32x4 v = *reinterpret_cast<const f32x4*>(y + i * 4);
    f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4);
    f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c4);
    f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c4);
    f32x4 mg = *reinterpret_cast<const f32x4*>(coef + c4);
    f32x4 mgx = *reinterpret_cast<const f32x4*>(coef + C + c4);
    f32x4 xhat = (v - mu) * is;
    if (zmask) {
      f32x4 zz = *reinterpret_cast<const f32x4*>(zmask + i * 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) g[k] = zz[k] > 0.f ? g[k] : 0.f;
    } else if (rc_beta) {
      const f32x4 zz = xhat * ga + *reinterpret_cast<const f32x4*>(rc_beta + c4);
#pragma unroll
      for (int k = 0; k < 4; ++k) g[k] = zz[k] > 0.f ? g[k] : 0.f;
    }
    f32x4 o = (g - mg - xhat * mgx) * (ga * is);
    *reinterpret_cast<f32x4*>(dy + i * 4) = o;
    if (g_out) *reinterpret_cast<f32x4*>(g_out + i * 4) = g;
  }
}

static int pick_blocks(int R, int* rows_per_block) {
  int nb = R < 512 ? (R + 7) / 8 : 512;
  if (nb < 1) nb = 1;
  int rpb = (R + nb - 1) / nb;
  nb = (R + rpb - 1) / rpb;
  *rows_per_block = rpb;
  return nb;
}

static size_t lds_for(int C) {
  int cg = C / 4;
  if (cg >= BN_THREADS) return 0;
  int rl = BN_THREADS / cg;
  return (size_t)rl * 2 * C * sizeof(float);
}

extern "C" {

// floats of workspace needed by the BN kernels for a [R][C] activation
long long sacv_bn_workspace_floats(int R, int C) {
  int rpb;
  int nb = pick_blocks(R, &rpb);
  return (long long)nb * 2 * C + 2 * C;
}

int sacv_bn_train_stats(const float* y, int R, int C, float eps, float momentum, float* mean, float* invstd,
                        float* running_mean, float* running_var, long long* num_batches_tracked, float* workspace,
                        void* stream) {
  SACV_CHECK_ARG(y && mean && invstd && workspace, "bn_train_stats: null pointer");
  SACV_CHECK_ARG(C % 4 == 0 && C > 0 && R > 0, "bn_train_stats: C must be a positive multiple of 4 (C=%d)", C);
  SACV_CHECK_ARG(lds_for(C) <= 64 * 1024, "bn_train_stats: C=%d unsupported", C);
  int rpb;
  int nb = pick_blocks(R, &rpb);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(nb), dim3(BN_THREADS), lds_for(C), st, y, workspace, R, C, rpb);
  SACV_CHECK_LAUNCH("bn_stats_partial");
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_PARTS), 0, st,
                     workspace, nb, R, C, eps,
                     momentum, mean, invstd, running_mean, running_var, num_batches_tracked);
  SACV_CHECK_LAUNCH("bn_stats_finalize");
  return SACV_OK;
}

// Second stage alone: `partial` is [nblocks][2][C] column sums / sums of squares produced elsewhere
// (the CF_STATS epilogue of sacv_conv2d_gemm), R = number of rows they cover.
int sacv_bn_finalize_stats(const float* partial, int nblocks, int R, int C, float eps, float momentum, float* mean,
                           float* invstd, float* running_mean, float* running_var, long long* num_batches_tracked,
                           void* stream) {
  SACV_CHECK_ARG(partial && mean && invstd && nblocks > 0 && R > 0 && C > 0, "bn_finalize_stats: bad args");
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_PARTS), 0,
                     (hipStream_t)stream, partial, nblocks, R, C, eps, momentum, mean, invstd, running_mean,
                     running_var, num_batches_tracked);
  SACV_CHECK_LAUNCH("bn_stats_finalize");
  return SACV_OK;
}

int sacv_bn_apply(const float* y, const float* mean, const float* invstd_or_var, const float* gamma,
                  const float* beta, const float* residual, float* z, long long numel, int C, int relu,
                  int stat_is_var, float eps, void* stream) {
  SACV_CHECK_ARG(y && mean && invstd_or_var && gamma && beta && z, "bn_apply: null pointer");
  SACV_CHECK_ARG(C % 4 == 0 && numel % C == 0, "bn_apply: bad shape");
  size_t n4 = (size_t)numel / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, mean, invstd_or_var, gamma,
                     beta, residual, z, n4, C, relu, stat_is_var, eps);
  SACV_CHECK_LAUNCH("bn_apply");
  return SACV_OK;
}

// dz: upstream grad; y: conv output (pre-BN).  ReLU mask: zmask = the block output (or null); with zmask null and
// beta non-null the mask is recomputed from y (layers whose output is relu(bn(y)), no residual).
int sacv_bn_train_bwd(const float* dz, const float* zmask, const float* y, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, float* dy, float* g_out, float* dgamma, float* dbeta,
                      int R, int C, int accumulate, float* workspace, void* stream) {
  SACV_CHECK_ARG(dz && y && mean && invstd && gamma && dy && dgamma && dbeta && workspace, "bn_train_bwd: null");
  SACV_CHECK_ARG(C % 4 == 0 && C > 0 && R > 0, "bn_train_bwd: bad shape");
  int rpb;
  int nb = pick_blocks(R, &rpb);
  hipStream_t st = (hipStream_t)stream;
  float* coef = workspace + (size_t)nb * 2 * C;
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nb), dim3(BN_THREADS), lds_for(C), st, dz, zmask, y, mean, invstd,
                     gamma, zmask ? nullptr : beta, workspace, R, C, rpb);
  SACV_CHECK_LAUNCH("bn_bwd_partial");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_PARTS), 0, st,
                     workspace, nb, R, C, dgamma,
                     dbeta, coef, accumulate);
  SACV_CHECK_LAUNCH("bn_bwd_finalize");
  size_t n4 = (size_t)R * C / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, st, dz, zmask, y, mean, invstd, gamma,
                     zmask ? nullptr : beta, coef, dy, g_out, n4, C);
  SACV_CHECK_LAUNCH("bn_bwd_apply");
  return SACV_OK;
}

}  // extern "C"
