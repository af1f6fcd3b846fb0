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

// Second stage: `fch` channels per 256-thread block (4, or 1 when there are many partial rows and few channels: a
// 64-channel layer at batch 256 has 6 272 rows of partials and would otherwise run on 16 workgroups), 256 / fch threads
// per channel each summing a fixed strided subset of the per-block partials, then a fixed-order fp64 combine in LDS.
#define FIN_THREADS 256
__device__ __forceinline__ int fin_channels(int nblocks, int C) { return (nblocks >= 2048 && C <= 1024) ? 1 : 4; }
static int fin_channels_host(int nblocks, int C) { return (nblocks >= 2048 && C <= 1024) ? 1 : 4; }
__device__ __forceinline__ void finalize_sums(const float* __restrict__ partial, int nblocks, int C, int c, int part,
                                              int fch, bool valid, double* red, double& s, double& q) {
  const int parts = FIN_THREADS / fch;
  double ls = 0.0, lq = 0.0;
  if (valid) {
    for (int b = part; b < nblocks; b += parts) {
      ls += (double)partial[(size_t)b * 2 * C + c];
      lq += (double)partial[(size_t)b * 2 * C + C + c];
    }
  }
  const int cl = threadIdx.x % fch;
  red[(part * fch + cl) * 2 + 0] = ls;
  red[(part * fch + cl) * 2 + 1] = lq;
  __syncthreads();
  s = 0.0;
  q = 0.0;
  if (part == 0) {
    for (int k = 0; k < parts; ++k) {
      s += red[(k * fch + cl) * 2 + 0];
      q += red[(k * fch + cl) * 2 + 1];
    }
  }
}

__global__ __launch_bounds__(FIN_THREADS) void bn_stats_finalize_kernel(
    const float* __restrict__ partial, int nblocks, int R, int C, float eps, float momentum, float* __restrict__ mean,
    float* __restrict__ invstd, float* __restrict__ running_mean, float* __restrict__ running_var,
    long long* __restrict__ num_batches_tracked) {
  __shared__ double red[FIN_THREADS * 2];
  const int fch = fin_channels(nblocks, C);
  const int cl = threadIdx.x % fch, part = threadIdx.x / fch;
  const int c = blockIdx.x * fch + cl;
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += 1;
  double s, q;
  finalize_sums(partial, nblocks, C, c, part, fch, c < C, red, s, q);
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

// z = act( (y - mean) * invstd * gamma + beta [+ residual] );  stat_is_var: invstd holds a variance.
// T = storage type of y / residual / z (fp32: 4 elements per thread and pass, bf16: 8); arithmetic in fp32.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ y, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const T* __restrict__ residual,
                                                        T* __restrict__ z, size_t nv, int C, int relu, int stat_is_var,
                                                        float eps) {
  constexpr int V = VecIO<T>::N;
  const int cg = C / V;
  // when the grid's thread count is a multiple of the channel-group count, a thread keeps its channel group for the
  // whole grid-stride loop: the four per-channel parameter vectors are loaded once instead of once per 16-byte access
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const bool fixed = (stride % (size_t)cg) == 0;
  float mu[V], is[V], g[V], b[V];
  int cprev = -1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    const int c0 = (int)(i % cg) * V;
    float v[V], o[V];
    VecIO<T>::load(y + i * V, v);
    if (!fixed || cprev < 0) {
      ld_param<V>(mean + c0, mu);
      ld_param<V>(invstd + c0, is);
      ld_param<V>(gamma + c0, g);
      ld_param<V>(beta + c0, b);
      if (stat_is_var) {
#pragma unroll
        for (int k = 0; k < V; ++k) is[k] = 1.0f / sqrtf(is[k] + eps);
      }
      cprev = c0;
    }
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = (v[k] - mu[k]) * is[k] * g[k] + b[k];
    if (residual) {
      float r[V];
      VecIO<T>::load(residual + i * V, r);
#pragma unroll
      for (int k = 0; k < V; ++k) o[k] += r[k];
    }
    if (relu) {
#pragma unroll
      for (int k = 0; k < V; ++k) o[k] = fmaxf(o[k], 0.f);
    }
    VecIO<T>::store(z + i * V, o);
  }
}

// backward partials: partial[b][0][c] = sum g, partial[b][1][c] = sum g*xhat,  g = dz * (z > 0).
// The ReLU mask comes from `zmask` (the stored block output) or, when rc_beta is given, is recomputed from y
// with the forward's own expression ((y - mean) * invstd) * gamma + beta (bit-identical: -ffp-contract=off),
// which saves reading z (layers without a residual input).  With bf16 storage the recomputed value is rounded to
// bf16 first, exactly like the forward pass stored it.
template <typename T>
__device__ __forceinline__ float storage_round(float v);
template <>
__device__ __forceinline__ float storage_round<float>(float v) { return v; }
template <>
__device__ __forceinline__ float storage_round<sacv_bf16>(float v) { return (float)(sacv_bf16)v; }

template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_partial_kernel(const T* __restrict__ dz,
                                                                     const T* __restrict__ zmask,
                                                                     const T* __restrict__ y,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd,
                                                                     const float* __restrict__ rc_gamma,
                                                                     const float* __restrict__ rc_beta,
                                                                     float* __restrict__ partial, int R, int C,
                                                                     int rows_per_block) {
  constexpr int V = VecIO<T>::N;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int cg = C / V;
  const int tid = threadIdx.x;
  const int r_begin = blockIdx.x * rows_per_block;
  const int r_end = min(R, r_begin + rows_per_block);
  float* out = partial + (size_t)blockIdx.x * 2 * C;
  float ga[V], be[V];  // recompute path: the thread's channel group is fixed in both branches below
  // one row: the arithmetic of a (dz, y[, z]) triple already in registers; rows are ALWAYS folded in ascending order,
  // so the sums do not depend on how many rows were loaded ahead
  auto fold = [&](const float (&g0)[V], const float (&v)[V], const float (&zz)[V], const float (&mu)[V],
                  const float (&is)[V], float (&s)[V], float (&q)[V]) {
    float g[V], xhat[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      g[k] = g0[k];
      xhat[k] = (v[k] - mu[k]) * is[k];
    }
    if (zmask) {
#pragma unroll
      for (int k = 0; k < V; ++k) g[k] = zz[k] > 0.f ? g[k] : 0.f;
    } else if (rc_beta) {
#pragma unroll
      for (int k = 0; k < V; ++k) g[k] = storage_round<T>(xhat[k] * ga[k] + be[k]) > 0.f ? g[k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
      s[k] += g[k];
      q[k] += g[k] * xhat[k];
    }
  };
  // rows r, r + step, ... < r_end of channel group cv.  Four rows are requested before the first is folded: with two
  // 256-thread blocks per CU a thread that waits for each row's two or three 16-byte loads in turn keeps ~16 KB per CU
  // in flight, half of what 5.5 TB/s x ~1.5 us of latency needs (the kernel ran at 0.53-0.60 of the HBM roofline).
  auto accum_rows = [&](int r0, int step, int cv, const float (&mu)[V], const float (&is)[V], float (&s)[V],
                        float (&q)[V]) {
    int r = r0;
#ifndef SACV_BN_AHEAD
#define SACV_BN_AHEAD 4  // rows requested before the first is folded (tuning builds: scripts/build_variant.sh)
#endif
    constexpr int AH = SACV_BN_AHEAD;
    for (; r + (AH - 1) * step < r_end; r += AH * step) {
      float g[AH][V], v[AH][V], zz[AH][V];
#pragma unroll
      for (int u = 0; u < AH; ++u) {
        const size_t o = (size_t)(r + u * step) * C + cv * V;
        VecIO<T>::load(dz + o, g[u]);
        VecIO<T>::load(y + o, v[u]);
        if (zmask) VecIO<T>::load(zmask + o, zz[u]);
      }
#pragma unroll
      for (int u = 0; u < AH; ++u) fold(g[u], v[u], zz[u], mu, is, s, q);
    }
    for (; r < r_end; r += step) {
      float g[V], v[V], zz[V];
      const size_t o = (size_t)r * C + cv * V;
      VecIO<T>::load(dz + o, g);
      VecIO<T>::load(y + o, v);
      if (zmask) VecIO<T>::load(zmask + o, zz);
      fold(g, v, zz, mu, is, s, q);
    }
  };
  if (cg >= BN_THREADS) {
    for (int cv = tid; cv < cg; cv += BN_THREADS) {
      float mu[V], is[V], s[V], q[V];
      ld_param<V>(mean + cv * V, mu);
      ld_param<V>(invstd + cv * V, is);
      if (rc_beta && !zmask) {
        ld_param<V>(rc_gamma + cv * V, ga);
        ld_param<V>(rc_beta + cv * V, be);
      }
#pragma unroll
      for (int k = 0; k < V; ++k) s[k] = q[k] = 0.f;
      accum_rows(r_begin, 1, cv, mu, is, s, q);
#pragma unroll
      for (int k = 0; k < V; ++k) {
        out[cv * V + k] = s[k];
        out[C + cv * V + k] = q[k];
      }
    }
    return;
  }
  const int rl = BN_THREADS / cg;
  const int cv = tid % cg, rlane = tid / cg;
  if (rlane < rl) {
    float mu[V], is[V], s[V], q[V];
    ld_param<V>(mean + cv * V, mu);
    ld_param<V>(invstd + cv * V, is);
    if (rc_beta && !zmask) {
      ld_param<V>(rc_gamma + cv * V, ga);
      ld_param<V>(rc_beta + cv * V, be);
    }
#pragma unroll
    for (int k = 0; k < V; ++k) s[k] = q[k] = 0.f;
    accum_rows(r_begin + rlane, rl, cv, mu, is, s, q);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      sm[(size_t)rlane * 2 * C + cv * V + k] = s[k];
      sm[(size_t)rlane * 2 * C + C + cv * V + k] = q[k];
    }
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
__global__ __launch_bounds__(FIN_THREADS) void bn_bwd_finalize_kernel(const float* __restrict__ partial,
                                                                              int nblocks, int R, int C,
                                                                              float* __restrict__ dgamma,
                                                                              float* __restrict__ dbeta,
                                                                              float* __restrict__ coef,
                                                                              int accumulate) {
  __shared__ double red[FIN_THREADS * 2];
  const int fch = fin_channels(nblocks, C);
  const int cl = threadIdx.x % fch, part = threadIdx.x / fch;
  const int c = blockIdx.x * fch + cl;
  double s, q;
  finalize_sums(partial, nblocks, C, c, part, fch, c < C, red, s, q);
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
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dz, const T* __restrict__ zmask,
                                                            const T* __restrict__ y, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ rc_beta,
                                                            const float* __restrict__ coef, T* __restrict__ dy,
                                                            T* __restrict__ g_out, size_t nv, int C) {
  constexpr int V = VecIO<T>::N;
  const int cg = C / V;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const bool fixed = (stride % (size_t)cg) == 0;  // see bn_apply_kernel
  float mu[V], is[V], ga[V], mg[V], mgx[V], be[V];
  int cprev = -1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    const int c0 = (int)(i % cg) * V;
    float g[V], v[V], xhat[V], o[V];
    VecIO<T>::load(dz + i * V, g);
    VecIO<T>::load(y + i * V, v);
    if (!fixed || cprev < 0) {
      ld_param<V>(mean + c0, mu);
      ld_param<V>(invstd + c0, is);
      ld_param<V>(gamma + c0, ga);
      ld_param<V>(coef + c0, mg);
      ld_param<V>(coef + C + c0, mgx);
      if (rc_beta) ld_param<V>(rc_beta + c0, be);
      cprev = c0;
    }
#pragma unroll
    for (int k = 0; k < V; ++k) xhat[k] = (v[k] - mu[k]) * is[k];
    if (zmask) {
      float zz[V];
      VecIO<T>::load(zmask + i * V, zz);
#pragma unroll
      for (int k = 0; k < V; ++k) g[k] = zz[k] > 0.f ? g[k] : 0.f;
    } else if (rc_beta) {
#pragma unroll
      for (int k = 0; k < V; ++k) g[k] = storage_round<T>(xhat[k] * ga[k] + be[k]) > 0.f ? g[k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = (g[k] - mg[k] - xhat[k] * mgx[k]) * (ga[k] * is[k]);
    VecIO<T>::store(dy + i * V, o);
    if (g_out) VecIO<T>::store(g_out + i * V, g);
  }
}

#ifndef SACV_BN_BLOCKS
#define SACV_BN_BLOCKS 512  // tuning builds only (scripts/build_variant.sh)
#endif
static int pick_blocks(int R, int* rows_per_block) {
  int nb = R < SACV_BN_BLOCKS ? (R + 7) / 8 : SACV_BN_BLOCKS;
  if (nb < 1) nb = 1;
  int rpb = (R + nb - 1) / nb;
  nb = (R + rpb - 1) / rpb;
  *rows_per_block = rpb;
  return nb;
}

static size_t lds_for(int C, int V = 4) {
  int cg = C / V;
  if (cg >= BN_THREADS) return 0;
  int rl = BN_THREADS / cg;
  return (size_t)rl * 2 * C * sizeof(float);
}

template <typename T>
static int bn_apply_impl(const T* y, const float* mean, const float* invstd_or_var, const float* gamma,
                         const float* beta, const T* residual, T* z, long long numel, int C, int relu, int stat_is_var,
                         float eps, void* stream) {
  constexpr int V = VecIO<T>::N;
  SACV_CHECK_ARG(y && mean && invstd_or_var && gamma && beta && z, "bn_apply: null pointer");
  SACV_CHECK_ARG(C % V == 0 && numel % C == 0, "bn_apply: bad shape (C=%d must be a multiple of %d)", C, V);
  size_t nv = (size_t)numel / V;
  int blocks = (int)((nv + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(bn_apply_kernel<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, mean, invstd_or_var, gamma,
                     beta, residual, z, nv, C, relu, stat_is_var, eps);
  SACV_CHECK_LAUNCH("bn_apply");
  return SACV_OK;
}

// dz: upstream grad; y: conv output (pre-BN).  ReLU mask: zmask = the block output (or null); with zmask null and
// beta non-null the mask is recomputed from y (layers whose output is relu(bn(y)), no residual).
// `partial` non-null: the first stage ran elsewhere (the producing data gradient's epilogue, sacv_conv2d_gemm_bnbwd):
// [nparts][2][C] rows of (sum g, sum g * xhat) covering the R rows
template <typename T>
static int bn_train_bwd_impl(const T* dz, const T* zmask, const T* y, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, T* dy, T* g_out, float* dgamma, float* dbeta,
                             int R, int C, int accumulate, float* workspace, void* stream,
                             const float* partial = nullptr, int nparts = 0) {
  constexpr int V = VecIO<T>::N;
  SACV_CHECK_ARG(dz && y && mean && invstd && gamma && dy && dgamma && dbeta && workspace, "bn_train_bwd: null");
  SACV_CHECK_ARG(C % V == 0 && C > 0 && R > 0, "bn_train_bwd: bad shape (C=%d must be a multiple of %d)", C, V);
  SACV_CHECK_ARG(lds_for(C, V) <= 64 * 1024, "bn_train_bwd: C=%d unsupported", C);
  int rpb;
  int nb = pick_blocks(R, &rpb);
  hipStream_t st = (hipStream_t)stream;
  float* coef = workspace + (size_t)nb * 2 * C;
  if (partial) {
    SACV_CHECK_ARG(nparts > 0, "bn_train_bwd_partial: nparts");
    nb = nparts;
  } else {
    hipLaunchKernelGGL(bn_bwd_partial_kernel<T>, dim3(nb), dim3(BN_THREADS), lds_for(C, V), st, dz, zmask, y, mean,
                       invstd, gamma, zmask ? nullptr : beta, workspace, R, C, rpb);
    SACV_CHECK_LAUNCH("bn_bwd_partial");
    partial = workspace;
  }
  const int FCH = fin_channels_host(nb, C);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + FCH - 1) / FCH), dim3(FIN_THREADS), 0, st,
                     partial, nb, R, C, dgamma,
                     dbeta, coef, accumulate);
  SACV_CHECK_LAUNCH("bn_bwd_finalize");
  size_t nv = (size_t)R * C / V;
  int blocks = (int)((nv + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(blocks), dim3(256), 0, st, dz, zmask, y, mean, invstd, gamma,
                     zmask ? nullptr : beta, coef, dy, g_out, nv, C);
  SACV_CHECK_LAUNCH("bn_bwd_apply");
  return SACV_OK;
}

// Eval-mode BatchNorm folded into the producing convolution's epilogue (SACV_CF_AFFINE): for every BatchNorm layer of a
// model, out[0][c] = beta - mean * scale (shift), out[1][c] = scale = gamma / sqrt(var + eps).  One launch for all
// layers: workgroup j handles record j of the table {gamma, beta, running_mean, running_var, out, C}.
struct BnFoldJob {
  const float* gamma;
  const float* beta;
  const float* mean;
  const float* var;
  float* out;  // [2][C]
  int C, pad;
};
__global__ __launch_bounds__(256) void bn_fold_batched_kernel(const BnFoldJob* __restrict__ jobs, float eps) {
  const BnFoldJob j = jobs[blockIdx.x];
  for (int c = threadIdx.x; c < j.C; c += 256) {
    const float is = 1.0f / sqrtf(j.var[c] + eps);  // as bn_apply's stat_is_var path
    const float sc = is * j.gamma[c];
    j.out[c] = j.beta[c] - j.mean[c] * sc;
    j.out[j.C + c] = sc;
  }
}

SACV_API_BEGIN

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
  const int FCH = fin_channels_host(nb, C);
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + FCH - 1) / FCH), dim3(FIN_THREADS), 0, st,
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
  const int FCH = fin_channels_host(nblocks, C);
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + FCH - 1) / FCH), dim3(FIN_THREADS), 0,
                     (hipStream_t)stream, partial, nblocks, R, C, eps, momentum, mean, invstd, running_mean,
                     running_var, num_batches_tracked);
  SACV_CHECK_LAUNCH("bn_stats_finalize");
  return SACV_OK;
}

int sacv_bn_apply(const float* y, const float* mean, const float* invstd_or_var, const float* gamma,
                  const float* beta, const float* residual, float* z, long long numel, int C, int relu,
                  int stat_is_var, float eps, void* stream) {
  return bn_apply_impl<float>(y, mean, invstd_or_var, gamma, beta, residual, z, numel, C, relu, stat_is_var, eps,
                              stream);
}
// bf16 activations (y, residual, z), fp32 statistics and affine parameters
int sacv_bn_apply_bf16(const void* y, const float* mean, const float* invstd_or_var, const float* gamma,
                       const float* beta, const void* residual, void* z, long long numel, int C, int relu,
                       int stat_is_var, float eps, void* stream) {
  return bn_apply_impl<sacv_bf16>((const sacv_bf16*)y, mean, invstd_or_var, gamma, beta, (const sacv_bf16*)residual,
                                  (sacv_bf16*)z, numel, C, relu, stat_is_var, eps, stream);
}

int sacv_bn_train_bwd(const float* dz, const float* zmask, const float* y, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, float* dy, float* g_out, float* dgamma, float* dbeta,
                      int R, int C, int accumulate, float* workspace, void* stream) {
  return bn_train_bwd_impl<float>(dz, zmask, y, mean, invstd, gamma, beta, dy, g_out, dgamma, dbeta, R, C, accumulate,
                                  workspace, stream);
}
// Second and third stage of sacv_bn_train_bwd from partial sums computed elsewhere: partial = [nparts][2][C] rows of
// (sum g, sum g * xhat) over the R rows, as written by sacv_conv2d_gemm_bnbwd (nparts = its stats rows).
int sacv_bn_train_bwd_partial(const float* dz, const float* zmask, const float* y, const float* mean,
                              const float* invstd, const float* gamma, const float* beta, float* dy, float* g_out,
                              float* dgamma, float* dbeta, int R, int C, int accumulate, const float* partial, int nparts,
                              float* workspace, void* stream) {
  SACV_CHECK_ARG(partial, "bn_train_bwd_partial: null partial sums");
  return bn_train_bwd_impl<float>(dz, zmask, y, mean, invstd, gamma, beta, dy, g_out, dgamma, dbeta, R, C, accumulate,
                                  workspace, stream, partial, nparts);
}
int sacv_bn_train_bwd_bf16(const void* dz, const void* zmask, const void* y, const float* mean, const float* invstd,
                           const float* gamma, const float* beta, void* dy, void* g_out, float* dgamma, float* dbeta,
                           int R, int C, int accumulate, float* workspace, void* stream) {
  return bn_train_bwd_impl<sacv_bf16>((const sacv_bf16*)dz, (const sacv_bf16*)zmask, (const sacv_bf16*)y, mean, invstd,
                                      gamma, beta, (sacv_bf16*)dy, (sacv_bf16*)g_out, dgamma, dbeta, R, C, accumulate,
                                      workspace, stream);
}
// bf16 twin of sacv_bn_train_bwd_partial: partial sums from sacv_conv2d_gemm_bf16_bnbwd
int sacv_bn_train_bwd_partial_bf16(const void* dz, const void* zmask, const void* y, const float* mean,
                                   const float* invstd, const float* gamma, const float* beta, void* dy, void* g_out,
                                   float* dgamma, float* dbeta, int R, int C, int accumulate, const float* partial,
                                   int nparts, float* workspace, void* stream) {
  SACV_CHECK_ARG(partial, "bn_train_bwd_partial_bf16: null partial sums");
  return bn_train_bwd_impl<sacv_bf16>((const sacv_bf16*)dz, (const sacv_bf16*)zmask, (const sacv_bf16*)y, mean, invstd,
                                      gamma, beta, (sacv_bf16*)dy, (sacv_bf16*)g_out, dgamma, dbeta, R, C, accumulate,
                                      workspace, stream, partial, nparts);
}

// jobs: device array of njobs records {const float* gamma, beta, running_mean, running_var; float* out /*[2][C]*/;
// int C; int pad} (48 bytes): out[0] = shift, out[1] = scale of the eval-mode BatchNorm (classification/backbones/
// resnet.py:41 in eval mode), the operand of SACV_CF_AFFINE.
int sacv_bn_fold_batched(const void* jobs, int njobs, float eps, void* stream) {
  SACV_CHECK_ARG(jobs && njobs > 0, "bn_fold_batched: bad args");
  hipLaunchKernelGGL(bn_fold_batched_kernel, dim3(njobs), dim3(256), 0, (hipStream_t)stream,
                     (const BnFoldJob*)jobs, eps);
  SACV_CHECK_LAUNCH("bn_fold_batched");
  return SACV_OK;
}

SACV_API_END
