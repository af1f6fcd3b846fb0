// HBM-bound NHWC helpers of the detection step: 3x3/s2 max-pool, bilinear(align_corners)
// upsample + lateral add of the FPN top-down path, activation backward passes and the bias
// gradient column sum.  All stream float4 along the channel axis; reductions are two-stage and
// deterministic.  References: nn.MaxPool2d(3,2,1) simpleAICV/detection/models/backbones/
// resnet.py:50; F.interpolate(mode='bilinear', align_corners=True)+add fpn.py:65-73;
// nn.ReLU head.py:27,67 / fpn.py:56; nn.Sigmoid head.py:50.
#include "sacv_common.h"

// ---------------------------------------------------------------- max pool
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           unsigned char* __restrict__ idx, int Nb, int H, int W,
                                                           int C, int Ho, int Wo, int k, int s, int p) {
  const int cg = C >> 2;
  size_t total = (size_t)Nb * Ho * Wo * cg;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c4 = (int)(i % cg);
    size_t q = i / cg;
    int wo = (int)(q % Wo);
    q /= Wo;
    int ho = (int)(q % Ho);
    int n = (int)(q / Ho);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
    bool first[4] = {true, true, true, true};
    for (int kh = 0; kh < k; ++kh) {
      int hi = ho * s - p + kh;
      if ((unsigned)hi >= (unsigned)H) continue;
      for (int kw = 0; kw < k; ++kw) {
        int wi = wo * s - p + kw;
        if ((unsigned)wi >= (unsigned)W) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(x + ((size_t)(n * H + hi) * W + wi) * C + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // first maximum in window scan order wins (torch CPU max_pool2d); NaN propagates
          if (first[e] || v[e] > best[e] || v[e] != v[e]) {
            best[e] = v[e];
            bi[e] = kh * k + kw;
            first[e] = false;
          }
        }
      }
    }
    size_t o = ((size_t)(n * Ho + ho) * Wo + wo) * C + c4 * 4;
    *reinterpret_cast<f32x4*>(y + o) = best;
    if (idx) {
      uchar4 b = make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]);
      *reinterpret_cast<uchar4*>(idx + o) = b;
    }
  }
}

// gather form of the backward: each input pixel sums the windows that selected it
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy,
                                                           const unsigned char* __restrict__ idx,
                                                           float* __restrict__ dx, int Nb, int H, int W, int C,
                                                           int Ho, int Wo, int k, int s, int p) {
  const int cg = C >> 2;
  size_t total = (size_t)Nb * H * W * cg;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c4 = (int)(i % cg);
    size_t q = i / cg;
    int wi = (int)(q % W);
    q /= W;
    int hi = (int)(q % H);
    int n = (int)(q / H);
    f32x4 acc = {0, 0, 0, 0};
    // windows ho with ho*s - p <= hi <= ho*s - p + k - 1
    int ho_lo = (hi + p - (k - 1) + s - 1) / s;
    if (hi + p - (k - 1) < 0) ho_lo = 0;
    int ho_hi = min(Ho - 1, (hi + p) / s);
    int wo_lo = (wi + p - (k - 1) + s - 1) / s;
    if (wi + p - (k - 1) < 0) wo_lo = 0;
    int wo_hi = min(Wo - 1, (wi + p) / s);
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
      int kh = hi - (ho * s - p);
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        int kw = wi - (wo * s - p);
        int code = kh * k + kw;
        size_t o = ((size_t)(n * Ho + ho) * Wo + wo) * C + c4 * 4;
        uchar4 b = *reinterpret_cast<const uchar4*>(idx + o);
        f32x4 g = *reinterpret_cast<const f32x4*>(dy + o);
        if (b.x == code) acc[0] += g[0];
        if (b.y == code) acc[1] += g[1];
        if (b.z == code) acc[2] += g[2];
        if (b.w == code) acc[3] += g[3];
      }
    }
    *reinterpret_cast<f32x4*>(dx + i * 4) = acc;
  }
}

// ---------------------------------------------------------------- bilinear upsample (align_corners) + add
__device__ __forceinline__ void src_index(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
  float src = scale * (float)dst;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = src - (float)i0;
}

// out[n,h,w,c] = lateral[n,h,w,c] + bilinear(top)[n,h,w,c]
__global__ __launch_bounds__(256) void upsample_add_fwd_kernel(const float* __restrict__ top,
                                                                const float* __restrict__ lateral,
                                                                float* __restrict__ out, int Nb, int Hs, int Ws,
                                                                int Hd, int Wd, int C, float sh, float sw) {
  const int cg = C >> 2;
  size_t total = (size_t)Nb * Hd * Wd * cg;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c4 = (int)(i % cg);
    size_t q = i / cg;
    int w = (int)(q % Wd);
    q /= Wd;
    int h = (int)(q % Hd);
    int n = (int)(q / Hd);
    int h0, h1, w0, w1;
    float lh, lw;
    src_index(h, sh, Hs, h0, h1, lh);
    src_index(w, sw, Ws, w0, w1, lw);
    float hh = 1.f - lh, hw = 1.f - lw;
    const float* b = top + (size_t)n * Hs * Ws * C + c4 * 4;
    f32x4 v00 = *reinterpret_cast<const f32x4*>(b + ((size_t)h0 * Ws + w0) * C);
    f32x4 v01 = *reinterpret_cast<const f32x4*>(b + ((size_t)h0 * Ws + w1) * C);
    f32x4 v10 = *reinterpret_cast<const f32x4*>(b + ((size_t)h1 * Ws + w0) * C);
    f32x4 v11 = *reinterpret_cast<const f32x4*>(b + ((size_t)h1 * Ws + w1) * C);
    f32x4 up = hh * (hw * v00 + lw * v01) + lh * (hw * v10 + lw * v11);
    f32x4 lat = *reinterpret_cast<const f32x4*>(lateral + i * 4);
    *reinterpret_cast<f32x4*>(out + i * 4) = up + lat;
  }
}

// dtop[n,hs,ws,c] (+)= sum over destination pixels that sampled (hs,ws) of weight * dout
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dtop,
                                                            int Nb, int Hs, int Ws, int Hd, int Wd, int C, float sh,
                                                            float sw, int accumulate) {
  const int cg = C >> 2;
  size_t total = (size_t)Nb * Hs * Ws * cg;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c4 = (int)(i % cg);
    size_t q = i / cg;
    int ws = (int)(q % Ws);
    q /= Ws;
    int hs = (int)(q % Hs);
    int n = (int)(q / Hs);
    // candidate destination range (generous; exact membership is re-tested with the forward formula)
    int h_lo = (sh > 0.f) ? max(0, (int)floorf((float)(hs - 1) / sh) - 1) : 0;
    int h_hi = (sh > 0.f) ? min(Hd - 1, (int)ceilf((float)(hs + 1) / sh) + 1) : Hd - 1;
    int w_lo = (sw > 0.f) ? max(0, (int)floorf((float)(ws - 1) / sw) - 1) : 0;
    int w_hi = (sw > 0.f) ? min(Wd - 1, (int)ceilf((float)(ws + 1) / sw) + 1) : Wd - 1;
    f32x4 acc = {0, 0, 0, 0};
    for (int h = h_lo; h <= h_hi; ++h) {
      int h0, h1;
      float lh;
      src_index(h, sh, Hs, h0, h1, lh);
      float wy = 0.f;
      if (h0 == hs) wy += 1.f - lh;
      if (h1 == hs) wy += lh;
      if (h0 != hs && h1 != hs) continue;
      for (int w = w_lo; w <= w_hi; ++w) {
        int w0, w1;
        float lw;
        src_index(w, sw, Ws, w0, w1, lw);
        if (w0 != ws && w1 != ws) continue;
        float wx = 0.f;
        if (w0 == ws) wx += 1.f - lw;
        if (w1 == ws) wx += lw;
        f32x4 g = *reinterpret_cast<const f32x4*>(dout + ((size_t)(n * Hd + h) * Wd + w) * C + c4 * 4);
        acc += (wy * wx) * g;
      }
    }
    if (accumulate) acc += *reinterpret_cast<const f32x4*>(dtop + i * 4);
    *reinterpret_cast<f32x4*>(dtop + i * 4) = acc;
  }
}

// ---------------------------------------------------------------- elementwise
// op 0: out = relu(a); 1: out = a * (b > 0); 2: out = a * b * (1 - b)  (sigmoid backward, b = prob);
// 3: out = a + b; 4: out = a * scalar; 5: out = a * scalar * b * (1 - b)
__global__ __launch_bounds__(256) void eltwise_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       float* __restrict__ out, size_t n, int op, float scalar) {
  size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 x = *reinterpret_cast<const f32x4*>(a + i * 4);
    f32x4 y = {0, 0, 0, 0};
    if (b) y = *reinterpret_cast<const f32x4*>(b + i * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v;
      switch (op) {
        case 0: v = fmaxf(x[e], 0.f); break;
        case 1: v = y[e] > 0.f ? x[e] : 0.f; break;
        case 2: v = x[e] * y[e] * (1.f - y[e]); break;
        case 3: v = x[e] + y[e]; break;
        case 4: v = x[e] * scalar; break;
        default: v = x[e] * scalar * y[e] * (1.f - y[e]); break;
      }
      o[e] = v;
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = o;
  }
  // tail
  size_t t = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) {
    float x = a[t], y = b ? b[t] : 0.f, v;
    switch (op) {
      case 0: v = fmaxf(x, 0.f); break;
      case 1: v = y > 0.f ? x : 0.f; break;
      case 2: v = x * y * (1.f - y); break;
      case 3: v = x + y; break;
      case 4: v = x * scalar; break;
      default: v = x * scalar * y * (1.f - y); break;
    }
    out[t] = v;
  }
}

// ---------------------------------------------------------------- column sum  out[c] (+)= sum_r x[r][c]
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                              int R, int C, int rows_per_block) {
  // C % 4 == 0: 16-byte loads, threads = (row lane, column group); the row lanes of a block are combined in a
  // fixed order through LDS (deterministic).  Other C: one column per thread.
  __shared__ float sm[256 * 4];
  const int r_begin = blockIdx.x * rows_per_block;
  const int r_end = min(R, r_begin + rows_per_block);
  const int cg = C >> 2;
  if ((C & 3) == 0 && cg <= 256) {
    const int rl = 256 / cg;  // row lanes
    const int c4 = threadIdx.x % cg, lane = threadIdx.x / cg;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (lane < rl) {
      int r = r_begin + lane;
      for (; r + rl < r_end; r += 2 * rl) {
        s0 += *reinterpret_cast<const f32x4*>(x + (size_t)r * C + c4 * 4);
        s1 += *reinterpret_cast<const f32x4*>(x + (size_t)(r + rl) * C + c4 * 4);
      }
      if (r < r_end) s0 += *reinterpret_cast<const f32x4*>(x + (size_t)r * C + c4 * 4);
      s0 += s1;
      *reinterpret_cast<f32x4*>(sm + (size_t)lane * C + c4 * 4) = s0;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      float acc = 0.f;
      for (int l = 0; l < rl; ++l) acc += sm[(size_t)l * C + c];
      partial[(size_t)blockIdx.x * C + c] = acc;
    }
    return;
  }
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int r = r_begin; r < r_end; ++r) s += x[(size_t)r * C + c];
    partial[(size_t)blockIdx.x * C + c] = s;
  }
}
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ partial, int nblocks,
                                                               int stride, int C, float* __restrict__ out,
                                                               int accumulate) {
  __shared__ double red[256];
  const int cl = threadIdx.x % 16, part = threadIdx.x / 16;
  const int c = blockIdx.x * 16 + cl;
  double s = 0.0;
  if (c < C)
    for (int b = part; b < nblocks; b += 16) s += (double)partial[(size_t)b * stride + c];
  red[part * 16 + cl] = s;
  __syncthreads();
  if (part == 0 && c < C) {
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t += red[k * 16 + cl];
    out[c] = accumulate ? out[c] + (float)t : (float)t;
  }
}

// up to 16 independent second stages in one launch (the parameter gradients of one transformer block: 4 Linear biases,
// 2 x (LayerNorm dbeta, dgamma)): blockIdx.y = job, same fixed summation order as colsum_finalize_kernel
#define SACV_COLSUM_MULTI 16
struct ColsumJobs {
  const float* partial[SACV_COLSUM_MULTI];
  float* out[SACV_COLSUM_MULTI];
  int nblocks[SACV_COLSUM_MULTI], stride[SACV_COLSUM_MULTI], C[SACV_COLSUM_MULTI], accumulate[SACV_COLSUM_MULTI];
};
__global__ __launch_bounds__(256) void colsum_finalize_multi_kernel(const ColsumJobs j) {
  __shared__ double red[256];
  const int job = blockIdx.y;
  const int C = j.C[job], nblocks = j.nblocks[job], stride = j.stride[job];
  const float* __restrict__ partial = j.partial[job];
  const int cl = threadIdx.x % 16, part = threadIdx.x / 16;
  const int c = blockIdx.x * 16 + cl;
  if (blockIdx.x * 16 >= C) return;
  double s = 0.0;
  if (c < C)
    for (int b = part; b < nblocks; b += 16) s += (double)partial[(size_t)b * stride + c];
  red[part * 16 + cl] = s;
  __syncthreads();
  if (part == 0 && c < C) {
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t += red[k * 16 + cl];
    float* out = j.out[job];
    out[c] = j.accumulate[job] ? out[c] + (float)t : (float)t;
  }
}

static int grid_for(size_t work, int cap = 8192) {
  size_t b = (work + 255) / 256;
  if (b < 1) b = 1;
  return (int)(b > (size_t)cap ? cap : b);
}

SACV_API_BEGIN

int sacv_maxpool_fwd(const float* x, float* y, unsigned char* idx, int Nb, int H, int W, int C, int Ho, int Wo,
                     int k, int s, int p, void* stream) {
  SACV_CHECK_ARG(x && y, "maxpool_fwd: null pointer");
  SACV_CHECK_ARG(C % 4 == 0 && k * k <= 255, "maxpool_fwd: bad C/k");
  SACV_CHECK_ARG(Ho == (H + 2 * p - k) / s + 1 && Wo == (W + 2 * p - k) / s + 1, "maxpool_fwd: output dims mismatch");
  size_t total = (size_t)Nb * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, idx, Nb, H,
                     W, C, Ho, Wo, k, s, p);
  SACV_CHECK_LAUNCH("maxpool_fwd");
  return SACV_OK;
}

int sacv_maxpool_bwd(const float* dy, const unsigned char* idx, float* dx, int Nb, int H, int W, int C, int Ho,
                     int Wo, int k, int s, int p, void* stream) {
  SACV_CHECK_ARG(dy && idx && dx, "maxpool_bwd: null pointer");
  SACV_CHECK_ARG(C % 4 == 0, "maxpool_bwd: C %% 4");
  size_t total = (size_t)Nb * H * W * (C / 4);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dy, idx, dx, Nb, H,
                     W, C, Ho, Wo, k, s, p);
  SACV_CHECK_LAUNCH("maxpool_bwd");
  return SACV_OK;
}

static float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

int sacv_upsample_bilinear_add_fwd(const float* top, const float* lateral, float* out, int Nb, int Hs, int Ws,
                                   int Hd, int Wd, int C, void* stream) {
  SACV_CHECK_ARG(top && lateral && out, "upsample_add_fwd: null pointer");
  SACV_CHECK_ARG(C % 4 == 0, "upsample_add_fwd: C %% 4");
  size_t total = (size_t)Nb * Hd * Wd * (C / 4);
  hipLaunchKernelGGL(upsample_add_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, top, lateral,
                     out, Nb, Hs, Ws, Hd, Wd, C, ac_scale(Hs, Hd), ac_scale(Ws, Wd));
  SACV_CHECK_LAUNCH("upsample_add_fwd");
  return SACV_OK;
}

int sacv_upsample_bilinear_bwd(const float* dout, float* dtop, int Nb, int Hs, int Ws, int Hd, int Wd, int C,
                               int accumulate, void* stream) {
  SACV_CHECK_ARG(dout && dtop, "upsample_bwd: null pointer");
  SACV_CHECK_ARG(C % 4 == 0, "upsample_bwd: C %% 4");
  size_t total = (size_t)Nb * Hs * Ws * (C / 4);
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dout, dtop, Nb, Hs,
                     Ws, Hd, Wd, C, ac_scale(Hs, Hd), ac_scale(Ws, Wd), accumulate);
  SACV_CHECK_LAUNCH("upsample_bwd");
  return SACV_OK;
}

int sacv_eltwise(const float* a, const float* b, float* out, long long n, int op, float scalar, void* stream) {
  SACV_CHECK_ARG(a && out && n >= 0, "eltwise: null pointer");
  SACV_CHECK_ARG(op >= 0 && op <= 5, "eltwise: unknown op %d", op);
  SACV_CHECK_ARG(b || op == 0 || op == 4, "eltwise: op %d needs a second operand", op);
  if (n == 0) return SACV_OK;
  hipLaunchKernelGGL(eltwise_kernel, dim3(grid_for((size_t)n / 4 + 4)), dim3(256), 0, (hipStream_t)stream, a, b, out,
                     (size_t)n, op, scalar);
  SACV_CHECK_LAUNCH("eltwise");
  return SACV_OK;
}

long long sacv_colsum_workspace_floats(int R, int C) {
  int nb = R < 256 ? 1 : (R < 4096 ? 64 : 512);
  return (long long)nb * C;
}

int sacv_colsum(const float* x, float* out, int R, int C, int accumulate, float* workspace, void* stream) {
  SACV_CHECK_ARG(x && out && workspace && R > 0 && C > 0, "colsum: bad args");
  int nb = R < 256 ? 1 : (R < 4096 ? 64 : 512);
  int rpb = (R + nb - 1) / nb;
  nb = (R + rpb - 1) / rpb;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb), dim3(256), 0, st, x, workspace, R, C, rpb);
  SACV_CHECK_LAUNCH("colsum_partial");
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, st, workspace, nb, C, C, out,
                     accumulate);
  SACV_CHECK_LAUNCH("colsum_finalize");
  return SACV_OK;
}

// out[c] (+)= sum_b partial[b*stride + c]: second stage for column sums produced elsewhere (the CF_STATS
// epilogue of sacv_conv2d_gemm: stride = 2*N) -- the bias gradient of the convolution that consumes that dy.
int sacv_colsum_finalize(const float* partial, int nblocks, int stride, int C, float* out, int accumulate,
                         void* stream) {
  SACV_CHECK_ARG(partial && out && nblocks > 0 && C > 0 && stride >= C, "colsum_finalize: bad args");
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, partial,
                     nblocks, stride, C, out, accumulate);
  SACV_CHECK_LAUNCH("colsum_finalize");
  return SACV_OK;
}

// n (<= 16) second stages in one launch: out[i][c] (+)= sum_b partial[i][b * stride[i] + c], c < C[i]
int sacv_colsum_finalize_multi(int n, const float* const* partial, const int* nblocks, const int* stride, const int* C,
                               float* const* out, const int* accumulate, void* stream) {
  SACV_CHECK_ARG(n >= 1 && n <= SACV_COLSUM_MULTI, "colsum_finalize_multi: n=%d (1..%d)", n, SACV_COLSUM_MULTI);
  ColsumJobs j;
  int cmax = 0;
  for (int i = 0; i < SACV_COLSUM_MULTI; ++i) {
    const int k = i < n ? i : 0;
    SACV_CHECK_ARG(partial[k] && out[k] && nblocks[k] > 0 && C[k] > 0 && stride[k] >= C[k],
                   "colsum_finalize_multi: bad job %d", k);
    j.partial[i] = partial[k];
    j.out[i] = out[k];
    j.nblocks[i] = nblocks[k];
    j.stride[i] = stride[k];
    j.C[i] = C[k];
    j.accumulate[i] = accumulate[k];
    if (i < n && C[k] > cmax) cmax = C[k];
  }
  hipLaunchKernelGGL(colsum_finalize_multi_kernel, dim3((cmax + 15) / 16, n), dim3(256), 0, (hipStream_t)stream, j);
  SACV_CHECK_LAUNCH("colsum_finalize_multi");
  return SACV_OK;
}

SACV_API_END

// ---------------------------------------------------------------- device-side data boundary
// DetectionCollater + Normalize (simpleAICV/detection/common.py:223-286): uint8 HWC images of different sizes ->
// one zero-padded fp32 batch [B][R][R][3] (channels_last storage of the logical [B,3,R,R] the model takes),
// value / 255 in fp32.  Moves 4x fewer bytes over PCIe than the reference's host-side float batch.
struct CollateArgs {
  const unsigned char* img[16];
  int h[16], w[16];
  int B, R;   // canvas R rows ...
  int RW;     // ... x RW columns (RW == R: the reference's square canvas)
};
__global__ __launch_bounds__(256) void collate_u8_kernel(const CollateArgs a, float* __restrict__ out) {
  const size_t per = (size_t)a.R * a.RW;
  const size_t total = per * a.B;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int b = (int)(i / per);
    size_t r = i - (size_t)b * per;
    int y = (int)(r / a.RW), x = (int)(r - (size_t)y * a.RW);
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if (y < a.h[b] && x < a.w[b]) {
      const unsigned char* p = a.img[b] + ((size_t)y * a.w[b] + x) * 3;
      v0 = (float)p[0] / 255.f;
      v1 = (float)p[1] / 255.f;
      v2 = (float)p[2] / 255.f;
    }
    float* o = out + i * 3;
    o[0] = v0;
    o[1] = v1;
    o[2] = v2;
  }
}

// evaluate_coco_detection post-steps (tools/scripts.py:742-758) in place on boxes [B][M][4]:
// boxes /= scale[b]; clip x1,y1 >= 0, x2 <= size[b][1], y2 <= size[b][0]; (x2,y2) -= (x1,y1)
__global__ void eval_postprocess_kernel(float* __restrict__ boxes, const float* __restrict__ scales,
                                        const float* __restrict__ sizes, int B, int M) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * M) return;
  int b = i / M;
  f32x4 v = *reinterpret_cast<f32x4*>(boxes + (size_t)i * 4);
  float s = scales[b];
  v[0] = v[0] / s;
  v[1] = v[1] / s;
  v[2] = v[2] / s;
  v[3] = v[3] / s;
  v[0] = fmaxf(v[0], 0.f);
  v[1] = fmaxf(v[1], 0.f);
  v[2] = fminf(v[2], sizes[b * 2 + 1]);
  v[3] = fminf(v[3], sizes[b * 2 + 0]);
  v[2] = v[2] - v[0];
  v[3] = v[3] - v[1];
  *reinterpret_cast<f32x4*>(boxes + (size_t)i * 4) = v;
}

SACV_API_BEGIN

// imgs: host array of B (<= 16) device pointers to uint8 [h][w][3]; out: fp32 [B][R][R][3]
int sacv_collate_u8(int B, const unsigned char* const* imgs, const int* h, const int* w, int R, float* out,
                    void* stream) {
  SACV_CHECK_ARG(B >= 1 && B <= 16 && imgs && h && w && out && R > 0, "collate_u8: bad args (B <= 16)");
  CollateArgs a;
  a.B = B;
  a.R = R;
  a.RW = R;
  for (int i = 0; i < B; ++i) {
    SACV_CHECK_ARG(imgs[i] && h[i] > 0 && w[i] > 0 && h[i] <= R && w[i] <= R, "collate_u8: image %d does not fit", i);
    a.img[i] = imgs[i];
    a.h[i] = h[i];
    a.w[i] = w[i];
  }
  size_t total = (size_t)B * R * R;
  hipLaunchKernelGGL(collate_u8_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a, out);
  SACV_CHECK_LAUNCH("collate_u8");
  return SACV_OK;
}

// as sacv_collate_u8 on a rectangular canvas out [B][Hc][Wc][3]: the padding-skip form of the collater -- the caller
// picks Hc x Wc >= the largest image of the batch (rounded up to the network's stride) instead of the reference's
// R x R square, so the zero padding of DetectionCollater (detection/common.py:251-268) is never convolved
int sacv_collate_u8_rect(int B, const unsigned char* const* imgs, const int* h, const int* w, int Hc, int Wc,
                         float* out, void* stream) {
  SACV_CHECK_ARG(B >= 1 && B <= 16 && imgs && h && w && out && Hc > 0 && Wc > 0, "collate_u8_rect: bad args (B <= 16)");
  CollateArgs a;
  a.B = B;
  a.R = Hc;
  a.RW = Wc;
  for (int i = 0; i < B; ++i) {
    SACV_CHECK_ARG(imgs[i] && h[i] > 0 && w[i] > 0 && h[i] <= Hc && w[i] <= Wc,
                   "collate_u8_rect: image %d does not fit", i);
    a.img[i] = imgs[i];
    a.h[i] = h[i];
    a.w[i] = w[i];
  }
  size_t total = (size_t)B * Hc * Wc;
  hipLaunchKernelGGL(collate_u8_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a, out);
  SACV_CHECK_LAUNCH("collate_u8_rect");
  return SACV_OK;
}

int sacv_eval_postprocess(float* boxes, const float* scales, const float* sizes, int B, int M, void* stream) {
  SACV_CHECK_ARG(boxes && scales && sizes && B > 0 && M > 0, "eval_postprocess: bad args");
  hipLaunchKernelGGL(eval_postprocess_kernel, dim3((B * M + 127) / 128), dim3(128), 0, (hipStream_t)stream, boxes,
                     scales, sizes, B, M);
  SACV_CHECK_LAUNCH("eval_postprocess");
  return SACV_OK;
}

SACV_API_END
