// ResNet50 image-classification tail: global average pool, softmax cross-entropy and the SGD
// update.  References: nn.AdaptiveAvgPool2d((1,1)) simpleAICV/classification/backbones/resnet.py:203, :241;
// CELoss = nn.CrossEntropyLoss(reduction='mean') simpleAICV/classification/losses.py:14-26; SGD with momentum
// as built by tools/utils.py::build_optimizer.  The FC layer itself runs on the conv GEMM kernels (1x1 conv).
#include "sacv_common.h"

// out[n][c] = mean_hw x[n][hw][c]
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int N,
                                                           int HW, int C) {
  const int cg = C >> 2;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * cg) return;
  int n = i / cg, c4 = i - n * cg;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const float* p = x + (size_t)n * HW * C + c4 * 4;
  for (int r = 0; r < HW; ++r) s += *reinterpret_cast<const f32x4*>(p + (size_t)r * C);
  const float inv = (float)HW;
  f32x4 o = {s[0] / inv, s[1] / inv, s[2] / inv, s[3] / inv};
  *reinterpret_cast<f32x4*>(out + (size_t)n * C + c4 * 4) = o;
}
// dx[n][hw][c] = dy[n][c] / HW
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N,
                                                           int HW, int C) {
  const int cg = C >> 2;
  size_t total = (size_t)N * HW * cg;
  const float inv = (float)HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c4 = (int)(i % cg);
    int n = (int)(i / ((size_t)HW * cg));
    f32x4 g = *reinterpret_cast<const f32x4*>(dy + (size_t)n * C + c4 * 4);
    f32x4 o = {g[0] / inv, g[1] / inv, g[2] / inv, g[3] / inv};
    *reinterpret_cast<f32x4*>(dx + i * 4) = o;
  }
}

// one wave per row: loss_row = logsumexp(x) - x[label]; dlogits = (softmax - onehot) * gscale
// partial[block] = sum of the block's 4 row losses
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ logits,
                                                          const long long* __restrict__ labels,
                                                          float* __restrict__ dlogits, float* __restrict__ partial,
                                                          int N, int C, float gscale) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  float loss = 0.f;
  if (row < N) {
    const float* x = logits + (size_t)row * C;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(x[c] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float lse = logf(s) + m;
    const int lab = (int)labels[row];
    loss = lse - x[lab];
    if (dlogits) {
      float* d = dlogits + (size_t)row * C;
      for (int c = lane; c < C; c += 64) d[c] = (expf(x[c] - lse) - (c == lab ? 1.f : 0.f)) * gscale;
    }
  }
  __shared__ float red[4];
  if (lane == 0) red[w] = loss;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void ce_finalize_kernel(const float* __restrict__ partial, int n, int N, float* __restrict__ loss_out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += (double)partial[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) loss_out[0] = (float)(s / (double)N);
}

// torch.optim.SGD (momentum, dampening 0): g += wd*p; buf = first ? g : mom*buf + g; p -= lr * (nesterov ? g + mom*buf : buf)
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ buf, size_t n, size_t split, float lr_lo,
                                                   float lr_hi, float momentum, float wd_lo, float wd_hi, int first,
                                                   int nesterov, float grad_scale, float clip_value,
                                                   const float* __restrict__ state) {
  float gmul = 1.f;
  if (state) {  // gated form (sacv_step_gate): skip flag, step count and clip multiplier live on the device
    if (state[1] > 0.f) return;
    // no first-step case here: the momentum buffer starts as zeros, and mom * 0 + g == g exactly, which is what
    // torch.optim.SGD's `buf = clone(g)` produces -- while a buffer loaded from a torch SGD checkpoint (whose
    // state has no step count) must NOT be overwritten by the first step after the resume
    first = 0;
    gmul = state[4];
  }
  size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 pp = *reinterpret_cast<f32x4*>(p + i * 4);
    f32x4 gg = *reinterpret_cast<const f32x4*>(g + i * 4);
    f32x4 bb = *reinterpret_cast<f32x4*>(buf + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool lo = (i * 4 + e) < split;
      const float wd = lo ? wd_lo : wd_hi;
      const float lr = lo ? lr_lo : lr_hi;
      float gr = gg[e] * grad_scale;
      if (clip_value > 0.f) gr = fminf(fmaxf(gr, -clip_value), clip_value);
      gr = gr * gmul + wd * pp[e];
      float b = first ? gr : momentum * bb[e] + gr;
      pp[e] = pp[e] - lr * (nesterov ? gr + momentum * b : b);
      bb[e] = b;
    }
    *reinterpret_cast<f32x4*>(p + i * 4) = pp;
    *reinterpret_cast<f32x4*>(buf + i * 4) = bb;
  }
}

SACV_API_BEGIN

int sacv_avgpool_fwd(const float* x, float* out, int N, int HW, int C, void* stream) {
  SACV_CHECK_ARG(x && out && N > 0 && HW > 0 && C % 4 == 0, "avgpool_fwd: bad args");
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(sacv_cdiv((long long)N * (C / 4), 256)), dim3(256), 0,
                     (hipStream_t)stream, x, out, N, HW, C);
  SACV_CHECK_LAUNCH("avgpool_fwd");
  return SACV_OK;
}

int sacv_avgpool_bwd(const float* dy, float* dx, int N, int HW, int C, void* stream) {
  SACV_CHECK_ARG(dy && dx && N > 0 && HW > 0 && C % 4 == 0, "avgpool_bwd: bad args");
  size_t total = (size_t)N * HW * (C / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, dx, N, HW, C);
  SACV_CHECK_LAUNCH("avgpool_bwd");
  return SACV_OK;
}

// loss_out[0] = mean cross-entropy; dlogits (nullable) = d loss / d logits * grad_scale; workspace >= ceil(N/4) floats
int sacv_softmax_ce_fwd_bwd(const float* logits, const long long* labels, float* dlogits, float* loss_out, int N,
                            int C, float grad_scale, float* workspace, void* stream) {
  SACV_CHECK_ARG(logits && labels && loss_out && workspace && N > 0 && C > 0, "softmax_ce: bad args");
  int blocks = (N + 3) / 4;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(softmax_ce_kernel, dim3(blocks), dim3(256), 0, st, logits, labels, dlogits, workspace, N, C,
                     grad_scale / (float)N);
  SACV_CHECK_LAUNCH("softmax_ce");
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, st, workspace, blocks, N, loss_out);
  SACV_CHECK_LAUNCH("ce_finalize");
  return SACV_OK;
}

int sacv_sgd_step(float* params, const float* grads, float* momentum_buf, long long n, long long split, float lr,
                  float momentum, float wd_lo, float wd_hi, int first_step, float grad_scale, void* stream) {
  SACV_CHECK_ARG(params && grads && momentum_buf && n > 0 && n % 4 == 0 && split % 4 == 0 && split <= n,
                 "sgd_step: bad args");
  size_t n4 = (size_t)n / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(sgd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, momentum_buf,
                     (size_t)n, (size_t)split, lr, lr, momentum, wd_lo, wd_hi, first_step, 0, grad_scale, 0.f,
                     (const float*)nullptr);
  SACV_CHECK_LAUNCH("sgd_step");
  return SACV_OK;
}

// as sacv_sgd_step, gated by the device-side state of sacv_step_gate (skip flag; a zero momentum buffer IS the first step,
// clip multiplier), with one learning rate per parameter group and torch's nesterov option
int sacv_sgd_step_gated(float* params, const float* grads, float* momentum_buf, long long n, long long split,
                        float lr_lo, float lr_hi, float momentum, float wd_lo, float wd_hi, int nesterov,
                        const float* state, float grad_scale, float clip_value, void* stream) {
  SACV_CHECK_ARG(params && grads && momentum_buf && state && n > 0 && n % 4 == 0 && split % 4 == 0 && split <= n,
                 "sgd_step_gated: bad args");
  size_t n4 = (size_t)n / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(sgd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, momentum_buf,
                     (size_t)n, (size_t)split, lr_lo, lr_hi, momentum, wd_lo, wd_hi, 0, nesterov, grad_scale,
                     clip_value, state);
  SACV_CHECK_LAUNCH("sgd_step_gated");
  return SACV_OK;
}

SACV_API_END
