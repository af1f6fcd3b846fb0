// Fused AdamW over the flat fp32 parameter arena (one launch for all 195 tensors).
// Semantics of torch.optim.AdamW as built by the reference (tools/utils.py:480-582, AdamW
// lr 1e-4, betas (0.9, 0.999), eps 1e-8, weight decay 1e-3 on ndim>1 tensors and 0 on 1-D
// tensors): decoupled decay p *= 1 - lr*wd, then p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps).
// The arena is laid out [1-D tensors | >=2-D tensors]; `split` is the element index of the
// boundary so the two parameter groups are one kernel.  HBM-bound: 16 B read + 12 B written
// per element, float4 streams.
#include "sacv_common.h"

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                     float* __restrict__ m, float* __restrict__ v, size_t n,
                                                     size_t split, float lr, float beta1, float beta2, float eps,
                                                     float wd_lo, float wd_hi, float bc1, float bc2_sqrt,
                                                     float grad_scale) {
  size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 pp = *reinterpret_cast<f32x4*>(p + i * 4);
    f32x4 gg = *reinterpret_cast<const f32x4*>(g + i * 4);
    f32x4 mm = *reinterpret_cast<f32x4*>(m + i * 4);
    f32x4 vv = *reinterpret_cast<f32x4*>(v + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float wd = (i * 4 + e) < split ? wd_lo : wd_hi;
      float gr = gg[e] * grad_scale;
      float x = pp[e] * (1.f - lr * wd);
      float m1 = beta1 * mm[e] + (1.f - beta1) * gr;
      float v1 = beta2 * vv[e] + (1.f - beta2) * gr * gr;
      float denom = sqrtf(v1) / bc2_sqrt + eps;
      x = x - (lr / bc1) * (m1 / denom);
      pp[e] = x;
      mm[e] = m1;
      vv[e] = v1;
    }
    *reinterpret_cast<f32x4*>(p + i * 4) = pp;
    *reinterpret_cast<f32x4*>(m + i * 4) = mm;
    *reinterpret_cast<f32x4*>(v + i * 4) = vv;
  }
}

// ---- step gate of the training loop (tools/scripts.py:852-1050), evaluated on the device so a step needs no host
// round trip.  state = float[8]:
//   [0] optimizer steps taken            [1] skip this step (0/1)
//   [2] 1 - beta1^t                      [3] sqrt(1 - beta2^t)
//   [4] gradient multiplier of clip_grad_norm_ (1 when off)   [5] total gradient norm (diagnostic)   [6],[7] reserved
//   nonfinite_flag : flag[0] = 1 when any element is inf/nan          (images / annotations guard, :878-884;
//                    nan_only = the reference's skip_inf_nan_grad test, which checks isnan twice, :951-961)
//   loss_skip_flag : flag[0] |= sum(values) == 0 or a value is not finite                     (:918-930)
//   [all-reduce SUM of flag over the ranks]                                                    (:963-966)
//   step_gate      : updates state[0..3]
//   grad_clip_coef : clip_grad_value_ + clip_grad_norm_ (:978-990) as a clamp inside the optimizer kernel and one
//                    global multiplier state[4] = min(1, max_norm / (||clamp(g * grad_scale)|| + 1e-6))
//   *_step_gated   : the optimizer kernels read `state` and return at once when the step is skipped
__global__ __launch_bounds__(256) void nonfinite_flag_kernel(const float* __restrict__ x, size_t n, int nan_only,
                                                              float* __restrict__ flag) {
  bool bad = false;
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) bad = bad || (nan_only ? (v[e] != v[e]) : !(fabsf(v[e]) <= 3.402823466e38f));
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
    float v = x[n4 * 4 + threadIdx.x];
    bad = bad || (nan_only ? (v != v) : !(fabsf(v) <= 3.402823466e38f));
  }
  if (bad) flag[0] = 1.f;  // every writer stores the same value
}
__global__ void loss_skip_flag_kernel(const float* __restrict__ values, int n, float* __restrict__ flag,
                                      int accumulate) {
  float s = 0.f;
  bool bad = false;
  for (int i = 0; i < n; ++i) {
    float v = values[i];
    bad = bad || !(fabsf(v) <= 3.402823466e38f);  // inf or nan
    s += v;
  }
  bad = bad || !(fabsf(s) <= 3.402823466e38f) || s == 0.f;
  if (accumulate) bad = bad || flag[0] > 0.f;
  flag[0] = bad ? 1.f : 0.f;
}
__global__ void step_gate_kernel(const float* __restrict__ flag, float* __restrict__ state, float beta1, float beta2) {
  const bool skip = flag[0] > 0.f;
  state[1] = skip ? 1.f : 0.f;
  state[4] = 1.f;
  if (!skip) {
    const float t = state[0] + 1.f;
    state[0] = t;
    state[2] = (float)(1.0 - pow((double)beta1, (double)t));
    state[3] = (float)sqrt(1.0 - pow((double)beta2, (double)t));
  }
}
// sum of squares of clamp(g * grad_scale, +-clip_value): per-block partials in double, fixed-order second stage
__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float* __restrict__ g, size_t n, float grad_scale,
                                                          float clip_value, double* __restrict__ partial) {
  __shared__ double red[4];
  double acc = 0.0;
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 v = *reinterpret_cast<const f32x4*>(g + i * 4);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = v[e] * grad_scale;
      if (clip_value > 0.f) x = fminf(fmaxf(x, -clip_value), clip_value);
      s += x * x;
    }
    acc += (double)s;
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void grad_clip_coef_kernel(const double* __restrict__ partial, int nblocks, float max_norm,
                                      float* __restrict__ state) {
  __shared__ double red[64];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 64) acc += partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 64; ++i) t += red[i];
    const float norm = (float)sqrt(t);
    state[5] = norm;
    float coef = 1.f;
    if (max_norm > 0.f) coef = fminf(max_norm / (norm + 1e-6f), 1.f);  // torch.nn.utils.clip_grad_norm_
    state[4] = coef;
  }
}
__global__ __launch_bounds__(256) void adamw_gated_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v, size_t n,
                                                           size_t split, float lr_lo, float lr_hi, float beta1,
                                                           float beta2, float eps, float wd_lo, float wd_hi,
                                                           const float* __restrict__ state, float grad_scale,
                                                           float clip_value) {
  if (state[1] > 0.f) return;
  const float bc1 = state[2], bc2_sqrt = state[3], gmul = state[4];
  size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 pp = *reinterpret_cast<f32x4*>(p + i * 4);
    f32x4 gg = *reinterpret_cast<const f32x4*>(g + i * 4);
    f32x4 mm = *reinterpret_cast<f32x4*>(m + i * 4);
    f32x4 vv = *reinterpret_cast<f32x4*>(v + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool lo = (i * 4 + e) < split;
      const float wd = lo ? wd_lo : wd_hi;
      const float lr = lo ? lr_lo : lr_hi;
      float gr = gg[e] * grad_scale;
      if (clip_value > 0.f) gr = fminf(fmaxf(gr, -clip_value), clip_value);
      gr *= gmul;
      float x = pp[e] * (1.f - lr * wd);
      float m1 = beta1 * mm[e] + (1.f - beta1) * gr;
      float v1 = beta2 * vv[e] + (1.f - beta2) * gr * gr;
      float denom = sqrtf(v1) / bc2_sqrt + eps;
      x = x - (lr / bc1) * (m1 / denom);
      pp[e] = x;
      mm[e] = m1;
      vv[e] = v1;
    }
    *reinterpret_cast<f32x4*>(p + i * 4) = pp;
    *reinterpret_cast<f32x4*>(m + i * 4) = mm;
    *reinterpret_cast<f32x4*>(v + i * 4) = vv;
  }
}

__global__ void fill_kernel(float* __restrict__ p, size_t n, float value) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = value;
}

// EmaModel.update (tools/utils.py:156-163): ema = decay * ema + (1 - decay) * model, elementwise over a flat block, in
// torch's arithmetic: both scalars arrive already rounded to fp32 (torch rounds the Python floats `decay` and
// `1. - decay` to the tensor's dtype), two multiplications and one addition, each rounded (-ffp-contract=off).
__global__ __launch_bounds__(256) void ema_update_kernel(float* __restrict__ ema, const float* __restrict__ model,
                                                          size_t n, float decay, float one_minus_decay,
                                                          const float* __restrict__ state) {
  if (state && state[1] > 0.f) return;  // the step was skipped (sacv_step_gate): the reference `continue`s before its update
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 e = *reinterpret_cast<const f32x4*>(ema + i * 4), m = *reinterpret_cast<const f32x4*>(model + i * 4);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = decay * e[k] + one_minus_decay * m[k];
    *reinterpret_cast<f32x4*>(ema + i * 4) = o;
  }
  const size_t t = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) ema[t] = decay * ema[t] + one_minus_decay * model[t];
}

SACV_API_BEGIN

// n must be a multiple of 4 (the arena pads every tensor to 4 floats); split likewise.
int sacv_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                    long long split, float lr, float beta1, float beta2, float eps, float wd_lo, float wd_hi,
                    int step, float grad_scale, void* stream) {
  SACV_CHECK_ARG(params && grads && exp_avg && exp_avg_sq, "adamw_step: null pointer");
  SACV_CHECK_ARG(n > 0 && n % 4 == 0 && split % 4 == 0 && split >= 0 && split <= n, "adamw_step: bad sizes");
  SACV_CHECK_ARG(step >= 1, "adamw_step: step counts from 1");
  double bc1 = 1.0 - pow((double)beta1, (double)step);
  double bc2 = 1.0 - pow((double)beta2, (double)step);
  size_t n4 = (size_t)n / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                     exp_avg_sq, (size_t)n, (size_t)split, lr, beta1, beta2, eps, wd_lo, wd_hi, (float)bc1,
                     (float)sqrt(bc2), grad_scale);
  SACV_CHECK_LAUNCH("adamw_step");
  return SACV_OK;
}

// flag[0] = 1 when x holds an inf/nan (nan_only: a nan).  `flag` must have been zeroed (sacv_fill) or written by an
// earlier flag kernel: this one only ever raises it.
int sacv_nonfinite_flag(const float* x, long long n, int nan_only, float* flag, void* stream) {
  SACV_CHECK_ARG(x && flag && n >= 0, "nonfinite_flag: bad args");
  SACV_CHECK_ARG(((uintptr_t)x & 15) == 0, "nonfinite_flag: x must be 16-byte aligned");
  if (n == 0) return SACV_OK;
  int blocks = (int)(((size_t)n / 4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(nonfinite_flag_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, nan_only,
                     flag);
  SACV_CHECK_LAUNCH("nonfinite_flag");
  return SACV_OK;
}

int sacv_loss_skip_flag(const float* values, int n, float* flag, int accumulate, void* stream) {
  SACV_CHECK_ARG(values && flag && n > 0, "loss_skip_flag: bad args");
  hipLaunchKernelGGL(loss_skip_flag_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, values, n, flag, accumulate);
  SACV_CHECK_LAUNCH("loss_skip_flag");
  return SACV_OK;
}

// state: float[8], see the layout above; flag[0] > 0 = skip
int sacv_step_gate(const float* flag, float* state, float beta1, float beta2, void* stream) {
  SACV_CHECK_ARG(flag && state, "step_gate: null pointer");
  hipLaunchKernelGGL(step_gate_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, state, beta1, beta2);
  SACV_CHECK_LAUNCH("step_gate");
  return SACV_OK;
}

long long sacv_grad_clip_workspace_bytes(void) { return 1024 * (long long)sizeof(double); }

// state[5] = || clamp(grads * grad_scale, +-clip_value) ||_2 over the whole arena (clip_value <= 0: no clamp),
// state[4] = min(1, max_norm / (norm + 1e-6)) (max_norm <= 0: 1).  Call after sacv_step_gate, before *_step_gated.
int sacv_grad_clip_coef(const float* grads, long long n, float grad_scale, float clip_value, float max_norm,
                        float* state, void* workspace, void* stream) {
  SACV_CHECK_ARG(grads && state && workspace && n > 0 && n % 4 == 0, "grad_clip_coef: bad args");
  hipStream_t st = (hipStream_t)stream;
  int blocks = (int)(((size_t)n / 4 + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(blocks), dim3(256), 0, st, grads, (size_t)n, grad_scale, clip_value,
                     (double*)workspace);
  SACV_CHECK_LAUNCH("grad_sumsq");
  hipLaunchKernelGGL(grad_clip_coef_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, blocks, max_norm,
                     state);
  SACV_CHECK_LAUNCH("grad_clip_coef");
  return SACV_OK;
}

int sacv_adamw_step_gated(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                          long long split, float lr_lo, float lr_hi, float beta1, float beta2, float eps, float wd_lo,
                          float wd_hi, const float* state, float grad_scale, float clip_value, void* stream) {
  SACV_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && state, "adamw_step_gated: null pointer");
  SACV_CHECK_ARG(n > 0 && n % 4 == 0 && split % 4 == 0 && split >= 0 && split <= n, "adamw_step_gated: bad sizes");
  size_t n4 = (size_t)n / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(adamw_gated_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                     exp_avg_sq, (size_t)n, (size_t)split, lr_lo, lr_hi, beta1, beta2, eps, wd_lo, wd_hi, state,
                     grad_scale, clip_value);
  SACV_CHECK_LAUNCH("adamw_step_gated");
  return SACV_OK;
}

int sacv_fill(float* p, long long n, float value, void* stream) {
  SACV_CHECK_ARG(p && n >= 0, "fill: bad args");
  if (n == 0) return SACV_OK;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, (size_t)n, value);
  SACV_CHECK_LAUNCH("fill");
  return SACV_OK;
}

int sacv_ema_update(float* ema, const float* model, long long n, float decay, float one_minus_decay, const float* state,
                    void* stream) {
  SACV_CHECK_ARG(ema && model && n > 0, "ema_update: bad args");
  SACV_CHECK_ARG((((size_t)ema | (size_t)model) & 15) == 0, "ema_update: blocks must be 16-byte aligned");
  const long long blocks = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(ema_update_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks))), dim3(256), 0,
                     (hipStream_t)stream, ema, model, (size_t)n, decay, one_minus_decay, state);
  SACV_CHECK_LAUNCH("ema_update");
  return SACV_OK;
}

SACV_API_END
