// ViT-B/16 MAE pre-training step (BASELINE config #5) -- the kernels around the bf16 MFMA GEMMs.
//
// Reference: simpleAICV/masked_image_modeling/models/vit_mae.py:25-461 (encoder with per-sample random masking,
// :187-215; decoder with mask tokens and un-shuffle, :343-372), simpleAICV/classification/backbones/vit.py:58-160
// (MultiHeadAttention, FeedForward, TransformerEncoderLayer), simpleAICV/masked_image_modeling/losses.py:11-23 (masked
// MSE), run under autocast (use_amp=True in 2.masked_image_modeling_training/imagenet/mae_vit_base_patch16_224/
// train_config.py).  Data flow mirrors autocast's dtypes: the residual stream is fp32 (the fp32 position encoding is
// added to it), LayerNorm reads fp32 and writes bf16, every Linear (patch embedding, qkv, proj, fc1, fc2, encoder_to_
// decoder, pred) is a bf16 MFMA GEMM (conv_bf16.hip, 1x1 problems) whose epilogue adds into the fp32 stream, GELU and
// the attention core work on bf16 tensors with fp32 arithmetic, softmax in fp32.
//
// All of these kernels are HBM- or latency-bound helpers; the FLOPs (98 %) are in the GEMMs.
#include "sacv_common.h"

typedef sacv_bf16 bf16;
typedef sacv_bf16x8 bf16x8;

static int grid_for_v(size_t total, int cap = 65536) {
  size_t b = (total + 255) / 256;
  return (int)(b > (size_t)cap ? cap : (b < 1 ? 1 : b));
}

// ------------------------------------------------------------------------------------------ random masking
// VITMAEPretrainModelEncoder.random_masking (vit_mae.py:187-215): shuffle = argsort(noise) per sample, restore =
// argsort(shuffle), keep = shuffle[:keep_len], mask[i] = restore[i] >= keep_len.  One workgroup per sample: bitonic
// sort of (noise, index) in LDS, index as the tie-break (torch.argsort on distinct values; ties are unpinned).
__global__ __launch_bounds__(256) void mae_masking_kernel(const float* __restrict__ noise, int n, int keep_len,
                                                           long long* __restrict__ shuffle_ids,
                                                           long long* __restrict__ restore_ids,
                                                           long long* __restrict__ keep_ids, float* __restrict__ mask) {
  __shared__ float key[1024];
  __shared__ int idx[1024];
  const int b = blockIdx.x;
  int np = 1;
  while (np < n) np <<= 1;
  for (int i = threadIdx.x; i < np; i += blockDim.x) {
    key[i] = i < n ? noise[(size_t)b * n + i] : INFINITY;
    idx[i] = i < n ? i : 0x7fffffff;
  }
  __syncthreads();
  for (int k = 2; k <= np; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < np; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const bool up = (i & k) == 0;
          const float a = key[i], c = key[l];
          const int ia = idx[i], ic = idx[l];
          const bool gt = (a > c) || (a == c && ia > ic);
          if (gt == up) {
            key[i] = c;
            key[l] = a;
            idx[i] = ic;
            idx[l] = ia;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int r = threadIdx.x; r < n; r += blockDim.x) {
    const int src = idx[r];
    shuffle_ids[(size_t)b * n + r] = src;
    restore_ids[(size_t)b * n + src] = r;
    if (r < keep_len) keep_ids[(size_t)b * keep_len + r] = src;
    mask[(size_t)b * n + src] = r < keep_len ? 0.f : 1.f;
  }
}

// ------------------------------------------------------------------------------------------ patches
// rows[b*keep + j][k], k = (c*P + py)*P + px  (the flattening of the conv weight [E][3][P][P]) of the KEPT patches
// only: PatchEmbeddingBlock (vit.py:20-55) is a P x P / stride-P convolution, i.e. a GEMM over non-overlapping
// patches, and only the visible quarter of them ever reaches the encoder (vit_mae.py:168-176).
__global__ __launch_bounds__(256) void mae_patchify_kernel(const float* __restrict__ img, const long long* __restrict__ ids,
                                                            bf16* __restrict__ rows, int B, int C, int H, int W, int P,
                                                            int keep, long long sN, long long sC, long long sH,
                                                            long long sW) {
  const int K = C * P * P, G = K >> 3, gw = W / P;
  const size_t total = (size_t)B * keep * G;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(t % G);
    const size_t row = t / G;
    const int b = (int)(row / keep);
    const int pid = ids ? (int)ids[row] : (int)(row % keep);
    const int ph = pid / gw, pw = pid - ph * gw;
    const int k0 = g * 8;
    const int c = k0 / (P * P);
    const int rem = k0 - c * P * P;
    const int py = rem / P, px = rem - py * P;  // 8 consecutive px (P % 8 == 0)
    const float* src = img + b * sN + c * sC + (long long)(ph * P + py) * sH + (long long)(pw * P + px) * sW;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[e * sW];
    VecIO<bf16>::store(rows + t * 8, v);
  }
}

// MAESelfSupervisedPretrainCollater's labels (masked_image_modeling/common.py:8-44; images_to_patch layout): [B][L][P*P*C] with
// k = (py*P + px)*C + c, optionally normalised per patch (norm_label, masked_image_modeling/common.py:36-39:
// (x - mean) / (var_unbiased + 1e-4)**0.5)
__global__ __launch_bounds__(256) void mae_patch_labels_kernel(const float* __restrict__ img, float* __restrict__ out,
                                                                int B, int C, int H, int W, int P, int norm,
                                                                long long sN, long long sC, long long sH, long long sW) {
  // one wave per patch
  const int gw = W / P, L = (H / P) * gw, K = P * P * C;
  const int lane = threadIdx.x & 63;
  const size_t patch = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (patch >= (size_t)B * L) return;
  const int b = (int)(patch / L), pid = (int)(patch % L);
  const int ph = pid / gw, pw = pid - ph * gw;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) {
    const int c = k % C, q = k / C;
    const int py = q / P, px = q - py * P;
    s += img[b * sN + c * sC + (long long)(ph * P + py) * sH + (long long)(pw * P + px) * sW];
  }
  s = wave_sum(s);
  s = __shfl(s, 0, 64);
  const float mean = s / (float)K;
  float var = 0.f;
  for (int k = lane; k < K; k += 64) {
    const int c = k % C, q = k / C;
    const int py = q / P, px = q - py * P;
    const float d = img[b * sN + c * sC + (long long)(ph * P + py) * sH + (long long)(pw * P + px) * sW] - mean;
    var += d * d;
  }
  var = wave_sum(var);
  var = __shfl(var, 0, 64) / (float)(K - 1);
  const float sd = sqrtf(var + 1e-4f);
  for (int k = lane; k < K; k += 64) {
    const int c = k % C, q = k / C;
    const int py = q / P, px = q - py * P;
    const float x = img[b * sN + c * sC + (long long)(ph * P + py) * sH + (long long)(pw * P + px) * sW];
    out[patch * K + k] = norm ? (x - mean) / sd : x;
  }
}

// ------------------------------------------------------------------------------------------ token assembly
// encoder input (vit_mae.py:160-180): x[b][0] = cls + pos[0]; x[b][1+j] = emb[b*keep + j] + pos[1 + keep_ids[b][j]]
__global__ __launch_bounds__(256) void mae_encoder_tokens_kernel(const bf16* __restrict__ emb,
                                                                  const float* __restrict__ cls,
                                                                  const float* __restrict__ pos,
                                                                  const long long* __restrict__ keep_ids,
                                                                  float* __restrict__ x, int B, int keep, int E) {
  const int G = E >> 3;
  const size_t total = (size_t)B * (keep + 1) * G;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(t % G);
    const size_t tok = t / G;
    const int b = (int)(tok / (keep + 1)), j = (int)(tok % (keep + 1));
    float v[8], pp[8];
    if (j == 0) {
      ld_param<8>(cls + g * 8, v);
      ld_param<8>(pos + g * 8, pp);
    } else {
      VecIO<bf16>::load(emb + ((size_t)b * keep + (j - 1)) * E + g * 8, v);
      ld_param<8>(pos + (size_t)(1 + keep_ids[(size_t)b * keep + j - 1]) * E + g * 8, pp);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += pp[e];
    *reinterpret_cast<f32x4*>(x + tok * E + g * 8) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(x + tok * E + g * 8 + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
}
// backward: demb[b*keep + j] = bf16(dx[b][1+j]);  dcls partial sums are taken by the column-sum kernel over dx[:,0]
__global__ __launch_bounds__(256) void mae_encoder_tokens_bwd_kernel(const float* __restrict__ dx,
                                                                      bf16* __restrict__ demb, int B, int keep, int E) {
  const int G = E >> 3;
  const size_t total = (size_t)B * keep * G;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(t % G);
    const size_t row = t / G;
    const int b = (int)(row / keep), j = (int)(row % keep);
    float v[8];
    ld_param<8>(dx + ((size_t)b * (keep + 1) + 1 + j) * E + g * 8, v);
    VecIO<bf16>::store(demb + row * E + g * 8, v);
  }
}

// decoder input (vit_mae.py:343-360): y = encoder_to_decoder(x) bf16 [B][keep+1][D];
// x[b][0] = y[b][0] + pos[0];  x[b][1+i] = (restore[b][i] < keep ? y[b][1 + restore[b][i]] : mask_token) + pos[1+i]
__global__ __launch_bounds__(256) void mae_decoder_tokens_kernel(const bf16* __restrict__ y,
                                                                  const float* __restrict__ mask_token,
                                                                  const float* __restrict__ pos,
                                                                  const long long* __restrict__ restore_ids,
                                                                  float* __restrict__ x, int B, int keep, int n, int D) {
  const int G = D >> 3;
  const size_t total = (size_t)B * (n + 1) * G;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(t % G);
    const size_t tok = t / G;
    const int b = (int)(tok / (n + 1)), i = (int)(tok % (n + 1));
    float v[8], pp[8];
    int src = 0;
    if (i > 0) {
      const int r = (int)restore_ids[(size_t)b * n + i - 1];
      src = r < keep ? 1 + r : -1;
    }
    if (src >= 0)
      VecIO<bf16>::load(y + ((size_t)b * (keep + 1) + src) * D + g * 8, v);
    else
      ld_param<8>(mask_token + g * 8, v);
    ld_param<8>(pos + (size_t)i * D + g * 8, pp);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += pp[e];
    *reinterpret_cast<f32x4*>(x + tok * D + g * 8) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(x + tok * D + g * 8 + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
}
// backward: dy[b][0] = dx[b][0]; dy[b][1+r] = dx[b][1 + shuffle[b][r]] (r < keep);
// gmask[b*(n-keep) + (r-keep)] = dx[b][1 + shuffle[b][r]] (r >= keep): rows whose column sum is d mask_token
__global__ __launch_bounds__(256) void mae_decoder_tokens_bwd_kernel(const float* __restrict__ dx,
                                                                      const long long* __restrict__ shuffle_ids,
                                                                      bf16* __restrict__ dy, float* __restrict__ gmask,
                                                                      int B, int keep, int n, int D) {
  const int G = D >> 3;
  const size_t total = (size_t)B * (n + 1) * G;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(t % G);
    const size_t q = t / G;
    const int b = (int)(q / (n + 1)), r = (int)(q % (n + 1)) - 1;  // r = -1: the cls row
    const int srow = r < 0 ? 0 : 1 + (int)shuffle_ids[(size_t)b * n + r];
    float v[8];
    ld_param<8>(dx + ((size_t)b * (n + 1) + srow) * D + g * 8, v);
    if (r < keep) {
      VecIO<bf16>::store(dy + ((size_t)b * (keep + 1) + 1 + r) * D + g * 8, v);
    } else {
      float* o = gmask + ((size_t)b * (n - keep) + (r - keep)) * D + g * 8;
      *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
  }
}

// ------------------------------------------------------------------------------------------ LayerNorm
// nn.LayerNorm(C, eps) over the last dim of the fp32 residual stream; one wave per row; y bf16 (the next Linear's input)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, bf16* __restrict__ y,
                                                             float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                             int R, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float* xr = x + (size_t)row * C;
  const int G = C >> 3;
  float s = 0.f;
  for (int g = lane; g < G; g += 64) {
    float v[8];
    ld_param<8>(xr + g * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e];
  }
  s = wave_sum(s);
  const float mean = __shfl(s, 0, 64) / (float)C;
  float q = 0.f;
  for (int g = lane; g < G; g += 64) {
    float v[8];
    ld_param<8>(xr + g * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) q += (v[e] - mean) * (v[e] - mean);
  }
  q = wave_sum(q);
  const float rstd = 1.0f / sqrtf(__shfl(q, 0, 64) / (float)C + eps);
  for (int g = lane; g < G; g += 64) {
    float v[8], ga[8], be[8];
    ld_param<8>(xr + g * 8, v);
    ld_param<8>(gamma + g * 8, ga);
    ld_param<8>(beta + g * 8, be);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (v[e] - mean) * rstd * ga[e] + be[e];
    VecIO<bf16>::store(y + (size_t)row * C + g * 8, v);
  }
  if (lane == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
}
// dx (+)= rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma;  partial[blk][0][c] = sum dy, [1][c] = sum dy*xhat
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, float* __restrict__ dx,
                                                             float* __restrict__ partial, int R, int C,
                                                             int rows_per_block, int accumulate) {
  extern __shared__ float sm[];  // [4 waves][2][C]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int G = C >> 3;
  const int r_begin = blockIdx.x * rows_per_block;
  const int r_end = min(R, r_begin + rows_per_block);
  constexpr int MAXG = 6;  // C <= 64 * 8 * MAXG = 3072
  float sdy[MAXG][8], sdx[MAXG][8];
#pragma unroll
  for (int i = 0; i < MAXG; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) sdy[i][e] = sdx[i][e] = 0.f;
  for (int row = r_begin + w; row < r_end; row += 4) {
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXG; ++i) {
      const int g = lane + 64 * i;
      if (g < G) {
        float d[8], v[8], ga[8];
        VecIO<bf16>::load(dy + (size_t)row * C + g * 8, d);
        ld_param<8>(x + (size_t)row * C + g * 8, v);
        ld_param<8>(gamma + g * 8, ga);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (v[e] - mu) * rs;
          const float gg = d[e] * ga[e];
          s1 += gg;
          s2 += gg * xh;
          sdy[i][e] += d[e];
          sdx[i][e] += d[e] * xh;
        }
      }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const float m1 = __shfl(s1, 0, 64) / (float)C, m2 = __shfl(s2, 0, 64) / (float)C;
#pragma unroll
    for (int i = 0; i < MAXG; ++i) {
      const int g = lane + 64 * i;
      if (g < G) {
        float d[8], v[8], ga[8], o[8];
        VecIO<bf16>::load(dy + (size_t)row * C + g * 8, d);
        ld_param<8>(x + (size_t)row * C + g * 8, v);
        ld_param<8>(gamma + g * 8, ga);
        float* dst = dx + (size_t)row * C + g * 8;
        if (accumulate) ld_param<8>(dst, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (v[e] - mu) * rs;
          const float r = rs * (d[e] * ga[e] - m1 - xh * m2);
          o[e] = accumulate ? o[e] + r : r;
        }
        *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{o[4], o[5], o[6], o[7]};
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXG; ++i) {
    const int g = lane + 64 * i;
    if (g < G) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sm[(size_t)(w * 2 + 0) * C + g * 8 + e] = sdy[i][e];
        sm[(size_t)(w * 2 + 1) * C + g * 8 + e] = sdx[i][e];
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += 256) {
    const int which = c / C, col = c - which * C;
    float acc = 0.f;
    for (int l = 0; l < 4; ++l) acc += sm[(size_t)(l * 2 + which) * C + col];
    partial[((size_t)blockIdx.x * 2 + which) * C + col] = acc;
  }
}

// ------------------------------------------------------------------------------------------ GELU (exact, erf)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    float v[8];
    VecIO<bf16>::load(x + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));
    VecIO<bf16>::store(y + i * 8, v);
  }
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                        bf16* __restrict__ dx, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    float v[8], d[8];
    VecIO<bf16>::load(x + i * 8, v);
    VecIO<bf16>::load(dy + i * 8, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float cdf = 0.5f * (1.0f + erff(v[e] * 0.70710678118654752f));
      const float pdf = 0.3989422804014327f * __expf(-0.5f * v[e] * v[e]);
      d[e] = d[e] * (cdf + v[e] * pdf);
    }
    VecIO<bf16>::store(dx + i * 8, d);
  }
}

// column sums of a bf16 [R][C] matrix (bias gradients): per-block partials for sacv_colsum_finalize
__global__ __launch_bounds__(256) void colsum_bf16_partial_kernel(const bf16* __restrict__ x, float* __restrict__ partial,
                                                                   int R, int C, int rows_per_block) {
  extern __shared__ float sm[];  // [row lanes][C]
  const int G = C >> 3;
  const int r_begin = blockIdx.x * rows_per_block;
  const int r_end = min(R, r_begin + rows_per_block);
  const int rl = G >= 256 ? 1 : 256 / G;
  for (int g0 = 0; g0 < G; g0 += 256) {  // G > 256 (C > 2048): column slabs
    const int g = g0 + (int)(threadIdx.x % (G >= 256 ? 256 : G));
    const int lane = G >= 256 ? 0 : threadIdx.x / G;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    if (g < G && lane < rl) {
      for (int r = r_begin + lane; r < r_end; r += rl) {
        float v[8];
        VecIO<bf16>::load(x + (size_t)r * C + g * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += v[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) sm[(size_t)lane * C + g * 8 + e] = s[e];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc = 0.f;
    for (int l = 0; l < rl; ++l) acc += sm[(size_t)l * C + c];
    partial[(size_t)blockIdx.x * C + c] = acc;
  }
}

// ------------------------------------------------------------------------------------------ attention core
// MultiHeadAttention.forward between qkv and proj (vit.py:73-86) for one (sample, head) per workgroup:
//   S = bf16(bf16(Q K^T) * scale); P = softmax_fp32(S); O = bf16(bf16(P) V)      (autocast's rounding points)
// qkv bf16 [B][n][3][H][d]; out bf16 [B][n][H*d].  Sequences are short (50 / 197 tokens, head dim 64 / 32): every
// operand of a (sample, head) lives in LDS as fp32 and the small products run as 4x4 register micro-tiles on the
// vector ALUs (2 % of the step's FLOPs; the matrix cores carry the Linear layers).  Queries are processed in chunks
// of QC rows so that K, V (and in the backward dK, dV) stay resident.
#define ATT_QC 16
__device__ __forceinline__ float bfr(float v) { return (float)(bf16)v; }

// C[m][n] (ldc) (+)= alpha * sum_k A(m,k) B(k,n);  A(m,k) = A[m*am + k*ak], B(k,n) = B[k*bk + n*bn]; M, N multiples of 4
// within zero-padded LDS arrays.  ROUND: result rounded to bf16 (a bf16 matmul output).
template <bool ACC, bool ROUND>
__device__ __forceinline__ void lds_gemm(float* C, int ldc, const float* A, int am, int ak, const float* B, int bk, int bn,
                                         int M, int N, int K, float alpha) {
  // thread -> rows m0..m0+3 and columns c, c+tn, c+2tn, c+3tn: consecutive threads touch consecutive columns, so B
  // reads (stride bn = 1 or an odd row stride) and C accesses are bank-conflict free; the A reads mostly broadcast
  const int tn = N >> 2, tiles = (M >> 2) * tn;
  for (int t = threadIdx.x; t < tiles; t += blockDim.x) {
    const int m0 = (t / tn) << 2, c = t % tn;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    // K % 4 == 0 (head dim, ATT_QC and the padded sequence length are multiples of 4): four k steps of LDS reads are
    // in flight before their FMAs -- with one or two waves per SIMD the loop is bound by LDS latency, not bandwidth
    for (int k0 = 0; k0 < K; k0 += 4) {
      float a[4][4], b[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[u][i] = A[(m0 + i) * am + (k0 + u) * ak];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[u][j] = B[(k0 + u) * bk + (c + j * tn) * bn];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] += a[u][i] * b[u][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = acc[i][j];
        if (ROUND) v = bfr(v);
        v *= alpha;
        float* o = C + (m0 + i) * ldc + c + j * tn;
        *o = ACC ? *o + v : v;
      }
  }
}

__device__ __forceinline__ void att_load_rows(float* dst, int ld, const bf16* src, size_t row_stride, int r0, int rows,
                                              int rows_pad, int d) {
  // dst[r][0..d) = src[(r0 + r) * row_stride + 0..d) for r < rows, zero rows up to rows_pad
  const int G = d >> 3;
  for (int t = threadIdx.x; t < rows_pad * G; t += blockDim.x) {
    const int r = t / G, g = t - r * G;
    float v[8];
    if (r < rows)
      VecIO<bf16>::load(src + (size_t)(r0 + r) * row_stride + g * 8, v);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[r * ld + g * 8 + e] = v[e];
  }
}

// softmax of S[qc][n] rows in place (fp32), one wave per row; padded columns [n, npad) are set to 0
__device__ __forceinline__ void att_softmax_rows(float* S, int ld, int rows, int n, int npad) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int r = w; r < rows; r += nw) {
    float* s = S + r * ld;
    float mx = -INFINITY;
    for (int k = lane; k < n; k += 64) mx = fmaxf(mx, s[k]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int k = lane; k < n; k += 64) {
      const float e = __expf(s[k] - mx);
      s[k] = e;
      sum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = 1.0f / sum;
    for (int k = lane; k < npad; k += 64) s[k] = k < n ? s[k] * inv : 0.f;
  }
}

__global__ __launch_bounds__(512) void attention_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out, int n,
                                                             int H, int d, float scale) {
  extern __shared__ float sm[];
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int npad = (n + 3) & ~3, ldk = d + 1, lds = npad + 1;
  float* Ks = sm;                    // [npad][ldk]
  float* Vs = Ks + npad * ldk;       // [npad][ldk]
  float* Qs = Vs + npad * ldk;       // [QC][ldk]
  float* Ss = Qs + ATT_QC * ldk;     // [QC][lds]
  float* Os = Ss + ATT_QC * lds;     // [QC][ldk]
  const size_t rs = (size_t)3 * H * d;
  const bf16* base = qkv + (size_t)b * n * rs + (size_t)h * d;
  att_load_rows(Ks, ldk, base + (size_t)H * d, rs, 0, n, npad, d);
  att_load_rows(Vs, ldk, base + (size_t)2 * H * d, rs, 0, n, npad, d);
  for (int q0 = 0; q0 < n; q0 += ATT_QC) {
    const int rows = min(ATT_QC, n - q0);
    __syncthreads();
    att_load_rows(Qs, ldk, base, rs, q0, rows, ATT_QC, d);
    __syncthreads();
    lds_gemm<false, true>(Ss, lds, Qs, ldk, 1, Ks, 1, ldk, ATT_QC, npad, d, scale);  // bf16(QK^T) * scale
    __syncthreads();
    for (int t = threadIdx.x; t < rows * n; t += blockDim.x) {  // the product with `scale` is a bf16 tensor too
      float* s = Ss + (t / n) * lds + t % n;
      *s = bfr(*s);
    }
    __syncthreads();
    att_softmax_rows(Ss, lds, rows, n, npad);
    __syncthreads();
    for (int t = threadIdx.x; t < rows * n; t += blockDim.x) {  // attn @ v casts the probabilities to bf16
      float* s = Ss + (t / n) * lds + t % n;
      *s = bfr(*s);
    }
    __syncthreads();
    lds_gemm<false, false>(Os, ldk, Ss, lds, 1, Vs, ldk, 1, ATT_QC, d, npad, 1.f);
    __syncthreads();
    const int G = d >> 3;
    for (int t = threadIdx.x; t < rows * G; t += blockDim.x) {
      const int r = t / G, g = t - r * G;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = Os[r * ldk + g * 8 + e];
      VecIO<bf16>::store(out + ((size_t)b * n + q0 + r) * ((size_t)H * d) + (size_t)h * d + g * 8, v);
    }
  }
}

// backward of the same block: recomputes S / P per query chunk; dqkv bf16 [B][n][3][H][d]
//   dV = bf16(P_b^T dO); dP = bf16(dO V^T); dS = P (dP - sum_k dP P) (fp32 softmax backward); d(QK^T) = bf16(bf16(dS) *
//   scale); dQ = bf16(d(QK^T) K); dK = bf16(d(QK^T)^T Q)
__global__ __launch_bounds__(512) void attention_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dout,
                                                             bf16* __restrict__ dqkv, int n, int H, int d, float scale) {
  extern __shared__ float sm[];
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int npad = (n + 3) & ~3, ldk = d + 1, lds = npad + 1;
  float* Ks = sm;                      // [npad][ldk]
  float* Vs = Ks + npad * ldk;
  float* dKs = Vs + npad * ldk;
  float* dVs = dKs + npad * ldk;
  float* Qs = dVs + npad * ldk;        // [QC][ldk]
  float* dOs = Qs + ATT_QC * ldk;
  float* dQs = dOs + ATT_QC * ldk;
  float* Ps = dQs + ATT_QC * ldk;      // [QC][lds] fp32 probabilities
  float* Gs = Ps + ATT_QC * lds;       // [QC][lds] dP, then d(QK^T)
  const size_t rs = (size_t)3 * H * d;
  const bf16* base = qkv + (size_t)b * n * rs + (size_t)h * d;
  att_load_rows(Ks, ldk, base + (size_t)H * d, rs, 0, n, npad, d);
  att_load_rows(Vs, ldk, base + (size_t)2 * H * d, rs, 0, n, npad, d);
  for (int t = threadIdx.x; t < 2 * npad * ldk; t += blockDim.x) dKs[t] = 0.f;  // dKs and dVs are adjacent
  const bf16* dobase = dout + (size_t)b * n * ((size_t)H * d) + (size_t)h * d;
  for (int q0 = 0; q0 < n; q0 += ATT_QC) {
    const int rows = min(ATT_QC, n - q0);
    __syncthreads();
    att_load_rows(Qs, ldk, base, rs, q0, rows, ATT_QC, d);
    att_load_rows(dOs, ldk, dobase, (size_t)H * d, q0, rows, ATT_QC, d);
    __syncthreads();
    lds_gemm<false, true>(Ps, lds, Qs, ldk, 1, Ks, 1, ldk, ATT_QC, npad, d, scale);
    lds_gemm<false, true>(Gs, lds, dOs, ldk, 1, Vs, 1, ldk, ATT_QC, npad, d, 1.f);  // dP = bf16(dO V^T)
    __syncthreads();
    for (int t = threadIdx.x; t < rows * n; t += blockDim.x) {
      float* s = Ps + (t / n) * lds + t % n;
      *s = bfr(*s);
    }
    __syncthreads();
    att_softmax_rows(Ps, lds, rows, n, npad);  // P fp32 (rows >= `rows` hold stale data: their dO rows are zero)
    __syncthreads();
    // softmax backward needs the fp32 P and dP: dS = P * (dP - rowsum(dP * P))
    {
      const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
      for (int r = w; r < ATT_QC; r += (int)(blockDim.x >> 6)) {
        float* p = Ps + r * lds;
        float* g = Gs + r * lds;
        float dot = 0.f;
        if (r < rows)
          for (int k = lane; k < n; k += 64) dot += g[k] * p[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
        for (int k = lane; k < npad; k += 64) {
          const float pk = (r < rows && k < n) ? p[k] : 0.f;
          const float ds = pk * (g[k] - dot);
          g[k] = (r < rows && k < n) ? bfr(bfr(ds) * scale) : 0.f;  // d(QK^T) as the bf16 tensors autograd passes
          p[k] = bfr(pk);                                           // P as attn @ v saw it
        }
      }
    }
    __syncthreads();
    lds_gemm<true, false>(dVs, ldk, Ps, 1, lds, dOs, ldk, 1, npad, d, ATT_QC, 1.f);   // dV += P_b^T dO
    lds_gemm<true, false>(dKs, ldk, Gs, 1, lds, Qs, ldk, 1, npad, d, ATT_QC, 1.f);    // dK += dS^T Q
    lds_gemm<false, false>(dQs, ldk, Gs, lds, 1, Ks, ldk, 1, ATT_QC, d, npad, 1.f);   // dQ = dS K
    __syncthreads();
    const int G = d >> 3;
    for (int t = threadIdx.x; t < rows * G; t += blockDim.x) {
      const int r = t / G, g = t - r * G;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = dQs[r * ldk + g * 8 + e];
      VecIO<bf16>::store(dqkv + ((size_t)b * n + q0 + r) * rs + (size_t)h * d + g * 8, v);
    }
  }
  __syncthreads();
  const int G = d >> 3;
  for (int t = threadIdx.x; t < n * G; t += blockDim.x) {
    const int r = t / G, g = t - r * G;
    float vk[8], vv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      vk[e] = dKs[r * ldk + g * 8 + e];
      vv[e] = dVs[r * ldk + g * 8 + e];
    }
    bf16* o = dqkv + ((size_t)b * n + r) * rs + (size_t)h * d + g * 8;
    VecIO<bf16>::store(o + (size_t)H * d, vk);
    VecIO<bf16>::store(o + (size_t)2 * H * d, vv);
  }
}

// ------------------------------------------------------------------------------------------ masked MSE
// MSELoss (masked_image_modeling/losses.py:11-23): loss = sum_p mask_p * mean_k (pred - label)^2 / (sum mask + 1e-4);
// pred bf16 [B][n+1][K] (row 0 = cls, dropped, vit_mae.py:369-370), label fp32 [B][n][K]; dpred bf16 (row 0 zero).
__global__ __launch_bounds__(256) void mae_mse_kernel(const bf16* __restrict__ pred, const float* __restrict__ label,
                                                       const float* __restrict__ mask, bf16* __restrict__ dpred,
                                                       float* __restrict__ partial, int B, int n, int K, float gscale) {
  // one wave per patch row; partial[block] = sum over its 4 rows of mask * mean_k diff^2
  __shared__ float red[4];
  __shared__ float msum_s;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x < 64) {  // sum(mask): every block recomputes it (B*n <= a few 10^5 floats, L2-resident)
    float s = 0.f;
    for (int i = lane; i < B * n; i += 64) s += mask[i];
    s = wave_sum(s);
    if (lane == 0) msum_s = s;
  }
  __syncthreads();
  const float denom = msum_s + 1e-4f;
  const size_t row = (size_t)blockIdx.x * 4 + w;
  float acc = 0.f;
  if (row < (size_t)B * (n + 1)) {
    const int b = (int)(row / (n + 1)), i = (int)(row % (n + 1));
    const int G = K >> 3;
    if (i == 0) {
      if (dpred)
        for (int g = lane; g < G; g += 64) {
          float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          VecIO<bf16>::store(dpred + row * K + g * 8, z);
        }
    } else {
      const float m = mask[(size_t)b * n + i - 1];
      const float* lab = label + ((size_t)b * n + i - 1) * K;
      const float gs = gscale * m * 2.0f / ((float)K * denom);
      for (int g = lane; g < G; g += 64) {
        float p[8], l[8];
        VecIO<bf16>::load(pred + row * K + g * 8, p);
        ld_param<8>(lab + g * 8, l);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float df = p[e] - l[e];
          acc += df * df;
          p[e] = df * gs;
        }
        if (dpred) VecIO<bf16>::store(dpred + row * K + g * 8, p);
      }
      acc = wave_sum(acc) * m / (float)K;
    }
  }
  if (lane == 0) red[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void mae_mse_finalize_kernel(const float* __restrict__ partial, int nblocks, const float* __restrict__ mask,
                                        int nmask, float* __restrict__ loss) {
  double s = 0.0, m = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 64) s += (double)partial[i];
  for (int i = threadIdx.x; i < nmask; i += 64) m += (double)mask[i];
  s = wave_sum_d(s);
  m = wave_sum_d(m);
  if (threadIdx.x == 0) loss[0] = (float)(s / ((double)(float)m + 1e-4));
}

SACV_API_BEGIN

int sacv_colsum_finalize(const float* partial, int nblocks, int stride, int C, float* out, int accumulate,
                         void* stream);

int sacv_mae_random_masking(const float* noise, int B, int n, int keep_len, long long* shuffle_ids,
                            long long* restore_ids, long long* keep_ids, float* mask, void* stream) {
  SACV_CHECK_ARG(noise && shuffle_ids && restore_ids && keep_ids && mask, "mae_random_masking: null pointer");
  SACV_CHECK_ARG(B > 0 && n > 0 && n <= 1024 && keep_len >= 0 && keep_len <= n, "mae_random_masking: bad sizes");
  hipLaunchKernelGGL(mae_masking_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, noise, n, keep_len, shuffle_ids,
                     restore_ids, keep_ids, mask);
  SACV_CHECK_LAUNCH("mae_masking");
  return SACV_OK;
}

int sacv_mae_patchify(const float* img, const long long* ids, void* rows, int B, int C, int H, int W, int P, int keep,
                      long long sN, long long sC, long long sH, long long sW, void* stream) {
  SACV_CHECK_ARG(img && rows && P % 8 == 0 && H % P == 0 && W % P == 0 && keep > 0, "mae_patchify: bad args");
  const size_t total = (size_t)B * keep * (C * P * P / 8);
  hipLaunchKernelGGL(mae_patchify_kernel, dim3(grid_for_v(total)), dim3(256), 0, (hipStream_t)stream, img, ids,
                     (bf16*)rows, B, C, H, W, P, keep, sN, sC, sH, sW);
  SACV_CHECK_LAUNCH("mae_patchify");
  return SACV_OK;
}

int sacv_mae_patch_labels(const float* img, float* out, int B, int C, int H, int W, int P, int norm, long long sN,
                          long long sC, long long sH, long long sW, void* stream) {
  SACV_CHECK_ARG(img && out && H % P == 0 && W % P == 0, "mae_patch_labels: bad args");
  const size_t patches = (size_t)B * (H / P) * (W / P);
  hipLaunchKernelGGL(mae_patch_labels_kernel, dim3((unsigned)((patches + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     img, out, B, C, H, W, P, norm, sN, sC, sH, sW);
  SACV_CHECK_LAUNCH("mae_patch_labels");
  return SACV_OK;
}

int sacv_mae_encoder_tokens(const void* emb, const float* cls, const float* pos, const long long* keep_ids, float* x,
                            int B, int keep, int E, void* stream) {
  SACV_CHECK_ARG(emb && cls && pos && keep_ids && x && E % 8 == 0, "mae_encoder_tokens: bad args");
  hipLaunchKernelGGL(mae_encoder_tokens_kernel, dim3(grid_for_v((size_t)B * (keep + 1) * (E / 8))), dim3(256), 0,
                     (hipStream_t)stream, (const bf16*)emb, cls, pos, keep_ids, x, B, keep, E);
  SACV_CHECK_LAUNCH("mae_encoder_tokens");
  return SACV_OK;
}

int sacv_mae_encoder_tokens_bwd(const float* dx, void* demb, int B, int keep, int E, void* stream) {
  SACV_CHECK_ARG(dx && demb && E % 8 == 0, "mae_encoder_tokens_bwd: bad args");
  hipLaunchKernelGGL(mae_encoder_tokens_bwd_kernel, dim3(grid_for_v((size_t)B * keep * (E / 8))), dim3(256), 0,
                     (hipStream_t)stream, dx, (bf16*)demb, B, keep, E);
  SACV_CHECK_LAUNCH("mae_encoder_tokens_bwd");
  return SACV_OK;
}

int sacv_mae_decoder_tokens(const void* y, const float* mask_token, const float* pos, const long long* restore_ids,
                            float* x, int B, int keep, int n, int D, void* stream) {
  SACV_CHECK_ARG(y && mask_token && pos && restore_ids && x && D % 8 == 0, "mae_decoder_tokens: bad args");
  hipLaunchKernelGGL(mae_decoder_tokens_kernel, dim3(grid_for_v((size_t)B * (n + 1) * (D / 8))), dim3(256), 0,
                     (hipStream_t)stream, (const bf16*)y, mask_token, pos, restore_ids, x, B, keep, n, D);
  SACV_CHECK_LAUNCH("mae_decoder_tokens");
  return SACV_OK;
}

int sacv_mae_decoder_tokens_bwd(const float* dx, const long long* shuffle_ids, void* dy, float* gmask, int B, int keep,
                                int n, int D, void* stream) {
  SACV_CHECK_ARG(dx && shuffle_ids && dy && gmask && D % 8 == 0, "mae_decoder_tokens_bwd: bad args");
  hipLaunchKernelGGL(mae_decoder_tokens_bwd_kernel, dim3(grid_for_v((size_t)B * (n + 1) * (D / 8))), dim3(256), 0,
                     (hipStream_t)stream, dx, shuffle_ids, (bf16*)dy, gmask, B, keep, n, D);
  SACV_CHECK_LAUNCH("mae_decoder_tokens_bwd");
  return SACV_OK;
}

int sacv_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int R,
                       int C, float eps, void* stream) {
  SACV_CHECK_ARG(x && gamma && beta && y && mean && rstd && R > 0 && C % 8 == 0, "layernorm_fwd: bad args");
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, beta,
                     (bf16*)y, mean, rstd, R, C, eps);
  SACV_CHECK_LAUNCH("layernorm_fwd");
  return SACV_OK;
}

// rows of the LayerNorm-backward partial buffer [rows][2C] actually written for R input rows
int sacv_layernorm_bwd_partial_rows(int R) {
  int nb = R < 1024 ? (R + 15) / 16 : 512;
  int rpb = (R + nb - 1) / nb;
  rpb = (rpb + 3) / 4 * 4;
  return (R + rpb - 1) / rpb;
}

long long sacv_layernorm_bwd_workspace_floats(int R, int C) {
  const int nb = R < 1024 ? (R + 15) / 16 : 512;
  return (long long)nb * 2 * C;
}

// dx (+)= LayerNorm backward; dgamma / dbeta (+)= their sums (fixed-order two-stage reduction)
int sacv_layernorm_bwd(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                       float* dx, float* dgamma, float* dbeta, int R, int C, int accumulate_dx, int accumulate_params,
                       float* workspace, void* stream) {
  SACV_CHECK_ARG(dy && x && mean && rstd && gamma && dx && workspace, "layernorm_bwd: null pointer");
  SACV_CHECK_ARG((dgamma != nullptr) == (dbeta != nullptr), "layernorm_bwd: dgamma and dbeta are given or deferred together");
  SACV_CHECK_ARG(R > 0 && C % 8 == 0 && C <= 3072, "layernorm_bwd: C=%d unsupported", C);
  int nb = R < 1024 ? (R + 15) / 16 : 512;
  int rpb = (R + nb - 1) / nb;
  rpb = (rpb + 3) / 4 * 4;
  nb = (R + rpb - 1) / rpb;  // = sacv_layernorm_bwd_partial_rows(R)
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nb), dim3(256), (size_t)8 * C * sizeof(float), st, (const bf16*)dy, x,
                     mean, rstd, gamma, dx, workspace, R, C, rpb, accumulate_dx);
  SACV_CHECK_LAUNCH("layernorm_bwd");
  if (!dgamma) return SACV_OK;  // deferred: the caller sums workspace [rows][2C] (dbeta | dgamma) later, e.g. with
                                // sacv_colsum_finalize_multi over all parameter gradients of a transformer block
  int rc = sacv_colsum_finalize(workspace, nb, 2 * C, C, dbeta, accumulate_params, stream);
  if (rc != SACV_OK) return rc;
  return sacv_colsum_finalize(workspace + C, nb, 2 * C, C, dgamma, accumulate_params, stream);
}

int sacv_gelu_fwd(const void* x, void* y, long long n, void* stream) {
  SACV_CHECK_ARG(x && y && n % 8 == 0, "gelu_fwd: bad args");
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for_v((size_t)n / 8, 8192)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)x, (bf16*)y, (size_t)n / 8);
  SACV_CHECK_LAUNCH("gelu_fwd");
  return SACV_OK;
}

int sacv_gelu_bwd(const void* dy, const void* x, void* dx, long long n, void* stream) {
  SACV_CHECK_ARG(dy && x && dx && n % 8 == 0, "gelu_bwd: bad args");
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for_v((size_t)n / 8, 8192)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)dy, (const bf16*)x, (bf16*)dx, (size_t)n / 8);
  SACV_CHECK_LAUNCH("gelu_bwd");
  return SACV_OK;
}

long long sacv_colsum_bf16_workspace_floats(int R, int C) {
  const int nb = R < 256 ? 1 : (R < 4096 ? 64 : 512);
  return (long long)nb * C;
}

// rows of the partial buffer [rows][C] actually written for R input rows
int sacv_colsum_bf16_partial_rows(int R) {
  int nb = R < 256 ? 1 : (R < 4096 ? 64 : 512);
  const int rpb = (R + nb - 1) / nb;
  return (R + rpb - 1) / rpb;
}

// out == NULL: first stage only (partials in workspace [sacv_colsum_bf16_partial_rows(R)][C]), finalized by the caller
int sacv_colsum_bf16(const void* x, float* out, int R, int C, int accumulate, float* workspace, void* stream) {
  SACV_CHECK_ARG(x && workspace && R > 0 && C % 8 == 0, "colsum_bf16: bad args");
  int nb = R < 256 ? 1 : (R < 4096 ? 64 : 512);
  int rpb = (R + nb - 1) / nb;
  nb = (R + rpb - 1) / rpb;
  const int G = C / 8;
  const int rl = G >= 256 ? 1 : 256 / G;
  hipLaunchKernelGGL(colsum_bf16_partial_kernel, dim3(nb), dim3(256), (size_t)rl * C * sizeof(float),
                     (hipStream_t)stream, (const bf16*)x, workspace, R, C, rpb);
  SACV_CHECK_LAUNCH("colsum_bf16_partial");
  if (!out) return SACV_OK;
  return sacv_colsum_finalize(workspace, nb, C, C, out, accumulate, stream);
}

static size_t att_lds_bytes(int n, int d, bool bwd) {
  const int npad = (n + 3) & ~3, ldk = d + 1, lds = npad + 1;
  size_t f = bwd ? (size_t)4 * npad * ldk + 3 * ATT_QC * ldk + 2 * ATT_QC * lds
                 : (size_t)2 * npad * ldk + 2 * ATT_QC * ldk + ATT_QC * lds;
  return f * sizeof(float);
}

int sacv_attention_fwd(const void* qkv, void* out, int B, int n, int H, int d, float scale, void* stream) {
  SACV_CHECK_ARG(qkv && out && B > 0 && n > 0 && H > 0 && d % 8 == 0, "attention_fwd: bad args");
  const size_t lds = att_lds_bytes(n, d, false);
  SACV_CHECK_ARG(lds <= 160 * 1024, "attention_fwd: sequence %d x head dim %d does not fit the LDS", n, d);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_fwd_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(attention_fwd_kernel, dim3(B * H), dim3(512), lds, (hipStream_t)stream, (const bf16*)qkv,
                     (bf16*)out, n, H, d, scale);
  SACV_CHECK_LAUNCH("attention_fwd");
  return SACV_OK;
}

int sacv_attention_bwd(const void* qkv, const void* dout, void* dqkv, int B, int n, int H, int d, float scale,
                       void* stream) {
  SACV_CHECK_ARG(qkv && dout && dqkv && B > 0 && n > 0 && H > 0 && d % 8 == 0, "attention_bwd: bad args");
  const size_t lds = att_lds_bytes(n, d, true);
  SACV_CHECK_ARG(lds <= 160 * 1024, "attention_bwd: sequence %d x head dim %d does not fit the LDS", n, d);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(attention_bwd_kernel, dim3(B * H), dim3(512), lds, (hipStream_t)stream, (const bf16*)qkv,
                     (const bf16*)dout, (bf16*)dqkv, n, H, d, scale);
  SACV_CHECK_LAUNCH("attention_bwd");
  return SACV_OK;
}

long long sacv_mae_mse_workspace_floats(int B, int n) { return ((long long)B * (n + 1) + 3) / 4; }

// loss_out[0] = MSELoss(pred[:, 1:], label, mask); dpred (nullable) = d loss / d pred * grad_scale
int sacv_mae_mse_loss(const void* pred, const float* label, const float* mask, void* dpred, float* loss_out, int B,
                      int n, int K, float grad_scale, float* workspace, void* stream) {
  SACV_CHECK_ARG(pred && label && mask && loss_out && workspace && K % 8 == 0, "mae_mse_loss: bad args");
  const int blocks = (int)(((size_t)B * (n + 1) + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mae_mse_kernel, dim3(blocks), dim3(256), 0, st, (const bf16*)pred, label, mask, (bf16*)dpred,
                     workspace, B, n, K, grad_scale);
  SACV_CHECK_LAUNCH("mae_mse");
  hipLaunchKernelGGL(mae_mse_finalize_kernel, dim3(1), dim3(64), 0, st, workspace, blocks, mask, B * n, loss_out);
  SACV_CHECK_LAUNCH("mae_mse_finalize");
  return SACV_OK;
}

SACV_API_END
