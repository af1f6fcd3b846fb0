// MultiHeadAttention core on the bf16 matrix cores (reference: simpleAICV/classification/backbones/
// vit.py:73-86 -- attn = softmax((q @ k^T) * scale); x = attn @ v -- under autocast: bf16 matmuls, fp32 softmax).
//
// One workgroup (4 waves) per (sample, head); the whole head lives in LDS (sequences of the MAE recipes: 50 tokens x
// head dim 64 in the encoder, 197 x 32 in the decoder).  Every product is v_mfma_f32_32x32x16_bf16:
//
//   forward, per block of 32 queries (one wave):   S^T = K Q^T   (keys in the accumulator registers, queries on lanes)
//       softmax over the keys = an in-register reduction + ONE cross-half exchange per query
//       O = P V = (P^T)^T V: the accumulator tile P^T IS the A operand of the next MFMA (its k order is permuted:
//       element j of lane half h is key 16s + 8(j>>2) + 4h + (j&3), so the V fragments are read in that order from a
//       transposed LDS image V^T[dd][key]) -- no LDS round trip for the probabilities.  Saves log-sum-exp per query.
//   backward (recomputes the probabilities from the saved log-sum-exp; deterministic, no atomics), two phases:
//       A  each wave owns key blocks:   S = Q K^T, dP = dO V^T (queries in registers, keys on lanes)
//                                        dV += P^T dO, dK += dS^T Q   (accumulator tile as A operand, dO^T / Q^T images)
//       B  each wave owns query blocks: S^T, dP^T (keys in registers)  ->  dQ += dS K   (K^T image)
//
// Rounding points follow autocast: bf16(bf16(QK^T) * scale), softmax in fp32, probabilities / dS cast to bf16 before
// their products, bf16 outputs.
#include "sacv_common.h"
#include <mutex>

typedef sacv_bf16 bf16;
typedef sacv_bf16x8 bf16x8;
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float bfr(float v) { return (float)(bf16)v; }
// row of the 32x32 accumulator tile held in register r of a lane in half lh
__device__ __forceinline__ int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// rows [0, npad) x D of a strided bf16 matrix -> row-major LDS image (row stride RS bytes), zero rows beyond n
template <int D>
__device__ __forceinline__ void load_rowmajor(unsigned char* dst, int RS, const bf16* src, size_t row_stride, int n,
                                              int npad) {
  constexpr int G = D / 8;
  for (int t = threadIdx.x; t < npad * G; t += blockDim.x) {
    const int r = t / G, g = t - r * G;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r < n) v = *reinterpret_cast<const u32x4*>(src + (size_t)r * row_stride + g * 8);
    *reinterpret_cast<u32x4*>(dst + r * RS + g * 16) = v;
  }
}
// the same rows as a TRANSPOSED image T[dd][row] (row stride TS bytes; 2-byte scatter)
template <int D>
__device__ __forceinline__ void load_transposed(unsigned char* dst, int TS, const bf16* src, size_t row_stride, int n,
                                                int npad) {
  constexpr int G = D / 8;
  for (int t = threadIdx.x; t < npad * G; t += blockDim.x) {
    const int r = t / G, g = t - r * G;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
    if (r < n) v = *reinterpret_cast<const bf16x8*>(src + (size_t)r * row_stride + g * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) *reinterpret_cast<bf16*>(dst + (g * 8 + e) * TS + r * 2) = v[e];
  }
}
// "nt" fragment: rows of a row-major image; lane (row l31, half lh) reads 8 consecutive k
__device__ __forceinline__ bf16x8 frag_rows(const unsigned char* img, int RS, int row0, int kstep, int l31, int lh) {
  return *reinterpret_cast<const bf16x8*>(img + (row0 + l31) * RS + kstep * 32 + lh * 16);
}
// B fragment matching an accumulator tile used as the A operand (k order 16s + 8(j>>2) + 4h + (j&3)) from a transposed
// image T[col][k]: two 8-byte reads
__device__ __forceinline__ bf16x8 frag_perm(const unsigned char* img, int TS, int col, int k0, int s, int lh) {
  const unsigned char* p = img + col * TS + (k0 + 16 * s + 4 * lh) * 2;
  const bf16x4_t lo = *reinterpret_cast<const bf16x4_t*>(p);
  const bf16x4_t hi = *reinterpret_cast<const bf16x4_t*>(p + 16);
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[e] = lo[e];
    v[4 + e] = hi[e];
  }
  return v;
}
// accumulator registers 8s..8s+7 as a bf16 A fragment
__device__ __forceinline__ bf16x8 acc_as_operand(const f32x16& x, int s) {
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (bf16)x[8 * s + j];
  return v;
}

}  // namespace

// ------------------------------------------------------------------------------------------ forward
template <int D, int NKB>
__global__ __launch_bounds__(256) void attention_fwd_mfma_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                                  float* __restrict__ lse, int n, int H, float scale) {
  constexpr int RS = D * 2 + 16, KS = D / 16, NB = D / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int npad = (n + 31) & ~31, nkb = npad >> 5;
  const int TS = npad * 2 + 8;
  unsigned char* Qs = smem;
  unsigned char* Ks = Qs + npad * RS;
  unsigned char* VT = Ks + npad * RS;
  const size_t rs = (size_t)3 * H * D;
  const bf16* base = qkv + (size_t)b * n * rs + (size_t)h * D;
  load_rowmajor<D>(Qs, RS, base, rs, n, npad);
  load_rowmajor<D>(Ks, RS, base + (size_t)H * D, rs, n, npad);
  load_transposed<D>(VT, TS, base + (size_t)2 * H * D, rs, n, npad);
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
  for (int qb = wid; qb < nkb; qb += 4) {
    bf16x8 bq[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) bq[s] = frag_rows(Qs, RS, qb * 32, s, l31, lh);
    f32x16 st[NKB];  // S^T blocks: rows = keys, cols (lanes) = queries
    float m = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if (kb < nkb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
          st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Ks, RS, kb * 32, s, l31, lh), bq[s], st[kb], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb * 32 + acc_row(r, lh);
          const float v = key < n ? bfr(bfr(st[kb][r]) * scale) : -INFINITY;
          st[kb][r] = v;
          m = fmaxf(m, v);
        }
      }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if (kb < nkb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __expf(st[kb][r] - m);
          st[kb][r] = e;
          l += e;
        }
      }
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int q = qb * 32 + l31;
    if (lh == 0 && q < n) lse[((size_t)b * H + h) * n + q] = m + __logf(l);
    f32x16 o[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if (kb < nkb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kb][r] *= inv;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bf16x8 pa = acc_as_operand(st[kb], s);
#pragma unroll
          for (int j = 0; j < NB; ++j)
            o[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, frag_perm(VT, TS, j * 32 + l31, kb * 32, s, lh), o[j], 0,
                                                           0, 0);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qq = qb * 32 + acc_row(r, lh);
        if (qq < n) out[((size_t)b * n + qq) * ((size_t)H * D) + (size_t)h * D + j * 32 + l31] = (bf16)o[j][r];
      }
  }
}

// ------------------------------------------------------------------------------------------ backward
template <int D, int NKB>
__global__ __launch_bounds__(256) void attention_bwd_mfma_kernel(const bf16* __restrict__ qkv,
                                                                  const bf16* __restrict__ out,
                                                                  const bf16* __restrict__ dout,
                                                                  const float* __restrict__ lse,
                                                                  bf16* __restrict__ dqkv, int n, int H, float scale) {
  constexpr int RS = D * 2 + 16, KS = D / 16, NB = D / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int npad = (n + 31) & ~31, nkb = npad >> 5;
  const int TS = npad * 2 + 8;
  unsigned char* Qs = smem;
  unsigned char* Ks = Qs + npad * RS;
  unsigned char* Vs = Ks + npad * RS;
  unsigned char* dOs = Vs + npad * RS;
  unsigned char* QT = dOs + npad * RS;
  unsigned char* KT = QT + D * TS;
  unsigned char* dOT = KT + D * TS;
  float* Ls = reinterpret_cast<float*>(dOT + D * TS);  // [npad] log-sum-exp (+inf beyond n: probability 0)
  float* Ds = Ls + npad;                                // [npad] delta = sum_dd dO * O
  const size_t rs = (size_t)3 * H * D, os = (size_t)H * D;
  const bf16* base = qkv + (size_t)b * n * rs + (size_t)h * D;
  const bf16* obase = out + (size_t)b * n * os + (size_t)h * D;
  const bf16* dobase = dout + (size_t)b * n * os + (size_t)h * D;
  load_rowmajor<D>(Qs, RS, base, rs, n, npad);
  load_rowmajor<D>(Ks, RS, base + (size_t)H * D, rs, n, npad);
  load_rowmajor<D>(Vs, RS, base + (size_t)2 * H * D, rs, n, npad);
  load_rowmajor<D>(dOs, RS, dobase, os, n, npad);
  load_transposed<D>(QT, TS, base, rs, n, npad);
  load_transposed<D>(KT, TS, base + (size_t)H * D, rs, n, npad);
  load_transposed<D>(dOT, TS, dobase, os, n, npad);
  for (int q = threadIdx.x; q < npad; q += blockDim.x) {
    float dl = 0.f;
    if (q < n) {
      for (int g = 0; g < D / 8; ++g) {
        float a[8], c[8];
        VecIO<bf16>::load(obase + (size_t)q * os + g * 8, a);
        VecIO<bf16>::load(dobase + (size_t)q * os + g * 8, c);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += a[e] * c[e];
      }
    }
    Ds[q] = dl;
    Ls[q] = q < n ? lse[((size_t)b * H + h) * n + q] : INFINITY;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;

  // ---- phase A: dV, dK of the key blocks this wave owns.  Tiles S / dP: rows = queries (registers), cols = keys (lanes)
  for (int kb = wid; kb < nkb; kb += 4) {
    bf16x8 bk[KS], bv[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bk[s] = frag_rows(Ks, RS, kb * 32, s, l31, lh);
      bv[s] = frag_rows(Vs, RS, kb * 32, s, l31, lh);
    }
    f32x16 dv[NB], dk[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) dv[j][r] = dk[j][r] = 0.f;
    const bool key_ok = kb * 32 + l31 < n;
    for (int qb = 0; qb < nkb; ++qb) {
      f32x16 sp, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) sp[r] = dp[r] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        sp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Qs, RS, qb * 32, s, l31, lh), bk[s], sp, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(dOs, RS, qb * 32, s, l31, lh), bv[s], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = qb * 32 + acc_row(r, lh);
        const float p = key_ok ? __expf(bfr(bfr(sp[r]) * scale) - Ls[q]) : 0.f;
        const float ds = p * (bfr(dp[r]) - Ds[q]);
        sp[r] = p;                             // -> bf16 in acc_as_operand (attn @ v saw bf16 probabilities)
        dp[r] = bfr(bfr(ds) * scale);          // d(QK^T) as autograd hands it on
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pa = acc_as_operand(sp, s), da = acc_as_operand(dp, s);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          dv[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, frag_perm(dOT, TS, j * 32 + l31, qb * 32, s, lh), dv[j], 0,
                                                          0, 0);
          dk[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, frag_perm(QT, TS, j * 32 + l31, qb * 32, s, lh), dk[j], 0,
                                                          0, 0);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb * 32 + acc_row(r, lh);
        if (key < n) {
          bf16* o = dqkv + ((size_t)b * n + key) * rs + (size_t)h * D + j * 32 + l31;
          o[(size_t)H * D] = (bf16)dk[j][r];
          o[(size_t)2 * H * D] = (bf16)dv[j][r];
        }
      }
  }

  // ---- phase B: dQ of the query blocks this wave owns.  Tiles S^T / dP^T: rows = keys (registers), cols = queries
  for (int qb = wid; qb < nkb; qb += 4) {
    bf16x8 bq[KS], bo[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bq[s] = frag_rows(Qs, RS, qb * 32, s, l31, lh);
      bo[s] = frag_rows(dOs, RS, qb * 32, s, l31, lh);
    }
    const int q = qb * 32 + l31;
    const float Lq = Ls[q], Dq = Ds[q];
    f32x16 dq[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[j][r] = 0.f;
    for (int kb = 0; kb < nkb; ++kb) {
      f32x16 sp, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) sp[r] = dp[r] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        sp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Ks, RS, kb * 32, s, l31, lh), bq[s], sp, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Vs, RS, kb * 32, s, l31, lh), bo[s], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb * 32 + acc_row(r, lh);
        const float p = key < n ? __expf(bfr(bfr(sp[r]) * scale) - Lq) : 0.f;
        dp[r] = bfr(bfr(p * (bfr(dp[r]) - Dq)) * scale);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 da = acc_as_operand(dp, s);
#pragma unroll
        for (int j = 0; j < NB; ++j)
          dq[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, frag_perm(KT, TS, j * 32 + l31, kb * 32, s, lh), dq[j], 0,
                                                          0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qq = qb * 32 + acc_row(r, lh);
        if (qq < n) dqkv[((size_t)b * n + qq) * rs + (size_t)h * D + j * 32 + l31] = (bf16)dq[j][r];
      }
  }
}

static size_t fwd_lds(int n, int d) {
  const int npad = (n + 31) & ~31;
  return (size_t)2 * npad * (d * 2 + 16) + (size_t)d * (npad * 2 + 8);
}
static size_t bwd_lds(int n, int d) {
  const int npad = (n + 31) & ~31;
  return (size_t)4 * npad * (d * 2 + 16) + (size_t)3 * d * (npad * 2 + 8) + (size_t)2 * npad * sizeof(float);
}

template <int D, int NKB>
static int launch_fwd(const void* qkv, void* out, float* lse, int B, int n, int H, float scale, hipStream_t st) {
  const size_t lds = fwd_lds(n, D);
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_fwd_mfma_kernel<D, NKB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  hipLaunchKernelGGL((attention_fwd_mfma_kernel<D, NKB>), dim3(B * H), dim3(256), lds, st, (const bf16*)qkv, (bf16*)out,
                     lse, n, H, scale);
  SACV_CHECK_LAUNCH("attention_fwd_mfma");
  return SACV_OK;
}
template <int D, int NKB>
static int launch_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int n,
                      int H, float scale, hipStream_t st) {
  const size_t lds = bwd_lds(n, D);
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_mfma_kernel<D, NKB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  hipLaunchKernelGGL((attention_bwd_mfma_kernel<D, NKB>), dim3(B * H), dim3(256), lds, st, (const bf16*)qkv,
                     (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, n, H, scale);
  SACV_CHECK_LAUNCH("attention_bwd_mfma");
  return SACV_OK;
}

SACV_API_BEGIN

// 1 when the matrix-core attention kernels cover (n, d): head dim 32 or 64, up to 224 tokens, LDS fit
int sacv_attention_mfma_supported(int n, int d) {
  if (d != 32 && d != 64) return 0;
  if (n < 1 || n > 224) return 0;
  return bwd_lds(n, d) <= 160 * 1024 ? 1 : 0;
}

// qkv bf16 [B][n][3][H][d] -> out bf16 [B][n][H*d], lse fp32 [B][H][n] (log-sum-exp of the scaled scores, saved for
// the backward)
int sacv_attention_fwd_mfma(const void* qkv, void* out, float* lse, int B, int n, int H, int d, float scale,
                            void* stream) {
  SACV_CHECK_ARG(qkv && out && lse && B > 0 && H > 0, "attention_fwd_mfma: bad args");
  SACV_CHECK_ARG(sacv_attention_mfma_supported(n, d), "attention_fwd_mfma: n=%d, d=%d not covered", n, d);
  hipStream_t st = (hipStream_t)stream;
  const int nkb = (n + 31) / 32;
  if (d == 64) return nkb <= 2 ? launch_fwd<64, 2>(qkv, out, lse, B, n, H, scale, st)
                               : launch_fwd<64, 7>(qkv, out, lse, B, n, H, scale, st);
  return nkb <= 2 ? launch_fwd<32, 2>(qkv, out, lse, B, n, H, scale, st)
                  : launch_fwd<32, 7>(qkv, out, lse, B, n, H, scale, st);
}

int sacv_attention_bwd_mfma(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B,
                            int n, int H, int d, float scale, void* stream) {
  SACV_CHECK_ARG(qkv && out && dout && lse && dqkv && B > 0 && H > 0, "attention_bwd_mfma: bad args");
  SACV_CHECK_ARG(sacv_attention_mfma_supported(n, d), "attention_bwd_mfma: n=%d, d=%d not covered", n, d);
  hipStream_t st = (hipStream_t)stream;
  const int nkb = (n + 31) / 32;
  if (d == 64) return nkb <= 2 ? launch_bwd<64, 2>(qkv, out, dout, lse, dqkv, B, n, H, scale, st)
                               : launch_bwd<64, 7>(qkv, out, dout, lse, dqkv, B, n, H, scale, st);
  return nkb <= 2 ? launch_bwd<32, 2>(qkv, out, dout, lse, dqkv, B, n, H, scale, st)
                  : launch_bwd<32, 7>(qkv, out, dout, lse, dqkv, B, n, H, scale, st);
}

SACV_API_END
