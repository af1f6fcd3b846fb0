// bf16 NT GEMM for the stride-1 1x1 convolutions and the ViT Linear layers on gfx950 (MI355X):
//   C[m][n] = sum_k A[m][k] * B[n][k]   A = activations [M][K] bf16 (rows contiguous), B = weight panel [N][K] bf16
// (forward: B = the layer's own weight; data gradient: B = the transposed panel [Cin][Cout], A = dy).  The reference runs
// these as nn.Conv2d 1x1 / nn.Linear under autocast (simpleAICV/classification/backbones/resnet.py:33-39,
// simpleAICV/masked_image_modeling/models/vit_mae.py:60-120).
//
// conv_nt_bf16_kernel (conv_bf16.hip) covers every convolution shape with register-staged 16-byte gathers, one K tile
// in flight and a barrier every 16 MFMAs per wave; for plain GEMMs that machinery is the bound (350-520 TFLOP/s).  This
// kernel is the lean form for contiguous rows, built like wino_gemm_kernel (winograd.hip):
//   * 256 x 128 output tile per workgroup, 8 waves (4 x 2) of 64 x 64 (2 x 2 blocks of v_mfma_f32_32x32x16_bf16),
//     one workgroup per CU (arithmetic intensity 85 flop per L2 byte instead of 64);
//   * K tile = 64 bf16 = one 128-byte line per row, brought in by LDS-DMA (global_load_lds_dwordx4: no staging
//     registers, no ds_write) into a ring of three 48 KB stages, two tiles in flight, counted vmcnt + raw s_barrier;
//   * unpadded rows, XOR-swizzled on the source address with key (row >> 1) & 7 -- exact for the 16-lane groups a
//     ds_read_b128 is served in (MI355X_MICROARCH.md, LDS);
//   * rows beyond M / N are clamped to the last valid row (the loads carry no masks), their results are not stored;
//   * epilogue as conv_nt_bf16_kernel: fp32 stage in LDS, + bias, (+ dst), round to bf16 or keep fp32, 16-byte stores,
//     column sums of the rounded values (CF_STATS).
// Persistent over tiles, XCD-contiguous tile order, deterministic.
#include "sacv_common.h"
#include <mutex>

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define GCF_BIAS 1
#define GCF_ACCUM 8
#define GCF_STATS 32
#define GCF_OUT_F32 64

typedef __attribute__((address_space(3))) void* glds_t;
__device__ __forceinline__ void glds16b(const void* src, glds_t dst) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
#else
  (void)src;
  (void)dst;
#endif
}

struct GemmArgsH {
  const bf16* A;  // [M][K]
  const bf16* B;  // [N][K]
  void* C;        // [M][N] bf16, or fp32 with GCF_OUT_F32
  const float* bias;
  const float* residual;  // optional fp32 [M][N] added in the epilogue
  float* stats;  // [ceil(M/128)][2][N]: the 128-row granule of conv_nt_bf16_kernel's CF_STATS buffer
  int M, N, K;
  int flags;
  int ntiles;
  SacvBnBwdH bn;  // bn.y non-null: C is the upstream gradient of this BatchNorm layer -- `stats` receives its backward
                  // first-stage sums (sum g, sum g * xhat) per 128-row granule (conv_bf16.hip: CFH_BNBWD)
};

template <int BM, int BN>
__global__ __launch_bounds__(512, 2) void gemm_nt_bf16_dma_kernel(const GemmArgsH a) {
  constexpr int NT = 512, NW = 8;
  constexpr int STAGE = (BM + BN) * 128;  // bytes per stage: rows of 64 bf16
  constexpr int WTM = BM / 4, WTN = BN / 2;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;  // 8-row DMA pieces per wave and tile
  static_assert(PA + PB == 6, "the counted waits are written for 6 DMA instructions per wave and tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int KT = a.K >> 6;
  const int ntn = (a.N + BN - 1) / BN;
  const int per = (a.ntiles + 7) >> 3;
  for (int it = blockIdx.x; it < per * 8; it += gridDim.x) {
    const int tile = (it & 7) * per + (it >> 3);  // XCD-contiguous order (workgroup b runs on XCD b % 8)
    if (tile >= a.ntiles) continue;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));  // keep the thread-index arithmetic inside the tile loop (winograd.hip)
    const int lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid >> 1) * WTM, wn0 = (wid & 1) * WTN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int nt = tile % ntn, mt = tile / ntn;
    const int m0 = mt * BM, n0 = nt * BN;

    __syncthreads();  // the previous tile's epilogue has left the LDS

    // byte offsets of this lane's 16 bytes inside a K tile of A / B
    const int prow = lane >> 3;
    unsigned aoff[PA], boff[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int row = (wid * PA + i) * 8 + prow;
      const int m = min(m0 + row, a.M - 1);
      aoff[i] = (unsigned)m * (unsigned)a.K * 2u + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int row = (wid * PB + i) * 8 + prow;
      const int n = min(n0 + row, a.N - 1);
      boff[i] = (unsigned)n * (unsigned)a.K * 2u + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
    const unsigned char* Ab = reinterpret_cast<const unsigned char*>(a.A);
    const unsigned char* Bb = reinterpret_cast<const unsigned char*>(a.B);
    auto issue = [&](int kt, int st) {
      const unsigned char* pa = Ab + (size_t)kt * 128;
      const unsigned char* pb = Bb + (size_t)kt * 128;
      unsigned char* base = smem + st * STAGE;
#pragma unroll
      for (int i = 0; i < PA; ++i) glds16b(pa + aoff[i], (glds_t)(base + (wid * PA + i) * 1024));
#pragma unroll
      for (int i = 0; i < PB; ++i) glds16b(pb + boff[i], (glds_t)(base + BM * 128 + (wid * PB + i) * 1024));
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int x7 = (l31 >> 1) & 7;
    issue(0, 0);
    if (KT > 1) {
      issue(1, 1);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int kt = 0; kt < KT; ++kt) {
      const bool more = kt + 2 < KT;
      int st2 = cur + 2;
      if (st2 >= 3) st2 -= 3;
      const unsigned char* cA = smem + cur * STAGE + (wm0 + l31) * 128;
      const unsigned char* cB = smem + cur * STAGE + BM * 128 + (wn0 + l31) * 128;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int o = ((kk * 2 + lh) ^ x7) << 4;
        bf16x8 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(cA + i * 32 * 128 + o);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(cB + j * 32 * 128 + o);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        // the DMA of tile kt + 2 goes out behind the first quarter's MFMAs (its stage was last read in iteration kt - 1)
        if (kk == 0 && more) issue(kt + 2, st2);
      }
      if (more)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // tile kt + 1 has landed, tile kt + 2 stays in flight
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur = cur == 2 ? 0 : cur + 1;
    }

    // ---- epilogue: accumulators -> LDS [BM][BN+4] fp32 -> rows of 8 channels.  The old values of an accumulating
    // bf16 destination are requested for all of the thread's rows BEFORE the accumulators are staged, the fp32 residual
    // rows of a 128-row half before that half is processed: one exposed HBM latency per tile / per half instead of one
    // per row (conv_bf16.hip has the measurement).
    constexpr int CS = BN + 4;
    constexpr int CG = BN / 8, RS = NT / CG;
    constexpr int NH = BM / 128, RPH = 128 / RS;  // 128-row halves (the CF_STATS granule), rows per thread and half
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int c8 = tid_e % CG, rr = tid_e / CG;
    const int n = n0 + c8 * 8;
    const bool nok = n < a.N;  // N % 8 == 0
    const int flags = a.flags;
    const bool acc_h = (flags & GCF_ACCUM) && !(flags & GCF_OUT_F32);
    bf16x8 oldh[NH][RPH];
    if (acc_h) {
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int i = 0; i < RPH; ++i) {
          const int m = m0 + h * 128 + rr + i * RS;
          if (nok && m < a.M)
            oldh[h][i] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(a.C) + (size_t)m * a.N + n);
        }
    }
    float* stage = reinterpret_cast<float*>(smem);
    {
      const int e_lane = tid_e & 63, e_wid = tid_e >> 6;
      const int e_wm0 = (e_wid >> 1) * WTM, e_wn0 = (e_wid & 1) * WTN, e_l31 = e_lane & 31, e_lh = e_lane >> 5;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            stage[(e_wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * e_lh) * CS + e_wn0 + j * 32 + e_l31] = acc[i][j][r];
    }
    __syncthreads();
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = ((flags & GCF_BIAS) && nok) ? a.bias[n + e] : 0.f;
    const bool bnb = a.bn.y != nullptr;
    const bool bn_hasz = bnb && a.bn.z != nullptr, bn_rc = bnb && a.bn.z == nullptr && a.bn.beta != nullptr;
    float bmu[8], bis[8], bga[8], bbe[8];
    if (bnb) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        bmu[e] = nok ? a.bn.mean[n + e] : 0.f;
        bis[e] = nok ? a.bn.invstd[n + e] : 0.f;
        bga[e] = (bn_rc && nok) ? a.bn.gamma[n + e] : 0.f;
        bbe[e] = (bn_rc && nok) ? a.bn.beta[n + e] : 0.f;
      }
    }
    // CF_STATS granule is 128 rows (two per tile): rows rr, rr + RS, ... of the first half, then of the second half
    float ssum[2][8], ssq[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 8; ++e) ssum[h][e] = ssq[h][e] = 0.f;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      f32x4 res[RPH][2];
      if (a.residual) {
#pragma unroll
        for (int i = 0; i < RPH; ++i) {
          const int m = m0 + h * 128 + rr + i * RS;
          if (nok && m < a.M) {
            res[i][0] = *reinterpret_cast<const f32x4*>(a.residual + (size_t)m * a.N + n);
            res[i][1] = *reinterpret_cast<const f32x4*>(a.residual + (size_t)m * a.N + n + 4);
          }
        }
      }
      bf16x8 bny[RPH], bnz[RPH];  // the half's rows of y (and z) of the BatchNorm layer, all requested before the first is used
      if (bnb) {
#pragma unroll
        for (int i = 0; i < RPH; ++i) {
          const int m = m0 + h * 128 + rr + i * RS;
          if (nok && m < a.M) {
            bny[i] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(a.bn.y) + (size_t)m * a.N + n);
            if (bn_hasz)
              bnz[i] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(a.bn.z) + (size_t)m * a.N + n);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < RPH; ++i) {
        const int row = h * 128 + rr + i * RS;
        const int m = m0 + row;
        if (!(nok && m < a.M)) continue;
        float v[8];
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(stage + row * CS + c8 * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(stage + row * CS + c8 * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = v0[e] + bv[e];
          v[4 + e] = v1[e] + bv[4 + e];
        }
        const size_t o = (size_t)m * a.N + n;
        if (a.residual) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] += res[i][0][e];
            v[4 + e] += res[i][1][e];
          }
        }
        if (flags & GCF_OUT_F32) {
          float* d = reinterpret_cast<float*>(a.C) + o;
          if (flags & GCF_ACCUM) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += d[e];
          }
          *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
          bf16* d = reinterpret_cast<bf16*>(a.C) + o;
          if (acc_h) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)oldh[h][i][e];
          }
          bf16x8 q;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            q[e] = (bf16)v[e];
            v[e] = (float)q[e];
          }
          *reinterpret_cast<bf16x8*>(d) = q;
        }
        if (bnb) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xh = ((float)bny[i][e] - bmu[e]) * bis[e];
            float g = v[e];
            if (bn_hasz)
              g = (float)bnz[i][e] > 0.f ? g : 0.f;
            else if (bn_rc)
              g = (float)(bf16)(xh * bga[e] + bbe[e]) > 0.f ? g : 0.f;
            ssum[h][e] += g;
            ssq[h][e] += g * xh;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            ssum[h][e] += v[e];
            ssq[h][e] += v[e] * v[e];
          }
        }
      }
    }
    if ((flags & GCF_STATS) || bnb) {
      __syncthreads();  // the stage is dead
      float* red = reinterpret_cast<float*>(smem);  // [2 halves][2][RS][BN]
#pragma unroll
      for (int h = 0; h < BM / 128; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          red[((size_t)(h * 2 + 0) * RS + rr) * BN + c8 * 8 + e] = ssum[h][e];
          red[((size_t)(h * 2 + 1) * RS + rr) * BN + c8 * 8 + e] = ssq[h][e];
        }
      __syncthreads();
      for (int c = tid_e; c < (BM / 128) * 2 * BN; c += NT) {
        const int hw = c / BN, col = c - hw * BN;  // hw = half * 2 + which
        const int h = hw >> 1, which = hw & 1;
        float s = 0.f;
        for (int l = 0; l < RS; ++l) s += red[((size_t)hw * RS + l) * BN + col];
        const int srow = mt * (BM / 128) + h;
        if (n0 + col < a.N && srow * 128 < a.M) a.stats[((size_t)srow * 2 + which) * a.N + n0 + col] = s;
      }
    }
  }
}

static int gemm_num_cus() {
  static int cus = 256;
  static std::once_flag once;
  std::call_once(once, [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
  });
  return cus;
}

template __global__ void gemm_nt_bf16_dma_kernel<256, 128>(const GemmArgsH);

// 1 when the plain-GEMM kernel takes this problem (conv_bf16.hip asks before it falls back to the gather kernel).
// Measured against conv_nt_bf16_kernel<128,128> (scripts/bench_gemm_bf16.py): +4...14 % from ~300 tiles up (MAE
// 6400 x 768 -> 2304 / 3072, 25216 x 512 -> 512 / 1536 / 2048, 25216 x 2048 -> 512; ResNet50 50176 x 1024 -> 256,
// 200704 x 128 -> 512), but 0.65x at 150-200 tiles (one 256 x 128 tile per CU leaves CUs idle where two 128 x 128
// tiles per CU do not) and 0.8x for N = 64.  Both kernels sit at 0.2-0.24 of the bf16 matrix peak: with 16 MFMAs of 32
// cycles per wave and K tile, the ~100 cycles a vector-memory instruction takes to issue (in-kernel stamps of
// wino_gemm_kernel) make the 48 loads of a K tile cost as much as its MFMAs -- the lever is fewer load
// instructions per MFMA (256 x 256 tiles), not the staging primitive.
int sacv_internal_gemm_nt_bf16_supported(long long M, int N, int K) {
#ifdef SACV_NO_DMA_GEMM
  return 0;
#endif
  if (K % 64 != 0 || K < 128 || N % 8 != 0 || N < 128) return 0;
  if (M * K >= (1LL << 30) || (long long)N * K >= (1LL << 30)) return 0;  // 32-bit byte offsets
  const long long tiles = ((M + 255) / 256) * ((N + 127) / 128);
  return tiles >= 300;
}

int sacv_internal_gemm_nt_bf16(const void* A, const void* B, void* C, const float* bias, const float* residual,
                               float* stats, long long M, int N, int K, int flags, void* stream,
                               const SacvBnBwdH* bn) {
  GemmArgsH a;
  a.A = (const bf16*)A;
  a.B = (const bf16*)B;
  a.C = C;
  a.bias = bias;
  a.residual = residual;
  a.stats = stats;
  a.M = (int)M;
  a.N = N;
  a.K = K;
  a.flags = flags;
  a.bn = bn ? *bn : SacvBnBwdH{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  constexpr int BM = 256, BN = 128;
  a.ntiles = (int)(((M + BM - 1) / BM) * ((N + BN - 1) / BN));
  const int lds = 3 * (BM + BN) * 128;
  static_assert(3 * (256 + 128) * 128 >= 256 * (128 + 4) * 4, "the epilogue stage must fit in the operand ring");
  static std::once_flag attr;
  std::call_once(attr, [lds] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_bf16_dma_kernel<BM, BN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  int grid = gemm_num_cus();
  if (grid > a.ntiles) grid = a.ntiles;
  hipLaunchKernelGGL((gemm_nt_bf16_dma_kernel<BM, BN>), dim3(grid), dim3(512), lds, (hipStream_t)stream, a);
  SACV_CHECK_LAUNCH("gemm_nt_bf16_dma_kernel");
  return SACV_OK;
}
