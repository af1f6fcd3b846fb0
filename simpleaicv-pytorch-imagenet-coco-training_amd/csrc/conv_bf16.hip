// bf16 implicit-GEMM convolution engine for gfx950 (MI355X): v_mfma_f32_32x32x16_bf16, fp32 accumulate, bf16 NHWC
// activations in HBM, fp32 master weights cast to bf16 panels once per step (the autocast recipe of
// 0.classification_training/imagenet/resnet50/train_config.py:97, tools/scripts.py:108-267; nn.Conv2d call sites
// simpleAICV/classification/backbones/resnet.py:33-39).
//
// Same decomposition as the fp32 engine (conv_gemm.hip):
//   conv_nt_bf16_kernel  C[m][n] = sum_k A(m,k) * B[n][k]   forward (A = im2col gather of x) and data gradient
//                        (A = transposed gather of dy), B = weight panel [N][taps*Cs] bf16
//   conv_tn_bf16_kernel  dW[co][tap,ci] = sum_pix dy[pix][co] * x[pix@tap][ci], split over pixels into fp32 slabs
//                        that the fp32 engine's wgrad_reduce sums in a fixed order into the fp32 gradient arena
// but in bf16 these convolutions sit at or below the HBM/L2 ridge (AI 60-570 flop/B against 310), so the kernels are
// built to stream: 16-byte gathers (8 channels), K tile = 64 channels = one 128-byte line per pixel row, LDS rows
// padded to 144 bytes (conflict-free ds_write_b128 / ds_read_b128), one barrier per K tile, persistent round-robin
// tiles.  Weight-gradient fragments are read with the hardware transposing LDS read (ds_read_b64_tr_b16): both
// operands arrive pixel-major ([pix][channel]) from HBM and the MFMA wants channel-major k-runs.
#include "sacv_common.h"
#include <mutex>

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CF_BIAS 1
#define CF_ACCUM 8
#define CF_STATS 32
#define CFH_BNBWD 256  // internal (sacv_conv2d_gemm_bf16_bnbwd): the stored value is the upstream gradient of a BatchNorm
#define CFH_OUT_F32 64  // dst is fp32 (the classifier's logits) instead of bf16

// 16 zero bytes that masked gathers read (see conv_gemm.hip: a select on the ADDRESS keeps the gathered registers
// untouched until the LDS store)
static __device__ __attribute__((aligned(16))) float sacv_zero16[4] = {0.f, 0.f, 0.f, 0.f};

struct ConvArgsH {
  const bf16* src;
  void* dst;
  int Nb, Hs, Ws;  // source tensor dims
  int Hd, Wd;      // M-space spatial dims
  int M;
  int remap_s, Hr, Wr, remap_oh, remap_ow;
  const bf16* w;
  const float* bias;
  int Cs, N, K;
  int KH, KW, stride_log2, pad_h, pad_w;
  int kh0, khs, kw0, kws, KWr;
  int flags;
  int ntiles;
  float* stats;  // [mtiles][2][N]
  const float* residual;  // optional fp32 [M][N] added in the epilogue (the ViT residual stream), never remapped
  SacvBnBwdH bn;          // CFH_BNBWD: the BatchNorm layer this launch stores the upstream gradient of (dst rows = its rows)
};

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }

// OCC / SWZ as conv_nt_kernel (conv_gemm.hip): SWZ = unpadded 128-byte rows, chunks XOR-swizzled by (row >> 1) & 7
// BNB: the CFH_BNBWD epilogue is compiled in (its own instantiations: the operand rows it holds cost the plain kernels
// 50-60 VGPRs and spills when it was a run-time flag)
template <int BM, int BN, int MODE, int OCC = 2, bool SWZ = false, bool BNB = false>
__global__ __launch_bounds__(256, OCC) void conv_nt_bf16_kernel(const ConvArgsH a) {
  constexpr int LSB = SWZ ? 128 : 144;          // LDS row stride in bytes: 64 bf16 (+ 16 bytes of padding)
  constexpr int RA = BM / 32, RB = BN / 32;     // 16-byte gathers per thread per K tile
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + 2 * BM * LSB;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int Cs = a.Cs;
  const bf16* __restrict__ wgt = a.w;
  const bf16* __restrict__ src = a.src;
  const int cpt = (Cs + 63) >> 6;  // K tiles per tap
  const int KT = a.KH * a.KW * cpt;
  const int smask = (1 << a.stride_log2) - 1;
  const int Hs = a.Hs, Ws = a.Ws;
  const int ntn = (a.N + BN - 1) / BN;
  const int kq = tid & 7, r0 = tid >> 3;
  const int wm0 = (wid >> 1) * WTM, wn0 = (wid & 1) * WTN;
  const int l31 = lane & 31, lh = lane >> 5;

  // XCD-aware raster (cdna_hip_programming T1): workgroups are dealt round-robin over the 8 XCDs, so work item `it`
  // runs on XCD it % 8; giving each XCD a CONTIGUOUS range of tiles keeps the N tiles of one M tile (which share the
  // A rows) and neighbouring M tiles (which share 3x3 halo rows) in one XCD's L2.  These kernels are L2/HBM-bound.
  const int per = (a.ntiles + 7) >> 3;
  for (int it = blockIdx.x; it < per * 8; it += gridDim.x) {
    const int tile = (it & 7) * per + (it >> 3);
    if (tile >= a.ntiles) continue;
    const int nt = tile % ntn, mt = tile / ntn;
    const int m0 = mt * BM, n0 = nt * BN;
    int hb[RA], wb[RA], rowoff[RA], wrow[RB];
    bool wok[RB];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int m = m0 + r0 + 32 * i;
      if (m < a.M) {
        const int hw = a.Hd * a.Wd;
        const int n = m / hw;
        const int rem = m - n * hw;
        const int hd = rem / a.Wd;
        const int wd = rem - hd * a.Wd;
        if (MODE == 0) {
          hb[i] = (hd << a.stride_log2) - a.pad_h;
          wb[i] = (wd << a.stride_log2) - a.pad_w;
        } else {
          hb[i] = hd + a.pad_h;
          wb[i] = wd + a.pad_w;
        }
        rowoff[i] = (n * Hs * Ws + hb[i] * Ws + wb[i]) * Cs + kq * 8;
        if (MODE == 1 && a.stride_log2 > 0) rowoff[i] = n * Hs * Ws * Cs + kq * 8;
      } else {
        hb[i] = -100000;
        wb[i] = -100000;
        rowoff[i] = 0;
      }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int n = n0 + r0 + 32 * i;
      wok[i] = n < a.N;
      wrow[i] = n * a.K + kq * 8;
    }

    u32x4 ra[RA], rb[RB];
    auto load_tile = [&](int kt) {
      const int tap = kt / cpt;
      const int c0 = (kt - tap * cpt) << 6;
      const int kh = tap / a.KW;
      const int kw = tap - kh * a.KW;
      const int rtap = (a.kh0 + a.khs * kh) * a.KWr + (a.kw0 + a.kws * kw);
      const int boff = rtap * Cs + c0;
      const int aoff = (MODE == 0 ? (kh * Ws + kw) : -(kh * Ws + kw)) * Cs + c0;
      const bool kok = (c0 + kq * 8) < Cs;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        bool ok;
        int off;
        if (MODE == 0) {
          const int hs = hb[i] + kh, ws = wb[i] + kw;
          ok = ((unsigned)hs < (unsigned)Hs) & ((unsigned)ws < (unsigned)Ws);
          off = rowoff[i] + aoff;
        } else if (a.stride_log2 == 0) {
          const int hs = hb[i] - kh, ws = wb[i] - kw;
          ok = ((unsigned)hs < (unsigned)Hs) & ((unsigned)ws < (unsigned)Ws);
          off = rowoff[i] + aoff;
        } else {
          const int th = hb[i] - kh, tw = wb[i] - kw;
          ok = (th >= 0) & (tw >= 0) & (((th | tw) & smask) == 0);
          const int hs = th >> a.stride_log2, ws = tw >> a.stride_log2;
          ok = ok & (hs < Hs) & (ws < Ws);
          off = rowoff[i] + (hs * Ws + ws) * Cs + c0;
        }
        ok = ok & kok;
        const void* ptr = ok ? (const void*)(src + (unsigned)off) : (const void*)sacv_zero16;
        ra[i] = ld16(ptr);
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const bool ok = wok[i] & kok;
        const void* ptr = ok ? (const void*)(wgt + (unsigned)(wrow[i] + boff)) : (const void*)sacv_zero16;
        rb[i] = ld16(ptr);
      }
    };
    auto store_tile = [&](int buf) {
      unsigned char* dA = sA + buf * BM * LSB + r0 * LSB;
      unsigned char* dB = sB + buf * BN * LSB + r0 * LSB;
      // (r0 + 32 i) >> 1 & 7 == (r0 >> 1) & 7: one chunk offset for all of this thread's rows
      const int ck = (SWZ ? (kq ^ ((r0 >> 1) & 7)) : kq) * 16;
#pragma unroll
      for (int i = 0; i < RA; ++i) *reinterpret_cast<u32x4*>(dA + 32 * i * LSB + ck) = ra[i];
#pragma unroll
      for (int i = 0; i < RB; ++i) *reinterpret_cast<u32x4*>(dB + 32 * i * LSB + ck) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < KT; ++kt) {
      const bool last = kt + 1 == KT;
      if (!last) load_tile(kt + 1);
      const unsigned char* cA = sA + cur * BM * LSB + (wm0 + l31) * LSB;
      const unsigned char* cB = sB + cur * BN * LSB + (wn0 + l31) * LSB;
      const int x7 = (l31 >> 1) & 7;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int co = (SWZ ? ((kk * 2 + lh) ^ x7) : (kk * 2 + lh)) * 16;
        bf16x8 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(cA + i * 32 * LSB + co);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(cB + j * 32 * LSB + co);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
      if (!last) store_tile(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }

    // ---- epilogue: accumulators -> LDS [BM][BN+4] fp32 -> rows of 8 channels: + bias, (+ dst), round to the output
    // type, 16-byte stores; CF_STATS: per-column sum / sum of squares of the ROUNDED values (what BatchNorm will read).
    // The thread's NR destination rows are decoded first and, for an accumulating bf16 destination (the data gradient
    // that joins a shortcut gradient), ALL their old values are requested before the accumulators are staged: the
    // loads fly during the LDS round trip instead of one exposed HBM latency per row (these launches ran at 0.14-0.20
    // of the HBM roofline with a load -> add -> store chain per row).
    constexpr int CS = BN + 4;
    constexpr int CG = BN / 8, RS = 256 / CG, NR = BM / RS;
    const int c8 = tid % CG, rr = tid / CG;
    const int n = n0 + c8 * 8;
    const bool nok = n < a.N;  // N % 8 == 0
    const int flags = a.flags;
    const int hw = a.Hd * a.Wd;
    size_t ooff[NR];
    bool rok[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int m = m0 + rr + i * RS;
      rok[i] = nok && m < a.M;
      size_t orow = (size_t)m;
      if (a.remap_s > 0 && rok[i]) {
        const int nn = m / hw;
        const int rem = m - nn * hw;
        const int hd = rem / a.Wd;
        const int wd = rem - hd * a.Wd;
        orow = ((size_t)nn * a.Hr + (size_t)hd * a.remap_s + a.remap_oh) * a.Wr + (size_t)wd * a.remap_s + a.remap_ow;
      }
      ooff[i] = orow * a.N + n;
    }
    const bool acc_h = (flags & CF_ACCUM) && !(flags & CFH_OUT_F32);
    bf16x8 oldh[NR];
    if (acc_h) {
#pragma unroll
      for (int i = 0; i < NR; ++i)
        if (rok[i]) oldh[i] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(a.dst) + ooff[i]);
    }
    float* stage = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          stage[(wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * CS + wn0 + j * 32 + l31] = acc[i][j][r];
    __syncthreads();
    // CFH_BNBWD: the rows of y (and z) of the BatchNorm layer these values are the upstream gradient of -- all of the
    // thread's rows requested at once, now that the accumulators have left the registers
    constexpr bool bnb = BNB;
    const bool bn_hasz = bnb && a.bn.z != nullptr, bn_rc = bnb && a.bn.z == nullptr && a.bn.beta != nullptr;
    bf16x8 bny[NR], bnz[NR];
    float bmu[8], bis[8], bga[8], bbe[8];
    if constexpr (BNB) {
#pragma unroll
      for (int i = 0; i < NR; ++i)
        if (rok[i]) {
          bny[i] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(a.bn.y) + ooff[i]);
          if (bn_hasz) bnz[i] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(a.bn.z) + ooff[i]);
        }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        bmu[e] = nok ? a.bn.mean[n + e] : 0.f;
        bis[e] = nok ? a.bn.invstd[n + e] : 0.f;
        bga[e] = (bn_rc && nok) ? a.bn.gamma[n + e] : 0.f;
        bbe[e] = (bn_rc && nok) ? a.bn.beta[n + e] : 0.f;
      }
    }
    float bv[8], ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bv[e] = ((flags & CF_BIAS) && nok) ? a.bias[n + e] : 0.f;
      ssum[e] = 0.f;
      ssq[e] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int row = rr + i * RS;
      if (!rok[i]) continue;
      float v[8];
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(stage + row * CS + c8 * 8);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(stage + row * CS + c8 * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = v0[e] + bv[e];
        v[4 + e] = v1[e] + bv[4 + e];
      }
      const size_t o = ooff[i];
      if (a.residual) {
        const f32x4 res0 = *reinterpret_cast<const f32x4*>(a.residual + o);
        const f32x4 res1 = *reinterpret_cast<const f32x4*>(a.residual + o + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] += res0[e];
          v[4 + e] += res1[e];
        }
      }
      if (flags & CFH_OUT_F32) {
        float* d = reinterpret_cast<float*>(a.dst) + o;
        if (flags & CF_ACCUM) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += d[e];
        }
        *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
      } else {
        bf16* d = reinterpret_cast<bf16*>(a.dst) + o;
        if (acc_h) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)oldh[i][e];
        }
        bf16x8 q;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          q[e] = (bf16)v[e];
          v[e] = (float)q[e];
        }
        *reinterpret_cast<bf16x8*>(d) = q;
      }
      if constexpr (BNB) {
        // first stage of that layer's BatchNorm backward (norm.hip: bn_bwd_partial_kernel<bf16>) on the ROUNDED values
        // just stored: g = dz * (z > 0), z read (residual layers) or recomputed from y by bn_apply's own expression,
        // rounded to the storage type as the forward pass stored it; sums of g and g * xhat
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = ((float)bny[i][e] - bmu[e]) * bis[e];
          float g = v[e];
          if (bn_hasz)
            g = (float)bnz[i][e] > 0.f ? g : 0.f;
          else if (bn_rc)
            g = (float)(bf16)(xh * bga[e] + bbe[e]) > 0.f ? g : 0.f;
          ssum[e] += g;
          ssq[e] += g * xh;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          ssum[e] += v[e];
          ssq[e] += v[e] * v[e];
        }
      }
    }
    __syncthreads();  // the stage is dead: its memory is the reduction scratch / the next tile's buffers
    if (BNB || (flags & CF_STATS)) {
      float* red = reinterpret_cast<float*>(smem);  // [2][RS][BN]
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(size_t)rr * BN + c8 * 8 + e] = ssum[e];
        red[(size_t)(RS + rr) * BN + c8 * 8 + e] = ssq[e];
      }
      __syncthreads();
      for (int c = tid; c < 2 * BN; c += 256) {
        const int which = c / BN, col = c - which * BN;
        float s = 0.f;
        for (int l = 0; l < RS; ++l) s += red[(size_t)(which * RS + l) * BN + col];
        if (n0 + col < a.N) a.stats[((size_t)mt * 2 + which) * a.N + n0 + col] = s;
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------ weight gradient
struct WgradArgsH {
  const bf16* dy;  // [npix][Cout]
  const bf16* x;   // [Nb][Hx][Wx][Cin]
  int Nb, Hx, Wx, Ho, Wo, npix;
  int nchunks, chunk;  // pixels per split (multiple of 64)
  float* part;         // [nchunks][Cout][N2] fp32
  int Cin, Cout, N2;
  int KH, KW, stride, pad;
};

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void conv_tn_bf16_kernel(const WgradArgsH a) {
  // LDS images [64 pixels][BM (+32) channels] bf16: rows padded by 64 bytes so that the four pixel rows a
  // ds_read_b64_tr_b16 touches per half-wave fall on disjoint banks
  constexpr int RSA = BM * 2 + 64, RSB = BN * 2 + 64;  // row strides in bytes
  constexpr int CGA = BM / 8, CGB = BN / 8;            // 16-byte chunks per pixel row
  constexpr int PPA = 256 / CGA, PPB = 256 / CGB;      // pixel rows covered per pass
  constexpr int RA = 64 / PPA, RB = 64 / PPB;
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;                 // [2][64][RSA]
  unsigned char* sB = smem + 2 * 64 * RSA;  // [2][64][RSB]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ntm = (a.Cout + BM - 1) / BM, ntn = (a.N2 + BN - 1) / BN;
  int bid = blockIdx.x;
  const int tm = bid % ntm;
  bid /= ntm;
  const int tn = bid % ntn;
  const int ck = bid / ntn;
  const int pix0 = ck * a.chunk;
  const int pix1 = min(pix0 + a.chunk, a.npix);
  const int m0 = tm * BM, n0 = tn * BN;
  const bf16* __restrict__ dy = a.dy;
  const bf16* __restrict__ x = a.x;

  const int ca = tid % CGA, pa = tid / CGA;
  const int cb = tid % CGB, pb = tid / CGB;
  const int co = m0 + ca * 8;
  const bool co_ok = co < a.Cout;  // Cout % 8 == 0
  const int n2 = n0 + cb * 8;
  const bool n2_ok = n2 < a.N2;  // Cin % 8 == 0: a chunk never straddles two taps
  int tap = 0, ci = 0, kh = 0, kw = 0;
  if (n2_ok) {
    tap = n2 / a.Cin;
    ci = n2 - tap * a.Cin;
    kh = tap / a.KW;
    kw = tap - kh * a.KW;
  }
  const int HoWo = a.Ho * a.Wo, Wo = a.Wo, Hx = a.Hx, Wx = a.Wx;
  const int dh = kh - a.pad, dw = kw - a.pad;

  // pixel coordinates of this thread's RB gather rows, advanced incrementally by 64 pixels per K tile (no integer
  // divisions in the loop: the MFMA block of a K tile is only 512 cycles long)
  int pn[RB], ph[RB], pw[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int pix = pix0 + pb + i * PPB;
    pn[i] = pix / HoWo;
    const int rem = pix - pn[i] * HoWo;
    ph[i] = rem / Wo;
    pw[i] = rem - ph[i] * Wo;
  }
  const int Ho = a.Ho;
  const int adv_n = 64 / HoWo;
  const int adv_h = (64 - adv_n * HoWo) / Wo;
  const int adv_w = 64 - adv_n * HoWo - adv_h * Wo;
  auto advance = [&]() {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      int w = pw[i] + adv_w;
      const int cw = w >= Wo ? 1 : 0;
      w -= cw ? Wo : 0;
      int h = ph[i] + adv_h + cw;
      const int chh = h >= Ho ? 1 : 0;
      h -= chh ? Ho : 0;
      pw[i] = w;
      ph[i] = h;
      pn[i] += adv_n + chh;
    }
  };
  u32x4 ra[RA], rb[RB];
  auto load_tile = [&](int p0) {  // the pixel state (pn, ph, pw) must already stand at p0
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int pix = p0 + pa + i * PPA;
      const bool ok = co_ok & (pix < pix1);
      const void* ptr = ok ? (const void*)(dy + (unsigned)(pix * a.Cout + co)) : (const void*)sacv_zero16;
      ra[i] = ld16(ptr);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int pix = p0 + pb + i * PPB;
      const int hi = ph[i] * a.stride + dh, wi = pw[i] * a.stride + dw;
      const bool ok = n2_ok & (pix < pix1) & ((unsigned)hi < (unsigned)Hx) & ((unsigned)wi < (unsigned)Wx);
      const void* ptr = ok ? (const void*)(x + ((unsigned)((pn[i] * Hx + hi) * Wx + wi) * (unsigned)a.Cin + (unsigned)ci))
                           : (const void*)sacv_zero16;
      rb[i] = ld16(ptr);
    }
  };
  auto store_tile = [&](int buf) {
    unsigned char* dA = sA + buf * 64 * RSA + pa * RSA + ca * 16;
    unsigned char* dB = sB + buf * 64 * RSB + pb * RSB + cb * 16;
#pragma unroll
    for (int i = 0; i < RA; ++i) *reinterpret_cast<u32x4*>(dA + i * PPA * RSA) = ra[i];
#pragma unroll
    for (int i = 0; i < RB; ++i) *reinterpret_cast<u32x4*>(dB + i * PPB * RSB) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int wm0 = (wid >> 1) * WTM, wn0 = (wid & 1) * WTN;
  const int l31 = lane & 31, lh = lane >> 5;
  // transposing fragment read (MI355X_MICROARCH / cdna_hip_programming T10): per 16-lane group g a block of
  // 4 pixel rows x 16 channels; lane 4q+p of the group addresses row q, channels 4p..4p+3, and receives channel
  // (lane & 15) of rows 0..3.  Groups 0,1 serve k = 0..7 of the MFMA's 16-deep step, groups 2,3 k = 8..15.
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int tr_row = 8 * (g >> 1) + q;            // + 4 for the second read, + 16 per k step
  const int tr_col = 16 * (g & 1) + 4 * p;        // channel inside the 32-wide MFMA block
  const int KT = (pix1 - pix0 + 63) >> 6;
  if (KT > 0) {
    load_tile(pix0);
    store_tile(0);
  }
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < KT; ++kt) {
    const bool last = kt + 1 == KT;
    advance();
    if (!last) load_tile(pix0 + ((kt + 1) << 6));
    const unsigned char* cA = sA + cur * 64 * RSA + tr_row * RSA + (wm0 + tr_col) * 2;
    const unsigned char* cB = sB + cur * 64 * RSB + tr_row * RSB + (wn0 + tr_col) * 2;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const unsigned char* pA = cA + kk * 16 * RSA + i * 64;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(pA));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(pA + 4 * RSA));
        union { s16x4 s[2]; bf16x8 v; } u;
        u.s[0] = lo;
        u.s[1] = hi;
        fa[i] = u.v;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const unsigned char* pB = cB + kk * 16 * RSB + j * 64;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(pB));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(pB + 4 * RSB));
        union { s16x4 s[2]; bf16x8 v; } u;
        u.s[0] = lo;
        u.s[1] = hi;
        fb[j] = u.v;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (!last) store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  // epilogue: accumulators -> LDS [BM][BN+4] -> fp32 slab rows with 16-byte stores
  constexpr int CS = BN + 4;
  float* stage = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        stage[(wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * CS + wn0 + j * 32 + l31] = acc[i][j][r];
  __syncthreads();
  float* __restrict__ part = a.part + (size_t)ck * a.Cout * a.N2;
  constexpr int CG = BN / 4, RS = 256 / CG;
  const int c4 = tid % CG, rr = tid / CG;
  const int n = n0 + c4 * 4;
  if (n < a.N2) {
#pragma unroll 4
    for (int row = rr; row < BM; row += RS) {
      const int m = m0 + row;
      if (m < a.Cout)
        *reinterpret_cast<f32x4*>(part + (size_t)m * a.N2 + n) =
            *reinterpret_cast<const f32x4*>(stage + row * CS + c4 * 4);
    }
  }
}

// fp32 -> bf16 elementwise (round to nearest even, NaN stays NaN): the per-step cast of the >=2-D section of the
// parameter arena (channels_last k x k weights and 1x1 weights ARE their forward GEMM panels), dlogits, inputs
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ in, bf16* __restrict__ out,
                                                         size_t n) {
  const size_t n8 = n >> 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(in + i * 8);
    const f32x4 b = *reinterpret_cast<const f32x4*>(in + i * 8 + 4);
    bf16x8 q;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      q[e] = (bf16)a[e];
      q[4 + e] = (bf16)b[e];
    }
    *reinterpret_cast<bf16x8*>(out + i * 8) = q;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) out[n8 * 8 + threadIdx.x] = (bf16)in[n8 * 8 + threadIdx.x];
}
__global__ __launch_bounds__(256) void cast_f32_kernel(const bf16* __restrict__ in, float* __restrict__ out,
                                                        size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = (float)in[i];
}

// data-gradient panels [Cin][T*Cout] bf16 of many fp32 weights in one launch (same descriptor table as the fp32
// sacv_pack_dgrad_batched: {src fp32, dst, Cout, Cin, T, src_ohwi}, 32x32 tiles transposed through LDS)
struct PackDescH {
  const float* src;
  bf16* dst;
  int Cout, Cin, T, src_ohwi;
};
__global__ __launch_bounds__(256) void pack_dgrad_batched_bf16_kernel(const PackDescH* __restrict__ descs,
                                                                       const int* __restrict__ chunk_map) {
  __shared__ float tile[32][33];
  const int di = chunk_map[2 * blockIdx.x], t = chunk_map[2 * blockIdx.x + 1];
  const PackDescH d = descs[di];
  const int nci = (d.Cin + 31) >> 5, nco = (d.Cout + 31) >> 5;
  const int cob = t % nco;
  const int qq = t / nco;
  const int cib = qq % nci;
  const int tap = qq / nci;
  const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int co = cob * 32 + r + 8 * rr, ci = cib * 32 + c;
    float v = 0.f;
    if (co < d.Cout && ci < d.Cin)
      v = d.src[d.src_ohwi ? ((size_t)co * d.T + tap) * d.Cin + ci : ((size_t)co * d.Cin + ci) * d.T + tap];
    tile[r + 8 * rr][c] = v;
  }
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int ci = cib * 32 + r + 8 * rr, co = cob * 32 + c;
    if (ci < d.Cin && co < d.Cout) d.dst[((size_t)ci * d.T + tap) * d.Cout + co] = (bf16)tile[c][r + 8 * rr];
  }
}

// im2col of the stem for the bf16 GEMM: fp32 image (any strides) -> bf16 [M][Kpad], Kpad a multiple of 64.
// A block = STEM_PW consecutive output pixels of one output row: the KH input rows they touch are staged in LDS as
// [kh][column][ci] with coalesced, bounds-masked loads (zero outside the image), and a matrix row is then KH runs of
// KW * Cin contiguous LDS floats.  A thread owns one 8-wide k group (LDS offsets computed once) and walks the pixels;
// the 16-byte stores of a block cover contiguous rows.  (Gathering straight from the image -- 8 scattered 4-byte loads
// per 16-byte store -- ran at 1.65 TB/s of the 1.2 GB this writes at batch 256.)
#define STEM_PW 32
__global__ __launch_bounds__(256) void im2col_stem_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ col,
                                                               int Nb, int Cin, int H, int W, long long sN,
                                                               long long sC, long long sH, long long sW, int Ho,
                                                               int Wo, int KH, int KW, int stride, int pad, int Kpad) {
  extern __shared__ float patch[];  // [KH][span][Cin]
  const int span = (STEM_PW - 1) * stride + KW;
  const int wo0 = blockIdx.x * STEM_PW, ho = blockIdx.y, n = blockIdx.z;
  const int hb = ho * stride - pad, wb0 = wo0 * stride - pad;
  const float* __restrict__ src = x + n * sN;
  const int per_row = span * Cin;
  for (int i = threadIdx.x; i < KH * per_row; i += 256) {
    const int kh = i / per_row, r = i - kh * per_row;
    const int cidx = r / Cin, ci = r - cidx * Cin;
    const int hi = hb + kh, wi = wb0 + cidx;
    float v = 0.f;
    if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) v = src[ci * sC + hi * sH + wi * sW];
    patch[i] = v;
  }
  __syncthreads();
  const int G = Kpad >> 3;
  const int RPB = 256 / G;
  const int gq = threadIdx.x % G, pr = threadIdx.x / G;
  if (pr >= RPB) return;
  const int K = KH * KW * Cin, run = KW * Cin;
  int off[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * gq + e;
    if (k < K) {
      const int kh = k / run;
      off[e] = kh * per_row + (k - kh * run);
    } else {
      off[e] = -1;
    }
  }
  const int npw = min(STEM_PW, Wo - wo0);
  const size_t m0 = ((size_t)n * Ho + ho) * Wo + wo0;
  for (int pq = pr; pq < npw; pq += RPB) {
    const float* base = patch + pq * stride * Cin;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)(off[e] >= 0 ? base[off[e]] : 0.f);
    *reinterpret_cast<bf16x8*>(col + ((m0 + pq) * G + gq) * 8) = v;
  }
}

// ------------------------------------------------------------------------------------------ host side
static int log2_exact_h(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return ((1 << l) == v) ? l : -1;
}
static int num_cus_h() {
  static std::once_flag once;
  static int cus = 256;
  std::call_once(once, [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
  });
  return cus;
}

template <int BM, int BN, int MODE, int OCC = 2, bool SWZ = false, bool BNB = false>
static int launch_nt_h(ConvArgsH& a, hipStream_t st) {
  size_t lds = (size_t)2 * (BM + BN) * (SWZ ? 128 : 144);
  const size_t lds_epi = (size_t)BM * (BN + 4) * sizeof(float);
  if (lds_epi > lds) lds = lds_epi;
  static std::once_flag attr_once;
  std::call_once(attr_once, [lds] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_nt_bf16_kernel<BM, BN, MODE, OCC, SWZ, BNB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  a.ntiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  int grid = num_cus_h() * OCC;
  const int items = ((a.ntiles + 7) / 8) * 8;
  if (grid > items) grid = items;
  hipLaunchKernelGGL((conv_nt_bf16_kernel<BM, BN, MODE, OCC, SWZ, BNB>), dim3(grid), dim3(256), lds, st, a);
  SACV_CHECK_LAUNCH("conv_nt_bf16_kernel");
  return SACV_OK;
}

template <int BM, int BN>
static int launch_tn_h(const WgradArgsH& a, hipStream_t st) {
  const int ntm = (a.Cout + BM - 1) / BM, ntn = (a.N2 + BN - 1) / BN;
  size_t lds = (size_t)2 * 64 * ((BM * 2 + 64) + (BN * 2 + 64));
  const size_t lds_epi = (size_t)BM * (BN + 4) * sizeof(float);
  if (lds_epi > lds) lds = lds_epi;
  static std::once_flag attr_once;
  std::call_once(attr_once, [lds] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tn_bf16_kernel<BM, BN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  hipLaunchKernelGGL((conv_tn_bf16_kernel<BM, BN>), dim3(ntm * ntn * a.nchunks), dim3(256), lds, st, a);
  SACV_CHECK_LAUNCH("conv_tn_bf16_kernel");
  return SACV_OK;
}

SACV_API_BEGIN

// bf16 convolution (forward mode 0 / data gradient mode 1) of ONE problem.  src NHWC bf16 [Nb][Hs][Ws][Cs]; w bf16
// panel [N][KH*KW*Cs] (mode 0: the weight itself when stored channels_last or 1x1; mode 1: the [Cin][taps*Cout]
// panel); dst [M][N] bf16, or fp32 with SACV_CFH_OUT_F32.  flags: SACV_CF_BIAS | SACV_CF_ACCUM | SACV_CF_STATS |
// SACV_CFH_OUT_F32.  remap / ext as in sacv_conv2d_gemm.  Cs and N must be multiples of 8.
static int conv2d_gemm_bf16_impl(const void* src, void* dst, int Nb, int Hs, int Ws, int Hd, int Wd, const void* w,
                                 const float* bias, int Cs, int N, int KH, int KW, int stride, int pad, int mode,
                                 int flags, int remap_s, int Hr, int Wr, const int* ext, float* stats,
                                 const float* residual, void* stream, const SacvBnBwdH* bn = nullptr) {
  SACV_CHECK_ARG(src && dst && w, "conv2d_gemm_bf16: null pointer");
  SACV_CHECK_ARG(!residual || (remap_s == 0 && !ext), "conv2d_gemm_bf16: residual with a remapped destination");
  SACV_CHECK_ARG(Cs % 8 == 0 && Cs > 0 && N % 8 == 0 && N > 0, "conv2d_gemm_bf16: Cs=%d and N=%d must be multiples of 8",
                 Cs, N);
  SACV_CHECK_ARG(KH > 0 && KW > 0 && (mode == 0 || mode == 1), "conv2d_gemm_bf16: bad KH/KW/mode");
  const int sl = log2_exact_h(stride);
  SACV_CHECK_ARG(sl >= 0, "conv2d_gemm_bf16: stride %d must be a power of two", stride);
  SACV_CHECK_ARG(!(flags & CF_BIAS) || bias, "conv2d_gemm_bf16: bias flag without bias pointer");
  SACV_CHECK_ARG(!(flags & CF_STATS) || stats, "conv2d_gemm_bf16: CF_STATS without a stats buffer");
  SACV_CHECK_ARG((flags & ~(CF_BIAS | CF_ACCUM | CF_STATS | CFH_OUT_F32)) == 0, "conv2d_gemm_bf16: unknown flag bits");
  if (bn) {
    SACV_CHECK_ARG(!(flags & (CF_STATS | CFH_OUT_F32)) && stats && remap_s == 0 && !ext && !residual && mode == 1,
                   "conv2d_gemm_bf16_bnbwd: a data gradient (mode 1) into a bf16 destination without row remap / tap "
                   "subset / SACV_CF_STATS, with a stats buffer");
    SACV_CHECK_ARG(bn->y && bn->mean && bn->invstd && (bn->z || !bn->beta || bn->gamma),
                   "conv2d_gemm_bf16_bnbwd: y / mean / invstd (and gamma with beta) are required");
    flags |= CFH_BNBWD;
  }
  SACV_CHECK_ARG(Nb > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, "conv2d_gemm_bf16: bad dims");
  const long long M = (long long)Nb * Hd * Wd;
  SACV_CHECK_ARG(M * (long long)(N > Cs ? N : Cs) < (1LL << 31) && (long long)Nb * Hs * Ws * Cs < (1LL << 31) &&
                     (long long)N * KH * KW * Cs < (1LL << 31),
                 "conv2d_gemm_bf16: tensor too large for 32-bit element indexing");
  ConvArgsH a;
  a.src = (const bf16*)src;
  a.dst = dst;
  a.Nb = Nb;
  a.Hs = Hs;
  a.Ws = Ws;
  a.Hd = Hd;
  a.Wd = Wd;
  a.M = (int)M;
  a.remap_s = remap_s;
  a.Hr = Hr;
  a.Wr = Wr;
  a.remap_oh = ext ? ext[9] : 0;
  a.remap_ow = ext ? ext[10] : 0;
  a.w = (const bf16*)w;
  a.bias = bias;
  a.Cs = Cs;
  a.N = N;
  a.KH = KH;
  a.KW = KW;
  a.stride_log2 = sl;
  a.pad_h = pad;
  a.pad_w = pad;
  a.kh0 = 0;
  a.khs = 1;
  a.kw0 = 0;
  a.kws = 1;
  a.KWr = KW;
  a.K = KH * KW * Cs;
  if (ext) {
    a.pad_h = ext[0];
    a.pad_w = ext[1];
    a.kh0 = ext[2];
    a.khs = ext[3];
    a.kw0 = ext[4];
    a.kws = ext[5];
    a.KWr = ext[7];
    a.K = ext[6] * ext[7] * Cs;
    SACV_CHECK_ARG(a.kh0 + a.khs * (KH - 1) < ext[6] && a.kw0 + a.kws * (KW - 1) < ext[7],
                   "conv2d_gemm_bf16: tap subset exceeds the weight panel");
  }
  a.flags = flags;
  a.stats = stats;
  a.residual = residual;
  a.bn = bn ? *bn : SacvBnBwdH{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipStream_t st = (hipStream_t)stream;
  // stride-1 1x1 convolutions / Linear layers are plain GEMMs over contiguous rows: the LDS-DMA kernel (gemm_bf16.hip)
  if (KH == 1 && KW == 1 && stride == 1 && pad == 0 && !ext && remap_s == 0 && Hs == Hd && Ws == Wd &&
      sacv_internal_gemm_nt_bf16_supported(M, N, Cs))
    return sacv_internal_gemm_nt_bf16(src, w, dst, bias, residual, stats, M, N, Cs, flags & ~CFH_BNBWD, stream, bn);
  // 3x3 / stride 1 / pad 1 with 64 output channels (layer1 of the classifier): the halo-resident kernel reads the input
  // once per tile instead of once per tap (conv3x3_halo_bf16.hip).  Its statistics rows cover 256 padded positions
  // each -- fewer rows than the 128-row granule the caller allocated: the rest is zeroed for the finalize pass.
  if (KH == 3 && KW == 3 && stride == 1 && pad == 1 && !ext && remap_s == 0 && Hs == Hd && Ws == Wd && !residual &&
      (flags & ~CF_STATS) == 0 && sacv_internal_conv3x3_halo_bf16_supported(Nb, Hs, Ws, Cs, N)) {
    if (flags & CF_STATS) {
      const int used = sacv_internal_conv3x3_halo_bf16_stats_rows(Nb, Hs, Ws);
      const int rows = (int)((M + 127) / 128);
      SACV_CHECK_ARG(used <= rows, "conv2d_gemm_bf16: statistics buffer smaller than the halo kernel's tiles");
      if (rows > used) {
        hipError_t e = hipMemsetAsync(stats + (size_t)used * 2 * N, 0, (size_t)(rows - used) * 2 * N * sizeof(float), st);
        SACV_CHECK_ARG(e == hipSuccess, "conv2d_gemm_bf16: hipMemsetAsync failed: %s", hipGetErrorString(e));
      }
    }
    return sacv_internal_conv3x3_halo_bf16(src, dst, Nb, Hs, Ws, w, Cs, N, mode, flags, stats, stream);
  }
  if (bn)  // data gradients only (the kernels with the BatchNorm-backward epilogue compiled in)
    return N <= 64 ? launch_nt_h<128, 64, 1, 3, true, true>(a, st) : launch_nt_h<128, 128, 1, 2, false, true>(a, st);
  if (N <= 64)  // three workgroups per CU on the unpadded, swizzled 48 KB LDS image (29.05 -> 28.8 ms on the bf16 classifier;
                // for wider outputs the two-per-CU 128 x 128 tile stays ahead)
    return mode == 0 ? launch_nt_h<128, 64, 0, 3, true>(a, st) : launch_nt_h<128, 64, 1, 3, true>(a, st);
  return mode == 0 ? launch_nt_h<128, 128, 0>(a, st) : launch_nt_h<128, 128, 1>(a, st);
}

int sacv_conv2d_gemm_bf16(const void* src, void* dst, int Nb, int Hs, int Ws, int Hd, int Wd, const void* w,
                          const float* bias, int Cs, int N, int KH, int KW, int stride, int pad, int mode, int flags,
                          int remap_s, int Hr, int Wr, const int* ext, float* stats, void* stream) {
  return conv2d_gemm_bf16_impl(src, dst, Nb, Hs, Ws, Hd, Wd, w, bias, Cs, N, KH, KW, stride, pad, mode, flags, remap_s,
                               Hr, Wr, ext, stats, nullptr, stream);
}

// One data gradient (mode 1) whose bf16 output dz [M][N] is the upstream gradient of the
// BatchNorm (+ ReLU) layer with input bn_y: the epilogue also writes that layer's backward first-stage sums (the bf16
// twin of sacv_conv2d_gemm_bnbwd); stats: [ceil(M / 128)][2][N] rows of (sum g, sum g * xhat).
int sacv_conv2d_gemm_bf16_bnbwd(const void* src, void* dst, int Nb, int Hs, int Ws, int Hd, int Wd, const void* w,
                                const float* bias, int Cs, int N, int KH, int KW, int stride, int pad, int mode,
                                int flags, float* stats, const void* bn_y, const void* bn_z, const float* bn_mean,
                                const float* bn_invstd, const float* bn_gamma, const float* bn_beta, void* stream) {
  SACV_CHECK_ARG((flags & ~(CF_BIAS | CF_ACCUM)) == 0, "conv2d_gemm_bf16_bnbwd: flags SACV_CF_BIAS / SACV_CF_ACCUM only");
  const SacvBnBwdH bn = {bn_y, bn_z, bn_mean, bn_invstd, bn_gamma, bn_beta};
  return conv2d_gemm_bf16_impl(src, dst, Nb, Hs, Ws, Hd, Wd, w, bias, Cs, N, KH, KW, stride, pad, mode, flags, 0, 0, 0,
                               nullptr, stats, nullptr, stream, &bn);
}

// nn.Linear on bf16 rows: y[M][N] (bf16, or fp32 with SACV_CFH_OUT_F32) = x[M][K] w[N][K]^T (+ bias) (+ residual, an
// fp32 [M][N] tensor: the ViT residual stream, vit.py:139-160) (+ y with SACV_CF_ACCUM).  K and N multiples of 8.
int sacv_linear_bf16(const void* x, const void* w, void* y, const float* bias, const float* residual, long long M,
                     int N, int K, int flags, void* stream) {
  SACV_CHECK_ARG(M > 0 && M < (1LL << 31), "linear_bf16: M");
  return conv2d_gemm_bf16_impl(x, y, 1, 1, (int)M, 1, (int)M, w, bias, K, N, 1, 1, 1, 0, 0, flags, 0, 0, 0, nullptr,
                               nullptr, residual, stream);
}

long long sacv_conv2d_wgrad_bf16_workspace_bytes(int Nb, int Ho, int Wo, int Cin_pad, int Cout, int KH, int KW,
                                                 int chunk) {
  const long long chunks = ((long long)Nb * Ho * Wo + chunk - 1) / chunk;
  return chunks * Cout * (long long)KH * KW * Cin_pad * (long long)sizeof(float);
}

// Weight gradient from bf16 activations into the fp32 gradient: dw (+)= x (*) dy, fp32 slabs per `chunk` pixels
// (multiple of 64) summed in a fixed order.  Arguments as sacv_conv2d_wgrad, one segment.
int sacv_conv2d_wgrad_bf16(const void* dy, const void* x, int Nb, int Hx, int Wx, int Ho, int Wo, float* dw,
                           float* workspace, int Cin, int Cin_pad, int Cout, int KH, int KW, int stride, int pad,
                           int chunk, int accumulate, int dw_taps, int dw_layout, void* stream) {
  SACV_CHECK_ARG(dy && x && dw && workspace, "conv2d_wgrad_bf16: null pointer");
  SACV_CHECK_ARG(dw_layout == 0 || (dw_layout == 1 && dw_taps == 0), "conv2d_wgrad_bf16: dw_layout");
  SACV_CHECK_ARG(Cin_pad % 8 == 0 && Cout % 8 == 0, "conv2d_wgrad_bf16: channels must be multiples of 8");
  SACV_CHECK_ARG(chunk % 64 == 0 && chunk > 0, "conv2d_wgrad_bf16: chunk must be a multiple of 64");
  const long long np = (long long)Nb * Ho * Wo;
  SACV_CHECK_ARG(np * (long long)(Cout > Cin_pad ? Cout : Cin_pad) < (1LL << 31) &&
                     (long long)Nb * Hx * Wx * Cin_pad < (1LL << 32),
                 "conv2d_wgrad_bf16: too large");
  WgradArgsH a;
  a.dy = (const bf16*)dy;
  a.x = (const bf16*)x;
  a.Nb = Nb;
  a.Hx = Hx;
  a.Wx = Wx;
  a.Ho = Ho;
  a.Wo = Wo;
  a.npix = (int)np;
  a.chunk = chunk;
  a.nchunks = (int)((np + chunk - 1) / chunk);
  a.part = workspace;
  a.Cin = Cin_pad;
  a.Cout = Cout;
  a.N2 = KH * KW * Cin_pad;
  a.KH = KH;
  a.KW = KW;
  a.stride = stride;
  a.pad = pad;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (Cout <= 64)
    rc = (a.N2 <= 64) ? launch_tn_h<64, 64>(a, st) : launch_tn_h<64, 128>(a, st);
  else
    rc = (a.N2 <= 64) ? launch_tn_h<128, 64>(a, st) : launch_tn_h<128, 128>(a, st);
  if (rc != SACV_OK) return rc;
  return sacv_internal_wgrad_reduce(workspace, dw, a.nchunks, Cout, Cin, Cin_pad, KH, KW, accumulate, dw_taps,
                                    dw_layout, stream);
}

int sacv_cast_bf16(const float* in, void* out, long long n, void* stream) {
  SACV_CHECK_ARG(in && out && n >= 0, "cast_bf16: bad args");
  if (n == 0) return SACV_OK;
  int blocks = (int)(((size_t)n / 8 + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, (bf16*)out, (size_t)n);
  SACV_CHECK_LAUNCH("cast_bf16");
  return SACV_OK;
}

int sacv_cast_f32(const void* in, float* out, long long n, void* stream) {
  SACV_CHECK_ARG(in && out && n >= 0, "cast_f32: bad args");
  if (n == 0) return SACV_OK;
  int blocks = (int)(((size_t)n + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(cast_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)in, out,
                     (size_t)n);
  SACV_CHECK_LAUNCH("cast_f32");
  return SACV_OK;
}

int sacv_pack_dgrad_batched_bf16(const void* descs, const int* chunk_map, int nchunks, void* stream) {
  SACV_CHECK_ARG(descs && chunk_map && nchunks > 0, "pack_dgrad_batched_bf16: bad args");
  static_assert(sizeof(PackDescH) == 32, "PackDescH layout is part of the ABI");
  hipLaunchKernelGGL(pack_dgrad_batched_bf16_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream,
                     (const PackDescH*)descs, chunk_map);
  SACV_CHECK_LAUNCH("pack_dgrad_batched_bf16");
  return SACV_OK;
}

int sacv_im2col_stem_bf16(const float* x, void* col, int Nb, int Cin, int H, int W, long long sN, long long sC,
                          long long sH, long long sW, int Ho, int Wo, int KH, int KW, int stride, int pad, int Kpad,
                          void* stream) {
  SACV_CHECK_ARG(x && col, "im2col_stem_bf16: null pointer");
  SACV_CHECK_ARG(Kpad % 64 == 0 && Kpad >= Cin * KH * KW, "im2col_stem_bf16: Kpad must be a multiple of 64 >= K");
  SACV_CHECK_ARG(Kpad / 8 <= 256, "im2col_stem_bf16: Kpad %d too large", Kpad);
  const size_t lds = (size_t)KH * ((STEM_PW - 1) * stride + KW) * Cin * sizeof(float);
  SACV_CHECK_ARG(lds <= 64 * 1024, "im2col_stem_bf16: %d x %d taps, stride %d, %d channels do not fit the LDS patch", KH,
                 KW, stride, Cin);
  SACV_CHECK_ARG(Ho <= 65535 && Nb <= 65535, "im2col_stem_bf16: grid");
  hipLaunchKernelGGL(im2col_stem_bf16_kernel, dim3((Wo + STEM_PW - 1) / STEM_PW, Ho, Nb), dim3(256), lds,
                     (hipStream_t)stream, x, (bf16*)col, Nb, Cin, H, W, sN, sC, sH, sW, Ho, Wo, KH, KW, stride, pad,
                     Kpad);
  SACV_CHECK_LAUNCH("im2col_stem_bf16");
  return SACV_OK;
}

SACV_API_END
