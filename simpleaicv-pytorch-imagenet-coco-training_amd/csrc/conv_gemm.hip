// Implicit-GEMM convolution engine for gfx950 (MI355X), fp32 in / fp32 accumulate on the
// f32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, 155 TF peak).
//
// Activations are NHWC ([N][H][W][C], C contiguous) so that the GEMM-K slice of one filter
// tap is a contiguous, 16-byte-vectorisable run; weights are repacked per step into
// K-contiguous GEMM panels.  Three kernels cover every convolution on the detection path
// (reference call sites: simpleAICV/classification/backbones/resnet.py:33-39 nn.Conv2d in
// ConvBnActBlock, simpleAICV/detection/models/fpn.py:20-58, head.py:20-33,60-73):
//
//   conv_nt_kernel  C[m][n] = sum_k A(m,k) * B[n][k]
//        forward : A = im2col gather of x      (mode 0), B = w   as [Cout][KH*KW*Cin]
//        dgrad   : A = transposed gather of dy (mode 1), B = w^T as [Cin ][KH*KW*Cout]
//        up to 8 same-weight problems (FPN levels sharing one head) run in ONE launch so the
//        small pyramid levels do not serialise as latency-bound launches of a few blocks.
//   conv_tn_kernel  dW[co][tap,ci] = sum_pix dy[pix][co] * x[pix@tap][ci]   (split over pix)
//   wgrad_reduce    deterministic sum of the split slabs, written as OIHW (+ optional accumulate)
//
// Tiling: 256 threads = 4 waves (2x2), block tile BM x BN x 32, wave tile (BM/2)x(BN/2) as
// 32x32 MFMA blocks, LDS double-buffered with one barrier per K tile, global->register
// prefetch of tile k+1 issued before the MFMAs of tile k (T14 split staging).
// LDS image [row][36] floats: ds_write_b128 per 8-lane row group and ds_read_b128 per lane are
// both bank-conflict free (row stride 36 dwords = 9 x 16 B, 9 odd).
#include "sacv_common.h"
#include <mutex>
#include <stdlib.h>

#define SACV_MAXP 8
#ifndef SACV_SGB_VALU
#define SACV_SGB_VALU 5
#endif

#define CF_BIAS 1
#define CF_RELU 2
#define CF_SIGMOID 4
#define CF_ACCUM 8
#define CF_MASK 16
#define CF_STATS 32
#define CF_AFFINE 128  // bias is [2][N]: row 0 additive, row 1 multiplicative: v = acc * bias[1][n] + bias[0][n]
#define CF_BNBWD 256   // internal (sacv_conv2d_gemm_bnbwd): the stored value is the upstream gradient dz of a BatchNorm
//                        (+ ReLU) layer -- stats rows = [sum g, sum g * xhat], g = dz * relu mask, xhat from that layer's y
#define CF_PUBLIC_MASK (63 | CF_AFFINE)  // the flags the C ABI accepts (64 is the bf16 entry point's SACV_CFH_OUT_F32)

// Ablation switches (skip gathers / LDS stores / the K-loop barrier / fragment addressing) exist only in a
// -DSACV_ABLATE diagnostic build (scripts/ablate_conv.py); the shipped kernel has none of them.
#ifdef SACV_ABLATE
#define SACV_ABL(bit) ((a.flags & (bit)) != 0)
#else
#define SACV_ABL(bit) false
#endif

// 16 bytes of zeros in device memory: masked-out gathers (padding taps, rows beyond M, channels beyond Cs) READ
// THIS instead of being loaded and then replaced by zeros.  A select on the loaded VALUE makes the compiler
// wait for the load right where the select is scheduled -- in the middle of the MFMA block -- whereas a select
// on the ADDRESS keeps every gathered register untouched until the LDS store at the end of the K iteration.
__device__ __attribute__((aligned(16))) float sacv_zero16[4] = {0.f, 0.f, 0.f, 0.f};

struct ConvProb {
  const float* src;
  float* dst;
  const float* mask;
  int Nb, Hs, Ws;  // source tensor dims
  int Hd, Wd;      // M-space spatial dims
  int M;           // Nb*Hd*Wd
  int tile_begin;  // first M-tile of this problem
  int remap_s, Hr, Wr;  // dst row remap (strided scatter) when remap_s > 0
  int remap_oh, remap_ow;  // + offsets: row = ((n*Hr + h*remap_s + remap_oh)*Wr + w*remap_s + remap_ow)
};

struct ConvArgs {
  int ks_log2;  // log2 of the K slab of the launched conv_nt variant (5; 4 for the 128 x 128 split-operand item)
  ConvProb p[SACV_MAXP];
  int nprob;
  int total_mtiles;
  const float* w;
  const float* bias;
  int Cs;  // channels of src
  int N, K;
  int KH, KW, stride_log2, pad_h, pad_w;
  // tap subset (parity-class data gradient of a strided conv): loop tap (kh,kw) of the KH x KW loop reads
  // weight tap (kh0 + khs*kh, kw0 + kws*kw) of a panel that is KWr taps wide
  int kh0, khs, kw0, kws, KWr;
  int mode;
  int flags;
  // persistent schedule (hybrid data-parallel + stream-K):
  //   phase 1: block b owns whole tiles b, b+grid, ... (base rounds)
  //   phase 2: the remaining rem_tiles (< grid) tiles are laid end to end as rem_tiles*KT K-iterations and cut
  //            into sk_blocks equal contiguous ranges, one per block (stream-K): a range is the tail of one tile
  //            and/or the head of the next (never more: sk_blocks >= rem_tiles).  A range that is a whole tile is
  //            finished in place; partial accumulators go to slab 2*b + s of `partial` (segment s of block b) and
  //            conv_fixup_kernel sums the slabs of each tile in block order (= K order, deterministic).
  //            Every workgroup gets the same number of K-iterations, so all CUs finish together.
  int grid, base, rem_tiles, sk_blocks;
  float* partial;  // [slots][BM][BN] row-major K-slice tiles
  float* stats;    // [total_mtiles][2][N] column sum / sum of squares (CF_STATS); [sum g, sum g*xhat] (CF_BNBWD)
  // CF_BNBWD: the BatchNorm layer whose upstream gradient this launch stores (single problem, dst rows = its rows)
  const float* bn_y;       // its pre-normalisation input [M][N]
  const float* bn_z;       // ReLU mask source (the stored block output) or null: the mask is recomputed from y
  const float* bn_mean;    // [N]
  const float* bn_invstd;  // [N]
  const float* bn_gamma;   // [N] (mask recomputation)
  const float* bn_beta;    // [N] (mask recomputation; null with a layer that has no ReLU: no mask)
};

struct TileCoord {
  int pi, m0, n0;
};

template <int BM, int BN>
__device__ __forceinline__ TileCoord tile_coord(const ConvArgs& a, int tile) {
  const int ntn = (a.N + BN - 1) / BN;
  const int nt = tile % ntn;
  const int mt = tile / ntn;
  int pi = 0;
  for (int i = 1; i < a.nprob; ++i)
    if (mt >= a.p[i].tile_begin) pi = i;
  TileCoord t;
  t.pi = pi;
  t.m0 = (mt - a.p[pi].tile_begin) * BM;
  t.n0 = nt * BN;
  return t;
}

// Vectorised tile epilogue shared by the main kernel (whole tiles, rows staged in LDS) and the fix-up
// kernel (split tiles, rows summed from the K-slice slabs).  NT threads walk the BM x BN tile as rows of
// float4: thread -> (column group c4 = t % (BN/4), row lane rr = t / (BN/4)), rows rr, rr+RS, ...
// Per float4: + bias, (+ dst), ReLU, sigmoid, (* mask > 0), optional strided row scatter, 16-byte store.
// CF_STATS: per-column sum and sum of squares of the STORED values over the tile's valid rows go to
// stats[mt][0][n] / stats[mt][1][n] (mt = global M-tile index) -- the BatchNorm batch statistics of a
// forward conv, or the bias gradient when the stored value is a dy.  Fixed order => deterministic.
template <int BM, int BN, int NT, typename RowFn>
__device__ __forceinline__ void vec_epilogue(const ConvArgs& a, const TileCoord& tc, int mt, RowFn get_row,
                                             float* red /* LDS scratch, 2*(NT/(BN/4))*BN floats, CF_STATS only */) {
  constexpr int CG = BN / 4;
  constexpr int RS = NT / CG;  // row lanes
  const int t = threadIdx.x;
  const int c4 = t % CG, rr = t / CG;
  const ConvProb& P = a.p[tc.pi];
  const int flags = a.flags;
  float* __restrict__ dst = P.dst;
  const int N = a.N, M = P.M;
  const int n = tc.n0 + c4 * 4;
  const bool nok = n < N;  // N % 4 == 0
  f32x4 bv = {0.f, 0.f, 0.f, 0.f}, sv = {1.f, 1.f, 1.f, 1.f};
  if ((flags & CF_BIAS) && nok) bv = *reinterpret_cast<const f32x4*>(a.bias + n);
  // an eval-mode BatchNorm folded into the convolution: per-channel scale (gamma / sqrt(var + eps)) and shift
  if ((flags & CF_AFFINE) && nok) sv = *reinterpret_cast<const f32x4*>(a.bias + N + n);
  f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
  // CF_BNBWD: per-channel constants of the BatchNorm layer this tile's values are the upstream gradient of
  f32x4 bmu = {0.f, 0.f, 0.f, 0.f}, bis = bmu, bga = bmu, bbe = bmu;
  if ((flags & CF_BNBWD) && nok) {
    bmu = *reinterpret_cast<const f32x4*>(a.bn_mean + n);
    bis = *reinterpret_cast<const f32x4*>(a.bn_invstd + n);
    if (!a.bn_z && a.bn_beta) {
      bga = *reinterpret_cast<const f32x4*>(a.bn_gamma + n);
      bbe = *reinterpret_cast<const f32x4*>(a.bn_beta + n);
    }
  }
  const int hw = P.Hd * P.Wd;
#pragma unroll 4
  for (int row = rr; row < BM; row += RS) {
    const int m = tc.m0 + row;
    if (!(nok && m < M)) continue;
    f32x4 v = get_row(row, c4);
    if (flags & CF_AFFINE) v = v * sv;
    v += bv;
    size_t orow = (size_t)m;
    if (P.remap_s > 0) {
      int nn = m / hw;
      int rem = m - nn * hw;
      int hd = rem / P.Wd;
      int wd = rem - hd * P.Wd;
      orow = ((size_t)nn * P.Hr + (size_t)hd * P.remap_s + P.remap_oh) * P.Wr + (size_t)wd * P.remap_s + P.remap_ow;
    }
    const size_t o = orow * N + n;
    if (flags & CF_ACCUM) v += *reinterpret_cast<const f32x4*>(dst + o);
    if (flags & CF_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (flags & CF_SIGMOID) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
    }
    if (flags & CF_MASK) {
      f32x4 mk = *reinterpret_cast<const f32x4*>(P.mask + o);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
    }
    *reinterpret_cast<f32x4*>(dst + o) = v;
    if (flags & CF_BNBWD) {
      // the first stage of that layer's BatchNorm backward (norm.hip: bn_bwd_partial_kernel) on values that are in
      // registers anyway: g = dz * (z > 0) with z read (residual layers) or recomputed from y by the forward pass's
      // own expression ((y - mean) * invstd) * gamma + beta (bit-identical: -ffp-contract=off); sums of g and g * xhat
      const f32x4 yv = *reinterpret_cast<const f32x4*>(a.bn_y + o);
      f32x4 xh, g = v;
#pragma unroll
      for (int e = 0; e < 4; ++e) xh[e] = (yv[e] - bmu[e]) * bis[e];
      if (a.bn_z) {
        const f32x4 zv = *reinterpret_cast<const f32x4*>(a.bn_z + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = zv[e] > 0.f ? v[e] : 0.f;
      } else if (a.bn_beta) {
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = (xh[e] * bga[e] + bbe[e]) > 0.f ? v[e] : 0.f;
      }
      ssum += g;
      ssq += g * xh;
    } else {
      ssum += v;
      ssq += v * v;
    }
  }
  if (flags & (CF_STATS | CF_BNBWD)) {
    __syncthreads();  // callers' row sources (LDS stage) are dead from here on
    *reinterpret_cast<f32x4*>(red + (size_t)rr * BN + c4 * 4) = ssum;
    *reinterpret_cast<f32x4*>(red + (size_t)(RS + rr) * BN + c4 * 4) = ssq;
    __syncthreads();
    for (int c = t; c < 2 * BN; c += NT) {
      const int which = c / BN, col = c - which * BN;
      float acc = 0.f;
      for (int l = 0; l < RS; ++l) acc += red[(size_t)(which * RS + l) * BN + col];
      if (tc.n0 + col < N) a.stats[((size_t)mt * 2 + which) * N + tc.n0 + col] = acc;
    }
  }
}

// first K-iteration (in the end-to-end numbering of the remainder tiles) of stream-K block b
__device__ __forceinline__ int sk_range_begin(const ConvArgs& a, int KT, int b) {
  return (int)(((long long)b * a.rem_tiles * KT) / a.sk_blocks);
}

// OCC = workgroups per CU the launch bounds allow for; SWZ = unpadded LDS rows (32 floats) with the 16-byte chunks
// XOR-swizzled by key (row >> 1) & 7 (exact for the lane groups of ds_read_b128, as in winograd.hip) instead of rows
// padded to 36 floats: the 128 x 64 variant then needs 48 KB and three workgroups fit a CU.
#ifndef SACV_SP_XCD_ORDER
#define SACV_SP_XCD_ORDER 0  // measured neutral with the split-operand kernels too (23.20 vs 23.19 ms, three pairs)
#endif
typedef __bf16 cbf16;
typedef __bf16 cbf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 cbf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));
typedef float cf32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int cu32x2 __attribute__((ext_vector_type(2)));

// x = hi + mid + lo for TWO fp32 values at a time, the terms packed as bf16 pairs: v_cvt_pk_bf16_f32 rounds both, the
// pair goes back to fp32 with one shift and one mask, v_pk_add_f32 takes both residuals -- 9 vector instructions per pair
// (element-wise code compiled to 15: PMC round 4 showed the split-operand kernels bound by instruction ISSUE).
__device__ __forceinline__ void split3_pair(const cf32x2 x, unsigned& h, unsigned& m, unsigned& l) {
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, cbf16x2));
  const cf32x2 hf = {__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};
  const cf32x2 r1 = x - hf;  // exact
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, cbf16x2));
  const cf32x2 mf = {__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xffff0000u)};
  const cf32x2 r2 = r1 - mf;  // exact difference, rounded once below
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, cbf16x2));
}
// 4 consecutive values -> the three 8-byte plane pieces
__device__ __forceinline__ void split3_quad(const f32x4 v, cu32x2& h, cu32x2& m, cu32x2& l) {
  unsigned h0, m0, l0, h1, m1, l1;
  split3_pair(cf32x2{v[0], v[1]}, h0, m0, l0);
  split3_pair(cf32x2{v[2], v[3]}, h1, m1, l1);
  h = cu32x2{h0, h1};
  m = cu32x2{m0, m1};
  l = cu32x2{l0, l1};
}

// SP (split operands, round 4): the fp32 matrix pipe (v_mfma_f32_32x32x2_f32, 157 TFLOP/s) bounds this kernel; the
// bf16 pipe is 16x faster.  With SP the gathered fp32 values are split WHEN THEY ARE STORED TO LDS into three bf16 terms,
// x = hi + mid + lo exactly (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): three round-to-nearest terms
// carry >= 24 significant bits), and a.b is formed from the six products of weight >= 2^-16 (hi.hi, mid.hi, hi.mid,
// mid.mid, lo.hi, hi.lo) with v_mfma_f32_32x32x16_bf16 into the same fp32 accumulators: 6 x 32 cycles per 16 K instead
// of 8 x 64.  What is dropped (mid.lo, lo.mid, lo.lo <= 2^-25 |a.b|) lies below the rounding of one fp32 addition;
// against fp64 the results carry the error of the fp32 pipe (tests/test_gpu_kernels.py, test_gpu_winograd.py for the
// same scheme in the Winograd GEMM).  Global traffic is unchanged (fp32 tensors), the LDS image is 1.5x the bytes:
// [2 buffers][3 planes][rows][32 bf16], 16-byte chunks XOR-swizzled by (row >> 2) & 3 for the ds_read_b128 fragments.
// KS = K per slab: 32, or (SP only) 16 -- the 128 x 128 split-operand item: at the same LDS bytes as the 128 x 64 / K 32 item
// it splits and gathers 2/3 of the operand elements per MFMA (PMC round 4: the SP kernels spend 40 % of their wave cycles
// ISSUING instructions, the split and the gather addresses; matrix pipe busy 0.2).
template <int BM, int BN, int MODE, int WN, int OCC = WN, bool SWZ = false, bool SP = false, int KS = 32>
__global__ __launch_bounds__(128 * WN, OCC) void conv_nt_kernel(const ConvArgs a) {
  static_assert(KS == 32 || (SP && KS == 16), "16-deep slabs exist for the split-operand form only");
  constexpr int LS = SWZ ? 32 : 36;
  constexpr int NT = 128 * WN;                 // 2 x WN waves
  constexpr int KQ = KS / 4;                   // threads per tile row (16 B = 4 K each)
  constexpr int PR = KS / 2;                   // SP: floats per plane row (KS bf16)
  constexpr int RPP = NT / KQ;                 // tile rows fetched per pass
  constexpr int RA = BM / RPP, RB = BN / RPP;
  constexpr int WTM = BM / 2, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;
  float* sB = smem + (SP ? 2 * 3 * BM * PR : 2 * BM * LS);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int bid = blockIdx.x;
  const int Cs = a.Cs;
  const float* __restrict__ wgt = a.w;
  const int cpt = (Cs + KS - 1) / KS;  // K tiles per tap (last one zero-padded when Cs % KS != 0)
  const int KT = a.KH * a.KW * cpt;
  const int smask = (1 << a.stride_log2) - 1;

  // segments of this block: `base` whole tiles, then its stream-K range (parts of at most two remainder tiles)
  int sk_u0 = 0, sk_u1 = 0, nsk = 0;
  if (bid < a.sk_blocks) {
    sk_u0 = sk_range_begin(a, KT, bid);
    sk_u1 = sk_range_begin(a, KT, bid + 1);
    if (sk_u1 > sk_u0) nsk = (sk_u1 > (sk_u0 / KT + 1) * KT) ? 2 : 1;
  }
  const int nseg = a.base + nsk;
  if (nseg == 0) return;

  // per-thread gather state of the CURRENT segment.  All offsets are 32-bit element indices (tensors are
  // checked to be < 2^30 elements) relative to wave-uniform base pointers, so a gather is a handful of
  // cheap VALU ops + one saddr global load: no 64-bit multiplies, no branches (both would fence the
  // scheduler and cluster ~100 issue cycles of address arithmetic between two MFMAs).
  const int kq = tid % KQ;
  const int r0 = tid / KQ;
  int hb[RA], wb[RA], rowoff[RA];
  int wrow[RB];
  bool wok[RB];
  const float* src;
  int Hs, Ws, n0;
  TileCoord tc;
  int k_begin, k_end, slot, cur_tile = 0;

  auto setup = [&](int sidx) {
    int tile;
    if (sidx < a.base) {
      // plain round-robin.  (An XCD-contiguous order -- each XCD taking grid/8 consecutive tiles per
      // round -- was measured: +2 % on a single large problem, -7 % on the grouped head launches.)
      tile = bid + sidx * a.grid;
      if (SP && SACV_SP_XCD_ORDER && ((a.base * a.grid) & 7) == 0) {
        // XCD-contiguous order for the split-operand kernels: workgroup b runs on XCD b % 8; every XCD takes a contiguous
        // range of the whole tiles, so the N tiles of one M tile (same activation rows) meet in one L2 at about the
        // same time.  (For the fp32-pipe kernels, bound by the matrix pipe, this order measured +2 % / -7 %: round 1.)
        const int per = (a.base * a.grid) >> 3;  // a bijection of [0, base * grid) when that is a multiple of 8
        tile = (tile & 7) * per + (tile >> 3);
      }
      k_begin = 0;
      k_end = KT;
      slot = -1;
    } else {
      const int s = sidx - a.base;  // 0: first tile of the stream-K range, 1: the next tile
      const int t0 = sk_u0 / KT;
      const int rt = t0 + s;
      tile = a.base * a.grid + rt;
      k_begin = s ? 0 : sk_u0 - t0 * KT;
      k_end = sk_u1 - rt * KT;
      if (k_end > KT) k_end = KT;
      slot = (k_begin == 0 && k_end == KT) ? -1 : 2 * bid + s;
    }
    cur_tile = tile;
    tc = tile_coord<BM, BN>(a, tile);
    const ConvProb& P = a.p[tc.pi];
    src = P.src;
    Hs = P.Hs;
    Ws = P.Ws;
    n0 = tc.n0;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      int m = tc.m0 + r0 + RPP * i;
      if (m < P.M) {
        int hw = P.Hd * P.Wd;
        int n = m / hw;
        int rem = m - n * hw;
        int hd = rem / P.Wd;
        int wd = rem - hd * P.Wd;
        if (MODE == 0) {
          hb[i] = (hd << a.stride_log2) - a.pad_h;
          wb[i] = (wd << a.stride_log2) - a.pad_w;
        } else {
          hb[i] = hd + a.pad_h;
          wb[i] = wd + a.pad_w;
        }
        // element offset of source pixel (hb, wb) of image n, channel kq*4 (may be "negative": only used
        // when the tap lands inside the image)
        rowoff[i] = (n * Hs * Ws + hb[i] * Ws + wb[i]) * Cs + kq * 4;
        if (MODE == 1 && a.stride_log2 > 0) rowoff[i] = n * Hs * Ws * Cs + kq * 4;
      } else {
        hb[i] = -100000;  // never valid in either mode
        wb[i] = -100000;
        rowoff[i] = 0;
      }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      int n = n0 + r0 + RPP * i;
      wok[i] = n < a.N;
      wrow[i] = n * a.K + kq * 4;
    }
  };

  f32x4 ra[RA], rb[RB];
  // coordinates of the K slab being fetched (wave-uniform)
  int lt_c0 = 0, lt_kh = 0, lt_kw = 0, lt_aoff = 0, lt_boff = 0;
  bool lt_kok = false;
  auto load_begin = [&](int kt, bool enable) {
    int tap = kt / cpt;
    lt_c0 = (kt - tap * cpt) * KS;
    lt_kh = tap / a.KW;
    lt_kw = tap - lt_kh * a.KW;
    int rtap = (a.kh0 + a.khs * lt_kh) * a.KWr + (a.kw0 + a.kws * lt_kw);
    lt_boff = rtap * Cs + lt_c0;
    // source offset of this tap relative to (hb, wb): scalar
    lt_aoff = (MODE == 0 ? (lt_kh * Ws + lt_kw) : -(lt_kh * Ws + lt_kw)) * Cs + lt_c0;
    lt_kok = enable && ((lt_c0 + kq * 4) < Cs);
  };
  auto load_a = [&](int i) {
    bool ok;
    int off;
    if (MODE == 0) {
      int hs = hb[i] + lt_kh, ws = wb[i] + lt_kw;
      ok = ((unsigned)hs < (unsigned)Hs) && ((unsigned)ws < (unsigned)Ws);
      off = rowoff[i] + lt_aoff;
    } else if (a.stride_log2 == 0) {
      int hs = hb[i] - lt_kh, ws = wb[i] - lt_kw;
      ok = ((unsigned)hs < (unsigned)Hs) && ((unsigned)ws < (unsigned)Ws);
      off = rowoff[i] + lt_aoff;
    } else {
      int th = hb[i] - lt_kh, tw = wb[i] - lt_kw;
      ok = (th >= 0) && (tw >= 0) && (((th | tw) & smask) == 0);
      int hs = th >> a.stride_log2, ws = tw >> a.stride_log2;
      ok = ok && (hs < Hs) && (ws < Ws);
      off = rowoff[i] + (hs * Ws + ws) * Cs + lt_c0;
    }
    ok = ok && lt_kok;
    const float* ptr = ok ? src + (unsigned)off : sacv_zero16;
    ra[i] = *reinterpret_cast<const f32x4*>(ptr);
  };
  auto load_b = [&](int i) {
    bool ok = wok[i] && lt_kok;
    const float* ptr = ok ? wgt + (unsigned)(wrow[i] + lt_boff) : sacv_zero16;
    rb[i] = *reinterpret_cast<const f32x4*>(ptr);
  };
  auto load_tile = [&](int kt) {
    load_begin(kt, true);
#pragma unroll
    for (int i = 0; i < RA; ++i) load_a(i);
#pragma unroll
    for (int i = 0; i < RB; ++i) load_b(i);
  };
  // SP: 4 consecutive K of one row -> three 8-byte stores (planes hi, mid, lo)
  auto put3 = [&](float* img, int rows, int row, const f32x4 v) {
    cu32x2 h, m, l;
    split3_quad(v, h, m, l);
    const int key = KS == 32 ? (row >> 2) & 3 : (row >> 3) & 1;  // swizzle of the 16-byte chunks (ds_read_b128 groups)
    float* d = img + row * PR + ((((kq >> 1) ^ key) << 2) + ((kq & 1) << 1));
    *reinterpret_cast<cu32x2*>(d) = h;
    *reinterpret_cast<cu32x2*>(d + rows * PR) = m;
    *reinterpret_cast<cu32x2*>(d + 2 * rows * PR) = l;
  };
  auto store_tile = [&](int buf) {
    if constexpr (SP) {
      float* dA = sA + buf * 3 * BM * PR;
      float* dB = sB + buf * 3 * BN * PR;
#pragma unroll
      for (int i = 0; i < RA; ++i) put3(dA, BM, r0 + RPP * i, ra[i]);
#pragma unroll
      for (int i = 0; i < RB; ++i) put3(dB, BN, r0 + RPP * i, rb[i]);
      return;
    }
    float* dA = sA + buf * BM * LS;
    float* dB = sB + buf * BN * LS;
#pragma unroll
    for (int i = 0; i < RA; ++i)
      *reinterpret_cast<f32x4*>(dA + (r0 + RPP * i) * LS + (SWZ ? ((kq ^ (((r0 + RPP * i) >> 1) & 7)) << 2) : kq * 4)) = ra[i];
#pragma unroll
    for (int i = 0; i < RB; ++i)
      *reinterpret_cast<f32x4*>(dB + (r0 + RPP * i) * LS + (SWZ ? ((kq ^ (((r0 + RPP * i) >> 1) & 7)) << 2) : kq * 4)) = rb[i];
  };

  f32x16 acc[TM][TN];
  const int wm0 = (wid / WN) * WTM;
  const int wn0 = (wid % WN) * WTN;
  const int l31 = lane & 31;
  const int lh = lane >> 5;

  // (a static priority for the second-dispatched half of the grid was also measured: no gain)
#ifdef SACV_STAMP
  // diagnostic build: per-workgroup phase timestamps (s_memtime) into a.stats (never in the shipped library)
  long long* stamp = reinterpret_cast<long long*>(a.stats) + (size_t)bid * 64;
  int nstamp = 0;
#define SACV_STAMP_HERE()                                                            \
  do {                                                                               \
    if (tid == 0 && nstamp < 64) stamp[nstamp] = (long long)__builtin_amdgcn_s_memtime(); \
    ++nstamp;                                                                        \
  } while (0)
#else
#define SACV_STAMP_HERE() do {} while (0)
#endif
  SACV_STAMP_HERE();
  // (delaying half of the workgroups by half a K iteration at start-up -- by parity or by grid half -- to
  // break the lockstep of the two workgroups of a CU was measured: no effect, they re-synchronise)
  // (the same stagger restricted to short-K, store-heavy 1x1 problems -- odd workgroups sleeping 3-14 us first --
  // only adds the delay: 78 -> 81..91 us)
  setup(0);
  load_tile(k_begin);
  store_tile(0);
  __syncthreads();
  SACV_STAMP_HERE();
  int cur = 0;
  for (int sidx = 0; sidx < nseg; ++sidx) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const TileCoord my_tc = tc;
    const int my_slot = slot;
    const int my_tile = cur_tile;
    const int kb = k_begin, ke = k_end;
    const bool more = (sidx + 1) < nseg;
    for (int kt = kb; kt < ke; ++kt) {
      const bool last = (kt + 1 == ke);
      const bool have_next = !last || more;
      if (last && more) setup(sidx + 1);  // switch the gather state to the next segment
      load_begin(last ? k_begin : kt + 1, have_next && !SACV_ABL(1024));
      if constexpr (SP) {
        constexpr int NKS = KS / 16;  // MFMA K steps per slab
        const float* pA = sA + cur * 3 * BM * PR + (wm0 + l31) * PR;
        const float* pB = sB + cur * 3 * BN * PR + (wn0 + l31) * PR;
        // swizzle key of this lane's fragment rows (wm0, wn0, i * 32: multiples of 16)
        const int k3 = KS == 32 ? (l31 >> 2) & 3 : (l31 >> 3) & 1;
        cbf16x8 fa[NKS][3][TM], fb[NKS][3][TN];
        // the next slab's gathers first: the MFMA block of a slab is only 768 cycles per wave, all of it is needed to
        // cover the loads
#pragma unroll
        for (int i = 0; i < RA; ++i) load_a(i);
#pragma unroll
        for (int i = 0; i < RB; ++i) load_b(i);
        // reads and products in one order -- K step, plane (hi, mid, lo), A blocks then B blocks -- so that the reads
        // trickle in between the MFMAs; products per K step: hi.hi, mid.hi, hi.mid, mid.mid, lo.hi, hi.lo
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const int o = ((ks * 2 + lh) ^ k3) * 4;
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
              fa[ks][pl][i] = *reinterpret_cast<const cbf16x8*>(pA + pl * BM * PR + i * 32 * PR + o);
#pragma unroll
            for (int j = 0; j < TN; ++j)
              fb[ks][pl][j] = *reinterpret_cast<const cbf16x8*>(pB + pl * BN * PR + j * 32 * PR + o);
          }
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          constexpr int pa[6] = {0, 1, 0, 1, 2, 0};
          constexpr int pb[6] = {0, 0, 1, 1, 0, 2};
#pragma unroll
          for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][pa[t6]][i], fb[ks][pb[t6]][j], acc[i][j], 0, 0, 0);
        }
#ifndef SACV_NO_SGB
        {
          constexpr int NRD = NKS * 3 * (TM + TN), NMF = NKS * 6 * TM * TN, PRE = TM + TN + 1;
          __builtin_amdgcn_sched_group_barrier(0x100, PRE, 0);
#pragma unroll
          for (int m = 0; m < NMF; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                         // 1 MFMA
            if (m < NRD - PRE) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 LDS read
          }
        }
#endif
        __builtin_amdgcn_s_setprio(0);
        if (!last) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
        continue;
      }
      const float* cA = sA + (SACV_ABL(8192) ? 0 : cur * BM * LS + (wm0 + l31) * LS + (SWZ ? 0 : 4 * lh));
      const float* cB = sB + (SACV_ABL(8192) ? 0 : cur * BN * LS + (wn0 + l31) * LS + (SWZ ? 0 : 4 * lh));
      // float offset of the 16-byte chunk (2 * quarter + lh) of this lane's fragment row (wm0, wn0, i * 32: multiples of 16)
      const int x7 = (l31 >> 1) & 7;
      auto qoff = [&](int quarter) { return SWZ ? (((quarter * 2 + lh) ^ x7) << 2) : quarter * 8; };
      // The next slab's gathers (address arithmetic + 16-byte loads) and the next fragments' LDS reads
      // are issued in the shadow of the current MFMAs (an fp32 32x32x2 MFMA occupies the matrix pipe
      // for 64 cycles but the issue port for 8), so a wave keeps the pipe busy on its own instead of
      // alternating a load phase with a math phase in lockstep with its SIMD partner.
      f32x4 fa[2][TM], fb[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(cA + i * 32 * LS + qoff(0));
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const f32x4*>(cB + j * 32 * LS + qoff(0));
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int cb = jj & 1, nbuf = cb ^ 1;
        if (jj < 3) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
            fa[nbuf][i] = *reinterpret_cast<const f32x4*>(cA + i * 32 * LS + qoff(jj + 1));
#pragma unroll
          for (int j = 0; j < TN; ++j)
            fb[nbuf][j] = *reinterpret_cast<const f32x4*>(cB + j * 32 * LS + qoff(jj + 1));
        }
        // one A gather and one B gather per quarter of the MFMA block.  (Issuing all eight in the first two
        // quarters, to give them more time before the LDS store, was measured: 122 -> 106-115 TF.)
#ifndef SACV_GATHER_QUARTERS
#define SACV_GATHER_QUARTERS 4  // quarters of the MFMA block over which the next slab's gathers are spread
#endif
        constexpr int GPQ_A = (RA + SACV_GATHER_QUARTERS - 1) / SACV_GATHER_QUARTERS;
        constexpr int GPQ_B = (RB + SACV_GATHER_QUARTERS - 1) / SACV_GATHER_QUARTERS;
#pragma unroll
        for (int g = 0; g < GPQ_A; ++g)
          if (jj * GPQ_A + g < RA) load_a(jj * GPQ_A + g);
#pragma unroll
        for (int g = 0; g < GPQ_B; ++g)
          if (jj * GPQ_B + g < RB) load_b(jj * GPQ_B + g);
        // raise the wave's issue priority for the MFMA block (dropped again before the LDS stores and the
        // barrier): the wave in its math phase then wins arbitration over its SIMD partner's address/LDS
        // work instead of alternating with it.  Measured +7..9 % on the 3x3 convs (112 -> 121 TF).
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cb][i][e], fb[cb][j][e], acc[i][j], 0, 0, 0);
#ifndef SACV_NO_SGB
#pragma unroll
        for (int m = 0; m < 4 * TM * TN; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // 1 MFMA
          if (m < TM + TN) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 LDS read
          __builtin_amdgcn_sched_group_barrier(0x002, SACV_SGB_VALU, 0);           // address VALU
          if (SACV_GATHER_QUARTERS == 4 ? (m == 5 || m == 10) : (m == 2 || m == 5 || m == 8 || m == 11))
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 global load
        }
#endif
      }
      // the slab of the NEXT segment stays in registers across the epilogue (the LDS is needed to stage
      // the output tile); within a segment it goes straight to the other LDS buffer
      __builtin_amdgcn_s_setprio(0);
      if (!last && !SACV_ABL(2048)) store_tile(cur ^ 1);
      if (!SACV_ABL(4096)) __syncthreads();
      cur ^= 1;
    }
    SACV_STAMP_HERE();  // end of K loop
    // ---- epilogue: accumulators -> LDS [BM][BN+4] -> 16-byte row-major stores
    {
      constexpr int CS = BN + 4;
      float* stage = smem;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            stage[(wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * CS + wn0 + j * 32 + l31] = acc[i][j][r];
      __syncthreads();
      if (my_slot < 0) {
        const int ntn = (a.N + BN - 1) / BN;
        vec_epilogue<BM, BN, NT>(
            a, my_tc, my_tile / ntn,
            [&](int row, int c4) { return *reinterpret_cast<const f32x4*>(stage + row * CS + c4 * 4); },
            smem + BM * CS);
      } else {
        // K-slice of a split tile: row-major slab for the fix-up kernel
        float* dst = a.partial + (size_t)my_slot * (BM * BN);
        constexpr int CG = BN / 4, RS = NT / CG;
        const int c4 = tid % CG, rr = tid / CG;
#pragma unroll 4
        for (int row = rr; row < BM; row += RS)
          *reinterpret_cast<f32x4*>(dst + row * BN + c4 * 4) =
              *reinterpret_cast<const f32x4*>(stage + row * CS + c4 * 4);
      }
      if (more) {
        __syncthreads();  // stage fully consumed
        store_tile(0);    // first slab of the next segment (gathered under the last MFMAs)
        __syncthreads();
        cur = 0;
      }
      SACV_STAMP_HERE();  // end of epilogue (+ next slab in LDS)
    }
  }
}

// Sums the stream-K slabs of each remainder tile in block order (deterministic) and applies the epilogue.
// (Round 4 built the alternative -- the slabs summed and the tile finished INSIDE conv_nt_kernel by the workgroup that
// arrives last at a per-tile ticket: write-through slab stores, drained, one agent-scope add per workgroup, one acquire in
// the finisher, the finish deferred to the end of the workgroup's K loops, column sums in this kernel's order by four
// virtual row lanes per thread; commit f65708e, outputs and CF_STATS sums bit-identical to this kernel on 26 shapes.  72
// launches fewer per RetinaNet step and the same step time: 27.13 vs 27.11 ms in a same-session A/B -- one 256-thread
// workgroup per tile finishing after everybody else has left is as long a tail as this launch.  Not adopted.)
// NT = 1024: one workgroup per tile (the form CF_STATS needs: the tile's column sums come out of one workgroup).
// NT = 256, QS = 4: four workgroups per tile, 32 rows each -- for launches without CF_STATS.  In the backward pass the
// fix-up shares the chip with the weight gradients of the side stream, whose persistent workgroups hold most of every
// CU's registers: a 1024-thread workgroup has to wait for a CU to drain (13 us alone became 25-44 us there), 256-thread
// ones fit into what is left.
template <int BM, int BN, int NT = 1024, int QS = 1>
__global__ __launch_bounds__(NT) void conv_fixup_kernel(const ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float red[QS == 1 ? 2 * (NT / (BN / 4)) * BN : 4];
  __shared__ int slab_of[256];
  __shared__ int nparts_s;
  const int rt = blockIdx.x / QS;
  const int qrow = (blockIdx.x % QS) * (BM / QS);  // first tile row of this workgroup
  const int ntn = (a.N + BN - 1) / BN;
  const int KT = a.KH * a.KW * ((a.Cs + (1 << a.ks_log2) - 1) >> a.ks_log2);
  const int tile = a.base * a.grid + rt;
  // blocks whose ranges hold the first and the last K-iteration of this tile
  const int u_first = rt * KT, u_last = u_first + KT - 1;
  const long long U = (long long)a.rem_tiles * KT;
  int b_first = (int)(((long long)u_first * a.sk_blocks) / U);
  while (sk_range_begin(a, KT, b_first + 1) <= u_first) ++b_first;
  while (sk_range_begin(a, KT, b_first) > u_first) --b_first;
  int b_last = (int)(((long long)u_last * a.sk_blocks) / U);
  while (sk_range_begin(a, KT, b_last + 1) <= u_last) ++b_last;
  while (sk_range_begin(a, KT, b_last) > u_last) --b_last;
  if (b_first == b_last) return;  // one block held the whole tile and finished it in place
  if (threadIdx.x == 0) {
    int n = 0;
    for (int b = b_first; b <= b_last && n < 256; ++b) {
      const int ub = sk_range_begin(a, KT, b);
      if (sk_range_begin(a, KT, b + 1) == ub) continue;
      slab_of[n++] = 2 * b + ((ub / KT == rt) ? 0 : 1);
    }
    nparts_s = n;
  }
  __syncthreads();
  const int nparts = nparts_s;
  TileCoord tc = tile_coord<BM, BN>(a, tile);
  tc.m0 += qrow;
  vec_epilogue<BM / QS, BN, NT>(
      a, tc, tile / ntn,
      [&](int row, int c4) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int part = 0; part < nparts; ++part)
          v += *reinterpret_cast<const f32x4*>(a.partial + (size_t)slab_of[part] * (BM * BN) + (qrow + row) * BN +
                                               c4 * 4);
        return v;
      },
      red);
}

// ------------------------------------------------------------------------------------------
// wgrad: split-K (over pixels) TN GEMM.  part[s][co][n2], n2 = tap*Cin + ci.
struct WgradSeg {
  const float* dy;  // [pix][Cout]
  const float* x;   // [Nb][Hx][Wx][Cin]
  int Nb, Hx, Wx;   // x dims
  int Ho, Wo;       // dy dims
  int npix;         // Nb*Ho*Wo
  int chunk_begin;  // first split index of this segment
};
struct WgradArgs {
  WgradSeg s[SACV_MAXP];
  int nseg;
  int nchunks;  // total splits
  int chunk;    // pixels per split (multiple of 32)
  float* part;  // [nchunks][Cout][N2]
  int Cin, Cout, N2;
  int KH, KW, stride, pad;
};

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void conv_tn_kernel(const WgradArgs a) {
  constexpr int RA = BM / 32, RB = BN / 32;  // float4 per thread per tile (32 pix rows x BM/4 col groups / 256)
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                // [2][32][BM]
  float* sB = smem + 2 * 32 * BM;  // [2][32][BN]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ntm = (a.Cout + BM - 1) / BM;
  const int ntn = (a.N2 + BN - 1) / BN;
  int bid = blockIdx.x;
  const int tm = bid % ntm;
  bid /= ntm;
  const int tn = bid % ntn;
  const int ck = bid / ntn;
  int si = 0;
  for (int i = 1; i < a.nseg; ++i)
    if (ck >= a.s[i].chunk_begin) si = i;
  const WgradSeg& S = a.s[si];
  const int pix0 = (ck - S.chunk_begin) * a.chunk;
  const int pix1 = min(pix0 + a.chunk, S.npix);
  const int m0 = tm * BM, n0 = tn * BN;
  const float* __restrict__ dy = S.dy;
  const float* __restrict__ x = S.x;

  // A' loader: thread -> (col group ca = tid % (BM/4), rows tid / (BM/4) + k*(256/(BM/4)))
  constexpr int CGA = BM / 4, RSA = 256 / CGA;  // col groups, row step
  constexpr int CGB = BN / 4, RSB = 256 / CGB;
  const int ca = tid % CGA, rra = tid / CGA;
  const int cb = tid % CGB, rrb = tid / CGB;
  const int co = m0 + ca * 4;
  const bool co_ok = co < a.Cout;  // Cout % 4 == 0 required
  const int n2 = n0 + cb * 4;
  const bool n2_ok = n2 < a.N2;
  int tap = 0, ci = 0, kh = 0, kw = 0;
  if (n2_ok) {
    tap = n2 / a.Cin;
    ci = n2 - tap * a.Cin;
    kh = tap / a.KW;
    kw = tap - kh * a.KW;
  }
  const int HoWo = S.Ho * S.Wo;
  const int Wo = S.Wo, Ho = S.Ho, Hx = S.Hx, Wx = S.Wx;
  const int dh = kh - a.pad, dw = kw - a.pad;

  // pixel coordinates of this thread's RB gather rows, advanced incrementally by 32 pixels per K tile
  int pn[RB], ph[RB], pw[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    int pix = pix0 + rrb + i * RSB;
    pn[i] = pix / HoWo;
    int rem = pix - pn[i] * HoWo;
    ph[i] = rem / Wo;
    pw[i] = rem - ph[i] * Wo;
  }

  f32x4 ra[RA], rb[RB];
  auto load_a = [&](int i, int p0, bool en) {
    int pix = p0 + rra + i * RSA;
    // bitwise & on purpose: && short-circuits into exec-masked branches inside the MFMA loop.  (`en` is implied:
    // the slab after the last one starts at or beyond pix1)
    bool ok = co_ok & (pix < pix1);
    // 32-bit element offsets (tensors are checked to be < 2^32 elements) and a pointer select: no 64-bit
    // multiply, no exec-masked branch in the MFMA loop
    const float* ptr = dy + (unsigned)(pix * a.Cout + co);
    ptr = ok ? ptr : sacv_zero16;
    ra[i] = *reinterpret_cast<const f32x4*>(ptr);
  };
  auto load_b = [&](int i, int p0, bool en) {
    int pix = p0 + rrb + i * RSB;
    int hi = ph[i] * a.stride + dh;
    int wi = pw[i] * a.stride + dw;
    bool ok = n2_ok & (pix < pix1) & ((unsigned)hi < (unsigned)Hx) & ((unsigned)wi < (unsigned)Wx);
    const float* ptr = x + ((unsigned)((pn[i] * Hx + hi) * Wx + wi) * (unsigned)a.Cin + (unsigned)ci);
    ptr = ok ? ptr : sacv_zero16;
    rb[i] = *reinterpret_cast<const f32x4*>(ptr);
  };
  // +32 pixels, branch-free: (dn, dhh, dww) = the wave-uniform decomposition of 32 in (image, row, column) units
  const int adv_n = 32 / HoWo;
  const int adv_h = (32 - adv_n * HoWo) / Wo;
  const int adv_w = 32 - adv_n * HoWo - adv_h * Wo;
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
  auto store_tile = [&](int buf) {
    float* dA = sA + buf * 32 * BM;
    float* dB = sB + buf * 32 * BN;
#pragma unroll
    for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(dA + (rra + i * RSA) * BM + ca * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(dB + (rrb + i * RSB) * BN + cb * 4) = rb[i];
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

  const int KT = (pix1 - pix0 + 31) >> 5;
  if (KT > 0) {
#pragma unroll
    for (int i = 0; i < RA; ++i) load_a(i, pix0, true);
#pragma unroll
    for (int i = 0; i < RB; ++i) load_b(i, pix0, true);
    store_tile(0);
  }
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < KT; ++kt) {
    const bool have_next = kt + 1 < KT;
    const int pnext = pix0 + ((kt + 1) << 5);
    advance();
    // Fragment reads: the wave's two 32-row MFMA blocks are INTERLEAVED in the LDS row (row 2i+b of the wave's
    // 64-row range belongs to block b), so one 8-byte ds_read_b64 per lane fetches the operands of both blocks
    // (64 contiguous dwords per half-wave: conflict-free) -- half the LDS instructions of two ds_read_b32.
    const float* cA = sA + cur * 32 * BM + wm0 + (TM == 2 ? 2 * l31 : l31);
    const float* cB = sB + cur * 32 * BN + wn0 + (TN == 2 ? 2 * l31 : l31);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto read_a = [&](float (&dst)[TM], int krow) {
      if (TM == 2) {
        f32x2 v = *reinterpret_cast<const f32x2*>(cA + krow * BM);
        dst[0] = v[0];
        dst[TM - 1] = v[1];
      } else {
        dst[0] = cA[krow * BM];
      }
    };
    auto read_b = [&](float (&dst)[TN], int krow) {
      if (TN == 2) {
        f32x2 v = *reinterpret_cast<const f32x2*>(cB + krow * BN);
        dst[0] = v[0];
        dst[TN - 1] = v[1];
      } else {
        dst[0] = cB[krow * BN];
      }
    };
    float fa[2][TM], fb[2][TN];
    read_a(fa[0], lh);
    read_b(fb[0], lh);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int cbuf = s & 1, nbuf = cbuf ^ 1;
      if (s < 15) {
        read_a(fa[nbuf], 2 * (s + 1) + lh);
        read_b(fb[nbuf], 2 * (s + 1) + lh);
      }
      if ((s & 3) == 0 && (s >> 2) < RA) load_a(s >> 2, pnext, have_next);
      if ((s & 3) == 2 && (s >> 2) < RB) load_b(s >> 2, pnext, have_next);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cbuf][i], fb[cbuf][j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < TM * TN; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (m < 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        if (m == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    if (have_next) store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  // epilogue: accumulators -> LDS [BM][BN+4] -> row-major slab with 16-byte stores
  constexpr int CS = BN + 4;
  float* stage = smem;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ri = (r & 3) + 8 * (r >> 2) + 4 * lh;  // row of the 32x32 MFMA block held in register r
        const int row = wm0 + (TM == 2 ? 2 * ri + i : ri);
        const int col = wn0 + (TN == 2 ? 2 * l31 + j : l31);
        stage[row * CS + col] = acc[i][j][r];
      }
  __syncthreads();
  float* __restrict__ part = a.part + (size_t)ck * a.Cout * a.N2;
  constexpr int CG = BN / 4, RS = 256 / CG;
  const int c4 = tid % CG, rr = tid / CG;
  const int n = n0 + c4 * 4;
  if (n < a.N2) {
#pragma unroll 4
    for (int row = rr; row < BM; row += RS) {
      int m = m0 + row;
      if (m < a.Cout)
        *reinterpret_cast<f32x4*>(part + (size_t)m * a.N2 + n) =
            *reinterpret_cast<const f32x4*>(stage + row * CS + c4 * 4);
    }
  }
}

// The weight-gradient GEMM on the bf16 pipe (split operands, see conv_nt_kernel SP): the gathered fp32 values of dy and
// x are split into three bf16 terms when they are stored to LDS (global traffic unchanged) and the six products per K
// step run on v_mfma_f32_32x32x16_bf16.  The contraction index is the PIXEL, the slow index of both operands, so the
// LDS images are [KR pixel rows][channels] bf16 per plane and the fragments come from the transposing LDS read
// (ds_read_b64_tr_b16; conv_tn_bf16_kernel of conv_bf16.hip has the lane mapping), rows padded by 64 bytes.  KR = 32
// pixel rows per tile for the 64 x 64 item (73.7 KB double-buffered, two workgroups per CU), 16 for the wider items.
typedef short cs16x4 __attribute__((ext_vector_type(4)));

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void conv_tn3_kernel(const WgradArgs a) {
  constexpr int KR = (BM + BN <= 128) ? 32 : 16;          // pixel rows per K tile
  constexpr int RSA_B = BM * 2 + 64, RSB_B = BN * 2 + 64;  // LDS row strides in bytes
  constexpr int CGA = BM / 4, RSA = 256 / CGA;            // f32x4 column groups of a row, rows per pass
  constexpr int CGB = BN / 4, RSB = 256 / CGB;
  constexpr int RA = KR / RSA, RB = KR / RSB;             // f32x4 per thread and tile
  static_assert(RA >= 1 && RB >= 1, "a tile row pass per operand");
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int IMG_A = 3 * KR * RSA_B, IMG_B = 3 * KR * RSB_B;  // bytes of one buffer [3 planes][KR][row]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  unsigned char* sA = smem3;              // [2][IMG_A]
  unsigned char* sB = smem3 + 2 * IMG_A;  // [2][IMG_B]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ntm = (a.Cout + BM - 1) / BM;
  const int ntn = (a.N2 + BN - 1) / BN;
  int bid = blockIdx.x;
  const int tm = bid % ntm;
  bid /= ntm;
  const int tn = bid % ntn;
  const int ck = bid / ntn;
  int si = 0;
  for (int i = 1; i < a.nseg; ++i)
    if (ck >= a.s[i].chunk_begin) si = i;
  const WgradSeg& S = a.s[si];
  const int pix0 = (ck - S.chunk_begin) * a.chunk;
  const int pix1 = min(pix0 + a.chunk, S.npix);
  const int m0 = tm * BM, n0 = tn * BN;
  const float* __restrict__ dy = S.dy;
  const float* __restrict__ x = S.x;

  const int ca = tid % CGA, rra = tid / CGA;
  const int cb = tid % CGB, rrb = tid / CGB;
  const int co = m0 + ca * 4;
  const bool co_ok = co < a.Cout;
  const int n2 = n0 + cb * 4;
  const bool n2_ok = n2 < a.N2;
  int tap = 0, ci = 0, kh = 0, kw = 0;
  if (n2_ok) {
    tap = n2 / a.Cin;
    ci = n2 - tap * a.Cin;
    kh = tap / a.KW;
    kw = tap - kh * a.KW;
  }
  const int HoWo = S.Ho * S.Wo;
  const int Wo = S.Wo, Ho = S.Ho, Hx = S.Hx, Wx = S.Wx;
  const int dh = kh - a.pad, dw = kw - a.pad;
  int pn[RB], ph[RB], pw[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    int pix = pix0 + rrb + i * RSB;
    pn[i] = pix / HoWo;
    int rem = pix - pn[i] * HoWo;
    ph[i] = rem / Wo;
    pw[i] = rem - ph[i] * Wo;
  }
  f32x4 ra[RA], rb[RB];
  auto load_a = [&](int i, int p0) {
    int pix = p0 + rra + i * RSA;
    bool ok = co_ok & (pix < pix1);
    const float* ptr = dy + (unsigned)(pix * a.Cout + co);
    ptr = ok ? ptr : sacv_zero16;
    ra[i] = *reinterpret_cast<const f32x4*>(ptr);
  };
  auto load_b = [&](int i, int p0) {
    int pix = p0 + rrb + i * RSB;
    int hi = ph[i] * a.stride + dh;
    int wi = pw[i] * a.stride + dw;
    bool ok = n2_ok & (pix < pix1) & ((unsigned)hi < (unsigned)Hx) & ((unsigned)wi < (unsigned)Wx);
    const float* ptr = x + ((unsigned)((pn[i] * Hx + hi) * Wx + wi) * (unsigned)a.Cin + (unsigned)ci);
    ptr = ok ? ptr : sacv_zero16;
    rb[i] = *reinterpret_cast<const f32x4*>(ptr);
  };
  const int adv_n = KR / HoWo;
  const int adv_h = (KR - adv_n * HoWo) / Wo;
  const int adv_w = KR - adv_n * HoWo - adv_h * Wo;
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
  // 4 consecutive channels of one pixel row -> three 8-byte stores (planes hi, mid, lo)
  auto put3 = [&](unsigned char* img, int rs, int row, int col, const f32x4 v) {
    cu32x2 h, m, l;
    split3_quad(v, h, m, l);
    unsigned char* d = img + row * rs + col * 2;
    *reinterpret_cast<cu32x2*>(d) = h;
    *reinterpret_cast<cu32x2*>(d + KR * rs) = m;
    *reinterpret_cast<cu32x2*>(d + 2 * KR * rs) = l;
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < RA; ++i) put3(sA + buf * IMG_A, RSA_B, rra + i * RSA, ca * 4, ra[i]);
#pragma unroll
    for (int i = 0; i < RB; ++i) put3(sB + buf * IMG_B, RSB_B, rrb + i * RSB, cb * 4, rb[i]);
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
  const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int tr_row = 8 * (tg >> 1) + tq;      // + 4 for the second read, + 16 per K step
  const int tr_col = 16 * (tg & 1) + 4 * tp;  // channel inside the 32-wide MFMA block

  const int KT = (pix1 - pix0 + KR - 1) / KR;
  if (KT > 0) {
#pragma unroll
    for (int i = 0; i < RA; ++i) load_a(i, pix0);
#pragma unroll
    for (int i = 0; i < RB; ++i) load_b(i, pix0);
    store_tile(0);
  }
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < KT; ++kt) {
    const bool have_next = kt + 1 < KT;
    const int pnext = pix0 + (kt + 1) * KR;  // beyond pix1 after the last tile: every load is masked
    advance();
#pragma unroll
    for (int i = 0; i < RA; ++i) load_a(i, pnext);
#pragma unroll
    for (int i = 0; i < RB; ++i) load_b(i, pnext);
    const unsigned char* cA = sA + cur * IMG_A + tr_row * RSA_B + (wm0 + tr_col) * 2;
    const unsigned char* cB = sB + cur * IMG_B + tr_row * RSB_B + (wn0 + tr_col) * 2;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < KR / 16; ++kk) {
      cbf16x8 fa[3][TM], fb[3][TN];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const unsigned char* pA = cA + (pl * KR + kk * 16) * RSA_B + i * 64;
          union { cs16x4 s[2]; cbf16x8 v; } u;
          u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) cs16x4*)(pA));
          u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) cs16x4*)(pA + 4 * RSA_B));
          fa[pl][i] = u.v;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const unsigned char* pB = cB + (pl * KR + kk * 16) * RSB_B + j * 64;
          union { cs16x4 s[2]; cbf16x8 v; } u;
          u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) cs16x4*)(pB));
          u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) cs16x4*)(pB + 4 * RSB_B));
          fb[pl][j] = u.v;
        }
      }
      constexpr int pa[6] = {0, 1, 0, 1, 2, 0};
      constexpr int pb[6] = {0, 0, 1, 1, 0, 2};
#pragma unroll
      for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pa[t6]][i], fb[pb[t6]][j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (have_next) store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  constexpr int CS = BN + 4;
  float* stage = reinterpret_cast<float*>(smem3);
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
      int m = m0 + row;
      if (m < a.Cout)
        *reinterpret_cast<f32x4*>(part + (size_t)m * a.N2 + n) =
            *reinterpret_cast<const f32x4*>(stage + row * CS + c4 * 4);
    }
  }
}

// source element (co, ci, tap) of a weight stored OIHW (src_ohwi = 0) or OHWI = channels_last (src_ohwi = 1)
__device__ __forceinline__ size_t w_src_index(int co, int ci, int tap, int Cin, int T, int src_ohwi) {
  return src_ohwi ? ((size_t)co * T + tap) * Cin + ci : ((size_t)co * Cin + ci) * T + tap;
}

// dw[co][ci][kh][kw] (OIHW; or [co][kh][kw][ci] when dw_ohwi) (+)= sum_s part[s][co][(kh*KW+kw)*Cin_pad + ci]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                            int nchunks, int Cout, int Cin, int Cin_pad, int T,
                                                            int accumulate, int dw_ohwi) {
  // 64 consecutive slab elements per block, 4 threads per element each summing every 4th chunk (coalesced
  // 256-byte rows), fixed-order combine through LDS: short serial chains even for hundreds of chunks.
  __shared__ float red[256];
  const int N2 = T * Cin_pad;
  const size_t total = (size_t)Cout * N2;
  const int el = threadIdx.x & 63, part_id = threadIdx.x >> 6;
  for (size_t base = (size_t)blockIdx.x * 64; base < total; base += (size_t)gridDim.x * 64) {
    size_t idx = base + el;
    float s = 0.f;
    if (idx < total)
      for (int c = part_id; c < nchunks; c += 4) s += part[(size_t)c * total + idx];
    red[threadIdx.x] = s;
    __syncthreads();
    if (part_id == 0 && idx < total) {
      float t = (red[el] + red[64 + el]) + (red[128 + el] + red[192 + el]);
      int co = (int)(idx / N2);
      int n2 = (int)(idx - (size_t)co * N2);
      int tap = n2 / Cin_pad;
      int ci = n2 - tap * Cin_pad;
      if (ci < Cin) {
        size_t o = w_src_index(co, ci, tap, Cin, T, dw_ohwi);
        dw[o] = accumulate ? dw[o] + t : t;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// weight repacks from OIHW [Cout][Cin][KH][KW]
//   fwd  panel: wf[co][tap*Cin + ci]            (K = T*Cin contiguous)
//   dgrad panel: wd[ci][tap*Cout + co]           (K = T*Cout contiguous)
__global__ void pack_w_fwd_kernel(const float* __restrict__ w, float* __restrict__ wf, int Cout, int Cin, int T) {
  size_t total = (size_t)Cout * Cin * T;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    int ci = (int)(idx % Cin);
    size_t q = idx / Cin;
    int tap = (int)(q % T);
    int co = (int)(q / T);
    wf[idx] = w[((size_t)co * Cin + ci) * T + tap];
  }
}
__global__ void pack_w_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wd, int Cout, int Cin, int T,
                                    int src_ohwi) {
  size_t total = (size_t)Cout * Cin * T;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    int co = (int)(idx % Cout);
    size_t q = idx / Cout;
    int tap = (int)(q % T);
    int ci = (int)(q / T);
    wd[idx] = w[w_src_index(co, ci, tap, Cin, T, src_ohwi)];
  }
}
// all data-gradient panels of a model in ONE launch: chunk_map[b] = (descriptor, tile index), tiles enumerated
// as (tap, ci block of 32, co block of 32) with co fastest
struct PackDesc {
  const float* src;
  float* dst;
  int Cout, Cin, T, src_ohwi;
};
__global__ __launch_bounds__(256) void pack_dgrad_batched_kernel(const PackDesc* __restrict__ descs,
                                                                  const int* __restrict__ chunk_map) {
  // one 32(co) x 32(ci) tile of one tap per block, transposed through LDS: reads run along ci (contiguous in a
  // channels_last / 1x1 weight), writes along co (contiguous in the panel)
  __shared__ float tile[32][33];
  const int di = chunk_map[2 * blockIdx.x], t = chunk_map[2 * blockIdx.x + 1];
  const PackDesc d = descs[di];
  const int nci = (d.Cin + 31) >> 5, nco = (d.Cout + 31) >> 5;
  const int cob = t % nco;
  const int q = t / nco;
  const int cib = q % nci;
  const int tap = q / nci;
  const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int co = cob * 32 + r + 8 * rr, ci = cib * 32 + c;
    tile[r + 8 * rr][c] = (co < d.Cout && ci < d.Cin) ? d.src[w_src_index(co, ci, tap, d.Cin, d.T, d.src_ohwi)] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int ci = cib * 32 + r + 8 * rr, co = cob * 32 + c;
    if (ci < d.Cin && co < d.Cout) d.dst[((size_t)ci * d.T + tap) * d.Cout + co] = tile[c][r + 8 * rr];
  }
}
// stem panel: wf[co][k], k = (kh*KW+kw)*Cin + ci for k < T*Cin, zero up to Kpad
__global__ void pack_w_stem_kernel(const float* __restrict__ w, float* __restrict__ wf, int Cout, int Cin, int T,
                                   int Kpad) {
  int total = Cout * Kpad;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    int k = idx % Kpad, co = idx / Kpad;
    float v = 0.f;
    if (k < T * Cin) {
      int tap = k / Cin, ci = k - tap * Cin;
      v = w[((size_t)co * Cin + ci) * T + tap];
    }
    wf[idx] = v;
  }
}

// explicit im2col for the 7x7/s2 stem (Cin = 3 is too narrow for 16-byte gathers):
// col[m][k], k = (kh*KW+kw)*Cin + ci, zero padded to Kpad.  Input given by element strides so
// NCHW-contiguous and channels_last images are both accepted (SURVEY.md finding 3).
__global__ __launch_bounds__(256) void im2col_stem_kernel(const float* __restrict__ x, float* __restrict__ col, int Nb,
                                                          int Cin, int H, int W, long long sN, long long sC,
                                                          long long sH, long long sW, int Ho, int Wo, int KH, int KW,
                                                          int stride, int pad, int Kpad) {
  // a thread owns ONE group of 4 consecutive k (its k -> (kh, kw, ci) decode is done once) and walks output pixels;
  // the block is G = Kpad / 4 groups x RPB pixels, so its 16-byte stores cover contiguous rows of the matrix
  const int G = Kpad >> 2;
  const int RPB = 256 / G;
  const int g = threadIdx.x % G, pr = threadIdx.x / G;
  if (pr >= RPB) return;
  const int K = KH * KW * Cin;
  int kh[4], kw[4];
  long long koff[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int k = 4 * g + e;
    if (k < K) {
      const int tap = k / Cin, ci = k - tap * Cin;
      kh[e] = tap / KW;
      kw[e] = tap - kh[e] * KW;
      koff[e] = ci * sC + kh[e] * sH + kw[e] * sW;
    } else {
      kh[e] = -100000;  // never inside the image
      kw[e] = 0;
      koff[e] = 0;
    }
  }
  const size_t npix = (size_t)Nb * Ho * Wo;
  for (size_t m = (size_t)blockIdx.x * RPB + pr; m < npix; m += (size_t)gridDim.x * RPB) {
    const int wo = (int)(m % Wo);
    const size_t q = m / Wo;
    const int ho = (int)(q % Ho);
    const int n = (int)(q / Ho);
    const int hb = ho * stride - pad, wb = wo * stride - pad;
    const float* __restrict__ src = x + n * sN + hb * sH + wb * sW;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool ok = ((unsigned)(hb + kh[e]) < (unsigned)H) & ((unsigned)(wb + kw[e]) < (unsigned)W);
      v[e] = ok ? src[koff[e]] : 0.f;
    }
    *reinterpret_cast<f32x4*>(col + (m * G + g) * 4) = v;
  }
}

// stem: part[s][co][k], k = tap*Cin + ci (k < T*Cin), row length Kpad -> dw[co][ci][tap]
__global__ __launch_bounds__(256) void wgrad_reduce_stem_kernel(const float* __restrict__ part,
                                                                 float* __restrict__ dw, int nchunks, int Cout,
                                                                 int Cin, int T, int Kpad, int accumulate) {
  // 16 consecutive elements per block, 16 threads per element each summing every 16th slab; fixed-order combine
  // through LDS (hundreds of slabs: a single thread per element was a 512-deep serial chain of loads)
  __shared__ float red[256];
  const size_t slab = (size_t)Cout * Kpad;
  const int total = Cout * T * Cin;
  const int el = threadIdx.x & 15, lane = threadIdx.x >> 4;
  for (int base = blockIdx.x * 16; base < total; base += gridDim.x * 16) {
    const int idx = base + el;
    float s = 0.f;
    int k = 0, co = 0;
    if (idx < total) {
      k = idx % (T * Cin);
      co = idx / (T * Cin);
      for (int c = lane; c < nchunks; c += 16) s += part[(size_t)c * slab + (size_t)co * Kpad + k];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (lane == 0 && idx < total) {
      float t = 0.f;
      for (int l = 0; l < 16; ++l) t += red[l * 16 + el];
      const int tap = k / Cin, ci = k - tap * Cin;
      const size_t o = ((size_t)co * Cin + ci) * T + tap;
      dw[o] = accumulate ? dw[o] + t : t;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// host side
static int log2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return ((1 << l) == v) ? l : -1;
}

static int num_cus() {
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

#ifndef SACV_NT_BLOCKS_PER_CU
#define SACV_NT_BLOCKS_PER_CU 2
#endif
#ifndef SACV_NT_WN
#define SACV_NT_WN 2  // waves along N of the 128x128 kernel: 2 -> 4 waves of 64x64, 4 -> 8 waves of 64x32
#endif

struct NtPlan {
  int base, rem, sk_blocks;
  double cost;  // in units of one 128x128 K-iteration
};

#ifndef SACV_SK_POLICY
#define SACV_SK_POLICY 3  // tuning builds: -DSACV_SK_POLICY=n (scripts/), never an environment variable
#endif
#ifndef SACV_SK_EXTRA
#define SACV_SK_EXTRA 6.0
#endif
#ifndef SACV_EPI_SPLIT
#define SACV_EPI_SPLIT 7.0
#endif
#ifndef SACV_SK_MIN_ITERS
#define SACV_SK_MIN_ITERS 4  // a stream-K range shorter than this is not worth its slab + fix-up
#endif

// Schedule of `tiles` equal tiles of KT K-iterations on `grid` persistent workgroups.  w = relative cost of
// one K-iteration of this tile shape (1.0 for 128x128, ~0.6 for 128x64).  The split-K workspace holds two
// slabs per workgroup (sacv_conv2d_gemm_workspace_bytes).
// `ks` scales the per-tile constants (epilogue, slab + fix-up, stream-K extra, shortest range): they are given in
// K-iterations of the fp32-pipe kernels; a K-iteration of a split-operand kernel is shorter, its epilogue is not.
static NtPlan plan_nt(int tiles, int KT, int grid, bool has_ws, double w, double ks = 1.0) {
  const double epi = 3.0 * ks, epi_split = SACV_EPI_SPLIT * ks;  // epilogue / slab write + fix-up, in K-iterations
  NtPlan p;
  p.base = tiles / grid;
  p.rem = tiles - p.base * grid;
  p.sk_blocks = 0;
  if (p.rem == 0) {
    p.cost = p.base * (KT + epi) * w;
    return p;
  }
  // candidates for the remainder tiles (cost in K-iterations per workgroup):
  //   one workgroup per tile                      KT + epi
  //   uniform split, j slices per tile (rem*j<=grid)  ceil(KT/j) + epi_split
  //   stream-K over all workgroups                ceil(rem*KT/grid) + epi_split + sk_extra
  // Stream-K equalises the work of every CU but costs a second (partial) epilogue in the workgroups whose
  // range crosses a tile boundary and more slabs per tile in the fix-up: measured ~+12 us per launch
  // (~6 K-iterations) over a uniform split, so it wins on the long-K problems (3x3 convs, head towers).
  const int policy = sacv_share.split_policy == 0 ? 0 : SACV_SK_POLICY;  // 0 none, 1 uniform only, 2 stream-K only, 3 auto
  const double sk_extra = SACV_SK_EXTRA * ks;
  const long long sk_min_iters = (long long)(SACV_SK_MIN_ITERS * ks + 0.5);
  p.sk_blocks = p.rem;
  double tail = KT + epi;
  if (has_ws && policy > 0) {
    if (policy != 2) {
      int max_j = KT / 4 > 0 ? KT / 4 : 1;
      if (max_j > 64) max_j = 64;
      for (int j = 2; j <= max_j && (long long)p.rem * j <= grid; ++j) {
        const double c = (KT + j - 1) / j + epi_split;
        if (c < tail) {
          tail = c;
          p.sk_blocks = p.rem * j;
        }
      }
    }
    if (policy != 1) {
      const long long U = (long long)p.rem * KT;
      long long gs = grid;
      if (U / gs < sk_min_iters) gs = U / sk_min_iters;
      if (gs < p.rem) gs = p.rem;
      const double c = (double)((U + gs - 1) / gs) + epi_split + sk_extra;
      if (gs > p.rem && (c < tail || policy == 2)) {
        tail = c;
        p.sk_blocks = (int)gs;
      }
    }
  }
  p.cost = (p.base * (KT + epi) + tail) * w;
  return p;
}

// persistent workgroup slots a launch of the tile shape plans for (sacv_launch_share)
static int nt_slots(int bn) {
  const int per_cu = bn == 64 ? sacv_share.nt64_wg_per_cu : sacv_share.nt128_wg_per_cu;
  int cus = num_cus() - sacv_share.cu_reserve;
  if (cus < 8) cus = 8;
  return cus * per_cu;
}

// the split-operand form of the 128 x 64 kernel: 74 KB of LDS, two workgroups per CU
static int nt_slots_split() {
  const int per_cu = sacv_share.nt64_wg_per_cu < 2 ? sacv_share.nt64_wg_per_cu : 2;
  int cus = num_cus() - sacv_share.cu_reserve;
  if (cus < 8) cus = 8;
  return cus * per_cu;
}

template <int BM, int BN, int MODE, int WN, int OCC = SACV_NT_BLOCKS_PER_CU, bool SWZ = false, bool SP = false, int KS = 32>
static int launch_nt(ConvArgs& a, const NtPlan& plan, float* workspace, hipStream_t st) {
  a.ks_log2 = KS == 32 ? 5 : 4;
  int grid = SP ? nt_slots_split() : nt_slots(BN);
  a.grid = grid;
  a.base = plan.base;
  a.rem_tiles = plan.rem;
  a.sk_blocks = plan.sk_blocks;
  a.partial = workspace;
  size_t lds = SP ? (size_t)2 * 3 * (BM + BN) * (KS / 2) * sizeof(float) : (size_t)2 * (BM + BN) * (SWZ ? 32 : 36) * sizeof(float);
  size_t lds_epi = ((size_t)BM * (BN + 4) + 2 * (128 * WN / (BN / 4)) * BN) * sizeof(float);
  if (lds_epi > lds) lds = lds_epi;
  static std::once_flag attr_once;  // one per template instance; lds is a compile-time function of it
  std::call_once(attr_once, [lds] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_nt_kernel<BM, BN, MODE, WN, OCC, SWZ, SP, KS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  int launch_blocks = a.base > 0 ? grid : a.sk_blocks;
  SACV_CHECK_ARG(a.sk_blocks <= grid && (a.sk_blocks == 0 || a.sk_blocks >= a.rem_tiles),
                 "conv2d_gemm: stream-K plan outside the workspace (%d ranges)", a.sk_blocks);
  hipLaunchKernelGGL((conv_nt_kernel<BM, BN, MODE, WN, OCC, SWZ, SP, KS>), dim3(launch_blocks), dim3(128 * WN), lds, st, a);
  SACV_CHECK_LAUNCH("conv_nt_kernel");
  if (a.sk_blocks > a.rem_tiles) {
    if (a.flags & (CF_STATS | CF_BNBWD))
      hipLaunchKernelGGL((conv_fixup_kernel<BM, BN>), dim3(a.rem_tiles), dim3(1024), 0, st, a);
    else
      hipLaunchKernelGGL((conv_fixup_kernel<BM, BN, 256, 4>), dim3(a.rem_tiles * 4), dim3(256), 0, st, a);
    SACV_CHECK_LAUNCH("conv_fixup_kernel");
  }
  return SACV_OK;
}

template <int BM, int BN>
static int launch_tn(const WgradArgs& a, int chunks, hipStream_t st) {
  int ntm = (a.Cout + BM - 1) / BM, ntn = (a.N2 + BN - 1) / BN;
  if (sacv_share.nt_split) {
    constexpr int KR = (BM + BN <= 128) ? 32 : 16;
    size_t lds3 = (size_t)2 * 3 * KR * ((BM * 2 + 64) + (BN * 2 + 64));
    const size_t epi = (size_t)BM * (BN + 4) * sizeof(float);
    if (epi > lds3) lds3 = epi;
    static std::once_flag attr3;
    std::call_once(attr3, [lds3] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tn3_kernel<BM, BN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
    });
    hipLaunchKernelGGL((conv_tn3_kernel<BM, BN>), dim3(ntm * ntn * chunks), dim3(256), lds3, st, a);
    SACV_CHECK_LAUNCH("conv_tn3_kernel");
    return SACV_OK;
  }
  size_t lds = (size_t)2 * 32 * (BM + BN) * sizeof(float);
  size_t lds_epi = (size_t)BM * (BN + 4) * sizeof(float);
  if (lds_epi > lds) lds = lds_epi;
  static std::once_flag attr_once;
  std::call_once(attr_once, [lds] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tn_kernel<BM, BN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  hipLaunchKernelGGL((conv_tn_kernel<BM, BN>), dim3(ntm * ntn * chunks), dim3(256), lds, st, a);
  SACV_CHECK_LAUNCH("conv_tn_kernel");
  return SACV_OK;
}

// Tile of the weight-gradient GEMM dW[Cout][N2 = taps * Cin]: 128 x 128 (64 where a dimension is that narrow); the 1x1
// layers take 64 x 64 tiles whatever their widths -- their K (the pixels) is all there is to split, and a 128 x 128 tile
// cut 32-64 ways leaves a workgroup 4-8 K iterations per 64 KB slab it writes: with 16 KB slabs and four times the K
// per workgroup the same launches take 51-54 us instead of 57-62 (layer2-4 at 2 x 800 x 1333; scripts/bench_tn_tiles.py).
static void wgrad_tile(int Cout, int N2, int taps, int* bm, int* bn) {
  *bm = (Cout <= 64 || taps == 1) ? 64 : 128;
  *bn = (N2 <= 64 || taps == 1) ? 64 : 128;
  // the split-operand kernel (bf16 pipe) takes the wide tile for the 1x1 layers too: its matrix work per K tile is 2.7x
  // cheaper, what is left is slab traffic and barriers per tile (same-session A/B, headline step: 23.19 -> 23.04 ms)
  if (sacv_share.nt_split) {
    *bm = Cout <= 64 ? 64 : 128;
    *bn = N2 <= 64 ? 64 : 128;
  }
}

int sacv_internal_wgrad_reduce(const float* part, float* dw, int chunks, int Cout, int Cin, int Cin_pad, int KH,
                               int KW, int accumulate, int dw_taps, int dw_layout, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int T = dw_taps > 0 ? dw_taps : KH * KW;
  if (dw_taps > 0) {
    // stem (KH = KW = 1, Cin_pad = Kpad): part is [Cout][Kpad] with k = tap*Cin + ci -> dw [Cout][Cin][dw_taps]
    const int blocks = (Cout * dw_taps * Cin + 15) / 16;
    hipLaunchKernelGGL(wgrad_reduce_stem_kernel, dim3(blocks), dim3(256), 0, st, part, dw, chunks, Cout, Cin, dw_taps,
                       Cin_pad, accumulate);
  } else {
    const size_t total = (size_t)Cout * KH * KW * Cin_pad;
    int blocks = (int)((total + 63) / 64);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, part, dw, chunks, Cout, Cin, Cin_pad, T,
                       accumulate, dw_layout);
  }
  SACV_CHECK_LAUNCH("wgrad_reduce_kernel");
  return SACV_OK;
}

SACV_API_BEGIN

// The BatchNorm layer a data-gradient launch computes the backward partial sums of (CF_BNBWD)
struct BnFuse {
  const float *y, *z, *mean, *invstd, *gamma, *beta;
};

// Grouped implicit-GEMM convolution (forward when mode==0, data-gradient when mode==1).
// Every problem p: src[p] is NHWC [Nb[p]][Hs[p]][Ws[p]][Cs]; dst[p] is [M_p][N] with
// M_p = Nb*Hd*Wd.  w is the packed panel [N][KH*KW*Cs].
static int conv2d_gemm_impl(int nprob, const float* const* src, float* const* dst, const float* const* mask,
                            const int* Nb, const int* Hs, const int* Ws, const int* Hd, const int* Wd, const float* w,
                            const float* bias, int Cs, int N, int KH, int KW, int stride, int pad, int mode, int flags,
                            int remap_s, int Hr, int Wr, const int* ext, float* workspace, float* stats, void* stream,
                            const BnFuse* bn) {
  SACV_CHECK_ARG(nprob >= 1 && nprob <= SACV_MAXP, "conv2d_gemm: nprob %d out of range", nprob);
  SACV_CHECK_ARG(Cs % 4 == 0 && Cs > 0, "conv2d_gemm: Cs=%d must be a positive multiple of 4", Cs);
  SACV_CHECK_ARG(N > 0 && KH > 0 && KW > 0, "conv2d_gemm: bad N/KH/KW");
  int sl = log2_exact(stride);
  SACV_CHECK_ARG(sl >= 0, "conv2d_gemm: stride %d must be a power of two", stride);
  SACV_CHECK_ARG(mode == 0 || mode == 1, "conv2d_gemm: mode");
  SACV_CHECK_ARG(!(flags & CF_BIAS) || bias, "conv2d_gemm: bias flag without bias pointer");
  SACV_CHECK_ARG(!(flags & CF_AFFINE) || (flags & CF_BIAS), "conv2d_gemm: SACV_CF_AFFINE needs SACV_CF_BIAS (bias = [2][N])");
  SACV_CHECK_ARG(remap_s == 0 || nprob == 1, "conv2d_gemm: remap only for single problems");
  SACV_CHECK_ARG((long long)N * KH * KW * Cs < (1LL << 30), "conv2d_gemm: weight panel too large");
  SACV_CHECK_ARG(N % 4 == 0, "conv2d_gemm: N=%d must be a multiple of 4", N);
  SACV_CHECK_ARG(!(flags & CF_STATS) || stats, "conv2d_gemm: CF_STATS without a stats buffer");
#ifndef SACV_ABLATE
  SACV_CHECK_ARG((flags & ~CF_PUBLIC_MASK) == 0, "conv2d_gemm: unknown flag bits 0x%x", flags & ~CF_PUBLIC_MASK);
#endif
  ConvArgs a;
  const int BM = 128;
  int tiles = 0;
  for (int i = 0; i < nprob; ++i) {
    SACV_CHECK_ARG(src[i] && dst[i], "conv2d_gemm: null tensor pointer (problem %d)", i);
    SACV_CHECK_ARG(Nb[i] > 0 && Hs[i] > 0 && Ws[i] > 0 && Hd[i] > 0 && Wd[i] > 0, "conv2d_gemm: bad dims");
    SACV_CHECK_ARG(!(flags & CF_MASK) || (mask && mask[i]), "conv2d_gemm: mask flag without mask pointer");
    long long M = (long long)Nb[i] * Hd[i] * Wd[i];
    SACV_CHECK_ARG(M * (long long)(N > Cs ? N : Cs) < (1LL << 30) &&
                       (long long)Nb[i] * Hs[i] * Ws[i] * Cs < (1LL << 30),
                   "conv2d_gemm: tensor too large for 32-bit element indexing (2^30 elements)");
    ConvProb& p = a.p[i];
    p.src = src[i];
    p.dst = dst[i];
    p.mask = mask ? mask[i] : nullptr;
    p.Nb = Nb[i];
    p.Hs = Hs[i];
    p.Ws = Ws[i];
    p.Hd = Hd[i];
    p.Wd = Wd[i];
    p.M = (int)M;
    p.tile_begin = tiles;
    p.remap_s = remap_s;
    p.Hr = Hr;
    p.Wr = Wr;
    p.remap_oh = ext ? ext[9] : 0;
    p.remap_ow = ext ? ext[10] : 0;
    tiles += (p.M + BM - 1) / BM;
  }
  a.nprob = nprob;
  a.total_mtiles = tiles;
  a.w = w;
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
    // ext = {pad_h, pad_w, kh0, khs, kw0, kws, KHreal, KWreal, 0, remap_oh, remap_ow, 0}
    a.pad_h = ext[0];
    a.pad_w = ext[1];
    a.kh0 = ext[2];
    a.khs = ext[3];
    a.kw0 = ext[4];
    a.kws = ext[5];
    a.KWr = ext[7];
    a.K = ext[6] * ext[7] * Cs;
    SACV_CHECK_ARG(a.kh0 + a.khs * (KH - 1) < ext[6] && a.kw0 + a.kws * (KW - 1) < ext[7],
                   "conv2d_gemm: tap subset exceeds the weight panel");
  }
  a.mode = mode;
  a.flags = flags;
  a.stats = stats;
  a.bn_y = a.bn_z = a.bn_mean = a.bn_invstd = a.bn_gamma = a.bn_beta = nullptr;
  if (bn) {
    SACV_CHECK_ARG(nprob == 1 && remap_s == 0 && !(flags & CF_STATS) && stats,
                   "conv2d_gemm_bnbwd: one problem, no row remap, no SACV_CF_STATS, a stats buffer");
    SACV_CHECK_ARG(bn->y && bn->mean && bn->invstd && (bn->z || !bn->beta || bn->gamma),
                   "conv2d_gemm_bnbwd: y / mean / invstd (and gamma with beta) are required");
    a.flags |= CF_BNBWD;
    a.bn_y = bn->y;
    a.bn_z = bn->z;
    a.bn_mean = bn->mean;
    a.bn_invstd = bn->invstd;
    a.bn_gamma = bn->gamma;
    a.bn_beta = bn->beta;
  }
  hipStream_t st = (hipStream_t)stream;
  const int grid = nt_slots(128);
  const int KT = KH * KW * ((Cs + 31) / 32);
  const bool ws = workspace != nullptr;
  NtPlan p128 = plan_nt(tiles * ((N + 127) / 128), KT, grid, ws, 1.0);
  // 128 x 64 tiles run at THREE workgroups per CU (swizzled, unpadded 48 KB LDS image; 134-136 VGPRs): finer-grained
  // rounds and a third workgroup to fill the CU while two sit in their epilogues.  Relative cost of a K-iteration 0.8
  // (0.6-0.8 measured the same); against the two-per-CU form of the same kernel: 30.14 -> 29.85 ms per step.
#ifndef SACV_W64_DGRAD
#define SACV_W64_DGRAD 0.8  // tuning builds only (scripts/build_variant.sh): relative K-iteration cost in mode 1
#endif
  NtPlan p64 = plan_nt(tiles * ((N + 63) / 64), KT, nt_slots(64), ws, mode == 1 ? SACV_W64_DGRAD : 0.8);
  if (sacv_share.nt_split) {
#ifndef SACV_SP_WIDE_MIN_N
#define SACV_SP_WIDE_MIN_N 128  // outputs from which the split-operand form takes 128 x 128 items (K 16 slabs)
#endif
    if (N >= SACV_SP_WIDE_MIN_N) {
      const int KT16 = KH * KW * ((Cs + 15) / 16);
#ifndef SACV_SP_KS_WIDE
#define SACV_SP_KS_WIDE 1.0  // tuning builds (scripts/build_variant.sh): planner constants of the 16-deep split kernels
#endif
#ifndef SACV_SP_KS_NARROW
#define SACV_SP_KS_NARROW 1.0
#endif
      NtPlan pw = plan_nt(tiles * ((N + 127) / 128), KT16, nt_slots_split(), ws, 1.0, SACV_SP_KS_WIDE);
      return mode == 0 ? launch_nt<128, 128, 0, 2, 2, true, true, 16>(a, pw, workspace, st)
                       : launch_nt<128, 128, 1, 2, 2, true, true, 16>(a, pw, workspace, st);
    }
    NtPlan ps = plan_nt(tiles * ((N + 63) / 64), KT, nt_slots_split(), ws, 1.0, SACV_SP_KS_NARROW);
    return mode == 0 ? launch_nt<128, 64, 0, 2, 2, true, true>(a, ps, workspace, st)
                     : launch_nt<128, 64, 1, 2, 2, true, true>(a, ps, workspace, st);
  }
  const bool use64 = (N <= 64) || (p64.cost < p128.cost);
  if (use64)
    return mode == 0 ? launch_nt<128, 64, 0, 2, 3, true>(a, p64, workspace, st)
                     : launch_nt<128, 64, 1, 2, 3, true>(a, p64, workspace, st);
  return mode == 0 ? launch_nt<128, 128, 0, SACV_NT_WN>(a, p128, workspace, st)
                   : launch_nt<128, 128, 1, SACV_NT_WN>(a, p128, workspace, st);
}

int sacv_conv2d_gemm(int nprob, const float* const* src, float* const* dst, const float* const* mask,
                     const int* Nb, const int* Hs, const int* Ws, const int* Hd, const int* Wd, const float* w,
                     const float* bias, int Cs, int N, int KH, int KW, int stride, int pad, int mode, int flags,
                     int remap_s, int Hr, int Wr, const int* ext, float* workspace, float* stats, void* stream) {
  return conv2d_gemm_impl(nprob, src, dst, mask, Nb, Hs, Ws, Hd, Wd, w, bias, Cs, N, KH, KW, stride, pad, mode, flags,
                          remap_s, Hr, Wr, ext, workspace, stats, stream, nullptr);
}

// One problem of sacv_conv2d_gemm (usually a data gradient, mode 1) whose output dz [M][N] is the upstream gradient of a
// BatchNorm (+ ReLU) layer with pre-normalisation input bn_y [M][N]: the epilogue also emits that layer's backward
// partial sums -- stats[m tile][0][n] = sum g, [1][n] = sum g * xhat, g = dz * (z > 0), xhat = (y - mean) * invstd --
// i.e. the first stage of sacv_bn_train_bwd, so that sacv_bn_train_bwd_partial can go straight to the second.  The ReLU
// mask comes from bn_z (the layer's stored output) or, with bn_z NULL, is recomputed from y with bn_gamma / bn_beta
// (bn_beta NULL: the layer has no ReLU).  stats: sacv_conv2d_gemm_stats_rows rows.
int sacv_conv2d_gemm_bnbwd(const float* src, float* dst, const float* mask, int Nb, int Hs, int Ws, int Hd, int Wd,
                           const float* w, const float* bias, int Cs, int N, int KH, int KW, int stride, int pad, int mode,
                           int flags, float* workspace, float* stats, const float* bn_y, const float* bn_z,
                           const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                           void* stream) {
  const BnFuse bn = {bn_y, bn_z, bn_mean, bn_invstd, bn_gamma, bn_beta};
  const float* srcs[1] = {src};
  float* dsts[1] = {dst};
  const float* masks[1] = {mask};
  return conv2d_gemm_impl(1, srcs, dsts, mask ? masks : nullptr, &Nb, &Hs, &Ws, &Hd, &Wd, w, bias, Cs, N, KH, KW, stride,
                          pad, mode, flags, 0, 0, 0, nullptr, workspace, stats, stream, &bn);
}

// Rows of the CF_STATS buffer ([rows][2][N] floats): the number of 128-row M tiles over all problems.
int sacv_conv2d_gemm_stats_rows(int nprob, const int* Nb, const int* Hd, const int* Wd) {
  int rows = 0;
  for (int i = 0; i < nprob; ++i) rows += (int)(((long long)Nb[i] * Hd[i] * Wd[i] + 127) / 128);
  return rows;
}

// Bytes of split-K workspace sacv_conv2d_gemm wants (it runs unsplit when given NULL).
long long sacv_conv2d_gemm_workspace_bytes(void) {
  return 2LL * num_cus() * SACV_NT_BLOCKS_PER_CU * 128 * 128 * (long long)sizeof(float);  // 2 slabs per workgroup
}

// Pixels per split of sacv_conv2d_wgrad for a problem list (the `chunk` argument): the split count that fills the
// workgroup slots the weight gradients' queue plans for (sacv_launch_share).  One segment: equal splits, tiles x splits
// within one co-resident wave of workgroups (512 beats 544 and 1024 with 128-wide tiles; the 64 x 64 tiles of the 1x1
// layers run four to a CU: 768).  Several segments (pyramid levels sharing a weight): the splits are unequal (short
// levels), so twice as many, smaller splits balance better (1024 beats 512 by 20 %).
int sacv_conv2d_wgrad_chunk(int nseg, const int* npix, int Cout, int N2, int taps) {
  if (nseg < 1 || !npix || Cout <= 0 || N2 <= 0) return -1;
  int bm, bn;
  wgrad_tile(Cout, N2, taps, &bm, &bn);
  int cus = num_cus() - sacv_share.cu_reserve;
  if (cus < 8) cus = 8;
  const int per_cu = sacv_share.wgrad_wg_per_cu;  // 2 alone
  int target = (bm == 64 && bn == 64 && taps == 1) ? cus * per_cu * 3 / 2 : cus * per_cu;
  if (nseg > 1) target *= 2;
  const long long tiles = (long long)((Cout + bm - 1) / bm) * ((N2 + bn - 1) / bn);
  long long total = 0;
  for (int i = 0; i < nseg; ++i) total += npix[i];
  long long chunks = target / tiles;
  const long long max_chunks = (total + 127) / 128;
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  long long chunk = (total + chunks - 1) / chunks;
  chunk = (chunk + 31) / 32 * 32;
  return (int)(chunk < 32 ? 32 : chunk);
}

// Workspace (bytes) needed by sacv_conv2d_wgrad for the given problem list.
long long sacv_conv2d_wgrad_workspace_bytes(int nseg, const int* Nb, const int* Ho, const int* Wo, int Cin_pad,
                                            int Cout, int KH, int KW, int chunk) {
  long long chunks = 0;
  for (int i = 0; i < nseg; ++i) chunks += ((long long)Nb[i] * Ho[i] * Wo[i] + chunk - 1) / chunk;
  return chunks * Cout * (long long)KH * KW * Cin_pad * (long long)sizeof(float);
}

// Weight gradient.  dy[s]: [Nb*Ho*Wo][Cout]; x[s]: NHWC [Nb][Hx][Wx][Cin_pad]; dw: OIHW
// [Cout][Cin][KH][KW] (Cin <= Cin_pad; the stem uses its im2col matrix as a 1x1 problem).
int sacv_conv2d_wgrad(int nseg, const float* const* dy, const float* const* x, const int* Nb, const int* Hx,
                      const int* Wx, const int* Ho, const int* Wo, float* dw, float* workspace, int Cin, int Cin_pad,
                      int Cout, int KH, int KW, int stride, int pad, int chunk, int accumulate, int dw_taps,
                      int dw_layout, void* stream) {
  SACV_CHECK_ARG(dw_layout == 0 || (dw_layout == 1 && dw_taps == 0), "conv2d_wgrad: dw_layout");
  SACV_CHECK_ARG(nseg >= 1 && nseg <= SACV_MAXP, "conv2d_wgrad: nseg");
  SACV_CHECK_ARG(Cin_pad % 4 == 0 && Cout % 4 == 0, "conv2d_wgrad: channels must be multiples of 4");
  SACV_CHECK_ARG(chunk % 32 == 0 && chunk > 0, "conv2d_wgrad: chunk must be a multiple of 32");
  SACV_CHECK_ARG(workspace && dw, "conv2d_wgrad: null pointer");
  WgradArgs a;
  int chunks = 0;
  for (int i = 0; i < nseg; ++i) {
    WgradSeg& s = a.s[i];
    s.dy = dy[i];
    s.x = x[i];
    s.Nb = Nb[i];
    s.Hx = Hx[i];
    s.Wx = Wx[i];
    s.Ho = Ho[i];
    s.Wo = Wo[i];
    long long np = (long long)Nb[i] * Ho[i] * Wo[i];
    SACV_CHECK_ARG(np * (long long)(Cout > Cin_pad ? Cout : Cin_pad) < (1LL << 31) &&
                       (long long)Nb[i] * Hx[i] * Wx[i] * Cin_pad < (1LL << 32),
                   "conv2d_wgrad: too large");
    s.npix = (int)np;
    s.chunk_begin = chunks;
    chunks += (int)((np + chunk - 1) / chunk);
  }
  a.nseg = nseg;
  a.nchunks = chunks;
  a.chunk = chunk;
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
  int bm, bn;
  wgrad_tile(Cout, a.N2, KH * KW, &bm, &bn);
  if (bm == 64)
    rc = (bn == 64) ? launch_tn<64, 64>(a, chunks, st) : launch_tn<64, 128>(a, chunks, st);
  else
    rc = (a.N2 <= 64) ? launch_tn<128, 64>(a, chunks, st) : launch_tn<128, 128>(a, chunks, st);
  if (rc != SACV_OK) return rc;
  return sacv_internal_wgrad_reduce(workspace, dw, chunks, Cout, Cin, Cin_pad, KH, KW, accumulate, dw_taps, dw_layout,
                                    stream);
}

int sacv_pack_weight(const float* w_oihw, float* out, int Cout, int Cin, int KH, int KW, int kind, int Kpad,
                     void* stream) {
  SACV_CHECK_ARG(w_oihw && out, "pack_weight: null pointer");
  size_t total = (size_t)Cout * Cin * KH * KW;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0)
    hipLaunchKernelGGL(pack_w_fwd_kernel, dim3(blocks), dim3(256), 0, st, w_oihw, out, Cout, Cin, KH * KW);
  else if (kind == 1)
    hipLaunchKernelGGL(pack_w_dgrad_kernel, dim3(blocks), dim3(256), 0, st, w_oihw, out, Cout, Cin, KH * KW, 0);
  else if (kind == 3)  // data-gradient panel from a channels_last (OHWI) weight
    hipLaunchKernelGGL(pack_w_dgrad_kernel, dim3(blocks), dim3(256), 0, st, w_oihw, out, Cout, Cin, KH * KW, 1);
  else if (kind == 2) {
    SACV_CHECK_ARG(Kpad >= Cin * KH * KW, "pack_weight: Kpad too small");
    hipLaunchKernelGGL(pack_w_stem_kernel, dim3(sacv_cdiv((long long)Cout * Kpad, 256)), dim3(256), 0, st, w_oihw,
                       out, Cout, Cin, KH * KW, Kpad);
  } else {
    sacv_set_error("pack_weight: unknown kind %d", kind);
    return SACV_ERR_ARG;
  }
  SACV_CHECK_LAUNCH("pack_weight");
  return SACV_OK;
}

// descs: device array of {const float* src; float* dst; int Cout, Cin, T, src_ohwi} (32 bytes each);
// chunk_map: device int[nchunks][2] = (descriptor index, 32x32 tile index = (tap*ceil(Cin/32) + ci_blk)*ceil(Cout/32) + co_blk)
int sacv_pack_dgrad_batched(const void* descs, const int* chunk_map, int nchunks, void* stream) {
  SACV_CHECK_ARG(descs && chunk_map && nchunks > 0, "pack_dgrad_batched: bad args");
  static_assert(sizeof(PackDesc) == 32, "PackDesc layout is part of the ABI");
  hipLaunchKernelGGL(pack_dgrad_batched_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream,
                     (const PackDesc*)descs, chunk_map);
  SACV_CHECK_LAUNCH("pack_dgrad_batched");
  return SACV_OK;
}

int sacv_im2col_stem(const float* x, float* col, int Nb, int Cin, int H, int W, long long sN, long long sC,
                     long long sH, long long sW, int Ho, int Wo, int KH, int KW, int stride, int pad, int Kpad,
                     void* stream) {
  SACV_CHECK_ARG(x && col, "im2col_stem: null pointer");
  SACV_CHECK_ARG(Kpad % 32 == 0 && Kpad >= Cin * KH * KW, "im2col_stem: Kpad must be a multiple of 32 >= K");
  SACV_CHECK_ARG(Kpad / 4 <= 256, "im2col_stem: Kpad %d too large", Kpad);
  const size_t npix = (size_t)Nb * Ho * Wo;
  const int rpb = 256 / (Kpad / 4);
  const size_t nblk = (npix + rpb - 1) / rpb;
  const int blocks = (int)(nblk > 16384 ? 16384 : nblk);
  hipLaunchKernelGGL(im2col_stem_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, col, Nb, Cin, H, W, sN,
                     sC, sH, sW, Ho, Wo, KH, KW, stride, pad, Kpad);
  SACV_CHECK_LAUNCH("im2col_stem");
  return SACV_OK;
}

SACV_API_END

