// Winograd F(2x2, 3x3) path for the stride-1 3x3 convolutions of the detection heads (fp32, NHWC, gfx950).
//
// Y = A^T [ (G g G^T) .* (B^T d B) ] A per 2x2 output tile: 16 multiplies per output pair of pixels instead of 36,
// i.e. 2.25x fewer matrix-core flops than the implicit GEMM of conv_gemm.hip.  The reference computes these layers
// with nn.Conv2d (simpleAICV/detection/models/head.py:20-36, 55-70 RetinaClsHead/RetinaRegHead, :103-141 FCOS
// towers); the transforms only use the coefficients 0, +-1, +-1/2, so the result differs from the direct fp32
// sum by ordinary fp32 rounding (tests/test_gpu_winograd.py bounds it against the direct kernel and fp64).
//
//   wino_weight_kernel   U[g][n][k] = (G w[n] G^T)[g]        from a GEMM panel [N][9][K] (forward panel, or the
//                                                             data-gradient panel with the taps flipped)
//   wino_input_kernel    V[g][t][c] = (B^T d_t B)[g]          4x4 input patch of tile t (all levels of a head in one
//                                                             launch; t runs over level, image, tile row, tile col)
//   wino_gemm_kernel     M_g = V[g] U[g]^T for g = 0..15 back to back in ONE workgroup per (64 tiles x 128 channels)
//                        item, each M_g folded into the four output accumulators Y[a][b] += A^T[a][i] A^T[b][j] M_g
//                        (VALU fmas once per g), so M never goes to memory and the inner loop sees one long K
//                        (16 x Cin) with no epilogue in between; the epilogue writes the 2x2 pixels of every tile with
//                        the fused bias / ReLU / sigmoid / accumulate / ReLU-mask / column-sum options of the direct
//                        kernel (conv_gemm.hip vec_epilogue).
//
// Deterministic (no atomics, fixed summation order).
//
// SPLIT operands (round 4): the fp32 matrix pipe of gfx950 (v_mfma_f32_32x32x2_f32, 157 TFLOP/s) is the bound of the
// whole step, the bf16 pipe is 16x faster.  The transform kernels can therefore write U and V as THREE bf16 planes,
// x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): the three round-to-nearest terms
// carry >= 24 significant bits, so the sum reproduces the fp32 value exactly (the weight-gradient kernel rebuilds V
// that way, bit for bit).  The GEMM then forms a.b from the six products whose weight is >= 2^-16 (hi.hi, hi.mid,
// mid.hi, mid.mid, hi.lo, lo.hi; the dropped mid.lo / lo.mid / lo.lo are <= 2^-25 |a.b|, below the rounding of ONE
// fp32 addition) with v_mfma_f32_32x32x16_bf16 into the same fp32 accumulators: 6 x 32 cycles per 16 K instead of
// 8 x 64.  Measured against fp64 the error of a K = 256 product sum is that of the fp32 pipe (its accumulation
// rounding, ~2.5e-7 of the mean magnitude; the dropped terms add 6e-9) -- tests/test_gpu_winograd.py.
#include "sacv_common.h"
#include <mutex>

#define WCF_BIAS 1
#define WCF_RELU 2
#define WCF_SIGMOID 4
#define WCF_ACCUM 8
#define WCF_MASK 16
#define WCF_STATS 32

#ifndef WINO_SGB
#define WINO_SGB 0  // tuning builds: sched_group_barrier interleave of the MFMA block (1: LDS reads, 2: + VALU)
#endif
#ifndef WINO_FRAG2
#define WINO_FRAG2 1  // 1: fragments of the next quarter are read while the current one multiplies (two register sets)
#endif
#ifndef WINO_PIN
#define WINO_PIN 0  // 1: sched_barrier(0) after every quarter of the MFMA block (measured: 2-3 % slower)
#endif
#ifndef WINO_PRIO
#define WINO_PRIO 1  // 0: no s_setprio, 1: MFMA block at priority 1, 2: the glue (DMA issue, waits, fold) above the MFMAs
#endif
#ifndef WINO_ISSUE_AT
#define WINO_ISSUE_AT 0  // -1: DMA issue before the MFMA block, 0..3: after that quarter of the MFMA block
#endif
#ifndef WINO_SPLIT_SGB
#define WINO_SPLIT_SGB 1  // split-operand MFMA block: fragment reads interleaved with the products (sched_group_barrier)
#endif
#define WINO_MAXL 5  // pyramid levels per problem
#define WINO_MAXP 2  // problems (e.g. the two towers of a head) per GEMM launch

static __device__ __attribute__((aligned(16))) float wino_zero16[4] = {0.f, 0.f, 0.f, 0.f};

typedef __bf16 wbf16;
typedef __bf16 wbf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));

// Element (row, col) inside one bf16 plane of the split form.  The GEMM fetches a stage of 32 K per row = 64 bytes per
// plane, HALF a 128-byte line: with plain row-major planes every line went L2 -> CU twice (two consecutive stages),
// and the L2 line rate, not the matrix pipe, bounded the kernel (10 TB/s of useful bytes).  So the planes are stored
// with the 32-column chunks of a ROW PAIR side by side: [rows / 2][cols / 32][2][32] -- one LDS-DMA instruction (16 rows
// x 64 bytes) then reads 8 whole lines.  rows is even (padded to 64 / 128), cols a multiple of 32.
__device__ __forceinline__ size_t wino_split_off(int row, int col, int cols) {
  return ((size_t)(row >> 1) * (cols >> 5) + (col >> 5)) * 64 + ((row & 1) << 5) + (col & 31);
}

// One transformed value group (4 consecutive channels, col % 4 == 0) of plane g: fp32 [16][rows][cols] row-major, or
// the three bf16 planes [16][3][rows][cols] of the split form (layout: wino_split_off).  gs = rows * cols.
template <bool SPLIT>
__device__ __forceinline__ void wino_put4(void* base, size_t gs, int g, int row, int col, int cols, const f32x4 v) {
  if constexpr (!SPLIT) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + (size_t)g * gs + (size_t)row * cols + col) = v;
  } else {
    const size_t off = wino_split_off(row, col, cols);
    wbf16x4 h, m, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = (wbf16)v[e];
      const float r1 = v[e] - (float)h[e];  // exact
      m[e] = (wbf16)r1;
      l[e] = (wbf16)(r1 - (float)m[e]);  // exact difference, rounded once
    }
    wbf16* b = reinterpret_cast<wbf16*>(base) + (size_t)g * 3 * gs + off;
    *reinterpret_cast<wbf16x4*>(b) = h;
    *reinterpret_cast<wbf16x4*>(b + gs) = m;
    *reinterpret_cast<wbf16x4*>(b + 2 * gs) = l;
  }
}

// -------------------------------------------------------------------------------------------- weights
template <bool SPLIT>
__device__ __forceinline__ void wino_weight_body(const float* __restrict__ panel, void* __restrict__ U, int N, int K,
                                                  int Np, int Kp, int flip) {
  const int k4n = Kp >> 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Np * k4n) return;
  const int n = idx / k4n, k4 = idx - n * k4n;
  const size_t gs = (size_t)Np * Kp;
  if (n >= N || k4 * 4 >= K) {  // zero padding: the GEMM reads whole 128 x 32 operand tiles without masks
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 16; ++g) wino_put4<SPLIT>(U, gs, g, n, k4 * 4, Kp, z);
    return;
  }
  const float* p = panel + (size_t)n * 9 * K + k4 * 4;
  f32x4 g[3][3];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int tap = flip ? (2 - kh) * 3 + (2 - kw) : kh * 3 + kw;
      g[kh][kw] = *reinterpret_cast<const f32x4*>(p + (size_t)tap * K);
    }
  f32x4 t[4][3];  // G g
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    t[0][kw] = g[0][kw];
    t[1][kw] = 0.5f * (g[0][kw] + g[1][kw] + g[2][kw]);
    t[2][kw] = 0.5f * (g[0][kw] - g[1][kw] + g[2][kw]);
    t[3][kw] = g[2][kw];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    wino_put4<SPLIT>(U, gs, i * 4 + 0, n, k4 * 4, Kp, t[i][0]);
    wino_put4<SPLIT>(U, gs, i * 4 + 1, n, k4 * 4, Kp, 0.5f * (t[i][0] + t[i][1] + t[i][2]));
    wino_put4<SPLIT>(U, gs, i * 4 + 2, n, k4 * 4, Kp, 0.5f * (t[i][0] - t[i][1] + t[i][2]));
    wino_put4<SPLIT>(U, gs, i * 4 + 3, n, k4 * 4, Kp, t[i][2]);
  }
}

template <bool SPLIT>
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ panel, void* __restrict__ U, int N,
                                                          int K, int Np, int Kp, int flip) {
  wino_weight_body<SPLIT>(panel, U, N, K, Np, Kp, flip);
}

// up to 32 weight transforms in one launch (all Winograd layers of a model, forward and data-gradient panels):
// blockIdx.y = job
#define WINO_WBATCH 32
struct WinoWeightJobs {
  const float* panel[WINO_WBATCH];
  void* U[WINO_WBATCH];
  int N[WINO_WBATCH], K[WINO_WBATCH], flip[WINO_WBATCH];
};
template <bool SPLIT>
__global__ __launch_bounds__(256) void wino_weight_batched_kernel(const WinoWeightJobs j) {
  const int job = blockIdx.y;
  const int N = j.N[job], K = j.K[job];
  wino_weight_body<SPLIT>(j.panel[job], j.U[job], N, K, (N + 127) / 128 * 128, (K + 31) / 32 * 32, j.flip[job]);
}

// -------------------------------------------------------------------------------------------- geometry
struct WinoGeom {
  int nlev;
  int tile_begin[WINO_MAXL + 1];  // first tile row of each level; [nlev] = T
  int th[WINO_MAXL], tw[WINO_MAXL], H[WINO_MAXL], W[WINO_MAXL];
  int Nb;
};

struct WinoTile {
  int lev, n, ty, tx;
};
__device__ __forceinline__ WinoTile wino_tile(const WinoGeom& g, int t) {
  WinoTile r;
  r.lev = 0;
#pragma unroll
  for (int l = 1; l < WINO_MAXL; ++l)
    if (l < g.nlev && t >= g.tile_begin[l]) r.lev = l;
  int tb = g.tile_begin[0], th = g.th[0], tw = g.tw[0];
#pragma unroll
  for (int l = 1; l < WINO_MAXL; ++l)
    if (r.lev == l) {
      tb = g.tile_begin[l];
      th = g.th[l];
      tw = g.tw[l];
    }
  const int local = t - tb;
  const int per = th * tw;
  r.n = local / per;
  const int rem = local - r.n * per;
  r.ty = rem / tw;
  r.tx = rem - r.ty * tw;
  return r;
}

// -------------------------------------------------------------------------------------------- input transform
struct WinoInArgs {
  WinoGeom g;
  const float* src[WINO_MAXL];
  void* V;
  int C, Cp, T, Tpad;
};

template <bool SPLIT>
__global__ __launch_bounds__(256) void wino_input_kernel(const WinoInArgs a) {
  const int c4n = a.Cp >> 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)a.Tpad * c4n) return;
  int t = (int)(idx / c4n), c4 = (int)(idx - (long long)t * c4n);
  if constexpr (SPLIT) {
    // 16 consecutive threads = the 32-channel chunk of a ROW PAIR = one 128-byte line of every plane (wino_split_off):
    // with a thread row per tile each store instruction wrote 64-byte half lines (+25 % on the transform's time)
    const int kcn = a.Cp >> 5;
    const long long unit = idx >> 4;
    const int within = (int)(idx & 15);
    t = 2 * (int)(unit / kcn) + (within >> 3);
    c4 = (int)(unit % kcn) * 8 + (within & 7);
  }
  const size_t gs = (size_t)a.Tpad * a.Cp;
  if (t >= a.T || c4 * 4 >= a.C) {  // padding rows / channels of the GEMM's 64 x 32 operand tiles: zeros
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 16; ++g) wino_put4<SPLIT>(a.V, gs, g, t, c4 * 4, a.Cp, z);
    return;
  }
  const WinoTile tl = wino_tile(a.g, t);
  const float* src = a.src[0];
  int H = a.g.H[0], W = a.g.W[0];
#pragma unroll
  for (int l = 1; l < WINO_MAXL; ++l)
    if (tl.lev == l) {
      src = a.src[l];
      H = a.g.H[l];
      W = a.g.W[l];
    }
  const int h0 = 2 * tl.ty - 1, w0 = 2 * tl.tx - 1;
  const float* base = src + (((long long)tl.n * H + h0) * W + w0) * a.C + c4 * 4;
  f32x4 d[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bool ok = ((unsigned)(h0 + r) < (unsigned)H) & ((unsigned)(w0 + s) < (unsigned)W);
      const float* p = ok ? base + ((long long)r * W + s) * a.C : wino_zero16;
      d[r][s] = *reinterpret_cast<const f32x4*>(p);
    }
  f32x4 u[4][4];  // B^T d
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    u[0][s] = d[0][s] - d[2][s];
    u[1][s] = d[1][s] + d[2][s];
    u[2][s] = d[2][s] - d[1][s];
    u[3][s] = d[1][s] - d[3][s];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    wino_put4<SPLIT>(a.V, gs, i * 4 + 0, t, c4 * 4, a.Cp, u[i][0] - u[i][2]);
    wino_put4<SPLIT>(a.V, gs, i * 4 + 1, t, c4 * 4, a.Cp, u[i][1] + u[i][2]);
    wino_put4<SPLIT>(a.V, gs, i * 4 + 2, t, c4 * 4, a.Cp, u[i][2] - u[i][1]);
    wino_put4<SPLIT>(a.V, gs, i * 4 + 3, t, c4 * 4, a.Cp, u[i][1] - u[i][3]);
  }
}

// one LDS-DMA wave-instruction: 64 lanes x 16 bytes, per-lane global source, lane-linear LDS destination.  The builtin
// only exists for the gfx950 device pass (the host pass of the same translation unit must not see it).
typedef __attribute__((address_space(3))) void* lds_t;
__device__ __forceinline__ void glds16(const void* src, lds_t dst) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
#else
  (void)src;
  (void)dst;
#endif
}

// -------------------------------------------------------------------------------------------- GEMM + output transform
struct WinoProb {
  WinoGeom g;
  const void* V;      // fp32 [16][Tpad][K], or the split form: bf16 [16][3][Tpad][K]
  const void* U;      // fp32 [16][Np][K],   or bf16 [16][3][Np][K]
  const float* bias;  // [N] or null
  float* stats;       // [mtiles][2][N] (WCF_STATS)
  float* dst[WINO_MAXL];         // per level [Nb][H][W][N]
  const float* mask[WINO_MAXL];  // per level, same shape as dst (WCF_MASK)
  int T, Tpad, K, N, Np, flags;  // K: padded to 32; Np: rows of U (N padded to 128)
  int item_begin, ntn;
  int mt0;  // first M tile (of BM rows) this launch computes
};

struct WinoArgs {
  WinoProb p[WINO_MAXP];
  int nprob, total_items;
};

struct WinoRow {  // per tile row of the current item (LDS)
  float* dst;
  const float* mask;
  int off;    // element offset of output pixel (2ty, 2tx), channel 0
  int wn;     // W * N: element distance of one output row
  int valid;  // bit 0: tile row < T, bit 1: pixel row 2ty+1 < H, bit 2: pixel column 2tx+1 < W
  int pad;
};

// Operand tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4) into a ring of three unpadded, XOR-swizzled
// stages with TWO tiles in flight (counted vmcnt + raw s_barrier, cdna_hip_programming.md 5 "Pipelining across
// barriers").  WN = waves along N: 2 -> 4 waves of 32 x 64 (<= 256 VGPRs, 2 waves per SIMD with two workgroups per
// CU), 4 -> 8 waves of 32 x 32 (<= 128 VGPRs, 4 waves per SIMD).
//
// SPLIT: the operands are three bf16 planes each (file header).  A stage still holds 32 K of BM + BN rows, now as
// 3 planes x 64 B per row (1.5x the bytes); one LDS-DMA instruction fills 16 rows x 64 B of one plane, lane l = (row
// l >> 2, 16-byte slot l & 3) fetching logical chunk slot ^ key(row) with key(r) = (r >> 2) & 3: a ds_read_b128
// fragment read (lane = row, one logical chunk per half-wave) then meets, in each of its 16-lane groups, four rows
// of every residue mod 4 (= the four 64-byte quarters of the 256-byte bank line) with four different keys.  Two
// v_mfma_f32_32x32x16_bf16 K steps per stage, six products each.
template <int BM, int BN, int WN, int OCC, bool SPLIT>
__global__ __launch_bounds__(2 * BM * WN, OCC) void wino_gemm_kernel(const WinoArgs a) {
  constexpr int NT = 2 * BM * WN;  // BM / 32 waves along M x WN along N
  constexpr int NW = NT / 64;
  constexpr int STAGE = SPLIT ? (BM + BN) * 48 : (BM + BN) * 32;  // floats per stage (fp32: rows of 32 floats; split: 3 planes of 64-byte rows)
  constexpr int WTM = 32, WTN = BN / WN;  // one 32-row MFMA block per wave along M
  constexpr int TN = WTN / 32;
  constexpr int CS = BN + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  WinoRow* rows = reinterpret_cast<WinoRow*>(smem + 3 * STAGE);

  // XCD-contiguous item order: workgroup b runs on XCD b % 8; every XCD takes a contiguous range of items so that the
  // N tiles of one M tile (same V rows) meet in one L2
  const int per = (a.total_items + 7) >> 3;
  for (int it = blockIdx.x; it < per * 8; it += gridDim.x) {
    const int item = (it & 7) * per + (it >> 3);
    if (item >= a.total_items) continue;
    // thread-index arithmetic is re-derived per item (and again before the epilogue) behind an opaque copy of the
    // thread id: hoisted out of the persistent loop it occupied ~40 registers for the whole kernel, spilled, and the
    // reloads (scratch loads count on vmcnt) drained the LDS-DMA pipeline in every K iteration
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid / WN) * WTM, wn0 = (wid % WN) * WTN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int pi = (a.nprob > 1 && item >= a.p[1].item_begin) ? 1 : 0;
    const WinoProb& P = a.p[pi];
    const int local = item - P.item_begin;
    const int mt = P.mt0 + local / P.ntn, nt = local % P.ntn;  // this launch covers the M tiles from mt0 on
    const int m0 = mt * BM, n0 = nt * BN;
    const int K = P.K, N = P.N;
    const int KT = K >> 5;  // K is padded to whole tiles
    const size_t vgs = (size_t)P.Tpad * K, ugs = (size_t)P.Np * K;

    __syncthreads();  // the previous item's epilogue has left the LDS
    if (tid < BM) {
      WinoRow r;
      const int t = m0 + tid;
      r.valid = 0;
      r.dst = nullptr;
      r.mask = nullptr;
      r.off = 0;
      r.wn = 0;
      r.pad = 0;
      if (t < P.T) {
        const WinoTile tl = wino_tile(P.g, t);
        int H = P.g.H[0], W = P.g.W[0];
        r.dst = P.dst[0];
        r.mask = P.mask[0];
#pragma unroll
        for (int l = 1; l < WINO_MAXL; ++l)
          if (tl.lev == l) {
            H = P.g.H[l];
            W = P.g.W[l];
            r.dst = P.dst[l];
            r.mask = P.mask[l];
          }
        r.off = ((tl.n * H + 2 * tl.ty) * W + 2 * tl.tx) * N;
        r.wn = W * N;
        r.valid = 1 | ((2 * tl.ty + 1 < H) ? 2 : 0) | ((2 * tl.tx + 1 < W) ? 4 : 0);
      }
      rows[tid] = r;
    }

    f32x16 Y[4][TN], acc[TN];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) Y[q][j][r] = 0.f;
    // fold M_g into the output-pixel accumulators: Y[a][b] += AT[a][i] AT[b][j] M_g, AT = [[1,1,1,0],[0,1,-1,-1]].
    // Branch-free (fma with a wave-uniform coefficient in {-1, 0, 1}: exact): the version with a branch per plane made
    // the register allocator keep one 16-register accumulator in scratch, and its reloads (vmcnt) drained the DMA ring.
    auto fold = [&](int g) {
      const int gi = g >> 2, gj = g & 3;
      const float ca[2] = {gi < 3 ? 1.f : 0.f, gi == 0 ? 0.f : (gi == 1 ? 1.f : -1.f)};
      const float cb2[2] = {gj < 3 ? 1.f : 0.f, gj == 0 ? 0.f : (gj == 1 ? 1.f : -1.f)};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float c = ca[q >> 1] * cb2[q & 1];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) Y[q][j][r] = __builtin_fmaf(acc[j][r], c, Y[q][j][r]);
      }
    };

    // One LDS-DMA wave-instruction writes 64 lanes x 16 B = 1 KiB = 8 rows of 128 B, lane-linear; lane l fills
    // (row l >> 3, 16-byte slot l & 7) and FETCHES logical chunk slot ^ key(row), key(r) = (r >> 1) & 7, so slot p of
    // row r holds chunk p ^ key(r).  A ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15, 20-27},
    // {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- and a fragment read has lane = row, one
    // logical chunk per half-wave: rows of one parity share a 128-byte bank half, and within each group the eight rows
    // of a parity have eight different keys, so the reads are conflict-free without padding.  (key(r) = r & 7 looks
    // right for 8 consecutive lanes but is 2-way conflicting on the real groups: rows 12-15 meet rows 20-23.)
    constexpr int RPI = SPLIT ? 16 : 8;                    // rows per LDS-DMA instruction
    constexpr int PA = BM / RPI / NW, PB = BN / RPI / NW;  // row pieces per wave and tile (and plane)
    constexpr int NG = (SPLIT ? 3 : 1) * (PA + PB);        // LDS-DMA instructions per wave and tile
    static_assert(PA >= 1 && PB >= 1, "every wave fetches a piece of both operands");
    static_assert(NG == 6 || NG == 4 || NG == 3, "the counted waits below are written for 6, 4 or 3 DMA instructions");
    // element offsets of this lane's 16 bytes inside a (g, kc) operand tile; the operands are zero-padded to whole
    // tiles (sacv_wino_pad_k / _pad_n), so every load is unconditional: scalar base + 32-bit lane offset
    unsigned aoff[PA], wrow[PB];
    if constexpr (SPLIT) {
      const int prow = lane >> 2;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int row = (wid * PA + i) * 16 + prow;
        aoff[i] = (unsigned)wino_split_off(m0 + row, ((lane & 3) ^ ((row >> 2) & 3)) << 3, K);
      }
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int row = (wid * PB + i) * 16 + prow;
        wrow[i] = (unsigned)wino_split_off(n0 + row, ((lane & 3) ^ ((row >> 2) & 3)) << 3, K);
      }
    } else {
      const int prow = lane >> 3;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int row = (wid * PA + i) * 8 + prow;
        aoff[i] = (unsigned)((m0 + row) * K + (((lane & 7) ^ ((row >> 1) & 7)) << 2));
      }
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int row = (wid * PB + i) * 8 + prow;
        wrow[i] = (unsigned)((n0 + row) * K + (((lane & 7) ^ ((row >> 1) & 7)) << 2));
      }
    }
    const wbf16* Vb = reinterpret_cast<const wbf16*>(P.V);
    const wbf16* Ub = reinterpret_cast<const wbf16*>(P.U);
    unsigned vgs32 = (unsigned)vgs, ugs32 = (unsigned)ugs;
    if constexpr (SPLIT) asm volatile("" : "+s"(Vb), "+s"(Ub), "+s"(vgs32), "+s"(ugs32));
    auto issue = [&](int g, int kc, int st) {
      const int c0 = SPLIT ? kc << 6 : kc << 5;  // split: the chunks of a row pair lie side by side (wino_split_off)
      float* base = smem + st * STAGE;
      if constexpr (SPLIT) {
        // scalar part in 32-bit element offsets (48 Tpad K < 2^32, checked by the host) on bases that live in SGPRs
        // for the whole item: a kernel-argument reload here is a scalar-memory wait in the middle of the MFMA stream
        const wbf16* v = Vb + ((unsigned)g * 3u * vgs32 + (unsigned)c0);
        const wbf16* u = Ub + ((unsigned)g * 3u * ugs32 + (unsigned)c0);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
          for (int i = 0; i < PA; ++i)
            glds16(v + (pl * vgs32 + aoff[i]), (lds_t)(base + pl * BM * 16 + (wid * PA + i) * 256));
#pragma unroll
          for (int i = 0; i < PB; ++i)
            glds16(u + (pl * ugs32 + wrow[i]), (lds_t)(base + 3 * BM * 16 + pl * BN * 16 + (wid * PB + i) * 256));
        }
      } else {
        const float* v = reinterpret_cast<const float*>(P.V) + (size_t)g * vgs + c0;
        const float* u = reinterpret_cast<const float*>(P.U) + (size_t)g * ugs + c0;
#pragma unroll
        for (int i = 0; i < PA; ++i) glds16(v + aoff[i], (lds_t)(base + (wid * PA + i) * 256));
#pragma unroll
        for (int i = 0; i < PB; ++i) glds16(u + wrow[i], (lds_t)(base + BM * 32 + (wid * PB + i) * 256));
      }
    };
    auto wait_one_in_flight = [&]() {  // all but the youngest tile's DMA instructions of this wave have landed
      if (NG == 6)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (NG == 4)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    };
    const int total = 16 * KT;
    const int x7 = (l31 >> 1) & 7;  // key of this lane's fragment rows (wm0, wn0 and j * 32 are multiples of 16)
    issue(0, 0, 0);
    {
      int g1 = 0, k1 = 1;
      if (k1 >= KT) {
        k1 = 0;
        g1 = 1;
      }
      issue(g1, k1, 1);  // total >= 16
    }
    wait_one_in_flight();
    __builtin_amdgcn_s_barrier();
#ifdef WINO_STAMP
    long long st_acc[6] = {0, 0, 0, 0, 0, 0};
    long long st_t = (long long)__builtin_amdgcn_s_memtime();
#define WINO_STAMP_AT(k)                                          \
  do {                                                            \
    const long long now__ = (long long)__builtin_amdgcn_s_memtime(); \
    st_acc[k] += now__ - st_t;                                    \
    st_t = now__;                                                 \
  } while (0)
#else
#define WINO_STAMP_AT(k) do {} while (0)
#endif
    int cur = 0, q = 0;
    int ng = 0, nk = 2;  // coordinates of tile q + 2
    while (nk >= KT) {
      nk -= KT;
      ++ng;
    }
    for (int g = 0; g < 16; ++g) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      for (int kc = 0; kc < KT; ++kc, ++q) {
        const bool more = q + 2 < total;
        int st2 = cur + 2;
        if (st2 >= 3) st2 -= 3;
#if WINO_PRIO == 2
        __builtin_amdgcn_s_setprio(3);
#endif
#if WINO_ISSUE_AT < 0
        if (more) issue(ng, nk, st2);
#endif
        WINO_STAMP_AT(0);
        if constexpr (SPLIT) {
          // rows of 64 B (16 floats) per plane; the lane's row is wm0 / wn0 + l31, its key (l31 >> 2) & 3
          const float* cA = smem + cur * STAGE + (wm0 + l31) * 16;
          const float* cB = smem + cur * STAGE + 3 * BM * 16 + (wn0 + l31) * 16;
          const int k3 = (l31 >> 2) & 3;
          wbf16x8 fa[2][3], fb[2][TN][3];
#ifndef WINO_DIAG_NODMA  // (tuning builds: the matrix work on stale LDS contents, no operand traffic)
          if (more) issue(ng, nk, st2);  // tile q + 2 into the stage that iteration q - 1 read
#endif
          // Fragment reads and products in ONE order, plane by plane (hi, mid, lo), so that the reads can trickle in
          // between the MFMAs instead of all 9 (TN + 1) x 3 x 2 landing before the first one: right after the barrier
          // every wave of the workgroup wants its fragments at once, and with all reads up front the matrix pipe idled
          // for the whole burst (in-kernel stamps: 40 % of a wave's time issuing reads, 27 % in the MFMA block).
          // Products per K step, largest first: hi.hi, mid.hi, hi.mid, mid.mid, lo.hi, hi.lo (x TN).
#if WINO_PRIO == 1
          __builtin_amdgcn_s_setprio(1);  // (before the reads: s_setprio is a scheduling boundary)
#endif
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int o = ((ks * 2 + lh) ^ k3) * 4;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
              fa[ks][pl] = *reinterpret_cast<const wbf16x8*>(cA + pl * BM * 16 + o);
#pragma unroll
              for (int j = 0; j < TN; ++j)
                fb[ks][j][pl] = *reinterpret_cast<const wbf16x8*>(cB + pl * BN * 16 + j * 32 * 16 + o);
            }
          }
#ifdef WINO_STAMP
          WINO_STAMP_AT(0);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          WINO_STAMP_AT(5);
#endif
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            constexpr int pa[6] = {0, 1, 0, 1, 2, 0};
            constexpr int pb[6] = {0, 0, 1, 1, 0, 2};
#pragma unroll
#ifdef WINO_DIAG_NOMFMA  // (tuning builds: the data path without the matrix work; one product keeps the reads alive)
            for (int t6 = 0; t6 < 1; ++t6)
#else
            for (int t6 = 0; t6 < 6; ++t6)
#endif
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][pa[t6]], fb[ks][j][pb[t6]], acc[j], 0, 0, 0);
          }
#if !defined(WINO_STAMP) && !defined(WINO_DIAG_NOMFMA) && WINO_SPLIT_SGB
          // reads in source order: per K step (A, B_0 .. B_TN-1) of hi, then of mid, then of lo.  The first product
          // needs TN + 1 of them; then one read per MFMA keeps ahead of the products' needs.
          __builtin_amdgcn_sched_group_barrier(0x100, TN + 2, 0);
#pragma unroll
          for (int m = 0; m < 12 * TN; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (m < 6 * (TN + 1) - (TN + 2)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
#endif
        } else {
        const float* cA = smem + cur * STAGE + (wm0 + l31) * 32;
          const float* cB = smem + cur * STAGE + BM * 32 + (wn0 + l31) * 32;
  #if WINO_FRAG2
          f32x4 fa[2], fb[2][TN];
          {
            const int o = (lh ^ x7) * 4;
            fa[0] = *reinterpret_cast<const f32x4*>(cA + o);
  #pragma unroll
            for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const f32x4*>(cB + j * 32 * 32 + o);
          }
  #else
          f32x4 fa[1], fb[1][TN];
  #endif
  #pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
  #if WINO_FRAG2
            const int cb = jj & 1, nbuf = cb ^ 1;
            if (jj < 3) {
              const int o = (((jj + 1) * 2 + lh) ^ x7) * 4;
              fa[nbuf] = *reinterpret_cast<const f32x4*>(cA + o);
  #pragma unroll
              for (int j = 0; j < TN; ++j) fb[nbuf][j] = *reinterpret_cast<const f32x4*>(cB + j * 32 * 32 + o);
            }
  #else
            constexpr int cb = 0;
            {
              const int o = ((jj * 2 + lh) ^ x7) * 4;
              fa[0] = *reinterpret_cast<const f32x4*>(cA + o);
  #pragma unroll
              for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const f32x4*>(cB + j * 32 * 32 + o);
            }
  #endif
  #if WINO_PRIO == 1
            __builtin_amdgcn_s_setprio(1);
  #elif WINO_PRIO == 2
            __builtin_amdgcn_s_setprio(0);
  #endif
  #pragma unroll
            for (int e = 0; e < 4; ++e)
  #pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cb][e], fb[cb][j][e], acc[j], 0, 0, 0);
  #if WINO_ISSUE_AT >= 0
            if (jj == WINO_ISSUE_AT && more) issue(ng, nk, st2);
  #endif
  #if WINO_SGB
  #pragma unroll
            for (int m = 0; m < 4 * TN; ++m) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  // 1 MFMA
              if (m < 1 + TN) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 LDS read (next fragments)
              if (WINO_SGB > 1) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // address VALU
            }
  #endif
  #if WINO_PIN
            // keep the scheduler from hoisting all four quarters' fragment reads to the loop top: with the 128 output
            // accumulators live that spills one of them, and the spill reloads (vmcnt) drain the DMA ring at every fold
            __builtin_amdgcn_sched_barrier(0);
  #endif
          }
        }
#if WINO_PRIO == 1
        __builtin_amdgcn_s_setprio(0);
#elif WINO_PRIO == 2
        __builtin_amdgcn_s_setprio(3);
#endif
        if (++nk >= KT) {
          nk = 0;
          ++ng;
        }
        WINO_STAMP_AT(1);
        // tile q + 1 (issued one iteration ago) has landed for this wave; tile q + 2 stays in flight
        if (more)
          wait_one_in_flight();
        else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WINO_STAMP_AT(2);
        __builtin_amdgcn_s_barrier();
        WINO_STAMP_AT(3);
        cur = cur == 2 ? 0 : cur + 1;
      }
      fold(g);
      WINO_STAMP_AT(4);
    }
#ifdef WINO_STAMP
    if (lane == 0 && P.stats) {
      long long* o = reinterpret_cast<long long*>(P.stats) + ((size_t)blockIdx.x * NW + wid) * 8;
      for (int k = 0; k < 6; ++k) o[k] = st_acc[k];
    }
#endif

    // ---- epilogue: the four output pixels of every tile, one staged [BM][BN] plane at a time
    float* stage = smem;
    constexpr int CG = BN / 4, RS = NT / CG;
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int c4 = tid_e % CG, rr = tid_e / CG;
    const int n = n0 + c4 * 4;
    const bool nok = n < N;  // N % 4 == 0
    const int flags = P.flags;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if ((flags & WCF_BIAS) && nok) bv = *reinterpret_cast<const f32x4*>(P.bias + n);
    f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
    const int e_lane = tid_e & 63, e_wid = tid_e >> 6;
    const int e_wm0 = (e_wid / WN) * WTM, e_wn0 = (e_wid % WN) * WTN, e_l31 = e_lane & 31, e_lh = e_lane >> 5;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int pa = q4 >> 1, pb = q4 & 1;
      if (q4) __syncthreads();  // the previous plane has been consumed
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          stage[(e_wm0 + (r & 3) + 8 * (r >> 2) + 4 * e_lh) * CS + e_wn0 + j * 32 + e_l31] = Y[q4][j][r];
      __syncthreads();
      const int need = 1 | (pa ? 2 : 0) | (pb ? 4 : 0);
      // The rows of a thread are handled as ONE batch: every operand the epilogue reads from memory (the ReLU mask of a
      // data gradient, or the tensor it accumulates into) is requested for all of them before the first is used, and
      // the stores follow the last use.  (Until round 4 this was a loop over the rows with the load, its wait and the
      // store in every iteration: loads and stores share vmcnt, so each row waited for the previous row's store to be
      // acknowledged and then for its own load -- 16 round trips per work item, +100 us on a two-tower data-gradient
      // launch against the forward launch of the same GEMMs.)
      constexpr int RPT = BM / RS;
      float* dptr[RPT];
      bool ok[RPT];
      f32x4 pre[RPT];
      const bool pre_mask = (flags & WCF_MASK) != 0;
      const bool pre_acc = !pre_mask && (flags & WCF_ACCUM) != 0;
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        const WinoRow R = rows[rr + i * RS];
        ok[i] = nok && (R.valid & need) == need;
        const int o = R.off + pa * R.wn + pb * N + n;
        dptr[i] = R.dst + o;
        const float* src = pre_mask ? R.mask + o : dptr[i];
        pre[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if ((pre_mask || pre_acc) && ok[i]) pre[i] = *reinterpret_cast<const f32x4*>(src);
      }
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        if (!ok[i]) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(stage + (rr + i * RS) * CS + c4 * 4) + bv;
        if (flags & WCF_ACCUM) v += pre_acc ? pre[i] : *reinterpret_cast<const f32x4*>(dptr[i]);
        if (flags & WCF_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (flags & WCF_SIGMOID) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
        }
        if (pre_mask) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = pre[i][e] > 0.f ? v[e] : 0.f;
        }
        *reinterpret_cast<f32x4*>(dptr[i]) = v;
        ssum += v;
        ssq += v * v;
      }
    }
    if (flags & WCF_STATS) {
      __syncthreads();
      float* red = smem;  // [2][RS][BN]
      *reinterpret_cast<f32x4*>(red + (size_t)rr * BN + c4 * 4) = ssum;
      *reinterpret_cast<f32x4*>(red + (size_t)(RS + rr) * BN + c4 * 4) = ssq;
      __syncthreads();
      for (int c = tid_e; c < 2 * BN; c += NT) {
        const int which = c / BN, col = c - which * BN;
        float sum = 0.f;
        for (int l = 0; l < RS; ++l) sum += red[(size_t)(which * RS + l) * BN + col];
        // statistics rows are per 64 tiles (sacv_wino_stats_rows): a 128-row item owns two of them
        constexpr int SR = BM / 64;
        if (n0 + col < N) {
          P.stats[((size_t)mt * SR * 2 + which) * N + n0 + col] = sum;
          if (SR == 2 && (mt * SR + 1) * 64 < P.T) P.stats[((size_t)(mt * SR + 1) * 2 + which) * N + n0 + col] = 0.f;
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------- weight gradient
// dU_g[n][c] = sum_t dM_g[t][n] V_g[t][c] with dM = A dY A^T (the 2x2 output-gradient tile expanded to 4x4), then
// dW = G^T dU G: 16 split-K (over tiles) TN GEMMs in one launch, 2.25x fewer matrix-core flops than conv_tn_kernel.
struct WinoDyArgs {
  WinoGeom g;
  const float* dy[WINO_MAXL];  // per level [Nb][H][W][N]
  void* dM;                    // fp32 [16][Tpad][N], or three bf16 planes [16][3][Tpad][N] (plain rows)
  int N, T, Tpad;
};

// (A dY A^T)[g] of one tile, 4 channels: fp32, or hi / mid / lo bf16 planes (x = hi + mid + lo exactly)
template <bool SPLIT>
__device__ __forceinline__ void wino_put_dm(void* base, size_t gs, int g, size_t off, const f32x4 v) {
  if constexpr (!SPLIT) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + (size_t)g * gs + off) = v;
  } else {
    wbf16x4 h, m, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = (wbf16)v[e];
      const float r1 = v[e] - (float)h[e];
      m[e] = (wbf16)r1;
      l[e] = (wbf16)(r1 - (float)m[e]);
    }
    wbf16* b = reinterpret_cast<wbf16*>(base) + (size_t)g * 3 * gs + off;
    *reinterpret_cast<wbf16x4*>(b) = h;
    *reinterpret_cast<wbf16x4*>(b + gs) = m;
    *reinterpret_cast<wbf16x4*>(b + 2 * gs) = l;
  }
}

template <bool SPLIT>
__global__ __launch_bounds__(256) void wino_dy_kernel(const WinoDyArgs a) {
  const int c4n = a.N >> 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)a.Tpad * c4n) return;
  const int t = (int)(idx / c4n), c4 = (int)(idx - (long long)t * c4n);
  const size_t gs = (size_t)a.Tpad * a.N;
  const size_t off = (size_t)t * a.N + c4 * 4;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  if (t >= a.T) {
#pragma unroll
    for (int g = 0; g < 16; ++g) wino_put_dm<SPLIT>(a.dM, gs, g, off, z);
    return;
  }
  const WinoTile tl = wino_tile(a.g, t);
  const float* src = a.dy[0];
  int H = a.g.H[0], W = a.g.W[0];
#pragma unroll
  for (int l = 1; l < WINO_MAXL; ++l)
    if (tl.lev == l) {
      src = a.dy[l];
      H = a.g.H[l];
      W = a.g.W[l];
    }
  const int h0 = 2 * tl.ty, w0 = 2 * tl.tx;
  const float* base = src + (((long long)tl.n * H + h0) * W + w0) * a.N + c4 * 4;
  const bool hv = h0 + 1 < H, wv = w0 + 1 < W;
  const f32x4 d00 = *reinterpret_cast<const f32x4*>(base);
  const f32x4 d01 = wv ? *reinterpret_cast<const f32x4*>(base + a.N) : z;
  const f32x4 d10 = hv ? *reinterpret_cast<const f32x4*>(base + (long long)W * a.N) : z;
  const f32x4 d11 = (hv && wv) ? *reinterpret_cast<const f32x4*>(base + ((long long)W + 1) * a.N) : z;
  // rows: A d = (d0, d0 + d1, d0 - d1, -d1); then the same along the columns
  const f32x4 r[4][2] = {{d00, d01}, {d00 + d10, d01 + d11}, {d00 - d10, d01 - d11}, {-d10, -d11}};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    wino_put_dm<SPLIT>(a.dM, gs, i * 4 + 0, off, r[i][0]);
    wino_put_dm<SPLIT>(a.dM, gs, i * 4 + 1, off, r[i][0] + r[i][1]);
    wino_put_dm<SPLIT>(a.dM, gs, i * 4 + 2, off, r[i][0] - r[i][1]);
    wino_put_dm<SPLIT>(a.dM, gs, i * 4 + 3, off, -r[i][1]);
  }
}

struct WinoWgradArgs {
  const void* dM;   // fp32 [16][Tpad][Cout]; wino_wgrad3_kernel: bf16 [16][3][Tpad][Cout]
  const void* V;    // fp32 [16][Tpad][Kp], or the split form bf16 [16][3][Tpad][Kp] (rebuilt exactly: hi + mid + lo)
  float* part;      // [nchunks][Cout][16 * Kp]
  int Tpad, Cout, Kp;
  int nchunks, chunk;  // tile rows per split (multiple of 32)
};

// Same tile engine as conv_tn_kernel (conv_gemm.hip): 32-row K tiles of both operands staged [32][BM] / [32][BN] in
// LDS, register prefetch of the next tile, interleaved b64 fragment reads; the operands are plain row-major matrices
// here (no spatial gather).  A workgroup = (Cout tile, (g, Cin) tile, split of the tile rows); Kp % BN == 0, so a
// tile never straddles two g.
template <int BM, int BN, bool VSPLIT>
__global__ __launch_bounds__(256, 2) void wino_wgrad_kernel(const WinoWgradArgs a) {
  constexpr int RA = BM / 32, RB = BN / 32;
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                // [2][32][BM]
  float* sB = smem + 2 * 32 * BM;  // [2][32][BN]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int N2 = 16 * a.Kp;
  const int ntm = (a.Cout + BM - 1) / BM;
  const int ntn = N2 / BN;
  int bid = blockIdx.x;
  const int tm = bid % ntm;
  bid /= ntm;
  const int tn = bid % ntn;
  const int ck = bid / ntn;
  const int pix0 = ck * a.chunk;
  const int pix1 = min(pix0 + a.chunk, a.Tpad);
  const int m0 = tm * BM, n0 = tn * BN;
  const int g = n0 / a.Kp, c0 = n0 - g * a.Kp;
  const float* __restrict__ dm = reinterpret_cast<const float*>(a.dM) + (size_t)g * a.Tpad * a.Cout;
  const size_t vps = (size_t)a.Tpad * a.Kp;  // one plane of V
  const float* __restrict__ v = reinterpret_cast<const float*>(a.V) + (size_t)g * vps + c0;
  const wbf16* __restrict__ v3a = reinterpret_cast<const wbf16*>(a.V) + (size_t)g * 3 * vps;

  constexpr int CGA = BM / 4, RSA = 256 / CGA;
  constexpr int CGB = BN / 4, RSB = 256 / CGB;
  const int ca = tid % CGA, rra = tid / CGA;
  const int cb = tid % CGB, rrb = tid / CGB;
  const int co = m0 + ca * 4;
  const bool co_ok = co < a.Cout;  // Cout % 4 == 0

  f32x4 ra[RA], rb[RB];
  auto load_a = [&](int i, int p0) {
    const int pix = p0 + rra + i * RSA;
    const float* ptr = dm + (unsigned)(pix * a.Cout + co);
    ptr = co_ok ? ptr : wino_zero16;
    ra[i] = *reinterpret_cast<const f32x4*>(ptr);
  };
  auto load_b = [&](int i, int p0) {
    const int pix = p0 + rrb + i * RSB;
    if constexpr (VSPLIT) {
      const wbf16* q = v3a + wino_split_off(pix, c0 + cb * 4, a.Kp);
      const wbf16x4 h = *reinterpret_cast<const wbf16x4*>(q), m = *reinterpret_cast<const wbf16x4*>(q + vps),
                    l = *reinterpret_cast<const wbf16x4*>(q + 2 * vps);
#pragma unroll
      for (int e = 0; e < 4; ++e) rb[i][e] = ((float)h[e] + (float)m[e]) + (float)l[e];  // both sums are exact
    } else {
      rb[i] = *reinterpret_cast<const f32x4*>(v + (unsigned)(pix * a.Kp + cb * 4));
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

  const int KT = (pix1 - pix0) >> 5;  // chunk and Tpad are multiples of 32
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
    const int pnext = have_next ? pix0 + ((kt + 1) << 5) : pix0;  // the last prefetch re-reads a valid tile, unused
    const float* cA = sA + cur * 32 * BM + wm0 + (TM == 2 ? 2 * l31 : l31);
    const float* cB = sB + cur * 32 * BN + wn0 + (TN == 2 ? 2 * l31 : l31);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto read_a = [&](float (&dst)[TM], int krow) {
      if (TM == 2) {
        f32x2 q = *reinterpret_cast<const f32x2*>(cA + krow * BM);
        dst[0] = q[0];
        dst[TM - 1] = q[1];
      } else {
        dst[0] = cA[krow * BM];
      }
    };
    auto read_b = [&](float (&dst)[TN], int krow) {
      if (TN == 2) {
        f32x2 q = *reinterpret_cast<const f32x2*>(cB + krow * BN);
        dst[0] = q[0];
        dst[TN - 1] = q[1];
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
      if ((s & 3) == 0 && (s >> 2) < RA) load_a(s >> 2, pnext);
      if ((s & 3) == 2 && (s >> 2) < RB) load_b(s >> 2, pnext);
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
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        if (m == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    if (have_next) store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  constexpr int CS = BN + 4;
  float* stage = smem;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ri = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int row = wm0 + (TM == 2 ? 2 * ri + i : ri);
        const int col = wn0 + (TN == 2 ? 2 * l31 + j : l31);
        stage[row * CS + col] = acc[i][j][r];
      }
  __syncthreads();
  float* __restrict__ part = a.part + (size_t)ck * a.Cout * N2;
  constexpr int CG = BN / 4, RS = 256 / CG;
  const int c4 = tid % CG, rr = tid / CG;
  const int n = n0 + c4 * 4;
#pragma unroll 4
  for (int row = rr; row < BM; row += RS) {
    const int m = m0 + row;
    if (m < a.Cout)
      *reinterpret_cast<f32x4*>(part + (size_t)m * N2 + n) = *reinterpret_cast<const f32x4*>(stage + row * CS + c4 * 4);
  }
}

// The same split-K GEMMs on the bf16 pipe: dM and V as three bf16 planes each (exact, file header), six products per
// 16 tiles of K into fp32 accumulators.  The contraction runs over TILES, the slow index of both operands, so the
// operand images in LDS are [16 tiles][channels] and the fragments come from the transposing LDS read
// (ds_read_b64_tr_b16, as conv_tn_bf16_kernel of conv_bf16.hip: per 16-lane group a block of 4 tile rows x 16 channels;
// lane 4q + p addresses row q, channels 4p..4p+3 and receives channel (lane & 15) of rows 0..3).  Rows are padded by
// 64 bytes: the four rows of a read fall on disjoint banks.  Tiles of 16 K double-buffered in LDS (61 KB for the
// 128 x 128 item, two workgroups per CU), the next tile's 6 x 16 bytes per thread prefetched into registers.
typedef short ws16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int wu32x4 __attribute__((ext_vector_type(4)));

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void wino_wgrad3_kernel(const WinoWgradArgs a) {
  constexpr int RSA = BM * 2 + 64, RSB = BN * 2 + 64;  // row strides in bytes
  constexpr int CGA = BM / 8, CGB = BN / 8;            // 16-byte chunks per tile row
  constexpr int LA = (3 * 16 * CGA + 255) / 256, LB = (3 * 16 * CGB + 255) / 256;  // loads per thread and K tile
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int IMG_A = 3 * 16 * RSA, IMG_B = 3 * 16 * RSB;  // bytes of one buffer: [3 planes][16 rows][RS]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  unsigned char* sA = smem3;              // [2][IMG_A]
  unsigned char* sB = smem3 + 2 * IMG_A;  // [2][IMG_B]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int N2 = 16 * a.Kp;
  const int ntm = (a.Cout + BM - 1) / BM;
  const int ntn = N2 / BN;
  int bid = blockIdx.x;
  const int tm = bid % ntm;
  bid /= ntm;
  const int tn = bid % ntn;
  const int ck = bid / ntn;
  const int pix0 = ck * a.chunk;
  const int pix1 = min(pix0 + a.chunk, a.Tpad);
  const int m0 = tm * BM, n0 = tn * BN;
  const int g = n0 / a.Kp, c0 = n0 - g * a.Kp;
  const size_t dps = (size_t)a.Tpad * a.Cout, vps = (size_t)a.Tpad * a.Kp;  // one plane
  const wbf16* __restrict__ dm = reinterpret_cast<const wbf16*>(a.dM) + (size_t)g * 3 * dps;
  const wbf16* __restrict__ v3 = reinterpret_cast<const wbf16*>(a.V) + (size_t)g * 3 * vps;

  // this thread's chunks of a K tile: (plane, row, 16-byte chunk) of A (dM, plain rows) and B (V, wino_split_off)
  int a_pl[LA], a_row[LA], a_col[LA], b_pl[LB], b_row[LB], b_col[LB];
  bool a_ok[LA], b_ok[LB];
#pragma unroll
  for (int i = 0; i < LA; ++i) {
    const int idx = tid + i * 256;
    a_pl[i] = idx / (16 * CGA);
    const int rem = idx - a_pl[i] * 16 * CGA;
    a_row[i] = rem / CGA;
    a_col[i] = (rem - a_row[i] * CGA) * 8;
    a_ok[i] = idx < 3 * 16 * CGA && m0 + a_col[i] < a.Cout;  // Cout % 8 == 0
  }
#pragma unroll
  for (int i = 0; i < LB; ++i) {
    const int idx = tid + i * 256;
    b_pl[i] = idx / (16 * CGB);
    const int rem = idx - b_pl[i] * 16 * CGB;
    b_row[i] = rem / CGB;
    b_col[i] = (rem - b_row[i] * CGB) * 8;
    b_ok[i] = idx < 3 * 16 * CGB;
  }
  wu32x4 ra[LA], rb[LB];
  auto load_tile = [&](int p0) {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const void* ptr = a_ok[i] ? (const void*)(dm + a_pl[i] * dps + (size_t)(p0 + a_row[i]) * a.Cout + m0 + a_col[i])
                                : (const void*)wino_zero16;
      ra[i] = *reinterpret_cast<const wu32x4*>(ptr);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const void* ptr = b_ok[i] ? (const void*)(v3 + b_pl[i] * vps + wino_split_off(p0 + b_row[i], c0 + b_col[i], a.Kp))
                                : (const void*)wino_zero16;
      rb[i] = *reinterpret_cast<const wu32x4*>(ptr);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LA; ++i)
      if (3 * 16 * CGA % 256 == 0 || tid + i * 256 < 3 * 16 * CGA)
        *reinterpret_cast<wu32x4*>(sA + buf * IMG_A + (a_pl[i] * 16 + a_row[i]) * RSA + a_col[i] * 2) = ra[i];
#pragma unroll
    for (int i = 0; i < LB; ++i)
      if (3 * 16 * CGB % 256 == 0 || tid + i * 256 < 3 * 16 * CGB)
        *reinterpret_cast<wu32x4*>(sB + buf * IMG_B + (b_pl[i] * 16 + b_row[i]) * RSB + b_col[i] * 2) = rb[i];
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
  const int tr_row = 8 * (tg >> 1) + tq;      // + 4 for the second read
  const int tr_col = 16 * (tg & 1) + 4 * tp;  // channel inside the 32-wide MFMA block

  const int KT = (pix1 - pix0) >> 4;  // chunk and Tpad are multiples of 32
  if (KT > 0) {
    load_tile(pix0);
    store_tile(0);
  }
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < KT; ++kt) {
    const bool last = kt + 1 == KT;
    if (!last) load_tile(pix0 + ((kt + 1) << 4));
    const unsigned char* cA = sA + cur * IMG_A + tr_row * RSA + (wm0 + tr_col) * 2;
    const unsigned char* cB = sB + cur * IMG_B + tr_row * RSB + (wn0 + tr_col) * 2;
    wbf16x8 fa[3][TM], fb[3][TN];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const unsigned char* pA = cA + pl * 16 * RSA + i * 64;
        union { ws16x4 s[2]; wbf16x8 v; } u;
        u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ws16x4*)(pA));
        u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ws16x4*)(pA + 4 * RSA));
        fa[pl][i] = u.v;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const unsigned char* pB = cB + pl * 16 * RSB + j * 64;
        union { ws16x4 s[2]; wbf16x8 v; } u;
        u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ws16x4*)(pB));
        u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ws16x4*)(pB + 4 * RSB));
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
    if (!last) store_tile(cur ^ 1);
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
  float* __restrict__ part = a.part + (size_t)ck * a.Cout * N2;
  constexpr int CG = BN / 4, RS = 256 / CG;
  const int c4 = tid % CG, rr = tid / CG;
  const int n = n0 + c4 * 4;
#pragma unroll 4
  for (int row = rr; row < BM; row += RS) {
    const int m = m0 + row;
    if (m < a.Cout)
      *reinterpret_cast<f32x4*>(part + (size_t)m * N2 + n) = *reinterpret_cast<const f32x4*>(stage + row * CS + c4 * 4);
  }
}

// dw[co][kh][kw][ci] (channels_last, dw_ohwi = 1) or [co][ci][kh][kw] (=)/(+=) (G^T dU G)[kh][kw], dU summed over the
// splits in a fixed order.  A block = 16 outputs (co, 4 input channels) x 16 g: thread (g, o) sums the splits of its
// dU_g (coalesced 256-byte rows per g), the 16 sums of an output meet in LDS and 16 threads apply G^T . G.
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                 int nchunks, int Cout, int Cin, int Kp, int accumulate,
                                                                 int dw_ohwi) {
  __shared__ f32x4 us[16][16];
  const int c4n = Cin >> 2;
  const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + o;
  const bool ok = idx < Cout * c4n;
  const int co = ok ? idx / c4n : 0, ci = ok ? (idx - co * c4n) * 4 : 0;
  const size_t N2 = (size_t)16 * Kp, slab = (size_t)Cout * N2;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
    const float* p = part + (size_t)co * N2 + (size_t)g * Kp + ci;
    for (int c = 0; c < nchunks; ++c) s += *reinterpret_cast<const f32x4*>(p + c * slab);
  }
  us[g][o] = s;
  __syncthreads();
  if (g != 0 || !ok) return;
  f32x4 u[4][4];
#pragma unroll
  for (int q = 0; q < 16; ++q) u[q >> 2][q & 3] = us[q][o];
  f32x4 t[3][4];  // G^T u: rows kh = (u0 + (u1 + u2)/2, (u1 - u2)/2, u3 + (u1 + u2)/2)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 hs = 0.5f * (u[1][j] + u[2][j]);
    t[0][j] = u[0][j] + hs;
    t[1][j] = 0.5f * (u[1][j] - u[2][j]);
    t[2][j] = u[3][j] + hs;
  }
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const f32x4 hs = 0.5f * (t[kh][1] + t[kh][2]);
    const f32x4 w3[3] = {t[kh][0] + hs, 0.5f * (t[kh][1] - t[kh][2]), t[kh][3] + hs};
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      if (dw_ohwi) {
        float* d = dw + ((size_t)co * 9 + kh * 3 + kw) * Cin + ci;
        f32x4 val = w3[kw];
        if (accumulate) val += *reinterpret_cast<const f32x4*>(d);
        *reinterpret_cast<f32x4*>(d) = val;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float* d = dw + ((size_t)co * Cin + ci + e) * 9 + kh * 3 + kw;
          *d = accumulate ? *d + w3[kw][e] : w3[kw][e];
        }
      }
    }
  }
}

#ifndef WINO_WN
#define WINO_WN 2  // waves along N of the GEMM kernel (tuning builds: -DWINO_WN=4: 8 waves of 32 x 32, same speed)
#endif
template __global__ void wino_gemm_kernel<64, 128, WINO_WN, WINO_WN, false>(const WinoArgs);
template __global__ void wino_gemm_kernel<64, 64, 2, 3, false>(const WinoArgs);
#ifndef WINO_SPLIT_OCC
#define WINO_SPLIT_OCC 2  // workgroups per CU of the split-operand GEMM (74 KB of LDS each)
#endif
template __global__ void wino_gemm_kernel<64, 64, 2, WINO_SPLIT_OCC, true>(const WinoArgs);
template __global__ void wino_gemm_kernel<128, 128, 2, 1, true>(const WinoArgs);

// -------------------------------------------------------------------------------------------- host side
static int wino_num_cus() {
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

// dynamic LDS limit of one kernel instantiation, set once per process
template <auto Kern>
static void wino_set_lds(int lds) {
  static std::once_flag once;
  std::call_once(once, [lds] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
}

static int wino_fill_geom(WinoGeom& g, int nlev, int Nb, const int* H, const int* W, const char* who) {
  SACV_CHECK_ARG(nlev >= 1 && nlev <= WINO_MAXL, "%s: %d levels (1..%d)", who, nlev, WINO_MAXL);
  SACV_CHECK_ARG(Nb > 0, "%s: batch", who);
  g.nlev = nlev;
  g.Nb = Nb;
  long long t = 0;
  for (int l = 0; l < WINO_MAXL; ++l) {
    g.tile_begin[l] = (int)t;
    if (l < nlev) {
      SACV_CHECK_ARG(H[l] > 0 && W[l] > 0, "%s: bad level dims", who);
      g.H[l] = H[l];
      g.W[l] = W[l];
      g.th[l] = (H[l] + 1) / 2;
      g.tw[l] = (W[l] + 1) / 2;
      t += (long long)Nb * g.th[l] * g.tw[l];
    } else {
      g.H[l] = g.W[l] = g.th[l] = g.tw[l] = 1;
    }
  }
  SACV_CHECK_ARG(t < (1LL << 24), "%s: too many tiles", who);
  g.tile_begin[WINO_MAXL] = (int)t;
  for (int l = nlev; l <= WINO_MAXL; ++l) g.tile_begin[l] = (int)t;
  return SACV_OK;
}

SACV_API_BEGIN

// Number of 2x2 output tiles of a pyramid (rows of the transformed operand V before padding to a multiple of 64).
int sacv_wino_tiles(int nlev, int Nb, const int* H, const int* W) {
  WinoGeom g;
  if (wino_fill_geom(g, nlev, Nb, H, W, "wino_tiles") != SACV_OK) return -1;
  return g.tile_begin[WINO_MAXL];
}

// Operand padding: the GEMM reads whole 64 x 32 (V) and 128 x 32 (U) tiles without masks, so rows / columns are padded
// with zeros to these granules by the two transform kernels.
int sacv_wino_pad_k(int K) { return (K + 31) / 32 * 32; }
int sacv_wino_pad_n(int N) { return (N + 127) / 128 * 128; }

// U[16][pad_n(N)][pad_k(K)] = G w G^T from a GEMM weight panel [N][9][K] (conv_gemm.hip forward panel: N = Cout,
// K = Cin; data-gradient panel: N = Cin, K = Cout with flip = 1, the 180-degree rotated kernel of the transposed
// convolution).
int sacv_wino_weights(const float* panel, void* U, int N, int K, int flip, int split, void* stream) {
  SACV_CHECK_ARG(panel && U, "wino_weights: null pointer");
  SACV_CHECK_ARG(N > 0 && K > 0 && K % 4 == 0, "wino_weights: N=%d K=%d (K must be a multiple of 4)", N, K);
  const int Np = sacv_wino_pad_n(N), Kp = sacv_wino_pad_k(K);
  SACV_CHECK_ARG((long long)Np * Kp * 16 < (1LL << 30), "wino_weights: panel too large");
  const int total = Np * (Kp / 4);
  if (split)
    hipLaunchKernelGGL(wino_weight_kernel<true>, dim3(sacv_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, panel, U,
                       N, K, Np, Kp, flip);
  else
    hipLaunchKernelGGL(wino_weight_kernel<false>, dim3(sacv_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, panel, U,
                       N, K, Np, Kp, flip);
  SACV_CHECK_LAUNCH("wino_weight_kernel");
  return SACV_OK;
}

// njobs (<= 32) weight transforms in one launch: U[i] = transform of panel[i] ([N[i]][9][K[i]], flip[i])
int sacv_wino_weights_batched(int njobs, const float* const* panel, void* const* U, const int* N, const int* K,
                              const int* flip, int split, void* stream) {
  SACV_CHECK_ARG(njobs >= 1 && njobs <= WINO_WBATCH, "wino_weights_batched: njobs=%d (1..%d)", njobs, WINO_WBATCH);
  WinoWeightJobs j;
  int maxblocks = 0;
  for (int i = 0; i < WINO_WBATCH; ++i) {
    const int k = i < njobs ? i : 0;
    SACV_CHECK_ARG(panel[k] && U[k] && N[k] > 0 && K[k] > 0 && K[k] % 4 == 0, "wino_weights_batched: bad job %d", k);
    j.panel[i] = panel[k];
    j.U[i] = U[k];
    j.N[i] = N[k];
    j.K[i] = K[k];
    j.flip[i] = flip[k];
    const int total = sacv_wino_pad_n(N[k]) * (sacv_wino_pad_k(K[k]) / 4);
    SACV_CHECK_ARG((long long)sacv_wino_pad_n(N[k]) * sacv_wino_pad_k(K[k]) * 16 < (1LL << 30),
                   "wino_weights_batched: panel too large");
    if (i < njobs && sacv_cdiv(total, 256) > maxblocks) maxblocks = sacv_cdiv(total, 256);
  }
  if (split)
    hipLaunchKernelGGL(wino_weight_batched_kernel<true>, dim3(maxblocks, njobs), dim3(256), 0, (hipStream_t)stream, j);
  else
    hipLaunchKernelGGL(wino_weight_batched_kernel<false>, dim3(maxblocks, njobs), dim3(256), 0, (hipStream_t)stream, j);
  SACV_CHECK_LAUNCH("wino_weight_batched_kernel");
  return SACV_OK;
}

// V[16][Tpad][pad_k(C)] = B^T d B of every 4x4 input patch (pad 1) of the NHWC level tensors
// src[l] = [Nb][H[l]][W[l]][C].  Tpad = tiles rounded up to a multiple of 64 (padding rows / channels are zero-filled).
int sacv_wino_input(int nlev, const float* const* src, int Nb, const int* H, const int* W, int C, void* V, int Tpad,
                    int split, void* stream) {
  WinoInArgs a;
  int rc = wino_fill_geom(a.g, nlev, Nb, H, W, "wino_input");
  if (rc != SACV_OK) return rc;
  SACV_CHECK_ARG(C > 0 && C % 4 == 0, "wino_input: C=%d must be a positive multiple of 4", C);
  a.T = a.g.tile_begin[WINO_MAXL];
  SACV_CHECK_ARG(Tpad >= a.T && Tpad % 64 == 0, "wino_input: Tpad=%d must be a multiple of 64 >= %d", Tpad, a.T);
  a.Cp = sacv_wino_pad_k(C);
  SACV_CHECK_ARG((long long)Tpad * a.Cp * 16 < (1LL << 30), "wino_input: V too large for 32-bit element indexing");
  SACV_CHECK_ARG(V, "wino_input: null V");
  for (int l = 0; l < WINO_MAXL; ++l) {
    a.src[l] = l < nlev ? src[l] : nullptr;
    SACV_CHECK_ARG(l >= nlev || src[l], "wino_input: null level pointer");
    SACV_CHECK_ARG(l >= nlev || (long long)Nb * H[l] * W[l] * C < (1LL << 30), "wino_input: level too large");
  }
  a.V = V;
  a.C = C;
  a.Tpad = Tpad;
  const long long total = (long long)Tpad * (a.Cp / 4);
  if (split)
    hipLaunchKernelGGL(wino_input_kernel<true>, dim3(sacv_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(wino_input_kernel<false>, dim3(sacv_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, a);
  SACV_CHECK_LAUNCH("wino_input_kernel");
  return SACV_OK;
}

// Rows of one problem's WCF_STATS buffer ([rows][2][N]): 64-tile M tiles.
int sacv_wino_stats_rows(int tiles) { return (tiles + 63) / 64; }

// 3x3 / stride 1 / pad 1 convolution of nprob (1 or 2) independent problems from their transformed operands.
// Problem p: V[p] = [16][Tpad[p]][pad_k(K[p])], U[p] = [16][pad_n(N[p])][pad_k(K[p])], outputs dst[p*5 + l] = [Nb][H[l]][W[l]][N[p]]
// (the pyramid geometry nlev / Nb / H / W is shared by the problems), optional bias[p], mask[p*5 + l] (WCF_MASK),
// stats[p] (WCF_STATS, sacv_wino_stats_rows rows).  flags[p]: the CF_* epilogue bits of sacv_conv2d_gemm.
int sacv_wino_gemm(int nprob, int nlev, int Nb, const int* H, const int* W, const void* const* V,
                   const void* const* U, const int* Tpad, const int* K, const int* N, const float* const* bias,
                   float* const* dst, const float* const* mask, float* const* stats, const int* flags, int split,
                   void* stream) {
  SACV_CHECK_ARG(nprob >= 1 && nprob <= WINO_MAXP, "wino_gemm: nprob %d (1..%d)", nprob, WINO_MAXP);
  WinoArgs a;
  int items = 0;
  for (int p = 0; p < nprob; ++p) {
    WinoProb& P = a.p[p];
    int rc = wino_fill_geom(P.g, nlev, Nb, H, W, "wino_gemm");
    if (rc != SACV_OK) return rc;
    P.T = P.g.tile_begin[WINO_MAXL];
    SACV_CHECK_ARG(V[p] && U[p], "wino_gemm: null operand (problem %d)", p);
    SACV_CHECK_ARG(K[p] > 0 && K[p] % 4 == 0 && N[p] > 0 && N[p] % 4 == 0,
                   "wino_gemm: K=%d N=%d must be positive multiples of 4", K[p], N[p]);
    SACV_CHECK_ARG(Tpad[p] >= P.T && Tpad[p] % 64 == 0, "wino_gemm: Tpad");
    const int Kp = sacv_wino_pad_k(K[p]), Np = sacv_wino_pad_n(N[p]);
    SACV_CHECK_ARG((long long)Tpad[p] * Kp * 16 < (1LL << 30) && (long long)Np * Kp * 16 < (1LL << 30),
                   "wino_gemm: operand too large for 32-bit element indexing");
    SACV_CHECK_ARG((flags[p] & ~63) == 0, "wino_gemm: unknown flag bits 0x%x", flags[p] & ~63);
    P.V = V[p];
    P.U = U[p];
    P.bias = bias ? bias[p] : nullptr;
    SACV_CHECK_ARG(!(flags[p] & WCF_BIAS) || P.bias, "wino_gemm: bias flag without bias pointer");
    P.stats = stats ? stats[p] : nullptr;
    SACV_CHECK_ARG(!(flags[p] & WCF_STATS) || P.stats, "wino_gemm: WCF_STATS without a stats buffer");
    for (int l = 0; l < WINO_MAXL; ++l) {
      P.dst[l] = l < nlev ? dst[p * WINO_MAXL + l] : nullptr;
      P.mask[l] = (l < nlev && mask) ? mask[p * WINO_MAXL + l] : nullptr;
      SACV_CHECK_ARG(l >= nlev || P.dst[l], "wino_gemm: null output pointer");
      SACV_CHECK_ARG(l >= nlev || !(flags[p] & WCF_MASK) || P.mask[l], "wino_gemm: mask flag without mask pointer");
      SACV_CHECK_ARG(l >= nlev || (long long)Nb * H[l] * W[l] * N[p] < (1LL << 30), "wino_gemm: output too large");
    }
    P.Tpad = Tpad[p];
    P.K = Kp;
    P.Np = Np;
    P.N = N[p];
    P.flags = flags[p];
    P.item_begin = 0;
    P.ntn = 0;
    P.mt0 = 0;
  }
  a.nprob = nprob;
  // Item shape: 64 tiles x 128 channels at 2 workgroups per CU, or x 64 channels at 3 per CU.  The wide item does 13 %
  // more work per CU-second (fewer operand bytes per flop) but a launch is a few hundred items on 512 slots, so the
  // makespan is a matter of rounds: measured on the 256-channel pyramid of 2 x 800x1333 (us, GEMM only) -- one tower
  // (352 wide items) 296 wide / 249 narrow; two towers (704) 451 / 459; the 720-channel output conv (1056) 722 / 684.
  // Cost model in units of one wide round: a narrow round is 0.845, partial rounds cost their measured fractions.
  int mt_items = 0, wide = 0, narrow_items = 0;
  for (int p = 0; p < nprob; ++p) {
    mt_items = (a.p[p].T + 63) / 64;
    wide += mt_items * ((a.p[p].N + 127) / 128);
    narrow_items += mt_items * ((a.p[p].N + 63) / 64);
  }
  int cus = wino_num_cus() - sacv_share.cu_reserve;
  if (cus < 8) cus = 8;
  const int occ_n = sacv_share.wino_wg_per_cu, occ_w = occ_n < 2 ? occ_n : 2;  // slots per CU (sacv_launch_share)
  auto rounds_cost = [](double r, double t1, double t2, double t3) {  // r = items / slots
    const double full = (double)(long long)r, f = r - full;
    const double tail = f <= 1e-9 ? 0.0 : (f <= 1.0 / 3 ? t1 : (f <= 2.0 / 3 ? t2 : t3));
    return full + tail;
  };
  const double cost_wide = rounds_cost((double)wide / (occ_w * cus), 0.45, 0.64, 1.0);
  double cost_narrow = 0.845 * rounds_cost((double)narrow_items / (occ_n * cus), 0.5, 0.7, 0.85);
  if (cost_narrow < 0.845) cost_narrow = 0.845;
  const bool narrow = split || cost_narrow < cost_wide;  // the split-operand kernel has the 64-channel item only
  const int bn = narrow ? 64 : 128;
  for (int p = 0; p < nprob; ++p) {
    WinoProb& P = a.p[p];
    P.item_begin = items;
    P.ntn = (P.N + bn - 1) / bn;
    items += ((P.T + 63) / 64) * P.ntn;
  }
  a.total_items = items;
  hipStream_t st = (hipStream_t)stream;
#ifndef WINO_BIG_COST
#define WINO_BIG_COST 1.05  // duration of a round of 128 x 128 items in units of a round of 64 x 64 items (two per CU)
#endif
  if (split) {
    // Two item shapes.  BIG: 128 tiles x 128 channels, 8 waves of 32 x 64, one workgroup per CU (148 KB of LDS) --
    // half the operand bytes per flop of the small item, which is what bounds this kernel (the L2 -> LDS rate, not
    // the matrix pipe).  SMALL: 64 x 64, 4 waves, two per CU.  A launch of a few hundred big items on 256 slots has a
    // poor last round, so the big kernel takes whole rounds of M tiles and the small kernel the rest of the rows.
    const int T = a.p[0].T;  // the problems of a launch share the pyramid geometry
    const int mt128 = a.p[0].Tpad / 128;  // whole 128-row tiles of V (Tpad is a multiple of 64 only)
    int ntn_big = 0, ntn_small = 0;
    for (int p = 0; p < nprob; ++p) {
      ntn_big += (a.p[p].N + 127) / 128;
      ntn_small += (a.p[p].N + 63) / 64;
    }
    const int occ_s = occ_n < WINO_SPLIT_OCC ? occ_n : WINO_SPLIT_OCC;
    auto plan_cost = [&](int mb) {  // mb M tiles of 128 rows to the big kernel; unit: one small item's duration
      const long long big = (long long)mb * ntn_big;
      const int rows_left = T - mb * 128;
      const long long small = rows_left > 0 ? (long long)((rows_left + 63) / 64) * ntn_small : 0;
      const double rb = (double)((big + cus - 1) / cus), rs = (double)((small + (long long)occ_s * cus - 1) / (occ_s * cus));
      return rb * WINO_BIG_COST + rs;
    };
    int mb = 0;
    double best = plan_cost(0);
    for (int r = 1; r <= 64 && mt128 > 0; ++r) {
      int cand = (int)((long long)r * cus / ntn_big);
      if (cand > mt128) cand = mt128;
      const double c = plan_cost(cand);
      if (c < best - 1e-9) {
        best = c;
        mb = cand;
      }
      if (cand == mt128) break;
    }
    if (sacv_share.wino_big == 0) mb = 0;
    if (sacv_share.wino_big == 2) mb = mt128;
    if (mb > 0) {
      constexpr int BM = 128, BN = 128;
      items = 0;
      for (int p = 0; p < nprob; ++p) {
        WinoProb& P = a.p[p];
        P.item_begin = items;
        P.ntn = (P.N + BN - 1) / BN;
        P.mt0 = 0;
        items += mb * P.ntn;
      }
      a.total_items = items;
      const int lds = 3 * (BM + BN) * 48 * 4 + BM * (int)sizeof(WinoRow);
      static_assert(3 * (BM + BN) * 48 >= BM * (BN + 4), "the epilogue stage must fit in the operand buffers");
      wino_set_lds<wino_gemm_kernel<BM, BN, 2, 1, true>>(lds);
      int grid = cus;
      if (grid > items) grid = items;
      hipLaunchKernelGGL((wino_gemm_kernel<BM, BN, 2, 1, true>), dim3(grid), dim3(512), lds, st, a);
      SACV_CHECK_LAUNCH("wino_gemm_kernel (split operands, 128 x 128 items)");
    }
    if (mb * 128 < T) {
      constexpr int BM = 64, BN = 64;
      items = 0;
      for (int p = 0; p < nprob; ++p) {
        WinoProb& P = a.p[p];
        P.item_begin = items;
        P.ntn = (P.N + BN - 1) / BN;
        P.mt0 = mb * 2;
        items += ((P.T + 63) / 64 - mb * 2) * P.ntn;
      }
      a.total_items = items;
      const int lds = 3 * (BM + BN) * 48 * 4 + BM * (int)sizeof(WinoRow);
      static_assert(3 * (BM + BN) * 48 >= BM * (BN + 4), "the epilogue stage must fit in the operand buffers");
      wino_set_lds<wino_gemm_kernel<BM, BN, 2, WINO_SPLIT_OCC, true>>(lds);
      int grid = cus * occ_s;
      if (grid > items) grid = items;
      hipLaunchKernelGGL((wino_gemm_kernel<BM, BN, 2, WINO_SPLIT_OCC, true>), dim3(grid), dim3(256), lds, st, a);
      SACV_CHECK_LAUNCH("wino_gemm_kernel (split operands)");
    }
    return SACV_OK;
  }
  if (narrow) {
    constexpr int BM = 64, BN = 64;
    const int lds = 3 * (BM + BN) * 32 * 4 + BM * (int)sizeof(WinoRow);
    static std::once_flag attr;
    std::call_once(attr, [lds] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_gemm_kernel<BM, BN, 2, 3, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    });
    int grid = cus * occ_n;
    if (grid > items) grid = items;
    hipLaunchKernelGGL((wino_gemm_kernel<BM, BN, 2, 3, false>), dim3(grid), dim3(256), lds, st, a);
    SACV_CHECK_LAUNCH("wino_gemm_kernel");
    return SACV_OK;
  }
  constexpr int BM = 64, BN = 128;
  constexpr int WN = WINO_WN;
  const int lds = 3 * (BM + BN) * 32 * 4 + BM * (int)sizeof(WinoRow);
  static_assert(2 * (BM + BN) * 32 >= BM * (BN + 4), "the epilogue stage must fit in the operand buffers");
  static std::once_flag attr;
  std::call_once(attr, [lds] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_gemm_kernel<BM, BN, WN, WN, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  int grid = cus * occ_w;
  if (grid > items) grid = items;
  hipLaunchKernelGGL((wino_gemm_kernel<BM, BN, WN, WN, false>), dim3(grid), dim3(128 * WN), lds, st, a);
  SACV_CHECK_LAUNCH("wino_gemm_kernel");
  return SACV_OK;
}

// Workspace of sacv_wino_wgrad in bytes: dM [16][Tpad][Cout] followed by the split slabs [nchunks][Cout][16 * Kp].
static int wino_wgrad_plan(int Tpad, int Cout, int Kp, int* BNout, int* chunk_out) {
  const int BN = (Kp % 128 == 0) ? 128 : 64;
  const int tiles = ((Cout + 127) / 128) * (16 * Kp / BN);
  int cus = wino_num_cus() - sacv_share.cu_reserve;
  if (cus < 8) cus = 8;
  const int slots = sacv_share.wgrad_wg_per_cu * cus;
  int best = 1;
  double best_cost = 1e30;
  for (int c = 1; c <= 32; ++c) {
    const int chunk = ((Tpad + c - 1) / c + 31) / 32 * 32;
    if (chunk < 256 && c > 1) break;
    const int n = (Tpad + chunk - 1) / chunk;
    const double rounds = (double)(((long long)tiles * n + slots - 1) / slots);
    const double cost = rounds * (chunk + 48.0) + 0.02 * n * 32;  // + epilogue per workgroup + slab traffic
    if (cost < best_cost) {
      best_cost = cost;
      best = chunk;
    }
  }
  *BNout = BN;
  *chunk_out = best;
  return (Tpad + best - 1) / best;
}

long long sacv_wino_wgrad_workspace_bytes(int Tpad, int Cout, int K) {
  if (Tpad <= 0 || Cout <= 0 || K <= 0) return -1;
  const int Kp = sacv_wino_pad_k(K);
  int BN, chunk;
  const int n = wino_wgrad_plan(Tpad, Cout, Kp, &BN, &chunk);
  // dM: fp32 [16][Tpad][Cout] or three bf16 planes (6 bytes per element: sized for those), then the fp32 slabs
  return 6LL * 16 * Tpad * Cout + 4LL * n * Cout * 16 * Kp;
}

// Weight gradient of a 3x3 / stride 1 / pad 1 convolution over a pyramid: dw (+)= sum over levels x (*) dy, from
// V = sacv_wino_input(x) (kept from the forward pass) and the output gradients dy[l] = [Nb][H[l]][W[l]][Cout].
// dw_layout 0: OIHW, 1: [Cout][3][3][Cin] (channels_last).  Cin must be a multiple of 64.
int sacv_wino_wgrad(int nlev, const float* const* dy, int Nb, const int* H, const int* W, int Cout, const void* V,
                    int v_split, int Tpad, int Cin, float* dw, int accumulate, int dw_layout, float* workspace,
                    void* stream) {
  WinoDyArgs d;
  int rc = wino_fill_geom(d.g, nlev, Nb, H, W, "wino_wgrad");
  if (rc != SACV_OK) return rc;
  d.T = d.g.tile_begin[WINO_MAXL];
  SACV_CHECK_ARG(V && dw && workspace, "wino_wgrad: null pointer");
  SACV_CHECK_ARG(Cout > 0 && Cout % 4 == 0 && Cin > 0 && Cin % 4 == 0, "wino_wgrad: Cout=%d Cin=%d must be multiples of 4",
                 Cout, Cin);
  const int Kp = sacv_wino_pad_k(Cin);
  SACV_CHECK_ARG(Kp % 64 == 0, "wino_wgrad: Cin=%d padded to %d is not a multiple of 64 (use sacv_conv2d_wgrad)", Cin, Kp);
  SACV_CHECK_ARG(Tpad >= d.T && Tpad % 64 == 0, "wino_wgrad: Tpad");
  SACV_CHECK_ARG((long long)Tpad * Cout * 16 < (1LL << 30) && (long long)Tpad * Kp * 16 < (1LL << 30) &&
                     (long long)Cout * Kp * 16 < (1LL << 30),
                 "wino_wgrad: operand too large for 32-bit element indexing");
  for (int l = 0; l < WINO_MAXL; ++l) {
    d.dy[l] = l < nlev ? dy[l] : nullptr;
    SACV_CHECK_ARG(l >= nlev || dy[l], "wino_wgrad: null level pointer");
    SACV_CHECK_ARG(l >= nlev || (long long)Nb * H[l] * W[l] * Cout < (1LL << 30), "wino_wgrad: level too large");
  }
  hipStream_t st = (hipStream_t)stream;
  SACV_CHECK_ARG(v_split >= 0 && v_split <= 2, "wino_wgrad: v_split=%d (0 fp32 V, 1 split V / fp32 pipe, 2 bf16 pipe)", v_split);
  const bool pipe16 = v_split == 2;
  SACV_CHECK_ARG(!pipe16 || Cout % 8 == 0, "wino_wgrad: the bf16-pipe form needs Cout %% 8 == 0 (Cout=%d)", Cout);
  d.dM = workspace;
  d.N = Cout;
  d.Tpad = Tpad;
  const long long total = (long long)Tpad * (Cout / 4);
  if (pipe16)
    hipLaunchKernelGGL(wino_dy_kernel<true>, dim3(sacv_cdiv(total, 256)), dim3(256), 0, st, d);
  else
    hipLaunchKernelGGL(wino_dy_kernel<false>, dim3(sacv_cdiv(total, 256)), dim3(256), 0, st, d);
  SACV_CHECK_LAUNCH("wino_dy_kernel");

  WinoWgradArgs a;
  int BN;
  a.nchunks = wino_wgrad_plan(Tpad, Cout, Kp, &BN, &a.chunk);
  a.dM = workspace;
  a.V = V;
  a.part = workspace + ((size_t)16 * Tpad * Cout * 3 + 1) / 2;  // behind dM in either form (6 bytes per element)
  a.Tpad = Tpad;
  a.Cout = Cout;
  a.Kp = Kp;
  const int grid = ((Cout + 127) / 128) * (16 * Kp / BN) * a.nchunks;
#define WINO_WGRAD_LAUNCH(BN_, SPLIT_, LDS_)                                                             \
  do {                                                                                                  \
    wino_set_lds<wino_wgrad_kernel<128, BN_, SPLIT_>>(LDS_);                                            \
    hipLaunchKernelGGL((wino_wgrad_kernel<128, BN_, SPLIT_>), dim3(grid), dim3(256), LDS_, st, a);      \
  } while (0)
  if (pipe16) {
    if (BN == 128) {
      const int img = 2 * 3 * 16 * ((128 * 2 + 64) + (128 * 2 + 64));
      const int lds = img > 128 * 132 * 4 ? img : 128 * 132 * 4;
      wino_set_lds<wino_wgrad3_kernel<128, 128>>(lds);
      hipLaunchKernelGGL((wino_wgrad3_kernel<128, 128>), dim3(grid), dim3(256), lds, st, a);
    } else {
      const int img = 2 * 3 * 16 * ((128 * 2 + 64) + (64 * 2 + 64));
      const int lds = img > 128 * 68 * 4 ? img : 128 * 68 * 4;
      wino_set_lds<wino_wgrad3_kernel<128, 64>>(lds);
      hipLaunchKernelGGL((wino_wgrad3_kernel<128, 64>), dim3(grid), dim3(256), lds, st, a);
    }
  } else if (BN == 128) {
    const int lds = 2 * 32 * (128 + 128) * 4 > 128 * 132 * 4 ? 2 * 32 * (128 + 128) * 4 : 128 * 132 * 4;
    if (v_split)
      WINO_WGRAD_LAUNCH(128, true, lds);
    else
      WINO_WGRAD_LAUNCH(128, false, lds);
  } else {
    const int lds = 2 * 32 * (128 + 64) * 4 > 128 * 68 * 4 ? 2 * 32 * (128 + 64) * 4 : 128 * 68 * 4;
    if (v_split)
      WINO_WGRAD_LAUNCH(64, true, lds);
    else
      WINO_WGRAD_LAUNCH(64, false, lds);
  }
#undef WINO_WGRAD_LAUNCH
  SACV_CHECK_LAUNCH("wino_wgrad_kernel");
  const int rtotal = Cout * (Cin / 4);
  hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(sacv_cdiv(rtotal, 16)), dim3(256), 0, st, a.part, dw, a.nchunks, Cout,
                     Cin, Kp, accumulate, dw_layout);
  SACV_CHECK_LAUNCH("wino_wgrad_reduce_kernel");
  return SACV_OK;
}

SACV_API_END
