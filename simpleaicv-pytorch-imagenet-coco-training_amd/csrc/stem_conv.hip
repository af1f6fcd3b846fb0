// Forward pass of the stem convolution (7x7 / stride 2 / pad 3, Cin = 3 -> 64: classification/backbones/resnet.py:176-181,
// reached through detection/models/backbones/resnet.py:96-100) WITHOUT the im2col matrix.
//
// Cin = 3 is too narrow for the implicit-GEMM kernel's 16-byte channel gathers, so rounds 1-2 wrote the [pixels][160]
// im2col matrix (341 MB at 2 x 800x1333: 144 us of writes at 2.5 TB/s) and ran a GEMM over it (140 us).  Here a
// workgroup owns a tile of TR x 32 output pixels, loads the (TR*2+5) x 69 x 3 input patch it needs into LDS once
// (10.8 KB; each input pixel is read ~1.4 times from HBM instead of being written 12 times and read back; the next tile's patch is requested before the current tile's MFMA loop) and feeds
// the fp32 matrix cores straight from it: for v_mfma_f32_32x32x2_f32 a lane supplies ONE A element (pixel = lane % 32,
// k = 2 s + lane / 32), which is a single ds_read_b32 at (row*2+ky, col*2+kx, c) of the patch -- consecutive lanes are
// 6 floats apart, 32 different banks.  K is ordered (ky, kx, c) with every ky row of 21 padded to 22 (a zero weight), so
// a k pair never straddles two patch rows; the weights sit in LDS as B[154][64] (39 KB, rebuilt per workgroup from the
// packed stem panel [64][160] the GEMM path uses).  4 waves = 4 output rows of 32 pixels x 64 channels, 2 x 77 MFMAs per
// tile and wave, fp32 accumulation in the same k order as the GEMM path.  Epilogue: 128-byte row stores from the
// accumulators and the per-tile column sums / sums of squares BatchNorm's statistics are finalised from (CF_STATS rows).
// Persistent workgroups, 3 per CU.
#include "sacv_common.h"
#include <algorithm>

namespace {

constexpr int TR = 4, TC = 32;            // output tile: TR rows x TC columns = 4 waves x 32 pixels
constexpr int KH = 7, KW = 7, CIN = 3, ST = 2, PAD = 3, NOUT = 64;
constexpr int PR = (TR - 1) * ST + KH;    // 13 patch rows
constexpr int PC = (TC - 1) * ST + KW;    // 69 patch columns
constexpr int KROW = KW * CIN + 1;        // 22: one ky row of k, padded to an even length
constexpr int KTOT = KH * KROW;           // 154
constexpr int PATCH = PR * PC * CIN + 4;  // + the floats a padded k may touch past the last pixel
constexpr int LDS_FLOATS = KTOT * NOUT + PATCH;

struct StemArgs {
  const float* x;
  long long sN, sC, sH, sW;
  const float* wpanel;  // [64][Kpad], k = (ky*7+kx)*3 + c
  float* y;             // [Nb][Ho][Wo][64]
  float* stats;         // [tiles][2][64] or null
  int Nb, H, W, Ho, Wo, Kpad, tiles_x, tiles_y, ntiles;
};

__global__ __launch_bounds__(256, 3) void stem_conv_kernel(const StemArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Bs = smem;                 // [KTOT][64]
  float* patch = smem + KTOT * NOUT;  // [PR][PC][3]
  __shared__ float red[2][4][NOUT];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  // weights -> LDS, once per workgroup
  // (coalesced reads along k; the transposing LDS writes are bank-conflicted, 38 per thread, once per workgroup)
  // (ten requests in flight per thread: one element at a time was 37 exposed L2 latencies at the start of every workgroup)
  for (int i0 = tid; i0 < NOUT * KH * KW * CIN; i0 += 256 * 10) {
    float wv_[10];
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      const int i = i0 + u * 256;
      const int n = i / (KH * KW * CIN), k = i - n * (KH * KW * CIN);
      wv_[u] = i < NOUT * KH * KW * CIN ? a.wpanel[(size_t)n * a.Kpad + k] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      const int i = i0 + u * 256;
      const int n = i / (KH * KW * CIN), k = i - n * (KH * KW * CIN);
      const int ky = k / (KW * CIN), r = k - ky * (KW * CIN);
      if (i < NOUT * KH * KW * CIN) Bs[(ky * KROW + r) * NOUT + n] = wv_[u];
    }
  }
  for (int i = tid; i < KH * NOUT; i += 256) Bs[((i / NOUT) * KROW + KW * CIN) * NOUT + (i % NOUT)] = 0.f;
  // patch elements of one tile: (c, row, col) with col fastest = the contiguous direction of an NCHW image (any strides
  // work); NLD loads per thread, all in flight at once, and the NEXT tile's are issued before the current tile's MFMA
  // loop (the first version loaded and stored one element at a time: eleven exposed global latencies per tile, 136 us)
  constexpr int NLD = (CIN * PR * PC + 255) / 256;
  int pofs[NLD];  // LDS float index of this thread's elements; -1: none
  int ppy[NLD], ppx[NLD], pc[NLD];
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    const int i = tid + u * 256;
    const int c = i / (PR * PC), rem = i - c * (PR * PC);
    ppy[u] = rem / PC;
    ppx[u] = rem - ppy[u] * PC;
    pc[u] = c;
    pofs[u] = i < CIN * PR * PC ? (ppy[u] * PC + ppx[u]) * CIN + c : -1;
  }
  float pv[NLD];
  auto load_patch = [&](int t) {
    const int per_img = a.tiles_x * a.tiles_y;
    const int n = t / per_img, tt = t - n * per_img;
    const int ty = tt / a.tiles_x, tx = tt - ty * a.tiles_x;
    const int iy0 = ty * TR * ST - PAD, ix0 = tx * TC * ST - PAD;
    const float* xn = a.x + (long long)n * a.sN;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int iy = iy0 + ppy[u], ix = ix0 + ppx[u];
      pv[u] = 0.f;
      if (pofs[u] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
        pv[u] = xn[pc[u] * a.sC + iy * a.sH + ix * a.sW];
    }
  };
  if ((int)blockIdx.x < a.ntiles) load_patch(blockIdx.x);
  for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
    const int per_img = a.tiles_x * a.tiles_y;
    const int n = t / per_img, tt = t - n * per_img;
    const int ty = tt / a.tiles_x, tx = tt - ty * a.tiles_x;
    const int oy0 = ty * TR, ox0 = tx * TC;
    __syncthreads();  // the previous tile's MFMA loop has left the patch (and the weights are in place)
#pragma unroll
    for (int u = 0; u < NLD; ++u)
      if (pofs[u] >= 0) patch[pofs[u]] = pv[u];
    if (tid < 4) patch[PR * PC * CIN + tid] = 0.f;
    __syncthreads();
    if (t + (int)gridDim.x < a.ntiles) load_patch(t + gridDim.x);
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // wave wv = output row oy0 + wv; lane's pixel = column l31
    const float* prow = patch + ((wv * ST) * PC + l31 * ST) * CIN + lh;
    const float* brow = Bs + lh * NOUT + l31;
#pragma unroll
    for (int ky = 0; ky < KH; ++ky) {
      const float* pa = prow + ky * PC * CIN;
      const float* pb = brow + ky * KROW * NOUT;
#pragma unroll
      for (int s = 0; s < KROW / 2; ++s) {
        const float av = pa[2 * s];
        const float b0 = pb[2 * s * NOUT], b1 = pb[2 * s * NOUT + 32];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[1], 0, 0, 0);
      }
    }
    // epilogue: accumulator register r of block j is pixel column (r & 3) + 8 (r >> 2) + 4 lh, channel l31 + 32 j
    const int oy = oy0 + wv;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    if (oy < a.Ho) {
      float* yrow = a.y + (((size_t)n * a.Ho + oy) * a.Wo + ox0) * NOUT;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int col = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (ox0 + col < a.Wo) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float v = acc[j][r];
            yrow[(size_t)col * NOUT + l31 + 32 * j] = v;
            s1[j] += v;
            s2[j] += v * v;
          }
        }
      }
    }
    if (a.stats) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        s1[j] += __shfl_xor(s1[j], 32, 64);
        s2[j] += __shfl_xor(s2[j], 32, 64);
      }
      if (lh == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          red[0][wv][l31 + 32 * j] = s1[j];
          red[1][wv][l31 + 32 * j] = s2[j];
        }
      }
      __syncthreads();
      if (tid < 2 * NOUT) {
        const int which = tid / NOUT, c = tid - which * NOUT;
        const float s = ((red[which][0][c] + red[which][1][c]) + red[which][2][c]) + red[which][3][c];
        a.stats[((size_t)t * 2 + which) * NOUT + c] = s;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same convolution, again without the im2col matrix (round 4): dW[co][k] = sum over output pixels
// of dy[pix][co] * patch(pix, k).  A workgroup of 8 waves owns the tile of the forward kernel (4 rows x 32 output pixels):
// its dy rows (128 x 64 floats, 32 KB) and the 13 x 69 x 3 image patch go to LDS once; wave w contracts the 32 pixels of
// output row w & 3 for the channel half w >> 2 against all 160 (154 padded) k: v_mfma_f32_32x32x2_f32 with A = dy
// (lane: channel l31, pixel 2 s + lh: consecutive floats of a dy row), B = patch (lane: k index l31 of the block, pixel
// 2 s + lh: the forward kernel's (ky, kx, c) offsets), five 32 x 32 accumulator blocks per wave that live across ALL the
// tiles of the (persistent) workgroup.  At the end the four row waves of a channel half are summed in wave order through
// LDS and the workgroup writes ONE slab [64][160] (k in the GEMM path's tap-major order); the slabs are summed in
// workgroup order by the stem reduction of conv_gemm.hip (deterministic).  10.9 GFLOP on the matrix cores (70 us at
// peak) and 162 MB of HBM reads instead of the 478 MB the [pixels][160] matrix and dy cost the conv_tn path (182 us).
constexpr int WG_THREADS = 512;
constexpr int KBLK = 5;  // 160 / 32 column blocks of k

struct StemWgradArgs {
  const float* x;
  long long sN, sC, sH, sW;
  const float* dy;  // [Nb][Ho][Wo][64]
  float* part;      // [grid][64][160]
  int Nb, H, W, Ho, Wo, tiles_x, tiles_y, ntiles;
};

__global__ __launch_bounds__(WG_THREADS, 2) void stem_wgrad_kernel(const StemWgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dyt = smem;                        // [TR * TC][64]; reused as the wave-partial exchange at the end
  float* patch = smem + TR * TC * NOUT;     // [PR][PC][3] (+ 4)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int row = wv & 3, jh = wv >> 2;
  // patch offset of this lane's k index in each of the five column blocks (padded layout: 22 per ky row)
  int koff[KBLK];
#pragma unroll
  for (int nb = 0; nb < KBLK; ++nb) {
    const int n = nb * 32 + l31;
    const int ky = n / KROW, r = n - ky * KROW;
    koff[nb] = (ky < KH && r < KW * CIN) ? ky * PC * CIN + r : 0;  // (columns of invalid k are never written out)
  }
  f32x16 acc[KBLK];
#pragma unroll
  for (int nb = 0; nb < KBLK; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  constexpr int NLD = (CIN * PR * PC + WG_THREADS - 1) / WG_THREADS;  // patch elements per thread
  constexpr int NDY = TR * TC * NOUT / 4 / WG_THREADS;                // 16-byte dy chunks per thread
  float pv[NLD];
  f32x4 dv[NDY];
  auto tile_origin = [&](int t, int& n, int& oy0, int& ox0) {
    const int per_img = a.tiles_x * a.tiles_y;
    n = t / per_img;
    const int tt = t - n * per_img;
    const int ty = tt / a.tiles_x, tx = tt - ty * a.tiles_x;
    oy0 = ty * TR;
    ox0 = tx * TC;
  };
  auto load_tile = [&](int t) {
    int n, oy0, ox0;
    tile_origin(t, n, oy0, ox0);
    const int iy0 = oy0 * ST - PAD, ix0 = ox0 * ST - PAD;
    const float* xn = a.x + (long long)n * a.sN;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int i = tid + u * WG_THREADS;
      const int c = i / (PR * PC), rem = i - c * (PR * PC);
      const int py = rem / PC, px = rem - py * PC;
      const int iy = iy0 + py, ix = ix0 + px;
      pv[u] = 0.f;
      if (i < CIN * PR * PC && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
        pv[u] = xn[c * a.sC + iy * a.sH + ix * a.sW];
    }
#pragma unroll
    for (int u = 0; u < NDY; ++u) {
      const int ch = tid + u * WG_THREADS;  // chunk: pixel ch / 16, channels 4 (ch % 16) ..
      const int pix = ch >> 4, r = pix / TC, c = pix - r * TC;
      dv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (oy0 + r < a.Ho && ox0 + c < a.Wo)
        dv[u] = *reinterpret_cast<const f32x4*>(a.dy + (((size_t)n * a.Ho + oy0 + r) * a.Wo + ox0 + c) * NOUT +
                                                (ch & 15) * 4);
    }
  };
  if ((int)blockIdx.x < a.ntiles) load_tile(blockIdx.x);
  for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
    __syncthreads();  // the previous tile's MFMA loop has left the LDS
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int i = tid + u * WG_THREADS;
      const int c = i / (PR * PC), rem = i - c * (PR * PC);
      if (i < CIN * PR * PC) patch[rem * CIN + c] = pv[u];
    }
    if (tid < 4) patch[PR * PC * CIN + tid] = 0.f;
#pragma unroll
    for (int u = 0; u < NDY; ++u) *reinterpret_cast<f32x4*>(dyt + (size_t)(tid + u * WG_THREADS) * 4) = dv[u];
    __syncthreads();
    if (t + (int)gridDim.x < a.ntiles) load_tile(t + gridDim.x);
    const float* pa = dyt + (row * TC + lh) * NOUT + l31 + 32 * jh;
    const float* pb = patch + ((row * ST) * PC + lh * ST) * CIN;
#pragma unroll
    for (int s = 0; s < TC / 2; ++s) {
      const float av = pa[2 * s * NOUT];
#pragma unroll
      for (int nb = 0; nb < KBLK; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, pb[2 * s * ST * CIN + koff[nb]], acc[nb], 0, 0, 0);
    }
  }
  // the four row waves of each channel half, summed in wave order; one column block per round through LDS
  float* slab = a.part + (size_t)blockIdx.x * NOUT * 160;
  float* ex = dyt;  // [8 waves][32 co][32 k]
  for (int nb = 0; nb < KBLK; ++nb) {
    __syncthreads();
#pragma unroll
    for (int nb2 = 0; nb2 < KBLK; ++nb2)
      if (nb2 == nb) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ex[(wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + l31] = acc[nb2][r];
      }
    __syncthreads();
    for (int e = tid; e < 2 * 32 * 32; e += WG_THREADS) {
      const int j = e >> 10, co = (e >> 5) & 31, kk = e & 31;
      const float v = ((ex[((j * 4 + 0) * 32 + co) * 32 + kk] + ex[((j * 4 + 1) * 32 + co) * 32 + kk]) +
                       ex[((j * 4 + 2) * 32 + co) * 32 + kk]) + ex[((j * 4 + 3) * 32 + co) * 32 + kk];
      const int n = nb * 32 + kk;
      const int ky = n / KROW, r = n - ky * KROW;
      if (ky < KH && r < KW * CIN) slab[(size_t)(j * 32 + co) * 160 + ky * KW * CIN + r] = v;
    }
  }
}

}  // namespace

SACV_API_BEGIN

// rows of the statistics buffer of sacv_stem_conv_fwd (one per output tile)
int sacv_stem_conv_stats_rows(int Nb, int Ho, int Wo) { return Nb * ((Ho + TR - 1) / TR) * ((Wo + TC - 1) / TC); }

// y[Nb][Ho][Wo][64] = conv7x7/s2/p3(x) for a 3-channel image given by element strides (NCHW-contiguous or
// channels_last), weights as the packed stem panel [64][Kpad] (sacv_pack_weight, stem mode: k = (ky*7+kx)*3 + c).
// stats (optional): [sacv_stem_conv_stats_rows][2][64] per-tile column sums and sums of squares of y.
int sacv_stem_conv_fwd(const float* x, int Nb, int H, int W, long long sN, long long sC, long long sH, long long sW,
                       const float* wpanel, int Kpad, float* y, int Ho, int Wo, float* stats, void* stream) {
  SACV_CHECK_ARG(x && wpanel && y, "stem_conv_fwd: null pointer");
  SACV_CHECK_ARG(Nb > 0 && H > 0 && W > 0, "stem_conv_fwd: empty image");
  SACV_CHECK_ARG(Ho == (H + 2 * PAD - KH) / ST + 1 && Wo == (W + 2 * PAD - KW) / ST + 1,
                 "stem_conv_fwd: output %d x %d does not belong to a %d x %d image (7x7, stride 2, pad 3)", Ho, Wo, H, W);
  SACV_CHECK_ARG(Kpad >= KH * KW * CIN, "stem_conv_fwd: Kpad %d < 147", Kpad);
  SACV_CHECK_ARG((long long)Nb * Ho * Wo * NOUT < (1LL << 40), "stem_conv_fwd: output too large");
  StemArgs a;
  a.x = x;
  a.sN = sN;
  a.sC = sC;
  a.sH = sH;
  a.sW = sW;
  a.wpanel = wpanel;
  a.y = y;
  a.stats = stats;
  a.Nb = Nb;
  a.H = H;
  a.W = W;
  a.Ho = Ho;
  a.Wo = Wo;
  a.Kpad = Kpad;
  a.tiles_y = (Ho + TR - 1) / TR;
  a.tiles_x = (Wo + TC - 1) / TC;
  const long long nt = (long long)Nb * a.tiles_x * a.tiles_y;
  SACV_CHECK_ARG(nt < (1LL << 30), "stem_conv_fwd: too many tiles");
  a.ntiles = (int)nt;
  const int lds = LDS_FLOATS * 4;
  const int grid = std::min(a.ntiles, 3 * 256);
  hipLaunchKernelGGL(stem_conv_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
  SACV_CHECK_LAUNCH("stem_conv_kernel");
  return SACV_OK;
}

// Persistent grid of sacv_stem_conv_wgrad and its workspace: one slab [64][160] per workgroup.
// (one 8-wave workgroup per CU: 177 VGPRs -- the five accumulator blocks, the next tile's loads in flight and the unrolled
// operand reads; capped at 128 for two workgroups per CU the kernel spills 51 registers)
static int stem_wgrad_grid(long long ntiles) { return (int)std::min<long long>(ntiles, 256); }
long long sacv_stem_conv_wgrad_workspace_bytes(int Nb, int Ho, int Wo) {
  const long long nt = (long long)Nb * ((Ho + TR - 1) / TR) * ((Wo + TC - 1) / TC);
  return (long long)stem_wgrad_grid(nt) * NOUT * 160 * (long long)sizeof(float);
}

// dw[64][3][7][7] (OIHW) (+)= weight gradient of the 7x7/s2/p3 stem convolution from the image x (element strides, as
// sacv_stem_conv_fwd) and dy [Nb][Ho][Wo][64]: replaces sacv_im2col_stem + sacv_conv2d_wgrad on the [pixels][160] matrix
// (the backward of nn.Conv2d(3, 64, 7, 2, 3), classification/backbones/resnet.py:176-181).
int sacv_stem_conv_wgrad(const float* x, int Nb, int H, int W, long long sN, long long sC, long long sH, long long sW,
                         const float* dy, int Ho, int Wo, float* dw, float* workspace, int accumulate, void* stream) {
  SACV_CHECK_ARG(x && dy && dw && workspace, "stem_conv_wgrad: null pointer");
  SACV_CHECK_ARG(Nb > 0 && H > 0 && W > 0, "stem_conv_wgrad: empty image");
  SACV_CHECK_ARG(Ho == (H + 2 * PAD - KH) / ST + 1 && Wo == (W + 2 * PAD - KW) / ST + 1,
                 "stem_conv_wgrad: output %d x %d does not belong to a %d x %d image (7x7, stride 2, pad 3)", Ho, Wo, H, W);
  StemWgradArgs a;
  a.x = x;
  a.sN = sN;
  a.sC = sC;
  a.sH = sH;
  a.sW = sW;
  a.dy = dy;
  a.part = workspace;
  a.Nb = Nb;
  a.H = H;
  a.W = W;
  a.Ho = Ho;
  a.Wo = Wo;
  a.tiles_y = (Ho + TR - 1) / TR;
  a.tiles_x = (Wo + TC - 1) / TC;
  const long long nt = (long long)Nb * a.tiles_x * a.tiles_y;
  SACV_CHECK_ARG(nt < (1LL << 30), "stem_conv_wgrad: too many tiles");
  a.ntiles = (int)nt;
  const int grid = stem_wgrad_grid(nt);
  const int lds = (TR * TC * NOUT + PATCH) * 4;
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3(grid), dim3(WG_THREADS), lds, (hipStream_t)stream, a);
  SACV_CHECK_LAUNCH("stem_wgrad_kernel");
  // slabs [grid][64][160], k = tap * 3 + c  ->  dw[co][c][tap]
  return sacv_internal_wgrad_reduce(workspace, dw, grid, NOUT, CIN, 160, 1, 1, accumulate, KH * KW, 0, stream);
}

SACV_API_END
