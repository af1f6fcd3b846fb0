// Halo-resident bf16 3x3 / stride 1 / pad 1 convolution for gfx950 (MI355X): forward and data gradient of the
// 64-output-channel 3x3 layers of the ResNet50 classifier under autocast (layer1: 64 -> 64 at 56 x 56, batch 256;
// reference call site: nn.Conv2d(planes, planes, 3, padding=1) of Bottleneck.conv2,
// simpleAICV/classification/backbones/resnet.py:100-155, run under torch.autocast, tools/scripts.py:108-267).
//
// Why another kernel.  conv_nt_bf16_kernel gathers the A operand per filter tap: every input row is fetched nine times
// through L2 (925 MB of taps for 206 MB of tensors on layer1), 48 vector-memory instructions per 128 MFMAs, and the
// launch sits at 0.20 of the HBM roofline although its arithmetic intensity (288 flop/B) is below the bf16 ridge.  Here
// the input of a tile is brought into LDS ONCE and all nine taps read it there:
//   * "Flat halo": outputs are indexed by their position q in the zero-PADDED image, linearised row-major over
//     (H + 2) x (W + 2) and over the batch.  Tap (kh, kw) of output q then reads padded input q + kh (W + 2) + kw: a
//     constant shift.  A tile is P = 256 consecutive positions; its input is ONE contiguous run of P + 2 (W + 2) + 2
//     positions (at most 384 rows of 128 bytes per 64-channel chunk), loaded by LDS-DMA (global_load_lds_dwordx4, a
//     per-lane source address: the real pixel, or a page of zeros for padding / beyond the batch).  Positions in the
//     padding ring (6.8 % at 56 x 56) are computed and not stored.
//   * The weight panel [N][9 Cin] streams through a 3-stage ring of per-tap tiles (64 rows x 128 bytes), two taps in
//     flight, counted vmcnt + one barrier per tap (as wino_gemm_kernel / gemm_nt_bf16_dma_kernel).
//   * LDS rows are unpadded and XOR-swizzled on the source side with key (row >> 1) & 7 (conflict-free for the 16-lane
//     groups of ds_read_b128); a fragment address is (row + tap shift) with the key of THAT row.
//   * 256 threads = 4 waves of 64 positions x 64 channels (2 x 2 blocks of v_mfma_f32_32x32x16_bf16, one LDS read per
//     MFMA), 72 KB of LDS, two workgroups per CU (one computes while the other loads / stores), persistent tiles in
//     XCD-contiguous order.
//   * Epilogue as conv_nt_bf16_kernel: fp32 stage in LDS, bf16 rounding, 16-byte stores of the valid positions,
//     per-tile column sums of the rounded values (CF_STATS) for BatchNorm.  For Cin = 64 the summation order per
//     output (taps outer, channels inner, 16 per MFMA) equals conv_nt_bf16_kernel's: results are bit-identical.
// Measured (scripts/bench_halo.py, layer1 shape, random data): 98-105 us against 138-145 us for conv_nt_bf16_kernel
// (565-605 vs 407-428 TFLOP/s; 0.25-0.26 vs 0.18 of the HBM roofline); in the classifier step 127 -> 98 us per launch.
// What bounds it (ablation builds, layer1 shape, 98 us): without the epilogue 59 us (1 005 TFLOP/s), without the image
// loads 88 us, without the MFMAs 78 us -- the epilogue is 40 % of the launch.  Two other epilogues were built and
// measured against this one: (a) the TRANSPOSED product (weights as the first MFMA operand: a lane then holds four
// consecutive channels of one position) stored straight from the accumulators, the next tile's image requested before
// the epilogue and the BatchNorm sums kept in registers across tiles: 111-115 us -- a store instruction then writes 16-
// byte pieces of 32 different lines; (b) the same with a per-wave 4 KB LDS transposition (16 ds_write_b64 + 8
// ds_read_b128 per lane, no workgroup barrier, whole 128-byte rows per store): 92-100 us in the microbenchmark, equal
// to this version in the classifier step (25.72 vs 25.67 ms).  What the three share is the wait for the stores: on
// gfx950 loads and stores retire through the ONE vmcnt counter and may pass each other, so the counted waits of the
// next tile's tap ring cannot skip the older stores and every tile ends up waiting for its HBM write acknowledgements.
// The form that avoids it keeps the weight panel resident in LDS (no counted waits inside a tile at all) at one
// workgroup per CU: not built.  (Also measured: per-row position decoding by division cost 7 us -> incremental; a
// start-up stagger of the second workgroup of each CU by half a tile: no effect; the nine taps inline: -5 us.)
#include "sacv_common.h"
#include <mutex>

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define HCF_STATS 32

namespace {

typedef __attribute__((address_space(3))) void* glds_t;
__device__ __forceinline__ void glds16b(const void* src, glds_t dst) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
#else
  (void)src;
  (void)dst;
#endif
}

__device__ __attribute__((aligned(128))) unsigned int halo_zero_page[32];  // 128 zero bytes: one whole LDS row

constexpr int P = 256;                 // positions per tile
constexpr int ROWS = 384;              // LDS rows of the input run (>= P + 2 (W + 2) + 2, multiple of 32)
constexpr int BN = 64;                 // output channels
constexpr int A_BYTES = ROWS * 128;    // 49 152
constexpr int B_STAGE = BN * 128;      // 8 192
constexpr int LDS_BYTES = A_BYTES + 3 * B_STAGE;  // 73 728
constexpr int CS = BN + 4;             // fp32 stage row (floats)
static_assert(P * CS * 4 <= LDS_BYTES, "the epilogue stage must fit in the operand image");

struct HaloArgs {
  const bf16* src;   // [Nb][H][W][Cs]
  bf16* dst;         // [Nb][H][W][64]
  const bf16* w;     // panel [64][9 * Cs], k = tap * Cs + c
  float* stats;      // [ntiles][2][64] (HCF_STATS)
  int Nb, H, W, Cs;
  int Wp, Lp;        // W + 2, (H + 2) * (W + 2)
  int Q;             // Nb * Lp positions
  int ntiles;
  int shift[9];      // LDS row shift of panel tap t (forward: kh Wp + kw; data gradient: (2 - kh) Wp + (2 - kw))
  int flags;
};

__global__ __launch_bounds__(256, 2) void conv3x3_halo_bf16_kernel(const HaloArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + A_BYTES;
  const int nchunks = a.Cs >> 6;
  const int K = 9 * a.Cs;
  const unsigned char* srcb = reinterpret_cast<const unsigned char*>(a.src);
  const unsigned char* wb = reinterpret_cast<const unsigned char*>(a.w);
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(halo_zero_page);

  const int per = (a.ntiles + 7) >> 3;
  for (int it = blockIdx.x; it < per * 8; it += gridDim.x) {
    const int tile = (it & 7) * per + (it >> 3);  // XCD-contiguous order (workgroup b runs on XCD b % 8)
    if (tile >= a.ntiles) continue;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));  // keep the thread-index arithmetic inside the tile loop
    const int lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int q0 = tile * P;

    __syncthreads();  // the previous tile's epilogue has left the LDS

    // ---- source byte offsets of this lane's 12 input rows (rows (wid * 12 + i) * 8 + lane / 8 of the run)
    // a row is a padded position q = q0 + r: image q / Lp, padded coordinates (A, B) = ((q % Lp) / Wp, (q % Lp) % Wp),
    // real pixel (A - 1, B - 1) when 1 <= A <= H and 1 <= B <= W, otherwise zeros
    // (one division pair per tile and lane; the next row of this lane is 8 positions further: incremental, Wp > 8.
    //  The first version decoded every row with two 32-bit divisions -- 24 per lane and tile plus 16 in the epilogue,
    //  ~5 000 VALU cycles per wave against the tile's 4 608 MFMA cycles: 105 us per layer1 convolution)
    unsigned aoff[12];  // byte offsets (tensors are checked to be < 2^31 elements); ~0u = zeros
    {
      const unsigned q = (unsigned)(q0 + wid * 96 + (lane >> 3));
      unsigned n = q / (unsigned)a.Lp;
      const unsigned rem = q - n * (unsigned)a.Lp;
      unsigned A = rem / (unsigned)a.Wp;
      unsigned B = rem - A * (unsigned)a.Wp;
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const bool ok = n < (unsigned)a.Nb && A >= 1u && A <= (unsigned)a.H && B >= 1u && B <= (unsigned)a.W;
        const unsigned pix = (n * (unsigned)a.H + (A - 1u)) * (unsigned)a.W + (B - 1u);
        // chunk position lane & 7 of LDS row r holds source chunk (lane & 7) ^ key(r); r = wid * 96 + i * 8 + lane / 8,
        // key(r) = (r >> 1) & 7 = (i & 1) * 4 + lane / 16
        const unsigned key = (unsigned)((i & 1) * 4 + (lane >> 4));
        aoff[i] = ok ? pix * (unsigned)a.Cs * 2u + ((((unsigned)lane & 7u) ^ key) << 4) : ~0u;
        B += 8u;
        if (B >= (unsigned)a.Wp) {
          B -= (unsigned)a.Wp;
          A += 1u;
          if (A >= (unsigned)a.H + 2u) {
            A = 0u;
            n += 1u;
          }
        }
      }
    }
    // weight rows: 64 rows, 8 per DMA instruction, 2 instructions per wave and tap
    unsigned boff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n = (wid * 2 + i) * 8 + (lane >> 3);
      boff[i] = (unsigned)n * (unsigned)K * 2u + (unsigned)(((lane & 7) ^ ((n >> 1) & 7)) << 4);
    }
    auto issue_a = [&](int c0) {
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const unsigned char* s = aoff[i] != ~0u ? srcb + (size_t)aoff[i] + (size_t)c0 * 2 : zero + ((lane & 7) << 4);
        glds16b(s, (glds_t)(sA + (wid * 12 + i) * 1024));
      }
    };
    auto issue_b = [&](int tap, int c0, int st) {
      const unsigned char* pb = wb + ((size_t)tap * a.Cs + c0) * 2;
#pragma unroll
      for (int i = 0; i < 2; ++i) glds16b(pb + boff[i], (glds_t)(sB + st * B_STAGE + (wid * 2 + i) * 1024));
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // this lane's fragment rows: A rows wid * 64 + i * 32 + l31 (+ the tap's shift), B rows j * 32 + l31
    const int arow0 = wid * 64 + l31;
    unsigned boffs[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = j * 32 + l31;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) boffs[j][kk] = (unsigned)(n * 128 + (((kk * 2 + lh) ^ ((n >> 1) & 7)) << 4));
    }

    for (int ch = 0; ch < nchunks; ++ch) {
      const int c0 = ch << 6;
      if (ch) __builtin_amdgcn_s_barrier();  // every wave has read the previous chunk's image and weight stages
      issue_a(c0);
      issue_b(0, c0, 0);
      issue_b(1, c0, 1);
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // the image and tap 0 have landed for this wave; tap 1 in flight
      __builtin_amdgcn_s_barrier();
#pragma unroll  // (9 x 16 MFMAs inline: the tap shifts and ring stages become constants; 105 -> 98 us on layer1)
      for (int tap = 0; tap < 9; ++tap) {
        const int st = tap % 3;
        const int sh = a.shift[tap];
        const unsigned char* cB = sB + st * B_STAGE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          bf16x8 fa[2], fb[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int row = arow0 + i * 32 + sh;
            fa[i] = *reinterpret_cast<const bf16x8*>(sA + row * 128 + (((kk * 2 + lh) ^ ((row >> 1) & 7)) << 4));
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(cB + boffs[j][kk]);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
          // tap + 2 goes out behind the first quarter's MFMAs (its stage was last read in iteration tap - 1)
          if (kk == 0 && tap + 2 < 9) issue_b(tap + 2, c0, (tap + 2) % 3);
        }
        if (tap + 2 < 9)
          asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // tap + 1 has landed, tap + 2 stays in flight
        else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tap < 8) __builtin_amdgcn_s_barrier();
      }
    }

    // ---- epilogue: accumulators -> fp32 stage [P][64 + 4] -> bf16 rows of the valid positions
    __syncthreads();  // all MFMAs' operand reads are done: the image becomes the stage
    float* stage = reinterpret_cast<float*>(smem);
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    {
      const int e_lane = tid_e & 63, e_wid = tid_e >> 6, e_l31 = e_lane & 31, e_lh = e_lane >> 5;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            stage[(e_wid * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * e_lh) * CS + j * 32 + e_l31] = acc[i][j][r];
    }
    __syncthreads();
    const int c8 = tid_e & 7, rr = tid_e >> 3;  // 8 column groups of 8 channels, 32 row lanes
    float ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) ssum[e] = ssq[e] = 0.f;
    {
      const unsigned q = (unsigned)(q0 + rr);
      unsigned n = q / (unsigned)a.Lp;
      const unsigned rem = q - n * (unsigned)a.Lp;
      unsigned A = rem / (unsigned)a.Wp;
      unsigned B = rem - A * (unsigned)a.Wp;
#pragma unroll
      for (int i = 0; i < P / 32; ++i) {
        const int row = rr + i * 32;
        if (n < (unsigned)a.Nb && A < (unsigned)a.H && B < (unsigned)a.W) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(stage + row * CS + c8 * 8);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(stage + row * CS + c8 * 8 + 4);
          float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            o[e] = (bf16)v[e];
            v[e] = (float)o[e];
            ssum[e] += v[e];
            ssq[e] += v[e] * v[e];
          }
          const size_t pix = ((size_t)n * a.H + A) * a.W + B;
          *reinterpret_cast<bf16x8*>(a.dst + pix * BN + c8 * 8) = o;
        }
        // the thread's next row is 32 positions further (Wp >= 9: at most four row wraps)
        B += 32u;
        while (B >= (unsigned)a.Wp) {
          B -= (unsigned)a.Wp;
          A += 1u;
        }
        while (A >= (unsigned)a.H + 2u) {
          A -= (unsigned)a.H + 2u;
          n += 1u;
        }
      }
    }
    if (a.flags & HCF_STATS) {
      __syncthreads();  // the stage is dead
      float* red = reinterpret_cast<float*>(smem);  // [2][32][64]
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(size_t)rr * BN + c8 * 8 + e] = ssum[e];
        red[(size_t)(32 + rr) * BN + c8 * 8 + e] = ssq[e];
      }
      __syncthreads();
      if (tid_e < 2 * BN) {
        const int which = tid_e / BN, col = tid_e - which * BN;
        float s = 0.f;
        for (int l = 0; l < 32; ++l) s += red[(size_t)(which * 32 + l) * BN + col];
        a.stats[((size_t)tile * 2 + which) * BN + col] = s;
      }
    }
  }
}

int halo_num_cus() {
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

}  // namespace

// 1 when the halo-resident kernel takes this 3x3 / stride 1 / pad 1 problem (sacv_conv2d_gemm_bf16 asks first).
// N == 64 (the weight tap tiles and the wave layout are built for it), Cs a multiple of 64, the input run of a tile
// must fit the 384-row image, enough tiles to fill the chip, 32-bit positions.
int sacv_internal_conv3x3_halo_bf16_supported(int Nb, int H, int W, int Cs, int N) {
#ifdef SACV_NO_HALO_CONV
  return 0;
#endif
  if (N != BN || Cs % 64 != 0 || Cs <= 0) return 0;
  if (P + 2 * (W + 2) + 2 > ROWS || W + 2 <= 8) return 0;
  const long long Q = (long long)Nb * (H + 2) * (W + 2);
  if (Q >= (1LL << 30) || (long long)Nb * H * W * (Cs > N ? Cs : N) >= (1LL << 31)) return 0;  // 32-bit byte offsets
  return (Q + P - 1) / P >= 2LL * halo_num_cus();
}

// rows of the statistics buffer this kernel fills ([rows][2][64]); the caller zeroes what lies beyond
int sacv_internal_conv3x3_halo_bf16_stats_rows(int Nb, int H, int W) {
  return (int)(((long long)Nb * (H + 2) * (W + 2) + P - 1) / P);
}

int sacv_internal_conv3x3_halo_bf16(const void* src, void* dst, int Nb, int H, int W, const void* w, int Cs, int N,
                                    int mode, int flags, float* stats, void* stream) {
  SACV_CHECK_ARG(src && dst && w, "conv3x3_halo_bf16: null pointer");
  SACV_CHECK_ARG(sacv_internal_conv3x3_halo_bf16_supported(Nb, H, W, Cs, N), "conv3x3_halo_bf16: unsupported shape");
  SACV_CHECK_ARG(!(flags & HCF_STATS) || stats, "conv3x3_halo_bf16: CF_STATS without a stats buffer");
  SACV_CHECK_ARG((flags & ~HCF_STATS) == 0, "conv3x3_halo_bf16: unsupported epilogue flags 0x%x", flags);
  HaloArgs a;
  a.src = (const bf16*)src;
  a.dst = (bf16*)dst;
  a.w = (const bf16*)w;
  a.stats = stats;
  a.Nb = Nb;
  a.H = H;
  a.W = W;
  a.Cs = Cs;
  a.Wp = W + 2;
  a.Lp = (H + 2) * (W + 2);
  a.Q = Nb * a.Lp;
  a.ntiles = (a.Q + P - 1) / P;
  for (int kh = 0; kh < 3; ++kh)
    for (int kw = 0; kw < 3; ++kw)
      a.shift[kh * 3 + kw] = mode == 0 ? kh * a.Wp + kw : (2 - kh) * a.Wp + (2 - kw);
  a.flags = flags;
  static std::once_flag attr;
  std::call_once(attr, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_bf16_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  });
  int grid = 2 * halo_num_cus();
  if (grid > a.ntiles) grid = a.ntiles;
  hipLaunchKernelGGL(conv3x3_halo_bf16_kernel, dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a);
  SACV_CHECK_LAUNCH("conv3x3_halo_bf16_kernel");
  return SACV_OK;
}
