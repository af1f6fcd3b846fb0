// Device-side detection postprocess, four launches per call: (1) score scan at HBM rate -- per prediction the class
// argmax, threshold, one 64-bit candidate key into atomic-free per-workgroup segments; (2) per image the exact top-n by
// radix select, bitonic sort, and the box decode of the survivors only; (3) the suppression bit-mask over the whole
// chip; (4) the greedy walk of hard NMS (or the batched Soft-NMS loop), padded to max_object_num.
// Replaces the host NumPy path of simpleAICV/detection/decode.py:
//   RetinaDecoder.__call__ :201-249, snap_txtytwth_to_x1y1x2y2 :251-271 (int32 truncation),
//   FCOSDecoder :293-364, DecodeMethod.__call__ :121-172, DetNMSMethod python_nms /
//   diou_python_nms :45-102 (class-agnostic, area = max(w*h,0), union clamped at 1e-4); torch_nms :39-43 =
//   torchvision.ops.nms as documented (unclamped IoU, drop when IoU > threshold; parity unpinned: no torchvision).
// Equal scores: the reference's np.argsort(-scores) order is unspecified; here ties are broken
// by ascending anchor index (deterministic).
// Compiled with -ffp-contract=off so IoU arithmetic rounds like the NumPy fp32 ops.
#include "sacv_common.h"
#include <algorithm>
#include <mutex>
#include <math.h>

#define DMAXL 8
#define TOPN_MAX 1024
#define SEL_THREADS 1024
#define SCAN_THREADS 256
#define IDX_BITS 22  // anchors per image in a candidate key (A <= 4 194 304)
#define CLS_BITS 10  // class index in a candidate key (C <= 1024)
typedef unsigned long long u64;

struct DecodeArgs {
  const float* cls[DMAXL];
  const float* reg[DMAXL];
  const float* ctr[DMAXL];  // FCOS only
  int off[DMAXL + 1];
  int tile0[DMAXL + 1];  // first scan tile of a level (the tiles of all B images of a level are consecutive)
  int nlev;
};

// A candidate = one prediction whose score passes the threshold, as ONE 64-bit key
//   [ score bits : 32 | (2^22 - 1 - anchor index) : 22 | class : 10 ]
// scores are > min_score >= 0, so their bit patterns order like the values; equal scores order by ascending anchor
// index (the tie rule of this file's header); the class rides along below the index and never decides an order.
__device__ __forceinline__ u64 cand_key(float score, int a, int klass) {
  return ((u64)__float_as_uint(score) << 32) | ((u64)(unsigned)(((1 << IDX_BITS) - 1) - a) << CLS_BITS) | (u64)(unsigned)klass;
}
__device__ __forceinline__ int key_anchor(u64 k) { return ((1 << IDX_BITS) - 1) - (int)((k >> CLS_BITS) & ((1u << IDX_BITS) - 1)); }
__device__ __forceinline__ int key_class(u64 k) { return (int)(k & ((1u << CLS_BITS) - 1)); }
__device__ __forceinline__ float key_score(u64 k) { return __uint_as_float((unsigned)(k >> 32)); }

// Candidate lists without atomics: image b is scanned by NSEG persistent workgroups; workgroup (b, g) takes the tiles
// g, g + NSEG, ... of the image and appends what passes the threshold to ITS OWN segment cand[b][g][0 .. segcap), keeping
// the running count in a register (one LDS round over the 4 waves per tile) and writing segcnt[b][g] once at the end.
// (The first version appended to one list per image with an atomic per workgroup: 1 568 same-address device-scope
// atomics per image took 24 us of a 48 us kernel.)  The selection kernel reads the segments through a prefix sum of their
// counts.  Deterministic down to the memory image.
struct SegArgs {
  u64* cand;    // [B][nseg][segcap]
  int* segcnt;  // [B][nseg]
  int nseg, segcap, tiles_img;
};

__device__ __forceinline__ int append_local(bool valid, u64 key, u64* __restrict__ seg, int seg_n, int* wave_n) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const u64 m = __ballot(valid);
  if (lane == 0) wave_n[wv] = __popcll(m);
  __syncthreads();
  int pos = seg_n, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; ++w) {
    const int c = wave_n[w];
    if (w < wv) pos += c;
    tot += c;
  }
  if (valid) {
    const u64 below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    seg[pos + __popcll(m & below)] = key;
  }
  return seg_n + tot;
}

struct ScanTile {
  int l, r0, nrows, nl;
};
template <int ROWS>
__device__ __forceinline__ ScanTile scan_tile(const DecodeArgs& d, int t) {  // t: tile number inside an image
  ScanTile s;
  s.l = 0;
  for (int k = 1; k < d.nlev; ++k)
    if (t >= d.tile0[k]) s.l = k;
  s.nl = d.off[s.l + 1] - d.off[s.l];
  s.r0 = (t - d.tile0[s.l]) * ROWS;
  s.nrows = min(ROWS, s.nl - s.r0);
  return s;
}

// Score scan: per prediction the first maximum over the C class scores (np.argmax), FCOS: sqrt(max * centerness),
// threshold, append.  HBM-bound: every class score is read once, as 16-byte loads in the tensor's own order -- a tile of
// ROWS predictions x C classes is contiguous in [B][n_l][C] -- and laid into LDS with a row stride of C + 4 floats, where
// 256 / ROWS threads per prediction reduce their share of the row with ds_read_b128 (stride 84 floats at C = 80: the 16
// lanes a b128 read is served for hit 16 different bank quads).  Persistent workgroups (3 per CU at C = 80), and the
// loads of a workgroup's NEXT tile are issued before the current tile is reduced, so the argmax and the barriers run
// under a full set of requests in flight.  kind 0: RetinaNet, kind 1: FCOS.
template <int ROWS>
__global__ __launch_bounds__(SCAN_THREADS) void decode_scan_kernel(const DecodeArgs d, const SegArgs sg, int kind, int C,
                                                                    float thr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f32x4* tile = reinterpret_cast<f32x4*>(smem_raw);
  __shared__ int wave_n[2][SCAN_THREADS / 64];
  constexpr int TPR = SCAN_THREADS / ROWS;  // threads per prediction
  constexpr int NLD = 16;                   // 16-byte loads per thread and tile, at most (host: ROWS * C / 4 <= 4096)
  const int tid = threadIdx.x;
  const int b = blockIdx.x / sg.nseg, g = blockIdx.x - b * sg.nseg;
  u64* seg = sg.cand + ((size_t)b * sg.nseg + g) * sg.segcap;
  const int C4 = C >> 2, S4 = C4 + 1;
  const int row0 = tid / C4, c40 = tid - row0 * C4;  // (row, chunk) of float4 number tid; + 256 per load, no divisions
  const int dr = SCAN_THREADS / C4, dc = SCAN_THREADS - dr * C4;
  const int r = tid / TPR, sub = tid % TPR;
  const int per = (C4 + TPR - 1) / TPR;
  const int k0 = min(sub * per, C4), k1 = min(C4, k0 + per);
  f32x4 v[NLD];
  int t = g, seg_n = 0, it = 0;
  ScanTile cur = scan_tile<ROWS>(d, t);  // (host: nseg <= tiles_img, so every workgroup has a first tile)
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(d.cls[cur.l] + ((size_t)b * cur.nl + cur.r0) * C);
    const int total = cur.nrows * C4;
#pragma unroll
    for (int u = 0; u < NLD; ++u)
      if (tid + u * SCAN_THREADS < total) v[u] = __builtin_nontemporal_load(src + tid + u * SCAN_THREADS);
  }
  while (true) {
    {
      const int total = cur.nrows * C4;
      int row = row0, c4 = c40;
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        if (tid + u * SCAN_THREADS < total) tile[row * S4 + c4] = v[u];
        row += dr;
        c4 += dc;
        if (c4 >= C4) {
          c4 -= C4;
          ++row;
        }
      }
    }
    __syncthreads();
    const int tn = t + sg.nseg;
    ScanTile nxt = cur;
    if (tn < sg.tiles_img) {  // uniform
      nxt = scan_tile<ROWS>(d, tn);
      const f32x4* src = reinterpret_cast<const f32x4*>(d.cls[nxt.l] + ((size_t)b * nxt.nl + nxt.r0) * C);
      const int total = nxt.nrows * C4;
#pragma unroll
      for (int u = 0; u < NLD; ++u)
        if (tid + u * SCAN_THREADS < total) v[u] = __builtin_nontemporal_load(src + tid + u * SCAN_THREADS);
    }
    float best = -INFINITY;
    int bi = k0 * 4;
    if (r < cur.nrows) {
      for (int k = k0; k < k1; ++k) {
        const f32x4 x = tile[r * S4 + k];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (x[e] > best) {  // strict: the first maximum, like np.argmax
            best = x[e];
            bi = 4 * k + e;
          }
      }
    }
#pragma unroll
    for (int s = 1; s < TPR; s <<= 1) {  // the lower share wins a tie
      const float ov = __shfl_xor(best, s, 64);
      const int oi = __shfl_xor(bi, s, 64);
      const bool take = (sub & s) ? !(best > ov) : (ov > best);
      if (take) {
        best = ov;
        bi = oi;
      }
    }
    const size_t grow = (size_t)b * cur.nl + cur.r0 + r;
    float sc = best;
    const bool owner = sub == 0 && r < cur.nrows;
    if (kind == 1 && owner) sc = sqrtf(best * d.ctr[cur.l][grow]);
    const bool valid = owner && sc > thr;
    // (the barrier inside also fences this tile's LDS reads from the next tile's LDS writes; wave_n alternates)
    seg_n = append_local(valid, cand_key(sc, d.off[cur.l] + cur.r0 + r, bi), seg, seg_n, wave_n[it & 1]);
    ++it;
    if (tn >= sg.tiles_img) break;
    t = tn;
    cur = nxt;
  }
  if (tid == 0) sg.segcnt[(size_t)b * sg.nseg + g] = seg_n;
}

// The same for class counts that are not a multiple of 4 (Objects365: 365) or too wide for the LDS tile: tiles of 256
// predictions, one thread per prediction, scalar reads.
__global__ __launch_bounds__(SCAN_THREADS) void decode_scan_generic_kernel(const DecodeArgs d, const SegArgs sg, int kind,
                                                                            int C, float thr) {
  __shared__ int wave_n[2][SCAN_THREADS / 64];
  const int b = blockIdx.x / sg.nseg, g = blockIdx.x - b * sg.nseg;
  u64* seg = sg.cand + ((size_t)b * sg.nseg + g) * sg.segcap;
  int seg_n = 0, it = 0;
  for (int t = g; t < sg.tiles_img; t += sg.nseg, ++it) {
    const ScanTile cur = scan_tile<SCAN_THREADS>(d, t);
    const int r = threadIdx.x;
    bool valid = false;
    u64 key = 0ull;
    if (r < cur.nrows) {
      const size_t row = (size_t)b * cur.nl + cur.r0 + r;
      const float* c = d.cls[cur.l] + row * C;
      float best = c[0];
      int bi = 0;
      for (int k = 1; k < C; ++k) {
        const float x = c[k];
        if (x > best) {
          best = x;
          bi = k;
        }
      }
      const float sc = kind == 1 ? sqrtf(best * d.ctr[cur.l][row]) : best;
      valid = sc > thr;
      key = cand_key(sc, d.off[cur.l] + cur.r0 + r, bi);
    }
    seg_n = append_local(valid, key, seg, seg_n, wave_n[it & 1]);
  }
  if (threadIdx.x == 0) sg.segcnt[(size_t)b * sg.nseg + g] = seg_n;
}

// ------------------------------------------------------------------------------------------
// Exact top-n of an image's candidates + sort + box decode of the selected ones, one workgroup per image.
// Radix select on the 64-bit keys from the top (12 + 12 + 8 bits of the score, then the index bits), each level a
// histogram in LDS and a parallel suffix scan over the bins; the walk stops at the first level whose boundary bin is
// taken whole (with distinct scores: after two or three levels).  Everything >= the threshold key is then collected
// (exactly K keys, any order), sorted by a bitonic network in LDS, and only these K predictions get their boxes decoded
// (snap_txtytwth_to_x1y1x2y2, decode.py:251-271 / FCOS :333-345, int32 truncation).
__global__ __launch_bounds__(SEL_THREADS) void select_sort_kernel(const SegArgs sg, const DecodeArgs d,
                                                                   const float* __restrict__ geom, int kind,
                                                                   int topn, float* __restrict__ s_scores,
                                                                   float* __restrict__ s_classes,
                                                                   float* __restrict__ s_boxes,
                                                                   int* __restrict__ s_count) {
  __shared__ int hist[4096];
  __shared__ u64 item[TOPN_MAX];
  __shared__ int seg_start[SEL_THREADS + 1];  // exclusive prefix of the segment counts (nseg <= 1024), then n
  __shared__ int wsum[SEL_THREADS / 64];
  __shared__ int sh_bin, sh_rem, sh_done, sh_n;
  constexpr int KREG = 8;  // keys a thread keeps in registers: the first 8192 candidates are read from memory once
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const u64* segs = sg.cand + (size_t)b * sg.nseg * sg.segcap;
  // n and the segment starts: block-wide exclusive scan of segcnt[b][:]
  int n;
  {
    const int c = tid < sg.nseg ? sg.segcnt[(size_t)b * sg.nseg + tid] : 0;
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int x = __shfl_up(incl, o, 64);
      if (lane >= o) incl += x;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SEL_THREADS / 64; ++w) {
      const int x = wsum[w];
      if (w < wv) before += x;
      tot += x;
    }
    seg_start[tid] = before + incl - c;
    if (tid == 0) seg_start[SEL_THREADS] = tot;
    n = tot;
    __syncthreads();
  }
  // Two regimes.  n <= 8192 (the usual case: a few thousand candidates in a few hundred short segments): every thread
  // reads its <= 8 keys ONCE, through a binary search of the segment starts, and all passes below run on registers.
  // n > 8192 (dense lists: up to every prediction of the image): the passes walk the segments, one wave per segment,
  // 4 x 64 keys per step (coalesced; no search).
  const bool in_regs = n <= KREG * SEL_THREADS;
  u64 kreg[KREG];
#pragma unroll
  for (int q = 0; q < KREG; ++q) {
    const int i = tid + q * SEL_THREADS;
    kreg[q] = 0ull;  // valid keys are never 0
    if (in_regs && i < n) {  // the last segment that starts at or before i
      int lo = 0;
#pragma unroll
      for (int step = SEL_THREADS / 2; step > 0; step >>= 1) {
        const int mid = lo + step;
        if (mid < sg.nseg && seg_start[mid] <= i) lo = mid;
      }
      kreg[q] = segs[(size_t)lo * sg.segcap + (i - seg_start[lo])];
    }
  }
  auto for_each_key = [&](auto&& f) {
    if (in_regs) {
#pragma unroll
      for (int q = 0; q < KREG; ++q)
        if (kreg[q] != 0ull) f(kreg[q]);
    } else {
      for (int g = wv; g < sg.nseg; g += SEL_THREADS / 64) {
        const int g1 = g + 1 < sg.nseg ? seg_start[g + 1] : seg_start[SEL_THREADS];
        const int c = g1 - seg_start[g];
        const u64* sp = segs + (size_t)g * sg.segcap;
        for (int j = lane; j < c; j += 256) {
          u64 k4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) k4[u] = j + 64 * u < c ? sp[j + 64 * u] : 0ull;
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (k4[u] != 0ull) f(k4[u]);
        }
      }
    }
  };
  const int K = min(n, topn);
  u64 T = 0ull;  // take every key >= T
  if (n > topn) {
    u64 prefix = 0ull;
    int remaining = K;
    const int shifts[6] = {52, 40, 32, 20, 8, 0};
    const int bitsn[6] = {12, 12, 8, 12, 12, 8};
    for (int lev = 0; lev < 6; ++lev) {
      const int nb = 1 << bitsn[lev];
      for (int i = tid; i < nb; i += SEL_THREADS) hist[i] = 0;
      __syncthreads();
      const u64 hi_mask = (lev == 0) ? 0ull : (~0ull << shifts[lev - 1]);
      const int sh = shifts[lev];
      for_each_key([&](u64 k) {
        if ((k & hi_mask) == prefix) atomicAdd(&hist[(int)(k >> sh) & (nb - 1)], 1);
      });
      __syncthreads();
      // thread t owns bins [t * per, (t + 1) * per); `above` = number of keys in the bins above its own
      const int per = nb >= SEL_THREADS ? nb / SEL_THREADS : 1;
      const int lo = tid * per;
      int own = 0;
      if (lo < nb)
        for (int q = 0; q < per; ++q) own += hist[lo + q];
      int incl = own;  // inclusive sum over the lanes >= mine of this wave
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int x = __shfl_down(incl, o, 64);
        if (lane + o < 64) incl += x;
      }
      if (lane == 0) wsum[wv] = incl;
      if (tid == 0) sh_done = 0;
      __syncthreads();
      int above = incl - own;
      for (int w = wv + 1; w < SEL_THREADS / 64; ++w) above += wsum[w];
      if (above < remaining && remaining <= above + own) {  // exactly one thread: the boundary bin is one of its own
        int cum = above, bin = lo + per - 1;
        for (; bin > lo; --bin) {
          if (cum + hist[bin] >= remaining) break;
          cum += hist[bin];
        }
        sh_bin = bin;
        sh_rem = remaining - cum;
        sh_done = hist[bin] == remaining - cum;
      }
      __syncthreads();
      prefix |= (u64)sh_bin << shifts[lev];
      remaining = sh_rem;
      const int done = sh_done;
      __syncthreads();
      if (done) break;
    }
    T = prefix;
  }
  if (tid == 0) sh_n = 0;
  for (int i = tid; i < TOPN_MAX; i += SEL_THREADS) item[i] = 0ull;
  __syncthreads();
  for_each_key([&](u64 k) {
    if (k >= T) {
      const int slot = atomicAdd(&sh_n, 1);
      if (slot < TOPN_MAX) item[slot] = k;
    }
  });
  __syncthreads();
  // bitonic sort, descending, TOPN_MAX elements (zeros sink to the end): thread i holds element i; exchanges at a
  // distance below 64 are wave shuffles, the ten longer ones go through LDS
  {
    u64 x = item[tid];
    for (int k2 = 2; k2 <= TOPN_MAX; k2 <<= 1) {
      const bool desc = (tid & k2) == 0;
      for (int j = k2 >> 1; j > 0; j >>= 1) {
        u64 y;
        if (j >= 64) {
          __syncthreads();  // (the partner's previous read of item[] is behind this barrier)
          item[tid] = x;
          __syncthreads();
          y = item[tid ^ j];
        } else {
          y = __shfl_xor(x, j, 64);
        }
        const bool lower = (tid & j) == 0;  // the lower index of the pair keeps the larger key when descending
        const bool take_max = lower == desc;
        x = take_max ? (x > y ? x : y) : (x < y ? x : y);
      }
    }
    __syncthreads();
    item[tid] = x;
    __syncthreads();
  }
  if (tid == 0) s_count[b] = K;
  for (int i = tid; i < topn; i += SEL_THREADS) {
    const size_t o = (size_t)b * topn + i;
    if (i < K) {
      const u64 k = item[i];
      const int a = key_anchor(k);
      int l = 0;
      for (int q = 1; q < d.nlev; ++q)
        if (a >= d.off[q]) l = q;
      const int nl = d.off[l + 1] - d.off[l];
      const size_t row = (size_t)b * nl + (a - d.off[l]);
      const f32x4 r = *reinterpret_cast<const f32x4*>(d.reg[l] + row * 4);
      float x1, y1, x2, y2;
      if (kind == 0) {
        const f32x4 an = *reinterpret_cast<const f32x4*>(geom + (size_t)a * 4);
        float aw = an[2] - an[0], ah = an[3] - an[1];
        float acx = an[0] + 0.5f * aw, acy = an[1] + 0.5f * ah;
        float bw = expf(r[2]) * aw, bh = expf(r[3]) * ah;
        float cx = r[0] * aw + acx, cy = r[1] * ah + acy;
        x1 = cx - 0.5f * bw;
        y1 = cy - 0.5f * bh;
        x2 = cx + 0.5f * bw;
        y2 = cy + 0.5f * bh;
      } else {
        float px = geom[(size_t)a * 2], py = geom[(size_t)a * 2 + 1];
        x1 = px - expf(r[0]);
        y1 = py - expf(r[1]);
        x2 = px + expf(r[2]);
        y2 = py + expf(r[3]);
      }
      s_scores[o] = key_score(k);
      s_classes[o] = (float)key_class(k);
      // truncation toward zero (astype(np.int32)), then the float the decoder hands on
      f32x4 fb = {(float)(int)x1, (float)(int)y1, (float)(int)x2, (float)(int)y2};
      *reinterpret_cast<f32x4*>(s_boxes + o * 4) = fb;
    } else {
      s_scores[o] = -1.f;
      s_classes[o] = -1.f;
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(s_boxes + o * 4) = z;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Hard NMS on <= 1024 score-sorted boxes.  Suppression bit-mask over the whole chip: workgroup (i, image), wave w decides
// the 64 pairs (i, 64 w + lane) and one ballot is the mask word; words left of the diagonal are zero.
__global__ __launch_bounds__(1024) void nms_mask_kernel(const float* __restrict__ s_boxes,
                                                         const int* __restrict__ s_count, int topn, float thr, int mode,
                                                         u64* __restrict__ mask) {
  const int b = blockIdx.y, i = blockIdx.x;
  const int n = s_count[b];
  if (i >= n) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nwords = (n + 63) >> 6;
  if (w >= nwords) return;
  u64* out = mask + ((size_t)b * TOPN_MAX + i) * 16 + w;
  if (w * 64 + 63 <= i) {  // wave-uniform
    if (lane == 0) *out = 0ull;
    return;
  }
  const int j = w * 64 + lane;
  bool drop = false;
  if (j > i && j < n) {
    const f32x4 bi = *reinterpret_cast<const f32x4*>(s_boxes + ((size_t)b * topn + i) * 4);
    const f32x4 bj = *reinterpret_cast<const f32x4*>(s_boxes + ((size_t)b * topn + j) * 4);
    const float wi = bi[2] - bi[0], hi = bi[3] - bi[1], wj = bj[2] - bj[0], hj = bj[3] - bj[1];
    const float ai = mode == 2 ? wi * hi : fmaxf(wi * hi, 0.f);  // torchvision.ops.nms does not clamp
    const float aj = mode == 2 ? wj * hj : fmaxf(wj * hj, 0.f);
    float tlx = fmaxf(bi[0], bj[0]), tly = fmaxf(bi[1], bj[1]);
    float brx = fminf(bi[2], bj[2]), bry = fminf(bi[3], bj[3]);
    float sw = fmaxf(brx - tlx, 0.f), sh = fmaxf(bry - tly, 0.f);
    float inter = sw * sh;
    float uni = (ai + aj) - inter;
    if (mode != 2) uni = fmaxf(uni, 1e-4f);
    float iou = inter / uni;
    if (mode == 1) {
      float ex = fmaxf(fmaxf(bi[2], bj[2]) - fminf(bi[0], bj[0]), 0.f);
      float ey = fmaxf(fmaxf(bi[3], bj[3]) - fminf(bi[1], bj[1]), 0.f);
      float c2 = fmaxf(ex * ex + ey * ey, 1e-4f);
      float kx = (bi[2] + bi[0]) / 2.f, ky = (bi[3] + bi[1]) / 2.f;
      float ox = (bj[2] + bj[0]) / 2.f, oy = (bj[3] + bj[1]) / 2.f;
      float dx = kx - ox, dy = ky - oy;
      float p2 = dx * dx + dy * dy;
      iou = iou - p2 / c2;
    }
    // python_nms / diou_python_nms keep iou < thr; torch_nms (torchvision) drops iou > thr (nan keeps)
    drop = mode == 2 ? (iou > thr) : !(iou < thr);
  }
  const u64 bits = __ballot(drop);
  if (lane == 0) *out = bits;
}

// The greedy pass, one workgroup per image: the image's mask rows are staged into LDS (128 KB, 16-byte loads, all of a
// thread's requests in flight at once), then a single wave walks the boxes in score order, 64 at a time: the block's
// "removed" word and its 64 diagonal words (lane i: who box i suppresses inside the block) are wave-uniform bit sets, so
// the walk visits only the boxes that are KEPT (find-first-set; scalar arithmetic), and the kept boxes' rows are OR-ed
// into the later blocks' removed words (lane w owns word w) before the next block starts.  Same result as the
// one-box-at-a-time loop of DetNMSMethod (decode.py:45-102); stops at max_obj kept boxes.
__device__ __forceinline__ u64 readlane_u64(u64 v, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((u64)hi << 32) | lo;
}

__global__ __launch_bounds__(1024) void nms_kernel(const float* __restrict__ s_scores,
                                                    const float* __restrict__ s_classes,
                                                    const float* __restrict__ s_boxes,
                                                    const int* __restrict__ s_count, const u64* __restrict__ gmask,
                                                    int topn, int max_obj, float* __restrict__ out_scores,
                                                    float* __restrict__ out_classes, float* __restrict__ out_boxes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int* keep = reinterpret_cast<int*>(smem_raw);                      // [1024]
  int* nkeep_p = reinterpret_cast<int*>(smem_raw + 1024 * 4);        // [4] (16 B slot)
  u64* mask = reinterpret_cast<u64*>(smem_raw + 1024 * 4 + 16);      // [1024][16]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = s_count[b];
  const int nwords = (n + 63) >> 6;
  {
    // rows [0, n) x 16 words = n * 8 16-byte pieces, <= 8 per thread.  (Words >= nwords of a row were never written by
    // nms_mask_kernel and are never read below.)
    const uint4* g4 = reinterpret_cast<const uint4*>(gmask + (size_t)b * TOPN_MAX * 16);
    uint4* m4 = reinterpret_cast<uint4*>(mask);
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (tid + u * 1024 < n * 8) v[u] = g4[tid + u * 1024];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (tid + u * 1024 < n * 8) m4[tid + u * 1024] = v[u];
  }
  __syncthreads();
  if (tid < 64) {
    const int lane = tid;
    u64 removed = 0ull;  // lane w (< nwords) owns word w of the removed bitmap
    int nk = 0;
    bool full = false;
    for (int blk = 0; blk < nwords && !full; ++blk) {
      const int cnt = min(64, n - blk * 64);
      const u64 vmask = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
      u64 alive = ~readlane_u64(removed, blk) & vmask;
      const u64 diag = (lane < cnt) ? mask[(size_t)(blk * 64 + lane) * 16 + blk] : 0ull;
      u64 kept = 0ull;
      while (alive != 0ull) {
        const int i = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)alive) - 1);
        kept |= 1ull << i;
        if (lane == 0) keep[nk] = blk * 64 + i;
        ++nk;
        if (nk >= max_obj) {
          full = true;
          break;
        }
        alive &= ~(1ull << i);
        alive &= ~readlane_u64(diag, i);
      }
      if (!full && blk + 1 < nwords) {
        while (kept != 0ull) {
          const int i = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)kept) - 1);
          kept &= kept - 1ull;
          if (lane < nwords) removed |= mask[(size_t)(blk * 64 + i) * 16 + lane];
        }
      }
    }
    if (lane == 0) nkeep_p[0] = nk;
  }
  __syncthreads();
  const int nk = nkeep_p[0];
  for (int i = tid; i < max_obj; i += 1024) {
    size_t o = (size_t)b * max_obj + i;
    if (i < nk) {
      int k = keep[i];
      size_t s = (size_t)b * topn + k;
      out_scores[o] = s_scores[s];
      out_classes[o] = s_classes[s];
      *reinterpret_cast<f32x4*>(out_boxes + o * 4) = *reinterpret_cast<const f32x4*>(s_boxes + s * 4);
    } else {
      out_scores[o] = -1.f;
      out_classes[o] = -1.f;
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(out_boxes + o * 4) = z;
    }
  }
}

// max over each row of 16 lanes by DPP (max is idempotent: overlapping exchanges are fine); every lane gets its row's
__device__ __forceinline__ unsigned row_max_u32(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));  // row_half_mirror
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {  // all 64 lanes active
  v = row_max_u32(v);
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return max(max(a, b), max(c, d));
}

// ------------------------------------------------------------------------------------------
// Soft-NMS (Bodla, Singh, Chellappa, Davis: "Soft-NMS -- Improving Object Detection With One Line of Code", ICCV
// 2017) on <= 1024 score-sorted boxes, one block per image, batched over the images of the launch.  The reference has
// no soft-NMS (DetNMSMethod, decode.py:26-104, knows torch_nms / python_nms / diou_python_nms): this is the mode
// BASELINE.json's north_star asks for on top of it, stated from the paper, PARITY UNPINNED (oracle/decode.py::soft_nms
// is the CPU restatement it is tested against).  Per pick: M = the live box with the highest current score (ties: the
// earlier box of the score-sorted list); every other live box i decays by its overlap with M,
//   linear   (method 0): s_i *= (1 - iou)         if iou >= nms_thr      (paper eq. 2)
//   gaussian (method 1): s_i *= exp(-iou^2/sigma)  for every i             (paper eq. 3)
// and leaves the candidate set once s_i <= prune_score.  IoU is the decoder family's (python_nms, decode.py:45-72:
// area = max(w*h, 0), union clamped at 1e-4; class-agnostic like DecodeMethod).  Picks come out in descending
// decayed-score order, so the first max_obj picks ARE the top max_obj of the full procedure: the loop stops there.
// Each thread owns one box (score in a register); the pick is a wave-level max of the score bits by DPP row operations
// + the lowest lane holding it (ballot), then one LDS round over the 16 waves: one barrier per pick.
__global__ __launch_bounds__(1024) void soft_nms_kernel(const float* __restrict__ s_scores,
                                                         const float* __restrict__ s_classes,
                                                         const float* __restrict__ s_boxes,
                                                         const int* __restrict__ s_count, int topn, float thr,
                                                         int method, float sigma, float prune_score, int max_obj,
                                                         float* __restrict__ out_scores,
                                                         float* __restrict__ out_classes,
                                                         float* __restrict__ out_boxes) {
  __shared__ unsigned long long wave_best[2][16];
  __shared__ f32x4 box_s[1024];
  __shared__ float pick_score[1024];
  __shared__ int pick_idx[1024];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = s_count[b];
  f32x4 bx = {0.f, 0.f, 0.f, 0.f};
  float sc = 0.f, ar = 0.f;
  bool live = tid < n;
  if (live) {
    bx = *reinterpret_cast<const f32x4*>(s_boxes + ((size_t)b * topn + tid) * 4);
    sc = s_scores[(size_t)b * topn + tid];
    ar = fmaxf((bx[2] - bx[0]) * (bx[3] - bx[1]), 0.f);
    live = sc > prune_score;
  }
  box_s[tid] = bx;
  __syncthreads();
  // One barrier per pick: the picks (index, decayed score) are kept in LDS and written out after the loop, so no thread
  // waits on global memory inside it; the picked box is read from the LDS copy of the list.
  int nk = 0;
  for (; nk < max_obj; ++nk) {
    // scores are > 0, so their bit patterns order like the values; 0 = no live box.  Wave maximum of the score bits by
    // DPP row operations, then the lowest lane holding it (ballot): ties go to the earlier box of the sorted list.
    const unsigned sbits = live ? __float_as_uint(sc) : 0u;
    const unsigned wmax = wave_max_u32(sbits);
    const int wl = __ffsll((unsigned long long)__ballot(sbits == wmax)) - 1;
    if (lane == 0) wave_best[nk & 1][wv] = ((unsigned long long)wmax << 32) | (unsigned)(wv * 64 + wl);
    __syncthreads();  // (slot nk & 1 is rewritten two picks later, behind the next pick's barrier)
    const unsigned long long e = wave_best[nk & 1][lane & 15];  // every row of 16 lanes holds the 16 waves' entries
    const unsigned eb = (unsigned)(e >> 32);
    const unsigned bmax = (unsigned)__builtin_amdgcn_readfirstlane((int)row_max_u32(eb));
    if (bmax == 0u) break;
    const int wbest = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)__ballot(eb == bmax)) - 1);
    const int m = __builtin_amdgcn_readlane((int)(unsigned)(e & 0xFFFFFFFFull), wbest);
    if (tid == m) {
      live = false;
      pick_idx[nk] = tid;
      pick_score[nk] = sc;  // the decayed score the box was picked with
    }
    if (live) {
      const f32x4 pm = box_s[m];
      const float am = fmaxf((pm[2] - pm[0]) * (pm[3] - pm[1]), 0.f);
      const float tlx = fmaxf(pm[0], bx[0]), tly = fmaxf(pm[1], bx[1]);
      const float brx = fminf(pm[2], bx[2]), bry = fminf(pm[3], bx[3]);
      const float sw = fmaxf(brx - tlx, 0.f), sh = fmaxf(bry - tly, 0.f);
      const float inter = sw * sh;
      const float uni = fmaxf((am + ar) - inter, 1e-4f);
      const float iou = inter / uni;
      if (method == 0) {
        if (iou >= thr) sc = sc * (1.f - iou);
      } else {
        sc = sc * expf(-(iou * iou) / sigma);
      }
      live = sc > prune_score;
    }
  }
  __syncthreads();
  for (int i = tid; i < max_obj; i += 1024) {
    const size_t o = (size_t)b * max_obj + i;
    if (i < nk) {
      const int k = pick_idx[i];
      out_scores[o] = pick_score[i];
      out_classes[o] = s_classes[(size_t)b * topn + k];
      *reinterpret_cast<f32x4*>(out_boxes + o * 4) = box_s[k];
    } else {
      out_scores[o] = -1.f;
      out_classes[o] = -1.f;
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(out_boxes + o * 4) = z;
    }
  }
}

SACV_API_BEGIN

// Workspace layout (256-byte aligned pieces): segment counters [B][nseg <= 1024] | candidate segments
// [B][nseg][segcap] u64 (nseg * segcap <= A + (DMAXL + nseg) * rows per tile <= A + SEG_SLACK) | suppression mask
// [B][1024][16] u64 | selected boxes [B][topn][4], scores [B][topn], classes [B][topn], counts [B]
#define SEG_SLACK (65536 + DMAXL * 256)  // nseg <= 65536 / rows per tile
static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

// bytes of workspace for sacv_decode_topk_nms / sacv_decode_topk_soft_nms
long long sacv_decode_workspace_bytes(int B, int A, int topn) {
  size_t n = al256((size_t)B * 1024 * 4) + al256((size_t)B * ((size_t)A + SEG_SLACK) * 8) +
             al256((size_t)B * TOPN_MAX * 16 * 8) + al256((size_t)B * topn * 16) + 2 * al256((size_t)B * topn * 4) +
             al256((size_t)B * 4);
  return (long long)n + 256;
}

static int decode_and_select(const char* who, int kind, int nlev, const float* const* cls, const float* const* reg,
                             const float* const* ctr, const int* level_off, const float* geom, int B, int A, int C,
                             float min_score, int topn, int max_obj, void* workspace, float* out_scores,
                             float* out_classes, float* out_boxes, hipStream_t st, float** s_scores_o,
                             float** s_classes_o, float** s_boxes_o, int** s_count_o, u64** mask_o) {
  (void)who;
  SACV_CHECK_ARG(kind == 0 || kind == 1, "decode: kind");
  SACV_CHECK_ARG(nlev >= 1 && nlev <= DMAXL, "decode: nlev");
  SACV_CHECK_ARG(topn >= 1 && topn <= TOPN_MAX, "decode: topn must be in [1, %d]", TOPN_MAX);
  SACV_CHECK_ARG(max_obj >= 1 && max_obj <= 1024, "decode: max_object_num must be in [1, 1024]");
  SACV_CHECK_ARG(B >= 1 && A >= 1 && C >= 1, "decode: empty batch");
  SACV_CHECK_ARG(A <= (1 << IDX_BITS) && C <= (1 << CLS_BITS),
                 "decode: at most %d predictions per image and %d classes", 1 << IDX_BITS, 1 << CLS_BITS);
  SACV_CHECK_ARG(cls && reg && level_off && level_off[nlev] == A, "decode: level offsets do not sum to A");
  SACV_CHECK_ARG(geom && workspace && out_scores && out_classes && out_boxes, "decode: null pointer");
  SACV_CHECK_ARG(min_score >= 0.f, "decode: min_score_threshold must be >= 0");
  // the LDS-tiled scan needs 16-byte rows; 128 predictions per tile up to 124 classes, 32 up to 508; else 256 (scalar)
  const int rows = (C % 4 != 0 || C > 508) ? SCAN_THREADS : (C <= 124 ? 128 : 32);
  DecodeArgs d;
  d.nlev = nlev;
  int tiles = 0;  // per image
  for (int l = 0; l < nlev; ++l) {
    SACV_CHECK_ARG(cls[l] && reg[l] && (kind == 0 || (ctr && ctr[l])), "decode: null level pointer");
    SACV_CHECK_ARG(level_off[l + 1] > level_off[l], "decode: empty level");
    d.cls[l] = cls[l];
    d.reg[l] = reg[l];
    d.ctr[l] = kind == 1 ? ctr[l] : nullptr;
    d.off[l] = level_off[l];
    d.tile0[l] = tiles;
    tiles += (level_off[l + 1] - level_off[l] + rows - 1) / rows;
  }
  d.off[nlev] = level_off[nlev];
  d.tile0[nlev] = tiles;
  const size_t lds = rows == SCAN_THREADS ? 0 : (size_t)rows * (C / 4 + 1) * 16;
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (150 * 1024) / (lds + 64)));  // resident workgroups
  SegArgs sg;
  sg.nseg = std::max(1, std::min(std::min(tiles, 65536 / rows), std::min(1024, 256 * per_cu / B)));
  sg.segcap = (tiles + sg.nseg - 1) / sg.nseg * rows;
  sg.tiles_img = tiles;
  unsigned char* ws = (unsigned char*)al256((size_t)workspace);
  sg.segcnt = (int*)ws;
  ws += al256((size_t)B * 1024 * 4);
  sg.cand = (u64*)ws;
  ws += al256((size_t)B * ((size_t)A + SEG_SLACK) * 8);
  u64* mask = (u64*)ws;
  ws += al256((size_t)B * TOPN_MAX * 16 * 8);
  float* s_boxes = (float*)ws;
  ws += al256((size_t)B * topn * 16);
  float* s_scores = (float*)ws;
  ws += al256((size_t)B * topn * 4);
  float* s_classes = (float*)ws;
  ws += al256((size_t)B * topn * 4);
  int* s_count = (int*)ws;
  SACV_CHECK_ARG((size_t)sg.nseg * sg.segcap <= (size_t)A + SEG_SLACK, "decode: internal segment bound");
  if (rows == 128)
    hipLaunchKernelGGL(decode_scan_kernel<128>, dim3(B * sg.nseg), dim3(SCAN_THREADS), lds, st, d, sg, kind, C,
                       min_score);
  else if (rows == 32)
    hipLaunchKernelGGL(decode_scan_kernel<32>, dim3(B * sg.nseg), dim3(SCAN_THREADS), lds, st, d, sg, kind, C,
                       min_score);
  else
    hipLaunchKernelGGL(decode_scan_generic_kernel, dim3(B * sg.nseg), dim3(SCAN_THREADS), 0, st, d, sg, kind, C,
                       min_score);
  SACV_CHECK_LAUNCH("decode_scan_kernel");
  hipLaunchKernelGGL(select_sort_kernel, dim3(B), dim3(SEL_THREADS), 0, st, sg, d, geom, kind, topn, s_scores,
                     s_classes, s_boxes, s_count);
  SACV_CHECK_LAUNCH("select_sort_kernel");
  *s_scores_o = s_scores;
  *s_classes_o = s_classes;
  *s_boxes_o = s_boxes;
  *s_count_o = s_count;
  *mask_o = mask;
  return SACV_OK;
}

// kind 0: RetinaNet (cls [B][n_l][C] probs, reg [B][n_l][4], geom = anchors [A][4]);
// kind 1: FCOS (plus ctr [B][n_l][1], geom = points [A][2]).
// Outputs: out_scores/out_classes [B][max_obj] (-1 padded), out_boxes [B][max_obj][4] (0 padded).
int sacv_decode_topk_nms(int kind, int nlev, const float* const* cls, const float* const* reg,
                         const float* const* ctr, const int* level_off, const float* geom, int B, int A, int C,
                         float min_score, int topn, int nms_type, float nms_thr, int max_obj, void* workspace,
                         float* out_scores, float* out_classes, float* out_boxes, void* stream) {
  SACV_CHECK_ARG(nms_type >= 0 && nms_type <= 2,
                 "decode: nms_type must be 0 (python_nms), 1 (diou_python_nms) or 2 (torch_nms)");
  hipStream_t st = (hipStream_t)stream;
  float *s_scores, *s_classes, *s_boxes;
  int* s_count;
  u64* mask;
  int rc = decode_and_select("decode_topk_nms", kind, nlev, cls, reg, ctr, level_off, geom, B, A, C, min_score, topn,
                             max_obj, workspace, out_scores, out_classes, out_boxes, st, &s_scores, &s_classes,
                             &s_boxes, &s_count, &mask);
  if (rc != SACV_OK) return rc;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(topn, B), dim3(1024), 0, st, s_boxes, s_count, topn, nms_thr, nms_type,
                     mask);
  SACV_CHECK_LAUNCH("nms_mask_kernel");
  size_t lds = 1024 * 4 + 16 + (size_t)1024 * 16 * 8;
  static std::once_flag attr_once;
  std::call_once(attr_once, [lds] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nms_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  });
  hipLaunchKernelGGL(nms_kernel, dim3(B), dim3(1024), lds, st, s_scores, s_classes, s_boxes, s_count, mask, topn,
                     max_obj, out_scores, out_classes, out_boxes);
  SACV_CHECK_LAUNCH("nms_kernel");
  return SACV_OK;
}

// The same front end (argmax / box decode / score threshold / exact top-n / sort) followed by batched Soft-NMS
// (soft_nms_kernel above).  method 0 = linear (decay when IoU >= nms_thr), 1 = gaussian (sigma); boxes whose decayed
// score falls to <= prune_score leave the candidate set.  Outputs as sacv_decode_topk_nms, scores = decayed scores.
int sacv_decode_topk_soft_nms(int kind, int nlev, const float* const* cls, const float* const* reg,
                              const float* const* ctr, const int* level_off, const float* geom, int B, int A, int C,
                              float min_score, int topn, int method, float nms_thr, float sigma, float prune_score,
                              int max_obj, void* workspace, float* out_scores, float* out_classes, float* out_boxes,
                              void* stream) {
  SACV_CHECK_ARG(method == 0 || method == 1, "soft_nms: method must be 0 (linear) or 1 (gaussian)");
  SACV_CHECK_ARG(sigma > 0.f && prune_score >= 0.f && nms_thr >= 0.f, "soft_nms: sigma > 0, thresholds >= 0");
  hipStream_t st = (hipStream_t)stream;
  float *s_scores, *s_classes, *s_boxes;
  int* s_count;
  u64* mask;
  int rc = decode_and_select("decode_topk_soft_nms", kind, nlev, cls, reg, ctr, level_off, geom, B, A, C, min_score,
                             topn, max_obj, workspace, out_scores, out_classes, out_boxes, st, &s_scores, &s_classes,
                             &s_boxes, &s_count, &mask);
  (void)mask;
  if (rc != SACV_OK) return rc;
  hipLaunchKernelGGL(soft_nms_kernel, dim3(B), dim3(1024), 0, st, s_scores, s_classes, s_boxes, s_count, topn, nms_thr,
                     method, sigma, prune_score, max_obj, out_scores, out_classes, out_boxes);
  SACV_CHECK_LAUNCH("soft_nms_kernel");
  return SACV_OK;
}

SACV_API_END
