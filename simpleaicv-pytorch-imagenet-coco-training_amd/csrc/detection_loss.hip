// Detection training-target and loss kernels (RetinaNet path; FCOS re-uses focal / IoU pieces).
//
//   retina_anchors   simpleAICV/detection/models/anchor.py:18-86  (bit-exact vs NumPy fp32)
//   retina_assign    simpleAICV/detection/losses.py:322-409 + IoU :54-70 (bit-exact labels)
//   focal_fwd_bwd    losses.py:196 (clamp), :220-261 (RetinaLoss) / :513-548 (FCOSLoss)
//   retina_box_loss  losses.py:263-320 (SmoothL1), :282-293 + :411-429 (IoU / GIoU)
//
// This translation unit is compiled with -ffp-contract=off: the IoU / target arithmetic must
// round exactly like the reference's separate fp32 torch ops (no fused multiply-add).
// Loss sums use wavefront shuffles + one LDS hop per block and a fixed-order fp64 second
// stage: deterministic, no float atomics.  The positive count is an integer atomic.
#include "sacv_common.h"
#include <math.h>

#define MAXL 8
#define MAXA 16
#define MAXG 128

struct AnchorArgs {
  float base[MAXL][MAXA][4];
  int fh[MAXL], fw[MAXL];
  float stride[MAXL];
  int off[MAXL + 1];  // anchor offset of each level
  int nlev, A;
};

__global__ void retina_anchors_kernel(const AnchorArgs a, float* __restrict__ out) {
  int total = a.off[a.nlev];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int l = 0;
    for (int k = 1; k < a.nlev; ++k)
      if (i >= a.off[k]) l = k;
    int r = i - a.off[l];
    int an = r % a.A;
    int pos = r / a.A;
    int x = pos % a.fw[l], y = pos / a.fw[l];
    float sx = ((float)x + 0.5f) * a.stride[l];  // exact: (i + 0.5) * stride is representable
    float sy = ((float)y + 0.5f) * a.stride[l];
    f32x4 v;
    v[0] = a.base[l][an][0] + sx;
    v[1] = a.base[l][an][1] + sy;
    v[2] = a.base[l][an][2] + sx;
    v[3] = a.base[l][an][3] + sy;
    *reinterpret_cast<f32x4*>(out + (size_t)i * 4) = v;
  }
}

__global__ void fcos_points_kernel(const AnchorArgs a, float* __restrict__ out) {
  int total = a.off[a.nlev];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int l = 0;
    for (int k = 1; k < a.nlev; ++k)
      if (i >= a.off[k]) l = k;
    int pos = i - a.off[l];
    int x = pos % a.fw[l], y = pos / a.fw[l];
    out[(size_t)i * 2 + 0] = ((float)x + 0.5f) * a.stride[l];
    out[(size_t)i * 2 + 1] = ((float)y + 0.5f) * a.stride[l];
  }
}

// ------------------------------------------------------------------------------------------ assign
// targets[b][a][0:4] = box target (tx,ty,tw,th when smooth_l1 else matched GT xyxy), [4] = label
// (-1 ignore, 0 background, cls+1 positive).  counts[0] += #positive, counts[1] += #(label >= 0).
__global__ __launch_bounds__(256) void retina_assign_kernel(const float* __restrict__ anchors,
                                                             const float* __restrict__ annots,
                                                             float* __restrict__ targets, int* __restrict__ counts,
                                                             int A, int G, int smooth_l1) {
  __shared__ float gt[MAXG][5];
  __shared__ int ng_s;
  __shared__ int cnt_s[2];
  const int b = blockIdx.y;
  if (threadIdx.x == 0) {
    int n = 0;
    const float* an = annots + (size_t)b * G * 5;
    for (int j = 0; j < G; ++j) {
      if (an[j * 5 + 4] >= 0.f) {
        for (int e = 0; e < 5; ++e) gt[n][e] = an[j * 5 + e];
        ++n;
      }
    }
    ng_s = n;
    cnt_s[0] = 0;
    cnt_s[1] = 0;
  }
  __syncthreads();
  const int ng = ng_s;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int is_pos = 0, is_valid = 0;
  if (i < A) {
    float* out = targets + ((size_t)b * A + i) * 5;
    if (ng == 0) {
      for (int e = 0; e < 5; ++e) out[e] = -1.f;
    } else {
      f32x4 an = *reinterpret_cast<const f32x4*>(anchors + (size_t)i * 4);
      float aw = fmaxf(an[2] - an[0], 0.f), ah = fmaxf(an[3] - an[1], 0.f);
      float a_area = aw * ah;
      float best = -1.f;
      int bi = 0;
      for (int g = 0; g < ng; ++g) {
        float tlx = fmaxf(an[0], gt[g][0]), tly = fmaxf(an[1], gt[g][1]);
        float brx = fminf(an[2], gt[g][2]), bry = fminf(an[3], gt[g][3]);
        float w = fmaxf(brx - tlx, 0.f), h = fmaxf(bry - tly, 0.f);
        float inter = w * h;
        float gw = fmaxf(gt[g][2] - gt[g][0], 0.f), gh = fmaxf(gt[g][3] - gt[g][1], 0.f);
        float g_area = gw * gh;
        float uni = fmaxf((a_area + g_area) - inter, 1e-4f);
        float iou = inter / uni;
        if (iou > best) {  // strict: first maximal index wins, like torch.max(dim=1)
          best = iou;
          bi = g;
        }
      }
      float label = -1.f;
      if (best < 0.4f) label = 0.f;
      if (best >= 0.5f) label = gt[bi][4] + 1.f;
      float t0, t1, t2, t3;
      if (smooth_l1) {
        float awr = an[2] - an[0], ahr = an[3] - an[1];
        float acx = an[0] + 0.5f * awr, acy = an[1] + 0.5f * ahr;
        float gw = fmaxf(gt[bi][2] - gt[bi][0], 1e-4f), gh = fmaxf(gt[bi][3] - gt[bi][1], 1e-4f);
        float gcx = gt[bi][0] + 0.5f * gw, gcy = gt[bi][1] + 0.5f * gh;
        t0 = (gcx - acx) / awr;
        t1 = (gcy - acy) / ahr;
        t2 = logf(gw / awr);
        t3 = logf(gh / ahr);
      } else {
        t0 = gt[bi][0];
        t1 = gt[bi][1];
        t2 = gt[bi][2];
        t3 = gt[bi][3];
      }
      out[0] = t0;
      out[1] = t1;
      out[2] = t2;
      out[3] = t3;
      out[4] = label;
      is_pos = label > 0.f;
      is_valid = label >= 0.f;
    }
  }
  unsigned long long mp = __ballot(is_pos), mv = __ballot(is_valid);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&cnt_s[0], __popcll(mp));
    atomicAdd(&cnt_s[1], __popcll(mv));
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (cnt_s[0]) atomicAdd(&counts[0], cnt_s[0]);
    if (cnt_s[1]) atomicAdd(&counts[1], cnt_s[1]);
  }
}

// ------------------------------------------------------------------------------------------ focal
struct LevelPtrs {
  const float* in[MAXL];
  float* grad[MAXL];
  int off[MAXL + 1];  // per-image anchor/point offset of each level
  int nlev;
};

// One thread = one float4 of class probabilities of one anchor.
//   loss_e = alpha_t * (1 - p_t)^gamma * (-log p_t),  p clamped to [1e-4, 1-1e-4]
//   anchors with label < 0 are skipped (RetinaLoss); grad is zero outside the clamp interval.
// grad = d(sum loss)/dp * weight / num_pos (written for every element, zeros included).
__global__ __launch_bounds__(256) void focal_fwd_bwd_kernel(const LevelPtrs lp, const float* __restrict__ targets,
                                                             int tstride, int B, int A, int C, float alpha,
                                                             float gamma, float weight,
                                                             const int* __restrict__ counts,
                                                             float* __restrict__ partial) {
  const int cg = (C + 3) >> 2;        // class groups of 4; the last one is partial when C % 4 != 0 (Objects365: 365)
  const bool vec = (C & 3) == 0;      // class rows are 16-byte aligned only then: otherwise element-wise accesses
  const size_t total = (size_t)B * A * cg;
  const float lo = 1e-4f, hi = (float)(1. - 1e-4);
  const int npos = counts[0];
  const float gscale = npos > 0 ? weight / (float)npos : 0.f;
  float lsum = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c4 = (int)(i % cg);
    size_t q = i / cg;
    int a = (int)(q % A);
    int b = (int)(q / A);
    int l = 0;
    for (int k = 1; k < lp.nlev; ++k)
      if (a >= lp.off[k]) l = k;
    int nl = lp.off[l + 1] - lp.off[l];
    size_t o = ((size_t)b * nl + (a - lp.off[l])) * C + c4 * 4;
    float label = targets[((size_t)b * A + a) * tstride + 4];
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    const int ne = vec ? 4 : min(4, C - c4 * 4);  // valid classes of this group
    if (label >= 0.f) {
      f32x4 praw = {0.5f, 0.5f, 0.5f, 0.5f};
      if (vec) {
        praw = *reinterpret_cast<const f32x4*>(lp.in[l] + o);
      } else {
        for (int e = 0; e < ne; ++e) praw[e] = lp.in[l][o + e];
      }
      int cls = (int)label - 1;  // -1 for background
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (e >= ne) break;
        float p = fminf(fmaxf(praw[e], lo), hi);
        bool inside = (praw[e] >= lo) && (praw[e] <= hi);
        bool t = (c4 * 4 + e) == cls;
        float pt = t ? p : 1.f - p;
        float at = t ? alpha : 1.f - alpha;
        float om = 1.f - pt;
        float lg = logf(pt);
        float mod, dmod;  // (1-pt)^gamma and its derivative w.r.t. pt
        if (gamma == 2.f) {
          mod = om * om;
          dmod = -2.f * om;
        } else {
          mod = powf(om, gamma);
          dmod = -gamma * powf(om, gamma - 1.f);
        }
        lsum += at * mod * (-lg);
        // dL/dpt = at * (dmod * (-lg) - mod / pt);  dpt/dp = t ? 1 : -1
        float dpt = at * (dmod * (-lg) - mod / pt);
        float dp = t ? dpt : -dpt;
        g[e] = inside ? dp * gscale : 0.f;
      }
    }
    if (lp.grad[l]) {
      if (vec) {
        *reinterpret_cast<f32x4*>(lp.grad[l] + o) = g;
      } else {
        for (int e = 0; e < ne; ++e) lp.grad[l][o + e] = g[e];
      }
    }
  }
  __shared__ float red[4];
  float w = wave_sum(lsum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// loss_out[slot] = weight * sum(partial) / num_pos   (0 when there are no positives)
__global__ void loss_finalize_kernel(const float* __restrict__ partial, int n, const int* __restrict__ counts,
                                     float weight, float* __restrict__ loss_out, int slot) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += (double)partial[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) {
    int npos = counts[0];
    loss_out[slot] = npos > 0 ? (float)(s / (double)npos) * weight : 0.f;
  }
}

// ------------------------------------------------------------------------------------------ box loss
// torch subgradient conventions: maximum/minimum split 0.5/0.5 on ties, clamp passes on the bound.
__device__ __forceinline__ float dmax_a(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }
__device__ __forceinline__ float dmin_a(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }

// ---- DIoU / CIoU / EIoU (IoUMethod, losses.py:83-116) by forward-mode differentiation: a value with its four
// partial derivatives w.r.t. the predicted box, propagated through the reference's operation sequence with
// torch's subgradient conventions (max/min split ties 0.5/0.5, clamp passes the gradient on the bound, the CIoU
// alpha is a constant as under torch.no_grad()).
struct D4 {
  float v, d[4];
};
__device__ __forceinline__ D4 d4_var(float v, int k) {
  D4 r = {v, {0.f, 0.f, 0.f, 0.f}};
  r.d[k] = 1.f;
  return r;
}
__device__ __forceinline__ D4 d4_scale(D4 a, float f) {  // derivative chain: value given separately
  for (int k = 0; k < 4; ++k) a.d[k] *= f;
  return a;
}
__device__ __forceinline__ D4 operator+(D4 a, const D4& b) {
  a.v += b.v;
  for (int k = 0; k < 4; ++k) a.d[k] += b.d[k];
  return a;
}
__device__ __forceinline__ D4 operator-(D4 a, const D4& b) {
  a.v -= b.v;
  for (int k = 0; k < 4; ++k) a.d[k] -= b.d[k];
  return a;
}
__device__ __forceinline__ D4 operator*(const D4& a, const D4& b) {
  D4 r;
  r.v = a.v * b.v;
  for (int k = 0; k < 4; ++k) r.d[k] = a.d[k] * b.v + a.v * b.d[k];
  return r;
}
__device__ __forceinline__ D4 operator/(const D4& a, const D4& b) {
  D4 r;
  r.v = a.v / b.v;
  const float inv = 1.f / b.v;
  for (int k = 0; k < 4; ++k) r.d[k] = (a.d[k] - r.v * b.d[k]) * inv;
  return r;
}
__device__ __forceinline__ D4 d4_addc(D4 a, float c) {
  a.v += c;
  return a;
}
__device__ __forceinline__ D4 d4_rsubc(float c, D4 a) {  // c - a
  a.v = c - a.v;
  for (int k = 0; k < 4; ++k) a.d[k] = -a.d[k];
  return a;
}
__device__ __forceinline__ D4 d4_mulc(D4 a, float c) {
  a.v *= c;
  for (int k = 0; k < 4; ++k) a.d[k] *= c;
  return a;
}
__device__ __forceinline__ D4 d4_maxc(D4 a, float c) {
  const float f = a.v > c ? 1.f : (a.v == c ? 0.5f : 0.f);
  a = d4_scale(a, f);
  a.v = fmaxf(a.v, c);
  return a;
}
__device__ __forceinline__ D4 d4_minc(D4 a, float c) {
  const float f = a.v < c ? 1.f : (a.v == c ? 0.5f : 0.f);
  a = d4_scale(a, f);
  a.v = fminf(a.v, c);
  return a;
}
__device__ __forceinline__ D4 d4_clamp_min(D4 a, float m) {  // torch.clamp(min=m): gradient where a >= m
  const float f = a.v >= m ? 1.f : 0.f;
  a = d4_scale(a, f);
  a.v = fmaxf(a.v, m);
  return a;
}
__device__ __forceinline__ D4 d4_sq(const D4& a) {
  D4 r;
  r.v = a.v * a.v;
  for (int k = 0; k < 4; ++k) r.d[k] = 2.f * a.v * a.d[k];
  return r;
}
__device__ __forceinline__ D4 d4_atan(const D4& a) {
  D4 r;
  r.v = atanf(a.v);
  const float f = 1.f / (1.f + a.v * a.v);
  for (int k = 0; k < 4; ++k) r.d[k] = a.d[k] * f;
  return r;
}

// type: 2 DIoU, 3 CIoU, 4 EIoU
__device__ float iou_family_loss_dual(const float p[4], const float g[4], int type, float dp[4]) {
  const D4 x0 = d4_var(p[0], 0), y0 = d4_var(p[1], 1), x1 = d4_var(p[2], 2), y1 = d4_var(p[3], 3);
  const D4 w = d4_clamp_min(d4_minc(x1, g[2]) - d4_maxc(x0, g[0]), 0.f);
  const D4 h = d4_clamp_min(d4_minc(y1, g[3]) - d4_maxc(y0, g[1]), 0.f);
  const D4 inter = w * h;
  const D4 pw = d4_clamp_min(x1 - x0, 0.f), ph = d4_clamp_min(y1 - y0, 0.f);
  const float gw = fmaxf(g[2] - g[0], 0.f), gh = fmaxf(g[3] - g[1], 0.f);
  const D4 uni = d4_clamp_min(d4_addc(pw * ph, gw * gh) - inter, 1e-4f);
  const D4 iou = inter / uni;
  const D4 ew = d4_clamp_min(d4_maxc(x1, g[2]) - d4_minc(x0, g[0]), 0.f);
  const D4 eh = d4_clamp_min(d4_maxc(y1, g[3]) - d4_minc(y0, g[1]), 0.f);
  const D4 c2 = d4_clamp_min(d4_sq(ew) + d4_sq(eh), 1e-4f);
  const D4 pcx = d4_mulc(x1 + x0, 0.5f), pcy = d4_mulc(y1 + y0, 0.5f);
  const float gcx = (g[2] + g[0]) / 2.f, gcy = (g[3] + g[1]) / 2.f;
  const D4 p2 = d4_sq(d4_addc(pcx, -gcx)) + d4_sq(d4_addc(pcy, -gcy));
  D4 pen = p2 / c2;
  if (type == 3) {
    const D4 da = d4_rsubc(atanf(gw / gh), d4_atan(pw / ph));  // atan(gw/gh) - atan(pw/ph)
    const D4 v = d4_mulc(d4_sq(da), 0.40528473456935109f);     // 4 / pi^2
    const float alpha = v.v / fmaxf(1.f - iou.v + v.v, 1e-4f);
    pen = pen + d4_mulc(v, alpha);
  } else if (type == 4) {
    const D4 cw2 = d4_clamp_min(d4_sq(ew), 1e-4f), ch2 = d4_clamp_min(d4_sq(eh), 1e-4f);
    pen = pen + d4_sq(d4_rsubc(gw, pw)) / cw2 + d4_sq(d4_rsubc(gh, ph)) / ch2;
  }
  const D4 val = iou - pen;
  for (int k = 0; k < 4; ++k) dp[k] = -val.d[k];
  return 1.f - val.v;
}

// loss = 1 - IoU-family(pred, gt);  returns loss, writes d loss / d pred[4].
// type: 0 IoU, 1 GIoU (hand-derived below), 2 DIoU, 3 CIoU, 4 EIoU (forward-mode, above)
__device__ float iou_loss_fwd_bwd(const float p[4], const float g[4], int type, float dp[4]) {
  if (type >= 2) return iou_family_loss_dual(p, g, type, dp);
  float tlx = fmaxf(p[0], g[0]), tly = fmaxf(p[1], g[1]);
  float brx = fminf(p[2], g[2]), bry = fminf(p[3], g[3]);
  float rw = brx - tlx, rh = bry - tly;
  float w = fmaxf(rw, 0.f), h = fmaxf(rh, 0.f);
  float inter = w * h;
  float pwr = p[2] - p[0], phr = p[3] - p[1];
  float pw = fmaxf(pwr, 0.f), ph = fmaxf(phr, 0.f);
  float gw = fmaxf(g[2] - g[0], 0.f), gh = fmaxf(g[3] - g[1], 0.f);
  float parea = pw * ph, garea = gw * gh;
  float ur = (parea + garea) - inter;
  float uni = fmaxf(ur, 1e-4f);
  float iou = inter / uni;
  // ---- backward of iou w.r.t. inter, uni
  float d_inter = 1.f / uni;
  float d_uni = -inter / (uni * uni);
  float value = iou;
  float d_earea = 0.f;
  float ewr = 0.f, ehr = 0.f, ew = 0.f, eh = 0.f, ear = 0.f;
  float etlx = 0.f, etly = 0.f, ebrx = 0.f, ebry = 0.f;
  if (type == 1) {
    etlx = fminf(p[0], g[0]);
    etly = fminf(p[1], g[1]);
    ebrx = fmaxf(p[2], g[2]);
    ebry = fmaxf(p[3], g[3]);
    ewr = ebrx - etlx;
    ehr = ebry - etly;
    ew = fmaxf(ewr, 0.f);
    eh = fmaxf(ehr, 0.f);
    ear = ew * eh;
    float earea = fmaxf(ear, 1e-4f);
    // giou = iou - (earea - uni) / earea
    value = iou - (earea - uni) / earea;
    d_uni += 1.f / earea;
    d_earea = -(uni / (earea * earea));  // d/dearea of -(earea-uni)/earea = -(uni/earea^2)
    if (!(ear >= 1e-4f)) d_earea = 0.f;
  }
  float d_ur = (ur >= 1e-4f) ? d_uni : 0.f;
  // uni_raw = parea + garea - inter
  float d_parea = d_ur;
  d_inter += -d_ur;
  // inter = w*h
  float d_w = d_inter * h, d_h = d_inter * w;
  float d_rw = (rw >= 0.f) ? d_w : 0.f, d_rh = (rh >= 0.f) ? d_h : 0.f;
  float d_brx = d_rw, d_tlx = -d_rw, d_bry = d_rh, d_tly = -d_rh;
  float d0 = d_tlx * dmax_a(p[0], g[0]);
  float d1 = d_tly * dmax_a(p[1], g[1]);
  float d2 = d_brx * dmin_a(p[2], g[2]);
  float d3 = d_bry * dmin_a(p[3], g[3]);
  // parea = pw*ph
  float d_pw = d_parea * ph, d_ph = d_parea * pw;
  float d_pwr = (pwr >= 0.f) ? d_pw : 0.f, d_phr = (phr >= 0.f) ? d_ph : 0.f;
  d2 += d_pwr;
  d0 -= d_pwr;
  d3 += d_phr;
  d1 -= d_phr;
  if (type == 1) {
    float d_ew = d_earea * eh, d_eh = d_earea * ew;
    float d_ewr = (ewr >= 0.f) ? d_ew : 0.f, d_ehr = (ehr >= 0.f) ? d_eh : 0.f;
    d2 += d_ewr * dmax_a(p[2], g[2]);
    d0 += -d_ewr * dmin_a(p[0], g[0]);
    d3 += d_ehr * dmax_a(p[3], g[3]);
    d1 += -d_ehr * dmin_a(p[1], g[1]);
  }
  // loss = 1 - value
  dp[0] = -d0;
  dp[1] = -d1;
  dp[2] = -d2;
  dp[3] = -d3;
  return 1.f - value;
}

struct RegPtrs {
  const float* in[MAXL];
  float* grad[MAXL];
  int off[MAXL + 1];
  int nlev;
};

// box_type: 0 SmoothL1 (targets are tx,ty,tw,th), 1 IoU, 2 GIoU, 3 DIoU, 4 CIoU, 5 EIoU (targets are GT xyxy)
__global__ __launch_bounds__(256) void retina_box_loss_kernel(const RegPtrs rp, const float* __restrict__ targets,
                                                               const float* __restrict__ anchors, int B, int A,
                                                               int box_type, float beta, float weight,
                                                               const int* __restrict__ counts,
                                                               float* __restrict__ partial) {
  const size_t total = (size_t)B * A;
  const int npos = counts[0];
  const float gscale = npos > 0 ? weight / (float)npos : 0.f;
  float lsum = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int a = (int)(i % A);
    int b = (int)(i / A);
    int l = 0;
    for (int k = 1; k < rp.nlev; ++k)
      if (a >= rp.off[k]) l = k;
    int nl = rp.off[l + 1] - rp.off[l];
    size_t o = ((size_t)b * nl + (a - rp.off[l])) * 4;
    const float* tg = targets + i * 5;
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (tg[4] > 0.f) {
      f32x4 r = *reinterpret_cast<const f32x4*>(rp.in[l] + o);
      if (box_type == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float d = r[e] - tg[e];
          float x = fabsf(d);
          float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
          if (x >= beta) {
            lsum += x - 0.5f * beta;
            g[e] = sg * gscale;
          } else {
            lsum += 0.5f * (x * x) / beta;
            g[e] = (x / beta) * sg * gscale;
          }
        }
      } else {
        f32x4 an = *reinterpret_cast<const f32x4*>(anchors + (size_t)a * 4);
        float aw = an[2] - an[0], ah = an[3] - an[1];
        float acx = an[0] + 0.5f * aw, acy = an[1] + 0.5f * ah;
        float ew = expf(r[2]), eh = expf(r[3]);
        float bw = ew * aw, bh = eh * ah;
        float cx = r[0] * aw + acx, cy = r[1] * ah + acy;
        float p[4] = {cx - 0.5f * bw, cy - 0.5f * bh, cx + 0.5f * bw, cy + 0.5f * bh};
        float gg[4] = {tg[0], tg[1], tg[2], tg[3]};
        float dp[4];
        lsum += iou_loss_fwd_bwd(p, gg, box_type - 1, dp);
        // chain to tx,ty,tw,th
        float dcx = dp[0] + dp[2], dcy = dp[1] + dp[3];
        float dbw = 0.5f * (dp[2] - dp[0]), dbh = 0.5f * (dp[3] - dp[1]);
        g[0] = dcx * aw * gscale;
        g[1] = dcy * ah * gscale;
        g[2] = dbw * bw * gscale;
        g[3] = dbh * bh * gscale;
      }
    }
    if (rp.grad[l]) *reinterpret_cast<f32x4*>(rp.grad[l] + o) = g;
  }
  __shared__ float red[4];
  float w = wave_sum(lsum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}


// ------------------------------------------------------------------------------------------ FCOS
// FCOSLoss.get_batch_position_annotations (losses.py:612-833): per (image, point) one pass over the GTs.
struct FcosLevels {
  int off[MAXL + 1];
  float stride[MAXL];
  float mi_lo[MAXL], mi_hi[MAXL];
  int nlev;
};

// targets[b][p] = (l, t, r, b, cls+1 | 0, centerness, px, py);  counts[0] += #positive
__global__ __launch_bounds__(256) void fcos_assign_kernel(const FcosLevels lv, const float* __restrict__ points,
                                                           const float* __restrict__ annots,
                                                           float* __restrict__ targets, int* __restrict__ counts,
                                                           int P, int G, float radius, int use_center_sample) {
  __shared__ float gt[MAXG][5];
  __shared__ int ng_s;
  __shared__ int cnt_s;
  const int b = blockIdx.y;
  if (threadIdx.x == 0) {
    int n = 0;
    const float* an = annots + (size_t)b * G * 5;
    for (int j = 0; j < G; ++j) {
      if (an[j * 5 + 4] >= 0.f) {
        for (int e = 0; e < 5; ++e) gt[n][e] = an[j * 5 + e];
        ++n;
      }
    }
    ng_s = n;
    cnt_s = 0;
  }
  __syncthreads();
  const int ng = ng_s;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int is_pos = 0;
  if (i < P) {
    int l = 0;
    for (int k = 1; k < lv.nlev; ++k)
      if (i >= lv.off[k]) l = k;
    const float px = points[(size_t)i * 2], py = points[(size_t)i * 2 + 1];
    const float judge = lv.stride[l] * radius;
    const float lo = lv.mi_lo[l], hi = lv.mi_hi[l];
    float best_area = 100000000.f;
    int bi = -1;
    float bl = 0.f, bt = 0.f, br = 0.f, bb = 0.f;
    for (int g = 0; g < ng; ++g) {
      float cl = px - gt[g][0], ct = py - gt[g][1], cr = gt[g][2] - px, cb = gt[g][3] - py;
      bool ok = fminf(fminf(cl, ct), fminf(cr, cb)) > 0.f;
      if (use_center_sample) {
        float cx = (gt[g][2] + gt[g][0]) / 2.f, cy = (gt[g][3] + gt[g][1]) / 2.f;
        float dx = px - cx, dy = py - cy;
        float dist = sqrtf(dx * dx + dy * dy);
        ok = ok && (dist < judge);
      }
      float cmax = ok ? fmaxf(fmaxf(cl, ct), fmaxf(cr, cb)) : 0.f;
      ok = ok && (cmax > lo) && (cmax < hi);
      float gw = gt[g][2] - gt[g][0], gh = gt[g][3] - gt[g][1];
      float area = ok ? gw * gh : 100000000.f;
      // argmin over ALL columns with first-index ties (torch.min): an invalid column carries INF
      if (g == 0 || area < best_area) {
        best_area = area;
        bi = ok ? g : -1;
        bl = cl;
        bt = ct;
        br = cr;
        bb = cb;
      }
    }
    float* out = targets + ((size_t)b * P + i) * 8;
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, px, py};
    if (bi >= 0) {
      o0[0] = bl;
      o0[1] = bt;
      o0[2] = br;
      o0[3] = bb;
      o1[0] = gt[bi][4] + 1.f;
      o1[1] = sqrtf((fminf(bl, br) / fmaxf(bl, br)) * (fminf(bt, bb) / fmaxf(bt, bb)));
      is_pos = 1;
    }
    *reinterpret_cast<f32x4*>(out) = o0;
    *reinterpret_cast<f32x4*>(out + 4) = o1;
  }
  unsigned long long mp = __ballot(is_pos);
  if ((threadIdx.x & 63) == 0 && mp) atomicAdd(&cnt_s, __popcll(mp));
  __syncthreads();
  if (threadIdx.x == 0 && cnt_s) atomicAdd(&counts[0], cnt_s);
}

struct FcosPtrs {
  const float* reg[MAXL];
  const float* ctr[MAXL];
  float* dreg[MAXL];
  float* dctr[MAXL];
  int off[MAXL + 1];
  int nlev;
};

// FCOSLoss.compute_batch_iou_loss (losses.py:550-586) and compute_batch_centerness_loss (:588-610),
// forward + gradients in one pass over the points.  partial[0][blk] = IoU-loss sum, partial[1][blk] = BCE sum.
__global__ __launch_bounds__(256) void fcos_reg_ctr_loss_kernel(const FcosPtrs fp, const float* __restrict__ targets,
                                                                 int B, int P, int iou_type, float w_reg,
                                                                 float w_ctr, const int* __restrict__ counts,
                                                                 float* __restrict__ partial, int nblk) {
  const size_t total = (size_t)B * P;
  const int npos = counts[0];
  const float s_reg = npos > 0 ? w_reg / (float)npos : 0.f;
  const float s_ctr = npos > 0 ? w_ctr / (float)npos : 0.f;
  const float lo = 1e-4f, hi = (float)(1. - 1e-4);
  float sum_r = 0.f, sum_c = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int a = (int)(i % P), b = (int)(i / P);
    int l = 0;
    for (int k = 1; k < fp.nlev; ++k)
      if (a >= fp.off[k]) l = k;
    int nl = fp.off[l + 1] - fp.off[l];
    size_t row = (size_t)b * nl + (a - fp.off[l]);
    const float* tg = targets + i * 8;
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    float gc = 0.f;
    if (tg[4] > 0.f) {
      f32x4 r = *reinterpret_cast<const f32x4*>(fp.reg[l] + row * 4);
      float e0 = expf(r[0]), e1 = expf(r[1]), e2 = expf(r[2]), e3 = expf(r[3]);
      float px = tg[6], py = tg[7], ct = tg[5];
      float p[4] = {px - e0, py - e1, px + e2, py + e3};
      float gg[4] = {px - tg[0], py - tg[1], px + tg[2], py + tg[3]};
      float dp[4];
      float li = iou_loss_fwd_bwd(p, gg, iou_type, dp);
      sum_r += li * ct;
      g[0] = -dp[0] * e0 * ct * s_reg;
      g[1] = -dp[1] * e1 * ct * s_reg;
      g[2] = dp[2] * e2 * ct * s_reg;
      g[3] = dp[3] * e3 * ct * s_reg;
      float craw = fp.ctr[l][row];
      float c = fminf(fmaxf(craw, lo), hi);
      sum_c += -(ct * logf(c) + (1.f - ct) * logf(1.f - c));
      bool inside = (craw >= lo) && (craw <= hi);
      gc = inside ? (-(ct / c) + (1.f - ct) / (1.f - c)) * s_ctr : 0.f;
    }
    if (fp.dreg[l]) *reinterpret_cast<f32x4*>(fp.dreg[l] + row * 4) = g;
    if (fp.dctr[l]) fp.dctr[l][row] = gc;
  }
  __shared__ float red[8];
  float wr = wave_sum(sum_r), wc = wave_sum(sum_c);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = wr;
    red[4 + (threadIdx.x >> 6)] = wc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    partial[nblk + blockIdx.x] = red[4] + red[5] + red[6] + red[7];
  }
}

// FCOS output head split (fcos.py:70-80, head.py:176-179): y8 [R][8] = (reg_raw[4], ctr_logit, pad[3])
//   reg[R][4] = reg_raw * exp(scale),  ctr[R][1] = sigmoid(ctr_logit)
__global__ __launch_bounds__(256) void fcos_split_fwd_kernel(const float* __restrict__ y8,
                                                              const float* __restrict__ scale,
                                                              float* __restrict__ reg, float* __restrict__ ctr,
                                                              int R) {
  const float es = expf(scale[0]);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < R; i += gridDim.x * blockDim.x) {
    f32x4 a = *reinterpret_cast<const f32x4*>(y8 + (size_t)i * 8);
    float c = y8[(size_t)i * 8 + 4];
    *reinterpret_cast<f32x4*>(reg + (size_t)i * 4) = a * es;
    ctr[i] = 1.f / (1.f + __expf(-c));
  }
}
// dy8 = (dreg*exp(scale), dctr*c*(1-c), 0, 0, 0);  partial[blk] = sum dreg*reg  (= d loss / d scale)
__global__ __launch_bounds__(256) void fcos_split_bwd_kernel(const float* __restrict__ dreg,
                                                              const float* __restrict__ dctr,
                                                              const float* __restrict__ reg,
                                                              const float* __restrict__ ctr,
                                                              const float* __restrict__ scale, float gscale,
                                                              float* __restrict__ dy8, float* __restrict__ partial,
                                                              int R) {
  const float es = expf(scale[0]);
  float s = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < R; i += gridDim.x * blockDim.x) {
    f32x4 d = *reinterpret_cast<const f32x4*>(dreg + (size_t)i * 4) * gscale;
    f32x4 r = *reinterpret_cast<const f32x4*>(reg + (size_t)i * 4);
    float c = ctr[i];
    s += d[0] * r[0] + d[1] * r[1] + d[2] * r[2] + d[3] * r[3];
    f32x4 o0 = d * es;
    f32x4 o1 = {dctr[i] * gscale * c * (1.f - c), 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(dy8 + (size_t)i * 8) = o0;
    *reinterpret_cast<f32x4*>(dy8 + (size_t)i * 8 + 4) = o1;
  }
  __shared__ float red[4];
  float w = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// out[idx] (+)= sum(partial[0..n))
__global__ void sum_to_scalar_kernel(const float* __restrict__ partial, int n, float* __restrict__ out, int idx,
                                     int accumulate) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += (double)partial[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[idx] = accumulate ? out[idx] + (float)s : (float)s;
}

// ------------------------------------------------------------------------------------------ host
static void fill_anchor_args(AnchorArgs& a, int nlev, const int* fw, const int* fh, const float* strides, int A) {
  a.nlev = nlev;
  a.A = A;
  int off = 0;
  for (int l = 0; l < nlev; ++l) {
    a.fw[l] = fw[l];
    a.fh[l] = fh[l];
    a.stride[l] = strides[l];
    a.off[l] = off;
    off += fw[l] * fh[l] * A;
  }
  a.off[nlev] = off;
}

SACV_API_BEGIN

// anchors_out: [sum_l fh*fw*A, 4] fp32 (A = nratios*nscales), order ((y*W + x)*A + a), a = r*S + s.
int sacv_retina_anchors(int nlev, const int* fw, const int* fh, const float* areas /*[nlev][2]*/,
                        const float* ratios, int nratios, const float* scales, int nscales, const float* strides,
                        float* anchors_out, void* stream) {
  SACV_CHECK_ARG(nlev >= 1 && nlev <= MAXL, "retina_anchors: nlev");
  SACV_CHECK_ARG(nratios * nscales >= 1 && nratios * nscales <= MAXA, "retina_anchors: too many anchors per cell");
  SACV_CHECK_ARG(anchors_out, "retina_anchors: null output");
  AnchorArgs a;
  int A = nratios * nscales;
  fill_anchor_args(a, nlev, fw, fh, strides, A);
  for (int l = 0; l < nlev; ++l) {
    int row = 0;
    for (int r = 0; r < nratios; ++r) {
      volatile float sq_w = (float)sqrt((double)ratios[r]);
      volatile float inv = 1.0f / ratios[r];
      volatile float sq_h = (float)sqrt((double)inv);
      for (int s = 0; s < nscales; ++s) {
        volatile float aw = scales[s] * sq_w;
        volatile float ah = scales[s] * sq_h;
        volatile float w = areas[l * 2 + 0] * aw;
        volatile float h = areas[l * 2 + 1] * ah;
        float hw = w / 2.0f, hh = h / 2.0f;
        a.base[l][row][0] = -hw;
        a.base[l][row][1] = -hh;
        a.base[l][row][2] = hw;
        a.base[l][row][3] = hh;
        ++row;
      }
    }
  }
  int total = a.off[nlev];
  hipLaunchKernelGGL(retina_anchors_kernel, dim3(sacv_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, a,
                     anchors_out);
  SACV_CHECK_LAUNCH("retina_anchors");
  return SACV_OK;
}

int sacv_fcos_points(int nlev, const int* fw, const int* fh, const float* strides, float* points_out, void* stream) {
  SACV_CHECK_ARG(nlev >= 1 && nlev <= MAXL && points_out, "fcos_points: bad args");
  AnchorArgs a;
  fill_anchor_args(a, nlev, fw, fh, strides, 1);
  int total = a.off[nlev];
  hipLaunchKernelGGL(fcos_points_kernel, dim3(sacv_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, a,
                     points_out);
  SACV_CHECK_LAUNCH("fcos_points");
  return SACV_OK;
}

// counts: int[2] device buffer, zeroed here.
int sacv_retina_assign(const float* anchors, const float* annots, float* targets, int* counts, int B, int A, int G,
                       int smooth_l1, void* stream) {
  SACV_CHECK_ARG(anchors && annots && targets && counts, "retina_assign: null pointer");
  SACV_CHECK_ARG(G >= 0 && G <= MAXG, "retina_assign: at most %d annotation rows per image (got %d)", MAXG, G);
  SACV_CHECK_ARG(B > 0 && A > 0, "retina_assign: empty");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(counts, 0, 2 * sizeof(int), st) != hipSuccess) {
    sacv_set_error("retina_assign: memset failed");
    return SACV_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(retina_assign_kernel, dim3(sacv_cdiv(A, 256), B), dim3(256), 0, st, anchors, annots, targets,
                     counts, A, G, smooth_l1);
  SACV_CHECK_LAUNCH("retina_assign");
  return SACV_OK;
}

#define FOCAL_BLOCKS 2048

long long sacv_loss_workspace_floats(void) { return FOCAL_BLOCKS; }

// cls_in[l]: [B][n_l][C] probabilities (n_l = level_off[l+1]-level_off[l]); grad[l] same shape or null.
// targets: [B][A][tstride] with the label at [4].  loss_out[slot] receives weight*sum/num_pos.
int sacv_focal_loss_fwd_bwd(int nlev, const float* const* cls_in, float* const* grad, const int* level_off,
                            const float* targets, int tstride, int B, int A, int C, float alpha, float gamma,
                            float weight, const int* counts, float* workspace, float* loss_out, int slot,
                            void* stream) {
  SACV_CHECK_ARG(nlev >= 1 && nlev <= MAXL, "focal: nlev");
  SACV_CHECK_ARG(C >= 1, "focal: num_classes");
  SACV_CHECK_ARG(level_off[nlev] == A, "focal: level offsets do not sum to A");
  SACV_CHECK_ARG(targets && counts && workspace && loss_out, "focal: null pointer");
  LevelPtrs lp;
  lp.nlev = nlev;
  for (int l = 0; l < nlev; ++l) {
    SACV_CHECK_ARG(cls_in[l], "focal: null level pointer");
    lp.in[l] = cls_in[l];
    lp.grad[l] = grad ? grad[l] : nullptr;
    lp.off[l] = level_off[l];
  }
  lp.off[nlev] = level_off[nlev];
  hipStream_t st = (hipStream_t)stream;
  size_t total = (size_t)B * A * ((C + 3) / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > FOCAL_BLOCKS) blocks = FOCAL_BLOCKS;
  hipLaunchKernelGGL(focal_fwd_bwd_kernel, dim3(blocks), dim3(256), 0, st, lp, targets, tstride, B, A, C, alpha,
                     gamma, weight, counts, workspace);
  SACV_CHECK_LAUNCH("focal_fwd_bwd");
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, workspace, blocks, counts, weight, loss_out,
                     slot);
  SACV_CHECK_LAUNCH("loss_finalize");
  return SACV_OK;
}

int sacv_retina_box_loss_fwd_bwd(int nlev, const float* const* reg_in, float* const* grad, const int* level_off,
                                 const float* targets, const float* anchors, int B, int A, int box_type, float beta,
                                 float weight, const int* counts, float* workspace, float* loss_out, int slot,
                                 void* stream) {
  SACV_CHECK_ARG(nlev >= 1 && nlev <= MAXL, "box_loss: nlev");
  SACV_CHECK_ARG(box_type >= 0 && box_type <= 5,
                 "box_loss: box_type %d unsupported (0 SmoothL1, 1 IoU, 2 GIoU, 3 DIoU, 4 CIoU, 5 EIoU)", box_type);
  SACV_CHECK_ARG(level_off[nlev] == A, "box_loss: level offsets do not sum to A");
  SACV_CHECK_ARG(targets && anchors && counts && workspace && loss_out, "box_loss: null pointer");
  RegPtrs rp;
  rp.nlev = nlev;
  for (int l = 0; l < nlev; ++l) {
    SACV_CHECK_ARG(reg_in[l], "box_loss: null level pointer");
    rp.in[l] = reg_in[l];
    rp.grad[l] = grad ? grad[l] : nullptr;
    rp.off[l] = level_off[l];
  }
  rp.off[nlev] = level_off[nlev];
  hipStream_t st = (hipStream_t)stream;
  size_t total = (size_t)B * A;
  int blocks = (int)((total + 255) / 256);
  if (blocks > FOCAL_BLOCKS) blocks = FOCAL_BLOCKS;
  hipLaunchKernelGGL(retina_box_loss_kernel, dim3(blocks), dim3(256), 0, st, rp, targets, anchors, B, A, box_type,
                     beta, weight, counts, workspace);
  SACV_CHECK_LAUNCH("retina_box_loss");
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, workspace, blocks, counts, weight, loss_out,
                     slot);
  SACV_CHECK_LAUNCH("loss_finalize");
  return SACV_OK;
}

// FCOSLoss.get_batch_position_annotations: targets [B][P][8]; level tables are host arrays.
int sacv_fcos_assign(int nlev, const int* level_off, const float* strides, const float* mi /*[nlev][2]*/,
                     const float* points, const float* annots, float* targets, int* counts, int B, int P, int G,
                     float center_sample_radius, int use_center_sample, void* stream) {
  SACV_CHECK_ARG(nlev >= 1 && nlev <= MAXL, "fcos_assign: nlev");
  SACV_CHECK_ARG(points && annots && targets && counts, "fcos_assign: null pointer");
  SACV_CHECK_ARG(G >= 0 && G <= MAXG, "fcos_assign: at most %d annotation rows per image (got %d)", MAXG, G);
  SACV_CHECK_ARG(level_off[nlev] == P, "fcos_assign: level offsets do not sum to P");
  FcosLevels lv;
  lv.nlev = nlev;
  for (int l = 0; l < nlev; ++l) {
    lv.off[l] = level_off[l];
    lv.stride[l] = strides[l];
    lv.mi_lo[l] = mi[l * 2];
    lv.mi_hi[l] = mi[l * 2 + 1];
  }
  lv.off[nlev] = level_off[nlev];
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(counts, 0, 2 * sizeof(int), st) != hipSuccess) {
    sacv_set_error("fcos_assign: memset failed");
    return SACV_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(fcos_assign_kernel, dim3(sacv_cdiv(P, 256), B), dim3(256), 0, st, lv, points, annots, targets,
                     counts, P, G, center_sample_radius, use_center_sample);
  SACV_CHECK_LAUNCH("fcos_assign");
  return SACV_OK;
}

// IoU (type 0) / GIoU (type 1) regression loss weighted by the centerness target + centerness BCE.
// loss_out[slot_reg], loss_out[slot_ctr]; gradients w.r.t. the (scaled) regression outputs and centerness probs.
int sacv_fcos_reg_ctr_loss_fwd_bwd(int nlev, const float* const* reg_in, const float* const* ctr_in,
                                   float* const* dreg, float* const* dctr, const int* level_off,
                                   const float* targets, int B, int P, int iou_type, float w_reg, float w_ctr,
                                   const int* counts, float* workspace, float* loss_out, int slot_reg, int slot_ctr,
                                   void* stream) {
  SACV_CHECK_ARG(nlev >= 1 && nlev <= MAXL, "fcos_loss: nlev");
  SACV_CHECK_ARG(iou_type >= 0 && iou_type <= 4,
                 "fcos_loss: iou_type %d unsupported (0 IoU, 1 GIoU, 2 DIoU, 3 CIoU, 4 EIoU)", iou_type);
  SACV_CHECK_ARG(level_off[nlev] == P && targets && counts && workspace && loss_out, "fcos_loss: bad args");
  FcosPtrs fp;
  fp.nlev = nlev;
  for (int l = 0; l < nlev; ++l) {
    SACV_CHECK_ARG(reg_in[l] && ctr_in[l], "fcos_loss: null level pointer");
    fp.reg[l] = reg_in[l];
    fp.ctr[l] = ctr_in[l];
    fp.dreg[l] = dreg ? dreg[l] : nullptr;
    fp.dctr[l] = dctr ? dctr[l] : nullptr;
    fp.off[l] = level_off[l];
  }
  fp.off[nlev] = level_off[nlev];
  hipStream_t st = (hipStream_t)stream;
  size_t total = (size_t)B * P;
  int blocks = (int)((total + 255) / 256);
  if (blocks > FOCAL_BLOCKS / 2) blocks = FOCAL_BLOCKS / 2;
  hipLaunchKernelGGL(fcos_reg_ctr_loss_kernel, dim3(blocks), dim3(256), 0, st, fp, targets, B, P, iou_type, w_reg,
                     w_ctr, counts, workspace, blocks);
  SACV_CHECK_LAUNCH("fcos_reg_ctr_loss");
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, workspace, blocks, counts, w_reg, loss_out,
                     slot_reg);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, workspace + blocks, blocks, counts, w_ctr,
                     loss_out, slot_ctr);
  SACV_CHECK_LAUNCH("loss_finalize");
  return SACV_OK;
}

int sacv_fcos_split_fwd(const float* y8, const float* scale, float* reg, float* ctr, int R, void* stream) {
  SACV_CHECK_ARG(y8 && scale && reg && ctr && R > 0, "fcos_split_fwd: bad args");
  int blocks = sacv_cdiv(R, 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(fcos_split_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y8, scale, reg, ctr, R);
  SACV_CHECK_LAUNCH("fcos_split_fwd");
  return SACV_OK;
}

// dscale[idx] (+)= sum dreg*reg; workspace >= 1024 floats
int sacv_fcos_split_bwd(const float* dreg, const float* dctr, const float* reg, const float* ctr, const float* scale,
                        float gscale, float* dy8, float* dscale, int idx, int accumulate, int R, float* workspace,
                        void* stream) {
  SACV_CHECK_ARG(dreg && dctr && reg && ctr && scale && dy8 && dscale && workspace && R > 0, "fcos_split_bwd: bad args");
  int blocks = sacv_cdiv(R, 256);
  if (blocks > 1024) blocks = 1024;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(fcos_split_bwd_kernel, dim3(blocks), dim3(256), 0, st, dreg, dctr, reg, ctr, scale, gscale, dy8,
                     workspace, R);
  SACV_CHECK_LAUNCH("fcos_split_bwd");
  hipLaunchKernelGGL(sum_to_scalar_kernel, dim3(1), dim3(64), 0, st, workspace, blocks, dscale, idx, accumulate);
  SACV_CHECK_LAUNCH("sum_to_scalar");
  return SACV_OK;
}

SACV_API_END
