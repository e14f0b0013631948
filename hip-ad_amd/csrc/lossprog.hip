// hip-ad_amd/csrc/lossprog.hip -- the decoder's training objective as a handful of launches.
//
// Replaces: SparseOneDecoder.loss (reference models/sparse_onedecoder.py:1094-1579) with its samplers
//   det/target.py:66-162 (SparseBox3DTarget), map/target.py:38-62 + 105-160 + map/match_cost.py (HungarianLinesAssigner),
//   motion/target.py:5-35 + 71-99, plan/target.py:7-37 + 80-162
// and loss modules det/losses.py:11-93 (SparseBox3DLoss), map/loss.py:10-120 (SparseLineLoss / LinesL1Loss) plus
// mmdet==2.28.2's FocalLoss / L1Loss / CrossEntropyLoss(use_sigmoid) / GaussianFocalLoss / FocalLossCost formulas.
//
// The torch-op formulation of the same arithmetic (projects/mmdet3d_plugin/models/criterion.py, which stays as the
// CPU / reference-parity form) is ~750 launches per training step -- cost terms, gathers, scatters, masks, where()s and
// their autograd mirror images, each a one-workgroup kernel of a few microseconds.  Here every task is
//   cost matrix (1 launch) -> Hungarian assignment (assign.hip, 1) -> inverse map + positive counts (1)
//   -> loss values of all six decoder layers AND d(loss)/d(prediction) in one pass (1),
// the motion / planning / ego terms one launch each.  Layers are read through a table of per-layer pointers (the heads'
// outputs as they are: no stacking copy); gradients land in one caller-provided buffer per tensor kind, so the
// backward of the whole objective is a single scaling by the upstream gradient.
//
// Arithmetic follows criterion.py expression by expression (same association order where it can matter: NaN targets,
// the 1e8 substitution, the smooth-L1 switch, first-minimum tie rules); sums over rows are block reductions + one
// fp32 atomic per workgroup and term.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "hipad.h"
#include "daf_common.h"

namespace hipad {

constexpr int kMaxGt = 64;       // padded ground-truth items per sample the LDS tables hold
constexpr int kBoxDims = 10;     // encoded box: x y z log(w) log(l) log(h) sin cos vx vy
constexpr float kReduceEps = 1.1920929e-07f;   // torch.finfo(float32).eps: mmdet weight_reduce_loss

__device__ __forceinline__ const float *layer_ptr(const hipad_layer_ptrs &t, int l) {
  const float *p = t.p[0];
#pragma unroll
  for (int i = 1; i < HIPAD_LOSS_MAX_LAYERS; ++i) p = (l == i) ? t.p[i] : p;
  return p;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float powg(float v, float gamma) { return gamma == 2.f ? v * v : powf(v, gamma); }

// mmdet FocalLossCost / det/target.py:126-149: cost of giving class `label` to logit x
__device__ __forceinline__ float focal_cost(float x, float alpha, float gamma, float eps) {
  const float p = sigmoidf_(x);
  const float neg = -logf(1.f - p + eps) * (1.f - alpha) * powg(p, gamma);
  const float pos = -logf(p + eps) * alpha * powg(1.f - p, gamma);
  return pos - neg;
}

// sigmoid focal loss of one logit against a 0/1 target: value and d/dx (losses.hip, same expressions)
__device__ __forceinline__ void focal_term(float xv, bool t, float alpha, float gamma, float &l, float &g) {
  const float p = sigmoidf_(xv);
  const float bce = fmaxf(xv, 0.f) - (t ? xv : 0.f) + log1pf(expf(-fabsf(xv)));
  if (t) {
    const float q = powf(1.f - p, gamma);
    l = alpha * q * bce;
    g = alpha * q * (gamma * p * logf(fmaxf(p, 1e-38f)) - (1.f - p));
  } else {
    const float q = powf(p, gamma);
    l = (1.f - alpha) * q * bce;
    g = (1.f - alpha) * q * (gamma * (1.f - p) * bce + p);
  }
}

__device__ __forceinline__ float sgn(float v) { return (float)((v > 0.f) - (v < 0.f)); }

// sum `v` over the workgroup (<= 256 threads, whole waves) and add it to *dst with one atomic
__device__ __forceinline__ void block_add(float v, float *dst, float *sh /* [4] */) {
  v = wave_sum(v);
  const int wv = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[wv] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < (int)((blockDim.x + 63) >> 6); ++i) s += sh[i];
    if (s != 0.f) atomicAdd(dst, s);
  }
}

// encoded regression target of a ground-truth box (det/target.py:49-64) and its weight row (NaN-aware, class-wise)
__device__ __forceinline__ float encode_box(const float *bx, int d) {
  if (d < 3) return bx[d];
  if (d < 6) return logf(bx[d]);
  if (d == 6) return sinf(bx[6]);
  if (d == 7) return cosf(bx[6]);
  return bx[d - 1];
}

__device__ __forceinline__ float box_weight_of(float target, long long label, int d, const hipad_det_loss_cfg &c) {
  float w = isnan(target) ? 0.f : 1.f;
#pragma unroll
  for (int k = 0; k < HIPAD_LOSS_MAX_CLSWISE; ++k)
    if (k < c.num_cls_wise && label == (long long)c.cls_wise_label[k]) w = c.cls_wise_weights[k][d];
  return w;
}

// =========================================================================================================
// DET 1: cost[lb][g][p] (ground-truth major, as hipad_linear_assignment takes it) + rows per problem
// =========================================================================================================
__global__ __launch_bounds__(256) void det_cost_kernel(float *__restrict__ cost, int *__restrict__ n_rows,
                                                       const hipad_layer_ptrs cls, const hipad_layer_ptrs box,
                                                       const float *__restrict__ gt_boxes, const long long *__restrict__ labels,
                                                       const int *__restrict__ count, const hipad_det_loss_cfg cfg, int bs,
                                                       int P, int C, int D, int G, int gt_dim) {
  __shared__ float tgt[kMaxGt][kBoxDims], wgt[kMaxGt][kBoxDims];
  __shared__ int lab[kMaxGt];
  const int lb = blockIdx.y, l = lb / bs, b = lb - l * bs;
  const int n = min(count[b], G);
  for (int i = threadIdx.x; i < G * kBoxDims; i += blockDim.x) {
    const int g = i / kBoxDims, d = i - g * kBoxDims;
    const float t = encode_box(gt_boxes + ((size_t)b * G + g) * gt_dim, d);
    tgt[g][d] = t;
    wgt[g][d] = box_weight_of(t, labels[(size_t)b * G + g], d, cfg);
  }
  for (int g = threadIdx.x; g < G; g += blockDim.x) lab[g] = (int)labels[(size_t)b * G + g];
  if (blockIdx.x == 0 && threadIdx.x == 0) n_rows[lb] = n;
  __syncthreads();
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float *cp = layer_ptr(cls, l) + ((size_t)b * P + p) * C;
  const float *bp = layer_ptr(box, l) + ((size_t)b * P + p) * D;
  float bv[kBoxDims];
#pragma unroll
  for (int d = 0; d < kBoxDims; ++d) bv[d] = bp[d];
  float *out = cost + (size_t)lb * G * P + p;
  for (int g = 0; g < G; ++g) {
    float c = 0.f;
    if (g < n) {
      const int lbl = min(max(lab[g], 0), C - 1);
      const float cls_cost = focal_cost(cp[lbl], cfg.cost_alpha, cfg.cost_gamma, cfg.cost_eps) * cfg.cost_cls_weight;
      float bsum = 0.f;
#pragma unroll
      for (int d = 0; d < kBoxDims; ++d) bsum += fabsf(bv[d] - tgt[g][d]) * wgt[g][d] * cfg.cost_reg_weights[d];
      c = cls_cost + bsum * cfg.cost_box_weight;
      if (isnan(c) || (isinf(c) && c < 0.f)) c = 1e8f;
    }
    out[(size_t)g * P] = c;
  }
}

// =========================================================================================================
// DET / MAP 2: inverse map prediction -> matched ground-truth item, positive count per layer
//   matched[lb][p] = g or -1;  count_out[0][l] = number of matched predictions whose target row is not all zero
//   (criterion.py: matched = not all(reg_target == 0); the flag comes from the caller-specific functor),
//   count_out[1][l] = number of assigned ground-truth items
// =========================================================================================================
template <typename NzFn>
__device__ __forceinline__ void finish_body(int *__restrict__ matched, float *__restrict__ count_out /* [2][L] */,
                                            const int *__restrict__ index, const int *__restrict__ count, int bs, int P,
                                            int G, NzFn nz) {
  __shared__ float sh[4];
  __shared__ float total, total_raw;
  const int l = blockIdx.x;
  for (int i = threadIdx.x; i < bs * P; i += blockDim.x) matched[(size_t)l * bs * P + i] = -1;
  if (threadIdx.x == 0) total = total_raw = 0.f;
  __syncthreads();
  float mine = 0.f, raw = 0.f;
  for (int i = threadIdx.x; i < bs * G; i += blockDim.x) {
    const int b = i / G, g = i - b * G;
    if (g < min(count[b], G)) {
      const int idx = index[((size_t)l * bs + b) * G + g];
      if (idx >= 0 && idx < P) {
        matched[((size_t)l * bs + b) * P + idx] = g;
        raw += 1.f;                                   // every assigned item (the motion head's positive count)
        if (nz(l * bs + b, b, g, idx)) mine += 1.f;   // items whose target row is not all zero (the task's own)
      }
    }
  }
  block_add(mine, &total, sh);
  block_add(raw, &total_raw, sh);
  __syncthreads();
  if (threadIdx.x == 0) {
    count_out[l] = total;
    count_out[gridDim.x + l] = total_raw;
  }
}

__global__ __launch_bounds__(256) void det_finish_kernel(int *__restrict__ matched, float *__restrict__ count_out,
                                                         const int *__restrict__ index, const int *__restrict__ count,
                                                         const float *__restrict__ gt_boxes, int bs, int P, int G, int gt_dim) {
  finish_body(matched, count_out, index, count, bs, P, G, [&](int, int b, int g, int) {
    bool any = false;
    for (int d = 0; d < kBoxDims; ++d) {
      const float t = encode_box(gt_boxes + ((size_t)b * G + g) * gt_dim, d);
      any = any || !(t == 0.f);     // NaN != 0 counts, as in torch
    }
    return any;
  });
}

// =========================================================================================================
// DET 3: focal class loss over all predictions; box L1 + centerness + yawness over the matched, class-gated ones.
//   terms[0..3][l] += cls / box / cns / yns;  gradients written for every element of the three tensors.
// =========================================================================================================
__global__ __launch_bounds__(256) void det_loss_kernel(float *__restrict__ terms /* [4][L] */, float *__restrict__ g_cls,
                                                       float *__restrict__ g_box, float *__restrict__ g_box_cns /* [..][3] */,
                                                       float *__restrict__ g_qt,
                                                       const hipad_layer_ptrs cls, const hipad_layer_ptrs box,
                                                       const hipad_layer_ptrs qt, const int *__restrict__ matched,
                                                       const float *__restrict__ num_pos, const float *__restrict__ gt_boxes,
                                                       const long long *__restrict__ labels, const hipad_det_loss_cfg cfg,
                                                       int L, int bs, int P, int C, int D, int Q, int G, int gt_dim) {
  __shared__ float sh[4];
  const int l = blockIdx.y;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;   // (b, p)
  float s_cls = 0.f, s_box = 0.f, s_cns = 0.f, s_yns = 0.f;
  if (r < bs * P) {
    const int b = r / P;
    const size_t row = (size_t)l * bs * P + r;
    const float avg = fmaxf(num_pos[l], 1.f) + kReduceEps;
    const float *cp = layer_ptr(cls, l) + (size_t)r * C;
    const float *bp = layer_ptr(box, l) + (size_t)r * D;
    const int g = matched[row];
    const int tcls = g >= 0 ? (int)labels[(size_t)b * G + g] : C;
    float xmax = -INFINITY;
    const float sc = cfg.w_cls / avg;
    for (int c = 0; c < C; ++c) {
      const float xv = cp[c];
      xmax = fmaxf(xmax, xv);
      float lv, gv;
      focal_term(xv, tcls == c, cfg.focal_alpha, cfg.focal_gamma, lv, gv);
      s_cls += lv * sc;
      g_cls[row * C + c] = gv * sc;
    }
    float gb[kBoxDims];
#pragma unroll
    for (int d = 0; d < kBoxDims; ++d) gb[d] = 0.f;
    float gq_cns = 0.f, gq_yns = 0.f;
    float gc3[3] = {0.f, 0.f, 0.f};   // d(centerness term) / d(box centre): kept apart so each term can be scaled on its own
    if (g >= 0) {
      const float *bx = gt_boxes + ((size_t)b * G + g) * gt_dim;
      const long long lbl = labels[(size_t)b * G + g];
      float tg[kBoxDims];
      bool nzero = false;
#pragma unroll
      for (int d = 0; d < kBoxDims; ++d) {
        tg[d] = encode_box(bx, d);
        nzero = nzero || !(tg[d] == 0.f);
      }
      const bool gate = cfg.cls_threshold > 0.f ? (sigmoidf_(xmax) > cfg.cls_threshold) : true;
      if (nzero && gate) {
        float bv[kBoxDims];
#pragma unroll
        for (int d = 0; d < kBoxDims; ++d) {
          bv[d] = bp[d];
          const float w = box_weight_of(tg[d], lbl, d, cfg) * cfg.loss_reg_weights[d];
          tg[d] = isnan(tg[d]) ? 0.f : tg[d];
          const float diff = bv[d] - tg[d];
          s_box += fabsf(diff) * w * (cfg.w_box / avg);
          gb[d] = sgn(diff) * w * (cfg.w_box / avg);
        }
        if (Q > 0) {
          const float *qp = layer_ptr(qt, l) + (size_t)r * Q;
          // centerness: BCE with logits against exp(-|centre error|); the target carries gradient into the box centre
          const float x = qp[cfg.cns_index];
          const float dx0 = tg[0] - bv[0], dx1 = tg[1] - bv[1], dx2 = tg[2] - bv[2];
          const float nrm = sqrtf(dx0 * dx0 + dx1 * dx1 + dx2 * dx2);
          const float t = expf(-nrm);
          const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
          const float sq = cfg.w_cns / avg;
          s_cns += bce * sq;
          gq_cns = (sigmoidf_(x) - t) * sq;
          if (nrm > 0.f) {     // d bce / d t = -x;  d t / d box_i = t * (tg_i - box_i) / nrm
            const float k = -x * t / nrm * sq;
            gc3[0] = k * dx0;
            gc3[1] = k * dx1;
            gc3[2] = k * dx2;
          }
          // yawness: Gaussian focal loss of sigmoid(logit) against [cos(target yaw, predicted yaw) > 0]
          const float y = qp[cfg.yns_index];
          const float p = sigmoidf_(y);
          const bool pos = (tg[6] * bv[6] + tg[7] * bv[7]) > 0.f;
          const float e = 1e-12f;
          float lv, dp;
          if (pos) {
            lv = -logf(p + e) * powg(1.f - p, cfg.gauss_alpha);
            dp = -powg(1.f - p, cfg.gauss_alpha) / (p + e) + cfg.gauss_alpha * powf(1.f - p, cfg.gauss_alpha - 1.f) * logf(p + e);
          } else {
            lv = -logf(1.f - p + e) * powg(p, cfg.gauss_alpha);
            dp = powg(p, cfg.gauss_alpha) / (1.f - p + e) - cfg.gauss_alpha * powf(p, cfg.gauss_alpha - 1.f) * logf(1.f - p + e);
          }
          const float sy = cfg.w_yns / avg;
          s_yns += lv * sy;
          gq_yns = dp * p * (1.f - p) * sy;
        }
      }
    }
    for (int d = 0; d < D; ++d) g_box[row * D + d] = d < kBoxDims ? gb[d] : 0.f;
    for (int d = 0; d < 3; ++d) g_box_cns[row * 3 + d] = gc3[d];
    for (int q = 0; q < Q; ++q) g_qt[row * Q + q] = q == cfg.cns_index ? gq_cns : (q == cfg.yns_index ? gq_yns : 0.f);
  }
  block_add(s_cls, terms + 0 * L + l, sh);
  block_add(s_box, terms + 1 * L + l, sh);
  block_add(s_cns, terms + 2 * L + l, sh);
  block_add(s_yns, terms + 3 * L + l, sh);
}

// =========================================================================================================
// MAP 1: focal class cost + permutation-invariant smooth-L1 line cost (map/match_cost.py:36-56): cost and the
//        best point order of every (prediction, ground-truth line) pair
// =========================================================================================================
__device__ __forceinline__ float smooth_l1(float d, float beta) {
  return beta > 0.f ? (d < beta ? 0.5f * d * d / beta : d - 0.5f * beta) : d;
}
__device__ __forceinline__ float norm_coord(float v, int k, const hipad_map_loss_cfg &c) {   // k even: x, odd: y
  return (k & 1) ? (v - c.origin_y) / c.norm_y : (v - c.origin_x) / c.norm_x;
}

constexpr int kLineDims = 40;   // 20 points x (x, y)
constexpr int kMaxPermute = 64;  // point orders of a ground-truth line held in LDS (more: read from memory)

// One workgroup per (layer-sample, ground-truth line): the line's roi-normalised point orders are prepared once in LDS
// (the divisions), a thread then owns one prediction and walks the orders.  (First version: one thread per prediction
// over ALL lines and orders -- 600 threads on the whole chip, 1.6 ms.)
__global__ __launch_bounds__(128) void map_cost_kernel(float *__restrict__ cost, unsigned char *__restrict__ perm,
                                                       int *__restrict__ n_rows, const hipad_layer_ptrs cls,
                                                       const hipad_layer_ptrs pts, const float *__restrict__ gt_pts,
                                                       const long long *__restrict__ labels, const int *__restrict__ count,
                                                       const hipad_map_loss_cfg cfg, int bs, int P, int C, int G, int NP) {
  __shared__ float gn[kMaxPermute][kLineDims];
  const int lb = blockIdx.y, l = lb / bs, b = lb - l * bs, g = blockIdx.x;
  const int n = min(count[b], G);
  if (g == 0 && threadIdx.x == 0) n_rows[lb] = n;
  const bool live = g < n;
  const float *gp = gt_pts + ((size_t)b * G + g) * NP * kLineDims;
  const int nlds = min(NP, kMaxPermute);
  if (live)
    for (int i = threadIdx.x; i < nlds * kLineDims; i += blockDim.x) gn[i / kLineDims][i % kLineDims] = norm_coord(gp[i], i % kLineDims, cfg);
  __syncthreads();
  const int lbl = live ? min(max((int)labels[(size_t)b * G + g], 0), C - 1) : 0;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    float c = 0.f;
    int best = 0;
    if (live) {
      const float *cp = layer_ptr(cls, l) + ((size_t)b * P + p) * C;
      const float *pp = layer_ptr(pts, l) + ((size_t)b * P + p) * kLineDims;
      float pn[kLineDims];
#pragma unroll
      for (int k = 0; k < kLineDims; ++k) pn[k] = norm_coord(pp[k], k, cfg);
      const float cls_cost = focal_cost(cp[lbl], 0.25f, 2.f, 1e-12f) * cfg.cost_cls_weight;
      float dmin = INFINITY;
      for (int q = 0; q < NP; ++q) {
        float s = 0.f;
        if (q < nlds) {
#pragma unroll
          for (int k = 0; k < kLineDims; ++k) s += smooth_l1(fabsf(pn[k] - gn[q][k]), cfg.cost_beta);
        } else {
          for (int k = 0; k < kLineDims; ++k) s += smooth_l1(fabsf(pn[k] - norm_coord(gp[q * kLineDims + k], k, cfg)), cfg.cost_beta);
        }
        s = s / (float)(kLineDims / 2);
        // first minimum wins; a NaN distance wins outright and stays (torch.min propagates NaN)
        if (q == 0 || s < dmin || (isnan(s) && !isnan(dmin))) { dmin = s; best = q; }
      }
      c = cls_cost + dmin * cfg.cost_reg_weight;
      if (isnan(c)) c = 0.f;                          // torch.nan_to_num
      else if (isinf(c)) c = c > 0.f ? 3.4028234663852886e38f : -3.4028234663852886e38f;
    }
    cost[((size_t)lb * G + g) * P + p] = c;
    perm[((size_t)lb * G + g) * P + p] = (unsigned char)best;
  }
}

__global__ __launch_bounds__(256) void map_finish_kernel(int *__restrict__ matched, float *__restrict__ count_out,
                                                         int *__restrict__ order /* [LB][G] */, const int *__restrict__ index,
                                                         const unsigned char *__restrict__ perm, const int *__restrict__ count,
                                                         const float *__restrict__ gt_pts, int bs, int P, int G, int NP) {
  finish_body(matched, count_out, index, count, bs, P, G, [&](int lb, int b, int g, int idx) {
    const int q = perm[((size_t)lb * G + g) * P + idx];
    order[(size_t)lb * G + g] = q;
    const float *gp = gt_pts + (((size_t)b * G + g) * NP + q) * kLineDims;
    bool any = false;
    for (int k = 0; k < kLineDims; ++k) any = any || !(gp[k] == 0.f);
    return any;
  });
}

// MAP 3: focal class loss over all predictions; smooth-L1 line loss on roi-normalised points over the matched, gated ones
__global__ __launch_bounds__(128) void map_loss_kernel(float *__restrict__ terms /* [2][L] */, float *__restrict__ g_cls,
                                                       float *__restrict__ g_pts, const hipad_layer_ptrs cls,
                                                       const hipad_layer_ptrs pts, const int *__restrict__ matched,
                                                       const int *__restrict__ order, const float *__restrict__ num_pos,
                                                       const float *__restrict__ gt_pts, const long long *__restrict__ labels,
                                                       const hipad_map_loss_cfg cfg, int L, int bs, int P, int C, int G, int NP) {
  __shared__ float sh[4];
  const int l = blockIdx.y;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  float s_cls = 0.f, s_line = 0.f;
  if (r < bs * P) {
    const int b = r / P;
    const size_t row = (size_t)l * bs * P + r;
    const float avg = fmaxf(num_pos[l], 1.f) + kReduceEps;
    const float *cp = layer_ptr(cls, l) + (size_t)r * C;
    const float *pp = layer_ptr(pts, l) + (size_t)r * kLineDims;
    const int g = matched[row];
    const int tcls = g >= 0 ? (int)labels[(size_t)b * G + g] : C;
    float xmax = -INFINITY;
    const float sc = cfg.w_cls / avg;
    for (int c = 0; c < C; ++c) {
      const float xv = cp[c];
      xmax = fmaxf(xmax, xv);
      float lv, gv;
      focal_term(xv, tcls == c, cfg.focal_alpha, cfg.focal_gamma, lv, gv);
      s_cls += lv * sc;
      g_cls[row * C + c] = gv * sc;
    }
    bool active = false;
    const float *gp = nullptr;
    if (g >= 0) {
      gp = gt_pts + (((size_t)b * G + g) * NP + order[((size_t)l * bs + b) * G + g]) * kLineDims;
      bool nzero = false;
      for (int k = 0; k < kLineDims; ++k) nzero = nzero || !(gp[k] == 0.f);
      const bool gate = cfg.cls_threshold > 0.f ? (sigmoidf_(xmax) > cfg.cls_threshold) : true;
      active = nzero && gate;
    }
    const float sl = cfg.w_line / (float)(kLineDims / 2) / avg;
    for (int k = 0; k < kLineDims; ++k) {
      float gk = 0.f;
      if (active) {
        const float t = isnan(gp[k]) ? 0.f : gp[k];
        const float diff = norm_coord(pp[k], k, cfg) - norm_coord(t, k, cfg);
        const float a = fabsf(diff);
        const float w = cfg.reg_weights[k];
        s_line += smooth_l1(a, cfg.loss_beta) * w * sl;
        const float da = (cfg.loss_beta > 0.f && a < cfg.loss_beta) ? a / cfg.loss_beta : 1.f;
        gk = da * sgn(diff) * w * sl / ((k & 1) ? cfg.norm_y : cfg.norm_x);
      }
      g_pts[row * kLineDims + k] = gk;
    }
  }
  block_add(s_cls, terms + 0 * L + l, sh);
  block_add(s_line, terms + 1 * L + l, sh);
}

// =========================================================================================================
// MOTION: targets scattered by the LAST layer's box matching (sparse_onedecoder.py:1287), winner-take-all mode,
//         focal loss over the modes + L1 on the cumulative trajectory of the winning mode
// =========================================================================================================
constexpr int kMaxTs = 12, kMaxModes = 8;

__global__ __launch_bounds__(256) void motion_loss_kernel(float *__restrict__ terms /* [2][L] */, float *__restrict__ g_cls,
                                                          float *__restrict__ g_reg, const hipad_layer_ptrs cls,
                                                          const hipad_layer_ptrs reg, const int *__restrict__ matched_last,
                                                          const float *__restrict__ num_pos, int num_pos_stride,
                                                          const float *__restrict__ trajs, const float *__restrict__ masks,
                                                          const hipad_motion_loss_cfg cfg, int L, int bs, int A, int M, int T,
                                                          int G) {
  __shared__ float sh[4];
  const int l = blockIdx.y;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;   // (b, a)
  float s_cls = 0.f, s_reg = 0.f;
  if (r < bs * A) {
    const int b = r / A;
    const size_t row = (size_t)l * bs * A + r;
    const float avg = fmaxf(num_pos[(size_t)l * num_pos_stride], 1.f) + kReduceEps;
    const float *cp = layer_ptr(cls, l) + (size_t)r * M;
    const float *rp = layer_ptr(reg, l) + (size_t)r * M * T * 2;
    const int g = matched_last[(size_t)(L - 1) * bs * A + r];   // the last decoder layer's matching serves every layer
    float tx[kMaxTs], ty[kMaxTs], wt[kMaxTs];
    bool any = false;
    for (int t = 0; t < T; ++t) {
      tx[t] = g >= 0 ? trajs[(((size_t)b * G + g) * T + t) * 2 + 0] : 0.f;
      ty[t] = g >= 0 ? trajs[(((size_t)b * G + g) * T + t) * 2 + 1] : 0.f;
      wt[t] = g >= 0 ? masks[((size_t)b * G + g) * T + t] : 0.f;
      any = any || (wt[t] != 0.f);
    }
    // winner-take-all: mean over time of |cumsum(target) - cumsum(pred_m)| * weight (first minimum)
    int best = 0;
    float dbest = INFINITY;
    for (int m = 0; m < M; ++m) {
      float cx = 0.f, cy = 0.f, px = 0.f, py = 0.f, s = 0.f;
      for (int t = 0; t < T; ++t) {
        cx += tx[t]; cy += ty[t];
        px += rp[(m * T + t) * 2 + 0]; py += rp[(m * T + t) * 2 + 1];
        const float ex = cx - px, ey = cy - py;
        s += sqrtf(ex * ex + ey * ey) * wt[t];
      }
      s = s / (float)T;
      if (m == 0 || s < dbest) { dbest = s; best = m; }
    }
    // focal over the modes, row weight = any(mask)
    const float rw = any ? 1.f : 0.f;
    const float sc = cfg.w_cls * rw / avg;
    for (int m = 0; m < M; ++m) {
      float lv, gv;
      focal_term(cp[m], m == best, cfg.focal_alpha, cfg.focal_gamma, lv, gv);
      s_cls += lv * sc;
      g_cls[row * M + m] = gv * sc;
    }
    // L1 on cumulative way-points of the winning mode; d/d(offset_t) = sum over t' >= t of the way-point gradients
    float gx[kMaxTs], gy[kMaxTs];
    {
      float cx = 0.f, cy = 0.f, px = 0.f, py = 0.f;
      const float sr = cfg.w_reg / avg;
      for (int t = 0; t < T; ++t) {
        cx += tx[t]; cy += ty[t];
        px += rp[(best * T + t) * 2 + 0]; py += rp[(best * T + t) * 2 + 1];
        const float ex = px - cx, ey = py - cy;
        s_reg += (fabsf(ex) + fabsf(ey)) * wt[t] * sr;
        gx[t] = sgn(ex) * wt[t] * sr;
        gy[t] = sgn(ey) * wt[t] * sr;
      }
      for (int t = T - 2; t >= 0; --t) { gx[t] += gx[t + 1]; gy[t] += gy[t + 1]; }
    }
    float *go = g_reg + row * M * T * 2;
    for (int m = 0; m < M; ++m)
      for (int t = 0; t < T; ++t) {
        go[(m * T + t) * 2 + 0] = m == best ? gx[t] : 0.f;
        go[(m * T + t) * 2 + 1] = m == best ? gy[t] : 0.f;
      }
  }
  block_add(s_cls, terms + 0 * L + l, sh);
  block_add(s_reg, terms + 1 * L + l, sh);
}

// =========================================================================================================
// PLAN (+ ego status): one workgroup per (layer, sample).  Single driving command, anchor groups of kind
// temp / spat (aligned to the reference group's winning mode) and speed (buckets of an interval; class = the bucket of
// the ground-truth average speed) -- criterion.py::_loss_plan_batched, reference sparse_onedecoder.py:1315-1443.
//   terms[kind 0..2][cls 0 / reg 1][l], kind: 0 temp, 1 spat, 2 speed;  terms[6][l] = ego status
// =========================================================================================================
__global__ __launch_bounds__(64) void plan_loss_kernel(float *__restrict__ terms /* [7][L] */, float *__restrict__ g_cls,
                                                       float *__restrict__ g_reg, float *__restrict__ g_status,
                                                       const hipad_layer_ptrs cls, const hipad_layer_ptrs reg,
                                                       const hipad_layer_ptrs status, const hipad_plan_loss_cfg cfg, int L,
                                                       int bs, int NG, int M, int T, int S) {
  __shared__ float dist[64];
  __shared__ int best_s;
  const int lb = blockIdx.x, l = lb / bs, b = lb - l * bs;
  const int tid = threadIdx.x;
  const float *cp = layer_ptr(cls, l) + (size_t)b * NG * M;
  const float *rp = layer_ptr(reg, l) + (size_t)b * NG * M * T * 2;
  float *gc = g_cls + (size_t)lb * NG * M;
  float *gr = g_reg + (size_t)lb * NG * M * T * 2;
  for (int i = tid; i < NG * M; i += 64) gc[i] = 0.f;
  for (int i = tid; i < NG * M * T * 2; i += 64) gr[i] = 0.f;
  // ---- winning mode of the reference group
  {
    const float *gt = cfg.gt_traj[cfg.ref_group] + (size_t)b * T * 2, *gm = cfg.gt_mask[cfg.ref_group] + (size_t)b * T;
    float s = INFINITY;
    if (tid < M) {
      float cx = 0.f, cy = 0.f, px = 0.f, py = 0.f;
      s = 0.f;
      const float *q = rp + ((size_t)cfg.ref_group * M + tid) * T * 2;
      for (int t = 0; t < T; ++t) {
        cx += gt[2 * t]; cy += gt[2 * t + 1];
        px += q[2 * t]; py += q[2 * t + 1];
        const float ex = cx - px, ey = cy - py;
        s += sqrtf(ex * ex + ey * ey) * gm[t];
      }
      s = s / (float)T;
    }
    dist[tid] = s;
    __syncthreads();
    if (tid == 0) {
      int bi = 0;
      for (int m = 1; m < M; ++m)
        if (dist[m] < dist[bi]) bi = m;
      best_s = bi;
    }
    __syncthreads();
  }
  const int ref = best_s;
  __syncthreads();   // everyone has read best_s / the zero fill of gc, gr above is ordered before the stores below
  bool ref_any = false;
  {
    const float *gm = cfg.gt_mask[cfg.ref_group] + (size_t)b * T;
    for (int t = 0; t < T; ++t) ref_any = ref_any || (gm[t] != 0.f);
  }
  float acc[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) acc[i] = 0.f;
  // ---- aligned groups: focal over the group's modes (target = reference mode), L1 of the reference mode's way-points
  for (int gi = 0; gi < NG; ++gi) {
    const int kind = cfg.kind[gi];
    if (kind > 1) continue;
    if (tid < M) {
      float lv, gv;
      focal_term(cp[gi * M + tid], tid == ref, cfg.focal_alpha, cfg.focal_gamma, lv, gv);
      const float sc = cfg.w_cls * (ref_any ? 1.f : 0.f) / (float)(bs * M);   // mean over the (sample, mode) elements
      acc[kind * 2] += lv * sc;
      gc[gi * M + tid] = gv * sc;
    }
    if (tid == 0) {
      const float *gt = cfg.gt_traj[gi] + (size_t)b * T * 2, *gm = cfg.gt_mask[gi] + (size_t)b * T;
      const float *q = rp + ((size_t)gi * M + ref) * T * 2;
      float *go = gr + ((size_t)gi * M + ref) * T * 2;
      float gx[kMaxTs], gy[kMaxTs];
      float cx = 0.f, cy = 0.f, px = 0.f, py = 0.f;
      const float sr = cfg.w_reg / (float)(bs * T * 2);                        // mean over (sample, step, xy)
      for (int t = 0; t < T; ++t) {
        cx += gt[2 * t]; cy += gt[2 * t + 1];
        px += q[2 * t]; py += q[2 * t + 1];
        const float ex = px - cx, ey = py - cy;
        acc[kind * 2 + 1] += (fabsf(ex) + fabsf(ey)) * gm[t] * sr;
        gx[t] = sgn(ex) * gm[t] * sr;
        gy[t] = sgn(ey) * gm[t] * sr;
      }
      for (int t = T - 2; t >= 0; --t) { gx[t] += gx[t + 1]; gy[t] += gy[t + 1]; }
      for (int t = 0; t < T; ++t) { go[2 * t] = gx[t]; go[2 * t + 1] = gy[t]; }
    }
  }
  // ---- speed intervals: the buckets' reference-mode logits are the classes; the ground-truth speed picks the bucket
  if (tid == 0 && cfg.num_intervals > 0) {
    const float *st = cfg.speed_traj + (size_t)b * T * 2, *sm = cfg.speed_mask + (size_t)b * T;
    float dsum = 0.f, msum = 0.f;
    bool sp_any = false;
    for (int t = 0; t < T; ++t) {
      dsum += sqrtf(st[2 * t] * st[2 * t] + st[2 * t + 1] * st[2 * t + 1]);
      msum += sm[t];
      sp_any = sp_any || (sm[t] != 0.f);
    }
    const float speed = dsum / (msum * cfg.speed_interval + 1e-4f);
    for (int iv = 0; iv < cfg.num_intervals; ++iv) {
      const int K = cfg.interval_size[iv];
      int bucket = 1;
      for (int k = 0; k < K; ++k)
        if (speed >= cfg.bucket_lo[iv][k] && speed < cfg.bucket_hi[iv][k]) bucket = k;
      // focal over the K bucket logits (each the reference-mode logit of its group), class = bucket
      for (int k = 0; k < K; ++k) {
        const int gi = cfg.interval_group[iv][k];
        float lv, gv;
        focal_term(cp[gi * M + ref], k == bucket, cfg.focal_alpha, cfg.focal_gamma, lv, gv);
        const float sc = cfg.w_cls * (sp_any ? 1.f : 0.f) / (float)(bs * K);
        acc[4] += lv * sc;
        gc[gi * M + ref] = gv * sc;
      }
      // L1 of the bucket group's reference-mode way-points against the interval's ground truth
      const int gb_ = cfg.interval_group[iv][min(bucket, K - 1)];
      const int g0 = cfg.interval_group[iv][0];
      const float *gt = cfg.gt_traj[g0] + (size_t)b * T * 2, *gm = cfg.gt_mask[g0] + (size_t)b * T;
      const float *q = rp + ((size_t)gb_ * M + ref) * T * 2;
      float *go = gr + ((size_t)gb_ * M + ref) * T * 2;
      float gx[kMaxTs], gy[kMaxTs];
      float cx = 0.f, cy = 0.f, px = 0.f, py = 0.f;
      const float sr = cfg.w_reg / (float)(bs * T * 2);
      for (int t = 0; t < T; ++t) {
        cx += gt[2 * t]; cy += gt[2 * t + 1];
        px += q[2 * t]; py += q[2 * t + 1];
        const float ex = px - cx, ey = py - cy;
        acc[5] += (fabsf(ex) + fabsf(ey)) * gm[t] * sr;
        gx[t] = sgn(ex) * gm[t] * sr;
        gy[t] = sgn(ey) * gm[t] * sr;
      }
      for (int t = T - 2; t >= 0; --t) { gx[t] += gx[t + 1]; gy[t] += gy[t + 1]; }
      for (int t = 0; t < T; ++t) { go[2 * t] = gx[t]; go[2 * t + 1] = gy[t]; }
    }
  }
  // ---- ego status: masked L1, mean over the (sample, S) elements of the layer
  if (S > 0 && tid < S) {
    const float x = layer_ptr(status, l)[(size_t)b * S + tid];
    const float t = cfg.ego_status[(size_t)b * S + tid], w = cfg.ego_status_mask[(size_t)b * S + tid];
    const float sc = cfg.w_status / (float)(bs * S);
    const float diff = x - t;
    acc[6] += fabsf(diff) * w * sc;
    g_status[(size_t)lb * S + tid] = sgn(diff) * w * sc;
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const float s = wave_sum(acc[i]);
    if (tid == 0 && s != 0.f) atomicAdd(terms + i * L + l, s);
  }
}

// =========================================================================================================
// Backward of the whole objective: every gradient element times the upstream gradient of ITS loss term.
// Segment = one gradient tensor [rows, width] inside the flat buffer; the term of an element depends on its column only
// (term_table[table_offset + column]); `extra` adds a second term's contribution to the first `extra_cols` columns
// (the centerness term's gradient into the box centre).
// =========================================================================================================
struct ScaleArgs {
  hipad_loss_segment seg[HIPAD_LOSS_MAX_SEGMENTS];
};

__global__ __launch_bounds__(256) void loss_scale_kernel(float *__restrict__ out, const float *__restrict__ grads,
                                                         const float *__restrict__ g_terms,
                                                         const signed char *__restrict__ term_table, const ScaleArgs a) {
  const hipad_loss_segment sg = a.seg[blockIdx.y];
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < sg.count; i += stride) {
    const int col = (int)(i % sg.width);
    float v = grads[sg.offset + i] * g_terms[term_table[sg.table_offset + col]];
    if (sg.extra_offset >= 0 && col < sg.extra_cols)
      v += grads[sg.extra_offset + (i / sg.width) * sg.extra_cols + col] * g_terms[sg.extra_term];
    out[sg.offset + i] = v;
  }
}

static int ok_or_launch() { return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH; }

}  // namespace hipad

using namespace hipad;

extern "C" {

int hipad_loss_det_assign(float *cost, int *n_rows, int *index, int *matched, float *count_out, const hipad_layer_ptrs *cls,
                          const hipad_layer_ptrs *box, const float *gt_boxes, const long long *labels, const int *count,
                          const hipad_det_loss_cfg *cfg, int layers, int bs, int P, int C, int D, int G, int gt_dim,
                          hipad_stream_t stream_) {
  if (!cost || !n_rows || !index || !matched || !count_out || !cls || !box || !gt_boxes || !labels || !count || !cfg)
    return HIPAD_EINVAL;
  if (layers <= 0 || layers > HIPAD_LOSS_MAX_LAYERS || bs <= 0 || P <= 0 || C <= 0 || D < kBoxDims || G <= 0 || G > kMaxGt ||
      gt_dim < 9)
    return HIPAD_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  const int LB = layers * bs;
  hipLaunchKernelGGL(det_cost_kernel, dim3((P + 255) / 256, LB), dim3(256), 0, stream, cost, n_rows, *cls, *box, gt_boxes,
                     labels, count, *cfg, bs, P, C, D, G, gt_dim);
  int rc = hipad_linear_assignment(index, cost, n_rows, LB, G, P, stream_);
  if (rc != HIPAD_OK) return rc;
  hipLaunchKernelGGL(det_finish_kernel, dim3(layers), dim3(256), 0, stream, matched, count_out, (const int *)index, count,
                     gt_boxes, bs, P, G, gt_dim);
  return ok_or_launch();
}

int hipad_loss_det(float *terms, float *grad_cls, float *grad_box, float *grad_box_cns, float *grad_quality,
                   const hipad_layer_ptrs *cls,
                   const hipad_layer_ptrs *box, const hipad_layer_ptrs *quality, const int *matched, const float *num_pos,
                   const float *gt_boxes, const long long *labels, const hipad_det_loss_cfg *cfg, int layers, int bs, int P,
                   int C, int D, int Q, int G, int gt_dim, hipad_stream_t stream_) {
  if (!terms || !grad_cls || !grad_box || !grad_box_cns || !cls || !box || !matched || !num_pos || !gt_boxes || !labels || !cfg)
    return HIPAD_EINVAL;
  if (Q > 0 && (!grad_quality || !quality)) return HIPAD_EINVAL;
  if (layers <= 0 || layers > HIPAD_LOSS_MAX_LAYERS || bs <= 0 || P <= 0 || C <= 0 || D < kBoxDims || G <= 0 || gt_dim < 9)
    return HIPAD_EINVAL;
  if (Q > 0 && (cfg->cns_index < 0 || cfg->cns_index >= Q || cfg->yns_index < 0 || cfg->yns_index >= Q)) return HIPAD_EINVAL;
  const hipad_layer_ptrs none = {};
  hipLaunchKernelGGL(det_loss_kernel, dim3((bs * P + 255) / 256, layers), dim3(256), 0, (hipStream_t)stream_, terms, grad_cls,
                     grad_box, grad_box_cns, grad_quality, *cls, *box, Q > 0 ? *quality : none, matched, num_pos, gt_boxes, labels,
                     *cfg,
                     layers, bs, P, C, D, Q, G, gt_dim);
  return ok_or_launch();
}

int hipad_loss_map_assign(float *cost, unsigned char *perm, int *n_rows, int *index, int *matched, int *order,
                          float *count_out, const hipad_layer_ptrs *cls, const hipad_layer_ptrs *pts, const float *gt_pts,
                          const long long *labels, const int *count, const hipad_map_loss_cfg *cfg, int layers, int bs, int P,
                          int C, int pts_dim, int G, int num_permute, hipad_stream_t stream_) {
  if (!cost || !perm || !n_rows || !index || !matched || !order || !count_out || !cls || !pts || !gt_pts || !labels || !count ||
      !cfg)
    return HIPAD_EINVAL;
  if (layers <= 0 || layers > HIPAD_LOSS_MAX_LAYERS || bs <= 0 || P <= 0 || C <= 0 || pts_dim != kLineDims || G <= 0 ||
      G > kMaxGt || num_permute <= 0 || num_permute > 255)
    return HIPAD_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  const int LB = layers * bs;
  hipLaunchKernelGGL(map_cost_kernel, dim3(G, LB), dim3(128), 0, stream, cost, perm, n_rows, *cls, *pts, gt_pts,
                     labels, count, *cfg, bs, P, C, G, num_permute);
  int rc = hipad_linear_assignment(index, cost, n_rows, LB, G, P, stream_);
  if (rc != HIPAD_OK) return rc;
  hipLaunchKernelGGL(map_finish_kernel, dim3(layers), dim3(256), 0, stream, matched, count_out, order, (const int *)index,
                     (const unsigned char *)perm, count, gt_pts, bs, P, G, num_permute);
  return ok_or_launch();
}

int hipad_loss_map(float *terms, float *grad_cls, float *grad_pts, const hipad_layer_ptrs *cls, const hipad_layer_ptrs *pts,
                   const int *matched, const int *order, const float *num_pos, const float *gt_pts, const long long *labels,
                   const hipad_map_loss_cfg *cfg, int layers, int bs, int P, int C, int pts_dim, int G, int num_permute,
                   hipad_stream_t stream_) {
  if (!terms || !grad_cls || !grad_pts || !cls || !pts || !matched || !order || !num_pos || !gt_pts || !labels || !cfg)
    return HIPAD_EINVAL;
  if (layers <= 0 || layers > HIPAD_LOSS_MAX_LAYERS || bs <= 0 || P <= 0 || C <= 0 || pts_dim != kLineDims || G <= 0 ||
      num_permute <= 0)
    return HIPAD_EINVAL;
  hipLaunchKernelGGL(map_loss_kernel, dim3((bs * P + 127) / 128, layers), dim3(128), 0, (hipStream_t)stream_, terms, grad_cls,
                     grad_pts, *cls, *pts, matched, order, num_pos, gt_pts, labels, *cfg, layers, bs, P, C, G, num_permute);
  return ok_or_launch();
}

int hipad_loss_motion(float *terms, float *grad_cls, float *grad_reg, const hipad_layer_ptrs *cls, const hipad_layer_ptrs *reg,
                      const int *det_matched, const float *num_pos, int num_pos_stride, const float *trajs, const float *masks,
                      const hipad_motion_loss_cfg *cfg, int layers, int bs, int A, int M, int T, int G,
                      hipad_stream_t stream_) {
  if (!terms || !grad_cls || !grad_reg || !cls || !reg || !det_matched || !num_pos || !trajs || !masks || !cfg)
    return HIPAD_EINVAL;
  if (layers <= 0 || layers > HIPAD_LOSS_MAX_LAYERS || bs <= 0 || A <= 0 || M <= 0 || M > kMaxModes || T <= 0 || T > kMaxTs ||
      G <= 0)
    return HIPAD_EINVAL;
  hipLaunchKernelGGL(motion_loss_kernel, dim3((bs * A + 255) / 256, layers), dim3(256), 0, (hipStream_t)stream_, terms,
                     grad_cls, grad_reg, *cls, *reg, det_matched, num_pos, num_pos_stride, trajs, masks, *cfg, layers, bs, A, M, T,
                     G);
  return ok_or_launch();
}

int hipad_loss_plan(float *terms, float *grad_cls, float *grad_reg, float *grad_status, const hipad_layer_ptrs *cls,
                    const hipad_layer_ptrs *reg, const hipad_layer_ptrs *status, const hipad_plan_loss_cfg *cfg, int layers,
                    int bs, int num_groups, int M, int T, int S, hipad_stream_t stream_) {
  if (!terms || !grad_cls || !grad_reg || !cls || !reg || !cfg) return HIPAD_EINVAL;
  if (S > 0 && (!grad_status || !status || !cfg->ego_status || !cfg->ego_status_mask)) return HIPAD_EINVAL;
  if (layers <= 0 || layers > HIPAD_LOSS_MAX_LAYERS || bs <= 0 || num_groups <= 0 || num_groups > HIPAD_LOSS_MAX_GROUPS ||
      M <= 0 || M > 64 || T <= 0 || T > kMaxTs || S > 64)
    return HIPAD_EINVAL;
  if (cfg->ref_group < 0 || cfg->ref_group >= num_groups || cfg->num_intervals < 0 ||
      cfg->num_intervals > HIPAD_LOSS_MAX_INTERVALS)
    return HIPAD_EINVAL;
  for (int g = 0; g < num_groups; ++g)
    if (!cfg->gt_traj[g] || !cfg->gt_mask[g] || cfg->kind[g] < 0 || cfg->kind[g] > 2) return HIPAD_EINVAL;
  for (int iv = 0; iv < cfg->num_intervals; ++iv) {
    if (cfg->interval_size[iv] <= 0 || cfg->interval_size[iv] > HIPAD_LOSS_MAX_BUCKETS) return HIPAD_EINVAL;
    for (int k = 0; k < cfg->interval_size[iv]; ++k)
      if (cfg->interval_group[iv][k] < 0 || cfg->interval_group[iv][k] >= num_groups) return HIPAD_EINVAL;
  }
  if (cfg->num_intervals > 0 && (!cfg->speed_traj || !cfg->speed_mask)) return HIPAD_EINVAL;
  const hipad_layer_ptrs none = {};
  hipLaunchKernelGGL(plan_loss_kernel, dim3(layers * bs), dim3(64), 0, (hipStream_t)stream_, terms, grad_cls, grad_reg,
                     grad_status, *cls, *reg, S > 0 ? *status : none, *cfg, layers, bs, num_groups, M, T, S);
  return ok_or_launch();
}

int hipad_loss_scale(float *out, const float *grads, const float *g_terms, const signed char *term_table,
                     const hipad_loss_segment *segments, int num_segments, hipad_stream_t stream_) {
  if (!out || !grads || !g_terms || !term_table || !segments || num_segments <= 0 || num_segments > HIPAD_LOSS_MAX_SEGMENTS)
    return HIPAD_EINVAL;
  ScaleArgs a;
  long long longest = 0;
  for (int i = 0; i < num_segments; ++i) {
    a.seg[i] = segments[i];
    if (segments[i].count < 0 || segments[i].width <= 0) return HIPAD_EINVAL;
    if (segments[i].count > longest) longest = segments[i].count;
  }
  long long blocks = (longest + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
  hipLaunchKernelGGL(loss_scale_kernel, dim3((unsigned)blocks, num_segments), dim3(256), 0, (hipStream_t)stream_, out, grads,
                     g_terms, term_table, a);
  return ok_or_launch();
}

}  // extern "C"
