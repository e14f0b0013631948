// hip-ad_amd/csrc/depthloss.hip -- the auxiliary dense-depth heads and their loss on the flat bf16 pyramid (gfx950).
//
// Replaces: DenseDepthNet.forward + .loss of the reference (models/blocks.py:266-326): per pyramid level a 1x1
// convolution 256 -> 1 on the level widened to fp32, exp, x focal / equal_focal, then
//   loss = sum_l  loss_weight * sum_valid |clamp(pred, 0, max_depth) - gt| / max(1, n_valid_l * num_levels)
// with valid = gt > 0 and pred not NaN.  As torch operators that is a widening copy of three pyramid levels (91 MB
// written), three library convolutions with their backward, and ~110 small elementwise / reduction launches per frame.
// Here: ONE pass over the rows of the flat pyramid (the encoder's bf16 output, read in place) per direction.
//   forward   one wave per row: dot(row, w_l) + b_l -> exp -> scale -> |clamp - gt| ; per-level error sums as 64-bit
//             fixed-point integers (order-independent: the loss value is bitwise reproducible) and valid counts
//   finish    loss and the per-level gradient coefficient loss_weight / max(1, n_valid_l * num_levels)
//   backward  rows with a valid target only (LiDAR-sparse): d loss / d logit, then  grad_feat row += g * w_l  (plain
//             read-modify-write into the frame's fp32 pyramid gradient -- every row is visited once),
//             grad w_l += g * row, grad b_l += g  (per workgroup partial sums, then fp32 atomics into the gradient buffers)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "hipad.h"
#include "wave_ops.h"
#include "daf_common.h"

namespace hipad {

constexpr double kDepthScale = 16777216.0;  // 2^24: |clamp - gt| <= ~1e3 per row, 1e5 rows, resolution 6e-8

struct DepthTable {
  const float *gt[HIPAD_DEPTH_MAX_LEVELS];
  const float *weight[HIPAD_DEPTH_MAX_LEVELS];
  const float *bias[HIPAD_DEPTH_MAX_LEVELS];
  float *grad_weight[HIPAD_DEPTH_MAX_LEVELS];
  float *grad_bias[HIPAD_DEPTH_MAX_LEVELS];
  int rows_per_sample[HIPAD_DEPTH_MAX_LEVELS];  // cams * h * w
  int rows_per_cam[HIPAD_DEPTH_MAX_LEVELS];     // h * w
  int row_off[HIPAD_DEPTH_MAX_LEVELS];          // first row of the level inside a sample's pyramid
  int cum[HIPAD_DEPTH_MAX_LEVELS + 1];          // cumulative bs * rows_per_sample
  int nlevels;
};

struct DepthRow {
  int level, cam;
  long flat_row, local;
};

__device__ __forceinline__ DepthRow depth_row(const DepthTable &t, int idx, long pyramid_rows, int cams) {
  DepthRow r;
  int l = 0;
#pragma unroll
  for (int i = 1; i < HIPAD_DEPTH_MAX_LEVELS; ++i)
    if (i < t.nlevels && idx >= t.cum[i]) l = i;
  r.level = l;
  r.local = idx - t.cum[l];
  const int n = t.rows_per_sample[l];
  const long b = r.local / n;
  const int in_sample = (int)(r.local - b * n);
  r.cam = (int)(b * cams) + in_sample / t.rows_per_cam[l];
  r.flat_row = b * pyramid_rows + t.row_off[l] + in_sample;
  return r;
}

__device__ __forceinline__ float4 bf16x4(const uint16_t *p) {
  const uint2 u = *reinterpret_cast<const uint2 *>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}

__global__ __launch_bounds__(256) void depth_fwd_kernel(float *__restrict__ pred, long long *__restrict__ err_fixed,
                                                        int *__restrict__ count, const uint16_t *__restrict__ feat,
                                                        const float *__restrict__ focal, const DepthTable tab,
                                                        long pyramid_rows, int cams, float inv_equal_focal, float max_depth) {
  __shared__ DepthTable t;
  for (int i = threadIdx.x; i < (int)(sizeof(DepthTable) / 4); i += blockDim.x)
    reinterpret_cast<int *>(&t)[i] = reinterpret_cast<const int *>(&tab)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int total = t.cum[t.nlevels];
  const int nwaves = gridDim.x * 4;
  float err[HIPAD_DEPTH_MAX_LEVELS];
  int cnt[HIPAD_DEPTH_MAX_LEVELS];
#pragma unroll
  for (int l = 0; l < HIPAD_DEPTH_MAX_LEVELS; ++l) err[l] = 0.f, cnt[l] = 0;
  for (int idx = uni(blockIdx.x * 4 + wv); idx < total; idx += nwaves) {
    const DepthRow r = depth_row(t, idx, pyramid_rows, cams);
    const float4 x = bf16x4(feat + r.flat_row * 256 + 4 * lane);
    const float *wp = t.weight[r.level] + 4 * lane;   // (a parameter inside a flat buffer: 4-byte alignment only)
    const float4 w = make_float4(wp[0], wp[1], wp[2], wp[3]);
    const float logit = wave_sum((x.x * w.x + x.y * w.y) + (x.z * w.z + x.w * w.w)) + t.bias[r.level][0];
    float d = expf(logit);
    if (focal) d *= focal[r.cam] * inv_equal_focal;
    const float gt = t.gt[r.level][r.local];
    if (lane == 0) pred[idx] = d;
    const bool valid = gt > 0.f && d == d;
    const float diff = fabsf(fminf(fmaxf(d, 0.f), max_depth) - gt);
#pragma unroll
    for (int l = 0; l < HIPAD_DEPTH_MAX_LEVELS; ++l)
      if (l == r.level && valid) err[l] += diff, cnt[l] += 1;
  }
  if (lane == 0) {
#pragma unroll
    for (int l = 0; l < HIPAD_DEPTH_MAX_LEVELS; ++l)
      if (l < t.nlevels && cnt[l] > 0) {
        const double v = (double)err[l] * kDepthScale;
        atomicAdd(reinterpret_cast<unsigned long long *>(err_fixed + l), (unsigned long long)__double2ll_rn(v));
        atomicAdd(count + l, cnt[l]);
      }
  }
}

// loss[0] = total, loss[1 + l] = per-level term; coef[l] = loss_weight / max(1, n_valid_l * num_levels)
__global__ void depth_finish_kernel(float *__restrict__ loss, float *__restrict__ coef, const long long *__restrict__ err_fixed,
                                    const int *__restrict__ count, int nlevels, float loss_weight) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float total = 0.f;
  for (int l = 0; l < nlevels; ++l) {
    const float n = fmaxf((float)count[l] * (float)nlevels, 1.f);
    const float e = (float)((double)err_fixed[l] / kDepthScale);
    const float term = e / n * loss_weight;
    coef[l] = loss_weight / n;
    loss[1 + l] = term;
    total += term;
  }
  loss[0] = total;
}

__global__ __launch_bounds__(256) void depth_bwd_kernel(float *__restrict__ grad_feat, const float *__restrict__ pred,
                                                        const float *__restrict__ coef, const float *__restrict__ upstream,
                                                        const uint16_t *__restrict__ feat, const DepthTable tab,
                                                        long pyramid_rows, int cams, float max_depth) {
  __shared__ DepthTable t;
  __shared__ float s_w[4][HIPAD_DEPTH_MAX_LEVELS][256];
  __shared__ float s_b[4][HIPAD_DEPTH_MAX_LEVELS];
  for (int i = threadIdx.x; i < (int)(sizeof(DepthTable) / 4); i += blockDim.x)
    reinterpret_cast<int *>(&t)[i] = reinterpret_cast<const int *>(&tab)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int total = t.cum[t.nlevels];
  const int nwaves = gridDim.x * 4;
  const float up = upstream ? upstream[0] : 1.f;
  float4 gw[HIPAD_DEPTH_MAX_LEVELS];
  float gb[HIPAD_DEPTH_MAX_LEVELS];
#pragma unroll
  for (int l = 0; l < HIPAD_DEPTH_MAX_LEVELS; ++l) gw[l] = make_float4(0.f, 0.f, 0.f, 0.f), gb[l] = 0.f;
  // a wave looks at 64 rows at a time (one per lane: prediction + target), then walks the rows that carry a gradient
  for (int base = uni(blockIdx.x * 4 + wv) * 64; base < total; base += nwaves * 64) {
    const int idx = base + lane;
    float g = 0.f;
    if (idx < total) {
      const DepthRow r = depth_row(t, idx, pyramid_rows, cams);
      const float d = pred[idx], gt = t.gt[r.level][r.local];
      const bool valid = gt > 0.f && d == d;
      const float dc = fminf(fmaxf(d, 0.f), max_depth);
      const float sgn = dc > gt ? 1.f : (dc < gt ? -1.f : 0.f);
      if (valid && d >= 0.f && d <= max_depth) g = up * coef[r.level] * sgn * d;   // d pred / d logit = pred
    }
    unsigned long long live = __ballot(g != 0.f);
    while (live) {
      const int src = (int)__builtin_ctzll(live);
      live &= live - 1;
      const float gl = rl_f(g, src);
      const DepthRow r = depth_row(t, base + src, pyramid_rows, cams);
      const float4 x = bf16x4(feat + r.flat_row * 256 + 4 * lane);
      const float *wp = t.weight[r.level] + 4 * lane;
      const float4 w = make_float4(wp[0], wp[1], wp[2], wp[3]);
      float4 *gf = reinterpret_cast<float4 *>(grad_feat + r.flat_row * 256) + lane;
      float4 o = *gf;
      o.x += gl * w.x; o.y += gl * w.y; o.z += gl * w.z; o.w += gl * w.w;
      *gf = o;
#pragma unroll
      for (int l = 0; l < HIPAD_DEPTH_MAX_LEVELS; ++l)
        if (l == r.level) {
          gw[l].x += gl * x.x; gw[l].y += gl * x.y; gw[l].z += gl * x.z; gw[l].w += gl * x.w;
          gb[l] += gl;
        }
    }
  }
#pragma unroll
  for (int l = 0; l < HIPAD_DEPTH_MAX_LEVELS; ++l) {
    reinterpret_cast<float4 *>(&s_w[wv][l][0])[lane] = gw[l];
    if (lane == 0) s_b[wv][l] = gb[l];
  }
  __syncthreads();
  for (int l = 0; l < t.nlevels; ++l) {
    const int c = threadIdx.x;
    const float v = (s_w[0][l][c] + s_w[1][l][c]) + (s_w[2][l][c] + s_w[3][l][c]);
    if (t.grad_weight[l] && v != 0.f) atomicAdd(t.grad_weight[l] + c, v);
    if (c == 0 && t.grad_bias[l]) {
      const float bsum = (s_b[0][l] + s_b[1][l]) + (s_b[2][l] + s_b[3][l]);
      if (bsum != 0.f) atomicAdd(t.grad_bias[l], bsum);
    }
  }
}

static int depth_table(DepthTable &t, const hipad_depth_level *levels, int nlevels, int bs, int cams, bool backward) {
  if (!levels || nlevels <= 0 || nlevels > HIPAD_DEPTH_MAX_LEVELS || bs <= 0 || cams <= 0) return HIPAD_EINVAL;
  memset(&t, 0, sizeof(t));
  long long cum = 0;
  for (int l = 0; l < nlevels; ++l) {
    const hipad_depth_level &lv = levels[l];
    if (!lv.gt || !lv.weight || !lv.bias || lv.rows_per_cam <= 0 || lv.row_offset < 0) return HIPAD_EINVAL;
    t.gt[l] = lv.gt; t.weight[l] = lv.weight; t.bias[l] = lv.bias;
    t.grad_weight[l] = backward ? lv.grad_weight : nullptr;
    t.grad_bias[l] = backward ? lv.grad_bias : nullptr;
    t.rows_per_cam[l] = lv.rows_per_cam;
    t.rows_per_sample[l] = lv.rows_per_cam * cams;
    t.row_off[l] = lv.row_offset;
    t.cum[l] = (int)cum;
    cum += (long long)bs * lv.rows_per_cam * cams;
    if (cum > 0x7fffffffll - 64 * 8192) return HIPAD_ERANGE;
  }
  t.cum[nlevels] = (int)cum;
  t.nlevels = nlevels;
  return HIPAD_OK;
}

}  // namespace hipad

using namespace hipad;

extern "C" {

size_t hipad_depth_loss_workspace(void) { return 64; }

int hipad_depth_loss_forward(float *loss, float *coef, float *pred, void *workspace, size_t workspace_bytes,
                             const unsigned short *feat, long long pyramid_rows, const float *focal,
                             const hipad_depth_level *levels, int nlevels, int bs, int cams, float equal_focal,
                             float max_depth, float loss_weight, hipad_stream_t stream_) {
  DepthTable t;
  const int rc = depth_table(t, levels, nlevels, bs, cams, false);
  if (rc != HIPAD_OK) return rc;
  if (!loss || !coef || !pred || !feat || pyramid_rows <= 0 || !(equal_focal > 0.f)) return HIPAD_EINVAL;
  if (!workspace || workspace_bytes < hipad_depth_loss_workspace() || ((uintptr_t)workspace & 7)) return HIPAD_EWORKSPACE;
  if (((uintptr_t)feat & 7) != 0) return HIPAD_EINVAL;
  for (int l = 0; l < nlevels; ++l)
    if ((long long)levels[l].row_offset + (long long)levels[l].rows_per_cam * cams > pyramid_rows) return HIPAD_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  long long *err_fixed = (long long *)workspace;
  int *count = (int *)((char *)workspace + 8 * HIPAD_DEPTH_MAX_LEVELS);
  if (fill_zero(workspace, 64, stream) != HIPAD_OK) return HIPAD_ELAUNCH;   // a kernel, not a memset node
  const int total = t.cum[nlevels];
  int blocks = (total + 4 * 8 - 1) / (4 * 8);   // about eight rows per wave
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(depth_fwd_kernel, dim3(blocks), dim3(256), 0, stream, pred, err_fixed, count, feat, focal, t,
                     (long)pyramid_rows, cams, 1.f / equal_focal, max_depth);
  hipLaunchKernelGGL(depth_finish_kernel, dim3(1), dim3(64), 0, stream, loss, coef, (const long long *)err_fixed,
                     (const int *)count, nlevels, loss_weight);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_depth_loss_backward(float *grad_feat, const float *pred, const float *coef, const float *upstream,
                              const unsigned short *feat, long long pyramid_rows, const hipad_depth_level *levels,
                              int nlevels, int bs, int cams, float max_depth, hipad_stream_t stream_) {
  DepthTable t;
  const int rc = depth_table(t, levels, nlevels, bs, cams, true);
  if (rc != HIPAD_OK) return rc;
  if (!grad_feat || !pred || !coef || !feat || pyramid_rows <= 0) return HIPAD_EINVAL;
  if ((((uintptr_t)grad_feat & 15) | ((uintptr_t)feat & 7)) != 0) return HIPAD_EINVAL;
  const int total = t.cum[nlevels];
  int blocks = (total + 4 * 64 * 4 - 1) / (4 * 64 * 4);   // about four 64-row looks per wave
  if (blocks > 256) blocks = 256;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(depth_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, grad_feat, pred, coef, upstream, feat,
                     t, (long)pyramid_rows, cams, max_depth);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

}  // extern "C"
