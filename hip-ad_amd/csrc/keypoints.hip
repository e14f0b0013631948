// hip-ad_amd/csrc/keypoints.hip -- key points of a 3D-box query, generated AND projected into every camera in one
// launch (forward) / one launch + a zero fill (backward).
//
// Replaces: SparseBox3DKeyPointsGenerator.forward (reference models/det/blocks.py:183-224: exp of the log-size, fixed
// offsets = fix_scale * size, learnable offsets = (sigmoid(fc(x)) - 0.5) * size, rotation by the box yaw, translation
// to the box centre) chained with DeformableFeatureAggregation.project_points + permute (models/blocks.py:216-225,
// 144-145).  In torch that chain is ~15 small elementwise kernels forward and ~25 backward per call, 12 calls per
// frame (det and ego queries, 6 layers); here the key points never exist in memory.
//
// Arithmetic: the same operations in the same order as the torch expression (no fma contraction), then the
// bit-exact projection of proj_wsm.hip (project_one is restated here with the identical expression).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hipad.h"
#include "daf_common.h"

namespace hipad {

struct KpProj {
  float p0, p1, p2, zc;
};

__device__ __forceinline__ KpProj kp_project_one(const float *__restrict__ M, float x, float y, float z) {
#pragma clang fp contract(off)
  KpProj r;
  r.p0 = ((M[0] * x + M[1] * y) + M[2] * z) + M[3];
  r.p1 = ((M[4] * x + M[5] * y) + M[6] * z) + M[7];
  r.p2 = ((M[8] * x + M[9] * y) + M[10] * z) + M[11];
  r.zc = fmaxf(r.p2, 1e-5f);
  if (r.p2 != r.p2) r.zc = r.p2;
  return r;
}

struct BoxPoint {
  float ox, oy, oz;     // offset in the box frame
  float sx, sy, sz;     // exp(log size)
  float s0, s1, s2;     // sigmoid of the learnable logits (0 for fixed points)
  float kx, ky, kz;     // key point in the lidar frame
  float sn, cs;
};

// anchor row layout: [x, y, z, log w, log l, log h, sin, cos, v...] (core/box3d.py)
__device__ __forceinline__ BoxPoint box_point(const float *__restrict__ an, const float *__restrict__ fix,
                                              const float *__restrict__ learn, int p, int n_fix) {
#pragma clang fp contract(off)
  BoxPoint q;
  q.sx = expf(an[3]); q.sy = expf(an[4]); q.sz = expf(an[5]);
  q.sn = an[6]; q.cs = an[7];
  if (p < n_fix) {
    q.s0 = q.s1 = q.s2 = 0.f;
    q.ox = fix[p * 3 + 0] * q.sx; q.oy = fix[p * 3 + 1] * q.sy; q.oz = fix[p * 3 + 2] * q.sz;
  } else {
    const float *l = learn + (p - n_fix) * 3;
    q.s0 = 1.f / (1.f + expf(-l[0])); q.s1 = 1.f / (1.f + expf(-l[1])); q.s2 = 1.f / (1.f + expf(-l[2]));
    q.ox = (q.s0 - 0.5f) * q.sx; q.oy = (q.s1 - 0.5f) * q.sy; q.oz = (q.s2 - 0.5f) * q.sz;
  }
  q.kx = (q.cs * q.ox - q.sn * q.oy) + an[0];
  q.ky = (q.sn * q.ox + q.cs * q.oy) + an[1];
  q.kz = q.oz + an[2];
  return q;
}

__global__ __launch_bounds__(256) void box_points_project_fwd_kernel(
    float *__restrict__ loc, float *__restrict__ kp_out, const float *__restrict__ anchor, const float *__restrict__ fix,
    const float *__restrict__ learn, const float *__restrict__ pm, const float *__restrict__ wh, long npt /* bs*A*P */,
    int A, int P, int n_fix, int cams, int anchor_dim) {
  const long pt = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pt >= npt) return;
  const int p = (int)(pt % P);
  const long ba = pt / P;
  const long b = ba / A;
  const BoxPoint q = box_point(anchor + ba * anchor_dim, fix, learn ? learn + ba * (long)(P - n_fix) * 3 : nullptr, p, n_fix);
  if (kp_out) {
    kp_out[pt * 3 + 0] = q.kx; kp_out[pt * 3 + 1] = q.ky; kp_out[pt * 3 + 2] = q.kz;
  }
  for (int cam = 0; cam < cams; ++cam) {
    const float *M = pm + (b * cams + cam) * 16;
    const KpProj r = kp_project_one(M, q.kx, q.ky, q.kz);
    float u = r.p0 / r.zc, v = r.p1 / r.zc;
    if (wh) {
      u = u / wh[(b * cams + cam) * 2];
      v = v / wh[(b * cams + cam) * 2 + 1];
    }
    reinterpret_cast<float2 *>(loc)[pt * cams + cam] = make_float2(u, v);
  }
}

// grad_anchor [bs*A, anchor_dim] must be zero on entry (columns 0..7 are accumulated with atomics over the points);
// grad_learn [bs*A, (P - n_fix)*3] is overwritten.
__global__ __launch_bounds__(256) void box_points_project_bwd_kernel(
    float *__restrict__ g_anchor, float *__restrict__ g_learn, const float *__restrict__ gloc,
    const float *__restrict__ anchor, const float *__restrict__ fix, const float *__restrict__ learn,
    const float *__restrict__ pm, const float *__restrict__ wh, long npt, int A, int P, int n_fix, int cams, int anchor_dim) {
  const long pt = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pt >= npt) return;
  const int p = (int)(pt % P);
  const long ba = pt / P;
  const long b = ba / A;
  const float *an = anchor + ba * anchor_dim;
  const BoxPoint q = box_point(an, fix, learn ? learn + ba * (long)(P - n_fix) * 3 : nullptr, p, n_fix);
  float gx = 0.f, gy = 0.f, gz = 0.f;
  for (int cam = 0; cam < cams; ++cam) {
    const float *M = pm + (b * cams + cam) * 16;
    const KpProj r = kp_project_one(M, q.kx, q.ky, q.kz);
    const float2 g = reinterpret_cast<const float2 *>(gloc)[pt * cams + cam];
    float gu = g.x, gv = g.y;
    if (wh) {
      gu = gu / wh[(b * cams + cam) * 2];
      gv = gv / wh[(b * cams + cam) * 2 + 1];
    }
    const float inv = 1.f / r.zc;
    const float gp0 = gu * inv, gp1 = gv * inv;
    const float gp2 = (r.p2 >= 1e-5f) ? -(gu * r.p0 + gv * r.p1) * inv * inv : 0.f;
    gx += gp0 * M[0] + gp1 * M[4] + gp2 * M[8];
    gy += gp0 * M[1] + gp1 * M[5] + gp2 * M[9];
    gz += gp0 * M[2] + gp1 * M[6] + gp2 * M[10];
  }
  // key point = R(yaw) offset + centre
  const float gox = q.cs * gx + q.sn * gy, goy = -q.sn * gx + q.cs * gy, goz = gz;
  const float gcs = q.ox * gx + q.oy * gy, gsn = -q.oy * gx + q.ox * gy;
  float gsx, gsy, gsz;  // d / d size
  if (p < n_fix) {
    gsx = gox * fix[p * 3 + 0]; gsy = goy * fix[p * 3 + 1]; gsz = goz * fix[p * 3 + 2];
  } else {
    gsx = gox * (q.s0 - 0.5f); gsy = goy * (q.s1 - 0.5f); gsz = goz * (q.s2 - 0.5f);
    float *gl = g_learn + ba * (long)(P - n_fix) * 3 + (p - n_fix) * 3;
    gl[0] = gox * q.sx * q.s0 * (1.f - q.s0);
    gl[1] = goy * q.sy * q.s1 * (1.f - q.s1);
    gl[2] = goz * q.sz * q.s2 * (1.f - q.s2);
  }
  float *ga = g_anchor + ba * anchor_dim;
  atomicAdd(ga + 0, gx); atomicAdd(ga + 1, gy); atomicAdd(ga + 2, gz);
  atomicAdd(ga + 3, gsx * q.sx); atomicAdd(ga + 4, gsy * q.sy); atomicAdd(ga + 5, gsz * q.sz);  // size = exp(log size)
  atomicAdd(ga + 6, gsn); atomicAdd(ga + 7, gcs);
}

// ------------------------------------------------------------------------------------------------------------
// Poly-line queries (map elements, plan trajectories): key point (s, h, k) of anchor a =
//   (anchor[a, s, :2] + offset[a, s, h, k, :2],  height[h])        reference models/map/blocks.py:193-218
// projected into every camera.  offset = the learnable_fc output, P = S * Hn * K points per anchor.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void line_points_project_fwd_kernel(
    float *__restrict__ loc, const float *__restrict__ anchor, const float *__restrict__ offset,
    const float *__restrict__ heights, const float *__restrict__ pm, const float *__restrict__ wh, long npt, int A, int S,
    int Hn, int K, int cams) {
#pragma clang fp contract(off)
  const long pt = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pt >= npt) return;
  const int P = S * Hn * K;
  const int p = (int)(pt % P);
  const long ba = pt / P;
  const long b = ba / A;
  const int s = p / (Hn * K), h = (p / K) % Hn;
  const float x = anchor[(ba * S + s) * 2 + 0] + offset[pt * 2 + 0];
  const float y = anchor[(ba * S + s) * 2 + 1] + offset[pt * 2 + 1];
  const float z = heights[h];
  for (int cam = 0; cam < cams; ++cam) {
    const float *M = pm + (b * cams + cam) * 16;
    const KpProj r = kp_project_one(M, x, y, z);
    float u = r.p0 / r.zc, v = r.p1 / r.zc;
    if (wh) {
      u = u / wh[(b * cams + cam) * 2];
      v = v / wh[(b * cams + cam) * 2 + 1];
    }
    reinterpret_cast<float2 *>(loc)[pt * cams + cam] = make_float2(u, v);
  }
}

// grad_offset [bs*A*P, 2] overwritten; grad_anchor [bs*A*S, 2] (zero on entry) += sum over (h, k), by atomics:
// one thread per key point keeps 30-40 thousand threads in flight where one per (anchor, sample) had 2 000
__global__ __launch_bounds__(256) void line_points_project_bwd_kernel(
    float *__restrict__ g_anchor, float *__restrict__ g_offset, const float *__restrict__ gloc,
    const float *__restrict__ anchor, const float *__restrict__ offset, const float *__restrict__ heights,
    const float *__restrict__ pm, const float *__restrict__ wh, long npt, int A, int S, int Hn, int K, int cams) {
  const long pt = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pt >= npt) return;
  const int HK = Hn * K;
  const long as = pt / HK;               // (b, a, s)
  const int hk = (int)(pt - as * HK);
  const long b = as / ((long)A * S);
  float x, y;
  {
#pragma clang fp contract(off)
    x = anchor[as * 2 + 0] + offset[pt * 2 + 0];
    y = anchor[as * 2 + 1] + offset[pt * 2 + 1];
  }
  const float z = heights[hk / K];
  float gx = 0.f, gy = 0.f;
  for (int cam = 0; cam < cams; ++cam) {
    const float *M = pm + (b * cams + cam) * 16;
    const KpProj r = kp_project_one(M, x, y, z);
    const float2 g = reinterpret_cast<const float2 *>(gloc)[pt * cams + cam];
    float gu = g.x, gv = g.y;
    if (wh) {
      gu = gu / wh[(b * cams + cam) * 2];
      gv = gv / wh[(b * cams + cam) * 2 + 1];
    }
    const float inv = 1.f / r.zc;
    const float gp0 = gu * inv, gp1 = gv * inv;
    const float gp2 = (r.p2 >= 1e-5f) ? -(gu * r.p0 + gv * r.p1) * inv * inv : 0.f;
    gx += gp0 * M[0] + gp1 * M[4] + gp2 * M[8];
    gy += gp0 * M[1] + gp1 * M[5] + gp2 * M[9];
  }
  g_offset[pt * 2 + 0] = gx;
  g_offset[pt * 2 + 1] = gy;
  atomicAdd(g_anchor + as * 2 + 0, gx);
  atomicAdd(g_anchor + as * 2 + 1, gy);
}

}  // namespace hipad

using namespace hipad;

extern "C" {

int hipad_box_points_project_forward(float *loc, float *key_points, const float *anchor, const float *fix_scale,
                                     const float *learn, const float *projection_mat, const float *image_wh, int bs,
                                     int A, int n_fix, int n_learn, int cams, int anchor_dim, hipad_stream_t stream) {
  if (!loc || !anchor || !fix_scale || !projection_mat || (n_learn > 0 && !learn)) return HIPAD_EINVAL;
  if (bs <= 0 || A <= 0 || n_fix < 0 || n_learn < 0 || n_fix + n_learn <= 0 || cams <= 0 || anchor_dim < 8) return HIPAD_EINVAL;
  const int P = n_fix + n_learn;
  const long npt = (long)bs * A * P;
  hipLaunchKernelGGL(box_points_project_fwd_kernel, dim3((unsigned)((npt + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     loc, key_points, anchor, fix_scale, learn, projection_mat, image_wh, npt, A, P, n_fix, cams, anchor_dim);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_box_points_project_backward(float *grad_anchor, float *grad_learn, const float *grad_loc, const float *anchor,
                                      const float *fix_scale, const float *learn, const float *projection_mat,
                                      const float *image_wh, int bs, int A, int n_fix, int n_learn, int cams,
                                      int anchor_dim, hipad_stream_t stream_) {
  if (!grad_anchor || !grad_loc || !anchor || !fix_scale || !projection_mat || (n_learn > 0 && (!learn || !grad_learn)))
    return HIPAD_EINVAL;
  if (bs <= 0 || A <= 0 || n_fix < 0 || n_learn < 0 || n_fix + n_learn <= 0 || cams <= 0 || anchor_dim < 8) return HIPAD_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  const int P = n_fix + n_learn;
  const long npt = (long)bs * A * P;
  if (fill_zero(grad_anchor, (size_t)bs * A * anchor_dim * sizeof(float), stream) != HIPAD_OK) return HIPAD_ELAUNCH;
  hipLaunchKernelGGL(box_points_project_bwd_kernel, dim3((unsigned)((npt + 255) / 256)), dim3(256), 0, stream, grad_anchor,
                     grad_learn, grad_loc, anchor, fix_scale, learn, projection_mat, image_wh, npt, A, P, n_fix, cams,
                     anchor_dim);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_line_points_project_forward(float *loc, const float *anchor, const float *offset, const float *heights,
                                      const float *projection_mat, const float *image_wh, int bs, int A, int num_sample,
                                      int num_heights, int num_learnable, int cams, hipad_stream_t stream) {
  if (!loc || !anchor || !offset || !heights || !projection_mat) return HIPAD_EINVAL;
  if (bs <= 0 || A <= 0 || num_sample <= 0 || num_heights <= 0 || num_learnable <= 0 || cams <= 0) return HIPAD_EINVAL;
  const long npt = (long)bs * A * num_sample * num_heights * num_learnable;
  hipLaunchKernelGGL(line_points_project_fwd_kernel, dim3((unsigned)((npt + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     loc, anchor, offset, heights, projection_mat, image_wh, npt, A, num_sample, num_heights, num_learnable, cams);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_line_points_project_backward(float *grad_anchor, float *grad_offset, const float *grad_loc, const float *anchor,
                                       const float *offset, const float *heights, const float *projection_mat,
                                       const float *image_wh, int bs, int A, int num_sample, int num_heights,
                                       int num_learnable, int cams, hipad_stream_t stream) {
  if (!grad_anchor || !grad_offset || !grad_loc || !anchor || !offset || !heights || !projection_mat) return HIPAD_EINVAL;
  if (bs <= 0 || A <= 0 || num_sample <= 0 || num_heights <= 0 || num_learnable <= 0 || cams <= 0) return HIPAD_EINVAL;
  const long npt = (long)bs * A * num_sample * num_heights * num_learnable;
  if (fill_zero(grad_anchor, (size_t)bs * A * num_sample * 2 * sizeof(float), (hipStream_t)stream) != HIPAD_OK) return HIPAD_ELAUNCH;
  hipLaunchKernelGGL(line_points_project_bwd_kernel, dim3((unsigned)((npt + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     grad_anchor, grad_offset, grad_loc, anchor, offset, heights, projection_mat, image_wh, npt, A, num_sample,
                     num_heights, num_learnable, cams);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

}  // extern "C"
