// hip-ad_amd/csrc/batchnorm.hip -- training-mode BatchNorm of the image encoder on bf16 channels-last activations, fused
// with what follows it in a ResNet bottleneck: the identity add and the ReLU.
//
// Replaces (reference: mmdet 2.28.2 ResNet / FPN built by projects/configs/hipad_b2d_stage2.py:112-134, run from
// models/sparse_detector.py:66-94): per norm layer the library's three forward launches (mean/variance partials, their
// finalisation, normalisation) + the ReLU (+ the residual add) and its three backward launches + the ReLU backward --
// 61 layers, ~500 launches per frame -- by TWO launches forward and TWO backward:
//   bn_stats_kernel      per-channel sum and sum of squares (fp32) of x
//   bn_apply_kernel      y = relu((x - mean) * rstd * gamma + beta + residual); the first workgroup also stores
//                        (mean, rstd) for the backward and updates the running statistics
//   bn_bwd_reduce_kernel per-channel sum of dy' and of dy' * xhat, dy' = dy gated by y > 0
//   bn_bwd_apply_kernel  dx = gamma * rstd * (dy' - mean(dy') - xhat * mean(dy' xhat)), d(residual) = dy'; the first
//                        workgroup adds d(gamma), d(beta) into the parameters' fp32 gradient buffers
// Layout: x is (rows, C) row-major bf16 (NHWC), rows = N*H*W.  A thread owns 8 consecutive channels (one 16-byte load);
// C / 8 threads cover a row, 256 / (C / 8) rows are in flight per workgroup.  HBM-bound: forward reads x twice and
// writes y once (6 B / element), backward reads dy, y, x twice and writes dx (14 B / element).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hipad.h"

namespace hipad {

// The reductions need many workgroups in flight to reach HBM speed, but every workgroup ends with one atomic per channel:
// same-address atomics retire at ~20 ns each, so the per-channel sums are kept in kReplicas copies (workgroup b adds to
// copy b % kReplicas) and the consumers add the copies up.
//
// The sums are accumulated as 64-bit FIXED-POINT integers: integer addition is associative, so the result does not
// depend on the order in which the workgroups' atomics arrive and the statistics -- hence the whole encoder forward and
// every activation derived from it -- are bitwise reproducible from run to run.  With fp32 atomics the sums moved in the
// last bit, the bf16 activations of the following layers flipped roundings here and there, and two runs of the same
// frame ended 4e-3 apart in the pyramid (relative L2; tools/diag_forward_determinism.py), which the random-init decoder
// amplifies to tens of percent in the gradient.  A workgroup's own partial sum is a fixed-order fp32 sum as before; its
// conversion (round to nearest at 2^-24 resp. 2^-40) costs 1e-7 relative on sums of magnitude >= 1, fp32 level.
constexpr int kReplicas = 4;
constexpr double kFwdScale = 16777216.0;              // 2^24: range 5e11 (sum of squares of 1e6 rows of |x| ~ 700)
constexpr double kBwdScale = 1099511627776.0;         // 2^40: range 8e6, resolution 9e-13 (gradient sums are small)

__device__ __forceinline__ long long to_fixed(float v, double scale) {
  const double d = (double)v * scale;
  if (!(d == d)) return 0;                            // NaN input: the statistics are garbage either way; keep the sum finite
  return d >= 9.0e18 ? 9000000000000000000ll : (d <= -9.0e18 ? -9000000000000000000ll : __double2ll_rn(d));
}
__device__ __forceinline__ float from_fixed(long long v, double scale) { return (float)((double)v / scale); }
__device__ __forceinline__ void add_fixed(long long *dst, float v, double scale) {
  atomicAdd(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)to_fixed(v, scale));
}
constexpr int kUnroll = 4;   // rows per thread in flight

__device__ __forceinline__ void unpack8(const uint4 &u, float *f) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(w[i] << 16);
    f[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
  }
}

__device__ __forceinline__ uint32_t bf16_rne(float v) {
  uint32_t u = __float_as_uint(v);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;   // NaN stays NaN
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}

__device__ __forceinline__ uint4 pack8(const float *f) {
  uint4 u;
  u.x = bf16_rne(f[0]) | (bf16_rne(f[1]) << 16);
  u.y = bf16_rne(f[2]) | (bf16_rne(f[3]) << 16);
  u.z = bf16_rne(f[4]) | (bf16_rne(f[5]) << 16);
  u.w = bf16_rne(f[6]) | (bf16_rne(f[7]) << 16);
  return u;
}

// Block-level combine of the 16 per-thread partials (two 8-channel vectors) over the rows in flight, then one atomic per
// channel and quantity.  All 256 threads take part: the 2C outputs are dealt over the threads (two threads per output
// when 2C < 256), so a thread reads at most 16 LDS words -- with only the C / 8 threads of one row doing the sums (first
// version) a 64-channel layer spent 15 us here.
__device__ __forceinline__ void combine_and_add(long long *__restrict__ gsum, double scale, const float *a, const float *b,
                                                int tpr, int C, float (*sh)[17], float *sh2) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sh[tid][k] = a[k];
    sh[tid][8 + k] = b[k];
  }
  __syncthreads();
  const int rpi = 256 / tpr, outs = 2 * C;
  if (outs >= 256) {
    for (int o = tid; o < outs; o += 256) {
      const int q = o / C, c = o - q * C, lc = c >> 3, k = (c & 7) + 8 * q;
      float acc = 0.f;
      for (int lr = 0; lr < rpi; ++lr) acc += sh[lr * tpr + lc][k];
      add_fixed(gsum + o, acc, scale);
    }
  } else {                       // 2C = 128: two threads per output, each over half of the rows in flight
    const int o = tid % outs, half = tid / outs, parts = 256 / outs;
    const int q = o / C, c = o - q * C, lc = c >> 3, k = (c & 7) + 8 * q;
    const int per = rpi / parts;
    float acc = 0.f;
    for (int lr = half * per; lr < (half + 1) * per; ++lr) acc += sh[lr * tpr + lc][k];
    sh2[tid] = acc;
    __syncthreads();
    if (tid < outs) {
      for (int p = 1; p < parts; ++p) acc += sh2[tid + p * outs];
      add_fixed(gsum + o, acc, scale);
    }
  }
}

__global__ __launch_bounds__(256) void bn_stats_kernel(long long *__restrict__ sums, const uint4 *__restrict__ x, long rows, int C,
                                                       long rows_per_block) {
  __shared__ float sh[256][17];
  __shared__ float sh2[256];
  const int tpr = C >> 3, rpi = 256 / tpr;
  const int lc = threadIdx.x % tpr, lr = threadIdx.x / tpr;
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  float s[8], ss[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = ss[k] = 0.f;
  for (long r = r0 + lr; r < r1; r += (long)kUnroll * rpi) {
    uint4 v[kUnroll];
    bool ok[kUnroll];
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {      // all loads first (clamped address, no branch), then the arithmetic
      const long rr = r + (long)j * rpi;
      ok[j] = rr < r1;
      v[j] = x[(ok[j] ? rr : r) * tpr + lc];
    }
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      float f[8];
      unpack8(v[j], f);
      const float m = ok[j] ? 1.f : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        s[k] += m * f[k];
        ss[k] += m * f[k] * f[k];
      }
    }
  }
  combine_and_add(sums + (blockIdx.x % kReplicas) * 2 * C, kFwdScale, s, ss, tpr, C, sh, sh2);
}

__global__ __launch_bounds__(256) void bn_apply_kernel(uint4 *__restrict__ y, const uint4 *__restrict__ x,
                                                       const uint4 *__restrict__ residual, const long long *__restrict__ sums,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       float *__restrict__ running_mean, float *__restrict__ running_var,
                                                       float *__restrict__ save /* [2C]: mean, rstd */, long rows, int C,
                                                       float eps, float momentum, int relu, long rows_per_block,
                                                       long y_group_rows, long y_group_stride /* in rows */) {
  const int tpr = C >> 3, rpi = 256 / tpr;
  const int lc = threadIdx.x % tpr, lr = threadIdx.x / tpr;
  const float inv_m = 1.f / (float)rows;
  // all prologue loads are issued before anything waits: a branch inside this loop (first version: the first workgroup's
  // stores) made every channel a separate memory round trip -- 27 us per launch instead of 7
  float scale[8], shift[8], mean_[8], var_[8], s1[8], s2[8], gm[8], bt[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = lc * 8 + k;
    long long q1 = 0, q2 = 0;                      // the replicas are summed as integers: order-free too
#pragma unroll
    for (int rep = 0; rep < kReplicas; ++rep) {
      q1 += sums[rep * 2 * C + c];
      q2 += sums[rep * 2 * C + C + c];
    }
    s1[k] = from_fixed(q1, kFwdScale);
    s2[k] = from_fixed(q2, kFwdScale);
    gm[k] = gamma[c];
    bt[k] = beta[c];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float mean = s1[k] * inv_m;
    float var = s2[k] * inv_m - mean * mean;
    var = var > 0.f ? var : 0.f;
    const float rstd = rsqrtf(var + eps);
    scale[k] = gm[k] * rstd;
    shift[k] = bt[k] - mean * scale[k];
    mean_[k] = mean;
    var_[k] = var;
  }
  if (blockIdx.x == 0 && lr == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = lc * 8 + k;
      save[c] = mean_[k];
      save[C + c] = rsqrtf(var_[k] + eps);
      if (running_mean) {
        const float unbiased = rows > 1 ? var_[k] * ((float)rows / (float)(rows - 1)) : var_[k];
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean_[k];
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
      }
    }
  }
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  for (long r = r0 + lr; r < r1; r += (long)kUnroll * rpi) {
    uint4 vx[kUnroll], vr[kUnroll];
    bool ok[kUnroll];
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      const long rr = r + (long)j * rpi;
      ok[j] = rr < r1;
      const long i = (ok[j] ? rr : r) * tpr + lc;
      vx[j] = x[i];
      if (residual) vr[j] = residual[i];     // uniform branch
    }
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      float f[8], o[8];
      unpack8(vx[j], f);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = f[k] * scale[k] + shift[k];
      if (residual) {
        float g[8];
        unpack8(vr[j], g);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] += g[k];
      }
      if (relu) {
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = o[k] > 0.f ? o[k] : 0.f;
      }
      if (ok[j]) {
        // output row: groups of y_group_rows consecutive rows sit y_group_stride rows apart (one group = one sample's
        // block of a level inside the flat pyramid; a plain tensor is one group)
        const long rr = r + (long)j * rpi;
        const long g = rr / y_group_rows;
        y[(g * y_group_stride + (rr - g * y_group_rows)) * tpr + lc] = pack8(o);
      }
    }
  }
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(long long *__restrict__ gsums, const uint4 *__restrict__ dy,
                                                            const uint4 *__restrict__ y /* NULL: no ReLU gate */,
                                                            const uint4 *__restrict__ x, const float *__restrict__ save,
                                                            long rows, int C, long rows_per_block) {
  __shared__ float sh[256][17];
  __shared__ float sh2[256];
  const int tpr = C >> 3, rpi = 256 / tpr;
  const int lc = threadIdx.x % tpr, lr = threadIdx.x / tpr;
  float mean[8], rstd[8], s[8], sx[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    mean[k] = save[lc * 8 + k];
    rstd[k] = save[C + lc * 8 + k];
    s[k] = sx[k] = 0.f;
  }
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  for (long r = r0 + lr; r < r1; r += (long)kUnroll * rpi) {
    uint4 vg[kUnroll], vx[kUnroll], vy[kUnroll];
    bool ok[kUnroll];
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      const long rr = r + (long)j * rpi;
      ok[j] = rr < r1;
      const long i = (ok[j] ? rr : r) * tpr + lc;
      vg[j] = dy[i];
      vx[j] = x[i];
      if (y) vy[j] = y[i];     // uniform branch
    }
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      float g[8], f[8];
      unpack8(vg[j], g);
      unpack8(vx[j], f);
      if (y) {
        float o[8];
        unpack8(vy[j], o);
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = o[k] > 0.f ? g[k] : 0.f;
      }
      const float m = ok[j] ? 1.f : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        s[k] += m * g[k];
        sx[k] += m * g[k] * ((f[k] - mean[k]) * rstd[k]);
      }
    }
  }
  combine_and_add(gsums + (blockIdx.x % kReplicas) * 2 * C, kBwdScale, s, sx, tpr, C, sh, sh2);
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(uint4 *__restrict__ dx, uint4 *__restrict__ dres /* may be NULL */,
                                                           float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                           const uint4 *__restrict__ dy, const uint4 *__restrict__ y,
                                                           const uint4 *__restrict__ x, const float *__restrict__ save,
                                                           const float *__restrict__ gamma, const long long *__restrict__ gsums,
                                                           long rows, int C, long rows_per_block) {
  const int tpr = C >> 3, rpi = 256 / tpr;
  const int lc = threadIdx.x % tpr, lr = threadIdx.x / tpr;
  const float inv_m = 1.f / (float)rows;
  float mean[8], rstd[8], a[8], b[8], gr[8], sdy[8], sdyx[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {           // loads only (no branch in here, see bn_apply_kernel)
    const int c = lc * 8 + k;
    mean[k] = save[c];
    rstd[k] = save[C + c];
    long long q1 = 0, q2 = 0;
#pragma unroll
    for (int rep = 0; rep < kReplicas; ++rep) {
      q1 += gsums[rep * 2 * C + c];
      q2 += gsums[rep * 2 * C + C + c];
    }
    sdy[k] = from_fixed(q1, kBwdScale);
    sdyx[k] = from_fixed(q2, kBwdScale);
    gr[k] = gamma[c];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a[k] = sdy[k] * inv_m;
    b[k] = sdyx[k] * inv_m;
    gr[k] *= rstd[k];
  }
  if (blockIdx.x == 0 && lr == 0) {         // single writer per channel: accumulate into the parameter gradients
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = lc * 8 + k;
      if (dgamma) dgamma[c] += sdyx[k];
      if (dbeta) dbeta[c] += sdy[k];
    }
  }
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  for (long r = r0 + lr; r < r1; r += (long)kUnroll * rpi) {
    uint4 vg[kUnroll], vx[kUnroll], vy[kUnroll];
    bool ok[kUnroll];
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      const long rr = r + (long)j * rpi;
      ok[j] = rr < r1;
      const long i = (ok[j] ? rr : r) * tpr + lc;
      vg[j] = dy[i];
      vx[j] = x[i];
      if (y) vy[j] = y[i];
    }
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      float g[8], f[8], o[8];
      unpack8(vg[j], g);
      unpack8(vx[j], f);
      if (y) {
        float yy[8];
        unpack8(vy[j], yy);
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = yy[k] > 0.f ? g[k] : 0.f;
      }
      const long i = (r + (long)j * rpi) * tpr + lc;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = gr[k] * (g[k] - a[k] - (f[k] - mean[k]) * rstd[k] * b[k]);
      if (ok[j]) {
        if (dres) dres[i] = pack8(g);
        dx[i] = pack8(o);
      }
    }
  }
}

static inline bool bn_shape_ok(long rows, int C) {
  if (rows <= 0 || C < 64 || C > 2048 || (C & 7)) return false;
  const int tpr = C >> 3;
  return 256 % tpr == 0;
}

static inline void bn_grid(long rows, int C, unsigned *blocks, long *rows_per_block) {
  const int rpi = 256 / (C >> 3);
  // at least one full unrolled iteration per workgroup; two for wide layers, whose per-workgroup fixed cost (the 2C-entry
  // prologue reads and 2C atomics) is as large as one iteration's data
  const long per_iter = (long)rpi * kUnroll * (C >= 512 ? 2 : 1);
  long want = (rows + per_iter - 1) / per_iter;
  if (want > 1024) want = 1024;
  long rpb = (rows + want - 1) / want;
  rpb = (rpb + rpi - 1) / rpi * rpi;
  *rows_per_block = rpb;
  *blocks = (unsigned)((rows + rpb - 1) / rpb);
}

}  // namespace hipad

using namespace hipad;

extern "C" {

int hipad_bn_supported(long long rows, int channels) { return bn_shape_ok((long)rows, channels) ? 1 : 0; }

int hipad_bn_forward_grouped(void *y, float *save, float *sums, const void *x, const void *residual, const float *gamma,
                             const float *beta, float *running_mean, float *running_var, long long rows, int channels,
                             float eps, float momentum, int relu, long long y_group_rows, long long y_group_stride_rows,
                             hipad_stream_t stream_);

int hipad_bn_forward(void *y, float *save, float *sums, const void *x, const void *residual, const float *gamma,
                     const float *beta, float *running_mean, float *running_var, long long rows, int channels, float eps,
                     float momentum, int relu, hipad_stream_t stream_) {
  return hipad_bn_forward_grouped(y, save, sums, x, residual, gamma, beta, running_mean, running_var, rows, channels, eps,
                                  momentum, relu, rows, rows, stream_);
}

int hipad_bn_forward_grouped(void *y, float *save, float *sums, const void *x, const void *residual, const float *gamma,
                             const float *beta, float *running_mean, float *running_var, long long rows, int channels,
                             float eps, float momentum, int relu, long long y_group_rows, long long y_group_stride_rows,
                             hipad_stream_t stream_) {
  if (!y || !save || !sums || !x || !gamma || !beta) return HIPAD_EINVAL;
  if (y_group_rows <= 0 || rows % y_group_rows || y_group_stride_rows < y_group_rows) return HIPAD_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return HIPAD_EINVAL;
  if (!bn_shape_ok((long)rows, channels)) return HIPAD_ERANGE;
  if ((((uintptr_t)y | (uintptr_t)x | (uintptr_t)residual) & 15) != 0) return HIPAD_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  unsigned blocks;
  long rpb;
  bn_grid((long)rows, channels, &blocks, &rpb);
  if ((uintptr_t)sums & 7) return HIPAD_EINVAL;
  hipLaunchKernelGGL(bn_stats_kernel, dim3(blocks), dim3(256), 0, stream, (long long *)sums, (const uint4 *)x, (long)rows,
                     channels, rpb);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, stream, (uint4 *)y, (const uint4 *)x,
                     (const uint4 *)residual, (const long long *)sums, gamma, beta, running_mean, running_var, save, (long)rows, channels, eps,
                     momentum, relu ? 1 : 0, rpb, (long)y_group_rows, (long)y_group_stride_rows);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_bn_backward(void *dx, void *dres, float *dgamma, float *dbeta, float *gsums, const void *dy, const void *y,
                      const void *x, const float *save, const float *gamma, long long rows, int channels,
                      hipad_stream_t stream_) {
  if (!dx || !gsums || !dy || !x || !save || !gamma) return HIPAD_EINVAL;
  if (!bn_shape_ok((long)rows, channels)) return HIPAD_ERANGE;
  if ((((uintptr_t)dx | (uintptr_t)dres | (uintptr_t)dy | (uintptr_t)y | (uintptr_t)x) & 15) != 0) return HIPAD_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  unsigned blocks;
  long rpb;
  bn_grid((long)rows, channels, &blocks, &rpb);
  if ((uintptr_t)gsums & 7) return HIPAD_EINVAL;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(blocks), dim3(256), 0, stream, (long long *)gsums, (const uint4 *)dy,
                     (const uint4 *)y, (const uint4 *)x, save, (long)rows, channels, rpb);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, stream, (uint4 *)dx, (uint4 *)dres, dgamma, dbeta,
                     (const uint4 *)dy, (const uint4 *)y, (const uint4 *)x, save, gamma, (const long long *)gsums, (long)rows,
                     channels, rpb);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

}  // extern "C"
