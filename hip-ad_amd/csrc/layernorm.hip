// hip-ad_amd/csrc/layernorm.hip -- LayerNorm over the last dimension, forward and backward.
//
// Replaces: the nn.LayerNorm calls of the decoder -- the tail of every linear_relu_ln stack (reference
// models/blocks.py:32-42), the "norm" ops of the decoder program (hipad_b2d_stage2.py:293) and the pre-norm
// of AsymmetricFFN (blocks.py:352-353): ~440 calls per frame.  torch runs the backward of each as three
// kernels (input gradient, two-stage gamma/beta reduction) plus two autograd accumulation adds; here it
// is ONE kernel: a wave owns a few rows, keeps its gamma / beta partial sums in registers, and adds them
// straight into the parameters' gradient buffers (like the Linear kernel does for dW / db).
//
// One wave per row at a time; lane l holds elements 4 l + 256 j (float4), N <= 1024, N % 4 == 0.
// fp32 throughout, two-pass statistics in registers (mean, then centred variance), rstd = rsqrt(var + eps).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hipad.h"
#include "wave_ops.h"

namespace hipad {

constexpr int kLnMaxJ = 4;  // N <= 1024

__device__ __forceinline__ float wave_sum64(float v) { return wave_sum(v); }  // DPP, wave-uniform (wave_ops.h)

template <int J>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(float *__restrict__ y, float *__restrict__ mean_out,
                                                            float *__restrict__ rstd_out, const float *__restrict__ x,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            int M, int N, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float4 *x4 = reinterpret_cast<const float4 *>(x + (size_t)row * N);
  float4 v[J];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int c = 4 * lane + 256 * j;
    v[j] = c < N ? x4[lane + 64 * j] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += v[j].x + v[j].y + v[j].z + v[j].w;
  }
  const float mean = wave_sum64(s) / (float)N;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int c = 4 * lane + 256 * j;
    if (c < N) {
      const float a = v[j].x - mean, b = v[j].y - mean, cc = v[j].z - mean, d = v[j].w - mean;
      q += a * a + b * b + cc * cc + d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum64(q) / (float)N + eps);
  float4 *y4 = reinterpret_cast<float4 *>(y + (size_t)row * N);
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int c = 4 * lane + 256 * j;
    if (c < N) {
      float4 g = make_float4(1.f, 1.f, 1.f, 1.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gamma) g = reinterpret_cast<const float4 *>(gamma)[lane + 64 * j];
      if (beta) b = reinterpret_cast<const float4 *>(beta)[lane + 64 * j];
      y4[lane + 64 * j] = make_float4((v[j].x - mean) * rstd * g.x + b.x, (v[j].y - mean) * rstd * g.y + b.y,
                                      (v[j].z - mean) * rstd * g.z + b.z, (v[j].w - mean) * rstd * g.w + b.w);
    }
  }
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
}

// dx = rstd * (gh - mean(gh) - xhat * mean(gh * xhat)), gh = dy * gamma, xhat = (x - mean) * rstd
// dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy    (atomics into the caller's buffers)
template <int J>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(float *__restrict__ dx, float *__restrict__ dgamma,
                                                            float *__restrict__ dbeta, const float *__restrict__ dy,
                                                            const float *__restrict__ x, const float *__restrict__ mean,
                                                            const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                            int M, int N, int rows_per_wave) {
  __shared__ float4 sg[4][64 * J];
  __shared__ float4 sb[4][64 * J];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + wv;
  const int r0 = wave * rows_per_wave, r1 = min(M, r0 + rows_per_wave);
  float4 ag[J], ab[J], g[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    ag[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = 4 * lane + 256 * j;
    g[j] = (gamma && c < N) ? reinterpret_cast<const float4 *>(gamma)[lane + 64 * j] : make_float4(1.f, 1.f, 1.f, 1.f);
  }
  // RB rows at a time with all their loads issued first: a wave owns a few rows to keep the parameter-gradient
  // atomics few, and walking them one by one cost one full memory latency per row
  constexpr int RB = J == 1 ? 4 : (J == 2 ? 2 : 1);
  for (int row = r0; row < r1; row += RB) {
    float4 xv[RB][J], dv[RB][J];
    float mu[RB], rs[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const int rr = min(row + b, r1 - 1);  // past the wave's last row: a repeat, computed but neither stored nor summed
      const float4 *x4 = reinterpret_cast<const float4 *>(x + (size_t)rr * N);
      const float4 *d4 = reinterpret_cast<const float4 *>(dy + (size_t)rr * N);
      mu[b] = mean[rr];
      rs[b] = rstd[rr];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const bool in = 4 * lane + 256 * j < N;
        xv[b][j] = in ? x4[lane + 64 * j] : make_float4(0.f, 0.f, 0.f, 0.f);
        dv[b][j] = in ? d4[lane + 64 * j] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const bool live = row + b < r1;
      float4 xh[J], gh[J];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int c = 4 * lane + 256 * j;
        if (c < N) {
          const float4 xq = xv[b][j], dq = dv[b][j];
          xh[j] = make_float4((xq.x - mu[b]) * rs[b], (xq.y - mu[b]) * rs[b], (xq.z - mu[b]) * rs[b], (xq.w - mu[b]) * rs[b]);
          gh[j] = make_float4(dq.x * g[j].x, dq.y * g[j].y, dq.z * g[j].z, dq.w * g[j].w);
          s1 += gh[j].x + gh[j].y + gh[j].z + gh[j].w;
          s2 += gh[j].x * xh[j].x + gh[j].y * xh[j].y + gh[j].z * xh[j].z + gh[j].w * xh[j].w;
          if (live) {
            ag[j].x += dq.x * xh[j].x; ag[j].y += dq.y * xh[j].y; ag[j].z += dq.z * xh[j].z; ag[j].w += dq.w * xh[j].w;
            ab[j].x += dq.x; ab[j].y += dq.y; ab[j].z += dq.z; ab[j].w += dq.w;
          }
        } else {
          xh[j] = gh[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      const float m1 = wave_sum64(s1) / (float)N, m2 = wave_sum64(s2) / (float)N;
      if (dx && live) {
        float4 *o4 = reinterpret_cast<float4 *>(dx + (size_t)(row + b) * N);
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const int c = 4 * lane + 256 * j;
          if (c < N)
            o4[lane + 64 * j] = make_float4(rs[b] * (gh[j].x - m1 - xh[j].x * m2), rs[b] * (gh[j].y - m1 - xh[j].y * m2),
                                            rs[b] * (gh[j].z - m1 - xh[j].z * m2), rs[b] * (gh[j].w - m1 - xh[j].w * m2));
        }
      }
    }
  }
  if (!dgamma && !dbeta) return;
  // combine the block's four waves in LDS, then one atomic per element and block
#pragma unroll
  for (int j = 0; j < J; ++j) {
    sg[wv][lane + 64 * j] = ag[j];
    sb[wv][lane + 64 * j] = ab[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N / 4; i += 256) {
    const int j = i >> 6, l = i & 63;  // element group i = lane l of slab j  (c = 4 l + 256 j = 4 i)
    float4 a = sg[0][l + 64 * j], b = sb[0][l + 64 * j];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 a2 = sg[w][l + 64 * j], b2 = sb[w][l + 64 * j];
      a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
      b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
    }
    if (dgamma) {
      atomicAdd(dgamma + 4 * i + 0, a.x); atomicAdd(dgamma + 4 * i + 1, a.y);
      atomicAdd(dgamma + 4 * i + 2, a.z); atomicAdd(dgamma + 4 * i + 3, a.w);
    }
    if (dbeta) {
      atomicAdd(dbeta + 4 * i + 0, b.x); atomicAdd(dbeta + 4 * i + 1, b.y);
      atomicAdd(dbeta + 4 * i + 2, b.z); atomicAdd(dbeta + 4 * i + 3, b.w);
    }
  }
}

static int check_ln(int M, int N) {
  if (M <= 0 || N <= 0) return HIPAD_EINVAL;
  if ((N & 3) || N > 256 * kLnMaxJ) return HIPAD_EINVAL;
  if ((long long)M * N >= (1ll << 31)) return HIPAD_ERANGE;
  return HIPAD_OK;
}

}  // namespace hipad

using namespace hipad;

extern "C" {

int hipad_layernorm_forward(float *y, float *mean, float *rstd, const float *x, const float *gamma, const float *beta,
                            int M, int N, float eps, hipad_stream_t stream_) {
  int rc = check_ln(M, N);
  if (rc != HIPAD_OK) return rc;
  if (!y || !x) return HIPAD_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  const dim3 grid((M + 3) / 4), block(256);
  const int J = (N + 255) / 256;
#define HIPAD_LN_FWD(JJ) hipLaunchKernelGGL((layernorm_fwd_kernel<JJ>), grid, block, 0, stream, y, mean, rstd, x, gamma, beta, M, N, eps)
  if (J == 1) HIPAD_LN_FWD(1); else if (J == 2) HIPAD_LN_FWD(2); else if (J == 3) HIPAD_LN_FWD(3); else HIPAD_LN_FWD(4);
#undef HIPAD_LN_FWD
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_layernorm_backward(float *dx, float *dgamma, float *dbeta, const float *dy, const float *x, const float *mean,
                             const float *rstd, const float *gamma, int M, int N, hipad_stream_t stream_) {
  int rc = check_ln(M, N);
  if (rc != HIPAD_OK) return rc;
  if (!dy || !x || !mean || !rstd) return HIPAD_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  // rows per wave: enough waves to cover the chip for large M, few atomics per parameter element for small M
  int rpw = (M + 1023) / 1024;
  if (rpw < 4) rpw = M >= 64 ? 4 : 1;
  const int waves = (M + rpw - 1) / rpw;
  const dim3 grid((waves + 3) / 4), block(256);
  const int J = (N + 255) / 256;
#define HIPAD_LN_BWD(JJ) hipLaunchKernelGGL((layernorm_bwd_kernel<JJ>), grid, block, 0, stream, dx, dgamma, dbeta, dy, x, mean, rstd, gamma, M, N, rpw)
  if (J == 1) HIPAD_LN_BWD(1); else if (J == 2) HIPAD_LN_BWD(2); else if (J == 3) HIPAD_LN_BWD(3); else HIPAD_LN_BWD(4);
#undef HIPAD_LN_BWD
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

}  // extern "C"
