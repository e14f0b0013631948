// hip-ad_amd/csrc/glue.hip -- small fused kernels for the decoder's "glue": each replaces a dozen or more elementwise
// / slice / cat launches (and their autograd mirror images) that sit between the fat kernels of a decoder layer.
//
//   chunk_mix_kernel       out chunk g = sum_k W[g][k] * (x0 chunk k + x1 chunk k): the query mixing of the planning
//                          refinement head (reference models/plan/blocks.py:120-135: "aligned" = sum of the temp / spat
//                          anchor groups' queries, speed queries = aligned + the speed groups of one interval) -- and,
//                          with W transposed, its backward.
//   motion_embed_kernel    class-conditioned motion-mode anchors rotated by the box yaw, last way-point, 2-D sine
//                          embedding (reference models/sparse_onedecoder.py:428-444 get_motion_anchor +
//                          models/attention.py:292-306 gen_sineembed_for_position): ~25 launches per decoder layer in
//                          torch, no gradient flows through it (argmax, detached boxes, frozen anchors).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hipad.h"

namespace hipad {

struct MixTable {
  float w[HIPAD_MIX_MAX][HIPAD_MIX_MAX];  // [out chunk][in chunk]
};

// x: (bs, K * rows, C), out: (bs, G * rows, C); C % 4 == 0; one float4 per thread
__global__ __launch_bounds__(256) void chunk_mix_kernel(float *__restrict__ out, const float *__restrict__ x0,
                                                        const float *__restrict__ x1, MixTable t, int bs, int K, int G,
                                                        int rows, int C4) {
  const long total = (long)bs * G * rows * C4;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c4 = (int)(idx % C4);
  long rem = idx / C4;
  const int r = (int)(rem % rows);
  rem /= rows;
  const int g = (int)(rem % G), b = (int)(rem / G);
  const float4 *p0 = reinterpret_cast<const float4 *>(x0) + ((long)b * K * rows + r) * C4 + c4;
  const float4 *p1 = x1 ? reinterpret_cast<const float4 *>(x1) + ((long)b * K * rows + r) * C4 + c4 : nullptr;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < K; ++k) {
    const float w = t.w[g][k];
    if (w != 0.f) {
      float4 v = p0[(long)k * rows * C4];
      if (p1) {
        const float4 u = p1[(long)k * rows * C4];
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
    }
  }
  reinterpret_cast<float4 *>(out)[idx] = acc;
}

// out (bs, A, modes, 2 * half): [sin/cos interleaved embedding of y | of x] of the rotated last way-point
//   cls (bs, A, ncls) logits, box (bs, A, D) with sin / cos yaw in columns sin_col / cos_col, table (ncls, modes, ts, 2),
//   freq (half): 10000 ** (2 * (k // 2) / half) as torch computed it.  One thread per (b, a, mode, k).
// Arithmetic order = the torch expression the reference evaluates (separate mul / sub kernels: no FMA contraction).
__global__ __launch_bounds__(256) void motion_embed_kernel(float *__restrict__ out, const float *__restrict__ cls,
                                                           const float *__restrict__ box, const float *__restrict__ table,
                                                           const float *__restrict__ freq, long n_anchor, int ncls, int D,
                                                           int sin_col, int cos_col, int modes, int ts, int half) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = n_anchor * modes * half;
  if (idx >= total) return;
  const int k = (int)(idx % half);
  const long am = idx / half;
  const int m = (int)(am % modes);
  const long a = am / modes;
  const float *cl = cls + a * ncls;
  int best = 0;
  float bv = cl[0];
  for (int c = 1; c < ncls; ++c) {  // torch.argmax: first maximum; NaN counts as the maximum
    const float v = cl[c];
    if (v > bv || (v != v && bv == bv)) { bv = v; best = c; }
  }
  const float *bx = box + a * D;
  const float yaw = atan2f(bx[sin_col], bx[cos_col]);
  const float cy = cosf(yaw), sy = sinf(yaw);
  const float *pt = table + (((long)best * modes + m) * ts + (ts - 1)) * 2;
  const float x = pt[0], y = pt[1];
  const float rx = __fsub_rn(__fmul_rn(cy, x), __fmul_rn(sy, y));
  const float ry = __fadd_rn(__fmul_rn(sy, x), __fmul_rn(cy, y));
  const float f = freq[k];
  const float ay = __fdiv_rn(__fmul_rn(ry, 6.283185307179586f), f);
  const float ax = __fdiv_rn(__fmul_rn(rx, 6.283185307179586f), f);
  float *o = out + am * (2 * half);
  o[k] = (k & 1) ? cosf(ay) : sinf(ay);
  o[half + k] = (k & 1) ? cosf(ax) : sinf(ax);
}

// out[i] = 1 / (1 - p) with probability 1 - p, else 0 -- one counter-based draw per element from (seed, device step
// counter, i): the Bernoulli keep mask of DeformableFeatureAggregation's attn_drop (reference models/blocks.py:209-212:
// torch.rand(...) > p, .float(), / (1 - p): four launches) in one launch, a fresh mask on every replay of a hipGraph.
__global__ __launch_bounds__(256) void keep_mask_kernel(float *__restrict__ out, long n, uint32_t thresh, float inv_keep,
                                                        uint32_t seed, const uint32_t *__restrict__ seed_dev) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t x = seed + (seed_dev ? *seed_dev * 0x9E3779B9u : 0u);
  x ^= (uint32_t)i * 0xC2B2AE3Du;
  x = (x ^ (x >> 15)) * 0x2C1B3C6Du;
  x ^= (uint32_t)(i >> 32) * 0x27D4EB2Fu + 0x165667B1u;
  x = (x ^ (x >> 13)) * 0x297A2D39u;
  x ^= x >> 16;
  out[i] = x >= thresh ? inv_keep : 0.f;
}

// out[i] = base[i] + keep(i) * x[i] / (1 - p)  (base may be NULL): the identity connection of the attention and FFN
// blocks, `identity + dropout(x)` (reference models/attention.py:283-289 dropout_layer + residual,
// models/blocks.py:383-396), as one launch; the same kernel with base = NULL applied to the output gradient is its
// backward -- the mask is recomputed from (seed, device step counter, i) like keep_mask_kernel's, never stored.
__global__ __launch_bounds__(256) void dropout_add_kernel(float *__restrict__ out, const float *__restrict__ x,
                                                          const float *__restrict__ base, long n4, uint32_t thresh,
                                                          float inv_keep, uint32_t seed, const uint32_t *__restrict__ seed_dev) {
  const long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i4 >= n4) return;
  const uint32_t s0 = seed + (seed_dev ? *seed_dev * 0x9E3779B9u : 0u);
  const float4 v = reinterpret_cast<const float4 *>(x)[i4];
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
  if (base) b = reinterpret_cast<const float4 *>(base)[i4];
  const float in[4] = {v.x, v.y, v.z, v.w};
  float o[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long i = i4 * 4 + k;
    uint32_t h = s0 ^ ((uint32_t)i * 0xC2B2AE3Du);
    h = (h ^ (h >> 15)) * 0x2C1B3C6Du;
    h ^= (uint32_t)(i >> 32) * 0x27D4EB2Fu + 0x165667B1u;
    h = (h ^ (h >> 13)) * 0x297A2D39u;
    h ^= h >> 16;
    if (h >= thresh) o[k] += in[k] * inv_keep;
  }
  reinterpret_cast<float4 *>(out)[i4] = make_float4(o[0], o[1], o[2], o[3]);
}

// GridMask on the device in one launch (reference models/grid_mask.py:92-138 for the detector's settings: rotate = 1,
// offset = False): out = x * mask, mask = product of the row and the column stripe patterns of this step's draw
// params = [apply, d, l, st_h, st_w] (device memory, written outside a hipGraph), optionally inverted (mode 1), identity
// when apply == 0.  Reads NCHW fp32, writes fp32 or bf16 with the caller's element strides (channels-last bf16 = what
// the encoder's first convolution consumes: no separate cast and re-layout passes).
__device__ __forceinline__ bool gm_inside(int idx, int length, int canvas, float d, float ln, float st) {
  const float pos = (float)(idx + (canvas - length) / 2);
  const float k = floorf((pos - st) / d);
  return pos >= st && k < floorf((float)canvas / d) && (pos - st) - k * d < ln;
}

__global__ __launch_bounds__(256) void grid_mask_kernel(void *__restrict__ out, const float *__restrict__ x,
                                                        const float *__restrict__ params, int n, int c, int h, int w,
                                                        int use_h, int use_w, int mode, int out_bf16, long sn, long sc,
                                                        long sy, long sx) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)n * h * w;
  if (idx >= total) return;
  const int px = (int)(idx % w);
  const long t = idx / w;
  const int py = (int)(t % h), img = (int)(t / h);
  const float apply = params[0], d = params[1], ln = params[2], st_h = params[3], st_w = params[4];
  float m = 1.f;
  if (apply > 0.f) {
    const float rows = (use_h && gm_inside(py, h, (int)(1.5 * h), d, ln, st_h)) ? 0.f : 1.f;
    const float cols = (use_w && gm_inside(px, w, (int)(1.5 * w), d, ln, st_w)) ? 0.f : 1.f;
    m = rows * cols;
    if (mode == 1) m = 1.f - m;
  }
  for (int ch = 0; ch < c; ++ch) {
    const float v = x[(((long)img * c + ch) * h + py) * w + px] * m;
    const long o = (long)img * sn + (long)ch * sc + (long)py * sy + (long)px * sx;
    if (out_bf16) {
      uint32_t u = __float_as_uint(v);
      u += 0x7FFFu + ((u >> 16) & 1u);                 // round to nearest even (finite inputs: pixel values)
      reinterpret_cast<uint16_t *>(out)[o] = (uint16_t)(u >> 16);
    } else {
      reinterpret_cast<float *>(out)[o] = v;
    }
  }
}

// Way-points -> per-step offsets of a trajectory (rows of T steps x D coordinates): out[t] = x[t] - x[t-1], out[0] =
// x[0] (the planning head's output format, reference models/sparse_onedecoder.py refine step of the plan branch:
// cat(wp[:1], wp[1:] - wp[:-1])); adjoint = 1 applies the transposed map to a gradient: out[t] = g[t] - g[t+1],
// out[T-1] = g[T-1].
__global__ __launch_bounds__(256) void step_offsets_kernel(float *__restrict__ out, const float *__restrict__ x, long n,
                                                           int T, int D, int adjoint) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int t = (int)((i / D) % T);
  float v = x[i];
  if (adjoint) {
    if (t + 1 < T) v -= x[i + D];
  } else {
    if (t > 0) v -= x[i - D];
  }
  out[i] = v;
}

// out[b, r, :] = base[b, r, :] + sum_k rows_k[b, :]  (up to three row vectors broadcast over the N rows of a sample): the
// planning branch's `embed + target-point embed + command embed + ego embed` and `feature + ego feature` (reference
// models/sparse_onedecoder.py, plan refinement), one launch instead of one per addend.  float4 per thread.
__global__ __launch_bounds__(256) void add_rows_kernel(float *__restrict__ out, const float *__restrict__ base,
                                                       const float *__restrict__ r0, const float *__restrict__ r1,
                                                       const float *__restrict__ r2, long n4, int N, int C4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const long row = i / C4;
  const int c = (int)(i - row * C4);
  const long b = row / N;
  float4 v = reinterpret_cast<const float4 *>(base)[i];
  const float *rs[3] = {r0, r1, r2};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (rs[k]) {
      const float4 a = reinterpret_cast<const float4 *>(rs[k])[b * C4 + c];
      v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
  }
  reinterpret_cast<float4 *>(out)[i] = v;
}

// out[b, c] = sum over the N rows of x[b, :, c] -- the gradient of a broadcast row vector.  One workgroup of 1024 threads
// per (sample, 64 columns): sixteen row phases of 64 lanes each read coalesced 256-byte rows, eight loads in flight per
// thread (480 rows = 4 dependent round trips), partial sums meet in LDS in a fixed order (no atomics: the result does
// not depend on scheduling).
__global__ __launch_bounds__(1024) void rows_sum_kernel(float *__restrict__ out, const float *__restrict__ x, int N, int C) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const long b = blockIdx.y;
  float s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = 0.f;
  if (c < C) {
    const float *xb = x + b * (long)N * C + c;
    int r = ph;
    for (; r + 7 * 16 < N; r += 8 * 16) {
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += xb[(long)(r + 16 * k) * C];
    }
    for (; r < N; r += 16) s[0] += xb[(long)r * C];
  }
  part[ph][lane] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (ph == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][lane];
    out[b * C + c] = t;
  }
}

}  // namespace hipad

using namespace hipad;

extern "C" {

int hipad_chunk_mix(float *out, const float *x0, const float *x1, const float *weights, int bs, int in_chunks,
                    int out_chunks, int rows, int channels, hipad_stream_t stream) {
  if (!out || !x0 || !weights || bs <= 0 || rows <= 0 || channels <= 0 || (channels & 3)) return HIPAD_EINVAL;
  if (in_chunks < 1 || out_chunks < 1 || in_chunks > HIPAD_MIX_MAX || out_chunks > HIPAD_MIX_MAX) return HIPAD_ERANGE;
  if ((((uintptr_t)out | (uintptr_t)x0 | (uintptr_t)x1) & 15) != 0) return HIPAD_EINVAL;
  MixTable t;
  for (int g = 0; g < HIPAD_MIX_MAX; ++g)
    for (int k = 0; k < HIPAD_MIX_MAX; ++k)
      t.w[g][k] = (g < out_chunks && k < in_chunks) ? weights[g * in_chunks + k] : 0.f;
  const long total = (long)bs * out_chunks * rows * (channels / 4);
  hipLaunchKernelGGL(chunk_mix_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, x0, x1,
                     t, bs, in_chunks, out_chunks, rows, channels / 4);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_keep_mask(float *out, long long n, float p_drop, unsigned seed, const unsigned *seed_dev, hipad_stream_t stream) {
  if (!out || n <= 0 || !(p_drop >= 0.f) || !(p_drop < 1.f)) return HIPAD_EINVAL;
  const double t = (double)p_drop * 4294967296.0;
  const uint32_t thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
  hipLaunchKernelGGL(keep_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, (long)n,
                     thresh, 1.f / (1.f - p_drop), seed, seed_dev);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_dropout_add(float *out, const float *x, const float *base, long long n, float p_drop, unsigned seed,
                      const unsigned *seed_dev, hipad_stream_t stream) {
  if (!out || !x || n <= 0 || (n & 3) || !(p_drop >= 0.f) || !(p_drop < 1.f)) return HIPAD_EINVAL;
  if ((((uintptr_t)out | (uintptr_t)x | (uintptr_t)base) & 15) != 0) return HIPAD_EINVAL;
  const double t = (double)p_drop * 4294967296.0;
  const uint32_t thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
  const long n4 = (long)(n / 4);
  hipLaunchKernelGGL(dropout_add_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, x, base,
                     n4, thresh, 1.f / (1.f - p_drop), seed, seed_dev);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_grid_mask(void *out, int out_bf16, const long long *out_strides, const float *x, const float *params, int n, int c,
                    int h, int w, int use_h, int use_w, int mode, hipad_stream_t stream) {
  if (!out || !out_strides || !x || !params || n <= 0 || c <= 0 || h <= 1 || w <= 1) return HIPAD_EINVAL;
  const long total = (long)n * h * w;
  hipLaunchKernelGGL(grid_mask_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, x, params,
                     n, c, h, w, use_h ? 1 : 0, use_w ? 1 : 0, mode, out_bf16 ? 1 : 0, (long)out_strides[0],
                     (long)out_strides[1], (long)out_strides[2], (long)out_strides[3]);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_motion_query_embed(float *out, const float *cls, const float *box, const float *table, const float *freq,
                             long long n_anchor, int num_classes, int box_dim, int sin_col, int cos_col, int modes,
                             int steps, int half_dim, hipad_stream_t stream) {
  if (!out || !cls || !box || !table || !freq) return HIPAD_EINVAL;
  if (n_anchor <= 0 || num_classes <= 0 || modes <= 0 || steps <= 0 || half_dim <= 0) return HIPAD_EINVAL;
  if (sin_col < 0 || cos_col < 0 || sin_col >= box_dim || cos_col >= box_dim) return HIPAD_EINVAL;
  const long total = (long)n_anchor * modes * half_dim;
  if (total >= (1l << 40)) return HIPAD_ERANGE;
  hipLaunchKernelGGL(motion_embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out,
                     cls, box, table, freq, (long)n_anchor, num_classes, box_dim, sin_col, cos_col, modes, steps, half_dim);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_step_offsets(float *out, const float *x, long long rows, int steps, int dims, int adjoint, hipad_stream_t stream) {
  if (!out || !x || out == x || rows <= 0 || steps <= 0 || dims <= 0) return HIPAD_EINVAL;
  const long n = (long)rows * steps * dims;
  if (n >= (1l << 40)) return HIPAD_ERANGE;
  hipLaunchKernelGGL(step_offsets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, x, n,
                     steps, dims, adjoint ? 1 : 0);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_add_rows(float *out, const float *base, const float *rows0, const float *rows1, const float *rows2, int bs,
                   int n_rows, int channels, hipad_stream_t stream) {
  if (!out || !base || !rows0 || bs <= 0 || n_rows <= 0 || channels <= 0 || (channels & 3)) return HIPAD_EINVAL;
  if ((((uintptr_t)out | (uintptr_t)base | (uintptr_t)rows0 | (uintptr_t)rows1 | (uintptr_t)rows2) & 15) != 0) return HIPAD_EINVAL;
  const long n4 = (long)bs * n_rows * (channels / 4);
  hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, base, rows0,
                     rows1, rows2, n4, n_rows, channels / 4);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_rows_sum(float *out, const float *x, int bs, int n_rows, int channels, hipad_stream_t stream) {
  if (!out || !x || bs <= 0 || n_rows <= 0 || channels <= 0) return HIPAD_EINVAL;
  hipLaunchKernelGGL(rows_sum_kernel, dim3((unsigned)((channels + 63) / 64), (unsigned)bs), dim3(1024), 0, (hipStream_t)stream,
                     out, x, n_rows, channels);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

}  // extern "C"
