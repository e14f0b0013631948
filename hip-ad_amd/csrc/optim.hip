// hip-ad_amd/csrc/optim.hip -- gradient clipping + AdamW over ONE flat parameter buffer.
//
// Replaces: the optimiser step of the reference's training loop -- torch AdamW (lr 2e-4, weight decay 1e-3,
// backbone lr x0.5; projects/configs/hipad_b2d_stage2.py:629-641) preceded by clip_grad_norm_(max_norm=25)
// (optimizer_config.grad_clip), which mmcv's OptimizerHook runs per parameter tensor.  With ~2000 parameter
// tensors torch's multi-tensor path needs ~170 launches and 6.5 ms per step (rocprofv3, profiles/); here all
// parameters / gradients / moments are views into four flat fp32 buffers and a step is two launches that
// move the compulsory 32 bytes per element once (~0.7 ms for 97 M parameters at HBM speed).
//
//   sqnorm kernel : partial[b] = sum of g^2 over the block's span      (deterministic two-level sum)
//   adamw kernel  : every block re-reduces the partials (4 KiB, L2), derives the clip coefficient
//                   c = min(1, max_norm / (norm + 1e-6)) -- torch.nn.utils.clip_grad_norm_ -- and applies
//                       g' = c g ; p *= 1 - lr wd ; m = b1 m + (1 - b1) g' ; v = b2 v + (1 - b2) g'^2
//                       p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
//                   -- torch.optim.AdamW -- with lr = lr0 for elements < n_group0 and lr1 after, then
//                   optionally zeroes g for the next step.  The step count t lives on the device
//                   (incremented here), so the whole thing replays from a hipGraph.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "hipad.h"

namespace hipad {

constexpr int kNormBlocks = 1024;

__device__ __forceinline__ float block_sum_256(float v, float *sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) sh[wv] = v;
  __syncthreads();
  v = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return v;
}

__global__ __launch_bounds__(256) void grad_sqnorm_kernel(float *__restrict__ partial, const float *__restrict__ g,
                                                          long n4 /* float4 count */, long n) {
  __shared__ float sh[4];
  const float4 *g4 = reinterpret_cast<const float4 *>(g);
  float s = 0.f;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 x = g4[i];
    s += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
  }
  if (blockIdx.x == 0) {  // tail (n not a multiple of 4)
    for (long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) s += g[i] * g[i];
  }
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

struct AdamCoef {
  float lr0, lr1, beta1, beta2, eps, wd, max_norm;
  hipad_lr_schedule sched;
};

// Learning-rate factor of iteration `it` (0-based count of steps already taken), mmcv 1.7.1 LrUpdaterHook with
// by_epoch=False as the reference configures it (projects/configs/hipad_b2d_stage2.py:643-649 lr_config):
//   regular(it) = CosineAnnealing: target + 0.5 (1 - target) (1 + cos(pi it / max_iters)), target = min_lr_ratio
//   warm-up (it < warmup_iters, linear): regular(it) * (1 - (1 - it / warmup_iters) (1 - warmup_ratio))
// policy 0 = constant.  The same closed form is hipad_amd.optim.lr_factor (tests compare the two).
__device__ __host__ __forceinline__ float lr_factor(const hipad_lr_schedule &s, int it) {
  if (s.policy == 0) return 1.f;
  float f = 1.f;
  if (s.policy == 1 && s.max_iters > 0) {
    const float x = fminf((float)it / (float)s.max_iters, 1.f);
    f = s.min_lr_ratio + 0.5f * (1.f - s.min_lr_ratio) * (1.f + cosf(3.14159265358979323846f * x));
  }
  if (s.warmup_iters > 0 && it < s.warmup_iters) {
    const float k = (1.f - (float)it / (float)s.warmup_iters) * (1.f - s.warmup_ratio);
    f *= 1.f - k;
  }
  return f;
}

__device__ __forceinline__ void adam_elem(float &p, float &g, float &m, float &v, float lr, float c, float b1, float b2,
                                          float eps, float wd, float inv_bc1, float inv_sqrt_bc2) {
  const float gg = g * c;
  p *= 1.f - lr * wd;
  m = b1 * m + (1.f - b1) * gg;
  v = b2 * v + (1.f - b2) * gg * gg;
  const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
  p -= lr * inv_bc1 * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_flat_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                         float *__restrict__ v, long n, long n_group0,
                                                         const float *__restrict__ partial, int *__restrict__ step,
                                                         float *__restrict__ norm_out, unsigned short *__restrict__ shadow,
                                                         AdamCoef k, int zero_grad) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < kNormBlocks; i += 256) s += partial[i];
  s = block_sum_256(s, sh);
  const float norm = sqrtf(s);
  float c = 1.f;
  if (k.max_norm > 0.f) c = fminf(1.f, k.max_norm / (norm + 1e-6f));
  const int t = *step + 1;  // every block reads the same (old) value; block 0 publishes the new one last
  const float lrf = lr_factor(k.sched, t - 1);
  k.lr0 *= lrf;
  k.lr1 *= lrf;
  const float inv_bc1 = 1.f / (1.f - powf(k.beta1, (float)t));
  const float inv_sqrt_bc2 = 1.f / sqrtf(1.f - powf(k.beta2, (float)t));
  const long n4 = n >> 2;
  float4 *p4 = reinterpret_cast<float4 *>(p), *g4 = reinterpret_cast<float4 *>(g);
  float4 *m4 = reinterpret_cast<float4 *>(m), *v4 = reinterpret_cast<float4 *>(v);
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 P = p4[i], G = g4[i], Mv = m4[i], V = v4[i];
    const long e = i * 4;
    const float l0 = e + 0 < n_group0 ? k.lr0 : k.lr1, l1 = e + 1 < n_group0 ? k.lr0 : k.lr1;
    const float l2 = e + 2 < n_group0 ? k.lr0 : k.lr1, l3 = e + 3 < n_group0 ? k.lr0 : k.lr1;
    adam_elem(P.x, G.x, Mv.x, V.x, l0, c, k.beta1, k.beta2, k.eps, k.wd, inv_bc1, inv_sqrt_bc2);
    adam_elem(P.y, G.y, Mv.y, V.y, l1, c, k.beta1, k.beta2, k.eps, k.wd, inv_bc1, inv_sqrt_bc2);
    adam_elem(P.z, G.z, Mv.z, V.z, l2, c, k.beta1, k.beta2, k.eps, k.wd, inv_bc1, inv_sqrt_bc2);
    adam_elem(P.w, G.w, Mv.w, V.w, l3, c, k.beta1, k.beta2, k.eps, k.wd, inv_bc1, inv_sqrt_bc2);
    p4[i] = P; m4[i] = Mv; v4[i] = V;
    if (shadow) {  // bf16 copy of the updated parameters (operand format of the MFMA kernels), same element order
      const __bf16 b0 = (__bf16)P.x, b1 = (__bf16)P.y, b2 = (__bf16)P.z, b3 = (__bf16)P.w;
      ushort4 sv;
      sv.x = __builtin_bit_cast(unsigned short, b0); sv.y = __builtin_bit_cast(unsigned short, b1);
      sv.z = __builtin_bit_cast(unsigned short, b2); sv.w = __builtin_bit_cast(unsigned short, b3);
      reinterpret_cast<ushort4 *>(shadow)[i] = sv;
    }
    if (zero_grad) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (blockIdx.x == 0) {
    for (long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
      float P = p[i], G = g[i], Mv = m[i], V = v[i];
      adam_elem(P, G, Mv, V, i < n_group0 ? k.lr0 : k.lr1, c, k.beta1, k.beta2, k.eps, k.wd, inv_bc1, inv_sqrt_bc2);
      p[i] = P; m[i] = Mv; v[i] = V;
      if (shadow) shadow[i] = __builtin_bit_cast(unsigned short, (__bf16)P);
      if (zero_grad) g[i] = 0.f;
    }
  }
  // the step counter is bumped by a trailing one-thread kernel (adamw_bump_kernel): blocks of THIS grid may
  // still be reading *step when block 0 gets here
  if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) {
    norm_out[0] = norm;
    norm_out[1] = k.lr0;  // the learning rate of group 0 this step actually used
  }
}

// bf16 copy of a flat fp32 buffer (initial fill of the shadow; afterwards adamw_flat_kernel maintains it)
__global__ __launch_bounds__(256) void shadow_bf16_kernel(unsigned short *__restrict__ dst, const float *__restrict__ src, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dst[i] = __builtin_bit_cast(unsigned short, (__bf16)src[i]);
}

__global__ void adamw_bump_kernel(int *step) { *step += 1; }

// bf16 copies of many fp32 matrices src = [R][C] in the MLP-chain operand layout (include/hipad.h): MFMA-fragment order.
//   P(A [rows][depth]) : blocks of 16 rows x 32 depth, block (tr, s) at ((tr * S + s) * 512) elements, S = ceil(depth / 32);
//                        inside a block lane = 16 * quad + l15 owns 8 consecutive elements A[16 tr + l15][32 s + 8 quad + j]
//   dst = P(src) (rows = R, depth = C);  dst_t = P(src^T) (rows = C, depth = R); out-of-range elements are zero.
// A wave then loads one block with ONE fully coalesced 1 KiB instruction (16 bytes per lane) straight into an MFMA B
// fragment.  One 32 x 32 tile of src per workgroup of 256 threads (two blocks of each output).
__global__ __launch_bounds__(256) void pack_weights_kernel(unsigned short *const *__restrict__ dst,
                                                           unsigned short *const *__restrict__ dst_t,
                                                           const float *const *__restrict__ src, const int *__restrict__ rows,
                                                           const int *__restrict__ cols, const int *__restrict__ tile_start,
                                                           int n_mats) {
  __shared__ unsigned short T[32][34];
  const int b = blockIdx.x;
  int lo = 0, hi = n_mats - 1;  // last matrix whose first tile is <= b
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tile_start[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const int m = lo, R = rows[m], C = cols[m];
  const int tc = (C + 31) >> 5, t = b - tile_start[m];
  const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
  const float *s = src[m];
  unsigned short *d = dst ? dst[m] : nullptr, *dt = dst_t ? dst_t[m] : nullptr;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    unsigned short v = 0;
    if (r < R && c < C) v = __builtin_bit_cast(unsigned short, (__bf16)s[(size_t)r * C + c]);
    T[ty + 8 * i][tx] = v;
  }
  __syncthreads();
  const int h = threadIdx.x >> 7, lane = (threadIdx.x >> 1) & 63, half = threadIdx.x & 1;
  const int l15 = lane & 15, quad = lane >> 4;
  if (d) {  // P(src): row tiles over R, steps over C
    const int TR = (R + 15) >> 4, S = (C + 31) >> 5, tr = (r0 >> 4) + h, st = c0 >> 5;
    if (tr < TR) {
      ushort4 o;
      const unsigned short *q = &T[16 * h + l15][8 * quad + 4 * half];
      o.x = q[0]; o.y = q[1]; o.z = q[2]; o.w = q[3];
      *reinterpret_cast<ushort4 *>(d + ((size_t)(tr * S + st) * 64 + lane) * 8 + 4 * half) = o;
    }
  }
  if (dt) {  // P(src^T): row tiles over C, steps over R
    const int TR = (C + 15) >> 4, S = (R + 31) >> 5, tr = (c0 >> 4) + h, st = r0 >> 5;
    if (tr < TR) {
      ushort4 o;
      const int rr = 8 * quad + 4 * half, cc = 16 * h + l15;
      o.x = T[rr + 0][cc]; o.y = T[rr + 1][cc]; o.z = T[rr + 2][cc]; o.w = T[rr + 3][cc];
      *reinterpret_cast<ushort4 *>(dt + ((size_t)(tr * S + st) * 64 + lane) * 8 + 4 * half) = o;
    }
  }
}

// dst[k] (fp32, contiguous) += src[k] (bf16, 4-D with element strides) for up to HIPAD_ACC_MAX tensors in ONE launch: the
// table travels by value in the kernel arguments (safe under hipGraph capture: no upload).  1024 elements per workgroup.
struct AccTable {
  float *dst[HIPAD_ACC_MAX];
  const unsigned short *src[HIPAD_ACC_MAX];
  int dims[HIPAD_ACC_MAX][3];      // sizes of dims 1..3 (dim 0 follows from the element count)
  int strides[HIPAD_ACC_MAX][4];   // source element strides
  int numel[HIPAD_ACC_MAX];
  int block_start[HIPAD_ACC_MAX + 1];
  int n;
};

__global__ __launch_bounds__(256) void accumulate_bf16_kernel(const AccTable t) {
  __shared__ int bs_s[HIPAD_ACC_MAX + 1];
  for (int i = threadIdx.x; i <= t.n; i += blockDim.x) bs_s[i] = t.block_start[i];
  __syncthreads();
  const int b = blockIdx.x;
  int lo = 0, hi = t.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (bs_s[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const int k = lo;
  const int d1 = t.dims[k][0], d2 = t.dims[k][1], d3 = t.dims[k][2];
  const int s0 = t.strides[k][0], s1 = t.strides[k][1], s2 = t.strides[k][2], s3 = t.strides[k][3];
  float *dst = t.dst[k];
  const unsigned short *src = t.src[k];
  const int n = t.numel[k];
  const int base = (b - bs_s[k]) * 1024;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = base + j * 256 + threadIdx.x;
    if (i < n) {
      const int i3 = i % d3, r3 = i / d3;
      const int i2 = r3 % d2, r2 = r3 / d2;
      const int i1 = r2 % d1, i0 = r2 / d1;
      const unsigned short v = src[(long)i0 * s0 + (long)i1 * s1 + (long)i2 * s2 + (long)i3 * s3];
      dst[i] += __uint_as_float((unsigned)v << 16);
    }
  }
}

}  // namespace hipad

using namespace hipad;

extern "C" {

size_t hipad_adamw_workspace(void) { return kNormBlocks * sizeof(float); }

float hipad_lr_factor(const hipad_lr_schedule *sched, int iteration) {
  if (!sched) return 1.f;
  return lr_factor(*sched, iteration);
}

int hipad_pack_weights(unsigned short *const *dst, unsigned short *const *dst_t, const float *const *src, const int *rows,
                       const int *cols, const int *tile_start, int n_mats, int total_tiles, hipad_stream_t stream) {
  if ((!dst && !dst_t) || !src || !rows || !cols || !tile_start || n_mats <= 0 || total_tiles <= 0) return HIPAD_EINVAL;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, dst, dst_t, src, rows, cols,
                     tile_start, n_mats);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_shadow_bf16(unsigned short *dst, const float *src, long long n, hipad_stream_t stream) {
  if (!dst || !src || n <= 0) return HIPAD_EINVAL;
  hipLaunchKernelGGL(shadow_bf16_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, dst, src, (long)n);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_adamw_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, long long n, long long n_group0,
                     float lr0, float lr1, float beta1, float beta2, float eps, float weight_decay, float max_norm,
                     int *step_dev, float *norm_out_dev, void *workspace, size_t workspace_bytes, int zero_grad,
                     const hipad_lr_schedule *sched, unsigned short *shadow_bf16, hipad_stream_t stream_) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !step_dev || n <= 0) return HIPAD_EINVAL;
  if (!workspace || workspace_bytes < hipad_adamw_workspace()) return HIPAD_EWORKSPACE;
  if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0) return HIPAD_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  float *partial = (float *)workspace;
  if (shadow_bf16 && ((uintptr_t)shadow_bf16 & 7) != 0) return HIPAD_EINVAL;   // every argument check before the first launch
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(kNormBlocks), dim3(256), 0, stream, partial, grad, (long)(n >> 2), (long)n);
  AdamCoef k{lr0, lr1, beta1, beta2, eps, weight_decay, max_norm, {0, 0, 1.f, 0, 1.f}};
  if (sched) k.sched = *sched;
  hipLaunchKernelGGL(adamw_flat_kernel, dim3(2048), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq, (long)n,
                     (long)n_group0, (const float *)partial, step_dev, norm_out_dev, shadow_bf16, k, zero_grad);
  hipLaunchKernelGGL(adamw_bump_kernel, dim3(1), dim3(1), 0, stream, step_dev);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_accumulate_bf16(const hipad_acc_item *items, int n_items, hipad_stream_t stream) {
  if (!items || n_items <= 0 || n_items > HIPAD_ACC_MAX) return HIPAD_EINVAL;
  AccTable t;
  memset(&t, 0, sizeof(t));
  long long blocks = 0;
  for (int k = 0; k < n_items; ++k) {
    const hipad_acc_item &it = items[k];
    if (!it.dst || !it.src) return HIPAD_EINVAL;
    long long numel = 1;
    for (int d = 0; d < 4; ++d) {
      if (it.sizes[d] <= 0 || it.strides[d] < 0) return HIPAD_EINVAL;
      numel *= it.sizes[d];
      if (numel > 0x7fffffffll || (long long)(it.sizes[d] - 1) * it.strides[d] > 0x7fffffffll) return HIPAD_ERANGE;
    }
    if (((uintptr_t)it.dst & 3) || ((uintptr_t)it.src & 1)) return HIPAD_EINVAL;
    t.dst[k] = it.dst;
    t.src[k] = it.src;
    for (int d = 0; d < 3; ++d) t.dims[k][d] = it.sizes[d + 1];
    for (int d = 0; d < 4; ++d) t.strides[k][d] = it.strides[d];
    t.numel[k] = (int)numel;
    t.block_start[k] = (int)blocks;
    blocks += (numel + 1023) / 1024;
    if (blocks > 0x7fffffffll) return HIPAD_ERANGE;
  }
  t.block_start[n_items] = (int)blocks;
  t.n = n_items;
  hipLaunchKernelGGL(accumulate_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

}  // extern "C"
