// hip-ad_amd/csrc/losses.hip -- sigmoid focal loss, value and logit gradient in one pass.
//
// Replaces: mmdet==2.28.2 FocalLoss(use_sigmoid=True) as the reference's loss() calls it for the det / map / motion /
// plan class heads (models/sparse_onedecoder.py:1146, 1201, 1302, 1360-1362): in torch ops that is one_hot, sigmoid,
// two pt terms, pow, bce-with-logits, three multiplies and a reduction forward and as many again backward, per call.
//   loss[n, c] = bce(x, t) * (alpha t + (1 - alpha)(1 - t)) * pt^gamma * w[n],   t = [target[n] == c],
//   pt = (1 - p) t + p (1 - t),  p = sigmoid(x)
// reduced per decoder layer (rows are layer-major): sum / (avg[l] + eps) when avg is given, else the mean.
// The same pass writes d(loss_l)/dx into grad_logits (the backward is then one scaling by the upstream gradient).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hipad.h"
#include "daf_common.h"

namespace hipad {

__global__ __launch_bounds__(256) void focal_loss_kernel(float *__restrict__ loss /* [L], zero on entry */,
                                                         float *__restrict__ grad /* [N*C] */, const float *__restrict__ x,
                                                         const long long *__restrict__ target, const float *__restrict__ w,
                                                         const float *__restrict__ avg /* [L] or NULL */, long n_elems, int C,
                                                         long elems_per_layer, float alpha, float gamma, float eps) {
  __shared__ float sh[4];
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float val = 0.f;
  int layer = 0;
  if (i < n_elems) {
    layer = (int)(i / elems_per_layer);
    const long n = i / C;
    const int c = (int)(i - n * C);
    const float xv = x[i];
    const bool t = target[n] == c;
    const float p = 1.f / (1.f + expf(-xv));
    const float scale = (w ? w[n] : 1.f) / (avg ? (avg[layer] + eps) : (float)elems_per_layer);
    // bce with logits: max(x, 0) - x t + log(1 + exp(-|x|))
    const float bce = fmaxf(xv, 0.f) - (t ? xv : 0.f) + log1pf(expf(-fabsf(xv)));
    float l, g;
    if (t) {
      const float q = powf(1.f - p, gamma);
      l = alpha * q * bce;
      g = alpha * q * (gamma * p * logf(fmaxf(p, 1e-38f)) - (1.f - p));
    } else {
      const float q = powf(p, gamma);
      l = (1.f - alpha) * q * bce;
      g = (1.f - alpha) * q * (gamma * (1.f - p) * bce + p);   // -log(1 - p) = bce for t = 0
    }
    val = l * scale;
    grad[i] = g * scale;
  }
  // one atomic per block when the whole block lies in one layer, one per wavefront when only the wavefront does, one per
  // element for the few wavefronts that straddle a layer boundary (element-wise atomics on `layers` addresses serialise:
  // measured 170 us for 48 600 elements before this)
  const long last = n_elems - 1;
  const long wave_first = i - (threadIdx.x & 63);
  const long block_first = (long)blockIdx.x * blockDim.x;
  const int wl0 = (int)((wave_first < last ? wave_first : last) / elems_per_layer);
  const int wl1 = (int)((wave_first + 63 < last ? wave_first + 63 : last) / elems_per_layer);
  const int bl0 = (int)((block_first < last ? block_first : last) / elems_per_layer);
  const int bl1 = (int)((block_first + 255 < last ? block_first + 255 : last) / elems_per_layer);
  const bool block_uniform = bl0 == bl1;          // uniform across the block: the barrier below is safe
  float wsum = val;
  if (wl0 == wl1) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wsum += __shfl_xor(wsum, o);
  }
  if (block_uniform) {
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = wsum;
    __syncthreads();
    if (threadIdx.x == 0 && block_first < n_elems) atomicAdd(loss + bl0, sh[0] + sh[1] + sh[2] + sh[3]);
  } else if (wl0 == wl1) {
    if ((threadIdx.x & 63) == 0 && wave_first < n_elems) atomicAdd(loss + wl0, wsum);
  } else if (i < n_elems) {
    atomicAdd(loss + layer, val);
  }
}

}  // namespace hipad

using namespace hipad;

extern "C" {

int hipad_focal_loss_forward(float *loss_per_layer, float *grad_logits, const float *logits, const long long *target,
                             const float *weight, const float *avg_factor, long long rows, int num_classes, int layers,
                             float alpha, float gamma, hipad_stream_t stream_) {
  if (!loss_per_layer || !grad_logits || !logits || !target) return HIPAD_EINVAL;
  if (rows <= 0 || num_classes <= 0 || layers <= 0 || rows % layers) return HIPAD_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  const long n = (long)rows * num_classes;
  if (fill_zero(loss_per_layer, (size_t)layers * sizeof(float), stream) != HIPAD_OK) return HIPAD_ELAUNCH;
  hipLaunchKernelGGL(focal_loss_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, loss_per_layer, grad_logits,
                     logits, target, weight, avg_factor, n, num_classes, n / layers, alpha, gamma, 1.1920929e-07f);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

}  // extern "C"
