// hip-ad_amd/csrc/imgpipe.hip -- the image leg of the training data pipeline on the device.
//
// Replaces (reference, per camera image and on the host, in PIL / cv2):
//   datasets/pipelines/augment.py:46-68   ResizeCropFlipImage._img_transform: Image.resize (bicubic, antialiased) ->
//                                         crop -> transpose(FLIP_LEFT_RIGHT) -> rotate (nearest) -> float32
//   datasets/pipelines/transform.py:286-321  NormalizeMultiviewImage (mmcv.imnormalize: BGR -> RGB, (x - mean) / std)
//   datasets/pipelines/transform.py:136-138  NuScenesSparse4DAdaptor's HWC -> CHW transpose + stack
// with two launches for all camera images of a sample: the raw uint8 frames (n, H, W, 3) stay in HBM, the horizontal
// resampling pass writes only the rows the vertical pass will read, and the second kernel walks the OUTPUT pixels:
// rotate / flip / crop are index arithmetic, the vertical pass is evaluated for exactly the pixels that survive, and the
// normalised value is stored in whatever layout the caller's strides describe (CHW fp32 for the reference's tensor,
// channels-last for the encoder).
//
// Bit-exactness with Pillow is the parity bar for the geometry (integer / byte work): the tap tables are computed on
// the host in double precision by the same expressions as Pillow's precompute_coeffs / normalize_coeffs_8bpc
// (src/libImaging/Resample.c) and Image.rotate + affine_fixed (PIL/Image.py, src/libImaging/Geometry.c); the kernels
// only do the 22-bit fixed-point dot products and the 16.16 index walk.  HBM-bound: 3 bytes read per source pixel of the
// used rows, 12 bytes written per output pixel.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "hipad.h"

namespace hipad {

constexpr int kPrecisionBits = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) {
  v >>= kPrecisionBits;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// tmp[(img, r, x, c)] = horizontal pass of source row row0 + r; one thread per (img, r, x), three channels
__global__ __launch_bounds__(256) void resize_h_kernel(uint8_t *__restrict__ tmp, const uint8_t *__restrict__ src,
                                                       const int *__restrict__ bounds, const int *__restrict__ coeffs,
                                                       int ksize, int n_img, int src_h, int src_w, int row0, int rows,
                                                       int out_w) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)n_img * rows * out_w;
  if (idx >= total) return;
  const int x = (int)(idx % out_w);
  const long ir = idx / out_w;
  const int r = (int)(ir % rows), img = (int)(ir / rows);
  const int first = bounds[2 * x], n = bounds[2 * x + 1];
  const int *k = coeffs + (long)x * ksize;
  const uint8_t *p = src + (((long)img * src_h + (row0 + r)) * src_w + first) * 3;
  int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
  for (int t = 0; t < n; ++t) {
    const int w = k[t];
    s0 += (int)p[3 * t] * w;
    s1 += (int)p[3 * t + 1] * w;
    s2 += (int)p[3 * t + 2] * w;
  }
  uint8_t *o = tmp + idx * 3;
  o[0] = (uint8_t)clip8(s0);
  o[1] = (uint8_t)clip8(s1);
  o[2] = (uint8_t)clip8(s2);
}

struct FinishArgs {
  int n_img, tmp_rows, res_w, res_h;  // tmp: (n_img, tmp_rows, res_w, 3); the resized image is res_h x res_w
  int ksize_v;
  int out_h, out_w;                   // final size = the crop box's size
  int crop_x, crop_y, flip;
  int rotate_on, a0, a1, a2, a3, a4, a5;
  int to_rgb, normalize;
  float mean[3], stdinv[3];
  long stride_n, stride_c, stride_y, stride_x;   // element strides of the output
};

// one thread per output pixel (img, y, x): inverse rotate -> flip -> crop -> vertical pass -> normalise -> store
__global__ __launch_bounds__(256) void finish_kernel(float *__restrict__ out, const uint8_t *__restrict__ tmp,
                                                     const int *__restrict__ bounds_v /* first already minus row0 */,
                                                     const int *__restrict__ coeffs_v, FinishArgs a) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)a.n_img * a.out_h * a.out_w;
  if (idx >= total) return;
  const int x = (int)(idx % a.out_w);
  const long iy = idx / a.out_w;
  const int y = (int)(iy % a.out_h), img = (int)(iy / a.out_h);
  int px = x, py = y;
  bool inside = true;
  if (a.rotate_on) {
    // Pillow walks xx += a0 per pixel and a2 += a1 per row in 32-bit ints: the closed form below is the same value
    // as long as nothing overflows, which check_fixed() guarantees on the host
    px = (a.a2 + a.a1 * y + a.a0 * x) >> 16;
    py = (a.a5 + a.a4 * y + a.a3 * x) >> 16;
    inside = px >= 0 && px < a.out_w && py >= 0 && py < a.out_h;
  }
  int v0 = 0, v1 = 0, v2 = 0;
  if (inside) {
    if (a.flip) px = a.out_w - 1 - px;
    const int rx = px + a.crop_x, ry = py + a.crop_y;
    if (rx >= 0 && rx < a.res_w && ry >= 0 && ry < a.res_h) {
      const int first = bounds_v[2 * ry], n = bounds_v[2 * ry + 1];
      const int *k = coeffs_v + (long)ry * a.ksize_v;
      const uint8_t *p = tmp + (((long)img * a.tmp_rows + first) * a.res_w + rx) * 3;
      const long pitch = (long)a.res_w * 3;
      int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
      for (int t = 0; t < n; ++t) {
        const int w = k[t];
        s0 += (int)p[0] * w;
        s1 += (int)p[1] * w;
        s2 += (int)p[2] * w;
        p += pitch;
      }
      v0 = clip8(s0); v1 = clip8(s1); v2 = clip8(s2);
    }
  }
  float f[3] = {(float)v0, (float)v1, (float)v2};
  if (a.to_rgb) { const float t = f[0]; f[0] = f[2]; f[2] = t; }
  float *o = out + (long)img * a.stride_n + (long)y * a.stride_y + (long)x * a.stride_x;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = f[c];
    if (a.normalize) v = __fmul_rn(__fsub_rn(v, a.mean[c]), a.stdinv[c]);
    o[c * a.stride_c] = v;
  }
}

static double bicubic(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

}  // namespace hipad

using namespace hipad;

extern "C" {

int hipad_resample_tables(int in_size, int out_size, int *bounds, int *coeffs) {
  if (in_size <= 0 || out_size <= 0) return HIPAD_EINVAL;
  if (in_size == out_size) {  // Pillow skips the pass: identity taps
    if (bounds && coeffs)
      for (int i = 0; i < out_size; ++i) {
        bounds[2 * i] = i;
        bounds[2 * i + 1] = 1;
        coeffs[i] = 1 << kPrecisionBits;
      }
    return 1;
  }
  const double scale = (double)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  if (!bounds || !coeffs) return ksize;
  const double ss = 1.0 / filterscale;
  double *w = new double[ksize];
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      w[x] = bicubic((x + xmin - center + 0.5) * ss);
      ww += w[x];
    }
    int *k = coeffs + (long)xx * ksize;
    for (int x = 0; x < ksize; ++x) {
      double v = 0.0;
      if (x < xmax) v = ww != 0.0 ? w[x] / ww : w[x];
      k[x] = v < 0 ? (int)(-0.5 + v * (1 << kPrecisionBits)) : (int)(0.5 + v * (1 << kPrecisionBits));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  delete[] w;
  return ksize;
}

int hipad_rotate_fixed(double angle_deg, int width, int height, int *a_out) {
  if (!a_out || width <= 0 || height <= 0 || !(angle_deg == angle_deg)) return HIPAD_EINVAL;
  double angle = fmod(angle_deg, 360.0);
  if (angle < 0) angle += 360.0;      // Python's % for a positive modulus
  if (angle == 0.0) return 0;         // Image.rotate returns a copy
  if (angle == 180.0 || ((angle == 90.0 || angle == 270.0) && width == height)) return HIPAD_ERANGE;  // PIL's transpose paths
  const double cx = width / 2.0, cy = height / 2.0;
  const double r = -(angle * (3.141592653589793 / 180.0));
  // Python's round(v, 15): the correctly rounded 15-decimal string, parsed back
  auto round15 = [](double v) {
    char buf[64];
    snprintf(buf, sizeof buf, "%.15f", v);
    return strtod(buf, nullptr);
  };
  double m[6] = {round15(cos(r)), round15(sin(r)), 0.0, round15(-sin(r)), round15(cos(r)), 0.0};
  m[2] = m[0] * -cx + m[1] * -cy + m[2];
  m[5] = m[3] * -cx + m[4] * -cy + m[5];
  m[2] += cx;
  m[5] += cy;
  auto fix = [](double v) {
    v = v * 65536.0 + 0.5;
    return v < 0.0 ? (int)floor(v) : (int)v;
  };
  // Pillow's check_fixed: the four corners must stay inside the 16.16 range, else it leaves the fixed-point path
  const double lim = 32768.0;
  const double xs[2] = {0.0, (double)width}, ys[2] = {0.0, (double)height};
  for (double x : xs)
    for (double y : ys) {
      const double u = m[0] * x + m[1] * y + m[2], v = m[3] * x + m[4] * y + m[5];
      if (fabs(u) >= lim || fabs(v) >= lim) return HIPAD_ERANGE;
    }
  a_out[0] = fix(m[0]);
  a_out[1] = fix(m[1]);
  a_out[2] = fix(m[2] + m[0] * 0.5 + m[1] * 0.5);
  a_out[3] = fix(m[3]);
  a_out[4] = fix(m[4]);
  a_out[5] = fix(m[5] + m[3] * 0.5 + m[4] * 0.5);
  return 1;
}

int hipad_image_resize_rows(unsigned char *tmp, const unsigned char *src, const int *bounds_h, const int *coeffs_h,
                            int ksize_h, int n_img, int src_h, int src_w, int row0, int rows, int out_w,
                            hipad_stream_t stream) {
  if (!tmp || !src || !bounds_h || !coeffs_h) return HIPAD_EINVAL;
  if (n_img <= 0 || src_h <= 0 || src_w <= 0 || out_w <= 0 || ksize_h <= 0) return HIPAD_EINVAL;
  if (row0 < 0 || rows <= 0 || row0 + rows > src_h) return HIPAD_ERANGE;
  const long total = (long)n_img * rows * out_w;
  hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tmp, src,
                     bounds_h, coeffs_h, ksize_h, n_img, src_h, src_w, row0, rows, out_w);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_image_finish(float *out, const long long *out_strides, const unsigned char *tmp, const int *bounds_v,
                       const int *coeffs_v, int ksize_v, int n_img, int tmp_rows, int resized_w, int resized_h,
                       const int *crop_box, int flip, const int *rotate_a, const float *mean, const float *std, int to_rgb,
                       hipad_stream_t stream) {
  if (!out || !out_strides || !tmp || !bounds_v || !coeffs_v || !crop_box) return HIPAD_EINVAL;
  if (n_img <= 0 || tmp_rows <= 0 || resized_w <= 0 || resized_h <= 0 || ksize_v <= 0) return HIPAD_EINVAL;
  if ((mean == nullptr) != (std == nullptr)) return HIPAD_EINVAL;
  FinishArgs a;
  a.n_img = n_img; a.tmp_rows = tmp_rows; a.res_w = resized_w; a.res_h = resized_h;
  a.ksize_v = ksize_v;
  a.out_w = crop_box[2] - crop_box[0];
  a.out_h = crop_box[3] - crop_box[1];
  if (a.out_w <= 0 || a.out_h <= 0) return HIPAD_EINVAL;
  a.crop_x = crop_box[0]; a.crop_y = crop_box[1]; a.flip = flip ? 1 : 0;
  a.rotate_on = rotate_a ? 1 : 0;
  a.a0 = a.a1 = a.a2 = a.a3 = a.a4 = a.a5 = 0;
  if (rotate_a) { a.a0 = rotate_a[0]; a.a1 = rotate_a[1]; a.a2 = rotate_a[2]; a.a3 = rotate_a[3]; a.a4 = rotate_a[4]; a.a5 = rotate_a[5]; }
  a.to_rgb = to_rgb ? 1 : 0;
  a.normalize = mean ? 1 : 0;
  for (int c = 0; c < 3; ++c) {
    a.mean[c] = mean ? mean[c] : 0.f;
    a.stdinv[c] = std ? (float)(1.0 / (double)std[c]) : 1.f;   // mmcv.imnormalize: stdinv = 1 / float64(std), applied in float32
  }
  a.stride_n = out_strides[0]; a.stride_c = out_strides[1]; a.stride_y = out_strides[2]; a.stride_x = out_strides[3];
  const long total = (long)n_img * a.out_h * a.out_w;
  hipLaunchKernelGGL(finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, tmp,
                     bounds_v, coeffs_v, a);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

}  // extern "C"
