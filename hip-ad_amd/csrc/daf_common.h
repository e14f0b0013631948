// hip-ad_amd/csrc/daf_common.h -- device helpers shared by the aggregation kernels.
#ifndef HIPAD_DAF_COMMON_H_
#define HIPAD_DAF_COMMON_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wave_ops.h"

namespace hipad {

constexpr int kWave = 64;
constexpr int kMaxPairsPerWave = 128;  // two (point,camera) pairs per lane

// pixel coordinate exactly as deformable_aggregation_cuda.cu:180-181 computes it:
// fl32(loc * size) then one correctly rounded subtraction of 0.5 -- never an fma.
__device__ __forceinline__ float pix_coord(float loc, int size) {
#pragma clang fp contract(off)
  float prod = loc * (float)size;
  asm volatile("" : "+v"(prod));  // opaque to the optimiser: no contraction across it
  return prod - 0.5f;
}

__device__ __forceinline__ bool loc_kept(float lw, float lh) {
  // cu:168-171; written so that NaN is kept, as there
  return !(lw <= 0.f || lw >= 1.f || lh <= 0.f || lh >= 1.f);
}

struct Taps {
  int h_low, w_low;
  float lh, lw, hh, hw;
  bool in_h0, in_h1, in_w0, in_w1;
};

__device__ __forceinline__ Taps make_taps(float loc_h, float loc_w, int H, int W) {
  Taps t;
  const float h_im = pix_coord(loc_h, H);
  const float w_im = pix_coord(loc_w, W);
  t.h_low = (int)floorf(h_im);
  t.w_low = (int)floorf(w_im);
  t.lh = h_im - (float)t.h_low;
  t.lw = w_im - (float)t.w_low;
  t.hh = 1.f - t.lh;
  t.hw = 1.f - t.lw;
  t.in_h0 = t.h_low >= 0;
  t.in_h1 = t.h_low + 1 <= H - 1;
  t.in_w0 = t.w_low >= 0;
  t.in_w1 = t.w_low + 1 <= W - 1;
  return t;
}

// One 256-channel pyramid row, 4 channels per lane: fp32 rows are 1 KiB (one dwordx4 per lane), bf16 rows 512 B (one
// dwordx2 per lane, widened in registers -- exact, the values ARE bf16: the encoder's output dtype).
template <typename FT>
__device__ __forceinline__ float4 load_row4(const FT *__restrict__ feat, size_t row, int lane);
template <>
__device__ __forceinline__ float4 load_row4<float>(const float *__restrict__ feat, size_t row, int lane) {
  return *(reinterpret_cast<const float4 *>(feat + row * 256) + lane);
}
template <>
__device__ __forceinline__ float4 load_row4<uint16_t>(const uint16_t *__restrict__ feat, size_t row, int lane) {
  const uint2 u = *(reinterpret_cast<const uint2 *>(feat + row * 256) + lane);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xFFFF0000u));
}

__device__ __forceinline__ float4 sel4(bool c, float4 v) {
  return c ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}

// Item geometry shared by forward and backward.
struct Item {
  int anchor;     // b*A + a
  int b;
  int npairs;     // (points in chunk) * cams, <= 128
  long pair0;     // global index of the item's first (point,camera) pair
};

__device__ __forceinline__ Item make_item(int item, int nchunks, int ppc, int cams, int A, int P) {
  Item it;
  it.anchor = item / nchunks;
  const int chunk = item - it.anchor * nchunks;
  it.b = it.anchor / A;
  const int p0 = chunk * ppc;
  const int p1 = min(P, p0 + ppc);
  it.npairs = (p1 - p0) * cams;
  it.pair0 = ((long)it.anchor * P + p0) * cams;
  return it;
}


// Zero fill as a kernel: the library never uses hipMemsetAsync -- inside a captured hipGraph every node
// is then a kernel node in one dependency chain (memset nodes were the one graph-node type only this
// library produced; see DESIGN.md "hipGraph").
__global__ void fill_zero_kernel(uint32_t *__restrict__ p, size_t n);
int fill_zero(void *ptr, size_t bytes, hipStream_t stream);

// ---- sorted (atomic-free) feature-gradient path, daf_bwd_sorted.hip -------------------
struct DafDims {
  int bs, cams, num_feat, C, L, A, P, G;
};
size_t daf_bwd_sorted_workspace(const DafDims &d);
bool daf_bwd_sorted_supported(const DafDims &d);
// grad_feat += scatter of the bilinear taps; returns HIPAD_* status
int daf_bwd_sorted_feat(const float *feat_unused, const int *ss, const int *start, const float *loc,
                        const float *wts, const float *gout, float *gfeat, const DafDims &d,
                        void *workspace, size_t workspace_bytes, hipStream_t stream);

// grad_loc / grad_w by one wave per (anchor, chunk of points); no atomics
int daf_bwd_lw(const void *feat, bool feat_bf16, const int *ss, const int *start, const float *loc, const float *wts,
               const float *gout, float *gloc, float *gw, const DafDims &d, int nchunks, int ppc,
               bool overwrite, hipStream_t stream);

}  // namespace hipad
#endif
