// hip-ad_amd/csrc/daf.hip -- deformable multi-view multi-scale aggregation for gfx950 (MI355X).
//
// What it computes (reference: projects/mmdet3d_plugin/ops/src/deformable_aggregation_cuda.cu
// :129-187 forward, :190-262 backward):
//   out[b,a,c] = sum_{p,cam,s} w[b,a,p,cam,s,c/(C/G)] * bilinear(feat[b, start[cam,s]+.., c],
//                                                               loc[b,a,p,cam]*(W_s,H_s) - 0.5)
//   a (p,cam) sample is dropped iff loc_w<=0 || loc_w>=1 || loc_h<=0 || loc_h>=1.
//
// How (this is not the reference's decomposition):
//   * The reference runs one thread per (b,a,p,cam,scale,channel) and float-atomicAdds every
//     product into out[].  Here ONE WAVEFRONT owns one work item = (anchor, chunk of points):
//     64 lanes x float4 = the 256 channels of one pyramid position = one coalesced 1 KiB row
//     per bilinear corner.
//   * The wave first loads the <=128 (point,camera) locations of its item (one or two per
//     lane), ballots the keep-mask and then walks ONLY the kept pairs (typically 1 of 6
//     cameras sees a point); location / camera index travel by v_readlane, so all index
//     arithmetic is wave-uniform and row base addresses live in SGPRs.
//   * Accumulation is in registers in a fixed order; items of one anchor are combined by a
//     tiny second pass in chunk order: no atomics in the forward, bitwise reproducible.
//   * Backward: same ownership.  grad_weights is reduced over the 32 channels of a group with
//     DPP/shuffle butterflies and stored once; grad_loc is reduced over the wave and over the
//     levels and stored once; only grad_feat (a true scatter) uses fp32 atomics, laid out so
//     that one wave-instruction adds 256 contiguous bytes (the shape the memory-side atomic
//     units want: MI355X_MICROARCH.md "Global float atomics").
//   * Index arithmetic follows the reference bit for bit: float product, then subtraction of
//     0.5 with a single rounding, floorf, int offsets -- kept out of reach of FMA contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hipad.h"
#include "daf_common.h"

namespace hipad {

__global__ __launch_bounds__(256) void fill_zero_kernel(uint32_t *__restrict__ p, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0u;
}

int fill_zero(void *ptr, size_t bytes, hipStream_t stream) {
  if (!ptr || bytes == 0) return HIPAD_OK;
  if (bytes & 3) return HIPAD_EINVAL;
  const size_t n = bytes >> 2;
  size_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (uint32_t *)ptr, n);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

// =====================================================================================
// Forward, fast path: C == 256, (C/G) % 4 == 0.
// grid = ceil(n_items / 4) blocks of 256 threads; one wave per item.
// dst = out (nchunks == 1) or partial slabs [n_items, 256] (nchunks > 1).
// =====================================================================================
// LT = compile-time number of levels (0 = runtime).  GEO: (height, width, first row) of every (camera, level) held in
// lanes 0..cams*L-1 and read back with v_readlane -- no scalar memory load between a pair's coordinates and its row
// loads (needs cams*L <= 64).
template <int LT, bool GEO, typename FT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) void daf_fwd_c256_kernel(
    float *__restrict__ dst, const FT *__restrict__ feat, const int *__restrict__ ss,
    const int *__restrict__ start, const float *__restrict__ loc, const float *__restrict__ wts,
    int n_items, int nchunks, int ppc, int cams, int num_feat, int L_rt, int A, int P, int G) {
  const int lane = threadIdx.x & (kWave - 1);
  const int item = uni(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  if (item >= n_items) return;
  const int L = LT ? LT : L_rt;
  const Item it = make_item(item, nchunks, ppc, cams, A, P);

  const float2 *loc2 = reinterpret_cast<const float2 *>(loc) + it.pair0;
  float2 l0 = make_float2(-1.f, -1.f), l1 = make_float2(-1.f, -1.f);
  if (lane < it.npairs) l0 = loc2[lane];
  if (lane + kWave < it.npairs) l1 = loc2[lane + kWave];
  const int cam0 = lane % cams, cam1 = (lane + kWave) % cams;
  const unsigned long long m0 = __ballot(lane < it.npairs && loc_kept(l0.x, l0.y));
  const unsigned long long m1 = __ballot(lane + kWave < it.npairs && loc_kept(l1.x, l1.y));

  const int lanes_per_group = (256 / G) >> 2;
  const int g = lane / lanes_per_group;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float *wbase = wts + (size_t)it.pair0 * L * G + g;
  const size_t frow0 = (size_t)it.b * num_feat;  // first pyramid row of this sample
  int geoH = 1, geoW = 1, geoS = 0;
  if (GEO && lane < cams * L) {
    geoH = ss[2 * lane];
    geoW = ss[2 * lane + 1];
    geoS = start[lane];
  }
  asm volatile("" : "+v"(geoH), "+v"(geoW), "+v"(geoS));  // the loads land here, not inside the pair loop

#pragma unroll
  for (int half = 0; half < 2; ++half) {
    unsigned long long m = half ? m1 : m0;
    while (m) {
      const int j = __builtin_ctzll(m);
      m &= m - 1;
      const float loc_w = rl_f(half ? l1.x : l0.x, j);
      const float loc_h = rl_f(half ? l1.y : l0.y, j);
      const int cam = rl_i(half ? cam1 : cam0, j);
      const int pidx = half * kWave + j;
      const float *wrow = wbase + (size_t)pidx * L * G;
#pragma unroll
      for (int s = 0; s < L; ++s) {
        const int cs = cam * L + s;
        const int H = GEO ? rl_i(geoH, cs) : ss[2 * cs], W = GEO ? rl_i(geoW, cs) : ss[2 * cs + 1];
        const Taps t = make_taps(loc_h, loc_w, H, W);
        // rows clamped into the map so every load is legal; out-of-map corners are zeroed
        const int h0 = max(t.h_low, 0), h1 = min(t.h_low + 1, H - 1);
        const int w0 = max(t.w_low, 0), w1 = min(t.w_low + 1, W - 1);
        const size_t base = frow0 + (size_t)(GEO ? rl_i(geoS, cs) : start[cs]);
        const float4 v1 = sel4(t.in_h0 && t.in_w0, load_row4<FT>(feat, base + (size_t)uni(h0 * W + w0), lane));
        const float4 v2 = sel4(t.in_h0 && t.in_w1, load_row4<FT>(feat, base + (size_t)uni(h0 * W + w1), lane));
        const float4 v3 = sel4(t.in_h1 && t.in_w0, load_row4<FT>(feat, base + (size_t)uni(h1 * W + w0), lane));
        const float4 v4 = sel4(t.in_h1 && t.in_w1, load_row4<FT>(feat, base + (size_t)uni(h1 * W + w1), lane));
        const float aw = wrow[s * G];
        const float w1c = t.hh * t.hw, w2c = t.hh * t.lw, w3c = t.lh * t.hw, w4c = t.lh * t.lw;
        acc.x += aw * (w1c * v1.x + w2c * v2.x + w3c * v3.x + w4c * v4.x);
        acc.y += aw * (w1c * v1.y + w2c * v2.y + w3c * v3.y + w4c * v4.y);
        acc.z += aw * (w1c * v1.z + w2c * v2.z + w3c * v3.z + w4c * v4.z);
        acc.w += aw * (w1c * v1.w + w2c * v2.w + w3c * v3.w + w4c * v4.w);
      }
    }
  }
  reinterpret_cast<float4 *>(dst + (size_t)item * 256)[lane] = acc;
}

// out[anchor, :] = sum over chunks (in chunk order) of partial[anchor*nchunks + k, :]
__global__ __launch_bounds__(256) void daf_fwd_combine_kernel(float *__restrict__ out,
                                                              const float *__restrict__ partial,
                                                              int n_anchor, int nchunks, int C4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (anchor, c4)
  if (i >= n_anchor * C4) return;
  const int a = i / C4, c4 = i - a * C4;
  const float4 *p = reinterpret_cast<const float4 *>(partial) + (size_t)a * nchunks * C4 + c4;
  float4 s = p[0];
  int k = 1;
  // eight partial rows in flight, added in chunk order (the map head has 38 chunks per anchor: one load per trip cost
  // 38 memory latencies, 12 us for 100 anchors)
  for (; k + 7 < nchunks; k += 8) {
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[(size_t)(k + j) * C4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s.x += v[j].x;
      s.y += v[j].y;
      s.z += v[j].z;
      s.w += v[j].w;
    }
  }
  for (; k < nchunks; ++k) {
    const float4 v = p[(size_t)k * C4];
    s.x += v.x;
    s.y += v.y;
    s.z += v.z;
    s.w += v.w;
  }
  reinterpret_cast<float4 *>(out)[i] = s;
}

// =====================================================================================
// Forward, generic path: any C, G (C % G == 0), L.  One wave per (item, block of 64 channels).
// Correct for every shape the reference accepts; not tuned.
// =====================================================================================
__global__ __launch_bounds__(256) void daf_fwd_generic_kernel(
    float *__restrict__ dst, const float *__restrict__ feat, const int *__restrict__ ss,
    const int *__restrict__ start, const float *__restrict__ loc, const float *__restrict__ wts,
    int n_items, int nchunks, int ppc, int cams, int num_feat, int C, int L, int A, int P, int G,
    int cblocks) {
  const int lane = threadIdx.x & (kWave - 1);
  const int widx = uni(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  if (widx >= n_items * cblocks) return;
  const int item = widx / cblocks;
  const int c = (widx - item * cblocks) * kWave + lane;
  const bool cok = c < C;
  const int cc = cok ? c : 0;
  const Item it = make_item(item, nchunks, ppc, cams, A, P);
  const float2 *loc2 = reinterpret_cast<const float2 *>(loc) + it.pair0;
  float2 l0 = make_float2(-1.f, -1.f), l1 = make_float2(-1.f, -1.f);
  if (lane < it.npairs) l0 = loc2[lane];
  if (lane + kWave < it.npairs) l1 = loc2[lane + kWave];
  const int cam0 = lane % cams, cam1 = (lane + kWave) % cams;
  const unsigned long long m0 = __ballot(lane < it.npairs && loc_kept(l0.x, l0.y));
  const unsigned long long m1 = __ballot(lane + kWave < it.npairs && loc_kept(l1.x, l1.y));
  const int g = cc / (C / G);
  float acc = 0.f;
  const float *wbase = wts + (size_t)it.pair0 * L * G + g;
  const size_t frow0 = (size_t)it.b * num_feat;
  for (int half = 0; half < 2; ++half) {
    unsigned long long m = half ? m1 : m0;
    while (m) {
      const int j = __builtin_ctzll(m);
      m &= m - 1;
      const float loc_w = rl_f(half ? l1.x : l0.x, j);
      const float loc_h = rl_f(half ? l1.y : l0.y, j);
      const int cam = rl_i(half ? cam1 : cam0, j);
      const int pidx = half * kWave + j;
      const float *wrow = wbase + (size_t)pidx * L * G;
      for (int s = 0; s < L; ++s) {
        const int cs = cam * L + s;
        const int H = ss[2 * cs], W = ss[2 * cs + 1];
        const Taps t = make_taps(loc_h, loc_w, H, W);
        const int h0 = max(t.h_low, 0), h1 = min(t.h_low + 1, H - 1);
        const int w0 = max(t.w_low, 0), w1 = min(t.w_low + 1, W - 1);
        const size_t base = frow0 + (size_t)start[cs];
        const float a1 = feat[(base + (size_t)(h0 * W + w0)) * C + cc];
        const float a2 = feat[(base + (size_t)(h0 * W + w1)) * C + cc];
        const float a3 = feat[(base + (size_t)(h1 * W + w0)) * C + cc];
        const float a4 = feat[(base + (size_t)(h1 * W + w1)) * C + cc];
        const float v1 = (t.in_h0 && t.in_w0) ? a1 : 0.f;
        const float v2 = (t.in_h0 && t.in_w1) ? a2 : 0.f;
        const float v3 = (t.in_h1 && t.in_w0) ? a3 : 0.f;
        const float v4 = (t.in_h1 && t.in_w1) ? a4 : 0.f;
        acc += wrow[s * G] * (t.hh * t.hw * v1 + t.hh * t.lw * v2 + t.lh * t.hw * v3 + t.lh * t.lw * v4);
      }
    }
  }
  if (cok) dst[(size_t)item * C + c] = acc;
}

__global__ __launch_bounds__(256) void daf_fwd_combine_generic_kernel(
    float *__restrict__ out, const float *__restrict__ partial, int n_anchor, int nchunks, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_anchor * C) return;
  const int a = i / C, c = i - a * C;
  const float *p = partial + (size_t)a * nchunks * C + c;
  float s = p[0];
  for (int k = 1; k < nchunks; ++k) s += p[(size_t)k * C];
  out[i] = s;
}

// =====================================================================================
// Backward, fast path: C == 256, G == 8 (32 channels per group).
// Lane l owns channels l, l+64, l+128, l+192 (register j <-> channel 64j + l), so that
//   - each feature / grad_feat wave-instruction touches 256 contiguous bytes,
//   - channel 64j + l belongs to group 2j + (l >> 5): a group is one 32-lane half of one
//     register, reduced with a 5-step butterfly.
// =====================================================================================
template <int LT, bool OVERWRITE>
__global__ __launch_bounds__(256) void daf_bwd_c256g8_kernel(
    const float *__restrict__ feat, const int *__restrict__ ss, const int *__restrict__ start,
    const float *__restrict__ loc, const float *__restrict__ wts, const float *__restrict__ gout,
    float *__restrict__ gfeat, float *__restrict__ gloc, float *__restrict__ gw, int n_items,
    int nchunks, int ppc, int cams, int num_feat, int L_rt, int A, int P) {
  constexpr int G = 8;
  const int lane = threadIdx.x & (kWave - 1);
  const int item = uni(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  if (item >= n_items) return;
  const int L = LT ? LT : L_rt;
  const Item it = make_item(item, nchunks, ppc, cams, A, P);

  const float2 *loc2 = reinterpret_cast<const float2 *>(loc) + it.pair0;
  float2 l0 = make_float2(-1.f, -1.f), l1 = make_float2(-1.f, -1.f);
  if (lane < it.npairs) l0 = loc2[lane];
  if (lane + kWave < it.npairs) l1 = loc2[lane + kWave];
  const int cam0 = lane % cams, cam1 = (lane + kWave) % cams;
  const unsigned long long m0 = __ballot(lane < it.npairs && loc_kept(l0.x, l0.y));
  const unsigned long long m1 = __ballot(lane + kWave < it.npairs && loc_kept(l1.x, l1.y));

  if (OVERWRITE) {
    // the kernel owns grad_loc / grad_w of its pairs: dropped pairs get zeros here, kept pairs
    // are written exactly once below (no address is stored twice)
    if (gloc) {
      float2 *g2 = reinterpret_cast<float2 *>(gloc) + it.pair0;
      if (lane < it.npairs && !((m0 >> lane) & 1ull)) g2[lane] = make_float2(0.f, 0.f);
      if (lane + kWave < it.npairs && !((m1 >> lane) & 1ull)) g2[lane + kWave] = make_float2(0.f, 0.f);
    }
    if (gw) {
      float4 *g4 = reinterpret_cast<float4 *>(gw + (size_t)it.pair0 * L * G);
      const int per_pair = L * (G / 4);
      const int n4 = it.npairs * per_pair;
      for (int i = lane; i < n4; i += kWave) {
        const int q = i / per_pair;
        const bool kept = q < kWave ? ((m0 >> q) & 1ull) : ((m1 >> (q - kWave)) & 1ull);
        if (!kept) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }

  float go[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) go[j] = gout[(size_t)it.anchor * 256 + 64 * j + lane];
  const int hsel = lane >> 5;
  const float *wbase = wts + (size_t)it.pair0 * L * G + hsel;
  const size_t frow0 = (size_t)it.b * num_feat;

#pragma unroll
  for (int half = 0; half < 2; ++half) {
    unsigned long long m = half ? m1 : m0;
    while (m) {
      const int jl = __builtin_ctzll(m);
      m &= m - 1;
      const float loc_w = rl_f(half ? l1.x : l0.x, jl);
      const float loc_h = rl_f(half ? l1.y : l0.y, jl);
      const int cam = rl_i(half ? cam1 : cam0, jl);
      const int pidx = half * kWave + jl;
      const size_t wofs = (size_t)pidx * L * G;
      float gl_w = 0.f, gl_h = 0.f;
#pragma unroll
      for (int s = 0; s < L; ++s) {
        const int cs = cam * L + s;
        const int H = ss[2 * cs], W = ss[2 * cs + 1];
        const Taps t = make_taps(loc_h, loc_w, H, W);
        const int h0 = max(t.h_low, 0), h1 = min(t.h_low + 1, H - 1);
        const int w0 = max(t.w_low, 0), w1 = min(t.w_low + 1, W - 1);
        const size_t base = frow0 + (size_t)start[cs];
        const size_t o00 = (base + (size_t)uni(h0 * W + w0)) * 256 + lane;
        const size_t o01 = (base + (size_t)uni(h0 * W + w1)) * 256 + lane;
        const size_t o10 = (base + (size_t)uni(h1 * W + w0)) * 256 + lane;
        const size_t o11 = (base + (size_t)uni(h1 * W + w1)) * 256 + lane;
        const bool i1 = t.in_h0 && t.in_w0, i2 = t.in_h0 && t.in_w1;
        const bool i3 = t.in_h1 && t.in_w0, i4 = t.in_h1 && t.in_w1;
        const float w1c = t.hh * t.hw, w2c = t.hh * t.lw, w3c = t.lh * t.hw, w4c = t.lh * t.lw;
        float gwp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v1 = i1 ? feat[o00 + 64 * j] : 0.f;
          const float v2 = i2 ? feat[o01 + 64 * j] : 0.f;
          const float v3 = i3 ? feat[o10 + 64 * j] : 0.f;
          const float v4 = i4 ? feat[o11 + 64 * j] : 0.f;
          const float aw = wbase[wofs + s * G + 2 * j];
          const float top = go[j] * aw;  // cu:86
          if (gfeat) {
            if (i1) atomicAdd(gfeat + o00 + 64 * j, w1c * top);
            if (i2) atomicAdd(gfeat + o01 + 64 * j, w2c * top);
            if (i3) atomicAdd(gfeat + o10 + 64 * j, w3c * top);
            if (i4) atomicAdd(gfeat + o11 + 64 * j, w4c * top);
          }
          // cu:92-118: d(val)/d(h_im), d(val)/d(w_im)
          const float gh = -t.hw * v1 - t.lw * v2 + t.hw * v3 + t.lw * v4;
          const float gwd = -t.hh * v1 + t.hh * v2 - t.lh * v3 + t.lh * v4;
          const float val = w1c * v1 + w2c * v2 + w3c * v3 + w4c * v4;
          gwp[j] = go[j] * val;          // cu:122
          gl_w += (float)W * gwd * top;  // cu:124
          gl_h += (float)H * gh * top;   // cu:125
        }
        if (gw) {
          // group 2j+h lives in half h of register j
          float mine[4], other[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            mine[j] = half_wave_sum(gwp[j]);
            other[j] = __shfl_xor(mine[j], 32);
          }
          // lane g (< 8) stores group g: j = g>>1, h = g&1; lanes 0..7 sit in half 0
          const int gj = lane >> 1;
          float lo = mine[0], hi = other[0];
          lo = gj == 1 ? mine[1] : lo; hi = gj == 1 ? other[1] : hi;
          lo = gj == 2 ? mine[2] : lo; hi = gj == 2 ? other[2] : hi;
          lo = gj == 3 ? mine[3] : lo; hi = gj == 3 ? other[3] : hi;
          const float sum = (lane & 1) ? hi : lo;
          if (lane < G) {
            float *dstw = gw + (size_t)it.pair0 * L * G + wofs + s * G + lane;
            *dstw = OVERWRITE ? sum : (*dstw + sum);
          }
        }
      }
      if (gloc) {
        const float sw = wave_sum(gl_w), sh = wave_sum(gl_h);
        if (lane == 0) {
          float2 *d = reinterpret_cast<float2 *>(gloc) + it.pair0 + pidx;
          if (OVERWRITE) {
            *d = make_float2(sw, sh);
          } else {
            float2 o = *d;
            *d = make_float2(o.x + sw, o.y + sh);
          }
        }
      }
    }
  }
}

// =====================================================================================
// Backward, generic path (any C, G, L): wave per (item, 64-channel block); all three
// gradients by atomics after a per-lane pass.  Correctness path, not tuned.
// =====================================================================================
__global__ __launch_bounds__(256) void daf_bwd_generic_kernel(
    const float *__restrict__ feat, const int *__restrict__ ss, const int *__restrict__ start,
    const float *__restrict__ loc, const float *__restrict__ wts, const float *__restrict__ gout,
    float *__restrict__ gfeat, float *__restrict__ gloc, float *__restrict__ gw, int n_items,
    int nchunks, int ppc, int cams, int num_feat, int C, int L, int A, int P, int G, int cblocks) {
  const int lane = threadIdx.x & (kWave - 1);
  const int widx = uni(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  if (widx >= n_items * cblocks) return;
  const int item = widx / cblocks;
  const int c = (widx - item * cblocks) * kWave + lane;
  const bool cok = c < C;
  const int cc = cok ? c : 0;
  const Item it = make_item(item, nchunks, ppc, cams, A, P);
  const float2 *loc2 = reinterpret_cast<const float2 *>(loc) + it.pair0;
  float2 l0 = make_float2(-1.f, -1.f), l1 = make_float2(-1.f, -1.f);
  if (lane < it.npairs) l0 = loc2[lane];
  if (lane + kWave < it.npairs) l1 = loc2[lane + kWave];
  const int cam0 = lane % cams, cam1 = (lane + kWave) % cams;
  const unsigned long long m0 = __ballot(lane < it.npairs && loc_kept(l0.x, l0.y));
  const unsigned long long m1 = __ballot(lane + kWave < it.npairs && loc_kept(l1.x, l1.y));
  const int g = cc / (C / G);
  const float go = cok ? gout[(size_t)it.anchor * C + c] : 0.f;
  const size_t frow0 = (size_t)it.b * num_feat;
  for (int half = 0; half < 2; ++half) {
    unsigned long long m = half ? m1 : m0;
    while (m) {
      const int j = __builtin_ctzll(m);
      m &= m - 1;
      const float loc_w = rl_f(half ? l1.x : l0.x, j);
      const float loc_h = rl_f(half ? l1.y : l0.y, j);
      const int cam = rl_i(half ? cam1 : cam0, j);
      const int pidx = half * kWave + j;
      const size_t wofs = ((size_t)it.pair0 + pidx) * L * G;
      float gl_w = 0.f, gl_h = 0.f;
      for (int s = 0; s < L; ++s) {
        const int cs = cam * L + s;
        const int H = ss[2 * cs], W = ss[2 * cs + 1];
        const Taps t = make_taps(loc_h, loc_w, H, W);
        const int h0 = max(t.h_low, 0), h1 = min(t.h_low + 1, H - 1);
        const int w0 = max(t.w_low, 0), w1 = min(t.w_low + 1, W - 1);
        const size_t base = frow0 + (size_t)start[cs];
        const size_t o00 = (base + (size_t)(h0 * W + w0)) * C + cc;
        const size_t o01 = (base + (size_t)(h0 * W + w1)) * C + cc;
        const size_t o10 = (base + (size_t)(h1 * W + w0)) * C + cc;
        const size_t o11 = (base + (size_t)(h1 * W + w1)) * C + cc;
        const bool i1 = t.in_h0 && t.in_w0, i2 = t.in_h0 && t.in_w1;
        const bool i3 = t.in_h1 && t.in_w0, i4 = t.in_h1 && t.in_w1;
        const float v1 = i1 ? feat[o00] : 0.f, v2 = i2 ? feat[o01] : 0.f;
        const float v3 = i3 ? feat[o10] : 0.f, v4 = i4 ? feat[o11] : 0.f;
        const float w1c = t.hh * t.hw, w2c = t.hh * t.lw, w3c = t.lh * t.hw, w4c = t.lh * t.lw;
        const float aw = wts[wofs + s * G + g];
        const float top = go * aw;
        if (cok) {
          if (gfeat) {
            if (i1) atomicAdd(gfeat + o00, w1c * top);
            if (i2) atomicAdd(gfeat + o01, w2c * top);
            if (i3) atomicAdd(gfeat + o10, w3c * top);
            if (i4) atomicAdd(gfeat + o11, w4c * top);
          }
          const float val = w1c * v1 + w2c * v2 + w3c * v3 + w4c * v4;
          if (gw) atomicAdd(gw + wofs + s * G + g, go * val);
          const float gh = -t.hw * v1 - t.lw * v2 + t.hw * v3 + t.lw * v4;
          const float gwd = -t.hh * v1 + t.hh * v2 - t.lh * v3 + t.lh * v4;
          gl_w += (float)W * gwd * top;
          gl_h += (float)H * gh * top;
        }
      }
      if (gloc) {
        const float sw = wave_sum(gl_w), sh = wave_sum(gl_h);
        if (lane == 0) {
          atomicAdd(gloc + 2 * ((size_t)it.pair0 + pidx), sw);
          atomicAdd(gloc + 2 * ((size_t)it.pair0 + pidx) + 1, sh);
        }
      }
    }
  }
}

// =====================================================================================
// Index work only (bit-exact class): one thread per (b,a,p,cam).
// =====================================================================================
__global__ __launch_bounds__(256) void daf_taps_kernel(uint8_t *__restrict__ valid,
                                                       int32_t *__restrict__ taps,
                                                       const int *__restrict__ ss,
                                                       const int *__restrict__ start,
                                                       const float *__restrict__ loc, long n,
                                                       int cams, int num_feat, int L, int A, int P) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int cam = (int)(i % cams);
  const int b = (int)(i / ((long)cams * P * A));
  const float lw = loc[2 * i], lh = loc[2 * i + 1];
  const bool keep = loc_kept(lw, lh);
  valid[i] = keep ? 1 : 0;
  for (int s = 0; s < L; ++s) {
    int32_t *o = taps + (i * L + s) * 4;
    if (!keep) {
      o[0] = o[1] = o[2] = o[3] = 0;
      continue;
    }
    const int cs = cam * L + s;
    const int H = ss[2 * cs], W = ss[2 * cs + 1];
    const Taps t = make_taps(lh, lw, H, W);
    o[0] = t.h_low;
    o[1] = t.w_low;
    o[2] = (int)(t.in_h0 && t.in_w0) | ((int)(t.in_h0 && t.in_w1) << 1) |
           ((int)(t.in_h1 && t.in_w0) << 2) | ((int)(t.in_h1 && t.in_w1) << 3);
    o[3] = b * num_feat + start[cs];
  }
}

// ----------------------------------------------------------------------------- host side
static int g_pairs_fwd = 0, g_pairs_bwd = 0;  // 0 = default

struct Plan {
  int ppc;      // points per chunk
  int nchunks;  // chunks per anchor
};

// Work-item sizing.  One wave owns <= 128 (point,camera) pairs.  Measured on MI355X
// (profiles/r01a_sweep_pairs_per_wave.txt): the kernels are latency-bound per wave, so more,
// smaller items win until the per-item fixed cost shows (< ~24 pairs): aim for ~4096 waves.
static Plan make_plan(int P, int cams, int target_pairs, long n_anchor) {
  if (target_pairs <= 0) {
    const long want_chunks = (4096 + n_anchor - 1) / n_anchor;
    long ppc_auto = (P + want_chunks - 1) / want_chunks;
    target_pairs = (int)(ppc_auto * cams);
    if (target_pairs < 24) target_pairs = 24;
  }
  if (target_pairs > kMaxPairsPerWave) target_pairs = kMaxPairsPerWave;
  int ppc = target_pairs / cams;
  if (ppc < 1) ppc = 1;
  if (ppc > P) ppc = P;
  int nchunks = (P + ppc - 1) / ppc;
  ppc = (P + nchunks - 1) / nchunks;  // balance
  nchunks = (P + ppc - 1) / ppc;
  return {ppc, nchunks};
}

static int check_dims(int bs, int cams, int num_feat, int C, int L, int A, int P, int G) {
  if (bs <= 0 || cams <= 0 || num_feat <= 0 || C <= 0 || L <= 0 || A <= 0 || P <= 0 || G <= 0)
    return HIPAD_EINVAL;
  if (C % G) return HIPAD_EINVAL;
  if (cams > kMaxPairsPerWave) return HIPAD_EINVAL;
  // 32-bit budgets: pyramid rows and (anchor,pair) counts are int in the kernels
  if ((long long)bs * num_feat >= (1ll << 31)) return HIPAD_ERANGE;
  if ((long long)bs * A * P * cams >= (1ll << 31)) return HIPAD_ERANGE;
  return HIPAD_OK;
}

static int launch_status() { return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH; }

}  // namespace hipad

using namespace hipad;

extern "C" {

int hipad_abi_version(void) { return 3; }  // 3: round-3 entry points (merged feature-gradient pass, fused objective, depth loss, glue)

const char *hipad_status_string(int s) {
  switch (s) {
    case HIPAD_OK: return "ok";
    case HIPAD_EINVAL: return "invalid argument (dimension, null pointer or unsupported combination)";
    case HIPAD_EWORKSPACE: return "workspace too small";
    case HIPAD_ELAUNCH: return "kernel launch failed";
    case HIPAD_ERANGE: return "sizes exceed the 32-bit index budget";
    default: return "unknown status";
  }
}

void hipad_daf_set_pairs_per_wave(int fwd, int bwd) {
  g_pairs_fwd = fwd > 0 ? fwd : 0;
  g_pairs_bwd = bwd > 0 ? bwd : 0;
}

size_t hipad_daf_forward_workspace(int bs, int cams, int num_feat, int C, int L, int A, int P, int G) {
  if (check_dims(bs, cams, num_feat, C, L, A, P, G) != HIPAD_OK) return 0;
  const Plan pl = make_plan(P, cams, g_pairs_fwd, (long)bs * A);
  if (pl.nchunks == 1) return 0;
  return (size_t)bs * A * pl.nchunks * C * sizeof(float);
}

static int daf_forward_impl(float *out, const void *feat_, bool feat_bf16, const int32_t *spatial_shape,
                            const int32_t *scale_start_index, const float *loc, const float *weights,
                            int bs, int cams, int num_feat, int C, int L, int A, int P, int G,
                            void *workspace, size_t workspace_bytes, hipad_stream_t stream_) {
  int rc = check_dims(bs, cams, num_feat, C, L, A, P, G);
  if (rc != HIPAD_OK) return rc;
  if (!out || !feat_ || !spatial_shape || !scale_start_index || !loc || !weights) return HIPAD_EINVAL;
  const float *feat = (const float *)feat_;
  if (feat_bf16 && !((C == 256) && ((C / G) % 4 == 0))) return HIPAD_EINVAL;   // bf16 rows: the 256-channel kernel only
  hipStream_t stream = (hipStream_t)stream_;
  const Plan pl = make_plan(P, cams, g_pairs_fwd, (long)bs * A);
  const int n_anchor = bs * A;
  const int n_items = n_anchor * pl.nchunks;
  float *dst = out;
  if (pl.nchunks > 1) {
    const size_t need = (size_t)n_items * C * sizeof(float);
    if (!workspace || workspace_bytes < need) return HIPAD_EWORKSPACE;
    dst = (float *)workspace;
  }
  const bool fast = (C == 256) && ((C / G) % 4 == 0);
  if (fast) {
    const int blocks = (n_items + 3) / 4;
#define HIPAD_FWD_T(LT, GEO, FT)                                                                            \
  hipLaunchKernelGGL((daf_fwd_c256_kernel<LT, GEO, FT>), dim3(blocks), dim3(256), 0, stream, dst,           \
                     (const FT *)feat_, spatial_shape, scale_start_index, loc, weights, n_items, pl.nchunks, \
                     pl.ppc, cams, num_feat, L, A, P, G)
#define HIPAD_FWD(LT, GEO)                                                              \
  do {                                                                                  \
    if (feat_bf16) HIPAD_FWD_T(LT, GEO, uint16_t); else HIPAD_FWD_T(LT, GEO, float);    \
  } while (0)
    const bool geo = cams * L <= kWave;
    if (L == 4) {
      if (geo) HIPAD_FWD(4, true); else HIPAD_FWD(4, false);
    } else if (L == 1) {
      if (geo) HIPAD_FWD(1, true); else HIPAD_FWD(1, false);
    } else {
      if (geo) HIPAD_FWD(0, true); else HIPAD_FWD(0, false);
    }
#undef HIPAD_FWD
#undef HIPAD_FWD_T
    if (pl.nchunks > 1) {
      const int n = n_anchor * 64;
      hipLaunchKernelGGL(daf_fwd_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, out,
                         (const float *)dst, n_anchor, pl.nchunks, 64);
    }
  } else {
    const int cblocks = (C + kWave - 1) / kWave;
    const long nw = (long)n_items * cblocks;
    if (nw >= (1l << 31)) return HIPAD_ERANGE;
    hipLaunchKernelGGL(daf_fwd_generic_kernel, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, stream,
                       dst, feat, spatial_shape, scale_start_index, loc, weights, n_items, pl.nchunks,
                       pl.ppc, cams, num_feat, C, L, A, P, G, cblocks);
    if (pl.nchunks > 1) {
      const long n = (long)n_anchor * C;
      hipLaunchKernelGGL(daf_fwd_combine_generic_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256),
                         0, stream, out, (const float *)dst, n_anchor, pl.nchunks, C);
    }
  }
  return launch_status();
}

size_t hipad_daf_backward_workspace(int bs, int cams, int num_feat, int C, int L, int A, int P, int G) {
  if (check_dims(bs, cams, num_feat, C, L, A, P, G) != HIPAD_OK) return 0;
  const DafDims d{bs, cams, num_feat, C, L, A, P, G};
  return daf_bwd_sorted_workspace(d);
}

static int daf_backward_impl(const void *feat_, bool feat_bf16, const int32_t *spatial_shape,
                             const int32_t *scale_start_index, const float *loc, const float *weights,
                             const float *grad_out, float *grad_feat, float *grad_loc, float *grad_w,
                             int bs, int cams, int num_feat, int C, int L, int A, int P, int G, int flags,
                             void *workspace, size_t workspace_bytes, hipad_stream_t stream_) {
  int rc = check_dims(bs, cams, num_feat, C, L, A, P, G);
  if (rc != HIPAD_OK) return rc;
  if (!feat_ || !spatial_shape || !scale_start_index || !loc || !weights || !grad_out) return HIPAD_EINVAL;
  const float *feat = (const float *)feat_;
  if (flags & ~(HIPAD_DAF_OVERWRITE_LOC_W | HIPAD_DAF_ATOMIC_FEAT)) return HIPAD_EINVAL;
  if (!grad_feat && !grad_loc && !grad_w) return HIPAD_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const Plan pl = make_plan(P, cams, g_pairs_bwd, (long)bs * A);
  const int n_items = bs * A * pl.nchunks;
  const bool overwrite = (flags & HIPAD_DAF_OVERWRITE_LOC_W) != 0;
  const DafDims d{bs, cams, num_feat, C, L, A, P, G};
  const bool fast = (C == 256) && (G == 8);
  if (feat_bf16 && !(fast && !(flags & HIPAD_DAF_ATOMIC_FEAT) && daf_bwd_sorted_supported(d))) return HIPAD_EINVAL;
  if (fast && !(flags & HIPAD_DAF_ATOMIC_FEAT) && daf_bwd_sorted_supported(d)) {
    // sorted path: grad_feat by row-sorted gather, grad_loc / grad_w by the wave-per-item kernel
    if (grad_feat) {
      rc = daf_bwd_sorted_feat(feat, spatial_shape, scale_start_index, loc, weights, grad_out, grad_feat, d,
                               workspace, workspace_bytes, stream);
      if (rc != HIPAD_OK) return rc;
    }
    if (grad_loc || grad_w) {
      rc = daf_bwd_lw(feat_, feat_bf16, spatial_shape, scale_start_index, loc, weights, grad_out, grad_loc, grad_w, d,
                      pl.nchunks, pl.ppc, overwrite, stream);
      if (rc != HIPAD_OK) return rc;
    }
    return launch_status();
  }
  if (fast) {
    const int blocks = (n_items + 3) / 4;
#define HIPAD_BWD(LT, OW)                                                                         \
  hipLaunchKernelGGL((daf_bwd_c256g8_kernel<LT, OW>), dim3(blocks), dim3(256), 0, stream, feat,   \
                     spatial_shape, scale_start_index, loc, weights, grad_out, grad_feat, grad_loc, \
                     grad_w, n_items, pl.nchunks, pl.ppc, cams, num_feat, L, A, P)
    if (L == 4) {
      if (overwrite) HIPAD_BWD(4, true); else HIPAD_BWD(4, false);
    } else {
      if (overwrite) HIPAD_BWD(0, true); else HIPAD_BWD(0, false);
    }
#undef HIPAD_BWD
  } else {
    if (overwrite) {
      // generic path accumulates with atomics: clear what the flag promises to overwrite
      const size_t npair = (size_t)bs * A * P * cams;
      if (grad_loc && fill_zero(grad_loc, npair * 2 * sizeof(float), stream) != HIPAD_OK) return HIPAD_ELAUNCH;
      if (grad_w && fill_zero(grad_w, npair * L * G * sizeof(float), stream) != HIPAD_OK) return HIPAD_ELAUNCH;
    }
    const int cblocks = (C + kWave - 1) / kWave;
    const long nw = (long)n_items * cblocks;
    if (nw >= (1l << 31)) return HIPAD_ERANGE;
    hipLaunchKernelGGL(daf_bwd_generic_kernel, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, stream,
                       feat, spatial_shape, scale_start_index, loc, weights, grad_out, grad_feat,
                       grad_loc, grad_w, n_items, pl.nchunks, pl.ppc, cams, num_feat, C, L, A, P, G,
                       cblocks);
  }
  return launch_status();
}

int hipad_daf_forward(float *out, const float *feat, const int32_t *spatial_shape,
                      const int32_t *scale_start_index, const float *loc, const float *weights,
                      int bs, int cams, int num_feat, int C, int L, int A, int P, int G,
                      void *workspace, size_t workspace_bytes, hipad_stream_t stream) {
  return daf_forward_impl(out, feat, false, spatial_shape, scale_start_index, loc, weights, bs, cams, num_feat, C, L, A, P, G,
                          workspace, workspace_bytes, stream);
}

int hipad_daf_forward_bf16(float *out, const void *feat_bf16, const int32_t *spatial_shape,
                           const int32_t *scale_start_index, const float *loc, const float *weights,
                           int bs, int cams, int num_feat, int C, int L, int A, int P, int G,
                           void *workspace, size_t workspace_bytes, hipad_stream_t stream) {
  return daf_forward_impl(out, feat_bf16, true, spatial_shape, scale_start_index, loc, weights, bs, cams, num_feat, C, L, A, P,
                          G, workspace, workspace_bytes, stream);
}

int hipad_daf_backward(const float *feat, const int32_t *spatial_shape,
                       const int32_t *scale_start_index, const float *loc, const float *weights,
                       const float *grad_out, float *grad_feat, float *grad_loc, float *grad_w,
                       int bs, int cams, int num_feat, int C, int L, int A, int P, int G, int flags,
                       void *workspace, size_t workspace_bytes, hipad_stream_t stream) {
  return daf_backward_impl(feat, false, spatial_shape, scale_start_index, loc, weights, grad_out, grad_feat, grad_loc, grad_w,
                           bs, cams, num_feat, C, L, A, P, G, flags, workspace, workspace_bytes, stream);
}

int hipad_daf_backward_bf16(const void *feat_bf16, const int32_t *spatial_shape,
                            const int32_t *scale_start_index, const float *loc, const float *weights,
                            const float *grad_out, float *grad_feat, float *grad_loc, float *grad_w,
                            int bs, int cams, int num_feat, int C, int L, int A, int P, int G, int flags,
                            void *workspace, size_t workspace_bytes, hipad_stream_t stream) {
  return daf_backward_impl(feat_bf16, true, spatial_shape, scale_start_index, loc, weights, grad_out, grad_feat, grad_loc,
                           grad_w, bs, cams, num_feat, C, L, A, P, G, flags, workspace, workspace_bytes, stream);
}

int hipad_daf_taps(uint8_t *valid, int32_t *taps, const int32_t *spatial_shape,
                   const int32_t *scale_start_index, const float *loc, int bs, int cams,
                   int num_feat, int L, int A, int P, hipad_stream_t stream_) {
  if (!valid || !taps || !spatial_shape || !scale_start_index || !loc) return HIPAD_EINVAL;
  if (bs <= 0 || cams <= 0 || L <= 0 || A <= 0 || P <= 0) return HIPAD_EINVAL;
  const long n = (long)bs * A * P * cams;
  hipLaunchKernelGGL(daf_taps_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream_, valid, taps, spatial_shape, scale_start_index, loc, n, cams,
                     num_feat, L, A, P);
  return launch_status();
}

}  // extern "C"
