// hip-ad_amd/csrc/chain.hip -- whole MLP stacks of the decoder in ONE launch per direction (gfx950).
//
// Replaces: the `linear_relu_ln` stacks and heads of the reference decoder -- ([Linear, ReLU] x in_loops, LayerNorm)
// x out_loops, then a Linear (+ mmcv Scale) (+ residual anchor) -- reference models/blocks.py:32-42 and their users:
// refinement heads det/blocks.py:77-156, map/blocks.py:80-135, plan/blocks.py:16-157, motion/blocks.py:16-50,
// ego/blocks.py:14-75, anchor encoders det/blocks.py:22-74, map/blocks.py:18-42, the camera / command / target-point
// encoders (blocks.py:104-105, sparse_onedecoder.py:300-330).  A stage-2 frame holds ~1 750 such Linear and ~850
// LayerNorm calls on 1 .. 5 400 rows of width <= 256; launched one by one each pays a 3-8 us dispatch + ramp although
// its arithmetic is < 1 us (profiles/r01h_*: Linear + LayerNorm = 34 % of the step).
//
// Here a CHAIN (up to 6 layers) runs row-tile resident: a workgroup owns R = 16 or 32 rows, keeps the activation tile in
// LDS (bf16 operand copy + fp32 result copy), streams each layer's bf16 weights from L2 straight into MFMA B fragments
// (a lane's fragment is 16 contiguous bytes of one weight row: no LDS staging -- each weight element is used once per
// workgroup), prefetches the NEXT layer's weights while the row phase (bias / ReLU / LayerNorm / saves) of the
// current one runs, and writes only what the backward needs.  Several independent chains (the ten plan branches, the
// three det heads, the four box-encoder parts) share one launch through a by-value descriptor table.
//
// Backward = two launches per group of chains:
//   chain_bwd_kernel   row-tile resident reverse sweep: LayerNorm / ReLU / Scale backward in the row phase (gamma, beta,
//                      scale gradients accumulated atomically), dX = dY W by MFMA against the TRANSPOSED bf16 weights;
//                      the gated dY of every layer is left in memory for
//   chain_dw_kernel    (gemm.hip) dW += dY^T X, db += colsum(dY) of every layer of every chain, one grid.
// Numerics = the per-layer kernels': bf16 operands, fp32 accumulation, fp32 LayerNorm (two-pass statistics).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hipad.h"
#include "wave_ops.h"

namespace hipad {

using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int CH_W = 256;              // widest layer
constexpr int CH_XS = CH_W + 8;        // bf16 operand tile stride (528 B: 16-byte fragments, conflict-free)
constexpr int CH_YS = CH_W + 4;        // fp32 result tile stride
constexpr int CH_MAXT = 4;             // 16-column MFMA tiles per wave (4 waves x 4 x 16 = 256)
constexpr int CH_MAXS = 8;             // 32-deep reduction steps (8 x 32 = 256)
constexpr unsigned CH_NONE = 0xffffffffu;

__device__ __forceinline__ short ch_bf16(float x) { return __builtin_bit_cast(short, (__bf16)x); }

struct ChainFwdLayer {                 // 56 bytes
  const unsigned short *w;             // bf16 [N][K]
  const float *bias, *gamma, *beta;
  unsigned off_h, off_y, off_stats;    // float offsets into ChainFwd::save (CH_NONE: not kept)
  unsigned short K, N;
  unsigned flags;                      // bit 0 ReLU, bit 1 LayerNorm
  float eps;
};
struct ChainFwd {                      // 88 + 6 * 56 = 424 bytes
  const float *x0, *x1;
  float *xsum, *out;
  const float *out_scale, *residual;
  float *save;
  int ldx0, ldx1, ldo, ldr;
  int M, nlayers, tile0, pad;
  ChainFwdLayer L[HIPAD_CHAIN_MAX_LAYERS];
};
struct ChainFwdArgs {
  int nchains, pad;
  unsigned long long *stamps;          // diagnostic build aid (hipad_chain_debug_stamps): cycle stamps of workgroup 0
  int tile0[HIPAD_CHAIN_MAX_CHAINS];   // first workgroup of every chain (also in c[i].tile0)
  ChainFwd c[HIPAD_CHAIN_MAX_CHAINS];
};

struct ChainBwdLayer {                 // 56 bytes
  const unsigned short *wt;            // bf16 [K][N] (transposed weights)
  const float *gamma;
  float *dgamma, *dbeta;
  unsigned off_h, off_stats, off_dy;   // float offsets: h / stats into ChainBwd::save, dy into ChainBwd::dy
  unsigned short K, N;
  unsigned flags;
  float eps;
};
struct ChainBwd {                      // 72 + 336 = 408 bytes
  const float *dout;
  const float *out_scale;
  float *dscale, *dx;
  const float *save;
  float *dy;
  int ldo, lddx, M, nlayers, tile0, pad;
  ChainBwdLayer L[HIPAD_CHAIN_MAX_LAYERS];
};
struct ChainBwdArgs {
  int nchains, pad;
  int tile0[HIPAD_CHAIN_MAX_CHAINS];
  ChainBwd c[HIPAD_CHAIN_MAX_CHAINS];
};

// The workgroup's chain descriptor, copied ONCE from the kernel-argument segment into LDS (one vector load per lane).
// Read in place, the ~60 scalar loads the compiler spreads over the layer loop each paid a trip to the argument buffer
// (host-visible memory: ~1.5 us per cache line; measured ~7 us per LAYER for a single-workgroup launch).
template <typename Desc, typename Args>
__device__ __forceinline__ const Desc &ch_stage_desc(Desc &lds, const Args &a) {
  int ci = 0;
#pragma unroll
  for (int i = 1; i < HIPAD_CHAIN_MAX_CHAINS; ++i)
    if (i < a.nchains && (int)blockIdx.x >= a.tile0[i]) ci = i;
  const unsigned *src = reinterpret_cast<const unsigned *>(&a.c[ci]);
  unsigned *dst = reinterpret_cast<unsigned *>(&lds);
  if (threadIdx.x < sizeof(Desc) / 4) dst[threadIdx.x] = src[threadIdx.x];
  __syncthreads();
  return lds;
}

// All B fragments this wave needs for one layer: tiles wv, wv + 4, ... of 16 output columns, ksteps reduction steps.
// The matrix is in MFMA-fragment order (hipad.h, hipad_pack_weights): block (t, s) = 64 lanes x 16 bytes, so a fragment
// is ONE fully coalesced 1 KiB wave load with no per-lane condition at all (the padding is stored as zeros).  History:
// row-major weights read as 16 rows x 64 bytes per instruction took ~190 cycles per load to issue (2.5 us per 256 x 256
// layer and workgroup); per-lane `if`s or a select behind the loads made the compiler wait for each load in turn
// (~10 us per layer).  The `t < ntiles` / `s < ksteps` guards are wave-uniform scalar branches.
template <bool FULL>
__device__ __forceinline__ void ch_fetch(bf16x8 (&bw)[CH_MAXT][CH_MAXS], const unsigned short *__restrict__ w, int rows,
                                         int depth, int wv, int lane) {
  const bf16x8 *base = reinterpret_cast<const bf16x8 *>(w) + lane;
  if (FULL) {  // rows = depth = 256: every tile and step exists, straight-line code, 32 loads in flight
#pragma unroll
    for (int tt = 0; tt < CH_MAXT; ++tt)
#pragma unroll
      for (int s = 0; s < CH_MAXS; ++s) bw[tt][s] = base[((wv + 4 * tt) * CH_MAXS + s) * 64];
    return;
  }
  const int ksteps = (depth + 31) >> 5, ntiles = (rows + 15) >> 4;
#pragma unroll
  for (int tt = 0; tt < CH_MAXT; ++tt) {
    const int t = wv + 4 * tt;
    if (t < ntiles) {  // wave-uniform
#pragma unroll
      for (int s = 0; s < CH_MAXS; ++s)
        if (s < ksteps) bw[tt][s] = base[(t * ksteps + s) * 64];  // uniform
    }
  }
}

// four consecutive elements p[c0 .. c0 + 3] of an n-vector (elements past n read as 0), branch-free
__device__ __forceinline__ float4 ch_ld4(const float *__restrict__ p, int c0, int n) {
  const float a = p[c0 + 0 < n ? c0 + 0 : 0], b = p[c0 + 1 < n ? c0 + 1 : 0], c = p[c0 + 2 < n ? c0 + 2 : 0],
              d = p[c0 + 3 < n ? c0 + 3 : 0];
  return make_float4(c0 + 0 < n ? a : 0.f, c0 + 1 < n ? b : 0.f, c0 + 2 < n ? c : 0.f, c0 + 3 < n ? d : 0.f);
}

// acc[tt][i] = A(LDS operand tile, rows 16 i ..) x B(bw[tt]) for the wave's tiles.  SPLIT: the activations are held as
// hi + lo bf16 pairs (lo = bf16(v - hi)) and multiplied in two MFMAs against the same weight fragment: the activation
// side of the product is then exact to ~2^-17, only the weights carry bf16 rounding -- the forward's error drops by
// ~1/sqrt(2) for one extra MFMA per fragment (the MFMA phase is ~10 % of a layer; no extra weight traffic).
template <int R, bool FULL, bool SPLIT>
__device__ __forceinline__ void ch_mma(f32x4 (&acc)[CH_MAXT][R / 16], const bf16x8 (&bw)[CH_MAXT][CH_MAXS],
                                       const short (*X)[CH_XS], const short (*XL)[CH_XS], int rows, int depth, int wv,
                                       int l15, int quad) {
#pragma unroll
  for (int tt = 0; tt < CH_MAXT; ++tt)
#pragma unroll
    for (int i = 0; i < R / 16; ++i) acc[tt][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (FULL) {  // one A fragment feeds the four independent accumulators of the wave's tiles
#pragma unroll
    for (int s = 0; s < CH_MAXS; ++s)
#pragma unroll
      for (int i = 0; i < R / 16; ++i) {
        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(&X[16 * i + l15][32 * s + 8 * quad]);
#pragma unroll
        for (int tt = 0; tt < CH_MAXT; ++tt)
          acc[tt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bw[tt][s], acc[tt][i], 0, 0, 0);
        if (SPLIT) {
          const bf16x8 al = *reinterpret_cast<const bf16x8 *>(&XL[16 * i + l15][32 * s + 8 * quad]);
#pragma unroll
          for (int tt = 0; tt < CH_MAXT; ++tt)
            acc[tt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bw[tt][s], acc[tt][i], 0, 0, 0);
        }
      }
    return;
  }
  const int ksteps = (depth + 31) >> 5, ntiles = (rows + 15) >> 4;
#pragma unroll
  for (int tt = 0; tt < CH_MAXT; ++tt) {
    if (wv + 4 * tt < ntiles) {
#pragma unroll
      for (int s = 0; s < CH_MAXS; ++s) {
        if (s < ksteps) {
#pragma unroll
          for (int i = 0; i < R / 16; ++i) {
            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(&X[16 * i + l15][32 * s + 8 * quad]);
            acc[tt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bw[tt][s], acc[tt][i], 0, 0, 0);
            if (SPLIT) {
              const bf16x8 al = *reinterpret_cast<const bf16x8 *>(&XL[16 * i + l15][32 * s + 8 * quad]);
              acc[tt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bw[tt][s], acc[tt][i], 0, 0, 0);
            }
          }
        }
      }
    }
  }
}

// wave-uniform dispatch on "the layer is 256 x 256"
__device__ __forceinline__ void ch_fetch_any(bf16x8 (&bw)[CH_MAXT][CH_MAXS], const unsigned short *__restrict__ w, int rows,
                                             int depth, int wv, int lane) {
  if (rows == CH_W && depth == CH_W) ch_fetch<true>(bw, w, rows, depth, wv, lane);
  else ch_fetch<false>(bw, w, rows, depth, wv, lane);
}
template <int R, bool SPLIT>
__device__ __forceinline__ void ch_mma_any(f32x4 (&acc)[CH_MAXT][R / 16], const bf16x8 (&bw)[CH_MAXT][CH_MAXS],
                                           const short (*X)[CH_XS], const short (*XL)[CH_XS], int rows, int depth, int wv,
                                           int l15, int quad) {
  if (rows == CH_W && depth == CH_W) ch_mma<R, true, SPLIT>(acc, bw, X, XL, rows, depth, wv, l15, quad);
  else ch_mma<R, false, SPLIT>(acc, bw, X, XL, rows, depth, wv, l15, quad);
}

// hi / lo bf16 halves of four activations into the two operand tiles
__device__ __forceinline__ void ch_store_split(short *hi, short *lo, float4 v) {
  const __bf16 h0 = (__bf16)v.x, h1 = (__bf16)v.y, h2 = (__bf16)v.z, h3 = (__bf16)v.w;
  *reinterpret_cast<short4 *>(hi) = make_short4(__builtin_bit_cast(short, h0), __builtin_bit_cast(short, h1),
                                                __builtin_bit_cast(short, h2), __builtin_bit_cast(short, h3));
  *reinterpret_cast<short4 *>(lo) = make_short4(ch_bf16(v.x - (float)h0), ch_bf16(v.y - (float)h1), ch_bf16(v.z - (float)h2),
                                                ch_bf16(v.w - (float)h3));
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void chain_fwd_kernel(const ChainFwdArgs a) {
  __shared__ short X[R][CH_XS];    // activations, bf16 hi halves (MFMA A operand)
  __shared__ short XL[R][CH_XS];   // ... lo halves
  __shared__ float Y[R][CH_YS];
  __shared__ ChainFwd desc;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, quad = lane >> 4;
  unsigned long long *stamps = (a.stamps && blockIdx.x == 0 && tid == 0) ? a.stamps : nullptr;
  int nstamp = 0;
#define CH_STAMP()                                                                   \
  do {                                                                               \
    if (stamps) {                                                                    \
      stamps[2 * nstamp] = __builtin_amdgcn_s_memtime();                             \
      stamps[2 * nstamp + 1] = __builtin_amdgcn_s_memrealtime();                     \
      ++nstamp;                                                                      \
    }                                                                                \
  } while (0)
  CH_STAMP();
  const ChainFwd &c = ch_stage_desc(desc, a);
  const int M = c.M, r0 = ((int)blockIdx.x - c.tile0) * R;
  CH_STAMP();

  bf16x8 bw[CH_MAXT][CH_MAXS];
  {  // input rows -> bf16 operand tile, zero-padded to a multiple of 32 columns.  Issue order: the input loads, THEN the
     // first layer's weights (vector-memory results return in order: the tile can be built while the weights still fly)
    const int K0 = c.L[0].K, Kp = (K0 + 31) & ~31;
    const float *x0 = c.x0, *x1 = c.x1;
    float *xs = c.xsum;
    const bool vec = (K0 & 3) == 0 && (c.ldx0 & 3) == 0 && ((uintptr_t)x0 & 15) == 0 &&
                     (!x1 || ((c.ldx1 & 3) == 0 && ((uintptr_t)x1 & 15) == 0));
    if (vec) {
      constexpr int IT = R * (CH_W / 4) / 256;  // float4 per thread for the widest input
      const int q = Kp >> 2;                    // float4 per padded row
      float4 v0[IT], v1[IT];
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int idx = tid + 256 * it;
        const int r = idx / q, col = (idx - r * q) * 4, row = r0 + r;
        const bool ok = idx < R * q && row < M && col < K0;
        // branch-free: out-of-range threads re-read the first element of the tensor and discard it
        v0[it] = *reinterpret_cast<const float4 *>(ok ? x0 + (size_t)row * c.ldx0 + col : x0);
        v1[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x1) v1[it] = *reinterpret_cast<const float4 *>(ok ? x1 + (size_t)row * c.ldx1 + col : x1);  // uniform branch
      }
      ch_fetch_any(bw, c.L[0].w, c.L[0].N, c.L[0].K, wv, lane);
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int idx = tid + 256 * it;
        const int r = idx / q, col = (idx - r * q) * 4, row = r0 + r;
        if (idx < R * q) {
          const bool ok = row < M && col < K0;
          float4 v = make_float4(v0[it].x + v1[it].x, v0[it].y + v1[it].y, v0[it].z + v1[it].z, v0[it].w + v1[it].w);
          if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (xs && x1 && ok) *reinterpret_cast<float4 *>(xs + (size_t)row * K0 + col) = v;
          ch_store_split(&X[r][col], &XL[r][col], v);
        }
      }
    } else {
      for (int idx = tid; idx < R * Kp; idx += 256) {
        const int r = idx / Kp, col = idx - r * Kp, row = r0 + r;
        float v = 0.f;
        if (row < M && col < K0) {
          v = x0[(size_t)row * c.ldx0 + col];
          if (x1) {
            v += x1[(size_t)row * c.ldx1 + col];
            if (xs) xs[(size_t)row * K0 + col] = v;
          }
        }
        const __bf16 hi = (__bf16)v;
        X[r][col] = __builtin_bit_cast(short, hi);
        XL[r][col] = ch_bf16(v - (float)hi);
      }
      ch_fetch_any(bw, c.L[0].w, c.L[0].N, c.L[0].K, wv, lane);
    }
  }

  const int nl = c.nlayers;
  const int c0 = 4 * lane;
  for (int l = 0; l < nl; ++l) {
    const ChainFwdLayer &L = c.L[l];
    const int K = L.K, N = L.N;
    const bool last = l + 1 == nl;
    const bool relu = L.flags & 1u, ln = L.flags & 2u;
    const int ntiles = (N + 15) >> 4;
    // this layer's small parameters, issued before anything waits: bias of the wave's tiles, LayerNorm gamma / beta and
    // the output scale of the lane's four columns (they land during the barrier and the MFMA phase)
    float bv[CH_MAXT];
#pragma unroll
    for (int tt = 0; tt < CH_MAXT; ++tt) {
      const int col = 16 * (wv + 4 * tt) + l15;
      bv[tt] = L.bias ? L.bias[col < N ? col : 0] : 0.f;
    }
    float4 gm = make_float4(1.f, 1.f, 1.f, 1.f), bt = make_float4(0.f, 0.f, 0.f, 0.f), sc = gm;
    if (ln) {  // uniform
      if (L.gamma) gm = ch_ld4(L.gamma, c0, N);
      if (L.beta) bt = ch_ld4(L.beta, c0, N);
    }
    if (last && c.out_scale) sc = ch_ld4(c.out_scale, c0, N);
    CH_STAMP();
    __syncthreads();  // operand tile complete
    CH_STAMP();
    f32x4 acc[CH_MAXT][R / 16];
    ch_mma_any<R, true>(acc, bw, X, XL, N, K, wv, l15, quad);
    CH_STAMP();
    // bias (+ ReLU) -> fp32 result tile; element (i, r): row 16 i + 4 quad + r, column 16 t + l15
#pragma unroll
    for (int tt = 0; tt < CH_MAXT; ++tt) {
      const int t = wv + 4 * tt;
      if (t < ntiles) {
        const int col = 16 * t + l15;
#pragma unroll
        for (int i = 0; i < R / 16; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[tt][i][r] + bv[tt];
            if (relu) v = fmaxf(v, 0.f);
            Y[16 * i + 4 * quad + r][col] = v;
          }
      }
    }
    CH_STAMP();
    if (!last) ch_fetch_any(bw, c.L[l + 1].w, c.L[l + 1].N, c.L[l + 1].K, wv, lane);  // lands during the row phase
    CH_STAMP();
    __syncthreads();  // result tile complete; operand tile free
    CH_STAMP();

    // row phase: a wave walks rows wv, wv + 4, ...; lane owns columns 4 lane .. 4 lane + 3
    const bool n4 = (N & 3) == 0;
    const int Np = (N + 31) & ~31;
    const float inv_n = 1.f / (float)N;
    float *sv = c.save;
    const unsigned off_h = L.off_h, off_y = L.off_y, off_stats = L.off_stats;
    const float eps = L.eps;
#pragma unroll
    for (int j = 0; j < R / 4; ++j) {
      const int r = wv + 4 * j;
      const int row = r0 + r;
      const bool rok = row < M;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c0 < N) {  // the tile is padded to CH_YS floats: reading past N inside the row is harmless, values masked below
        v = *reinterpret_cast<const float4 *>(&Y[r][c0]);
        if (c0 + 1 >= N) v.y = 0.f;
        if (c0 + 2 >= N) v.z = 0.f;
        if (c0 + 3 >= N) v.w = 0.f;
      }
      if (sv && off_h != CH_NONE && rok && c0 < N) {
        float *h = sv + off_h + (size_t)row * N + c0;
        if (n4) *reinterpret_cast<float4 *>(h) = v;
        else {
          h[0] = v.x;
          if (c0 + 1 < N) h[1] = v.y;
          if (c0 + 2 < N) h[2] = v.z;
          if (c0 + 3 < N) h[3] = v.w;
        }
      }
      if (ln) {
        const float mean = wave_sum(v.x + v.y + v.z + v.w) * inv_n;
        float4 d = make_float4(c0 + 0 < N ? v.x - mean : 0.f, c0 + 1 < N ? v.y - mean : 0.f,
                               c0 + 2 < N ? v.z - mean : 0.f, c0 + 3 < N ? v.w - mean : 0.f);
        const float rstd = rsqrtf(wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) * inv_n + eps);
        v = make_float4(d.x * rstd * gm.x + bt.x, d.y * rstd * gm.y + bt.y, d.z * rstd * gm.z + bt.z,
                        d.w * rstd * gm.w + bt.w);
        if (c0 + 1 >= N) v.y = 0.f;
        if (c0 + 2 >= N) v.z = 0.f;
        if (c0 + 3 >= N) v.w = 0.f;
        if (c0 >= N) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sv && rok) {
          if (off_stats != CH_NONE && lane == 0) {
            sv[off_stats + 2 * (size_t)row] = mean;
            sv[off_stats + 2 * (size_t)row + 1] = rstd;
          }
          if (off_y != CH_NONE && c0 < N) {
            float *y = sv + off_y + (size_t)row * N + c0;
            if (n4) *reinterpret_cast<float4 *>(y) = v;
            else {
              y[0] = v.x;
              if (c0 + 1 < N) y[1] = v.y;
              if (c0 + 2 < N) y[2] = v.z;
              if (c0 + 3 < N) y[3] = v.w;
            }
          }
        }
      }
      if (!last) {
        if (c0 < Np)  // next layer's operand tiles, zero-padded to a multiple of 32 columns
          ch_store_split(&X[r][c0], &XL[r][c0], v);
      } else if (rok && c0 < N) {
        float4 o = make_float4(v.x * sc.x, v.y * sc.y, v.z * sc.z, v.w * sc.w);
        if (c.residual) {
          const float *rs = c.residual + (size_t)row * c.ldr + c0;
          o.x += rs[0];
          if (c0 + 1 < N) o.y += rs[1];
          if (c0 + 2 < N) o.z += rs[2];
          if (c0 + 3 < N) o.w += rs[3];
        }
        float *op = c.out + (size_t)row * c.ldo + c0;
        if (n4 && (c.ldo & 3) == 0 && ((uintptr_t)c.out & 15) == 0) *reinterpret_cast<float4 *>(op) = o;
        else {
          op[0] = o.x;
          if (c0 + 1 < N) op[1] = o.y;
          if (c0 + 2 < N) op[2] = o.z;
          if (c0 + 3 < N) op[3] = o.w;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward, reverse sweep (dX chain + LayerNorm / ReLU / Scale backward; leaves every layer's dY for chain_dw_kernel)
// ---------------------------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void chain_bwd_kernel(const ChainBwdArgs a) {
  __shared__ short X[R][CH_XS];   // bf16 dY of the current layer (MFMA A operand)
  __shared__ float G[R][CH_YS];   // fp32 gradient w.r.t. the current layer's output
  __shared__ float red[2][4][CH_W];
  __shared__ ChainBwd desc;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, quad = lane >> 4;
  const ChainBwd &c = ch_stage_desc(desc, a);
  const int M = c.M, r0 = ((int)blockIdx.x - c.tile0) * R, nl = c.nlayers;
  const int c0 = 4 * lane;

  bf16x8 bw[CH_MAXT][CH_MAXS];
  {  // incoming gradient rows -> G (loads first, then the last layer's transposed weights, then the LDS stores)
    const int N = c.L[nl - 1].N;
    const bool n4 = (N & 3) == 0 && (c.ldo & 3) == 0 && ((uintptr_t)c.dout & 15) == 0;
    float4 g[R / 4];
#pragma unroll
    for (int j = 0; j < R / 4; ++j) {
      const int row = r0 + wv + 4 * j;
      const bool ok = row < M && c0 < N;
      const float *p = c.dout + (size_t)(ok ? row : 0) * c.ldo;
      if (n4) {
        g[j] = *reinterpret_cast<const float4 *>(p + (ok ? c0 : 0));
      } else {
        g[j] = make_float4(p[c0 + 0 < N ? c0 + 0 : 0], p[c0 + 1 < N ? c0 + 1 : 0], p[c0 + 2 < N ? c0 + 2 : 0],
                           p[c0 + 3 < N ? c0 + 3 : 0]);
      }
      if (!ok) g[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (nl > 1 || c.dx) ch_fetch_any(bw, c.L[nl - 1].wt, c.L[nl - 1].K, c.L[nl - 1].N, wv, lane);
#pragma unroll
    for (int j = 0; j < R / 4; ++j)
      if (c0 < CH_W) *reinterpret_cast<float4 *>(&G[wv + 4 * j][c0]) = g[j];
  }
  for (int l = nl - 1; l >= 0; --l) {
    const ChainBwdLayer &L = c.L[l];
    const int K = L.K, N = L.N;
    const bool last = l + 1 == nl;
    const bool relu = L.flags & 1u, ln = L.flags & 2u;
    const bool n4 = (N & 3) == 0;
    const int Np = (N + 31) & ~31;
    const float inv_n = 1.f / (float)N;
    const bool scaled = last && c.out_scale;
    const bool need_h = relu || ln || scaled;
    const float *sv = c.save;
    // everything the row phase reads from memory, issued before the barrier: gamma / scale of the lane's columns and,
    // for each of the wave's rows, the saved activation h and the LayerNorm statistics
    float4 gm = make_float4(1.f, 1.f, 1.f, 1.f), sc = gm;
    if (ln && L.gamma) gm = ch_ld4(L.gamma, c0, N);
    if (scaled) sc = ch_ld4(c.out_scale, c0, N);
    float4 hs[R / 4];
    float mu[R / 4], rs[R / 4];
#pragma unroll
    for (int j = 0; j < R / 4; ++j) {
      const int row = r0 + wv + 4 * j;
      const bool ok = row < M && c0 < N;
      hs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      mu[j] = rs[j] = 0.f;
      if (need_h) {  // uniform
        const float *hp = sv + L.off_h + (size_t)(ok ? row : 0) * N;
        if (n4) hs[j] = *reinterpret_cast<const float4 *>(hp + (ok ? c0 : 0));
        else hs[j] = make_float4(hp[c0 + 0 < N ? c0 + 0 : 0], hp[c0 + 1 < N ? c0 + 1 : 0], hp[c0 + 2 < N ? c0 + 2 : 0],
                                 hp[c0 + 3 < N ? c0 + 3 : 0]);
        if (!ok) hs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 + 1 >= N) hs[j].y = 0.f;
        if (c0 + 2 >= N) hs[j].z = 0.f;
        if (c0 + 3 >= N) hs[j].w = 0.f;
      }
      if (ln) {  // uniform
        const size_t so = L.off_stats + 2 * (size_t)(row < M ? row : 0);
        mu[j] = sv[so];
        rs[j] = sv[so + 1];
      }
    }
    __syncthreads();  // G complete
    float4 pg = make_float4(0.f, 0.f, 0.f, 0.f), pb = pg, ps = pg;  // gamma / beta / scale gradient partials
#pragma unroll
    for (int j = 0; j < R / 4; ++j) {
      const int r = wv + 4 * j;
      const int row = r0 + r;
      const bool rok = row < M;
      const float4 h = hs[j];
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c0 < N) {
        g = *reinterpret_cast<const float4 *>(&G[r][c0]);
        if (c0 + 1 >= N) g.y = 0.f;
        if (c0 + 2 >= N) g.z = 0.f;
        if (c0 + 3 >= N) g.w = 0.f;
      }
      if (!rok) g = make_float4(0.f, 0.f, 0.f, 0.f);
      if (scaled) {  // out = h * scale (+ residual)
        ps.x += g.x * h.x; ps.y += g.y * h.y; ps.z += g.z * h.z; ps.w += g.w * h.w;
        g.x *= sc.x; g.y *= sc.y; g.z *= sc.z; g.w *= sc.w;
      }
      if (ln) {
        const float mean = mu[j], rstd = rs[j];
        float4 xh = make_float4(c0 + 0 < N ? (h.x - mean) * rstd : 0.f, c0 + 1 < N ? (h.y - mean) * rstd : 0.f,
                                c0 + 2 < N ? (h.z - mean) * rstd : 0.f, c0 + 3 < N ? (h.w - mean) * rstd : 0.f);
        pg.x += g.x * xh.x; pg.y += g.y * xh.y; pg.z += g.z * xh.z; pg.w += g.w * xh.w;
        pb.x += g.x; pb.y += g.y; pb.z += g.z; pb.w += g.w;
        const float4 gh = make_float4(g.x * gm.x, g.y * gm.y, g.z * gm.z, g.w * gm.w);
        const float m1 = wave_sum(gh.x + gh.y + gh.z + gh.w) * inv_n;
        const float m2 = wave_sum(gh.x * xh.x + gh.y * xh.y + gh.z * xh.z + gh.w * xh.w) * inv_n;
        g = make_float4(rstd * (gh.x - m1 - xh.x * m2), rstd * (gh.y - m1 - xh.y * m2), rstd * (gh.z - m1 - xh.z * m2),
                        rstd * (gh.w - m1 - xh.w * m2));
        if (c0 + 0 >= N) g.x = 0.f;
        if (c0 + 1 >= N) g.y = 0.f;
        if (c0 + 2 >= N) g.z = 0.f;
        if (c0 + 3 >= N) g.w = 0.f;
      }
      if (relu) {
        if (!(h.x > 0.f)) g.x = 0.f;
        if (!(h.y > 0.f)) g.y = 0.f;
        if (!(h.z > 0.f)) g.z = 0.f;
        if (!(h.w > 0.f)) g.w = 0.f;
      }
      if (!rok) g = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rok && c0 < N) {  // dY of this layer, for the weight-gradient kernel
        float *dp = c.dy + L.off_dy + (size_t)row * N + c0;
        if (n4) *reinterpret_cast<float4 *>(dp) = g;
        else {
          dp[0] = g.x;
          if (c0 + 1 < N) dp[1] = g.y;
          if (c0 + 2 < N) dp[2] = g.z;
          if (c0 + 3 < N) dp[3] = g.w;
        }
      }
      if (c0 < Np)
        *reinterpret_cast<short4 *>(&X[r][c0]) = make_short4(ch_bf16(g.x), ch_bf16(g.y), ch_bf16(g.z), ch_bf16(g.w));
    }
    // parameter gradients of the row phase: combine the four waves in LDS, one atomic per column and workgroup
    if (ln || scaled) {
      if (c0 < N) {
        *reinterpret_cast<float4 *>(&red[0][wv][c0]) = ln ? pg : ps;
        *reinterpret_cast<float4 *>(&red[1][wv][c0]) = pb;
      }
    }
    __syncthreads();  // X (bf16 dY) complete, G free, red complete
    if (ln || scaled) {
      if (tid < N) {
        const float s0 = red[0][0][tid] + red[0][1][tid] + red[0][2][tid] + red[0][3][tid];
        if (ln) {
          const float s1 = red[1][0][tid] + red[1][1][tid] + red[1][2][tid] + red[1][3][tid];
          if (L.dgamma) atomicAdd(L.dgamma + tid, s0);
          if (L.dbeta) atomicAdd(L.dbeta + tid, s1);
        } else if (c.dscale) {
          atomicAdd(c.dscale + tid, s0);
        }
      }
    }
    const bool need_dx = l > 0 || c.dx != nullptr;
    if (need_dx) {
      // dX[r][k] = sum_n dY[r][n] Wt[k][n]: output columns k < K, reduction over n < N
      f32x4 acc[CH_MAXT][R / 16];
      ch_mma_any<R, false>(acc, bw, X, X, K, N, wv, l15, quad);
      const int ktiles = (K + 15) >> 4;
#pragma unroll
      for (int tt = 0; tt < CH_MAXT; ++tt) {
        const int t = wv + 4 * tt;
        if (t < ktiles) {
          const int col = 16 * t + l15;
#pragma unroll
          for (int i = 0; i < R / 16; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int rr = 16 * i + 4 * quad + r;
              if (l > 0) G[rr][col] = acc[tt][i][r];
              else if (col < K && r0 + rr < M) c.dx[(size_t)(r0 + rr) * c.lddx + col] = acc[tt][i][r];
            }
        }
      }
    }
    if (l > 1 || (l == 1 && c.dx)) ch_fetch_any(bw, c.L[l - 1].wt, c.L[l - 1].K, c.L[l - 1].N, wv, lane);
  }
}

static unsigned long long *g_stamps = nullptr;

static int ch_check_dims(int K, int N) { return (K >= 1 && K <= CH_W && N >= 1 && N <= CH_W) ? 1 : 0; }

}  // namespace hipad

using namespace hipad;

extern "C" {

void hipad_chain_debug_stamps(unsigned long long *device_buffer) { g_stamps = device_buffer; }

// every argument check of a call happens before its first launch (an EINVAL must not leave half the batches enqueued)
static int check_chain(const hipad_chain &s) {
  if (s.nlayers < 1 || s.nlayers > HIPAD_CHAIN_MAX_LAYERS || s.M <= 0 || !s.x0 || !s.out) return HIPAD_EINVAL;
  for (int l = 0; l < s.nlayers; ++l) {
    const hipad_chain_layer &sl = s.layers[l];
    if (!ch_check_dims(sl.K, sl.N) || !sl.w) return HIPAD_EINVAL;
    if (l > 0 && sl.K != s.layers[l - 1].N) return HIPAD_EINVAL;
    if ((uintptr_t)sl.w & 15) return HIPAD_EINVAL;
    if ((sl.flags & 2) && l + 1 == s.nlayers && s.out_scale) return HIPAD_EINVAL;  // Scale after a LayerNorm: unused
  }
  return HIPAD_OK;
}

static int check_chain_grad(const hipad_chain_grad &s) {
  if (s.nlayers < 1 || s.nlayers > HIPAD_CHAIN_MAX_LAYERS || s.M <= 0 || !s.dout || !s.dy || !s.save) return HIPAD_EINVAL;
  for (int l = 0; l < s.nlayers; ++l) {
    const hipad_chain_grad_layer &sl = s.layers[l];
    if (!ch_check_dims(sl.K, sl.N)) return HIPAD_EINVAL;
    const bool need_wt = l > 0 || s.dx;
    if (need_wt && (!sl.wt || ((uintptr_t)sl.wt & 15))) return HIPAD_EINVAL;
  }
  return HIPAD_OK;
}

int hipad_chain_forward(const hipad_chain *chains, int nchains, hipad_stream_t stream_) {
  if (!chains || nchains <= 0) return HIPAD_EINVAL;
  for (int i = 0; i < nchains; ++i)
    if (check_chain(chains[i]) != HIPAD_OK) return HIPAD_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  int maxM = 0;
  for (int i = 0; i < nchains; ++i) maxM = chains[i].M > maxM ? chains[i].M : maxM;
  const int R = maxM >= 2048 ? 32 : 16;
  for (int base = 0; base < nchains; base += HIPAD_CHAIN_MAX_CHAINS) {
    const int n = nchains - base < HIPAD_CHAIN_MAX_CHAINS ? nchains - base : HIPAD_CHAIN_MAX_CHAINS;
    ChainFwdArgs a;
    a.nchains = n;
    a.pad = 0;
    a.stamps = g_stamps;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
      const hipad_chain &s = chains[base + i];
      if (s.nlayers < 1 || s.nlayers > HIPAD_CHAIN_MAX_LAYERS || s.M <= 0 || !s.x0 || !s.out) return HIPAD_EINVAL;
      ChainFwd &c = a.c[i];
      c.x0 = s.x0; c.x1 = s.x1; c.xsum = s.xsum; c.out = s.out; c.out_scale = s.out_scale; c.residual = s.residual;
      c.save = s.save;
      c.ldx0 = s.ldx0; c.ldx1 = s.ldx1; c.ldo = s.ldo; c.ldr = s.ldr;
      c.M = s.M; c.nlayers = s.nlayers; c.tile0 = tiles; c.pad = 0;
      a.tile0[i] = tiles;
      tiles += (s.M + R - 1) / R;
      for (int l = 0; l < s.nlayers; ++l) {
        const hipad_chain_layer &sl = s.layers[l];
        if (!ch_check_dims(sl.K, sl.N) || !sl.w) return HIPAD_EINVAL;
        if (l > 0 && sl.K != s.layers[l - 1].N) return HIPAD_EINVAL;
        if ((uintptr_t)sl.w & 15) return HIPAD_EINVAL;
        if ((sl.flags & 2) && l + 1 == s.nlayers && s.out_scale) return HIPAD_EINVAL;  // Scale after a LayerNorm: unused
        ChainFwdLayer &L = c.L[l];
        L.w = sl.w; L.bias = sl.bias; L.gamma = sl.gamma; L.beta = sl.beta;
        L.off_h = s.save ? sl.off_h : CH_NONE; L.off_y = s.save ? sl.off_y : CH_NONE;
        L.off_stats = s.save ? sl.off_stats : CH_NONE;
        L.K = (unsigned short)sl.K; L.N = (unsigned short)sl.N; L.flags = (unsigned)sl.flags; L.eps = sl.eps;
      }
    }
    if (R == 32) hipLaunchKernelGGL((chain_fwd_kernel<32>), dim3(tiles), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((chain_fwd_kernel<16>), dim3(tiles), dim3(256), 0, stream, a);
  }
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_chain_backward_dx(const hipad_chain_grad *chains, int nchains, hipad_stream_t stream_) {
  if (!chains || nchains <= 0) return HIPAD_EINVAL;
  for (int i = 0; i < nchains; ++i)
    if (check_chain_grad(chains[i]) != HIPAD_OK) return HIPAD_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  int maxM = 0;
  for (int i = 0; i < nchains; ++i) maxM = chains[i].M > maxM ? chains[i].M : maxM;
  const int R = maxM >= 2048 ? 32 : 16;
  for (int base = 0; base < nchains; base += HIPAD_CHAIN_MAX_CHAINS) {
    const int n = nchains - base < HIPAD_CHAIN_MAX_CHAINS ? nchains - base : HIPAD_CHAIN_MAX_CHAINS;
    ChainBwdArgs a;
    a.nchains = n;
    a.pad = 0;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
      const hipad_chain_grad &s = chains[base + i];
      if (s.nlayers < 1 || s.nlayers > HIPAD_CHAIN_MAX_LAYERS || s.M <= 0 || !s.dout || !s.dy || !s.save)
        return HIPAD_EINVAL;
      ChainBwd &c = a.c[i];
      c.dout = s.dout; c.out_scale = s.out_scale; c.dscale = s.dscale; c.dx = s.dx; c.save = s.save; c.dy = s.dy;
      c.ldo = s.ldo; c.lddx = s.lddx; c.M = s.M; c.nlayers = s.nlayers; c.tile0 = tiles; c.pad = 0;
      a.tile0[i] = tiles;
      tiles += (s.M + R - 1) / R;
      for (int l = 0; l < s.nlayers; ++l) {
        const hipad_chain_grad_layer &sl = s.layers[l];
        if (!ch_check_dims(sl.K, sl.N)) return HIPAD_EINVAL;
        const bool need_wt = l > 0 || s.dx;
        if (need_wt && (!sl.wt || ((uintptr_t)sl.wt & 15))) return HIPAD_EINVAL;
        ChainBwdLayer &L = c.L[l];
        L.wt = sl.wt; L.gamma = sl.gamma; L.dgamma = sl.dgamma; L.dbeta = sl.dbeta;
        L.off_h = sl.off_h; L.off_stats = sl.off_stats; L.off_dy = sl.off_dy;
        L.K = (unsigned short)sl.K; L.N = (unsigned short)sl.N; L.flags = (unsigned)sl.flags; L.eps = sl.eps;
      }
    }
    if (R == 32) hipLaunchKernelGGL((chain_bwd_kernel<32>), dim3(tiles), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((chain_bwd_kernel<16>), dim3(tiles), dim3(256), 0, stream, a);
  }
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

}  // extern "C"
