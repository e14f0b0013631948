// hip-ad_amd/csrc/gemm.hip -- dense linear layers of the decoder on the matrix cores (gfx950).
//
// Replaces: every torch.nn.Linear / mmcv.cnn.Linear call of the decoder blocks (reference
// models/blocks.py:32-42 linear_relu_ln stacks, attention in/out projections attention.py:27-34,
// AsymmetricFFN blocks.py:367-396, refinement heads, anchor encoders) -- in the reference these are
// cuBLAS fp32 GEMMs plus separate bias / ReLU / grad-accumulate kernels.
//
// Why hand-written: a stage-2 frame makes ~1200 Linear calls forward; with library GEMMs under bf16
// autocast each costs ~13 launches forward+backward (casts of input and weight, GEMM, bias, ReLU,
// two backward GEMMs, a transposed-gradient copy, a bias reduction, gradient accumulation) and the
// frame is bound by the ~5 us per-dispatch floor, not by flops.  Here a Linear is
//   forward : ONE kernel   Y = relu?(X W^T + b)     fp32 in/out, operands rounded to bf16 on the way
//                                                   into LDS, fp32 accumulation (v_mfma_f32_16x16x32_bf16)
//   backward: TWO kernels  dX = (dY o [Y>0]) W
//                          dW += (dY o [Y>0])^T X ,  db += colsum(dY o [Y>0])   (atomic accumulation
//                          straight into the caller's gradient buffers: no separate accumulate pass)
// Tile: 64 x 32 outputs per 256-thread workgroup, reduction in chunks of 256 (see below).
// Both operands go through LDS as [out index][reduction index] bf16 rows (528-byte stride), which
// makes the three products (NT, NN, TN) one code path with two tile loaders (reduction index
// contiguous in memory, or strided = transposed in registers on the way in).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hipad.h"
#include "wave_ops.h"
#include "daf_common.h"

namespace hipad {

using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// Tile: 64 rows of A_op x 32 rows of B_op per 256-thread workgroup (4 waves, each 16 x 32 = two MFMA tiles),
// reduction consumed in chunks of KC = 256: a whole chunk of both operands is fetched with ALL its loads in
// flight at once (24-40 float4 per thread), rounded to bf16 into LDS, and multiplied with 8 MFMA steps.
// The decoder's GEMMs are tiny (M <= 1500, K <= 1024, mostly 256 x 256) and latency-bound: one round trip
// to memory per 256 of reduction instead of one per 32, and 2x the workgroups of a 64 x 64 tile.
constexpr int BM = 64, BN = 32, KC = 256;
constexpr int LDS_STRIDE = KC + 8;  // bf16 per tile row (528 bytes: 16-byte aligned fragments)

__device__ __forceinline__ short to_bf16(float x) { return __builtin_bit_cast(short, (__bf16)x); }

// Four consecutive floats, branch-free: out-of-range elements read a clamped (valid) address and are
// zeroed by a select, so every load of a chunk is issued before the first one is waited for.
// VEC: p is 16-byte aligned and nvalid is 0 or 4 (guaranteed by the host when ld % 4 == 0).
template <bool VEC>
__device__ __forceinline__ float4 ld4(const float *__restrict__ p, const float *__restrict__ safe, int nvalid) {
  const float *q = nvalid > 0 ? p : safe;
  if (VEC) {
    const float4 v = *reinterpret_cast<const float4 *>(q);
    return nvalid > 0 ? v : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float x = q[0], y = q[nvalid > 1 ? 1 : 0], z = q[nvalid > 2 ? 2 : 0], w = q[nvalid > 3 ? 3 : 0];
  return make_float4(nvalid > 0 ? x : 0.f, nvalid > 1 ? y : 0.f, nvalid > 2 ? z : 0.f, nvalid > 3 ? w : 0.f);
}

__device__ __forceinline__ float4 gate4(float4 v, float4 g) {
  return make_float4(g.x > 0.f ? v.x : 0.f, g.y > 0.f ? v.y : 0.f, g.z > 0.f ? v.z : 0.f, g.w > 0.f ? v.w : 0.f);
}

__device__ __forceinline__ bf16x8 pack8(float a, float b, float c, float d, float e, float f, float g, float h) {
  bf16x8 p;
  p[0] = to_bf16(a); p[1] = to_bf16(b); p[2] = to_bf16(c); p[3] = to_bf16(d);
  p[4] = to_bf16(e); p[5] = to_bf16(f); p[6] = to_bf16(g); p[7] = to_bf16(h);
  return p;
}

// ---- operand chunk, reduction index CONTIGUOUS in memory: T[r][kk] = src[(r0 + r) * ld + k0 + kk]
// thread -> 8 consecutive kk (kq = tid & 31), rows rr + 8 p (rr = tid >> 5): a wave reads 2 x 1 KiB rows.
template <int ROWS, bool VEC>
__device__ __forceinline__ void gload_rowmajor(float4 (&v)[ROWS / 4], const float *__restrict__ src, int ld, int r0,
                                               int k0, int rows, int kend, int tid) {
  const int kk = (tid & 31) * 8, rr = tid >> 5;
  const int gk = k0 + kk;
  const int n0 = max(0, min(4, kend - gk)), n1 = max(0, min(4, kend - gk - 4));
#pragma unroll
  for (int p = 0; p < ROWS / 8; ++p) {
    const int gr = r0 + rr + 8 * p;
    const bool in = gr < rows;
    const float *q = src + (size_t)(in ? gr : 0) * ld + gk;
    v[2 * p] = ld4<VEC>(q, src, in ? n0 : 0);
    v[2 * p + 1] = ld4<VEC>(q + 4, src, in ? n1 : 0);
  }
}
template <int ROWS>
__device__ __forceinline__ void lstore_rowmajor(short (*T)[LDS_STRIDE], const float4 (&v)[ROWS / 4], int tid) {
  const int kk = (tid & 31) * 8, rr = tid >> 5;
#pragma unroll
  for (int p = 0; p < ROWS / 8; ++p) {
    const float4 a = v[2 * p], b = v[2 * p + 1];
    *reinterpret_cast<bf16x8 *>(&T[rr + 8 * p][kk]) = pack8(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
  }
}

// the same with the activation kept as a hi + lo bf16 PAIR (x = hi + lo + O(2^-17 x)): the MFMA then sees ~16 mantissa
// bits of the activation and only the weight is rounded to bf16 -- as the MLP-chain kernels do (chain.hip)
__device__ __forceinline__ float bf16_f(short h) { return (float)__builtin_bit_cast(__bf16, h); }
template <int ROWS>
__device__ __forceinline__ void lstore_rowmajor_hilo(short (*T)[LDS_STRIDE], short (*TL)[LDS_STRIDE],
                                                     const float4 (&v)[ROWS / 4], int tid) {
  const int kk = (tid & 31) * 8, rr = tid >> 5;
#pragma unroll
  for (int p = 0; p < ROWS / 8; ++p) {
    const float4 a = v[2 * p], b = v[2 * p + 1];
    const bf16x8 hi = pack8(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
    *reinterpret_cast<bf16x8 *>(&T[rr + 8 * p][kk]) = hi;
    *reinterpret_cast<bf16x8 *>(&TL[rr + 8 * p][kk]) =
        pack8(a.x - bf16_f(hi[0]), a.y - bf16_f(hi[1]), a.z - bf16_f(hi[2]), a.w - bf16_f(hi[3]), b.x - bf16_f(hi[4]),
              b.y - bf16_f(hi[5]), b.z - bf16_f(hi[6]), b.w - bf16_f(hi[7]));
  }
}

// ---- operand chunk, reduction index STRIDED in memory: T[r][kk] = src[(k0 + kk) * ld + r0 + r]
// thread -> blocks of 4 r x 8 kk (8 float4 loads along r, transposed in registers, 4 16-byte LDS stores)
template <int ROWS, bool VEC>
__device__ __forceinline__ void gload_trans(float4 (&v)[ROWS / 4], const float *__restrict__ src, int ld, int r0, int k0,
                                            int rows, int kend, int tid) {
  constexpr int RG = ROWS / 4;  // r-groups per tile
#pragma unroll
  for (int j = 0; j < ROWS / 32; ++j) {
    const int id = tid + 256 * j;
    const int rg = id % RG, kg = id / RG;
    const int gr = r0 + 4 * rg;
    const int nr = max(0, min(4, rows - gr));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int gk = k0 + 8 * kg + i;
      const bool in = gk < kend;
      v[8 * j + i] = ld4<VEC>(src + (size_t)(in ? gk : 0) * ld + (nr > 0 ? gr : 0), src, in ? nr : 0);
    }
  }
}
template <int ROWS>
__device__ __forceinline__ void lstore_trans(short (*T)[LDS_STRIDE], const float4 (&v)[ROWS / 4], int tid) {
  constexpr int RG = ROWS / 4;
#pragma unroll
  for (int j = 0; j < ROWS / 32; ++j) {
    const int id = tid + 256 * j;
    const int rg = id % RG, kg = id / RG;
    const float4 *w = &v[8 * j];
    *reinterpret_cast<bf16x8 *>(&T[4 * rg + 0][8 * kg]) = pack8(w[0].x, w[1].x, w[2].x, w[3].x, w[4].x, w[5].x, w[6].x, w[7].x);
    *reinterpret_cast<bf16x8 *>(&T[4 * rg + 1][8 * kg]) = pack8(w[0].y, w[1].y, w[2].y, w[3].y, w[4].y, w[5].y, w[6].y, w[7].y);
    *reinterpret_cast<bf16x8 *>(&T[4 * rg + 2][8 * kg]) = pack8(w[0].z, w[1].z, w[2].z, w[3].z, w[4].z, w[5].z, w[6].z, w[7].z);
    *reinterpret_cast<bf16x8 *>(&T[4 * rg + 3][8 * kg]) = pack8(w[0].w, w[1].w, w[2].w, w[3].w, w[4].w, w[5].w, w[6].w, w[7].w);
  }
}

// C[m][n] = sum_k A_op[m][k] * B_op[n][k]
//   A_TRANS: A_op[m][k] = A[k * lda + m]  else A[m * lda + k]   (optional ReLU gate on A, same layout)
//   B_TRANS: B_op[n][k] = B[k * ldb + n]  else B[n * ldb + k]
//   EPI 0: C = relu?(acc + bias[n])  stored;   EPI 1: atomicAdd(C, acc) + optional row sums of A_op
//   grid = (ceil(N/32), ceil(M/64), splits over the reduction); k_per_split is a multiple of KC
// One 64 x 32 output tile (tile coordinates bx, by, split bz) of the product; TA / TB: the workgroup's LDS.
template <bool A_TRANS, bool B_TRANS, int EPI, bool VA, bool VB, bool HILO = false>
__device__ __forceinline__ void gemm_tile(short (*TA)[LDS_STRIDE], short (*TB)[LDS_STRIDE], float *__restrict__ C,
                                          const float *__restrict__ A, const float *__restrict__ A_gate,
                                          const float *__restrict__ B, const float *__restrict__ bias,
                                          float *__restrict__ rowsum_out, int M, int N, int K, int lda, int ldb, int ldc,
                                          int relu, int k_per_split, int bx, int by, int bz,
                                          short (*TAL)[LDS_STRIDE] = nullptr /* HILO: low halves of A_op */) {
  static_assert(!HILO || !A_TRANS, "hi + lo activations: row-major A only");
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  const int m0 = by * BM, n0 = bx * BN;
  const int kbeg = bz * k_per_split, kend = min(K, kbeg + k_per_split);
  f32x4 acc[2];
  acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
  acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float rsum = 0.f;

  float4 ra[BM / 4], rb[BN / 4];
  auto fetch = [&](int k0) {
    if (A_TRANS) gload_trans<BM, VA>(ra, A, lda, m0, k0, M, kend, tid);
    else gload_rowmajor<BM, VA>(ra, A, lda, m0, k0, M, kend, tid);
    if (A_gate) {
      float4 rg[BM / 4];
      if (A_TRANS) gload_trans<BM, VA>(rg, A_gate, lda, m0, k0, M, kend, tid);
      else gload_rowmajor<BM, VA>(rg, A_gate, lda, m0, k0, M, kend, tid);
#pragma unroll
      for (int i = 0; i < BM / 4; ++i) ra[i] = gate4(ra[i], rg[i]);
    }
    if (B_TRANS) gload_trans<BN, VB>(rb, B, ldb, n0, k0, N, kend, tid);
    else gload_rowmajor<BN, VB>(rb, B, ldb, n0, k0, N, kend, tid);
  };

  fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += KC) {
    if (A_TRANS) lstore_trans<BM>(TA, ra, tid);
    else if (HILO) lstore_rowmajor_hilo<BM>(TA, TAL, ra, tid);
    else lstore_rowmajor<BM>(TA, ra, tid);
    if (B_TRANS) lstore_trans<BN>(TB, rb, tid); else lstore_rowmajor<BN>(TB, rb, tid);
    __syncthreads();
    if (k0 + KC < kend) fetch(k0 + KC);  // next chunk's loads fly during the MFMA phase
    const int nsteps = (min(KC, kend - k0) + 31) >> 5;
    for (int s = 0; s < nsteps; ++s) {
      const bf16x8 a = *reinterpret_cast<const bf16x8 *>(&TA[16 * wv + l15][32 * s + 8 * quad]);
      const bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(&TB[l15][32 * s + 8 * quad]);
      const bf16x8 b1 = *reinterpret_cast<const bf16x8 *>(&TB[16 + l15][32 * s + 8 * quad]);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b1, acc[1], 0, 0, 0);
      if (HILO) {
        const bf16x8 al = *reinterpret_cast<const bf16x8 *>(&TAL[16 * wv + l15][32 * s + 8 * quad]);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, b1, acc[1], 0, 0, 0);
      }
    }
    if (EPI == 1 && rowsum_out && bx == 0) {
      // sum over the chunk of A_op[row = tid/4][.]: each thread a quarter of the row, then 2 shuffles
      const int row = tid >> 2, q = tid & 3;
      float s8 = 0.f;
#pragma unroll
      for (int i = 0; i < KC / 32; ++i) {
        const bf16x8 t = *reinterpret_cast<const bf16x8 *>(&TA[row][64 * q + 8 * i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s8 += (float)__builtin_bit_cast(__bf16, (short)t[e]);
      }
      rsum += quad_sum(s8);
    }
    __syncthreads();
  }
  // C fragment j: row = m0 + 16 wv + 4 quad + r ; col = n0 + 16 j + l15
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + 16 * j + l15;
    if (col >= N) continue;
    const float bv = (EPI == 0 && bias) ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 16 * wv + 4 * quad + r;
      if (row >= M) continue;
      float v = acc[j][r];
      if (EPI == 0) {
        v += bv;
        if (relu) v = fmaxf(v, 0.f);
        C[(size_t)row * ldc + col] = v;
      } else {
        atomicAdd(C + (size_t)row * ldc + col, v);
      }
    }
  }
  if (EPI == 1 && rowsum_out && bx == 0 && (tid & 3) == 0 && m0 + (tid >> 2) < M)
    atomicAdd(rowsum_out + m0 + (tid >> 2), rsum);
}

template <bool A_TRANS, bool B_TRANS, int EPI, bool VA, bool VB>
__global__ __launch_bounds__(256) void gemm_kernel(float *__restrict__ C, const float *__restrict__ A,
                                                   const float *__restrict__ A_gate, const float *__restrict__ B,
                                                   const float *__restrict__ bias, float *__restrict__ rowsum_out,
                                                   int M, int N, int K, int lda, int ldb, int ldc, int relu,
                                                   int k_per_split) {
  __shared__ short TA[BM][LDS_STRIDE];
  __shared__ short TB[BN][LDS_STRIDE];
  gemm_tile<A_TRANS, B_TRANS, EPI, VA, VB>(TA, TB, C, A, A_gate, B, bias, rowsum_out, M, N, K, lda, ldb, ldc, relu,
                                           k_per_split, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Forward Linear with hi + lo activations: Y = relu?(X W^T + b), X row-major.
template <bool VA, bool VB>
__global__ __launch_bounds__(256) void gemm_fwd_hilo_kernel(float *__restrict__ C, const float *__restrict__ A,
                                                            const float *__restrict__ B, const float *__restrict__ bias, int M,
                                                            int N, int K, int lda, int ldb, int ldc, int relu, int k_per_split) {
  __shared__ short TA[BM][LDS_STRIDE];
  __shared__ short TAL[BM][LDS_STRIDE];
  __shared__ short TB[BN][LDS_STRIDE];
  gemm_tile<false, false, 0, VA, VB, true>(TA, TB, C, A, nullptr, B, bias, nullptr, M, N, K, lda, ldb, ldc, relu, k_per_split,
                                           blockIdx.x, blockIdx.y, blockIdx.z, TAL);
}

// Backward of a Linear in ONE launch: the first `dx_tiles` workgroups compute dX = (dY o gate) W (overwrite), the
// others dW += (dY o gate)^T X and db += colsum(dY o gate) (atomics).  Two launches per layer were two trips through
// the ~4 us dispatch floor that dominates these tiny products.
//   VDY / VW / VX: 16-byte vector loads allowed on dY (and the gate), W, X.
struct LinBwdArgs {
  float *dx, *dw, *db;
  const float *dy, *gate, *x, *w;
  int M, N, K;
  int dx_tiles_x, dx_tiles;          // dX grid: dx_tiles_x column tiles, dx_tiles in total
  int dw_tiles_x, dw_tiles_xy;       // dW grid: column tiles, tiles per split
  int dx_kper, dw_kper;
};

template <bool VDY, bool VW, bool VX>
__global__ __launch_bounds__(256) void linear_bwd_fused_kernel(LinBwdArgs a) {
  __shared__ short TA[BM][LDS_STRIDE];
  __shared__ short TB[BN][LDS_STRIDE];
  const int b = blockIdx.x;
  if (b < a.dx_tiles) {
    // dX[m][k] = sum_n dYg[m][n] W[n][k]: A = dY row-major, B_op[k][n] = W[n*K + k] (strided)
    gemm_tile<false, true, 0, VDY, VW>(TA, TB, a.dx, a.dy, a.gate, a.w, nullptr, nullptr, a.M, a.K, a.N, a.N, a.K, a.K,
                                       0, a.dx_kper, b % a.dx_tiles_x, b / a.dx_tiles_x, 0);
  } else {
    // dW[n][k] += sum_m dYg[m][n] X[m][k]: A_op[n][m] = dY[m*N + n], B_op[k][m] = X[m*K + k] (both strided)
    const int t = b - a.dx_tiles;
    const int bz = t / a.dw_tiles_xy, r = t - bz * a.dw_tiles_xy;
    gemm_tile<true, true, 1, VDY, VX>(TA, TB, a.dw, a.dy, a.gate, a.x, nullptr, a.db, a.N, a.K, a.M, a.N, a.K, a.K, 0,
                                      a.dw_kper, r % a.dw_tiles_x, r / a.dw_tiles_x, bz);
  }
}

// ------------------------------------------------------------------------------------------------------------
// Linear + ReLU + LayerNorm forward in one launch (the [Linear, ReLU, LayerNorm] unit of every linear_relu_ln stack,
// reference models/blocks.py:32-42): a workgroup owns 32 full rows (N <= 256 outputs), so the LayerNorm statistics
// are available in its epilogue.  Writes xr = relu(x W^T + b) (kept for the backward) and y = LN(xr).
//   CT = 16-column MFMA tiles per wave (N = 64 * CT for CT in {1,2,4}; N = 32/16: CT = 1 with idle waves).
// Reduction in chunks of 128; weights are re-read from L2 by every workgroup (<= 256 KiB).
// ------------------------------------------------------------------------------------------------------------
constexpr int LR = 32;            // rows per workgroup
constexpr int KC2 = 128;          // reduction chunk
constexpr int LDS2 = KC2 + 8;

template <int ROWS>
__device__ __forceinline__ void gload_rm128(float4 (&v)[ROWS / 8], const float *__restrict__ src, int ld, int r0, int k0,
                                            int rows, int kend, int tid) {
  const int kk = (tid & 15) * 8, rr = tid >> 4;  // 16 threads per row, 16 rows per pass
  const int gk = k0 + kk;
  const int n0 = max(0, min(4, kend - gk)), n1 = max(0, min(4, kend - gk - 4));
#pragma unroll
  for (int p = 0; p < ROWS / 16; ++p) {
    const int gr = r0 + rr + 16 * p;
    const bool in = gr < rows;
    const float *q = src + (size_t)(in ? gr : 0) * ld + gk;
    v[2 * p] = ld4<true>(q, src, in ? n0 : 0);
    v[2 * p + 1] = ld4<true>(q + 4, src, in ? n1 : 0);
  }
}
template <int ROWS>
__device__ __forceinline__ void lstore_rm128(short (*T)[LDS2], const float4 (&v)[ROWS / 8], int tid) {
  const int kk = (tid & 15) * 8, rr = tid >> 4;
#pragma unroll
  for (int p = 0; p < ROWS / 16; ++p) {
    const float4 a = v[2 * p], b = v[2 * p + 1];
    *reinterpret_cast<bf16x8 *>(&T[rr + 16 * p][kk]) = pack8(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
  }
}

template <int CT>
__global__ __launch_bounds__(256) void linear_relu_ln_fwd_kernel(
    float *__restrict__ y, float *__restrict__ xr, float *__restrict__ mean_out, float *__restrict__ rstd_out,
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
    const float *__restrict__ gamma, const float *__restrict__ beta, int M, int N, int K, float eps) {
  constexpr int NB = 64 * CT;  // weight rows staged in LDS (>= N)
  __shared__ short TA[LR][LDS2];
  __shared__ short TB[NB][LDS2];
  __shared__ float red[4][LR];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  const int m0 = blockIdx.x * LR;
  f32x4 acc[2][CT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 ra[LR / 8], rb[NB / 8];
  gload_rm128<LR>(ra, x, K, m0, 0, M, K, tid);
  gload_rm128<NB>(rb, w, K, 0, 0, N, K, tid);
  for (int k0 = 0; k0 < K; k0 += KC2) {
    lstore_rm128<LR>(TA, ra, tid);
    lstore_rm128<NB>(TB, rb, tid);
    __syncthreads();
    if (k0 + KC2 < K) {
      gload_rm128<LR>(ra, x, K, m0, k0 + KC2, M, K, tid);
      gload_rm128<NB>(rb, w, K, 0, k0 + KC2, N, K, tid);
    }
    const int nsteps = (min(KC2, K - k0) + 31) >> 5;
    for (int s = 0; s < nsteps; ++s) {
      const bf16x8 a0 = *reinterpret_cast<const bf16x8 *>(&TA[l15][32 * s + 8 * quad]);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8 *>(&TA[16 + l15][32 * s + 8 * quad]);
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        const bf16x8 b = *reinterpret_cast<const bf16x8 *>(&TB[16 * (wv * CT + j) + l15][32 * s + 8 * quad]);
        acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b, acc[1][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // element (i, j, r): row = 16 i + 4 quad + r, col = 16 (wv CT + j) + l15
  float v[2][CT][4];
  float psum[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) psum[i][r] = 0.f;
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int col = 16 * (wv * CT + j) + l15;
    const bool cok = col < N;
    const float bv = (bias && cok) ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t = cok ? fmaxf(acc[i][j][r] + bv, 0.f) : 0.f;
        v[i][j][r] = t;
        psum[i][r] += t;
      }
  }
  auto row_total = [&](float (&ps)[2][4], float (&tot)[2][4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = ps[i][r];
        t = row_sum(t);  // the 16 lanes l15 = 0..15 of one DPP row (wave_ops.h)
        if (l15 == 0) red[wv][16 * i + 4 * quad + r] = t;
      }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * i + 4 * quad + r;
        tot[i][r] = red[0][row] + red[1][row] + red[2][row] + red[3][row];
      }
    __syncthreads();
  };
  float mean[2][4], var[2][4];
  row_total(psum, mean);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      mean[i][r] /= (float)N;
      psum[i][r] = 0.f;
    }
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const bool cok = 16 * (wv * CT + j) + l15 < N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = cok ? v[i][j][r] - mean[i][r] : 0.f;
        psum[i][r] += d * d;
      }
  }
  row_total(psum, var);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) var[i][r] = rsqrtf(var[i][r] / (float)N + eps);  // var now holds rstd
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int col = 16 * (wv * CT + j) + l15;
    if (col >= N) continue;
    const float gm = gamma ? gamma[col] : 1.f, bt = beta ? beta[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + 16 * i + 4 * quad + r;
        if (row >= M) continue;
        xr[(size_t)row * N + col] = v[i][j][r];
        y[(size_t)row * N + col] = (v[i][j][r] - mean[i][r]) * var[i][r] * gm + bt;
      }
  }
  if (wv == 0 && l15 == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + 16 * i + 4 * quad + r;
        if (row < M) {
          mean_out[row] = mean[i][r];
          rstd_out[row] = var[i][r];
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Weight / bias gradients of every layer of a group of MLP chains (chain.hip) in ONE grid: entry e computes
// dW_e[n][k] += sum_m dY_e[m][n] X_e[m][k] and db_e[n] += sum_m dY_e[m][n] with the transposed-operand tile loaders
// above (one 64 x 32 tile of dW and one 256-row slab of the reduction per workgroup, fp32 atomics into the
// caller's gradient buffers).  dY is already gated by the chain's reverse sweep.
// ------------------------------------------------------------------------------------------------------------
struct ChainDwEntry {
  const float *dy, *x;
  float *dw, *db;
  int M, N, K, ldx;
  int wg0, tiles_x, tiles_xy, vec;
};
struct ChainDwArgs {
  int n, pad;
  ChainDwEntry e[HIPAD_CHAIN_MAX_DW];
};

__global__ __launch_bounds__(256) void chain_dw_kernel(const ChainDwArgs a) {
  __shared__ short TA[BM][LDS_STRIDE];
  __shared__ short TB[BN][LDS_STRIDE];
  const int b = blockIdx.x;
  int ei = 0;
  for (int i = 1; i < a.n; ++i)
    if (b >= a.e[i].wg0) ei = i;
  const ChainDwEntry &e = a.e[ei];
  const int t = b - e.wg0;
  const int bz = t / e.tiles_xy, r = t - bz * e.tiles_xy;
  if (e.vec)
    gemm_tile<true, true, 1, true, true>(TA, TB, e.dw, e.dy, nullptr, e.x, nullptr, e.db, e.N, e.K, e.M, e.N, e.ldx, e.K, 0,
                                         KC, r % e.tiles_x, r / e.tiles_x, bz);
  else
    gemm_tile<true, true, 1, false, false>(TA, TB, e.dw, e.dy, nullptr, e.x, nullptr, e.db, e.N, e.K, e.M, e.N, e.ldx, e.K, 0,
                                           KC, r % e.tiles_x, r / e.tiles_x, bz);
}

static inline int vec_ok(const void *p, int ld) { return (((uintptr_t)p & 15) == 0 && (ld & 3) == 0) ? 1 : 0; }

// launch gemm_kernel<AT, BT, EPI, va, vb> with the two vector-load flags resolved at run time
#define HIPAD_GEMM(AT, BT, EPI, va, vb, grid, stream, ...)                                                      \
  do {                                                                                                          \
    if (va) {                                                                                                   \
      if (vb) hipLaunchKernelGGL((gemm_kernel<AT, BT, EPI, true, true>), grid, dim3(256), 0, stream, __VA_ARGS__);   \
      else hipLaunchKernelGGL((gemm_kernel<AT, BT, EPI, true, false>), grid, dim3(256), 0, stream, __VA_ARGS__);     \
    } else {                                                                                                    \
      if (vb) hipLaunchKernelGGL((gemm_kernel<AT, BT, EPI, false, true>), grid, dim3(256), 0, stream, __VA_ARGS__);  \
      else hipLaunchKernelGGL((gemm_kernel<AT, BT, EPI, false, false>), grid, dim3(256), 0, stream, __VA_ARGS__);    \
    }                                                                                                           \
  } while (0)

static int check_lin(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return HIPAD_EINVAL;
  if ((long long)M * N >= (1ll << 31) || (long long)M * K >= (1ll << 31) || (long long)N * K >= (1ll << 31))
    return HIPAD_ERANGE;
  return HIPAD_OK;
}

}  // namespace hipad

using namespace hipad;

extern "C" {

int hipad_linear_forward(float *y, const float *x, const float *weight, const float *bias, int M, int N, int K,
                         int relu, hipad_stream_t stream) {
  int rc = check_lin(M, N, K);
  if (rc != HIPAD_OK) return rc;
  if (!y || !x || !weight) return HIPAD_EINVAL;
  const dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, 1);
  const int kall = (K + KC - 1) / KC * KC;
  // C[m][n] = sum_k X[m][k] W[n][k], activations as hi + lo bf16 pairs (only the weights are rounded to bf16)
  const bool va = vec_ok(x, K), vb = vec_ok(weight, K);
#define HIPAD_FWD_HILO(VA_, VB_)                                                                                      \
  hipLaunchKernelGGL((gemm_fwd_hilo_kernel<VA_, VB_>), grid, dim3(256), 0, (hipStream_t)stream, y, x, weight, bias, M, N, \
                     K, K, K, N, relu, kall)
  if (va) { if (vb) HIPAD_FWD_HILO(true, true); else HIPAD_FWD_HILO(true, false); }
  else { if (vb) HIPAD_FWD_HILO(false, true); else HIPAD_FWD_HILO(false, false); }
#undef HIPAD_FWD_HILO
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_linear_relu_ln_supported(int N, int K) { return (N >= 16 && N <= 256 && N % 16 == 0 && K % 4 == 0) ? 1 : 0; }

int hipad_linear_relu_ln_forward(float *y, float *x_relu, float *mean, float *rstd, const float *x, const float *weight,
                                 const float *bias, const float *gamma, const float *beta, int M, int N, int K,
                                 float eps, hipad_stream_t stream) {
  int rc = check_lin(M, N, K);
  if (rc != HIPAD_OK) return rc;
  if (!y || !x_relu || !mean || !rstd || !x || !weight) return HIPAD_EINVAL;
  if (!hipad_linear_relu_ln_supported(N, K) || !vec_ok(x, K) || !vec_ok(weight, K)) return HIPAD_EINVAL;
  const dim3 grid((M + LR - 1) / LR);
#define HIPAD_LRL(CT_) hipLaunchKernelGGL((linear_relu_ln_fwd_kernel<CT_>), grid, dim3(256), 0, (hipStream_t)stream, y, \
                                           x_relu, mean, rstd, x, weight, bias, gamma, beta, M, N, K, eps)
  if (N <= 64) HIPAD_LRL(1); else if (N <= 128) HIPAD_LRL(2); else HIPAD_LRL(4);
#undef HIPAD_LRL
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_chain_backward_dw(const hipad_chain_dw *entries, int nentries, hipad_stream_t stream_) {
  if (!entries || nentries <= 0) return HIPAD_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  for (int base = 0; base < nentries; base += HIPAD_CHAIN_MAX_DW) {
    const int n = nentries - base < HIPAD_CHAIN_MAX_DW ? nentries - base : HIPAD_CHAIN_MAX_DW;
    ChainDwArgs a;
    a.n = n;
    a.pad = 0;
    long long wgs = 0;
    for (int i = 0; i < n; ++i) {
      const hipad_chain_dw &s = entries[base + i];
      int rc = check_lin(s.M, s.N, s.K);
      if (rc != HIPAD_OK) return rc;
      if (!s.dy || !s.x || !s.dw || s.ldx < s.K) return HIPAD_EINVAL;
      ChainDwEntry &e = a.e[i];
      e.dy = s.dy; e.x = s.x; e.dw = s.dw; e.db = s.db;
      e.M = s.M; e.N = s.N; e.K = s.K; e.ldx = s.ldx;
      e.tiles_x = (s.K + BN - 1) / BN;
      e.tiles_xy = e.tiles_x * ((s.N + BM - 1) / BM);
      e.vec = vec_ok(s.dy, s.N) & vec_ok(s.x, s.ldx);
      e.wg0 = (int)wgs;
      wgs += (long long)e.tiles_xy * ((s.M + KC - 1) / KC);
      if (wgs >= (1ll << 30)) return HIPAD_ERANGE;
    }
    hipLaunchKernelGGL(chain_dw_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, a);
  }
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_linear_backward(float *dx, float *dw, float *db, const float *dy, const float *y_relu,
                          const float *x, const float *weight, int M, int N, int K, hipad_stream_t stream_) {
  int rc = check_lin(M, N, K);
  if (rc != HIPAD_OK) return rc;
  if (!dy || !x || !weight) return HIPAD_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  const int vdy = vec_ok(dy, N) & (y_relu ? vec_ok(y_relu, N) : 1);
  if (dx && dw && N <= 4 * KC) {
    // the common case: both gradients, short reduction for dX -> one fused launch
    LinBwdArgs a;
    a.dx = dx; a.dw = dw; a.db = db; a.dy = dy; a.gate = y_relu; a.x = x; a.w = weight;
    a.M = M; a.N = N; a.K = K;
    a.dx_tiles_x = (K + BN - 1) / BN;
    a.dx_tiles = a.dx_tiles_x * ((M + BM - 1) / BM);
    a.dx_kper = (N + KC - 1) / KC * KC;
    a.dw_tiles_x = (K + BN - 1) / BN;
    a.dw_tiles_xy = a.dw_tiles_x * ((N + BM - 1) / BM);
    int per = KC;
    while ((long long)a.dw_tiles_xy * ((M + per - 1) / per) > 4096 && per < M) per += KC;
    a.dw_kper = per;
    const int splits = (M + per - 1) / per;
    const long long blocks = (long long)a.dx_tiles + (long long)a.dw_tiles_xy * splits;
    if (blocks < (1ll << 31)) {
      const dim3 grid((unsigned)blocks);
      const int vw = vec_ok(weight, K), vx = vec_ok(x, K);
#define HIPAD_LBF(A_, B_, C_) hipLaunchKernelGGL((linear_bwd_fused_kernel<A_, B_, C_>), grid, dim3(256), 0, stream, a)
      if (vdy) {
        if (vw) { if (vx) HIPAD_LBF(true, true, true); else HIPAD_LBF(true, true, false); }
        else { if (vx) HIPAD_LBF(true, false, true); else HIPAD_LBF(true, false, false); }
      } else {
        if (vw) { if (vx) HIPAD_LBF(false, true, true); else HIPAD_LBF(false, true, false); }
        else { if (vx) HIPAD_LBF(false, false, true); else HIPAD_LBF(false, false, false); }
      }
#undef HIPAD_LBF
      return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
    }
  }
  if (dx) {
    // dX[m][k] = sum_n dYm[m][n] W[n][k]  : A = dY (row-major over n), B_op[k][n] = W[n*K + k] (transposed)
    const int tiles = ((K + BN - 1) / BN) * ((M + BM - 1) / BM);
    int splits = 1;
    if (N > 4 * KC) {  // long reduction (the 256 -> 9600 / 2880 weight layers): split it, accumulate atomically
      splits = (N + 2 * KC - 1) / (2 * KC);
      while (splits > 1 && (long long)tiles * splits > 4096) --splits;
    }
    int per = (N + splits - 1) / splits;
    per = (per + KC - 1) / KC * KC;
    splits = (N + per - 1) / per;
    const dim3 grid((K + BN - 1) / BN, (M + BM - 1) / BM, splits);
    if (splits == 1) {
      HIPAD_GEMM(false, true, 0, vdy, vec_ok(weight, K), grid, stream, dx, dy, y_relu, weight,
                 (const float *)nullptr, (float *)nullptr, M, K, N, N, K, K, 0, per);
    } else {
      if (fill_zero(dx, (size_t)M * K * sizeof(float), stream) != HIPAD_OK) return HIPAD_ELAUNCH;
      HIPAD_GEMM(false, true, 1, vdy, vec_ok(weight, K), grid, stream, dx, dy, y_relu, weight,
                 (const float *)nullptr, (float *)nullptr, M, K, N, N, K, K, 0, per);
    }
  }
  if (dw || db) {
    if (!dw) return HIPAD_EINVAL;  // bias gradient alone is not needed by the model (see functional._Linear)
    // dW[n][k] += sum_m dYm[m][n] X[m][k] : A_op[n][m] = dY[m*N + n] (transposed), B_op[k][m] = X[m*K + k]
    // the reduction runs over the M rows: one chunk of KC rows per workgroup unless that makes too many
    const int tiles = ((N + BM - 1) / BM) * ((K + BN - 1) / BN);
    int per = KC;
    while ((long long)tiles * ((M + per - 1) / per) > 4096 && per < M) per += KC;
    const int splits = (M + per - 1) / per;
    const dim3 grid((K + BN - 1) / BN, (N + BM - 1) / BM, splits);
    HIPAD_GEMM(true, true, 1, vdy, vec_ok(x, K), grid, stream, dw, dy, y_relu, x,
               (const float *)nullptr, db, N, K, M, N, K, K, 0, per);
  }
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

}  // extern "C"
