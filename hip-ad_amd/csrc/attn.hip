// hip-ad_amd/csrc/attn.hip -- multi-head attention core for the decoder's query sets (gfx950).
//
// Replaces the flash-attn call of the reference (models/attention.py:76-80, 91-95:
// flash_attn_unpadded_kvpacked_func on fp16/bf16 q and packed kv; softmax(q k^T / sqrt(D)) v with
// dropout on the probabilities in training).  flash-attn is a CUDA-only third-party package, so
// this is a from-scratch CDNA4 kernel, shaped for the decoder's problem sizes: 1-2 samples,
// 8 heads, head_dim 32 or 64, 100..1481 queries x 100..1000 keys -- tiny: what matters is the
// critical path of one query tile, not peak MFMA rate.
//
// Layout: q [B, Nq, H*D], k / v [B, Nk, H*D], out [B, Nq, H*D] fp32 (the in/out projections are
// fp32 GEMMs around this kernel); operands are rounded to bf16 on load, products accumulate in
// fp32 on the matrix cores (v_mfma_f32_16x16x32_bf16), softmax in fp32 -- the reference computes
// this block in fp16/bf16 too (attention.py:63).
//
// Forward decomposition: workgroup = 16 query rows of one (batch, head); its 4 waves split the
// keys round-robin in steps of 32 and merge (max, sum, O) through LDS at the end.  Per step a wave
// computes S^T = K Q^T (keys on rows, queries on the lane) so the row statistics of a query live
// on lanes l, l+16, l+32, l+48 (two xor-shuffles), exponentiates, and feeds P^T straight back as
// the B operand of O^T += V^T P^T (accumulator register order = operand order, no LDS transpose;
// cdna_hip_programming.md section 3 "An accumulator tile as the next MFMA's operand").
//
// Backward: two kernels, both recompute P from the saved log-sum-exp.
//   dq kernel  : same ownership as forward (16 queries, keys split over waves, LDS merge).
//   dkv kernel : workgroup = 16 keys, waves split the queries; dK and dV need no atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hipad.h"

namespace hipad {

using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ short f2bf(float x) {
  // round-to-nearest-even fp32 -> bf16 (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
  return __builtin_bit_cast(short, (__bf16)x);
}

// 8 consecutive floats -> bf16x8 (two 16-byte loads)
__device__ __forceinline__ bf16x8 load8(const float *p, float mul) {
  const float4 a = reinterpret_cast<const float4 *>(p)[0];
  const float4 b = reinterpret_cast<const float4 *>(p)[1];
  bf16x8 r;
  r[0] = f2bf(a.x * mul); r[1] = f2bf(a.y * mul); r[2] = f2bf(a.z * mul); r[3] = f2bf(a.w * mul);
  r[4] = f2bf(b.x * mul); r[5] = f2bf(b.y * mul); r[6] = f2bf(b.z * mul); r[7] = f2bf(b.w * mul);
  return r;
}

// Transposed operand tiles of the backward kernels.  dK / dV / dQ contract over the 32 rows (queries or keys) of a step,
// so one MFMA operand is a COLUMN of Q, dO or K across 8 of those rows: read from global that is 8 scalar loads per
// operand (64 of them per step of the dK / dV kernel at D = 64, each with its own address arithmetic and conversion).
// Instead a wave stages the step's 32 x D tile once -- coalesced float4 loads, bf16 pairs of two adjacent rows written
// as one dword -- into a private LDS tile stored column-major, and every operand becomes two 8-byte LDS reads.
// Row order, rounding and accumulation order are unchanged: the results are bit-identical to the scalar-load form.
constexpr int kTS = 36;  // bf16 per tile row: 32 rows of the step + pad (rows stay 8-byte aligned)

template <int D>
__device__ __forceinline__ void stage_transposed(short (*tile)[kTS], const float *base, int row0, int nrows, int ld,
                                                 int lane) {
  constexpr int C4 = D / 4;                 // float4 per row
  constexpr int ITEMS = 16 * C4 / 64;       // (row pair, float4) items per lane
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int idx = lane + 64 * i;
    const int rp = idx / C4, c4 = idx - rp * C4;
    const int r0 = min(row0 + 2 * rp, nrows - 1), r1 = min(row0 + 2 * rp + 1, nrows - 1);
    const float4 a = *reinterpret_cast<const float4 *>(base + (size_t)r0 * ld + 4 * c4);
    const float4 b = *reinterpret_cast<const float4 *>(base + (size_t)r1 * ld + 4 * c4);
    const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned lo = (unsigned short)f2bf(av[e]), hi = (unsigned short)f2bf(bv[e]);
      *reinterpret_cast<unsigned *>(&tile[4 * c4 + e][2 * rp]) = lo | (hi << 16);
    }
  }
}

// operand of lane (l15, quad) for column `col`: rows 4 quad .. 4 quad + 3 and 16 + 4 quad .. 16 + 4 quad + 3 of the step
__device__ __forceinline__ bf16x8 tile_operand(const short (*tile)[kTS], int col, int quad) {
  using s4 = __attribute__((ext_vector_type(4))) short;
  const s4 lo = *reinterpret_cast<const s4 *>(&tile[col][4 * quad]);
  const s4 hi = *reinterpret_cast<const s4 *>(&tile[col][16 + 4 * quad]);
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

// counter-based keep decision for attention dropout: same (seed, b, h, q, k) -> same bit in fwd/bwd
__device__ __forceinline__ bool keep_prob(uint32_t seed, int b, int h, int q, int k, uint32_t thresh) {
  uint32_t x = seed ^ (uint32_t)(b * 0x9E3779B1u) ^ (uint32_t)(h * 0x85EBCA77u);
  x ^= (uint32_t)q * 0xC2B2AE3Du;
  x = (x ^ (x >> 15)) * 0x2C1B3C6Du;
  x ^= (uint32_t)k * 0x27D4EB2Fu;
  x = (x ^ (x >> 13)) * 0x297A2D39u;
  x ^= x >> 16;
  return x >= thresh;  // P(keep) = 1 - thresh / 2^32
}

struct AttnArgs {
  int B, H, Nq, Nk, ld;  // ld = H*D (row stride in floats)
  float scale_log2e;     // softmax scale * log2(e)
  float scale;           // softmax scale
  uint32_t seed, drop_thresh;
  float inv_keep;        // 1 / (1 - p_drop)
  const uint32_t *seed_dev;  // optional device-resident step counter added to seed (graph replay)
};

__device__ __forceinline__ uint32_t eff_seed(const AttnArgs &a) {
  return a.seed_dev ? a.seed + *a.seed_dev * 0x9E3779B9u : a.seed;
}

constexpr float kNegInf = -INFINITY;

// =====================================================================================
// forward
// =====================================================================================
template <int D>
__global__ __launch_bounds__(256) void attn_fwd_kernel(float *__restrict__ out, float *__restrict__ lse,
                                                       const float *__restrict__ q, const float *__restrict__ k,
                                                       const float *__restrict__ v, AttnArgs a) {
  constexpr int DC32 = D / 32;  // 32-wide d chunks (QK reduction)
  constexpr int DC16 = D / 16;  // 16-wide d chunks (O rows)
  // V tile of the step transposed in LDS (stage_transposed above): measured in the replayed step 15.2 -> 13.2 us at
  // D = 32 but 20.4 -> 21.6 us at D = 64 (the scalar loads overlap the S tiles' MFMAs; the staged path adds an LDS round
  // trip to every step) -- off, the dK / dV kernel is where it pays
  constexpr bool kStage = false;
  constexpr int kEpiBytes = 4 * 64 * DC16 * 4 * 4, kTileBytes = kStage ? 4 * D * kTS * 2 : 0;
  __shared__ float s_m[4][16], s_l[4][16];
  __shared__ __attribute__((aligned(16))) unsigned char raw[kEpiBytes > kTileBytes ? kEpiBytes : kTileBytes];
  float (*s_o)[64][DC16 * 4] = reinterpret_cast<float (*)[64][DC16 * 4]>(raw);
  short (*t_v)[D][kTS] = reinterpret_cast<short (*)[D][kTS]>(raw);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t seed = a.drop_thresh ? eff_seed(a) : 0u;
  const int l15 = lane & 15, quad = lane >> 4;
  const int q0 = blockIdx.x * 16, h = blockIdx.y, b = blockIdx.z;
  const int qi = min(q0 + l15, a.Nq - 1);
  const float *qp = q + ((size_t)b * a.Nq + qi) * a.ld + h * D;
  const float *kb = k + (size_t)b * a.Nk * a.ld + h * D;
  const float *vb = v + (size_t)b * a.Nk * a.ld + h * D;

  bf16x8 qf[DC32];
#pragma unroll
  for (int c = 0; c < DC32; ++c) qf[c] = load8(qp + 32 * c + 8 * quad, 1.f);  // bf16(q); scale in fp32 after the product

  f32x4 o[DC16];
#pragma unroll
  for (int c = 0; c < DC16; ++c) o[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m = kNegInf, l = 0.f;

  const int nsteps = (a.Nk + 31) / 32;
  for (int step = wv; step < nsteps; step += 4) {
    const int k0 = step * 32;
    if (kStage) {
      __builtin_amdgcn_wave_barrier();
      stage_transposed<D>(t_v[wv], vb, k0, a.Nk, a.ld, lane);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    // S^T tiles: rows = keys k0 + 16t + 4*quad + r, col = query l15
    f32x4 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int kr = min(k0 + 16 * t + l15, a.Nk - 1);
      const float *kp = kb + (size_t)kr * a.ld + 8 * quad;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < DC32; ++c)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(load8(kp + 32 * c, 1.f), qf[c], acc, 0, 0, 0);
      s[t] = acc * a.scale_log2e;
    }
    float mx = kNegInf;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + 16 * t + 4 * quad + r;
        if (key >= a.Nk) s[t][r] = kNegInf;
        mx = fmaxf(mx, s[t][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m, mx);
    const float alpha = (m_new == kNegInf) ? 1.f : exp2f(m - m_new);
    bf16x8 pf;
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = (m_new == kNegInf) ? 0.f : exp2f(s[t][r] - m_new);
        psum += p;
        if (a.drop_thresh) {
          const int key = k0 + 16 * t + 4 * quad + r;
          p = keep_prob(seed, b, h, q0 + l15, key, a.drop_thresh) ? p * a.inv_keep : 0.f;
        }
        pf[4 * t + r] = f2bf(p);
      }
    l = l * alpha + psum;
    m = m_new;
    // O^T[d][query] += V^T[d][key] P^T[key][query]; operand k-slot j <-> key k0 + 16(j>>2) + 4quad + (j&3)
#pragma unroll
    for (int c = 0; c < DC16; ++c) {
      bf16x8 vf;
      if (kStage) {
        vf = tile_operand(t_v[wv], 16 * c + l15, quad);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int key = min(k0 + 16 * (j >> 2) + 4 * quad + (j & 3), a.Nk - 1);
          vf[j] = f2bf(vb[(size_t)key * a.ld + 16 * c + l15]);
        }
      }
      o[c] = o[c] * alpha;
      o[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[c], 0, 0, 0);
    }
  }
  // row sum over the four quads of a query
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  if (kStage) __syncthreads();              // the V tiles and the merge buffer share LDS
  // ---- merge the 4 waves
  if (quad == 0) {
    s_m[wv][l15] = m;
    s_l[wv][l15] = l;
  }
#pragma unroll
  for (int c = 0; c < DC16; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) s_o[wv][lane][4 * c + r] = o[c][r];
  __syncthreads();
  if (wv == 0) {
    float mm = kNegInf;
#pragma unroll
    for (int w = 0; w < 4; ++w) mm = fmaxf(mm, s_m[w][l15]);
    float lt = 0.f, f[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      f[w] = (s_m[w][l15] == kNegInf) ? 0.f : exp2f(s_m[w][l15] - mm);
      lt += s_l[w][l15] * f[w];
    }
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    if (q0 + l15 < a.Nq) {
      float *op = out + ((size_t)b * a.Nq + q0 + l15) * a.ld + h * D;
#pragma unroll
      for (int c = 0; c < DC16; ++c) {
        float4 r4;
        float *rr = reinterpret_cast<float *>(&r4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float acc = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) acc += s_o[w][lane][4 * c + r] * f[w];
          rr[r] = acc * inv;
        }
        reinterpret_cast<float4 *>(op + 16 * c + 4 * quad)[0] = r4;
      }
      if (quad == 0 && lse) lse[((size_t)b * a.H + h) * a.Nq + q0 + l15] = mm + log2f(lt);  // base-2 lse
    }
  }
}

// =====================================================================================
// backward dQ: workgroup = 16 queries of one (b, h); waves split the keys.
//   S^T, P^T as in forward; dP^T = V dO^T (keys x queries); dS^T = P^T o (dP^T - delta) * scale
//   dQ^T[d][query] += K^T[d][key] dS^T[key][query]
// =====================================================================================
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(float *__restrict__ dq, const float *__restrict__ dout,
                                                          const float *__restrict__ q, const float *__restrict__ k,
                                                          const float *__restrict__ v, const float *__restrict__ lse,
                                                          const float *__restrict__ out, float *__restrict__ delta,
                                                          AttnArgs a) {
  constexpr int DC32 = D / 32, DC16 = D / 16;
  // K tile of the step transposed in LDS: 37.0 -> 36.4 us at D = 64, 17.2 -> 24.5 us at D = 32 (with the delta sum
  // folded in at the same time) -- off
  constexpr bool kStage = false;
  constexpr int kEpiBytes = 4 * 64 * DC16 * 4 * 4, kTileBytes = kStage ? 4 * D * kTS * 2 : 0;
  __shared__ __attribute__((aligned(16))) unsigned char raw[kEpiBytes > kTileBytes ? kEpiBytes : kTileBytes];
  float (*s_o)[64][DC16 * 4] = reinterpret_cast<float (*)[64][DC16 * 4]>(raw);   // epilogue: the waves' partial dQ
  short (*t_k)[D][kTS] = reinterpret_cast<short (*)[D][kTS]>(raw);               // main loop: K tile per wave, transposed
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t seed = a.drop_thresh ? eff_seed(a) : 0u;
  const int l15 = lane & 15, quad = lane >> 4;
  const int q0 = blockIdx.x * 16, h = blockIdx.y, b = blockIdx.z;
  const int qi = min(q0 + l15, a.Nq - 1);
  const float *qp = q + ((size_t)b * a.Nq + qi) * a.ld + h * D;
  const float *dop = dout + ((size_t)b * a.Nq + qi) * a.ld + h * D;
  const float *kb = k + (size_t)b * a.Nk * a.ld + h * D;
  const float *vb = v + (size_t)b * a.Nk * a.ld + h * D;
  const float my_lse = lse[((size_t)b * a.H + h) * a.Nq + qi];
  // delta[b, h, q] = sum_d dO * O in fp32, computed here (a lane holds 8 columns per 32-column chunk of its query, the
  // four quads cover the row) and left in `delta` for the dK / dV kernel that follows on the stream: no separate launch
  const float *op = out + ((size_t)b * a.Nq + qi) * a.ld + h * D;
  float my_delta = 0.f;
#pragma unroll
  for (int c = 0; c < DC32; ++c) {
    const float4 d0 = *reinterpret_cast<const float4 *>(dop + 32 * c + 8 * quad);
    const float4 d1 = *reinterpret_cast<const float4 *>(dop + 32 * c + 8 * quad + 4);
    const float4 o0 = *reinterpret_cast<const float4 *>(op + 32 * c + 8 * quad);
    const float4 o1 = *reinterpret_cast<const float4 *>(op + 32 * c + 8 * quad + 4);
    my_delta += (d0.x * o0.x + d0.y * o0.y + d0.z * o0.z + d0.w * o0.w) + (d1.x * o1.x + d1.y * o1.y + d1.z * o1.z + d1.w * o1.w);
  }
  my_delta += __shfl_xor(my_delta, 16);
  my_delta += __shfl_xor(my_delta, 32);
  if (wv == 0 && quad == 0 && q0 + l15 < a.Nq) delta[((size_t)b * a.H + h) * a.Nq + qi] = my_delta;

  bf16x8 qf[DC32], dof[DC32];
#pragma unroll
  for (int c = 0; c < DC32; ++c) {
    qf[c] = load8(qp + 32 * c + 8 * quad, 1.f);
    dof[c] = load8(dop + 32 * c + 8 * quad, 1.f);
  }
  f32x4 acc_dq[DC16];
#pragma unroll
  for (int c = 0; c < DC16; ++c) acc_dq[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nsteps = (a.Nk + 31) / 32;
  for (int step = wv; step < nsteps; step += 4) {
    const int k0 = step * 32;
    if (kStage) {
      __builtin_amdgcn_wave_barrier();      // the previous step's operand reads are done
      stage_transposed<D>(t_k[wv], kb, k0, a.Nk, a.ld, lane);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    bf16x8 dsf;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int kr = min(k0 + 16 * t + l15, a.Nk - 1);
      const float *kp = kb + (size_t)kr * a.ld + 8 * quad;
      const float *vp = vb + (size_t)kr * a.ld + 8 * quad;
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < DC32; ++c) {
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(load8(kp + 32 * c, 1.f), qf[c], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(load8(vp + 32 * c, 1.f), dof[c], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + 16 * t + 4 * quad + r;
        float p = (key < a.Nk) ? exp2f(s[r] * a.scale_log2e - my_lse) : 0.f;
        float dpr = dp[r];
        if (a.drop_thresh) {
          const bool keep = keep_prob(seed, b, h, q0 + l15, key, a.drop_thresh);
          dpr = keep ? dpr * a.inv_keep : 0.f;  // d(dropped P)/dP
        }
        dsf[4 * t + r] = f2bf(p * (dpr - my_delta) * a.scale);
      }
    }
#pragma unroll
    for (int c = 0; c < DC16; ++c) {
      bf16x8 kf;
      if (kStage) {
        kf = tile_operand(t_k[wv], 16 * c + l15, quad);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int key = min(k0 + 16 * (j >> 2) + 4 * quad + (j & 3), a.Nk - 1);
          kf[j] = f2bf(kb[(size_t)key * a.ld + 16 * c + l15]);
        }
      }
      acc_dq[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, dsf, acc_dq[c], 0, 0, 0);
    }
  }
  if (kStage) __syncthreads();              // the tiles and the epilogue buffer share LDS
#pragma unroll
  for (int c = 0; c < DC16; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) s_o[wv][lane][4 * c + r] = acc_dq[c][r];
  __syncthreads();
  if (wv == 0 && q0 + l15 < a.Nq) {
    float *op = dq + ((size_t)b * a.Nq + q0 + l15) * a.ld + h * D;
#pragma unroll
    for (int c = 0; c < DC16; ++c) {
      float4 r4;
      float *rr = reinterpret_cast<float *>(&r4);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        rr[r] = s_o[0][lane][4 * c + r] + s_o[1][lane][4 * c + r] + s_o[2][lane][4 * c + r] + s_o[3][lane][4 * c + r];
      reinterpret_cast<float4 *>(op + 16 * c + 4 * quad)[0] = r4;
    }
  }
}

// =====================================================================================
// backward dK, dV: workgroup = 16 keys of one (b, h); waves split the queries (steps of 32).
//   S tile t: rows = queries q0 + 16t + 4quad + r, col = key l15   (A = Q rows, B = K^T)
//   dV^T[d][key] += dO^T[d][query] Pd[query][key]      (Pd = dropped P)
//   dK^T[d][key] += Q^T [d][query] dS[query][key]
// =====================================================================================
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(float *__restrict__ dk, float *__restrict__ dv,
                                                           const float *__restrict__ dout, const float *__restrict__ q,
                                                           const float *__restrict__ k, const float *__restrict__ v,
                                                           const float *__restrict__ lse,
                                                           const float *__restrict__ delta, AttnArgs a) {
  constexpr int DC32 = D / 32, DC16 = D / 16;
  // Q and dO tiles of the step transposed in LDS: 56.6 -> 51.5 us at D = 64, 20.7 -> 19.7 us at D = 32 (64 scalar loads
  // per step before); a 128-wide head's tiles do not fit: scalar loads there
  constexpr bool kStage = D <= 64;
  constexpr int kEpiBytes = 2 * 4 * 64 * DC16 * 4 * 4, kTileBytes = kStage ? 2 * 4 * D * kTS * 2 : 0;
  __shared__ __attribute__((aligned(16))) unsigned char raw[kEpiBytes > kTileBytes ? kEpiBytes : kTileBytes];
  float (*s_k)[64][DC16 * 4] = reinterpret_cast<float (*)[64][DC16 * 4]>(raw);   // epilogue: the waves' partial dK, dV
  float (*s_v)[64][DC16 * 4] = s_k + 4;
  short (*t_q)[D][kTS] = reinterpret_cast<short (*)[D][kTS]>(raw);               // main loop: Q and dO tiles per wave
  short (*t_do)[D][kTS] = t_q + 4;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t seed = a.drop_thresh ? eff_seed(a) : 0u;
  const int l15 = lane & 15, quad = lane >> 4;
  const int kk0 = blockIdx.x * 16, h = blockIdx.y, b = blockIdx.z;
  const int ki = min(kk0 + l15, a.Nk - 1);
  const bool key_ok = kk0 + l15 < a.Nk;
  const float *kp = k + ((size_t)b * a.Nk + ki) * a.ld + h * D;
  const float *vp = v + ((size_t)b * a.Nk + ki) * a.ld + h * D;
  const float *qb = q + (size_t)b * a.Nq * a.ld + h * D;
  const float *dob = dout + (size_t)b * a.Nq * a.ld + h * D;
  const float *lse_b = lse + ((size_t)b * a.H + h) * a.Nq;
  const float *del_b = delta + ((size_t)b * a.H + h) * a.Nq;

  bf16x8 kf[DC32], vf[DC32];  // B operands: [k = d][col = key]
#pragma unroll
  for (int c = 0; c < DC32; ++c) {
    kf[c] = load8(kp + 32 * c + 8 * quad, 1.f);
    vf[c] = load8(vp + 32 * c + 8 * quad, 1.f);
  }
  f32x4 acc_k[DC16], acc_v[DC16];
#pragma unroll
  for (int c = 0; c < DC16; ++c) {
    acc_k[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc_v[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int nsteps = (a.Nq + 31) / 32;
  for (int step = wv; step < nsteps; step += 4) {
    const int qq0 = step * 32;
    if (kStage) {
      __builtin_amdgcn_wave_barrier();
      stage_transposed<D>(t_q[wv], qb, qq0, a.Nq, a.ld, lane);
      stage_transposed<D>(t_do[wv], dob, qq0, a.Nq, a.ld, lane);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    bf16x8 pf, dsf;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int qr = min(qq0 + 16 * t + l15, a.Nq - 1);
      const float *qp = qb + (size_t)qr * a.ld + 8 * quad;
      const float *dp_ = dob + (size_t)qr * a.ld + 8 * quad;
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < DC32; ++c) {
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(load8(qp + 32 * c, 1.f), kf[c], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(load8(dp_ + 32 * c, 1.f), vf[c], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = qq0 + 16 * t + 4 * quad + r;
        const bool ok = qi < a.Nq && key_ok;
        const int qc = min(qi, a.Nq - 1);
        float p = ok ? exp2f(s[r] * a.scale_log2e - lse_b[qc]) : 0.f;
        float pd = p, dpr = dp[r];
        if (a.drop_thresh) {
          const bool keep = keep_prob(seed, b, h, qi, kk0 + l15, a.drop_thresh);
          pd = keep ? p * a.inv_keep : 0.f;
          dpr = keep ? dpr * a.inv_keep : 0.f;
        }
        pf[4 * t + r] = f2bf(pd);
        dsf[4 * t + r] = f2bf(p * (dpr - del_b[qc]) * a.scale);
      }
    }
#pragma unroll
    for (int c = 0; c < DC16; ++c) {
      bf16x8 dot, qt;  // A operands: [row = d][k = query slot j]
      if (kStage) {
        dot = tile_operand(t_do[wv], 16 * c + l15, quad);
        qt = tile_operand(t_q[wv], 16 * c + l15, quad);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int qi = min(qq0 + 16 * (j >> 2) + 4 * quad + (j & 3), a.Nq - 1);
          dot[j] = f2bf(dob[(size_t)qi * a.ld + 16 * c + l15]);
          qt[j] = f2bf(qb[(size_t)qi * a.ld + 16 * c + l15]);
        }
      }
      acc_v[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pf, acc_v[c], 0, 0, 0);
      acc_k[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsf, acc_k[c], 0, 0, 0);
    }
  }
  if (kStage) __syncthreads();              // the tiles and the epilogue buffers share LDS
#pragma unroll
  for (int c = 0; c < DC16; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s_k[wv][lane][4 * c + r] = acc_k[c][r];
      s_v[wv][lane][4 * c + r] = acc_v[c][r];
    }
  __syncthreads();
  if (wv == 0 && key_ok) {
    float *okp = dk + ((size_t)b * a.Nk + kk0 + l15) * a.ld + h * D;
    float *ovp = dv + ((size_t)b * a.Nk + kk0 + l15) * a.ld + h * D;
#pragma unroll
    for (int c = 0; c < DC16; ++c) {
      float4 rk, rv;
      float *pk = reinterpret_cast<float *>(&rk), *pv = reinterpret_cast<float *>(&rv);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pk[r] = s_k[0][lane][4 * c + r] + s_k[1][lane][4 * c + r] + s_k[2][lane][4 * c + r] + s_k[3][lane][4 * c + r];
        pv[r] = s_v[0][lane][4 * c + r] + s_v[1][lane][4 * c + r] + s_v[2][lane][4 * c + r] + s_v[3][lane][4 * c + r];
      }
      reinterpret_cast<float4 *>(okp + 16 * c + 4 * quad)[0] = rk;
      reinterpret_cast<float4 *>(ovp + 16 * c + 4 * quad)[0] = rv;
    }
  }
}

static AttnArgs make_args(int B, int H, int Nq, int Nk, int D, float scale, float p_drop, uint32_t seed,
                          const uint32_t *seed_dev) {
  AttnArgs a;
  a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.ld = H * D;
  a.scale = scale;
  a.scale_log2e = scale * 1.4426950408889634f;
  a.seed = seed;
  a.seed_dev = seed_dev;
  if (p_drop > 0.f) {
    double t = (double)p_drop * 4294967296.0;
    a.drop_thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
    if (a.drop_thresh == 0) a.drop_thresh = 1;
    a.inv_keep = 1.f / (1.f - p_drop);
  } else {
    a.drop_thresh = 0;
    a.inv_keep = 1.f;
  }
  return a;
}

static int attn_check(int B, int H, int Nq, int Nk, int D, float p_drop) {
  if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return HIPAD_EINVAL;
  if (D != 32 && D != 64 && D != 128) return HIPAD_EINVAL;
  if (!(p_drop >= 0.f && p_drop < 1.f)) return HIPAD_EINVAL;
  if (H > 65535 || B > 65535) return HIPAD_ERANGE;
  return HIPAD_OK;
}

}  // namespace hipad

using namespace hipad;

extern "C" {

int hipad_attention_forward(float *out, float *lse, const float *q, const float *k, const float *v, int B,
                            int H, int Nq, int Nk, int D, float softmax_scale, float p_drop, unsigned seed,
                            const unsigned *seed_dev, hipad_stream_t stream_) {
  int rc = attn_check(B, H, Nq, Nk, D, p_drop);
  if (rc != HIPAD_OK) return rc;
  if (!out || !q || !k || !v) return HIPAD_EINVAL;
  const AttnArgs a = make_args(B, H, Nq, Nk, D, softmax_scale, p_drop, seed, seed_dev);
  const dim3 grid((Nq + 15) / 16, H, B), block(256);
  hipStream_t stream = (hipStream_t)stream_;
  if (D == 32) hipLaunchKernelGGL(attn_fwd_kernel<32>, grid, block, 0, stream, out, lse, q, k, v, a);
  else if (D == 64) hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, block, 0, stream, out, lse, q, k, v, a);
  else hipLaunchKernelGGL(attn_fwd_kernel<128>, grid, block, 0, stream, out, lse, q, k, v, a);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int hipad_attention_backward(float *dq, float *dk, float *dv, float *delta_ws, const float *dout, const float *out,
                             const float *lse, const float *q, const float *k, const float *v, int B, int H, int Nq,
                             int Nk, int D, float softmax_scale, float p_drop, unsigned seed, const unsigned *seed_dev,
                             hipad_stream_t stream_) {
  int rc = attn_check(B, H, Nq, Nk, D, p_drop);
  if (rc != HIPAD_OK) return rc;
  if (!dq || !dk || !dv || !delta_ws || !dout || !out || !lse || !q || !k || !v) return HIPAD_EINVAL;
  const AttnArgs a = make_args(B, H, Nq, Nk, D, softmax_scale, p_drop, seed, seed_dev);
  hipStream_t stream = (hipStream_t)stream_;
  const dim3 gq((Nq + 15) / 16, H, B), gk((Nk + 15) / 16, H, B), block(256);
#define HIPAD_ATTN_BWD(DD)                                                                                   \
  hipLaunchKernelGGL(attn_bwd_dq_kernel<DD>, gq, block, 0, stream, dq, dout, q, k, v, lse, out, delta_ws, a);           \
  hipLaunchKernelGGL(attn_bwd_dkv_kernel<DD>, gk, block, 0, stream, dk, dv, dout, q, k, v, lse, (const float *)delta_ws, a)
  if (D == 32) { HIPAD_ATTN_BWD(32); }
  else if (D == 64) { HIPAD_ATTN_BWD(64); }
  else { HIPAD_ATTN_BWD(128); }
#undef HIPAD_ATTN_BWD
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

}  // extern "C"
