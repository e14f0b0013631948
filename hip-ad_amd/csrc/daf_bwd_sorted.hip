// hip-ad_amd/csrc/daf_bwd_sorted.hip -- backward of the deformable aggregation without the
// atomic scatter.
//
// Reference semantics: deformable_aggregation_cuda.cu:190-262 + :62-126 -- per thread 4 float
// atomicAdds into grad_feat, 1 into grad_weights (32-way contention), 2 into
// grad_sampling_location (1024-way contention).
//
// Why not atomics: a stage-2 plan call scatters ~0.5 GB of fp32 adds; gfx950 retires float
// atomics at ~1.3 TB/s chip-wide whatever the schedule (MI355X_MICROARCH.md "Global float
// atomics"), i.e. >= 0.4 ms per call, 24 calls per frame.  The transposed problem
//     grad_feat[row, :] += sum over taps t hitting row:  coef_t * w[t's pair, level, group(c)]
//                                                                * grad_out[t's anchor, :]
// is a GATHER from grad_out (<= 1 MB per call, L2 resident) once the taps are grouped by
// destination row.  So:
//   1. tap_pass<count>   one thread per (point,camera) pair: histogram of taps per pyramid row
//   2. alloc             every row reserves a contiguous tap segment (block-local scan + one
//                        atomic per 1024 rows; segments need not be in row order)
//   3. tap_pass<place>   same walk, each tap takes a slot in its row's segment (4-byte tap id)
//   4. feat kernel       one wave per batch of 64 row-sorted taps: lanes decode their tap in
//                        parallel (coefficient, weights, anchor), then the wave walks the batch,
//                        accumulating in registers while the row is unchanged and adding the
//                        1 KiB row to grad_feat when it changes: plain read-modify-write when
//                        the row lies wholly inside the batch (the usual case), fp32 atomics
//                        only for rows that straddle batches.
//   5. lw kernel         grad_weights / grad_location: one wave per (anchor, chunk of points)
//                        as in the forward, float4 per lane, 8-lane / 64-lane butterflies,
//                        one plain store per result (deterministic).
// Steps 1-4 touch only index data + grad_out; step 5 re-gathers the features.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "hipad.h"
#include "daf_common.h"

namespace hipad {

// ------------------------------------------------------------------------------------------
// 1 + 3: tap enumeration.  tap id = ((pair * L + s) << 2) | corner.
//
// A per-row global atomic per tap serialises on hot rows (a coarse 8x22 map takes a quarter of
// all taps on ~1000 addresses: measured 0.4 ms per pass for a stage-2 plan call).  So the
// workgroup (1024 consecutive pairs ~ a couple of anchors, spatially clustered) privatises the
// coarse maps in LDS: phase 1 counts its taps per row in an LDS table, phase 2 issues ONE global
// atomic per touched row (count pass: add the count; place pass: reserve `count` slots and keep
// the returned base in the table), phase 3 (place only) re-walks the taps and draws slots from
// the LDS cursors.  Maps too large for the table (the fine levels, few taps per row) go straight
// to global atomics.  Which maps are "coarse" is decided on device from spatial_shape: coarsest
// level first while every camera's map of that level still fits the table.
// ------------------------------------------------------------------------------------------
// Multi-call form.  A decoder frame makes 24 aggregation calls whose feature gradients all land in ONE buffer, and the
// pipeline above is pure fixed cost at the size of one call (five launches of a few dozen to a few hundred workgroups,
// 2.4 ms per frame for 24 x 5 launches).  So the kernels take a TABLE of calls (location / weight / grad_out pointers and
// the (anchors, points) of each; cameras, levels, the pyramid and grad_feat are shared): the (point,camera) pairs of all
// calls form one global index space, a tap id is ((global pair * L + level) << 2) | corner, and count / alloc / place /
// accumulate run ONCE per frame over every tap of the frame.  A row that several calls touch is read-modify-written
// once instead of once per call.  The single-call entry (hipad_daf_backward) is the same code with a one-entry table.
struct MultiArgs {
  const float *loc[HIPAD_DAF_MAX_CALLS];
  const float *wts[HIPAD_DAF_MAX_CALLS];
  const float *gout[HIPAD_DAF_MAX_CALLS];
  int A[HIPAD_DAF_MAX_CALLS];
  int P[HIPAD_DAF_MAX_CALLS];
  int pbase[HIPAD_DAF_MAX_CALLS + 1];  // first global pair of each call; pbase[ncalls] = all pairs
  int ncalls;
  int pad;
};
static_assert(sizeof(MultiArgs) % 4 == 0 && sizeof(MultiArgs) <= 3072, "kernel-argument budget");

// the table, copied once per workgroup from the kernel-argument segment into LDS (runtime indexing of a by-value
// argument would otherwise go through scratch)
__device__ __forceinline__ void stage_calls(MultiArgs &lds, const MultiArgs &a) {
  const unsigned *src = reinterpret_cast<const unsigned *>(&a);
  unsigned *dst = reinterpret_cast<unsigned *>(&lds);
  for (unsigned i = threadIdx.x; i < sizeof(MultiArgs) / 4; i += blockDim.x) dst[i] = src[i];
}

// call that owns global pair gp (pbase ascending, pbase[0] == 0, gp < pbase[ncalls])
__device__ __forceinline__ int find_call(const int *pbase, int ncalls, int gp) {
  int lo = 0, hi = ncalls - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (pbase[mid] <= gp) lo = mid; else hi = mid - 1;
  }
  return lo;
}

constexpr int kTapBlock = 1024;
constexpr int kLdsRows = 24576;  // 96 KiB of counters: all four levels of the six 704x256 cameras (22 440 rows).  With the
                                 // finest level left to global atomics the plan call's taps -- clustered on a few hundred
                                 // rows of the front camera -- serialise in L2: 105 + 83 us per call instead of 42 + 37
constexpr int kMaxMaps = 64;     // cams * L handled by the LDS path

template <bool PLACE>
__global__ __launch_bounds__(kTapBlock) void daf_tap_pass_kernel(
    int *__restrict__ counter /* cnt (count pass) or cursor (place pass) */, int *__restrict__ taps,
    const int *__restrict__ ss, const int *__restrict__ start, const MultiArgs margs,
    int npair, int cams, int num_feat, int L, int cap /* slots in taps[] */,
    int nch /* chunks of kTapBlock consecutive pairs per workgroup */) {
  __shared__ int tab[kLdsRows];
  __shared__ MultiArgs m;
  __shared__ int map_base[kMaxMaps];  // LDS slot of the map's first row, or -1
  // geometry of every (camera, level) map, fetched once by nmaps threads in parallel: thread 0's slot assignment,
  // the per-thread tap walks and the per-row flush all read it from LDS instead of chasing global loads
  __shared__ int geo_h[kMaxMaps], geo_w[kMaxMaps], geo_s[kMaxMaps];
  __shared__ int cm_slot[kMaxMaps + 1], cm_map[kMaxMaps];  // LDS-resident maps in slot order: first slot, map id
  __shared__ int used_s, ncoarse_s;
  const int tid = threadIdx.x;
  const int nmaps = cams * L;
  const bool lds_geo = nmaps <= kMaxMaps;
  // A workgroup walks the pairs of ONE camera (blockIdx.y): its table then only has to hold that camera's maps, so
  // at 704x256 all four levels are privatised (14 960 rows; the finest level of six cameras together is 67 584 and
  // used to go tap by tap through global atomics: a quarter of all taps)
  const int cam_b = blockIdx.y;

  stage_calls(m, margs);
  if (lds_geo && tid < nmaps) {
    geo_h[tid] = ss[2 * tid];
    geo_w[tid] = ss[2 * tid + 1];
    geo_s[tid] = start[tid];
  }
  if (tid < kMaxMaps) map_base[tid] = -1;
  __syncthreads();
  // sample of the block's first pair (the LDS table privatises the rows of that sample)
  const int q0 = blockIdx.x * kTapBlock * nch;      // first (anchor, point) of the block; pair = point * cams + camera
  const int gp0 = q0 * cams + cam_b;
  const int k0 = find_call(m.pbase, m.ncalls, min(gp0, npair - 1));
  const int b0 = (gp0 - m.pbase[k0]) / (m.P[k0] * cams * m.A[k0]);
  if (tid == 0) {
    int used = 0, nc = 0;
    if (lds_geo) {
      for (int s = L - 1; s >= 0; --s) {    // coarsest level first, while the camera's map still fits
        const int c = cam_b;
        const int need = geo_h[c * L + s] * geo_w[c * L + s];
        if (used + need > kLdsRows) break;
        map_base[c * L + s] = used;
        cm_slot[nc] = used;
        cm_map[nc++] = c * L + s;
        used += need;
      }
    }
    cm_slot[nc] = used;
    used_s = used;
    ncoarse_s = nc;
  }
  __syncthreads();
  const int used = used_s;
  for (int i = tid; i < used; i += kTapBlock) tab[i] = 0;
  __syncthreads();

  // the thread's pair of chunk c: location, camera, sample, kept
  struct PairRef {
    float2 l;
    int pair, cam, b;
    bool kept;
  };
  auto resolve = [&](int c) {
    PairRef r;
    r.pair = (q0 + c * kTapBlock + tid) * cams + cam_b;
    r.l = make_float2(-1.f, -1.f);
    const int kc = find_call(m.pbase, m.ncalls, min(r.pair, npair - 1));
    const int lpair = r.pair - m.pbase[kc];  // pair index inside its call
    if (r.pair < npair) r.l = reinterpret_cast<const float2 *>(m.loc[kc])[lpair];
    r.kept = r.pair < npair && loc_kept(r.l.x, r.l.y);
    r.cam = cam_b;                            // == lpair % cams: every call's pair count is a multiple of cams
    r.b = lpair / (m.P[kc] * cams * m.A[kc]);
    return r;
  };

  // ---- phase 1: count into LDS (coarse maps of sample b0) or straight to global
  for (int c = 0; c < nch; ++c) {
    const PairRef r = resolve(c);
    if (!r.kept) continue;
    for (int s = 0; s < L; ++s) {
      const int cs = r.cam * L + s;
      const int H = lds_geo ? geo_h[cs] : ss[2 * cs], W = lds_geo ? geo_w[cs] : ss[2 * cs + 1];
      const Taps t = make_taps(r.l.y, r.l.x, H, W);
      const int lbase = (lds_geo && r.b == b0) ? map_base[cs] : -1;
      const int gbase = r.b * num_feat + (lds_geo ? geo_s[cs] : start[cs]);
      const int id0 = (r.pair * L + s) << 2;
#pragma unroll
      for (int corner = 0; corner < 4; ++corner) {
        const bool in_h = (corner >> 1) ? t.in_h1 : t.in_h0;
        const bool in_w = (corner & 1) ? t.in_w1 : t.in_w0;
        if (in_h && in_w) {
          const int rel = (t.h_low + (corner >> 1)) * W + (t.w_low + (corner & 1));
          if (lbase >= 0) {
            atomicAdd(&tab[lbase + rel], 1);
          } else {
            const int pos = atomicAdd(counter + gbase + rel, 1);
            // pos < cap always holds for a consistent workspace; the check keeps a corrupted one (e.g. two
            // launches sharing it from different streams) from turning into a wild store
            if (PLACE && (unsigned)pos < (unsigned)cap) taps[pos] = id0 | corner;
          }
        }
      }
    }
  }
  __syncthreads();
  // ---- phase 2: one global atomic per touched LDS row.  Four slots per thread and trip, so that (place pass) four
  // atomics are in flight before the first returned base is needed; the map a slot belongs to is found by a cursor
  // that only moves forward (a thread's slots ascend), i.e. ~nmaps LDS reads per thread in total.
  {
    int k = 0;
    for (int i0 = tid; i0 < used; i0 += 4 * kTapBlock) {
      int c[4], addr[4], base[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * kTapBlock;
        c[u] = i < used ? tab[i] : 0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * kTapBlock;
        addr[u] = 0;
        if (c[u] > 0) {
          while (cm_slot[k + 1] <= i) ++k;  // i < used == cm_slot[ncoarse]: stops inside the table
          addr[u] = b0 * num_feat + geo_s[cm_map[k]] + (i - cm_slot[k]);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        base[u] = 0;
        if (c[u] > 0) base[u] = atomicAdd(counter + addr[u], c[u]);
      }
      if (PLACE) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (c[u] > 0) tab[i0 + u * kTapBlock] = base[u];
      }
    }
  }
  if (!PLACE) return;
  __syncthreads();
  // ---- phase 3: draw slots from the LDS cursors
  if (!lds_geo) return;
  for (int c = 0; c < nch; ++c) {
    const PairRef r = resolve(c);
    if (!(r.kept && r.b == b0)) continue;
    for (int s = 0; s < L; ++s) {
      const int cs = r.cam * L + s;
      const int lbase = map_base[cs];
      if (lbase < 0) continue;
      const int H = geo_h[cs], W = geo_w[cs];
      const Taps t = make_taps(r.l.y, r.l.x, H, W);
      const int id0 = (r.pair * L + s) << 2;
#pragma unroll
      for (int corner = 0; corner < 4; ++corner) {
        const bool in_h = (corner >> 1) ? t.in_h1 : t.in_h0;
        const bool in_w = (corner & 1) ? t.in_w1 : t.in_w0;
        if (in_h && in_w) {
          const int rel = (t.h_low + (corner >> 1)) * W + (t.w_low + (corner & 1));
          const int pos = atomicAdd(&tab[lbase + rel], 1);
          if ((unsigned)pos < (unsigned)cap) taps[pos] = id0 | corner;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// 2: segment allocation.  Rows only need CONTIGUOUS tap segments, not segments in row order,
// so no global prefix sum: each workgroup scans its 1024 rows locally and reserves its span
// with ONE atomicAdd on the running total (cnt[R]).  offs[r] = cursor[r] = segment start.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void daf_alloc_kernel(int *__restrict__ cnt,
                                                         int *__restrict__ offs,
                                                         int *__restrict__ cursor, int R) {
  __shared__ int wsum[16];
  __shared__ int base_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = blockIdx.x * 1024 + tid;
  const int c = r < R ? cnt[r] : 0;
  // inclusive scan inside the wave
  int x = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(x, d);
    if (lane >= d) x += y;
  }
  if (lane == 63) wsum[wv] = x;
  __syncthreads();
  if (wv == 0) {
    int v = lane < 16 ? wsum[lane] : 0;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      const int y = __shfl_up(v, d);
      if (lane >= d) v += y;
    }
    if (lane < 16) wsum[lane] = v;  // inclusive over waves
    if (lane == 15) base_s = v > 0 ? atomicAdd(cnt + R, v) : 0;  // cnt[R] = running total
  }
  __syncthreads();
  if (r < R) {
    const int o = base_s + (wv ? wsum[wv - 1] : 0) + (x - c);
    offs[r] = o;
    cursor[r] = o;
  }
}

// ------------------------------------------------------------------------------------------
// 4: grad_feat from row-sorted taps.  C == 256, G == 8.  Persistent grid-stride over batches.
// ------------------------------------------------------------------------------------------
// consecutive 64-tap batches one wave walks with its row sum carried along (hipad_daf_set_feat_run overrides)
static int g_feat_run = 4;
void daf_set_feat_run(int batches) { g_feat_run = batches > 0 && batches <= 64 ? batches : 4; }
// most workgroups the accumulation pass is launched with (waves take runs round-robin).  Frame pass on MI355X
// (tools/sweep_tap_chunks.py blocks=...): 1024 545 us, 2048 514, 4096 499, 8192 497, 16384 498 -- with about one run
// per wave nobody waits for a wave that drew one run more
static int g_feat_blocks = 8192;
void daf_set_feat_blocks(int blocks) { g_feat_blocks = blocks >= 64 && blocks <= 65536 ? blocks : 8192; }

__global__ __launch_bounds__(256) void daf_bwd_feat_kernel(
    float *__restrict__ gfeat, const int *__restrict__ taps, const int *__restrict__ offs,
    const int *__restrict__ ends /* cursor after placement */, const int *__restrict__ total_p,
    const MultiArgs margs, const int *__restrict__ ss,
    const int *__restrict__ start, int R, int cams, int num_feat, int L, int npairL, int kFeatRun) {
  __shared__ float wc_s[4][kWave][8];
  __shared__ int htab[4][256];  // per wave: lowest lane of every hash bucket of (row, grad_out row) keys
  __shared__ float tr_s[4][256];  // per wave: one row's partial sums on their way to contiguous atomics
  __shared__ MultiArgs m;
  const int lane = threadIdx.x & (kWave - 1);
  const int wv = threadIdx.x >> 6;
  stage_calls(m, margs);
  __syncthreads();
  const int total = min(max(*total_p, 0), npairL * 4);  // never past the tap buffer, whatever the counter says
  const int nrun = (total + kFeatRun * kWave - 1) / (kFeatRun * kWave);
  const int nwaves = gridDim.x * 4;
  const int g = lane >> 3;  // group of this lane's 4 channels
  float4 *gfeat4 = reinterpret_cast<float4 *>(gfeat);

  // A row whose taps all lie inside the wave's run is written with one plain read-modify-write; the (at most two) rows
  // a run shares with its neighbours are added with atomics.  The atomics execute at the memory side (the neighbour's
  // workgroup sits on another XCD), so they are issued on CONTIGUOUS addresses: the lane's four channels go through
  // LDS and every atomic instruction of the wave covers 256 consecutive bytes.  With one lane adding floats 16 bytes
  // apart the write counter (profiles/r03_daf_pmc_traffic.json) showed 584 MB per frame for 51 MB of touched rows.
  auto flush = [&](int r, int ex, const float4 &o, const float4 &a4) {
    if (r < 0) return;
    if (ex) {
      gfeat4[(size_t)r * 64 + lane] = make_float4(o.x + a4.x, o.y + a4.y, o.z + a4.z, o.w + a4.w);
    } else {
      reinterpret_cast<float4 *>(&tr_s[wv][0])[lane] = a4;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float *pf = gfeat + (size_t)r * 256 + lane;
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(pf + k * kWave, tr_s[wv][k * kWave + lane]);
      __builtin_amdgcn_wave_barrier();
    }
  };

  for (int run = uni(blockIdx.x * 4 + wv); run < nrun; run += nwaves) {
   const int r0 = run * (kFeatRun * kWave);
   const int r1 = min(total, r0 + kFeatRun * kWave);
   int cur = -1, cur_excl = 0;
   float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
   float4 old = make_float4(0.f, 0.f, 0.f, 0.f);
   for (int t0 = r0; t0 < r1; t0 += kWave) {
    const int n = min(kWave, r1 - t0);
    // ---- parallel decode: lane <-> tap
    int row = -1, excl = 0;
    // grad_out row of the tap's anchor (a 64-bit address in two lanes' registers: the calls' grad_out tensors are
    // separate allocations); skipped slots keep the first call's first row, a legal address that is never used
    unsigned long long grow = (unsigned long long)m.gout[0];
    float wq[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) wq[k] = 0.f;
    const int tap_id = lane < n ? taps[t0 + lane] : -1;
    // a slot that is not a tap of THIS launch (never expected: count and place walk the same pairs)
    // must not be used as an index: it is skipped instead of trusted
    if (lane < n && tap_id >= 0 && (tap_id >> 2) < npairL) {
      const int id = tap_id;
      const int corner = id & 3;
      const int q = id >> 2;
      const int gpair = q / L, s = q - gpair * L;
      const int kc = find_call(m.pbase, m.ncalls, gpair);
      const int pair = gpair - m.pbase[kc];
      const int cam = pair % cams;
      const int anchor = pair / (m.P[kc] * cams);
      const int b = anchor / m.A[kc];
      grow = (unsigned long long)(m.gout[kc] + (size_t)anchor * 256);
      const float2 l = reinterpret_cast<const float2 *>(m.loc[kc])[pair];
      const int cs = cam * L + s;
      const int H = ss[2 * cs], W = ss[2 * cs + 1];
      const Taps t = make_taps(l.y, l.x, H, W);
      const float ch = (corner >> 1) ? t.lh : t.hh;
      const float cw = (corner & 1) ? t.lw : t.hw;
      const float coef = ch * cw;
      row = b * num_feat + start[cs] + (t.h_low + (corner >> 1)) * W + (t.w_low + (corner & 1));
      const float4 *w4 = reinterpret_cast<const float4 *>(m.wts[kc]) + ((size_t)pair * L + s) * 2;
      const float4 wa = w4[0], wb = w4[1];
      wq[0] = coef * wa.x; wq[1] = coef * wa.y; wq[2] = coef * wa.z; wq[3] = coef * wa.w;
      wq[4] = coef * wb.x; wq[5] = coef * wb.y; wq[6] = coef * wb.z; wq[7] = coef * wb.w;
      excl = (offs[row] >= r0 && ends[row] <= r1) ? 1 : 0;
    }
    __builtin_amdgcn_wave_barrier();
    reinterpret_cast<float4 *>(&wc_s[wv][lane][0])[0] = make_float4(wq[0], wq[1], wq[2], wq[3]);
    reinterpret_cast<float4 *>(&wc_s[wv][lane][0])[1] = make_float4(wq[4], wq[5], wq[6], wq[7]);
    reinterpret_cast<int4 *>(&htab[wv][0])[lane] = make_int4(kWave, kWave, kWave, kWave);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- merge the taps of the batch that hit the same row FROM the same grad_out row (the points of one anchor
    // crowd into few pixels of the coarse levels: 2-4 taps per such pair on a stage-2 frame): their coefficients add,
    // and ONE 1 KiB grad_out load serves them all -- the pass is bound by exactly those loads.  Leader = the lowest lane
    // of the key's hash bucket; a bucket shared by different keys merges nothing (each lane stays its own leader).
    const int grow_lo = (int)(unsigned)(grow & 0xffffffffull), grow_hi = (int)(unsigned)(grow >> 32);
    const unsigned hb = (((unsigned)row * 0x9E3779B1u) ^ ((unsigned)grow_lo * 0x85EBCA6Bu)) >> 24;
    if (row >= 0) atomicMin(&htab[wv][hb], lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int cand = row >= 0 ? htab[wv][hb] : lane;
    const bool same = __shfl(row, cand) == row && __shfl(grow_lo, cand) == grow_lo && __shfl(grow_hi, cand) == grow_hi;
    const int leader = (row >= 0 && same) ? cand : lane;
    if (leader != lane) {
#pragma unroll
      for (int k = 0; k < 8; ++k) atomicAdd(&wc_s[wv][leader][k], wq[k]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    unsigned long long live = __ballot(row >= 0 && leader == lane);

    // ---- walk the leaders (rows ascend with the lane index: a merged tap belongs to its leader's row)
    // kAhead grad_out rows in flight: with one load per iteration behind the row-change branch every tap paid a full
    // memory latency.  The row's present value is fetched when the row starts and wanted only when it ends.
    // Measured on the frame pass (tools/sweep_tap_chunks.py): 4 ahead 830 us, 8 ahead 870 (the last trip of a batch
    // loads rows nobody uses), 16 ahead 860; a wave-uniform `if` around each load 2 000 us (a wait per load).
#ifndef HIPAD_FEAT_AHEAD
#define HIPAD_FEAT_AHEAD 4
#endif
    constexpr int kAhead = HIPAD_FEAT_AHEAD;
    while (live) {
      int idx[kAhead];
      unsigned long long mm = live;
#pragma unroll
      for (int k = 0; k < kAhead; ++k) {
        idx[k] = mm ? (int)__builtin_ctzll(mm) : -1;
        mm = mm ? (mm & (mm - 1)) : 0ull;
      }
      float4 gq[kAhead];
      float wq8[kAhead];
#pragma unroll
      for (int k = 0; k < kAhead; ++k) {
        const int tt = idx[k] >= 0 ? idx[k] : idx[0];
        const unsigned long long ga = ((unsigned long long)(unsigned)rl_i(grow_hi, tt) << 32) | (unsigned)rl_i(grow_lo, tt);
        gq[k] = reinterpret_cast<const float4 *>(ga)[lane];
        wq8[k] = wc_s[wv][tt][g];
      }
#pragma unroll
      for (int k = 0; k < kAhead; ++k) {
        if (idx[k] >= 0) {
          const int rt = rl_i(row, idx[k]);
          if (rt != cur) {
            flush(cur, cur_excl, old, acc);
            cur = rt;
            cur_excl = rl_i(excl, idx[k]);
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cur_excl) old = gfeat4[(size_t)cur * 64 + lane];
          }
          const float wc = wq8[k];
          acc.x += wc * gq[k].x; acc.y += wc * gq[k].y; acc.z += wc * gq[k].z; acc.w += wc * gq[k].w;
        }
      }
      live = mm;
    }
    __builtin_amdgcn_wave_barrier();  // wc_s[wv] / htab[wv] are rewritten by the next batch
   }
   flush(cur, cur_excl, old, acc);
  }
}

// ------------------------------------------------------------------------------------------
// 5: grad_weights + grad_location.  C == 256, G == 8, one wave per item, float4 per lane.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float group8_sum(float v) { return oct_sum(v); }

// GEO: the (height, width, first row) of every (camera, level) sits in lanes 0..cams*L-1 and is read back with
// v_readlane, so the pair loop has no scalar memory load between a pair's coordinates and its 16 row loads.  The loop
// body is branch-free (null gw / gloc only skip the final stores): with uniform branches between the levels the
// compiler kept each level's loads behind the previous level's reduction, four memory latencies per pair instead of one.
template <int LT, bool OVERWRITE, bool GEO, typename FT>
__global__ __launch_bounds__(256) void daf_bwd_lw_kernel(
    const FT *__restrict__ feat, const int *__restrict__ ss, const int *__restrict__ start,
    const float *__restrict__ loc, const float *__restrict__ wts, const float *__restrict__ gout,
    float *__restrict__ gloc, float *__restrict__ gw, int n_items, int nchunks, int ppc, int cams,
    int num_feat, int L_rt, int A, int P) {
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

  const float4 go = reinterpret_cast<const float4 *>(gout + (size_t)it.anchor * 256)[lane];
  const int g = lane >> 3;
  const float *wbase = wts + (size_t)it.pair0 * L * G + g;
  const size_t frow0 = (size_t)it.b * num_feat;
  int geoH = 1, geoW = 1, geoS = 0;
  if (GEO && lane < cams * L) {
    geoH = ss[2 * lane];
    geoW = ss[2 * lane + 1];
    geoS = start[lane];
  }
  // land the loads above here (an empty asm that consumes the registers): left pending, the compiler's wait for them
  // sits inside the pair loop, where it is also a wait for the previous pair's stores
  asm volatile("" : "+v"(geoH), "+v"(geoW), "+v"(geoS));

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
      float keep = 0.f;  // lane (g, s = lane & 7) keeps the reduced grad_w of level s
      // the pair's sampling weights first: issued after the row loads, the wait for the last of them is a wait for
      // everything outstanding, this pair's grad_w store included
      float awl[LT ? LT : 1];
#pragma unroll
      for (int s = 0; s < LT; ++s) awl[s] = wbase[wofs + s * G];
#pragma unroll
      for (int s = 0; s < L; ++s) {
        const int cs = cam * L + s;
        const int H = GEO ? rl_i(geoH, cs) : ss[2 * cs], W = GEO ? rl_i(geoW, cs) : ss[2 * cs + 1];
        const Taps t = make_taps(loc_h, loc_w, H, W);
        const int h0 = max(t.h_low, 0), h1 = min(t.h_low + 1, H - 1);
        const int w0 = max(t.w_low, 0), w1 = min(t.w_low + 1, W - 1);
        const size_t base = frow0 + (size_t)(GEO ? rl_i(geoS, cs) : start[cs]);
        const float4 v1 = sel4(t.in_h0 && t.in_w0, load_row4<FT>(feat, base + (size_t)uni(h0 * W + w0), lane));
        const float4 v2 = sel4(t.in_h0 && t.in_w1, load_row4<FT>(feat, base + (size_t)uni(h0 * W + w1), lane));
        const float4 v3 = sel4(t.in_h1 && t.in_w0, load_row4<FT>(feat, base + (size_t)uni(h1 * W + w0), lane));
        const float4 v4 = sel4(t.in_h1 && t.in_w1, load_row4<FT>(feat, base + (size_t)uni(h1 * W + w1), lane));
        const float aw = LT ? awl[LT ? s : 0] : wbase[wofs + s * G];
        const float w1c = t.hh * t.hw, w2c = t.hh * t.lw, w3c = t.lh * t.hw, w4c = t.lh * t.lw;
        // value, d/dh, d/dw per channel (cu:92-118), dotted with grad_out
        float dot_val, dot_h, dot_w;
        {
          const float vx = w1c * v1.x + w2c * v2.x + w3c * v3.x + w4c * v4.x;
          const float vy = w1c * v1.y + w2c * v2.y + w3c * v3.y + w4c * v4.y;
          const float vz = w1c * v1.z + w2c * v2.z + w3c * v3.z + w4c * v4.z;
          const float vw = w1c * v1.w + w2c * v2.w + w3c * v3.w + w4c * v4.w;
          dot_val = go.x * vx + go.y * vy + go.z * vz + go.w * vw;
          const float hx = t.hw * (v3.x - v1.x) + t.lw * (v4.x - v2.x);
          const float hy = t.hw * (v3.y - v1.y) + t.lw * (v4.y - v2.y);
          const float hz = t.hw * (v3.z - v1.z) + t.lw * (v4.z - v2.z);
          const float hq = t.hw * (v3.w - v1.w) + t.lw * (v4.w - v2.w);
          dot_h = go.x * hx + go.y * hy + go.z * hz + go.w * hq;
          const float wx = t.hh * (v2.x - v1.x) + t.lh * (v4.x - v3.x);
          const float wy = t.hh * (v2.y - v1.y) + t.lh * (v4.y - v3.y);
          const float wz = t.hh * (v2.z - v1.z) + t.lh * (v4.z - v3.z);
          const float wq = t.hh * (v2.w - v1.w) + t.lh * (v4.w - v3.w);
          dot_w = go.x * wx + go.y * wy + go.z * wz + go.w * wq;
        }
        gl_w += (float)W * aw * dot_w;  // cu:124 with top = grad_out * weight
        gl_h += (float)H * aw * dot_h;  // cu:125
        const float gs = group8_sum(dot_val);  // cu:122 summed over the group's 32 channels
        keep = ((lane & 7) == s) ? gs : keep;
      }
      if (gw) {
        if (L <= 8) {
          if ((lane & 7) < L) {
            float *d = gw + (size_t)it.pair0 * L * G + wofs + (lane & 7) * G + g;
            *d = OVERWRITE ? keep : (*d + keep);
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
            const float2 o = *d;
            *d = make_float2(o.x + sw, o.y + sh);
          }
        }
      }
    }
  }
}

// ----------------------------------------------------------------------------- host side
static int g_tap_chunks = 0;  // 0 = automatic
void daf_set_tap_chunks(int n) { g_tap_chunks = n > 0 ? n : 0; }

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct SortedWs {
  int *cnt, *offs, *cursor, *taps;
  size_t bytes;
};

// R pyramid rows (all samples), npair (point,camera) pairs over all calls
static SortedWs carve(size_t R, size_t npair, int L, void *base) {
  const size_t tmax = npair * L * 4;
  char *p = (char *)base;
  SortedWs w;
  size_t o = 0;
  w.cnt = (int *)(p + o); o += align256((R + 1) * 4);
  w.offs = (int *)(p + o); o += align256((R + 1) * 4);
  w.cursor = (int *)(p + o); o += align256((R + 1) * 4);
  w.taps = (int *)(p + o); o += align256(tmax * 4);
  w.bytes = o;
  return w;
}

static bool sorted_budget_ok(long long bs, long long num_feat, long long npair, int L) {
  return npair * L * 4 < (1ll << 31) && bs * num_feat < (1ll << 30);
}

bool daf_bwd_sorted_supported(const DafDims &d) {
  if (d.C != 256 || d.G != 8 || d.L > 8) return false;
  return sorted_budget_ok(d.bs, d.num_feat, (long long)d.bs * d.A * d.P * d.cams, d.L);
}

size_t daf_bwd_sorted_workspace(const DafDims &d) {
  if (!daf_bwd_sorted_supported(d)) return 0;
  return carve((size_t)d.bs * d.num_feat, (size_t)d.bs * d.A * d.P * d.cams, d.L, nullptr).bytes;
}

// the pipeline over a table of calls (m.pbase filled in by the caller)
static int run_sorted(const MultiArgs &m, const int *ss, const int *start, float *gfeat, int bs, int cams, int num_feat,
                      int L, void *workspace, size_t workspace_bytes, hipStream_t stream) {
  const int npair = m.pbase[m.ncalls];
  const int R = bs * num_feat;
  const SortedWs w = carve((size_t)R, (size_t)npair, L, workspace);
  if (!workspace || workspace_bytes < w.bytes) return HIPAD_EWORKSPACE;
  if (fill_zero(w.cnt, (size_t)(R + 1) * 4, stream) != HIPAD_OK) return HIPAD_ELAUNCH;
  // Work split of the two tap passes: blockIdx.y = camera, blockIdx.x = a run of `nch` chunks of 1024 consecutive
  // (anchor, point) indices.  Every workgroup clears and flushes its LDS table once whatever it walks, and pays one global
  // atomic per pyramid row it touches: fewer, longer workgroups mean fewer of those (a frame: 1.5 M at 252 workgroups),
  // more workgroups mean a shorter walk -- about three per CU is the default, hipad_daf_set_tap_chunks() overrides it.
  const int npoint = npair / cams;
  int nch = g_tap_chunks;
  if (nch <= 0) {
    const long long per_cam = 768 / cams > 0 ? 768 / cams : 1;   // ~3 workgroups per CU (measured: tools/sweep_tap_chunks.py)
    nch = (int)(((long long)npoint + (long long)kTapBlock * per_cam - 1) / ((long long)kTapBlock * per_cam));
  }
  nch = nch < 1 ? 1 : (nch > 64 ? 64 : nch);
  const dim3 pg((npoint + kTapBlock * nch - 1) / (kTapBlock * nch), cams), pb(kTapBlock);
  const int cap = npair * L * 4;
  hipLaunchKernelGGL(daf_tap_pass_kernel<false>, pg, pb, 0, stream, w.cnt, (int *)nullptr, ss, start, m, npair, cams,
                     num_feat, L, cap, nch);
  hipLaunchKernelGGL(daf_alloc_kernel, dim3((R + 1023) / 1024), dim3(1024), 0, stream, w.cnt, w.offs,
                     w.cursor, R);
  hipLaunchKernelGGL(daf_tap_pass_kernel<true>, pg, pb, 0, stream, w.cursor, w.taps, ss, start, m, npair, cams,
                     num_feat, L, cap, nch);
  // grid: up to g_feat_blocks workgroups x 4 waves, each wave walks runs of g_feat_run batches of 64 taps
  const long long tmax = (long long)npair * L * 4;
  const int kFeatRun = g_feat_run;
  long long nb = (tmax + 256 * kFeatRun - 1) / (256 * kFeatRun);
  if (nb > g_feat_blocks) nb = g_feat_blocks;
  hipLaunchKernelGGL(daf_bwd_feat_kernel, dim3((unsigned)nb), dim3(256), 0, stream, gfeat, (const int *)w.taps,
                     (const int *)w.offs, (const int *)w.cursor, (const int *)(w.cnt + R), m, ss, start, R, cams, num_feat,
                     L, npair * L, kFeatRun);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

int daf_bwd_sorted_feat(const float *, const int *ss, const int *start, const float *loc,
                        const float *wts, const float *gout, float *gfeat, const DafDims &d,
                        void *workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!daf_bwd_sorted_supported(d)) return HIPAD_EINVAL;
  MultiArgs m;
  memset(&m, 0, sizeof(m));
  m.loc[0] = loc; m.wts[0] = wts; m.gout[0] = gout;
  m.A[0] = d.A; m.P[0] = d.P;
  m.pbase[0] = 0; m.pbase[1] = d.bs * d.A * d.P * d.cams;
  m.ncalls = 1;
  return run_sorted(m, ss, start, gfeat, d.bs, d.cams, d.num_feat, d.L, workspace, workspace_bytes, stream);
}

static int multi_table(MultiArgs &m, const hipad_daf_call *calls, int ncalls, int bs, int cams, int num_feat, int C,
                       int L, int G) {
  if (!calls || ncalls <= 0 || ncalls > HIPAD_DAF_MAX_CALLS) return HIPAD_EINVAL;
  if (bs <= 0 || cams <= 0 || num_feat <= 0 || L <= 0 || L > 8 || C != 256 || G != 8) return HIPAD_EINVAL;
  memset(&m, 0, sizeof(m));
  long long total = 0;
  for (int k = 0; k < ncalls; ++k) {
    const hipad_daf_call &c = calls[k];
    if (!c.loc || !c.weights || !c.grad_out || c.num_anchors <= 0 || c.num_pts <= 0) return HIPAD_EINVAL;
    m.loc[k] = c.loc; m.wts[k] = c.weights; m.gout[k] = c.grad_out;
    m.A[k] = c.num_anchors; m.P[k] = c.num_pts;
    m.pbase[k] = (int)total;
    total += (long long)bs * c.num_anchors * c.num_pts * cams;
    if (!sorted_budget_ok(bs, num_feat, total, L)) return HIPAD_ERANGE;
  }
  m.pbase[ncalls] = (int)total;
  m.ncalls = ncalls;
  return HIPAD_OK;
}

}  // namespace hipad

extern "C" {

void hipad_daf_set_tap_chunks(int chunks) { hipad::daf_set_tap_chunks(chunks); }
void hipad_daf_set_feat_run(int batches) { hipad::daf_set_feat_run(batches); }
void hipad_daf_set_feat_blocks(int blocks) { hipad::daf_set_feat_blocks(blocks); }

size_t hipad_daf_backward_feat_multi_workspace(const hipad_daf_call *calls, int ncalls, int bs, int cams, int num_feat,
                                               int C, int L, int G) {
  hipad::MultiArgs m;
  if (hipad::multi_table(m, calls, ncalls, bs, cams, num_feat, C, L, G) != HIPAD_OK) return 0;
  return hipad::carve((size_t)bs * num_feat, (size_t)m.pbase[ncalls], L, nullptr).bytes;
}

int hipad_daf_backward_feat_multi(const hipad_daf_call *calls, int ncalls, float *grad_feat, const int32_t *spatial_shape,
                                  const int32_t *scale_start_index, int bs, int cams, int num_feat, int C, int L, int G,
                                  void *workspace, size_t workspace_bytes, hipad_stream_t stream) {
  hipad::MultiArgs m;
  const int rc = hipad::multi_table(m, calls, ncalls, bs, cams, num_feat, C, L, G);
  if (rc != HIPAD_OK) return rc;
  if (!grad_feat || !spatial_shape || !scale_start_index) return HIPAD_EINVAL;
  return hipad::run_sorted(m, spatial_shape, scale_start_index, grad_feat, bs, cams, num_feat, L, workspace,
                           workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"

namespace hipad {

int daf_bwd_lw(const void *feat, bool feat_bf16, const int *ss, const int *start, const float *loc, const float *wts,
               const float *gout, float *gloc, float *gw, const DafDims &d, int nchunks, int ppc,
               bool overwrite, hipStream_t stream) {
  const int n_items = d.bs * d.A * nchunks;
  const int blocks = (n_items + 3) / 4;
#define HIPAD_LW_T(LT, OW, GEO, FT)                                                                       \
  hipLaunchKernelGGL((daf_bwd_lw_kernel<LT, OW, GEO, FT>), dim3(blocks), dim3(256), 0, stream,            \
                     (const FT *)feat, ss, start, loc, wts, gout, gloc, gw, n_items, nchunks, ppc, d.cams, \
                     d.num_feat, d.L, d.A, d.P)
#define HIPAD_LW(LT, OW, GEO)                                                      \
  do {                                                                             \
    if (feat_bf16) HIPAD_LW_T(LT, OW, GEO, uint16_t); else HIPAD_LW_T(LT, OW, GEO, float); \
  } while (0)
  const bool geo = d.cams * d.L <= kWave;
  if (d.L == 4 && geo) {
    if (overwrite) HIPAD_LW(4, true, true); else HIPAD_LW(4, false, true);
  } else if (d.L == 4) {
    if (overwrite) HIPAD_LW(4, true, false); else HIPAD_LW(4, false, false);
  } else if (geo) {
    if (overwrite) HIPAD_LW(0, true, true); else HIPAD_LW(0, false, true);
  } else {
    if (overwrite) HIPAD_LW(0, true, false); else HIPAD_LW(0, false, false);
  }
#undef HIPAD_LW
#undef HIPAD_LW_T
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

}  // namespace hipad
