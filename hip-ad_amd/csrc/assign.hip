// hip-ad_amd/csrc/assign.hip -- minimum-cost one-to-one assignment (Hungarian / shortest augmenting path)
// on the device, one workgroup per problem.
//
// Replaces: scipy.optimize.linear_sum_assignment as called by the reference's target assignment -- boxes
// (models/det/target.py:98-103: cost (num_pred, num_gt) -> .detach().cpu().numpy() -> SciPy) and map lines
// (models/map/target.py:150-155, HungarianLinesAssigner) -- twelve blocking device->host copies + host solves
// per training step there, which also make the step impossible to capture in a hipGraph.
//
// Problem shape: R rows = ground-truth items (a few dozen), C columns = predictions (100 / 900); every row is
// assigned a distinct column minimising the total cost (rows <= cols), which is what SciPy returns for the
// transposed (num_pred x num_gt) matrix.  Arithmetic in fp64 on the fp32 costs, as SciPy's (it converts its
// input to double): the optimum is the same assignment whenever it is unique.
//
// Algorithm: potentials u (rows), v (cols); rows are inserted one at a time; each insertion grows an
// alternating tree from the new row, relaxing minv[j] over all unvisited columns IN PARALLEL (one thread per
// column slice), picking the closest column by a block arg-min (ties: lowest column index), shifting the
// potentials, until a free column is reached; then the path is flipped.  O(R^2 C / 256) steps.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hipad.h"

namespace hipad {

constexpr int kMaxCols = 2048;
constexpr int kMaxRows = 256;
constexpr double kInf = 1e300;

__global__ __launch_bounds__(256) void linear_assignment_kernel(int *__restrict__ col_of_row,
                                                                const float *__restrict__ cost /* [B][R][C] */,
                                                                const int *__restrict__ n_rows, int R, int C) {
  __shared__ double v[kMaxCols + 1], minv[kMaxCols + 1];
  __shared__ double u[kMaxRows + 1];
  __shared__ int p[kMaxCols + 1], way[kMaxCols + 1];
  __shared__ unsigned char used[kMaxCols + 1];
  __shared__ double red_val[4];
  __shared__ int red_idx[4];
  __shared__ int j0_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float *cb = cost + (size_t)b * R * C;
  int n = n_rows[b];
  if (n > R) n = R;
  if (n > C) n = C;  // cannot happen for valid input (host checks R <= C)
  for (int j = tid; j <= C; j += 256) { v[j] = 0.0; p[j] = 0; way[j] = 0; }
  for (int i = tid; i <= R; i += 256) u[i] = 0.0;
  for (int r = tid; r < R; r += 256) col_of_row[(size_t)b * R + r] = -1;
  __syncthreads();
  for (int i = 1; i <= n; ++i) {
    for (int j = tid; j <= C; j += 256) { minv[j] = kInf; used[j] = 0; }
    if (tid == 0) { p[0] = i; j0_s = 0; }
    __syncthreads();
    bool infeasible = false;
    while (true) {
      const int j0 = j0_s;
      const int i0 = p[j0];
      __syncthreads();  // everyone has read j0_s / p[j0] before they change
      if (tid == 0) used[j0] = 1;
      const double ui0 = u[i0];
      const float *crow = cb + (size_t)(i0 - 1) * C;
      double best = kInf;
      int best_j = 0x7fffffff;
      for (int j = 1 + tid; j <= C; j += 256) {
        if (used[j] || j == j0) continue;
        const double cur = (double)crow[j - 1] - ui0 - v[j];
        double m = minv[j];
        if (cur < m) { m = cur; minv[j] = cur; way[j] = j0; }
        if (m < best) { best = m; best_j = j; }  // ascending j per thread: first (lowest) index wins ties
      }
      // block arg-min, ties to the lowest column index
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(best, o);
        const int oj = __shfl_xor(best_j, o);
        if (ov < best || (ov == best && oj < best_j)) { best = ov; best_j = oj; }
      }
      if (lane == 0) { red_val[wv] = best; red_idx[wv] = best_j; }
      __syncthreads();
      double delta = red_val[0];
      int j1 = red_idx[0];
#pragma unroll
      for (int w = 1; w < 4; ++w)
        if (red_val[w] < delta || (red_val[w] == delta && red_idx[w] < j1)) { delta = red_val[w]; j1 = red_idx[w]; }
      if (j1 == 0x7fffffff || !(delta < kInf)) {
        // no unvisited column with a finite reduced cost (+inf / NaN costs): the problem is infeasible for this row.
        // SciPy raises; a kernel cannot -- the row stays unassigned (-1) and the search for it ends here, without
        // indexing p[] / way[] by the sentinel.  (Uniform: every thread derived j1 / delta from the same LDS words.)
        infeasible = true;
        break;
      }
      // shift potentials (used[] now includes j0: thread 0 wrote it before the barrier above)
      for (int j = tid; j <= C; j += 256) {
        if (used[j]) { u[p[j]] += delta; v[j] -= delta; }
        else minv[j] -= delta;
      }
      __syncthreads();
      if (tid == 0) j0_s = j1;
      const bool done = p[j1] == 0;
      __syncthreads();
      if (done) break;
    }
    if (infeasible) {
      __syncthreads();
      continue;
    }
    if (tid == 0) {  // flip the alternating path
      int j0 = j0_s;
      do {
        const int j1 = way[j0];
        p[j0] = p[j1];
        j0 = j1;
      } while (j0);
    }
    __syncthreads();
  }
  for (int j = 1 + tid; j <= C; j += 256)
    if (p[j] > 0) col_of_row[(size_t)b * R + (p[j] - 1)] = j - 1;
}

}  // namespace hipad

using namespace hipad;

extern "C" {

int hipad_linear_assignment(int *col_of_row, const float *cost, const int *n_rows, int batch, int rows, int cols,
                            hipad_stream_t stream) {
  if (!col_of_row || !cost || !n_rows || batch <= 0 || rows <= 0 || cols <= 0) return HIPAD_EINVAL;
  if (rows > kMaxRows || cols > kMaxCols || rows > cols) return HIPAD_ERANGE;
  hipLaunchKernelGGL(linear_assignment_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, col_of_row, cost,
                     n_rows, rows, cols);
  return hipGetLastError() == hipSuccess ? HIPAD_OK : HIPAD_ELAUNCH;
}

}  // extern "C"
