// hip-ad_amd/csrc/wave_ops.h -- cross-lane helpers for 64-wide waves (gfx950), shared by all kernels.
#ifndef HIPAD_WAVE_OPS_H_
#define HIPAD_WAVE_OPS_H_
#include <hip/hip_runtime.h>

namespace hipad {

__device__ __forceinline__ float rl_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ int rl_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Cross-lane sums on the DPP path.  __shfl_xor compiles to ds_bpermute_b32 here: an LDS-crossbar round trip of
// ~100 cycles that the next step has to wait for, i.e. 6 serialised round trips per wave sum.  DPP operands are read
// inside the VALU (a few cycles each).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_take(float v) {  // rows outside ROW_MASK and lanes without a source read 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float quad_sum(float v) {  // every lane: sum over its 4 lanes
  v += dpp_take<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_take<0x4E>(v);   // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ float oct_sum(float v) {   // every lane: sum over its 8 lanes
  v = quad_sum(v);
  return v + dpp_take<0x141>(v);  // row_half_mirror
}
__device__ __forceinline__ float row_sum(float v) {   // every lane: sum over its row of 16 lanes
  v = oct_sum(v);
  return v + dpp_take<0x140>(v);  // row_mirror
}
__device__ __forceinline__ float half_wave_sum(float v) {
  // sum over the 32 lanes of this lane's half; every lane of the half gets the result
  v = row_sum(v);
  v += dpp_take<0x142, 0xA>(v);   // row_bcast15: rows 1 and 3 add the row before them
  const float lo = rl_f(v, 31), hi = rl_f(v, 63);
  return (threadIdx.x & 32) ? hi : lo;
}
__device__ __forceinline__ float wave_sum(float v) {  // wave-uniform result
  v = row_sum(v);
  v += dpp_take<0x142, 0xA>(v);   // row_bcast15 into rows 1, 3
  v += dpp_take<0x143, 0xC>(v);   // row_bcast31 into rows 2, 3
  return rl_f(v, 63);
}

}  // namespace hipad
#endif  // HIPAD_WAVE_OPS_H_
