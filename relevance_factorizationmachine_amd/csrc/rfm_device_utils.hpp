// Device helpers shared by the FM and MF translation units (gfx950): lane-group reductions
// through DPP moves, 8- / 16-byte factor packs and the reference's clipped sigmoid
// (src/base.py:63-66).
#pragma once

#include <hip/hip_runtime.h>

namespace rfm {

constexpr double kLogitClip = 700.0;  // src/base.py:65

// Sum over the LPR lanes of a lane group; every lane gets the total.  The steps
// inside a row of 16 lanes are DPP moves (quad permutes, half-row and row mirrors:
// a few cycles each) instead of ds_bpermute round trips through the LDS crossbar.
template <int CTRL>
__device__ inline double dpp_move(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

template <int LPR>
__device__ inline double group_sum(double v) {
  if (LPR >= 2) v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
  if (LPR >= 4) v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
  if (LPR >= 8) v += dpp_move<0x141>(v);  // row_half_mirror
  if (LPR >= 16) v += dpp_move<0x140>(v); // row_mirror
  if (LPR >= 32) v += __shfl_xor(v, 16, 64);
  if (LPR >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

__device__ inline double sigmoid_clipped(double z) {
  // np.clip keeps a NaN logit NaN (src/base.py:65); fmin/fmax alone would turn it into -700
  z = z != z ? z : fmin(fmax(z, -kLogitClip), kLogitClip);
  return 1.0 / (1.0 + exp(-z));
}

template <int VEC>
struct Pack;
template <>
struct Pack<1> {
  double v[1];
  __device__ inline void load(const double* p) { v[0] = *p; }
  __device__ inline void store(double* p) const { *p = v[0]; }
};
template <>
struct Pack<2> {
  double v[2];
  __device__ inline void load(const double* p) {
    const double2 t = *reinterpret_cast<const double2*>(p);
    v[0] = t.x;
    v[1] = t.y;
  }
  __device__ inline void store(double* p) const {
    *reinterpret_cast<double2*>(p) = make_double2(v[0], v[1]);
  }
};

}  // namespace rfm
