// Shared device-side definitions for libmile_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mile_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Flattened description of the FCN that kernels take by value.
struct DevSpec {
  int32_t n_layers;
  int32_t in_features;
  int32_t widths[MILE_MAX_LAYERS];
  int32_t w_off[MILE_MAX_LAYERS];   // kernel[in,out] offset in the raveled vector
  int32_t b_off[MILE_MAX_LAYERS];   // bias[out] offset
  int32_t act_off[MILE_MAX_LAYERS + 1];  // per-row activation record: [0]=input, [l+1]=output of layer l
  int32_t act_stride;
  int32_t max_width;
  int32_t activation;
  int32_t task;
  int32_t prior;
  float prior_loc;
  float prior_scale;
  int32_t d;
};

// ---------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller.  counter = (quad index, particle id, step, stage),
// key = (seed_lo, seed_hi).  Restated on the CPU in oracle/mclmc_oracle.py:philox_normal.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;   // one v_mad_u64_u32 instead of mul_hi + mul_lo
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Four N(0,1) draws for elements 4q..4q+3 of one particle's noise vector.
__device__ __forceinline__ f32x4 philox_normal4(uint32_t quad, uint32_t pid, uint32_t step, uint32_t stage,
                                                uint64_t seed) {
  uint32_t w[4];
  philox4x32_10(quad, pid, step, stage, (uint32_t)seed, (uint32_t)(seed >> 32), w);
  f32x4 z;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float u1 = ((float)w[2 * p] + 1.0f) * 2.3283064365386963e-10f;  // (w+1) * 2^-32 in (0,1]
    const float u2 = (float)w[2 * p + 1] * 2.3283064365386963e-10f;       // [0,1]: revolutions
    // hardware transcendentals: v_log_f32 is log2, v_sin/v_cos take revolutions
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    z[2 * p] = r * __builtin_amdgcn_cosf(u2);
    z[2 * p + 1] = r * __builtin_amdgcn_sinf(u2);
  }
  return z;
}

// ReLU in ONE instruction: fmaxf(x, 0) and v_med3_f32(x, 0, inf) both get a second one from the compiler (v_max x,x,x to
// quiet a signalling NaN).  As a signed-integer max the sign bit does the work: negative floats (and -0, and NaNs with the
// sign bit set) are negative integers -> 0, everything else passes through unchanged.
__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

__device__ __forceinline__ float act_fwd(int act, float z) {
  if (act == MILE_ACT_RELU) return fmaxf(z, 0.0f);
  if (act == MILE_ACT_TANH) return tanhf(z);
  return 1.0f / (1.0f + expf(-z));
}
// derivative expressed through the activation OUTPUT h (relu'(0) = 0 as in JAX)
__device__ __forceinline__ float act_bwd(int act, float h) {
  if (act == MILE_ACT_RELU) return h > 0.0f ? 1.0f : 0.0f;
  if (act == MILE_ACT_TANH) return 1.0f - h * h;
  return h * (1.0f - h);
}

// Sum over the 64 lanes of a wave, result in every lane.  DPP only (no LDS traffic: __shfl_xor
// compiles to ds_bpermute_b32): butterfly inside each 16-lane row, then row_bcast:15 / :31 carry
// the row sums up to lane 63, which is read back through an SGPR.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
  return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xF>(v);   // row_half_mirror
  v = dpp_add<0x140, 0xF>(v);   // row_mirror: every lane now holds its row's sum
  v = dpp_add<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3: lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
