// Layer-wise gradient path for wide FCNs (MILE_GRAD_GEMM_F32): the Dense products are plain strided
// batched SGEMMs (rocBLAS, one batch entry per particle, weights read in place from the [E, d]
// parameter array, weight gradients accumulated in place into the slab, bias gradients as the skinny product dZ^T 1),
// everything between them is
// the elementwise HIP kernels below.  fp32 throughout: same numerics class as k_grad_generic, which
// it replaces where hidden widths are large enough for a library GEMM to win (B3/B4-class nets; the
// fused kernels k_grad_w64 / k_grad_w128b cover the shapes they were written for).
//
// Same maths as k_grad_generic: Dense stack src/flax_building_blocks/basic.py:42-61, likelihoods
// src/training/probabilistic.py:92-109 (nansum), activations src/config/models/base.py:25-39.
#pragma once
#include "mile_device.h"
#include "mile_grad_generic.h"

// H = act(Z + b) in place.  Z [E][R][W] (row stride W), b = theta + b_off (particle stride d).
__global__ __launch_bounds__(256) void k_gemm_bias_act(float *Z, const float *theta, int b_off, int d, int W, long long RW,
                                                       int activation, int apply_act) {
  const int e = blockIdx.y;
  float *z = Z + (size_t)e * RW;
  const float *b = theta + (size_t)e * d + b_off;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < RW; i += (long long)gridDim.x * 256) {
    const float v = z[i] + b[i % W];
    z[i] = apply_act ? act_fwd(activation, v) : v;
  }
}

// dZ = dH * act'(H) in place on dH; the derivative is a function of the activation's output.
__global__ __launch_bounds__(256) void k_gemm_act_grad(float *dH, const float *H, long long RW, int activation) {
  const int e = blockIdx.y;
  float *g = dH + (size_t)e * RW;
  const float *hh = H + (size_t)e * RW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < RW; i += (long long)gridDim.x * 256)
    g[i] *= act_bwd(activation, hh[i]);
}

// Head: per-row log-likelihood and d/d(out), in place on out [E][R][K]; adds the chunk's
// log-likelihood to llacc[e] (one workgroup per particle: deterministic order).
__global__ __launch_bounds__(256) void k_gemm_head(float *out, const void *y, long long r0, int R, int K, int task, float *llacc,
                                                   int first_chunk) {
  __shared__ float red[4];
  const int e = blockIdx.x, tid = threadIdx.x;
  float *o = out + (size_t)e * R * K;
  float ll = 0.0f;
  for (int r = tid; r < R; r += 256) {
    float *z = o + (size_t)r * K;
    if (task == MILE_TASK_REGRESSION) {
      float dmu, ds;
      ll += row_loss_regr(z[0], z[1], ((const float *)y)[r0 + r], dmu, ds);
      z[0] = dmu; z[1] = ds;
    } else {
      const int yi = ((const int32_t *)y)[r0 + r];
      float m = z[0];
      for (int c = 1; c < K; ++c) m = fmaxf(m, z[c]);
      float se = 0.0f;
      for (int c = 0; c < K; ++c) se += expf(z[c] - m);
      const float lse = m + logf(se);
      const float l1 = z[yi] - lse;
      const bool bad = isnan(l1);
      for (int c = 0; c < K; ++c) z[c] = bad ? 0.0f : ((c == yi ? 1.0f : 0.0f) - expf(z[c] - lse));
      ll += bad ? 0.0f : l1;
    }
  }
  ll = wave_sum(ll);
  if ((tid & 63) == 0) red[tid >> 6] = ll;
  __syncthreads();
  if (tid == 0) {
    const float t = (red[0] + red[1]) + (red[2] + red[3]);
    llacc[e] = first_chunk ? t : llacc[e] + t;
  }
}

// Evaluation: per-row log-likelihood of the head outputs (no nansum zeroing: src/inference/metrics.py:247-294
// uses the distributions' log_prob directly).  out_ll[(s0 + s) * N + r0 + r].
__global__ __launch_bounds__(256) void k_gemm_rowll(const float *out, const void *y, long long r0, int R, int K, int task, float *out_ll,
                                                    long long N, long long s0) {
  const int s = blockIdx.y;
  const float *o = out + (size_t)s * R * K;
  for (int r = blockIdx.x * 256 + threadIdx.x; r < R; r += gridDim.x * 256) {
    const float *z = o + (size_t)r * K;
    float v;
    if (task == MILE_TASK_REGRESSION) {
      const float es = expf(z[1]);
      const float sig = isnan(es) ? es : fminf(fmaxf(es, 1e-6f), 1e6f);
      const float rr = (((const float *)y)[r0 + r] - z[0]) / sig;
      v = -0.5f * rr * rr - logf(sig) - 0.91893853320467274f;
    } else {
      const int yi = ((const int32_t *)y)[r0 + r];
      float m = z[0];
      for (int c = 1; c < K; ++c) m = fmaxf(m, z[c]);
      float se = 0.0f;
      for (int c = 0; c < K; ++c) se += expf(z[c] - m);
      v = z[yi] - (m + logf(se));
    }
    out_ll[(size_t)(s0 + s) * N + r0 + r] = v;
  }
}
