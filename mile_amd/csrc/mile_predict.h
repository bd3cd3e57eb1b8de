// Forward-only kernels for evaluation: pointwise log predictive density of every sample on a test set,
// i.e. the per-sample module.apply + Normal / Categorical log_prob of the reference's evaluation
// (src/inference/evaluation.py:16-43, src/inference/metrics.py:247-294).  out[s][n], NaNs are NOT zeroed
// (metrics.py calls log_prob directly, there is no nansum there).
#pragma once
#include "mile_grad_w64.h"

struct PredParams {
  DevSpec spec;
  const float *theta;   // [S, d]
  const float *X;       // [N, F]
  const float *Xp;      // [Npad, Fp]
  const void *y;        // [Npad]
  float *out;           // [S, N]
  int32_t N, Npad, Fp, SB, R;
};

__device__ __forceinline__ float row_logpdf_regr(float mu, float sr, float yv) {
  const float es = expf(sr);
  const float sig = isnan(es) ? es : fminf(fmaxf(es, 1e-6f), 1e6f);
  const float r = (yv - mu) / sig;
  return -0.5f * r * r - logf(sig) - 0.91893853320467274f;
}

// width-64 ReLU regression nets: the forward half of k_grad_w64 (same LDS images, same T layout)
template <int NH, int FQ>
__global__ __launch_bounds__(256, 1) void k_fwd_w64(const PredParams p) {
  using LY = W64Layout<NH, FQ>;
  constexpr int FP = LY::FP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const DevSpec &sp = p.spec;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int e = blockIdx.y, s = blockIdx.x;
  const int F = sp.in_features;
  const float *th = p.theta + (size_t)e * sp.d;
  float *WIMG = lds + LY::WIMG, *W1IMG = lds + LY::W1IMG, *BIAS = lds + LY::BIAS;
  float *WO = lds + LY::WO, *BO = lds + LY::BO;
  for (int l = 1; l < NH; ++l) {
    const float *W = th + sp.w_off[l];
    for (int idx = tid; idx < 4096; idx += 256) WIMG[(l - 1) * 64 * W64_RS + (idx >> 6) * W64_RS + (idx & 63)] = W[idx];
  }
  {
    const float *W = th + sp.w_off[0];
    for (int idx = tid; idx < FP * 64; idx += 256) {
      const int r = idx >> 6;
      W1IMG[r * W64_RS + (idx & 63)] = r < F ? W[idx] : 0.0f;
    }
    for (int idx = tid; idx < NH * 64; idx += 256) BIAS[idx] = th[sp.b_off[idx >> 6] + (idx & 63)];
    const float *Wo = th + sp.w_off[NH];
    if (tid < 128) WO[(tid & 1) * 64 + (tid >> 1)] = Wo[tid];
    if (tid < 2) BO[tid] = th[sp.b_off[NH] + tid];
  }
  __syncthreads();
  const int NB = p.Npad / 32;
  const int b0 = (int)(((long long)s * NB) / p.SB), b1 = (int)(((long long)(s + 1) * NB) / p.SB);
  for (int blk = b0 + wave; blk < b1; blk += 4) {
    const int row0 = blk * 32;
    f32x4 xv[FQ];
#pragma unroll
    for (int q = 0; q < FQ; ++q) xv[q] = *(const f32x4 *)(p.Xp + (size_t)(row0 + j) * FP + 8 * q + 4 * h);
    f32x16 H[2][2];   // ping-pong over layers
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
      f32x16 acc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *(const f32x4 *)(BIAS + 32 * ob + 8 * g + 4 * h);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[4 * g + m] = bv[m];
      }
#pragma unroll
      for (int q = 0; q < FQ; ++q)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W1IMG[(8 * q + 4 * h + m) * W64_RS + 32 * ob + j], xv[q][m], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) H[0][ob][r] = relu1(acc[r]);
    }
#pragma unroll
    for (int l = 1; l < NH; ++l) {
      const float *Wl = WIMG + (l - 1) * 64 * W64_RS;
#pragma unroll
      for (int ob = 0; ob < 2; ++ob) {
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 bv = *(const f32x4 *)(BIAS + l * 64 + 32 * ob + 8 * g + 4 * h);
#pragma unroll
          for (int m = 0; m < 4; ++m) acc[4 * g + m] = bv[m];
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int t = 0; t < 16; ++t)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Wl[(32 * kb + tfeat(t, h)) * W64_RS + 32 * ob + j],
                                                      H[(l - 1) & 1][kb][t], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) H[l & 1][ob][r] = relu1(acc[r]);
      }
    }
    float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 w0 = *(const f32x4 *)(WO + 32 * kb + 8 * g + 4 * h);
        const f32x4 w1 = *(const f32x4 *)(WO + 64 + 32 * kb + 8 * g + 4 * h);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          p0 = fmaf(w0[m], H[(NH - 1) & 1][kb][4 * g + m], p0);
          p1 = fmaf(w1[m], H[(NH - 1) & 1][kb][4 * g + m], p1);
        }
      }
    p0 += __shfl_xor(p0, 32);
    p1 += __shfl_xor(p1, 32);
    if (h == 0 && row0 + j < p.N)
      p.out[(size_t)e * p.N + row0 + j] = row_logpdf_regr(p0 + BO[0], p1 + BO[1], ((const float *)p.y)[row0 + j]);
  }
}

// any FCN: one thread per (row, unit), activations tiled through LDS (as k_grad_generic's forward)
__global__ __launch_bounds__(256) void k_fwd_generic(const PredParams p) {
  extern __shared__ float lds[];
  const DevSpec &sp = p.spec;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int e = blockIdx.y, s = blockIdx.x;
  const int nl = sp.n_layers, as = sp.act_stride, R = p.R, F = sp.in_features;
  const float *th = p.theta + (size_t)e * sp.d;
  float *act = lds;
  const int rows_per = (p.N + p.SB - 1) / p.SB;
  const int r_begin = s * rows_per, r_end = min(p.N, r_begin + rows_per);
  for (int t0 = r_begin; t0 < r_end; t0 += R) {
    const int nr = min(R, r_end - t0);
    for (int idx = tid; idx < nr * F; idx += nt) {
      const int r = idx / F, c = idx - r * F;
      act[r * as + c] = p.X[(size_t)(t0 + r) * F + c];
    }
    __syncthreads();
    for (int l = 0; l < nl; ++l) {
      const int win = l == 0 ? F : sp.widths[l - 1], wout = sp.widths[l];
      const float *W = th + sp.w_off[l], *b = th + sp.b_off[l];
      for (int idx = tid; idx < nr * wout; idx += nt) {
        const int r = idx / wout, o = idx - r * wout;
        const float *a = act + r * as + sp.act_off[l];
        float z = b[o];
        for (int i = 0; i < win; ++i) z = fmaf(a[i], W[i * wout + o], z);
        if (l < nl - 1) z = act_fwd(sp.activation, z);
        act[r * as + sp.act_off[l + 1] + o] = z;
      }
      __syncthreads();
    }
    const int C = sp.widths[nl - 1];
    for (int r = tid; r < nr; r += nt) {
      const float *out = act + r * as + sp.act_off[nl];
      float v;
      if (sp.task == MILE_TASK_REGRESSION) {
        v = row_logpdf_regr(out[0], out[1], ((const float *)p.y)[t0 + r]);
      } else {
        const int yi = ((const int32_t *)p.y)[t0 + r];
        float m = out[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, out[c]);
        float se = 0.0f;
        for (int c = 0; c < C; ++c) se += expf(out[c] - m);
        v = out[yi] - (m + logf(se));
      }
      p.out[(size_t)e * p.N + t0 + r] = v;
    }
    __syncthreads();
  }
}
