// Generic grad-log-likelihood kernel: any FCN spec (widths, activation, task).
// Correctness/coverage path; the MFMA kernels in mile_grad_w64.h are the fast path.
//
// Computes, per (particle e, row split s), the gradient of the LIKELIHOOD part of
// log_unnormalized_posterior (src/training/probabilistic.py:68-113) over the rows of
// the split, through FullyConnected (src/flax_building_blocks/basic.py:42-61), into
// slab[e][s][0..d) and the partial log-likelihood into llpart[e][s].  The prior and
// the sum over splits are applied by the consumer (k_finalize / k_update).
#pragma once
#include "mile_device.h"

struct GradParams {
  DevSpec spec;
  const float *theta;   // [E, d]
  const float *X;       // [N, F]
  const float *Xp;      // [Npad, Fp] zero padded (MFMA kernels)
  const void *Xb;       // [Npad, 16] bf16 zero padded, and
  const void *Xt;       // [32, Npb] bf16 transposed (k_grad_w128b); Npb = N rounded up to 64 rows
  const void *y;        // [Npad] fp32 (regr) or int32 (classification)
  float *slabs;         // [E, S, d]
  float *llpart;        // [E, S]
  int32_t N, Npad, Fp, S, R;
  int32_t Npb;
  int32_t dp;           // slab row stride: d rounded up to a multiple of 4 floats (128-bit stores)
  long long *dbg_buf;   // dev: 8 per-phase cycle counters (k_grad_w128b, MILE_DEBUG=16)
  int32_t dbg;          // debug knobs (MILE_DEBUG env): bit0 skip row blocks, bit1 skip staging, bit2 skip reduction
};

// Per-row log-likelihood and d/d(out).  NaN rows contribute nothing (jnp.nansum).
__device__ __forceinline__ float row_loss_regr(float mu, float sr, float yv, float &dmu, float &ds) {
  const float es = expf(sr);
  const float sig = fminf(fmaxf(es, 1e-6f), 1e6f);
  const bool unclipped = (es > 1e-6f) && (es < 1e6f);
  const float r = (yv - mu) / sig;
  float ll = -0.5f * r * r - logf(sig) - 0.91893853320467274f;
  dmu = r / sig;
  ds = unclipped ? (r * r - 1.0f) : 0.0f;
  if (isnan(ll) || isnan(es) || isnan(mu)) { ll = 0.0f; dmu = 0.0f; ds = 0.0f; }
  return ll;
}

static __global__ __launch_bounds__(256) void k_grad_generic(const GradParams p) {
  extern __shared__ float lds[];
  const DevSpec &sp = p.spec;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int e = blockIdx.y, s = blockIdx.x;
  const int d = sp.d, nl = sp.n_layers, as = sp.act_stride, mw = sp.max_width, R = p.R;
  const float *th = p.theta + (size_t)e * d;
  float *slab = p.slabs + ((size_t)e * p.S + s) * p.dp;
  float *act = lds;
  float *dzc = act + R * as;
  float *dzn = dzc + R * mw;
  float *red = dzn + R * mw;

  for (int i = tid; i < d; i += nt) slab[i] = 0.0f;

  const int rows_per = (p.N + p.S - 1) / p.S;
  const int r_begin = s * rows_per;
  const int r_end = min(p.N, r_begin + rows_per);
  const int F = sp.in_features;
  float ll_acc = 0.0f;

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
      float *dz = dzc + r * mw;
      if (sp.task == MILE_TASK_REGRESSION) {
        float dmu, ds;
        ll_acc += row_loss_regr(out[0], out[1], ((const float *)p.y)[t0 + r], dmu, ds);
        dz[0] = dmu; dz[1] = ds;
      } else {
        const int yi = ((const int32_t *)p.y)[t0 + r];
        float m = out[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, out[c]);
        float se = 0.0f;
        for (int c = 0; c < C; ++c) se += expf(out[c] - m);
        const float lse = m + logf(se);
        const float ll = out[yi] - lse;
        const bool bad = isnan(ll);
        for (int c = 0; c < C; ++c) dz[c] = bad ? 0.0f : ((c == yi ? 1.0f : 0.0f) - expf(out[c] - lse));
        ll_acc += bad ? 0.0f : ll;
      }
    }
    __syncthreads();
    for (int l = nl - 1; l >= 0; --l) {
      const int win = l == 0 ? F : sp.widths[l - 1], wout = sp.widths[l];
      const float *W = th + sp.w_off[l];
      for (int idx = tid; idx < win * wout; idx += nt) {
        const int i = idx / wout, o = idx - i * wout;
        const float *a = act + sp.act_off[l] + i;
        float acc = 0.0f;
        for (int r = 0; r < nr; ++r) acc = fmaf(a[r * as], dzc[r * mw + o], acc);
        slab[sp.w_off[l] + idx] += acc;
      }
      for (int o = tid; o < wout; o += nt) {
        float acc = 0.0f;
        for (int r = 0; r < nr; ++r) acc += dzc[r * mw + o];
        slab[sp.b_off[l] + o] += acc;
      }
      if (l > 0) {
        for (int idx = tid; idx < nr * win; idx += nt) {
          const int r = idx / win, i = idx - r * win;
          float dh = 0.0f;
          for (int o = 0; o < wout; ++o) dh = fmaf(dzc[r * mw + o], W[i * wout + o], dh);
          const float hval = act[r * as + sp.act_off[l] + i];
          dzn[r * mw + i] = dh * act_bwd(sp.activation, hval);
        }
      }
      __syncthreads();
      float *t = dzc; dzc = dzn; dzn = t;
    }
  }
  ll_acc = wave_sum(ll_acc);
  if ((tid & 63) == 0) red[tid >> 6] = ll_acc;
  __syncthreads();
  if (tid == 0) {
    float t = 0.0f;
    for (int w = 0; w < nt / 64; ++w) t += red[w];
    p.llpart[(size_t)e * p.S + s] = t;
  }
}
