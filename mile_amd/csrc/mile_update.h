// Integrator kernels: everything of one MCLMC step that is not the likelihood gradient.
//
// One workgroup per particle, state laid out [E, d] so a wave reads 64 consecutive
// parameters.  A step of blackjax's kernel (call sites
// src/training/warmup.py:286-291, src/training/sampling.py:151) is
//     O(eps/2, z1) . B(b1) . A(1/2) . grad . B(1-2 b1) . A(1/2) . grad . B(b1) . O(eps/2, z2)
// and every op between two gradient evaluations is a linear combination of the SAME
// four vectors {u, e = g~/|g~|, zA, zB} followed by a normalisation.  So k_update
//   pass 1: forms g = sum_s slab[s] + grad log prior, and reduces the 10 pairwise dot
//           products of those vectors (+ the log-prior value) in ONE sweep;
//   scalar: runs the B / O chain on 4 coefficients using the Gram matrix (fp64, a few
//           dozen flops) -- this is where |g|, u.e and the norms of A.2 / A.5 come from;
//   pass 2: writes u' = c . {u, e, zA, zB}, the thinned sample, and x' = x + eps a u' s.
// A whole "end of step i + start of step i+1" sequence is therefore one launch.
#pragma once
#include "mile_device.h"

enum : int32_t {
  UPD_FROM_SLABS = 1 << 0,  // g, logp come from the grad kernel's slabs (else from state)
  UPD_START = 1 << 1,       // first op of a step sequence: dK = 0, l_old = logp
  UPD_B1 = 1 << 2,          // B(coef_b1) before the record point
  UPD_OA = 1 << 3,          // O with noise A before the record point
  UPD_RECORD = 1 << 4,      // end of a kernel step: write info, (emit sample), restart dK / l_old
  UPD_OB = 1 << 5,          // O with noise B after the record point
  UPD_B2 = 1 << 6,          // B(coef_b2) after the record point
  UPD_A = 1 << 7,           // position update with coef_a
};

struct UpdParams {
  int32_t d, E, S, flags;
  int32_t prior;
  float prior_loc, prior_scale;
  float *x, *u, *g, *logp;
  const float *slabs, *llpart;
  const float *eps, *L, *sdc;
  const float *zA, *zB;          // explicit noise [E, d] or NULL -> Philox
  uint64_t seed;
  const int32_t *pids;
  uint32_t stepA, stageA, stepB, stageB;
  float coef_b1, coef_b2, coef_a;
  float hA, hB;                  // O-step time as a multiple of eps
  float *dK, *lold;              // workspace [E]
  float *out_sample;             // [E, d] or NULL
  float *out_info;               // [E, 3] or NULL
};

struct Chain {
  double M[4][4];
  double c[4];
  __device__ double norm() const {
    double s = 0.0;
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b < 4; ++b) s += c[a] * M[a][b] * c[b];
    return sqrt(s);
  }
  __device__ void normalize() {
    const double n = norm();
    if (n > 1e-13)  // blackjax normalized_flatten_array tolerance
      for (int a = 0; a < 4; ++a) c[a] /= n;
  }
  // esh_dynamics_momentum_update_one_step (A.2); returns the kinetic energy change
  __device__ double B(double eps, double coef, double gnorm, int d) {
    double ue = 0.0;
    for (int a = 0; a < 4; ++a) ue += c[a] * M[a][1];
    const double delta = eps * coef * gnorm / (double)(d - 1);
    const double zeta = exp(-delta);
    const double beta = (1.0 - zeta) * (1.0 + zeta + ue * (1.0 - zeta));
    for (int a = 0; a < 4; ++a) c[a] *= 2.0 * zeta;
    c[1] += beta;
    normalize();
    // (d-1) (delta - ln2 + ln(1 + ue + (1-ue) zeta^2)), written without the cancellation
    return (double)(d - 1) * (delta + log1p(0.5 * (1.0 - ue) * expm1(-2.0 * delta)));
  }
  // partially_refresh_momentum (A.5)
  __device__ void O(int k, double hstep, double L, int d) {
    const double nu = sqrt(expm1(2.0 * hstep / L) / (double)d);
    c[k] += nu;
    const double n = norm();
    for (int a = 0; a < 4; ++a) c[a] /= n;
  }
};

#define UPD_NT 256
#define UPD_NSUM 11

__global__ __launch_bounds__(UPD_NT) void k_update(const UpdParams p) {
  __shared__ float red[UPD_NT / 64][UPD_NSUM];
  __shared__ float bc[8];
  const int tid = threadIdx.x, e = blockIdx.x, d = p.d;
  const size_t base = (size_t)e * d;
  const int nq = (d + 3) >> 2;
  const bool from_slabs = p.flags & UPD_FROM_SLABS;
  const bool useA = p.flags & UPD_OA, useB = p.flags & UPD_OB;
  const uint32_t pid = p.pids ? (uint32_t)p.pids[e] : (uint32_t)e;
  const float *sl = p.slabs + (size_t)e * p.S * d;

  // ---- pass 1 ----------------------------------------------------------------------
  float sm[UPD_NSUM];
#pragma unroll
  for (int k = 0; k < UPD_NSUM; ++k) sm[k] = 0.0f;
  for (int q = tid; q < nq; q += UPD_NT) {
    f32x4 za = {0, 0, 0, 0}, zb = {0, 0, 0, 0};
    if (useA && !p.zA) za = philox_normal4(q, pid, p.stepA, p.stageA, p.seed);
    if (useB && !p.zB) zb = philox_normal4(q, pid, p.stepB, p.stageB, p.seed);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int i = 4 * q + m;
      if (i < d) {
        float gi;
        const float xi = p.x[base + i];
        if (from_slabs) {
          gi = 0.0f;
          for (int s = 0; s < p.S; ++s) gi += sl[(size_t)s * d + i];
          const float t = (xi - p.prior_loc) / p.prior_scale;
          if (p.prior == MILE_PRIOR_NORMAL) {
            gi -= t / p.prior_scale;
            sm[10] += -0.5f * t * t;
          } else {
            gi -= (t > 0.0f ? 1.0f : (t < 0.0f ? -1.0f : 0.0f)) / p.prior_scale;
            sm[10] += -fabsf(t);
          }
          p.g[base + i] = gi;
        } else {
          gi = p.g[base + i];
        }
        const float gs = p.sdc ? gi * p.sdc[base + i] : gi;
        const float ui = p.u[base + i];
        const float a = useA ? (p.zA ? p.zA[base + i] : za[m]) : 0.0f;
        const float b = useB ? (p.zB ? p.zB[base + i] : zb[m]) : 0.0f;
        sm[0] = fmaf(ui, ui, sm[0]); sm[1] = fmaf(ui, gs, sm[1]); sm[2] = fmaf(gs, gs, sm[2]);
        sm[3] = fmaf(ui, a, sm[3]); sm[4] = fmaf(gs, a, sm[4]); sm[5] = fmaf(a, a, sm[5]);
        sm[6] = fmaf(ui, b, sm[6]); sm[7] = fmaf(gs, b, sm[7]); sm[8] = fmaf(b, b, sm[8]);
        sm[9] = fmaf(a, b, sm[9]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < UPD_NSUM; ++k) sm[k] = wave_sum(sm[k]);
  if ((tid & 63) == 0)
#pragma unroll
    for (int k = 0; k < UPD_NSUM; ++k) red[tid >> 6][k] = sm[k];
  __syncthreads();

  // ---- scalar chain (every thread, redundantly, in fp64) ---------------------------
  double S[UPD_NSUM];
#pragma unroll
  for (int k = 0; k < UPD_NSUM; ++k) {
    double t = 0.0;
    for (int w = 0; w < UPD_NT / 64; ++w) t += (double)red[w][k];
    S[k] = t;
  }
  const double eps = p.eps[e], L = p.L[e];
  double logp_now;
  if (from_slabs) {
    double ll = 0.0;
    for (int s = 0; s < p.S; ++s) ll += (double)p.llpart[(size_t)e * p.S + s];
    const double cst = p.prior == MILE_PRIOR_NORMAL
                           ? -(double)d * (log((double)p.prior_scale) + 0.91893853320467274)
                           : -(double)d * log(2.0 * (double)p.prior_scale);
    logp_now = ll + S[10] + cst;
  } else {
    logp_now = p.logp[e];
  }
  const double gn = S[2] > 0.0 ? sqrt(S[2]) : 1.0;
  Chain ch;
  ch.M[0][0] = S[0]; ch.M[0][1] = S[1] / gn; ch.M[0][2] = S[3]; ch.M[0][3] = S[6];
  ch.M[1][1] = 1.0;  ch.M[1][2] = S[4] / gn; ch.M[1][3] = S[7] / gn;
  ch.M[2][2] = S[5]; ch.M[2][3] = S[9];
  ch.M[3][3] = S[8];
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < a; ++b) ch.M[a][b] = ch.M[b][a];
  ch.c[0] = 1.0; ch.c[1] = ch.c[2] = ch.c[3] = 0.0;

  double dk = (p.flags & UPD_START) ? 0.0 : (double)p.dK[e];
  double lold = (p.flags & UPD_START) ? logp_now : (double)p.lold[e];
  if (p.flags & UPD_B1) dk += ch.B(eps, p.coef_b1, gn, d);
  if (p.flags & UPD_OA) ch.O(2, (double)p.hA * eps, L, d);
  double info_dk = 0.0, info_de = 0.0;
  if (p.flags & UPD_RECORD) {
    info_dk = dk;
    info_de = dk - logp_now + lold;
    dk = 0.0;
    lold = logp_now;
  }
  if (p.flags & UPD_OB) ch.O(3, (double)p.hB * eps, L, d);
  if (p.flags & UPD_B2) dk += ch.B(eps, p.coef_b2, gn, d);
  __syncthreads();  // all threads have read dK/lold/logp before thread 0 rewrites them
  if (tid == 0) {
    p.dK[e] = (float)dk;
    p.lold[e] = (float)lold;
    if (from_slabs) p.logp[e] = (float)logp_now;
    if ((p.flags & UPD_RECORD) && p.out_info) {
      p.out_info[3 * e + 0] = (float)logp_now;
      p.out_info[3 * e + 1] = (float)info_dk;
      p.out_info[3 * e + 2] = (float)info_de;
    }
  }
  const bool any_op = p.flags & (UPD_B1 | UPD_OA | UPD_OB | UPD_B2);
  if (!any_op && !(p.flags & UPD_A) && !p.out_sample) return;

  // ---- pass 2 ----------------------------------------------------------------------
  const float c0 = (float)ch.c[0], c1 = (float)(ch.c[1] / gn), c2 = (float)ch.c[2], c3 = (float)ch.c[3];
  const float ea = (float)(eps * (double)p.coef_a);
  for (int q = tid; q < nq; q += UPD_NT) {
    f32x4 za = {0, 0, 0, 0}, zb = {0, 0, 0, 0};
    if (useA && !p.zA) za = philox_normal4(q, pid, p.stepA, p.stageA, p.seed);
    if (useB && !p.zB) zb = philox_normal4(q, pid, p.stepB, p.stageB, p.seed);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int i = 4 * q + m;
      if (i < d) {
        const float sd = p.sdc ? p.sdc[base + i] : 1.0f;
        const float gs = p.g[base + i] * sd;
        const float a = useA ? (p.zA ? p.zA[base + i] : za[m]) : 0.0f;
        const float b = useB ? (p.zB ? p.zB[base + i] : zb[m]) : 0.0f;
        float v = p.u[base + i];
        if (any_op) {
          v = fmaf(c0, v, fmaf(c1, gs, fmaf(c2, a, c3 * b)));
          p.u[base + i] = v;
        }
        const float xi = p.x[base + i];
        if (p.out_sample) p.out_sample[base + i] = xi;
        if (p.flags & UPD_A) p.x[base + i] = fmaf(ea * sd, v, xi);
      }
    }
  }
}

// g = sum_s slab + grad log prior;  logp = sum_s llpart + log prior.   (mile_logpost_grad)
__global__ __launch_bounds__(UPD_NT) void k_finalize(int d, int S, int prior, float loc, float scale,
                                                     const float *theta, const float *slabs,
                                                     const float *llpart, float *grad, float *logp) {
  __shared__ float red[UPD_NT / 64];
  const int tid = threadIdx.x, e = blockIdx.x;
  const size_t base = (size_t)e * d;
  const float *sl = slabs + (size_t)e * S * d;
  float pv = 0.0f;
  for (int i = tid; i < d; i += UPD_NT) {
    float gi = 0.0f;
    for (int s = 0; s < S; ++s) gi += sl[(size_t)s * d + i];
    const float t = (theta[base + i] - loc) / scale;
    if (prior == MILE_PRIOR_NORMAL) { gi -= t / scale; pv += -0.5f * t * t; }
    else { gi -= (t > 0.0f ? 1.0f : (t < 0.0f ? -1.0f : 0.0f)) / scale; pv += -fabsf(t); }
    grad[base + i] = gi;
  }
  pv = wave_sum(pv);
  if ((tid & 63) == 0) red[tid >> 6] = pv;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int w = 0; w < UPD_NT / 64; ++w) t += (double)red[w];
    for (int s = 0; s < S; ++s) t += (double)llpart[(size_t)e * S + s];
    t += prior == MILE_PRIOR_NORMAL ? -(double)d * (log((double)scale) + 0.91893853320467274)
                                    : -(double)d * log(2.0 * (double)scale);
    logp[e] = (float)t;
  }
}

// momentum = z / |z|  (generate_unit_vector of blackjax.mcmc.mclmc.init, A.1)
__global__ __launch_bounds__(UPD_NT) void k_init_momentum(int d, const float *z, uint64_t seed,
                                                          const int32_t *pids, float *u) {
  __shared__ float red[UPD_NT / 64];
  const int tid = threadIdx.x, e = blockIdx.x;
  const size_t base = (size_t)e * d;
  const uint32_t pid = pids ? (uint32_t)pids[e] : (uint32_t)e;
  const int nq = (d + 3) >> 2;
  float ss = 0.0f;
  for (int q = tid; q < nq; q += UPD_NT) {
    f32x4 zz = {0, 0, 0, 0};
    if (!z) zz = philox_normal4(q, pid, 0u, 2u, seed);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int i = 4 * q + m;
      if (i < d) {
        const float v = z ? z[base + i] : zz[m];
        u[base + i] = v;
        ss = fmaf(v, v, ss);
      }
    }
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  float t = 0.0f;
  for (int w = 0; w < UPD_NT / 64; ++w) t += red[w];
  const float inv = 1.0f / sqrtf(t);
  for (int i = tid; i < d; i += UPD_NT) u[base + i] *= inv;
}

__global__ __launch_bounds__(UPD_NT) void k_debug_noise(int d, uint64_t seed, const int32_t *pids,
                                                        uint32_t step, uint32_t stage, float *out) {
  const int tid = threadIdx.x, e = blockIdx.x;
  const uint32_t pid = pids ? (uint32_t)pids[e] : (uint32_t)e;
  const int nq = (d + 3) >> 2;
  for (int q = tid; q < nq; q += UPD_NT) {
    const f32x4 zz = philox_normal4(q, pid, step, stage, seed);
    for (int m = 0; m < 4; ++m)
      if (4 * q + m < d) out[(size_t)e * d + 4 * q + m] = zz[m];
  }
}
