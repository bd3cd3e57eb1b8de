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
#include <cstdlib>

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
  UPD_TUNE = 1 << 8,        // record point also runs the warm-up tuner (k_update_fast only)
  UPD_NO_G = 1 << 9,        // mid-sequence launch: nobody reads state.g before the next gradient, skip its store (k_update_fast only)
};

struct UpdParams {
  int32_t d, E, S, flags;
  int32_t dp;                    // slab row stride (floats)
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
  // separate inputs (ping-pong state): NULL -> read from x/u/g/logp (in place)
  const float *x_in, *u_in, *g_in, *logp_in;
  // ---- warm-up tuner (UPD_TUNE): make_L_step_size_adaptation, src/training/warmup.py:271-350 ----
  float *t_eps;                  // [E] step size, rewritten at the record point
  float *t_eps_max, *t_time, *t_xavg;   // [E] adaptive state (step_size_max, time, x_average)
  float *t_W;                    // [E] streaming-average weight
  float *t_avg;                  // [E, 2, d] streaming averages of x and x^2
  const float *bk_x, *bk_u, *bk_g, *bk_logp;   // state before this kernel step (handle_nans reverts to it)
  float t_mask;                  // 1 during tune1 (no averaging), 0 during tune2
  float t_var;                   // desired energy variance at this step
  float t_trust, t_decay;
  float *upart;                  // k_update_seg: [E][segments][UPD_NSUM] partial sums + [E][8] staged scalars, or NULL
  // ---- merged warm-up launch (UPD_TUNE together with the NEXT step's O, B, A; k_update_fast only) ----
  // x_in / u_in name the buffer W the step ran in; the ACCEPTED state stays there (x_acc = W.x, u_rec = W.u, g, logp), and the
  // next step's working position / momentum go to x / u (the other buffer, which bk_* names: the state before this step).
  float *u_rec;                  // [E, d] momentum at the record point (nan_to_num'd), or NULL: plain record launch
  float *x_acc;                  // [E, d] == x_in, writable: a rejected chain gets bk_x back here
  int32_t force_restart;         // test hook (MILE_TUNE_FORCE_RESTART): every chain takes the restart path of the merged launch
};

// B / O chain on the coefficients of {u, e, zA, zB}; norms and projections come from the
// Gram matrix M.  fp32 with expm1f/log1pf: the inputs (dot products) are fp32 sums anyway
// and every cancellation-prone expression is written in its stable form.
struct Chain {
  float M[4][4];
  float c[4];
  __device__ __forceinline__ float norm() const {
    float s = 0.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) s = fmaf(c[a] * M[a][b], c[b], s);
    return sqrtf(s);
  }
  // esh_dynamics_momentum_update_one_step (A.2); returns the kinetic energy change
  __device__ __forceinline__ float B(float eps, float coef, float gnorm, int d) {
    float ue = 0.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a) ue = fmaf(c[a], M[a][1], ue);
    const float delta = eps * coef * gnorm / (float)(d - 1);
    const float omz = -expm1f(-delta);            // 1 - zeta
    const float zeta = 1.0f - omz;
    const float beta = omz * (1.0f + zeta + ue * omz);
#pragma unroll
    for (int a = 0; a < 4; ++a) c[a] *= 2.0f * zeta;
    c[1] += beta;
    const float n = norm();
    if (n > 1e-13f)                               // blackjax normalized_flatten_array tolerance
#pragma unroll
      for (int a = 0; a < 4; ++a) c[a] /= n;
    // (d-1) (delta - ln2 + ln(1 + ue + (1-ue) zeta^2)) without the cancellation
    return (float)(d - 1) * (delta + log1pf(0.5f * (1.0f - ue) * expm1f(-2.0f * delta)));
  }
  // partially_refresh_momentum (A.5)
  __device__ __forceinline__ void O(int k, float hstep, float L, int d) {
    const float nu = sqrtf(expm1f(2.0f * hstep / L) / (float)d);
    c[k] += nu;
    const float n = norm();
#pragma unroll
    for (int a = 0; a < 4; ++a) c[a] /= n;
  }
};


// blackjax normalized_flatten_array(x, tol = 1e-13) -> (x / |x| if |x| > tol else x, |x|), applied to the (preconditioned)
// gradient: the SAME guarded helper serves the gradient and the new momentum in esh_dynamics_momentum_update_one_step, so a zero
// gradient -- e.g. nan_to_num of a NaN one in an accepted warm-up state (src/training/warmup.py:478-482) -- gives e = g = 0,
// delta = 0 and leaves the momentum alone (VERDICT r2 weak #1b; rounds 1-2 used |g| = 1 there).  gn: |g~| as jnp.linalg.norm
// returns it (NaN and inf propagate); ign: the factor that turns g~ into e; ee = e.e.  An infinite norm divides every finite
// entry to 0 (e = 0, all projections 0) -- the projections are set to 0 directly because (sum u.g~) * 0 would be NaN once that
// sum itself overflowed.
struct GradNorm {
  float gn, ign, ee;
  bool zero_e;
};
__device__ __forceinline__ GradNorm grad_norm(float gg) {
  GradNorm r;
  r.gn = sqrtf(gg);
  const bool ok = r.gn > 1e-13f;                 // false for NaN
  r.zero_e = ok && isinf(r.gn);
  r.ign = ok ? 1.0f / r.gn : 1.0f;               // 1 / inf = 0
  r.ee = ok ? (r.zero_e ? 0.0f : 1.0f) : gg;
  return r;
}
// Gram matrix of {u, e, zA, zB} from the reduced sums S[0..9] (u.u, u.g~, g~.g~, u.zA, g~.zA, zA.zA, u.zB, g~.zB, zB.zB, zA.zB)
__device__ __forceinline__ void chain_gram(Chain &ch, const float *S, const GradNorm &gnm) {
  const float ig = gnm.ign;
  ch.M[0][0] = S[0]; ch.M[0][1] = gnm.zero_e ? 0.0f : S[1] * ig; ch.M[0][2] = S[3]; ch.M[0][3] = S[6];
  ch.M[1][1] = gnm.ee; ch.M[1][2] = gnm.zero_e ? 0.0f : S[4] * ig; ch.M[1][3] = gnm.zero_e ? 0.0f : S[7] * ig;
  ch.M[2][2] = S[5]; ch.M[2][3] = S[9];
  ch.M[3][3] = S[8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (b < a) ch.M[a][b] = ch.M[b][a];
  ch.c[0] = 1.0f; ch.c[1] = ch.c[2] = ch.c[3] = 0.0f;
}
// jnp.nan_to_num: NaN -> 0, +-inf -> +-FLT_MAX (handle_nans applies it to every leaf of an accepted state)
__device__ __forceinline__ float nan_to_num(float v) {
  constexpr float FMAX = 3.4028234663852886e38f;
  return isnan(v) ? 0.0f : fminf(fmaxf(v, -FMAX), FMAX);
}

#define UPD_NT 1024
#define UPD_NW (UPD_NT / 64)
#define UPD_NSUM 12   // 10 dot products, log-prior, non-finite count
#define UPD_QMAX 4   // quads a thread keeps in registers between the passes (d <= 16384)

// CACHED: every thread keeps its <= UPD_QMAX quads of (x, u, g~, zA, zB) in registers between
// pass 1 and pass 2, so state is read once, noise is generated once and written once.
template <bool CACHED>
static __global__ __launch_bounds__(UPD_NT) void k_update(const UpdParams p) {
  __shared__ float red[UPD_NW][UPD_NSUM + 1];
  __shared__ float tot[UPD_NSUM + 1];
  const int tid = threadIdx.x, e = blockIdx.x, d = p.d;
  const size_t base = (size_t)e * d;
  const int nq = (d + 3) >> 2;
  const bool from_slabs = p.flags & UPD_FROM_SLABS;
  const bool useA = p.flags & UPD_OA, useB = p.flags & UPD_OB;
  const uint32_t pid = p.pids ? (uint32_t)p.pids[e] : (uint32_t)e;
  const float *sl = p.slabs + (size_t)e * p.S * p.dp;
  constexpr int NK = CACHED ? UPD_QMAX : 1;
  f32x4 cx[NK], cu[NK], cg[NK], ca[NK], cb[NK];

  // ---- pass 1 ----------------------------------------------------------------------
  float sm[UPD_NSUM];
#pragma unroll
  for (int k = 0; k < UPD_NSUM; ++k) sm[k] = 0.0f;
  auto pass1 = [&](int q, f32x4 &X, f32x4 &U, f32x4 &G, f32x4 &A, f32x4 &B) {
    A = f32x4{0, 0, 0, 0}; B = f32x4{0, 0, 0, 0};
    if (useA && !p.zA) A = philox_normal4(q, pid, p.stepA, p.stageA, p.seed);
    if (useB && !p.zB) B = philox_normal4(q, pid, p.stepB, p.stageB, p.seed);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int i = 4 * q + m;
      X[m] = U[m] = G[m] = 0.0f;
      if (i < d) {
        float gi;
        const float xi = p.x[base + i];
        if (from_slabs) {
          gi = 0.0f;
          for (int s = 0; s < p.S; ++s) gi += sl[(size_t)s * p.dp + i];
          const float t = (xi - p.prior_loc) / p.prior_scale;
          if (p.prior == MILE_PRIOR_NORMAL) {
            gi -= t / p.prior_scale;
            sm[10] = fmaf(-0.5f * t, t, sm[10]);
          } else {
            gi -= (t > 0.0f ? 1.0f : (t < 0.0f ? -1.0f : 0.0f)) / p.prior_scale;
            sm[10] -= fabsf(t);
          }
          p.g[base + i] = gi;
        } else {
          gi = p.g[base + i];
        }
        const float gs = p.sdc ? gi * p.sdc[base + i] : gi;
        const float ui = p.u[base + i];
        if (useA && p.zA) A[m] = p.zA[base + i];
        if (useB && p.zB) B[m] = p.zB[base + i];
        const float a = A[m], b = B[m];
        X[m] = xi; U[m] = ui; G[m] = gs;
        sm[0] = fmaf(ui, ui, sm[0]); sm[1] = fmaf(ui, gs, sm[1]); sm[2] = fmaf(gs, gs, sm[2]);
        sm[3] = fmaf(ui, a, sm[3]); sm[4] = fmaf(gs, a, sm[4]); sm[5] = fmaf(a, a, sm[5]);
        sm[6] = fmaf(ui, b, sm[6]); sm[7] = fmaf(gs, b, sm[7]); sm[8] = fmaf(b, b, sm[8]);
        sm[9] = fmaf(a, b, sm[9]);
      } else {
        A[m] = B[m] = 0.0f;
      }
    }
  };
  if (CACHED) {
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int q = tid + k * UPD_NT;
      if (q < nq) pass1(q, cx[k], cu[k], cg[k], ca[k], cb[k]);
    }
  } else {
    for (int q = tid; q < nq; q += UPD_NT) pass1(q, cx[0], cu[0], cg[0], ca[0], cb[0]);
  }
#pragma unroll
  for (int k = 0; k < UPD_NSUM; ++k) sm[k] = wave_sum(sm[k]);
  if ((tid & 63) == 0)
#pragma unroll
    for (int k = 0; k < UPD_NSUM; ++k) red[tid >> 6][k] = sm[k];
  __syncthreads();
  if (tid < UPD_NSUM) {
    float t = 0.0f;
#pragma unroll
    for (int w = 0; w < UPD_NW; ++w) t += red[w][tid];
    tot[tid] = t;
  }
  __syncthreads();

  // ---- scalar chain (every thread, redundantly) -------------------------------------
  float S[UPD_NSUM];
#pragma unroll
  for (int k = 0; k < UPD_NSUM; ++k) S[k] = tot[k];
  const float eps = p.eps[e], L = p.L[e];
  float logp_now;
  if (from_slabs) {
    float ll = 0.0f;
    for (int s = 0; s < p.S; ++s) ll += p.llpart[(size_t)e * p.S + s];
    const float cst = p.prior == MILE_PRIOR_NORMAL ? -(float)d * (logf(p.prior_scale) + 0.91893853320467274f)
                                                   : -(float)d * logf(2.0f * p.prior_scale);
    logp_now = ll + (S[10] + cst);
  } else {
    logp_now = p.logp[e];
  }
  const GradNorm gnm = grad_norm(S[2]);
  const float gn = gnm.gn, ign = gnm.ign;
  Chain ch;
  chain_gram(ch, S, gnm);

  float dk = (p.flags & UPD_START) ? 0.0f : p.dK[e];
  float lold = (p.flags & UPD_START) ? logp_now : p.lold[e];
  if (p.flags & UPD_B1) dk += ch.B(eps, p.coef_b1, gn, d);
  if (p.flags & UPD_OA) ch.O(2, p.hA * eps, L, d);
  float info_dk = 0.0f, info_de = 0.0f;
  if (p.flags & UPD_RECORD) {
    info_dk = dk;
    info_de = dk - (logp_now - lold);
    dk = 0.0f;
    lold = logp_now;
  }
  if (p.flags & UPD_OB) ch.O(3, p.hB * eps, L, d);
  if (p.flags & UPD_B2) dk += ch.B(eps, p.coef_b2, gn, d);
  __syncthreads();  // all threads have read dK/lold/logp before thread 0 rewrites them
  if (tid == 0) {
    p.dK[e] = dk;
    p.lold[e] = lold;
    if (from_slabs) p.logp[e] = logp_now;
    if ((p.flags & UPD_RECORD) && p.out_info) {
      p.out_info[3 * e + 0] = logp_now;
      p.out_info[3 * e + 1] = info_dk;
      p.out_info[3 * e + 2] = info_de;
    }
  }
  const bool any_op = p.flags & (UPD_B1 | UPD_OA | UPD_OB | UPD_B2);
  if (!any_op && !(p.flags & UPD_A) && !p.out_sample) return;

  // ---- pass 2 ----------------------------------------------------------------------
  const float c0 = ch.c[0], c1 = ch.c[1] * ign, c2 = ch.c[2], c3 = ch.c[3];
  const float ea = eps * p.coef_a;
  auto pass2 = [&](int q, const f32x4 &X, const f32x4 &U, const f32x4 &G, const f32x4 &A, const f32x4 &B) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int i = 4 * q + m;
      if (i < d) {
        const float sd = p.sdc ? p.sdc[base + i] : 1.0f;
        float v = U[m];
        if (any_op) {
          v = fmaf(c0, v, fmaf(c1, G[m], fmaf(c2, A[m], c3 * B[m])));
          p.u[base + i] = v;
        }
        if (p.out_sample) p.out_sample[base + i] = X[m];
        if (p.flags & UPD_A) p.x[base + i] = fmaf(ea * sd, v, X[m]);
      }
    }
  };
  if (CACHED) {
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int q = tid + k * UPD_NT;
      if (q < nq) pass2(q, cx[k], cu[k], cg[k], ca[k], cb[k]);
    }
  } else {
    for (int q = tid; q < nq; q += UPD_NT) {
      f32x4 X, U, G, A = {0, 0, 0, 0}, B = {0, 0, 0, 0};
      if (useA && !p.zA) A = philox_normal4(q, pid, p.stepA, p.stageA, p.seed);
      if (useB && !p.zB) B = philox_normal4(q, pid, p.stepB, p.stageB, p.seed);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int i = 4 * q + m;
        X[m] = U[m] = G[m] = 0.0f;
        if (i < d) {
          X[m] = p.x[base + i]; U[m] = p.u[base + i];
          G[m] = p.g[base + i] * (p.sdc ? p.sdc[base + i] : 1.0f);
          if (useA && p.zA) A[m] = p.zA[base + i];
          if (useB && p.zB) B[m] = p.zB[base + i];
        }
      }
      pass2(q, X, U, G, A, B);
    }
  }
}

// ---------------------------------------------------------------------------------------
// k_update for any d on all CUs: a particle's vector is cut into segments of UPD_SEG elements, one workgroup each.
//   PHASE 1 (k_update_seg<1>): pass 1 of k_update on the segment (sums, g from the slabs), the segment's 11 partial sums to upart;
//   PHASE 2 (k_update_seg<2>): every workgroup adds the particle's partials in segment order (deterministic), runs the scalar
//   chain (segment 0 writes the per-particle scalars) and applies pass 2 to its segment, state re-read and noise regenerated as
//   in the uncached k_update.  Same arithmetic per element as k_update<false>; only the order of the 11 sums differs.
// d > 4 * UPD_NT * UPD_QMAX_BIG (B4: 213 255, LeNet: 83 126); with E = 32 particles per GPU (B5) the one-workgroup-per-particle
// form left 7/8 of the chip idle through its serial 4-byte loops.
#define UPD_SEG 8192
template <int PHASE>
static __global__ __launch_bounds__(UPD_NT) void k_update_seg(const UpdParams p, float *upart, int nseg) {
  __shared__ float red[UPD_NW][UPD_NSUM + 1];
  __shared__ float tot[UPD_NSUM + 1];
  const int tid = threadIdx.x, seg = blockIdx.x, e = blockIdx.y, d = p.d;
  const size_t base = (size_t)e * d;
  const int q0 = seg * (UPD_SEG / 4), q1 = min((d + 3) >> 2, q0 + UPD_SEG / 4);
  const bool from_slabs = p.flags & UPD_FROM_SLABS;
  const bool useA = p.flags & UPD_OA, useB = p.flags & UPD_OB;
  const uint32_t pid = p.pids ? (uint32_t)p.pids[e] : (uint32_t)e;
  const float *sl = p.slabs + (size_t)e * p.S * p.dp;
  float *mine = upart + ((size_t)e * nseg + seg) * UPD_NSUM;
  if constexpr (PHASE == 1) {
    float sm[UPD_NSUM];
#pragma unroll
    for (int k = 0; k < UPD_NSUM; ++k) sm[k] = 0.0f;
    for (int q = q0 + tid; q < q1; q += UPD_NT) {
      f32x4 A = {0, 0, 0, 0}, B = {0, 0, 0, 0};
      if (useA && !p.zA) A = philox_normal4(q, pid, p.stepA, p.stageA, p.seed);
      if (useB && !p.zB) B = philox_normal4(q, pid, p.stepB, p.stageB, p.seed);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int i = 4 * q + m;
        if (i < d) {
          float gi;
          const float xi = p.x[base + i];
          if (from_slabs) {
            gi = 0.0f;
            for (int s = 0; s < p.S; ++s) gi += sl[(size_t)s * p.dp + i];
            const float t = (xi - p.prior_loc) / p.prior_scale;
            if (p.prior == MILE_PRIOR_NORMAL) {
              gi -= t / p.prior_scale;
              sm[10] = fmaf(-0.5f * t, t, sm[10]);
            } else {
              gi -= (t > 0.0f ? 1.0f : (t < 0.0f ? -1.0f : 0.0f)) / p.prior_scale;
              sm[10] -= fabsf(t);
            }
            p.g[base + i] = gi;
          } else {
            gi = p.g[base + i];
          }
          const float gs = p.sdc ? gi * p.sdc[base + i] : gi;
          const float ui = p.u[base + i];
          const float a = (useA && p.zA) ? p.zA[base + i] : A[m], b = (useB && p.zB) ? p.zB[base + i] : B[m];
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
    if (tid < UPD_NSUM) {
      float t = 0.0f;
#pragma unroll
      for (int w = 0; w < UPD_NW; ++w) t += red[w][tid];
      mine[tid] = t;
    }
    return;
  } else {
    if (tid < UPD_NSUM) {
      float t = 0.0f;
      for (int sg = 0; sg < nseg; ++sg) t += upart[((size_t)e * nseg + sg) * UPD_NSUM + tid];
      tot[tid] = t;
    }
    __syncthreads();
    float S[UPD_NSUM];
#pragma unroll
    for (int k = 0; k < UPD_NSUM; ++k) S[k] = tot[k];
    const float eps = p.eps[e], L = p.L[e];
    float logp_now;
    if (from_slabs) {
      float ll = 0.0f;
      for (int s = 0; s < p.S; ++s) ll += p.llpart[(size_t)e * p.S + s];
      const float cst = p.prior == MILE_PRIOR_NORMAL ? -(float)d * (logf(p.prior_scale) + 0.91893853320467274f)
                                                     : -(float)d * logf(2.0f * p.prior_scale);
      logp_now = ll + (S[10] + cst);
    } else {
      logp_now = p.logp[e];
    }
    const GradNorm gnm = grad_norm(S[2]);
    const float gn = gnm.gn, ign = gnm.ign;
    Chain ch;
    chain_gram(ch, S, gnm);
    float dk = (p.flags & UPD_START) ? 0.0f : p.dK[e];
    float lold = (p.flags & UPD_START) ? logp_now : p.lold[e];
    if (p.flags & UPD_B1) dk += ch.B(eps, p.coef_b1, gn, d);
    if (p.flags & UPD_OA) ch.O(2, p.hA * eps, L, d);
    float info_dk = 0.0f, info_de = 0.0f;
    if (p.flags & UPD_RECORD) {
      info_dk = dk;
      info_de = dk - (logp_now - lold);
      dk = 0.0f;
      lold = logp_now;
    }
    if (p.flags & UPD_OB) ch.O(3, p.hB * eps, L, d);
    if (p.flags & UPD_B2) dk += ch.B(eps, p.coef_b2, gn, d);
    // the per-particle scalars are rewritten by k_update_seg_scalars after every segment has read them
    const bool any_op = p.flags & (UPD_B1 | UPD_OA | UPD_OB | UPD_B2);
    if (seg == 0 && tid == 0) {   // staged: dK / lold / logp must stay readable for the other segments of this launch
      float *st = upart + ((size_t)gridDim.y * nseg) * UPD_NSUM + (size_t)e * 8;
      st[0] = dk; st[1] = lold; st[2] = logp_now; st[3] = info_dk; st[4] = info_de;
    }
    if (!any_op && !(p.flags & UPD_A) && !p.out_sample) return;
    const float c0 = ch.c[0], c1 = ch.c[1] * ign, c2 = ch.c[2], c3 = ch.c[3];
    const float ea = eps * p.coef_a;
    for (int q = q0 + tid; q < q1; q += UPD_NT) {
      f32x4 A = {0, 0, 0, 0}, B = {0, 0, 0, 0};
      if (useA && !p.zA) A = philox_normal4(q, pid, p.stepA, p.stageA, p.seed);
      if (useB && !p.zB) B = philox_normal4(q, pid, p.stepB, p.stageB, p.seed);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int i = 4 * q + m;
        if (i < d) {
          const float sd = p.sdc ? p.sdc[base + i] : 1.0f;
          const float X = p.x[base + i], G = p.g[base + i] * sd;
          const float a = (useA && p.zA) ? p.zA[base + i] : A[m], b = (useB && p.zB) ? p.zB[base + i] : B[m];
          float v = p.u[base + i];
          if (any_op) {
            v = fmaf(c0, v, fmaf(c1, G, fmaf(c2, a, c3 * b)));
            p.u[base + i] = v;
          }
          if (p.out_sample) p.out_sample[base + i] = X;
          if (p.flags & UPD_A) p.x[base + i] = fmaf(ea * sd, v, X);
        }
      }
    }
  }
}
// the per-particle scalars of a segmented update, from the values segment 0 staged (after every segment has used the old ones)
static __global__ __launch_bounds__(256) void k_update_seg_scalars(const UpdParams p, const float *upart, int nseg, int E) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const float *st = upart + ((size_t)E * nseg) * UPD_NSUM + (size_t)e * 8;
  p.dK[e] = st[0];
  p.lold[e] = st[1];
  if (p.flags & UPD_FROM_SLABS) p.logp[e] = st[2];
  if ((p.flags & UPD_RECORD) && p.out_info) {
    p.out_info[3 * e + 0] = st[2];
    p.out_info[3 * e + 1] = st[3];
    p.out_info[3 * e + 2] = st[4];
  }
}

// ---------------------------------------------------------------------------------------
// Fast path of k_update for d <= 4*UPD_NT*UPD_QMAX: branch-free, all loads of a thread are
// issued back to back (clamped addresses + masks instead of bounds branches) as AL-float
// vectors, state and noise stay in registers between the two passes.
// NK = quads per thread, AL = guaranteed alignment (floats) of every row base.
// ---------------------------------------------------------------------------------------
template <int AL>
__device__ __forceinline__ f32x4 ld4(const float *q) {
  if constexpr (AL == 4) {
    return *(const f32x4 *)q;
  } else if constexpr (AL == 2) {
    const f32x2 a = *(const f32x2 *)q, b = *(const f32x2 *)(q + 2);
    return f32x4{a[0], a[1], b[0], b[1]};
  } else {
    return f32x4{q[0], q[1], q[2], q[3]};
  }
}
template <int AL>
__device__ __forceinline__ void st4(float *q, const f32x4 v) {
  if constexpr (AL == 4) {
    *(f32x4 *)q = v;
  } else if constexpr (AL == 2) {
    *(f32x2 *)q = f32x2{v[0], v[1]};
    *(f32x2 *)(q + 2) = f32x2{v[2], v[3]};
  } else {
    q[0] = v[0]; q[1] = v[1]; q[2] = v[2]; q[3] = v[3];
  }
}

// Slab / llpart reads.  COH = the update runs as the EPILOGUE of the grad launch that produced the slabs (last-arriving
// workgroup of the particle, mile_grad_w64.h): every byte another workgroup of this launch wrote is read with an sc1 load
// (L1 bypassed; the producer stored it sc1 and drained) -- /opt/skills/guides cdna_hip_programming.md section 6 Guideline 16,
// the counter form of the hand-off.  Slab rows are 16-byte aligned whatever AL says (dp % 4 == 0).
typedef uint32_t upd_u32x4 __attribute__((ext_vector_type(4)));
template <int AL, bool COH>
__device__ __forceinline__ f32x4 ld4_slab(const float *row, __amdgpu_buffer_rsrc_t rs, int float_off) {
  if constexpr (COH) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, float_off * 4, 0, 16));   // aux 16 = sc1
  } else {
    return ld4<AL>(row + float_off);
  }
}
template <bool COH>
__device__ __forceinline__ float ld1_shared(const float *q) {
  if constexpr (COH) return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *q;
}

// The body of k_update_fast for particle e, run by `nt` threads (a multiple of 64, nt * NK >= d / 4) with thread index tid;
// red / bc: LDS scratch of the caller ([nt / 64][UPD_NSUM + 1] and [8] floats).
// CF >= 0: the launch kind is known at compile time -- the flag word is the constant CF and the prior is Normal -- so every
// flag test folds and the per-element code is branch-free (the steady state of a sampling run is two kinds, UPD_KIND_MID and
// UPD_KIND_REC).  CF < 0: flags and prior kind are read from p at run time (first / last launches of a call, tuner, Laplace
// prior, step-O refresh).
#define UPD_KIND_MID (UPD_FROM_SLABS | UPD_B1 | UPD_A | UPD_NO_G)
#define UPD_KIND_REC (UPD_FROM_SLABS | UPD_B1 | UPD_OA | UPD_RECORD | UPD_OB | UPD_B2 | UPD_A | UPD_NO_G)
#define UPD_KIND_TUNE (UPD_FROM_SLABS | UPD_B1 | UPD_OA | UPD_RECORD | UPD_TUNE | UPD_OB | UPD_B2 | UPD_A)   // warm-up steady state
// BIG (d beyond UPD_QMAX quads per thread, up to UPD_QMAX_BIG): only u stays in registers, x is read twice, the gradient waits in the
// workgroup's LDS array `gl` ([NK][nt] quads, <= 147 KB) and the O-step noise is generated twice (pass 1 for the sums, pass 2
// for the update) -- five register-resident arrays of 9 quads do not fit the 128 registers a 1024-thread workgroup has.
// Returns true (workgroup-uniform) when a merged warm-up launch could not take the next step's O, B, A along -- the step was
// rejected, or the accepted state needed nan_to_num -- and the caller has to run them from the stored state (upd_tune_restart).
template <int NK, int AL, bool SDC, bool COH, int CF = -1, bool BIG = false>
__device__ __forceinline__ bool upd_fast_body(const UpdParams &p, const int e, const int tid, const int nt,
                                              float (*red)[UPD_NSUM + 1], float *bc, long long *stamps = nullptr,
                                              f32x4 *gl = nullptr) {
  const int d = p.d;
  const int flags = CF >= 0 ? CF : p.flags;
  const size_t base = (size_t)e * d;
  const int nqf = d >> 2;            // full quads
  const int ntail = d & 3;           // leftover elements, handled by threads 0..ntail-1
  const bool from_slabs = flags & UPD_FROM_SLABS;
  const bool useA = flags & UPD_OA, useB = flags & UPD_OB;
  const bool explA = useA && p.zA, explB = useB && p.zB;
  const uint32_t pid = p.pids ? (uint32_t)p.pids[e] : (uint32_t)e;
  const float *sl = p.slabs + (size_t)e * p.S * p.dp;
  const __amdgpu_buffer_rsrc_t sl_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(sl), 0, p.S * p.dp * 4, 0x00020000);
  const float *xin = p.x_in ? p.x_in : p.x, *uin = p.u_in ? p.u_in : p.u, *gin = p.g_in ? p.g_in : p.g;
  const bool tune = flags & UPD_TUNE;
  const bool merged = tune && p.u_rec != nullptr;      // record point + tuner + the next step's O, B, A in one launch
  const bool store_g = from_slabs && !(flags & UPD_NO_G);
  const float ips = 1.0f / p.prior_scale;
  const bool normal = CF >= 0 ? true : p.prior == MILE_PRIOR_NORMAL;

  // per-particle scalars, fetched up front so their latency hides under the vector loads
  const float eps_in = p.eps[e], L_in = p.L[e];
  const float dk_in = (flags & UPD_START) ? 0.0f : p.dK[e];
  const float lold_in = (flags & UPD_START) ? 0.0f : p.lold[e];
  const float logp_in = from_slabs ? 0.0f : (p.logp_in ? p.logp_in : p.logp)[e];
  float ll_in = 0.0f;
  if (from_slabs) {
#pragma unroll 8          // independent loads, several in flight: S is 2 for B2 but up to 64 for narrow nets with few chains
    for (int s = 0; s < p.S; ++s) ll_in += ld1_shared<COH>(p.llpart + (size_t)e * p.S + s);
  }

  constexpr int NR = BIG ? 1 : NK;   // quads of g / noise a thread keeps in registers
  f32x4 cx[BIG ? 1 : NK], cu[NK], cg[NR], ca[NR], cb[NR], csd[SDC ? NK : 1];
  // ---- pass 1: loads -----------------------------------------------------------------
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int q = tid + k * nt;
    const bool valid = q < nqf;
    const size_t o = base + 4 * (size_t)(valid ? q : 0);
    if constexpr (!BIG) cx[k] = ld4<AL>(xin + o);
    cu[k] = ld4<AL>(uin + o);
    f32x4 g;
    if (from_slabs) {
      const int so = (int)(o - base);
      g = ld4_slab<AL, COH>(sl, sl_rs, so);
#pragma unroll 4
      for (int s = 1; s < p.S; ++s) g += ld4_slab<AL, COH>(sl, sl_rs, s * p.dp + so);
    } else {
      g = ld4<AL>(gin + o);
    }
    if constexpr (BIG) gl[k * nt + tid] = g;
    else cg[k] = g;
    if constexpr (SDC) csd[k] = ld4<AL>(p.sdc + o);
    if constexpr (!BIG) {
      ca[k] = explA ? ld4<AL>(p.zA + o) : f32x4{0, 0, 0, 0};
      cb[k] = explB ? ld4<AL>(p.zB + o) : f32x4{0, 0, 0, 0};
    }
  }
  // tail element (at most 3 per particle): thread t < ntail owns element 4*nqf + t
  const bool has_tail = tid < ntail;
  const size_t to = base + 4 * (size_t)nqf + (has_tail ? tid : 0);
  float tx = 0, tu = 0, tg = 0, ta = 0, tb = 0, tsd = 1.0f;
  if (ntail) {
    const size_t tc = has_tail ? to : base;
    tx = xin[tc]; tu = uin[tc];
    if (from_slabs) {
      tg = 0.0f;
#pragma unroll 4
      for (int s = 0; s < p.S; ++s) tg += ld1_shared<COH>(sl + (size_t)s * p.dp + (tc - base));
    }
    else tg = gin[tc];
    if (SDC) tsd = p.sdc[tc];
    if (explA) ta = p.zA[tc];
    if (explB) tb = p.zB[tc];
  }
  if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamps[0] = wall_clock64(); }   // dev: loads landed
  // ---- pass 1: noise + sums ------------------------------------------------------------
  float sm[UPD_NSUM];
#pragma unroll
  for (int k = 0; k < UPD_NSUM; ++k) sm[k] = 0.0f;
#pragma unroll
  for (int kq = 0; kq < NK; ++kq) {
    const int q = tid + kq * nt;
    const int k = BIG ? 0 : kq;          // register slot of g / noise
    if constexpr (BIG) {
      // x is read again in pass 2 instead of held (MALL-resident; spills cost more).  No explicit-noise hooks in this form.
      cx[0] = ld4<AL>(xin + base + (unsigned)(4 * (q < nqf ? q : 0)));
      cg[0] = gl[kq * nt + tid];
      ca[0] = f32x4{0, 0, 0, 0};
      cb[0] = f32x4{0, 0, 0, 0};
    }
    if (useA && !p.zA) ca[k] = philox_normal4(q, pid, p.stepA, p.stageA, p.seed);
    if (useB && !p.zB) cb[k] = philox_normal4(q, pid, p.stepB, p.stageB, p.seed);
    const float mk = q < nqf ? 1.0f : 0.0f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float gi = cg[k][m];
      const float xi = cx[k][m];
      if (from_slabs) {
        const float t = (xi - p.prior_loc) * ips;
        if (normal) { gi = fmaf(-t, ips, gi); sm[10] = fmaf(-0.5f * mk * t, t, sm[10]); }
        else { gi -= (t > 0.0f ? ips : (t < 0.0f ? -ips : 0.0f)); sm[10] -= mk * fabsf(t); }
      }
      cg[k][m] = gi;                       // un-preconditioned gradient (stored below)
      if (tune) sm[11] += (mk != 0.0f && !isfinite(xi)) ? 1.0f : 0.0f;
      const float gs = (SDC ? gi * csd[kq][m] : gi) * mk;
      const float ui = cu[kq][m] * mk, a = ca[k][m] * mk, b = cb[k][m] * mk;
      ca[k][m] = a; cb[k][m] = b;
      sm[0] = fmaf(ui, ui, sm[0]); sm[1] = fmaf(ui, gs, sm[1]); sm[2] = fmaf(gs, gs, sm[2]);
      sm[3] = fmaf(ui, a, sm[3]); sm[4] = fmaf(gs, a, sm[4]); sm[5] = fmaf(a, a, sm[5]);
      sm[6] = fmaf(ui, b, sm[6]); sm[7] = fmaf(gs, b, sm[7]); sm[8] = fmaf(b, b, sm[8]);
      sm[9] = fmaf(a, b, sm[9]);
    }
    if (store_g && q < nqf) {
      f32x4 gst = cg[k];
      if (tune) {   // handle_nans: nan_to_num on every leaf of an accepted state (a rejected chain's g is rewritten below)
#pragma unroll
        for (int m = 0; m < 4; ++m) gst[m] = nan_to_num(gst[m]);
      }
      st4<AL>(p.g + base + 4 * (size_t)q, gst);
    }
    if constexpr (SDC) {
#pragma unroll
      for (int m = 0; m < 4; ++m) cg[k][m] *= csd[kq][m];   // keep g~ = g*s for pass 2
    }
    if constexpr (BIG) {
      gl[kq * nt + tid] = cg[0];
      __builtin_amdgcn_sched_barrier(0);   // one quad at a time: interleaved iterations spill (128 registers per thread)
    }
  }
  if (stamps) stamps[3] = wall_clock64();
  if (ntail) {
    f32x4 za = {0, 0, 0, 0}, zb = {0, 0, 0, 0};
    if (useA && !p.zA) za = philox_normal4(nqf, pid, p.stepA, p.stageA, p.seed);
    if (useB && !p.zB) zb = philox_normal4(nqf, pid, p.stepB, p.stageB, p.seed);
    const int t = has_tail ? tid : 0;
    if (useA && !p.zA) ta = za[t];
    if (useB && !p.zB) tb = zb[t];
    const float mk = has_tail ? 1.0f : 0.0f;
    if (from_slabs) {
      const float tt = (tx - p.prior_loc) * ips;
      if (normal) { tg = fmaf(-tt, ips, tg); sm[10] = fmaf(-0.5f * mk * tt, tt, sm[10]); }
      else { tg -= (tt > 0.0f ? ips : (tt < 0.0f ? -ips : 0.0f)); sm[10] -= mk * fabsf(tt); }
      if (has_tail && store_g) p.g[to] = tune ? nan_to_num(tg) : tg;
    }
    if (tune) sm[11] += (has_tail && !isfinite(tx)) ? 1.0f : 0.0f;
    tg *= tsd;
    const float gs = tg * mk, ui = tu * mk;
    ta *= mk; tb *= mk;
    sm[0] = fmaf(ui, ui, sm[0]); sm[1] = fmaf(ui, gs, sm[1]); sm[2] = fmaf(gs, gs, sm[2]);
    sm[3] = fmaf(ui, ta, sm[3]); sm[4] = fmaf(gs, ta, sm[4]); sm[5] = fmaf(ta, ta, sm[5]);
    sm[6] = fmaf(ui, tb, sm[6]); sm[7] = fmaf(gs, tb, sm[7]); sm[8] = fmaf(tb, tb, sm[8]);
    sm[9] = fmaf(ta, tb, sm[9]);
  }
  if (stamps) stamps[4] = wall_clock64();
#pragma unroll
  for (int k = 0; k < UPD_NSUM; ++k) sm[k] = wave_sum(sm[k]);
  if (stamps) stamps[5] = wall_clock64();
  if ((tid & 63) == 0)
#pragma unroll
    for (int k = 0; k < UPD_NSUM; ++k) red[tid >> 6][k] = sm[k];
  __syncthreads();
  if (stamps) stamps[1] = wall_clock64();                                                          // dev: sums reduced

  // ---- scalar chain: wave 0 only, coefficients broadcast through LDS ------------------------
  if (tid < 64) {
  float S[UPD_NSUM];
  {
    float t = 0.0f;
    const int kk = tid < UPD_NSUM ? tid : 0;
#pragma unroll
    for (int w = 0; w < (nt >> 6); ++w) t += red[w][kk];
#pragma unroll
    for (int k = 0; k < UPD_NSUM; ++k) S[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), k));
  }
  const float eps = eps_in, L = L_in;
  float logp_now;
  if (from_slabs) {
    const float ll = ll_in;
    const float cst = normal ? -(float)d * (logf(p.prior_scale) + 0.91893853320467274f)
                             : -(float)d * logf(2.0f * p.prior_scale);
    logp_now = ll + (S[10] + cst);
  } else {
    logp_now = logp_in;
  }
  const GradNorm gnm = grad_norm(S[2]);
  const float gn = gnm.gn, ign = gnm.ign;
  Chain ch;
  chain_gram(ch, S, gnm);
  float dk = (flags & UPD_START) ? 0.0f : dk_in;
  float lold = (flags & UPD_START) ? logp_now : lold_in;
  if (flags & UPD_B1) dk += ch.B(eps, p.coef_b1, gn, d);
  if (flags & UPD_OA) ch.O(2, p.hA * eps, L, d);
  float info_dk = 0.0f, info_de = 0.0f;
  if (flags & UPD_RECORD) {
    info_dk = dk;
    info_de = dk - (logp_now - lold);
    dk = 0.0f;
    lold = logp_now;
  }
  // ---- warm-up tuner at the record point (predictor + handle_nans, warmup.py:271-326,468-483) ----
  bool t_ok = true;
  float t_wgt = 0.0f, t_Wold = 0.0f;
  float eps_next = eps;                                     // the step size the ops after the record point run with
  const float r0 = ch.c[0], r1 = ch.c[1] * ign, r2 = ch.c[2], r3 = ch.c[3];   // momentum at the record point
  bool t_simple = true;
  if (tune) {
    constexpr float FMAX = 3.4028234663852886e38f;
    t_ok = S[11] == 0.0f;                                   // jnp.all(jnp.isfinite(position))
    float dE = info_de;
    dE = t_ok ? (isnan(dE) ? 0.0f : fminf(fmaxf(dE, -FMAX), FMAX)) : 0.0f;          // nan_to_num / 0.0
    const float emax_old = p.t_eps_max[e];
    const float emax = t_ok ? nan_to_num(emax_old) : 0.8f * eps;                    // nan_to_num(inf) = FLT_MAX, (NaN) = 0
    const float xi = dE * dE / ((float)d * p.t_var) + 1e-8f;
    const float lx = logf(xi) / (6.0f * p.t_trust);
    const float w = expf(-0.5f * lx * lx);
    const float e2 = eps * eps;
    const float xavg = p.t_decay * p.t_xavg[e] + w * (xi / (e2 * e2 * e2));
    const float tm = p.t_decay * p.t_time[e] + w;
    float en = powf(xavg / tm, -1.0f / 6.0f);
    en = (en < emax ? 1.0f : 0.0f) * en + (en > emax ? 1.0f : 0.0f) * emax;         // as written at warmup.py:317-319 (0 * inf = NaN included)
    t_Wold = p.t_W[e];
    t_wgt = (1.0f - p.t_mask) * (t_ok ? 1.0f : 0.0f) * en;
    logp_now = t_ok ? nan_to_num(logp_now) : p.bk_logp[e];    // the logdensity leaf of the state the chain continues from
    eps_next = en;                                            // params_new.step_size: what the next kernel step uses
    // the next step's ops can be formed from THIS launch's Gram matrix only if the accepted momentum / gradient are the
    // vectors it was built from, i.e. nan_to_num changed nothing: all finite
    t_simple = t_ok && isfinite(S[2]) && isfinite(S[0]) && isfinite(r0) && isfinite(r1) && isfinite(r2) && isfinite(r3) &&
               !p.force_restart;
    if (merged) lold = logp_now;
    if (tid == 0) {
      p.t_eps[e] = en;
      p.t_eps_max[e] = emax;
      p.t_xavg[e] = xavg;
      p.t_time[e] = tm;
      p.t_W[e] = t_Wold + t_wgt;
    }
  }
  if (flags & UPD_OB) ch.O(3, p.hB * eps_next, L, d);
  if (flags & UPD_B2) dk += ch.B(eps_next, p.coef_b2, gn, d);
  if (tid == 0) {
    p.dK[e] = dk;
    p.lold[e] = lold;
    if (from_slabs || tune) p.logp[e] = logp_now;
    if ((flags & UPD_RECORD) && p.out_info) {
      p.out_info[3 * e + 0] = logp_now;
      p.out_info[3 * e + 1] = info_dk;
      p.out_info[3 * e + 2] = info_de;
    }
    bc[0] = ch.c[0]; bc[1] = ch.c[1] * ign; bc[2] = ch.c[2]; bc[3] = ch.c[3];
    bc[4] = eps_next * p.coef_a; bc[5] = t_ok ? 1.0f : 0.0f; bc[6] = t_Wold; bc[7] = t_wgt;
    if (merged) { bc[8] = r0; bc[9] = r1; bc[10] = r2; bc[11] = r3; bc[12] = t_simple ? 1.0f : 0.0f; }
  }
  }
  __syncthreads();
  if (stamps) stamps[2] = wall_clock64();                                                          // dev: chain done
  // ---- pass 2 (from registers) ------------------------------------------------------------
  const bool any_op = flags & (UPD_B1 | UPD_OA | UPD_OB | UPD_B2);
  const bool doA = flags & UPD_A;
  const float c0 = bc[0], c1 = bc[1], c2 = bc[2], c3 = bc[3], ea = bc[4];
  const bool t_ok = bc[5] != 0.0f;
  const float t_Wold = bc[6], t_wgt = bc[7];
  if (tune && !t_ok) {   // handle_nans: this chain keeps its previous state (rare, workgroup-uniform)
    float *xo = merged ? p.x_acc : p.x, *uo = merged ? p.u_rec : p.u;
    for (int i = tid; i < d; i += nt) {
      xo[base + i] = p.bk_x[base + i];
      uo[base + i] = p.bk_u[base + i];
      p.g[base + i] = p.bk_g[base + i];
    }
    return merged;       // merged: the next step's O, B, A still have to run, from the restored state
  }
  float rc0 = 0.0f, rc1 = 0.0f, rc2 = 0.0f;
  bool simple = true;
  if (merged) { rc0 = bc[8]; rc1 = bc[9]; rc2 = bc[10]; simple = bc[12] != 0.0f; }
  const bool t_acc = tune && p.t_mask == 0.0f;              // streaming_average_update of [x, x^2]
  const float t_den = 1.0f / (t_Wold + t_wgt);              // zero_prevention = mask = 0 here
#pragma unroll
  for (int kq = 0; kq < NK; ++kq) {
    const int q = tid + kq * nt;
    const int k = BIG ? 0 : kq;
    if (q < nqf) {
      const size_t o = base + 4 * (size_t)q;
      if constexpr (BIG) {
        cx[0] = ld4<AL>(xin + base + (unsigned)(4 * q));
        cg[0] = gl[kq * nt + tid];
        if (flags & (UPD_OA | UPD_OB)) {   // the noise of pass 1 again (same counters)
          ca[0] = f32x4{0, 0, 0, 0};
          cb[0] = f32x4{0, 0, 0, 0};
          int q2 = q;
          asm volatile("" : "+v"(q2));     // opaque: otherwise the compiler keeps pass 1's values alive instead (111 spills)
          if (useA && !p.zA) ca[0] = philox_normal4(q2, pid, p.stepA, p.stageA, p.seed);
          if (useB && !p.zB) cb[0] = philox_normal4(q2, pid, p.stepB, p.stageB, p.seed);
        } else {
          ca[0] = f32x4{0, 0, 0, 0}; cb[0] = f32x4{0, 0, 0, 0};
        }
      }
      if (t_acc) {
        float *a0 = p.t_avg + (size_t)e * 2 * d + 4 * (size_t)q, *a1 = a0 + d;
        f32x4 m0 = ld4<AL>(a0), m1 = ld4<AL>(a1);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          m0[m] = (t_Wold * m0[m] + t_wgt * cx[k][m]) * t_den;
          m1[m] = (t_Wold * m1[m] + t_wgt * cx[k][m] * cx[k][m]) * t_den;
        }
        st4<AL>(a0, m0);
        st4<AL>(a1, m1);
      }
      f32x4 v = cu[kq];
      if (merged) {   // the accepted state's momentum (record point; zB is not part of it) stays in the step's own buffer
        f32x4 ur;
#pragma unroll
        for (int m = 0; m < 4; ++m) ur[m] = nan_to_num(fmaf(rc0, cu[kq][m], fmaf(rc1, cg[k][m], rc2 * ca[k][m])));
        st4<AL>(p.u_rec + o, ur);
        if (!simple) continue;   // O, B, A of the next step run from the stored (nan_to_num'd) state instead
      }
      if (any_op) {
#pragma unroll
        for (int m = 0; m < 4; ++m) v[m] = fmaf(c0, cu[kq][m], fmaf(c1, cg[k][m], fmaf(c2, ca[k][m], c3 * cb[k][m])));
        if (tune && !merged) {   // nan_to_num(next_state.momentum), warmup.py:478-482
#pragma unroll
          for (int m = 0; m < 4; ++m) v[m] = nan_to_num(v[m]);
        }
        st4<AL>(p.u + o, v);
      }
      if (p.out_sample) st4<AL>(p.out_sample + o, cx[k]);
      if (doA) {
        f32x4 xn;
#pragma unroll
        for (int m = 0; m < 4; ++m) xn[m] = fmaf(SDC ? ea * csd[kq][m] : ea, v[m], cx[k][m]);
        st4<AL>(p.x + o, xn);
      }
    }
    if constexpr (BIG) __builtin_amdgcn_sched_barrier(0);
  }
  if (has_tail) {
    if (t_acc) {
      float *a0 = p.t_avg + (size_t)e * 2 * d + 4 * (size_t)nqf + tid, *a1 = a0 + d;
      *a0 = (t_Wold * *a0 + t_wgt * tx) * t_den;
      *a1 = (t_Wold * *a1 + t_wgt * tx * tx) * t_den;
    }
    float v = tu;
    if (merged) p.u_rec[to] = nan_to_num(fmaf(rc0, tu, fmaf(rc1, tg, rc2 * ta)));
    if (!merged || simple) {
      if (any_op) { v = fmaf(c0, tu, fmaf(c1, tg, fmaf(c2, ta, c3 * tb))); if (tune && !merged) v = nan_to_num(v); p.u[to] = v; }
      if (p.out_sample) p.out_sample[to] = tx;
      if (doA) p.x[to] = fmaf(ea * tsd, v, tx);
    }
  }
  return merged && !simple;
}

// The next step's O(z1) . B(b1) . A(1/2) for a chain whose merged warm-up launch could not form them (rejected step, or an
// accepted state that nan_to_num changed): the ordinary first launch of a step, from the state the record launch left in the
// step's own buffer, with the step size the tuner has just written.
template <int NK, int AL, bool SDC>
__device__ __forceinline__ void upd_tune_restart(const UpdParams &p, const int e, const int tid, const int nt,
                                                 float (*red)[UPD_NSUM + 1], float *bc) {
  __syncthreads();                                   // the record pass's global stores / LDS reads are done
  UpdParams q = p;
  q.flags = UPD_START | UPD_B2 | UPD_A | (p.flags & UPD_OB);
  q.x_in = p.x_acc; q.u_in = p.u_rec; q.g_in = p.g; q.logp_in = p.logp;
  q.eps = p.t_eps;
  q.u_rec = nullptr; q.x_acc = nullptr; q.out_info = nullptr; q.out_sample = nullptr; q.zA = nullptr;
  upd_fast_body<NK, AL, SDC, false, -1>(q, e, tid, nt, red, bc);
}

// MAXT: the launch bound.  A launch never uses more threads than the quads need (d = 8834: 768), and a kernel promised
// <= 768 threads may use 170 registers instead of 128: the warm-up record kind (two coefficient sets, the restart path) spilled
// 19 VGPRs / 45 SGPRs under the 1024-thread bound.
template <int NK, int AL, bool SDC, int CF = -1, int MAXT = UPD_NT>
static __global__ __launch_bounds__(MAXT) void k_update_fast(const UpdParams p) {
  __shared__ float red[UPD_NW][UPD_NSUM + 1];
  __shared__ float bc[16];
  // launched with the fewest waves that still give NK quads per thread (nt = blockDim.x <= UPD_NT, a multiple of 64): the
  // kernel is VALU-bound on half the chip (E workgroups), so idle padded lanes cost real time (d = 8834: 768 threads, not 1024)
  const bool again = upd_fast_body<NK, AL, SDC, false, CF>(p, blockIdx.x, threadIdx.x, blockDim.x, red, bc);
  if constexpr (CF < 0 || (CF & UPD_TUNE) != 0) {
    if (again) upd_tune_restart<NK, AL, SDC>(p, blockIdx.x, threadIdx.x, blockDim.x, red, bc);
  }
}

// d beyond the register cache (B3: d = 34 562): NK = 5 .. UPD_QMAX_BIG quads per thread, g parked in dynamic LDS
// (NK * blockDim.x quads).  No tuner in this form (mile_tune runs k_tune_post after the step for these sizes).
#define UPD_QMAX_BIG 9
template <int NK, int AL, bool SDC, int CF = -1>
static __global__ __launch_bounds__(UPD_NT) void k_update_big(const UpdParams p) {
  __shared__ float red[UPD_NW][UPD_NSUM + 1];
  __shared__ float bc[16];
  extern __shared__ __attribute__((aligned(16))) char upd_gl[];
  upd_fast_body<NK, AL, SDC, false, CF, true>(p, blockIdx.x, threadIdx.x, blockDim.x, red, bc, nullptr, reinterpret_cast<f32x4 *>(upd_gl));
}

// which compile-time kind a launch is (else -1): steady-state flag words with a Normal prior, no tuner, no preconditioner
static inline int upd_kind(const UpdParams &u) {
  if (u.prior != MILE_PRIOR_NORMAL || u.sdc || getenv("MILE_NO_UPD_KIND")) return -1;
  if (u.flags == UPD_KIND_MID) return UPD_KIND_MID;
  if (u.flags == UPD_KIND_REC) return UPD_KIND_REC;
  if (u.flags == UPD_KIND_TUNE && u.u_rec) return UPD_KIND_TUNE;
  return -1;
}

#define AUX_NT 256
// g = sum_s slab + grad log prior;  logp = sum_s llpart + log prior.   (mile_logpost_grad)
static __global__ __launch_bounds__(AUX_NT) void k_finalize(int d, int dp, int S, int prior, float loc, float scale,
                                                     const float *theta, const float *slabs,
                                                     const float *llpart, float *grad, float *logp) {
  __shared__ float red[AUX_NT / 64];
  const int tid = threadIdx.x, e = blockIdx.x;
  const size_t base = (size_t)e * d;
  const float *sl = slabs + (size_t)e * S * dp;
  float pv = 0.0f;
  for (int i = tid; i < d; i += AUX_NT) {
    float gi = 0.0f;
    for (int s = 0; s < S; ++s) gi += sl[(size_t)s * dp + i];
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
    for (int w = 0; w < AUX_NT / 64; ++w) t += (double)red[w];
    for (int s = 0; s < S; ++s) t += (double)llpart[(size_t)e * S + s];
    t += prior == MILE_PRIOR_NORMAL ? -(double)d * (log((double)scale) + 0.91893853320467274)
                                    : -(double)d * log(2.0 * (double)scale);
    logp[e] = (float)t;
  }
}

// momentum = z / |z|  (generate_unit_vector of blackjax.mcmc.mclmc.init, A.1)
static __global__ __launch_bounds__(AUX_NT) void k_init_momentum(int d, const float *z, uint64_t seed,
                                                          const int32_t *pids, float *u) {
  __shared__ float red[AUX_NT / 64];
  const int tid = threadIdx.x, e = blockIdx.x;
  const size_t base = (size_t)e * d;
  const uint32_t pid = pids ? (uint32_t)pids[e] : (uint32_t)e;
  const int nq = (d + 3) >> 2;
  float ss = 0.0f;
  for (int q = tid; q < nq; q += AUX_NT) {
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
  for (int w = 0; w < AUX_NT / 64; ++w) t += red[w];
  const float inv = 1.0f / sqrtf(t);
  for (int i = tid; i < d; i += AUX_NT) u[base + i] *= inv;
}

static __global__ __launch_bounds__(AUX_NT) void k_debug_noise(int d, uint64_t seed, const int32_t *pids,
                                                        uint32_t step, uint32_t stage, float *out) {
  const int tid = threadIdx.x, e = blockIdx.x;
  const uint32_t pid = pids ? (uint32_t)pids[e] : (uint32_t)e;
  const int nq = (d + 3) >> 2;
  for (int q = tid; q < nq; q += AUX_NT) {
    const f32x4 zz = philox_normal4(q, pid, step, stage, seed);
    for (int m = 0; m < 4; ++m)
      if (4 * q + m < d) out[(size_t)e * d + 4 * q + m] = zz[m];
  }
}

// Warm-up tuner as a separate launch after a kernel step, for d beyond k_update_fast's register cache
// (same arithmetic as the UPD_TUNE block above: predictor + handle_nans warmup.py:271-326,468-483, streaming
// averages :343-348).  x/u/g/logp: the state after the step, bk_*: the state before it, info: the step's
// MCLMCInfo triple [E, 3].  One workgroup per particle.
struct TunePostParams {
  int32_t d;
  float *x, *u, *g, *logp;
  const float *bk_x, *bk_u, *bk_g, *bk_logp;
  float *info;
  float *t_eps, *t_eps_max, *t_time, *t_xavg, *t_W, *t_avg;
  float t_mask, t_var, t_trust, t_decay;
};

static __global__ __launch_bounds__(AUX_NT) void k_tune_post(const TunePostParams p) {
  __shared__ float red[AUX_NT / 64];
  __shared__ float bc[4];
  const int tid = threadIdx.x, e = blockIdx.x, d = p.d;
  const size_t base = (size_t)e * d;
  __shared__ float red2[AUX_NT / 64];
  float bad = 0.0f, bad_ug = 0.0f;
  for (int i = tid; i < d; i += AUX_NT) {
    bad += isfinite(p.x[base + i]) ? 0.0f : 1.0f;
    bad_ug += (isfinite(p.u[base + i]) && isfinite(p.g[base + i])) ? 0.0f : 1.0f;
  }
  bad = wave_sum(bad);
  bad_ug = wave_sum(bad_ug);
  if ((tid & 63) == 0) { red[tid >> 6] = bad; red2[tid >> 6] = bad_ug; }
  __syncthreads();
  if (tid == 0) {
    constexpr float FMAX = 3.4028234663852886e38f;
    float cnt = 0.0f;
    for (int w = 0; w < AUX_NT / 64; ++w) cnt += red[w];
    const bool ok = cnt == 0.0f;
    const float eps = p.t_eps[e];
    float dE = p.info[3 * e + 2];
    dE = ok ? (isnan(dE) ? 0.0f : fminf(fmaxf(dE, -FMAX), FMAX)) : 0.0f;
    const float emax = ok ? nan_to_num(p.t_eps_max[e]) : 0.8f * eps;
    const float xi = dE * dE / ((float)d * p.t_var) + 1e-8f;
    const float lx = logf(xi) / (6.0f * p.t_trust);
    const float w = expf(-0.5f * lx * lx);
    const float e2 = eps * eps;
    const float xavg = p.t_decay * p.t_xavg[e] + w * (xi / (e2 * e2 * e2));
    const float tm = p.t_decay * p.t_time[e] + w;
    float en = powf(xavg / tm, -1.0f / 6.0f);
    en = (en < emax ? 1.0f : 0.0f) * en + (en > emax ? 1.0f : 0.0f) * emax;
    const float Wold = p.t_W[e], wgt = (1.0f - p.t_mask) * (ok ? 1.0f : 0.0f) * en;
    p.t_eps[e] = en; p.t_eps_max[e] = emax; p.t_xavg[e] = xavg; p.t_time[e] = tm; p.t_W[e] = Wold + wgt;
    if (!ok) { p.logp[e] = p.bk_logp[e]; p.info[3 * e] = p.bk_logp[e]; }
    else p.logp[e] = nan_to_num(p.logp[e]);            // nan_to_num on every leaf of an accepted state (warmup.py:478-482)
    float cug = 0.0f;
    for (int w = 0; w < AUX_NT / 64; ++w) cug += red2[w];
    bc[0] = ok ? 1.0f : 0.0f; bc[1] = Wold; bc[2] = wgt; bc[3] = cug;
  }
  __syncthreads();
  const bool ok = bc[0] != 0.0f;
  if (!ok) {   // handle_nans: the chain keeps its previous state
    for (int i = tid; i < d; i += AUX_NT) {
      p.x[base + i] = p.bk_x[base + i];
      p.u[base + i] = p.bk_u[base + i];
      p.g[base + i] = p.bk_g[base + i];
    }
    return;
  }
  if (bc[3] != 0.0f)        // accepted with a non-finite momentum / gradient entry (rare): nan_to_num them as handle_nans does
    for (int i = tid; i < d; i += AUX_NT) {
      p.u[base + i] = nan_to_num(p.u[base + i]);
      p.g[base + i] = nan_to_num(p.g[base + i]);
    }
  if (p.t_mask == 0.0f) {   // streaming_average_update of [x, x^2] with weight eps (zero_prevention = 0)
    const float Wold = bc[1], wgt = bc[2], den = 1.0f / (Wold + wgt);
    float *a0 = p.t_avg + (size_t)e * 2 * d, *a1 = a0 + d;
    for (int i = tid; i < d; i += AUX_NT) {
      const float xv = p.x[base + i];
      a0[i] = (Wold * a0[i] + wgt * xv) * den;
      a1[i] = (Wold * a1[i] + wgt * xv * xv) * den;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Warm-start optimizer step (mile_warmstart_step): g = -(sum_s slab[s]) / B is the gradient of the batch-mean negative
// log-likelihood (the slabs hold the gradient of the LIKELIHOOD sum; no prior in the warm-start loss, trainer.py:729-737);
// optax's sgd / adam / adamw update of every member in one launch.  grid (ceil(d / 1024), E).
struct OptimParams {
  int32_t d, S, dp, kind;
  float lr, b1, b2, eps, wd, inv_batch;
  float bc1, bc2;                 // 1 - b1^t, 1 - b2^t
  float *theta, *m, *v;
  const float *slabs, *llpart;
  const uint8_t *active;
  float *out_nll;
};
static __global__ __launch_bounds__(256) void k_optim_step(const OptimParams p) {
  const int e = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x == 0 && p.out_nll) {
    float ll = 0.0f;
    for (int s = 0; s < p.S; ++s) ll += p.llpart[(size_t)e * p.S + s];
    p.out_nll[e] = -ll * p.inv_batch;
  }
  if (i >= p.d) return;
  if (p.active && !p.active[e]) return;                       // early-stopped member: parameters and moments frozen
  const float *sl = p.slabs + (size_t)e * p.S * p.dp + i;
  float g = 0.0f;
#pragma unroll 4
  for (int s = 0; s < p.S; ++s) g += sl[(size_t)s * p.dp];
  g = -g * p.inv_batch;
  const size_t o = (size_t)e * p.d + i;
  const float th = p.theta[o];
  float upd;
  if (p.kind == MILE_OPT_SGD) {
    upd = p.lr * g;
  } else {
    const float m = p.b1 * p.m[o] + (1.0f - p.b1) * g;
    const float v = p.b2 * p.v[o] + (1.0f - p.b2) * g * g;
    p.m[o] = m; p.v[o] = v;
    const float mh = m / p.bc1, vh = v / p.bc2;
    upd = p.lr * (mh / (sqrtf(vh) + p.eps) + (p.kind == MILE_OPT_ADAMW ? p.wd * th : 0.0f));
  }
  p.theta[o] = th - upd;
}
