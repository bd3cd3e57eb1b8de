/* CPU restatement (C, fp32, OpenMP over particles) of the MCLMC hot path -- TEST / BASELINE
 * INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED against the reference (see the header of
 * oracle/mclmc_oracle.py); this file is checked against that NumPy oracle in tests/test_oracle_c.py
 * and exists so that bench.py's cpu_baseline is a fair CPU number: one chain per core, weights
 * in cache, vectorised inner loops -- the shape of the reference's own CPU run (one chain per XLA
 * host device, src/training/sampling.py:180-188).
 *
 * Follows: Dense stack src/flax_building_blocks/basic.py:42-61; Gaussian head + nansum
 * src/training/probabilistic.py:92-100; Normal prior src/training/priors.py:101-108; blackjax 1.2.2
 * isokinetic McLachlan MCLMC step with explicit noise.
 * Supported: ReLU, regression head, Normal prior (what BASELINE configs B1/B2 use).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXL 16

typedef struct {
  int n_layers, in_features, widths[MAXL], w_off[MAXL], b_off[MAXL], d;
  float prior_loc, prior_scale;
} cpu_spec;

void cpu_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int cpu_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ravel_pytree order for < 11 layers: per layer bias[out], kernel[in,out] */
void cpu_spec_init(cpu_spec *s, int in_features, int n_layers, const int *widths, float loc, float scale) {
  s->n_layers = n_layers; s->in_features = in_features; s->prior_loc = loc; s->prior_scale = scale;
  int off = 0, fin = in_features;
  for (int l = 0; l < n_layers; ++l) {
    s->widths[l] = widths[l];
    s->b_off[l] = off; off += widths[l];
    s->w_off[l] = off; off += fin * widths[l];
    fin = widths[l];
  }
  s->d = off;
}

/* one particle: log posterior and gradient; scratch holds activations for RB rows at a time */
#define RB 64
static float logpost_grad_one(const cpu_spec *s, const float *th, const float *X, const float *y, int N, float *g,
                              float *scratch) {
  const int nl = s->n_layers, F = s->in_features, d = s->d;
  int maxw = F, sumw = F;
  for (int l = 0; l < nl; ++l) { if (s->widths[l] > maxw) maxw = s->widths[l]; sumw += s->widths[l]; }
  float *act = scratch;                 /* [RB][sumw] */
  float *dz = act + RB * sumw;          /* [RB][maxw] */
  float *dzn = dz + RB * maxw;          /* [RB][maxw] */
  memset(g, 0, sizeof(float) * d);
  double ll = 0.0;
  for (int r0 = 0; r0 < N; r0 += RB) {
    const int nr = N - r0 < RB ? N - r0 : RB;
    for (int r = 0; r < nr; ++r) memcpy(act + r * sumw, X + (size_t)(r0 + r) * F, sizeof(float) * F);
    int aoff = 0, fin = F;
    for (int l = 0; l < nl; ++l) {
      const int fo = s->widths[l];
      const float *W = th + s->w_off[l], *b = th + s->b_off[l];
      for (int r = 0; r < nr; ++r) {
        const float *a = act + r * sumw + aoff;
        float *z = act + r * sumw + aoff + fin;
        for (int o = 0; o < fo; ++o) z[o] = b[o];
        for (int i = 0; i < fin; ++i) {
          const float ai = a[i];
          const float *Wi = W + (size_t)i * fo;
          for (int o = 0; o < fo; ++o) z[o] += ai * Wi[o];
        }
        if (l < nl - 1)
          for (int o = 0; o < fo; ++o) z[o] = z[o] > 0.0f ? z[o] : 0.0f;
      }
      aoff += fin; fin = fo;
    }
    /* Gaussian head */
    for (int r = 0; r < nr; ++r) {
      const float *out = act + r * sumw + aoff;
      const float mu = out[0], sr = out[1];
      const float es = expf(sr);
      float sig = es < 1e-6f ? 1e-6f : (es > 1e6f ? 1e6f : es);
      const int unclipped = es > 1e-6f && es < 1e6f;
      const float rr = (y[r0 + r] - mu) / sig;
      float l1 = -0.5f * rr * rr - logf(sig) - 0.91893853320467274f;
      float dmu = rr / sig, ds = unclipped ? rr * rr - 1.0f : 0.0f;
      if (isnan(l1) || isnan(es) || isnan(mu)) { l1 = 0.0f; dmu = 0.0f; ds = 0.0f; }
      ll += l1;
      dz[r * maxw + 0] = dmu; dz[r * maxw + 1] = ds;
    }
    /* backward */
    float *dzc = dz, *dzo = dzn;
    for (int l = nl - 1; l >= 0; --l) {
      const int fo = s->widths[l];
      const int fi = l == 0 ? F : s->widths[l - 1];
      aoff -= fi;
      const float *W = th + s->w_off[l];
      float *gW = g + s->w_off[l], *gb = g + s->b_off[l];
      for (int r = 0; r < nr; ++r) {
        const float *a = act + r * sumw + aoff;
        const float *dzr = dzc + r * maxw;
        for (int o = 0; o < fo; ++o) gb[o] += dzr[o];
        for (int i = 0; i < fi; ++i) {
          const float ai = a[i];
          float *gWi = gW + (size_t)i * fo;
          for (int o = 0; o < fo; ++o) gWi[o] += ai * dzr[o];
        }
        if (l > 0) {
          float *dn = dzo + r * maxw;
          for (int i = 0; i < fi; ++i) {
            const float *Wi = W + (size_t)i * fo;
            float acc = 0.0f;
            for (int o = 0; o < fo; ++o) acc += Wi[o] * dzr[o];
            dn[i] = a[i] > 0.0f ? acc : 0.0f;
          }
        }
      }
      float *t = dzc; dzc = dzo; dzo = t;
    }
  }
  /* prior */
  double lp = 0.0;
  const float sc = s->prior_scale, loc = s->prior_loc;
  for (int i = 0; i < d; ++i) {
    const float t = (th[i] - loc) / sc;
    lp += -0.5 * (double)t * t;
    g[i] -= t / sc;
  }
  lp += -(double)d * (log((double)sc) + 0.91893853320467274);
  return (float)(ll + lp);
}

static size_t scratch_floats(const cpu_spec *s) {
  int maxw = s->in_features, sumw = s->in_features;
  for (int l = 0; l < s->n_layers; ++l) { if (s->widths[l] > maxw) maxw = s->widths[l]; sumw += s->widths[l]; }
  return (size_t)RB * (sumw + 2 * maxw);
}

void cpu_logpost_grad(const cpu_spec *s, const float *theta, int E, const float *X, const float *y, int N,
                      float *logp, float *grad) {
#pragma omp parallel
  {
    float *scratch = (float *)malloc(sizeof(float) * scratch_floats(s));
#pragma omp for schedule(dynamic, 1)
    for (int e = 0; e < E; ++e)
      logp[e] = logpost_grad_one(s, theta + (size_t)e * s->d, X, y, N, grad + (size_t)e * s->d, scratch);
    free(scratch);
  }
}

static float bstep(float *u, const float *g, int d, float eps, float coef) {   /* A.2, returns dK */
  /* normalized_flatten_array (tol 1e-13) serves the gradient and the new momentum alike: a (near-)zero vector stays as it is */
  double gg = 0.0, ug = 0.0;
  for (int i = 0; i < d; ++i) { gg += (double)g[i] * g[i]; ug += (double)u[i] * g[i]; }
  const float gn = (float)sqrt(gg);
  const float ign = gn > 1e-13f ? 1.0f / gn : 1.0f;
  const float ue = (float)ug * ign;
  const float delta = eps * coef * gn / (float)(d - 1), zeta = expf(-delta);
  const float ce = (1.0f - zeta) * (1.0f + zeta + ue * (1.0f - zeta)) * ign, cu = 2.0f * zeta;
  double nn = 0.0;
  for (int i = 0; i < d; ++i) { u[i] = ce * g[i] + cu * u[i]; nn += (double)u[i] * u[i]; }
  const float un = (float)sqrt(nn);
  if (un > 1e-13f) {
    const float inv = 1.0f / un;
    for (int i = 0; i < d; ++i) u[i] *= inv;
  }
  return (float)(d - 1) * (delta - 0.69314718055994531f + logf(1.0f + ue + (1.0f - ue) * zeta * zeta));
}

static void ostep(float *u, const float *z, int d, float h, float L) {          /* A.5 */
  const float nu = sqrtf((expf(2.0f * h / L) - 1.0f) / (float)d);
  double nn = 0.0;
  for (int i = 0; i < d; ++i) { u[i] += nu * z[i]; nn += (double)u[i] * u[i]; }
  const float inv = (float)(1.0 / sqrt(nn));
  for (int i = 0; i < d; ++i) u[i] *= inv;
}

/* n_steps kernel steps (O . B A B A B . O) of every particle, explicit noise [n_steps, 2, E, d];
 * info [n_steps, E, 3] = (logdensity, kinetic_change, energy_change) or NULL */
void cpu_mclmc_steps(const cpu_spec *s, float *x, float *u, float *logp, float *g, int E, const float *eps,
                     const float *L, const float *noise, int n_steps, const float *X, const float *y, int N,
                     float *info) {
  const int d = s->d;
  const float b1 = 0.1931833275037836f, b2 = 1.0f - 2.0f * 0.1931833275037836f;
#pragma omp parallel
  {
    float *scratch = (float *)malloc(sizeof(float) * scratch_floats(s));
#pragma omp for schedule(dynamic, 1)
    for (int e = 0; e < E; ++e) {
      float *xe = x + (size_t)e * d, *ue = u + (size_t)e * d, *ge = g + (size_t)e * d;
      for (int t = 0; t < n_steps; ++t) {
        const float *z1 = noise + (((size_t)t * 2 + 0) * E + e) * d, *z2 = noise + (((size_t)t * 2 + 1) * E + e) * d;
        const float l_old = logp[e], h = eps[e];
        ostep(ue, z1, d, 0.5f * h, L[e]);
        float dK = bstep(ue, ge, d, h, b1);
        for (int i = 0; i < d; ++i) xe[i] += h * 0.5f * ue[i];
        logp[e] = logpost_grad_one(s, xe, X, y, N, ge, scratch);
        dK += bstep(ue, ge, d, h, b2);
        for (int i = 0; i < d; ++i) xe[i] += h * 0.5f * ue[i];
        logp[e] = logpost_grad_one(s, xe, X, y, N, ge, scratch);
        dK += bstep(ue, ge, d, h, b1);
        ostep(ue, z2, d, 0.5f * h, L[e]);
        if (info) {
          float *o = info + ((size_t)t * E + e) * 3;
          o[0] = logp[e]; o[1] = dK; o[2] = dK - logp[e] + l_old;
        }
      }
    }
    free(scratch);
  }
}
