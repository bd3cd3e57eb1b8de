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
 * Supported: ReLU / tanh / sigmoid (src/config/models/base.py:25-39), Gaussian regression head and softmax classification
 * head (probabilistic.py:92-109), Normal and Laplace priors (priors.py:101-128): BASELINE configs B1-B4.
 * Parallelism: one particle per thread when there are at least as many particles as threads (the reference's layout);
 * with fewer particles (B4's bounded sample) the rows of a particle are split over the threads instead, each with its own
 * gradient accumulator, summed in thread order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXL 16

enum { ACT_RELU = 0, ACT_TANH = 1, ACT_SIGMOID = 2 };
enum { TASK_REGR = 0, TASK_CLASS = 1 };
enum { PRIOR_NORMAL = 0, PRIOR_LAPLACE = 1 };

typedef struct {
  int n_layers, in_features, widths[MAXL], w_off[MAXL], b_off[MAXL], d;
  float prior_loc, prior_scale;
  int activation, task, prior;
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
void cpu_spec_init(cpu_spec *s, int in_features, int n_layers, const int *widths, float loc, float scale, int activation,
                   int task, int prior) {
  s->n_layers = n_layers; s->in_features = in_features; s->prior_loc = loc; s->prior_scale = scale;
  s->activation = activation; s->task = task; s->prior = prior;
  int off = 0, fin = in_features;
  for (int l = 0; l < n_layers; ++l) {
    s->widths[l] = widths[l];
    s->b_off[l] = off; off += widths[l];
    s->w_off[l] = off; off += fin * widths[l];
    fin = widths[l];
  }
  s->d = off;
}

/* log-likelihood of rows [r_begin, r_end) of one particle and its gradient ACCUMULATED into g (caller zeroes it);
 * scratch holds activations for RB rows at a time.  y: float targets (regr) or int32 labels (classification). */
#define RB 64
static double loglik_rows(const cpu_spec *s, const float *th, const float *X, const void *yv, int r_begin, int r_end, float *g,
                          float *scratch) {
  const int nl = s->n_layers, F = s->in_features, K = s->widths[nl - 1];
  int maxw = F, sumw = F;
  for (int l = 0; l < nl; ++l) { if (s->widths[l] > maxw) maxw = s->widths[l]; sumw += s->widths[l]; }
  float *act = scratch;                 /* [RB][sumw] */
  float *dz = act + RB * sumw;          /* [RB][maxw] */
  float *dzn = dz + RB * maxw;          /* [RB][maxw] */
  const float *yf = (const float *)yv;
  const int32_t *yi = (const int32_t *)yv;
  double ll = 0.0;
  for (int r0 = r_begin; r0 < r_end; r0 += RB) {
    const int nr = r_end - r0 < RB ? r_end - r0 : RB;
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
        if (l < nl - 1) {
          if (s->activation == ACT_RELU) for (int o = 0; o < fo; ++o) z[o] = z[o] > 0.0f ? z[o] : 0.0f;
          else if (s->activation == ACT_TANH) for (int o = 0; o < fo; ++o) z[o] = tanhf(z[o]);
          else for (int o = 0; o < fo; ++o) z[o] = 1.0f / (1.0f + expf(-z[o]));
        }
      }
      aoff += fin; fin = fo;
    }
    if (s->task == TASK_REGR) {           /* Gaussian head, probabilistic.py:92-100 */
      for (int r = 0; r < nr; ++r) {
        const float *out = act + r * sumw + aoff;
        const float mu = out[0], sr = out[1];
        const float es = expf(sr);
        float sig = es < 1e-6f ? 1e-6f : (es > 1e6f ? 1e6f : es);
        const int unclipped = es > 1e-6f && es < 1e6f;
        const float rr = (yf[r0 + r] - mu) / sig;
        float l1 = -0.5f * rr * rr - logf(sig) - 0.91893853320467274f;
        float dmu = rr / sig, ds = unclipped ? rr * rr - 1.0f : 0.0f;
        if (isnan(l1) || isnan(es) || isnan(mu)) { l1 = 0.0f; dmu = 0.0f; ds = 0.0f; }
        ll += l1;
        dz[r * maxw + 0] = dmu; dz[r * maxw + 1] = ds;
      }
    } else {                              /* log_softmax(out)[y], probabilistic.py:101-109 */
      for (int r = 0; r < nr; ++r) {
        const float *out = act + r * sumw + aoff;
        float m = out[0];
        for (int k = 1; k < K; ++k) m = out[k] > m ? out[k] : m;
        float se = 0.0f;
        for (int k = 0; k < K; ++k) se += expf(out[k] - m);
        const float lse = m + logf(se);
        const int lab = yi[r0 + r];
        float l1 = out[lab] - lse;
        const int bad = isnan(l1);
        if (bad) l1 = 0.0f;
        ll += l1;
        for (int k = 0; k < K; ++k) dz[r * maxw + k] = bad ? 0.0f : (k == lab ? 1.0f : 0.0f) - expf(out[k] - lse);
      }
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
            const float h = a[i];             /* the layer's activation output */
            dn[i] = s->activation == ACT_RELU ? (h > 0.0f ? acc : 0.0f)
                  : s->activation == ACT_TANH ? acc * (1.0f - h * h) : acc * h * (1.0f - h);
          }
        }
      }
      float *t = dzc; dzc = dzo; dzo = t;
    }
  }
  return ll;
}

static double log_prior_add(const cpu_spec *s, const float *th, float *g) {
  double lp = 0.0;
  const int d = s->d;
  const float sc = s->prior_scale, loc = s->prior_loc;
  if (s->prior == PRIOR_NORMAL) {
    for (int i = 0; i < d; ++i) {
      const float t = (th[i] - loc) / sc;
      lp += -0.5 * (double)t * t;
      g[i] -= t / sc;
    }
    lp += -(double)d * (log((double)sc) + 0.91893853320467274);
  } else {
    for (int i = 0; i < d; ++i) {
      const float t = (th[i] - loc) / sc;
      lp += -fabs((double)t);
      g[i] -= (t > 0.0f ? 1.0f : (t < 0.0f ? -1.0f : 0.0f)) / sc;
    }
    lp += -(double)d * log(2.0 * (double)sc);
  }
  return lp;
}

static size_t scratch_floats(const cpu_spec *s) {
  int maxw = s->in_features, sumw = s->in_features;
  for (int l = 0; l < s->n_layers; ++l) { if (s->widths[l] > maxw) maxw = s->widths[l]; sumw += s->widths[l]; }
  return (size_t)RB * (sumw + 2 * maxw);
}

/* one particle on the calling thread */
static float logpost_grad_one(const cpu_spec *s, const float *th, const float *X, const void *y, int N, float *g, float *scratch) {
  memset(g, 0, sizeof(float) * s->d);
  const double ll = loglik_rows(s, th, X, y, 0, N, g, scratch);
  return (float)(ll + log_prior_add(s, th, g));
}

/* one particle with its rows split over all threads (fewer particles than threads): per-thread gradient accumulators in
 * `gpart` [threads][d], summed in thread order */
static float logpost_grad_split(const cpu_spec *s, const float *th, const float *X, const void *y, int N, float *g, float *gpart,
                                int nthreads) {
  const int d = s->d;
  const int nblk = (N + RB - 1) / RB;
  double llp[256];
  if (nthreads > 256) nthreads = 256;
#pragma omp parallel num_threads(nthreads)
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
    const int t = 0, nt = 1;
#endif
    float *scratch = (float *)malloc(sizeof(float) * scratch_floats(s));
    float *gl = gpart + (size_t)t * d;
    memset(gl, 0, sizeof(float) * d);
    const int b0 = (int)((long long)nblk * t / nt), b1 = (int)((long long)nblk * (t + 1) / nt);
    const int r0 = b0 * RB, r1 = b1 * RB < N ? b1 * RB : N;
    llp[t] = r0 < r1 ? loglik_rows(s, th, X, y, r0, r1, gl, scratch) : 0.0;
    free(scratch);
#pragma omp barrier
#pragma omp for schedule(static)
    for (int i = 0; i < d; ++i) {
      float acc = 0.0f;
      for (int k = 0; k < nt; ++k) acc += gpart[(size_t)k * d + i];
      g[i] = acc;
    }
#pragma omp single
    { for (int k = nt; k < 256; ++k) llp[k] = 0.0; }
  }
  double ll = 0.0;
  for (int k = 0; k < nthreads; ++k) ll += llp[k];
  return (float)(ll + log_prior_add(s, th, g));
}

void cpu_logpost_grad(const cpu_spec *s, const float *theta, int E, const float *X, const void *y, int N,
                      float *logp, float *grad) {
  const int nth = cpu_threads();
  if (E < nth && nth > 1) {
    float *gpart = (float *)malloc(sizeof(float) * (size_t)nth * s->d);
    for (int e = 0; e < E; ++e)
      logp[e] = logpost_grad_split(s, theta + (size_t)e * s->d, X, y, N, grad + (size_t)e * s->d, gpart, nth);
    free(gpart);
    return;
  }
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

static void steps_one(const cpu_spec *s, float *xe, float *ue, float *lpe, float *ge, int e, int E, float h, float Le,
                      const float *noise, int n_steps, const float *X, const void *y, int N, float *info, float *scratch,
                      float *gpart, int nsplit) {
  const int d = s->d;
  const float b1 = 0.1931833275037836f, b2 = 1.0f - 2.0f * 0.1931833275037836f;
  for (int t = 0; t < n_steps; ++t) {
    const float *z1 = noise + (((size_t)t * 2 + 0) * E + e) * d, *z2 = noise + (((size_t)t * 2 + 1) * E + e) * d;
    const float l_old = *lpe;
    ostep(ue, z1, d, 0.5f * h, Le);
    float dK = bstep(ue, ge, d, h, b1);
    for (int i = 0; i < d; ++i) xe[i] += h * 0.5f * ue[i];
    *lpe = gpart ? logpost_grad_split(s, xe, X, y, N, ge, gpart, nsplit) : logpost_grad_one(s, xe, X, y, N, ge, scratch);
    dK += bstep(ue, ge, d, h, b2);
    for (int i = 0; i < d; ++i) xe[i] += h * 0.5f * ue[i];
    *lpe = gpart ? logpost_grad_split(s, xe, X, y, N, ge, gpart, nsplit) : logpost_grad_one(s, xe, X, y, N, ge, scratch);
    dK += bstep(ue, ge, d, h, b1);
    ostep(ue, z2, d, 0.5f * h, Le);
    if (info) {
      float *o = info + ((size_t)t * E + e) * 3;
      o[0] = *lpe; o[1] = dK; o[2] = dK - *lpe + l_old;
    }
  }
}

/* n_steps kernel steps (O . B A B A B . O) of every particle, explicit noise [n_steps, 2, E, d];
 * info [n_steps, E, 3] = (logdensity, kinetic_change, energy_change) or NULL */
void cpu_mclmc_steps(const cpu_spec *s, float *x, float *u, float *logp, float *g, int E, const float *eps,
                     const float *L, const float *noise, int n_steps, const float *X, const void *y, int N,
                     float *info) {
  const int d = s->d;
  const int nth = cpu_threads();
  if (E < nth && nth > 1) {   /* fewer particles than threads: particles in turn, rows over the threads */
    float *gpart = (float *)malloc(sizeof(float) * (size_t)nth * d);
    for (int e = 0; e < E; ++e)
      steps_one(s, x + (size_t)e * d, u + (size_t)e * d, logp + e, g + (size_t)e * d, e, E, eps[e], L[e], noise, n_steps, X, y, N,
                info, NULL, gpart, nth);
    free(gpart);
    return;
  }
#pragma omp parallel
  {
    float *scratch = (float *)malloc(sizeof(float) * scratch_floats(s));
#pragma omp for schedule(dynamic, 1)
    for (int e = 0; e < E; ++e)
      steps_one(s, x + (size_t)e * d, u + (size_t)e * d, logp + e, g + (size_t)e * d, e, E, eps[e], L[e], noise, n_steps, X, y, N,
                info, scratch, NULL, 1);
    free(scratch);
  }
}
