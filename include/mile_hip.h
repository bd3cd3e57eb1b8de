/*
 * mile_hip.h -- C ABI of the MI355X-native MCLMC ensemble sampler (libmile_hip.so).
 *
 * This is the drop-in boundary for ONE path of zhiyuan-yang/MILE: the MCLMC
 * integrator step over an ensemble of BNN-parameter particles with the
 * per-particle full-batch grad-log-posterior of the FCN MLP.  Every entry point
 * names the reference interface it replaces (paths relative to the reference
 * repository root).  The reference binds that path through Python callables
 * (blackjax.mclmc(logdensity_fn, L, step_size) -> SamplingAlgorithm(init, step));
 * a Python closure cannot cross a C ABI, so the closure's CONTENT crosses
 * instead: the model spec (src/config/models/fcn.py:7-30), the prior
 * (src/training/priors.py:67-91), the task (src/training/probabilistic.py:92-109)
 * and the training data (src/training/trainer.py:576-580).
 *
 * Conventions
 *   - All array pointers are DEVICE pointers into caller-owned memory (PyTorch-ROCm
 *     tensors), contiguous, row-major, fp32 unless stated.  The library never
 *     frees or retains caller memory beyond the call, except mile_set_data which
 *     COPIES X and y into its own padded layout.
 *   - [E, d] arrays: one row per particle (chain), d = mile_param_count(), in
 *     jax.flatten_util.ravel_pytree order of the FCN param tree: for each layer in
 *     sorted-name order ('layer0','layer1','layer10','layer11','layer2',...):
 *     bias[out] then kernel[in, out] row-major (src/training/priors.py:105,
 *     src/training/warmup.py:341,442).
 *   - `stream` is a hipStream_t passed as void*; work is enqueued on it and the
 *     call returns without synchronising.  No internal threads.
 *   - Return value: 0 on success, negative mile_status on error; the message is
 *     available from mile_last_error() (thread-local).
 *   - One handle per device; a handle is not thread-safe; distinct handles are
 *     independent.
 */
#ifndef MILE_HIP_H_
#define MILE_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MILE_ABI_VERSION 4
#define MILE_MAX_LAYERS 16

typedef enum mile_status {
  MILE_OK = 0,
  MILE_ERR_INVALID = -1,     /* bad argument / unsupported spec */
  MILE_ERR_STATE = -2,       /* call order (e.g. no data set, workspace too small) */
  MILE_ERR_HIP = -3,         /* HIP runtime error */
  MILE_ERR_NOMEM = -4
} mile_status;

/* src/config/models/base.py:25-39 (Activation) */
typedef enum mile_activation { MILE_ACT_RELU = 0, MILE_ACT_TANH = 1, MILE_ACT_SIGMOID = 2 } mile_activation;
/* src/config/data.py Task; likelihoods at src/training/probabilistic.py:92-109 */
typedef enum mile_task { MILE_TASK_REGRESSION = 0, MILE_TASK_CLASSIFICATION = 1 } mile_task;
/* src/training/priors.py:47-49 (StandardNormal == NORMAL with loc 0, scale 1) */
typedef enum mile_prior { MILE_PRIOR_NORMAL = 0, MILE_PRIOR_LAPLACE = 1 } mile_prior;
/* Placement of the partial momentum refresh inside one kernel step. */
typedef enum mile_refresh { MILE_REFRESH_O_STEP_O = 0, MILE_REFRESH_STEP_O = 1 } mile_refresh;
/* Which grad-log-posterior kernel to use.  AUTO picks the fastest fp32-accurate kernel that supports the spec
 * (MFMA_W64_BF16X3 reproduces fp32 products exactly from three-term bf16 splits and counts as one); the
 * bf16-operand kernel MFMA_W128_BF16 (operands ROUNDED to bf16, fp32 accumulate) is only ever selected explicitly. */
typedef enum mile_grad_kernel {
  MILE_GRAD_AUTO = 0,
  MILE_GRAD_GENERIC = 1,          /* any FCN, fp32 VALU */
  MILE_GRAD_MFMA_W64 = 2,         /* ReLU regression, 1-3 hidden layers of width 64, fp32 MFMA */
  MILE_GRAD_MFMA_W128_BF16 = 3,   /* ReLU regression, 1-3 hidden layers of width 128, bf16 MFMA */
  MILE_GRAD_GEMM_F32 = 4,         /* any FCN, fp32: rocBLAS strided-batched SGEMMs + elementwise HIP kernels (wide nets) */
  MILE_GRAD_LENET_F32 = 5,        /* MILE_MODEL_LENET only: im2col + the same SGEMMs, pooling / col2im HIP kernels */
  MILE_GRAD_MFMA_W64_BF16X3 = 6,  /* as MFMA_W64 with 2-3 hidden layers; the hidden->hidden forward / dH / dW products run as six
                                     bf16 MFMA products of exact three-term bf16 splits of the fp32 operands (fp32-faithful) */
  MILE_GRAD_MFMA_WIDE_BF16X3 = 7, /* any FCN, layer-wise: hand-written batched MFMA GEMMs (k_mm3) with the same fp32-faithful
                                     three-term bf16 products, bias / activation / activation-derivative fused into their
                                     epilogues; what AUTO picks for wide nets (hidden width >= 96: B4's 4 x 256 softmax net) */
  MILE_GRAD_MFMA_WIDE_BF16 = 8,   /* the same kernels with bf16-ROUNDED operands (one product instead of six); explicit only */
  MILE_GRAD_MFMA_NARROW_F32 = 10, /* FCNs with 1-3 hidden layers of width <= 64, F <= 64 inputs (or 4-10 hidden layers of width <= 16,
                                     F <= 16: the reference's depth ablations), <= 16 outputs, any activation,
                                     either head: fused forward + backward on v_mfma_f32_16x16x4_f32 (fp32 operands -- exact fp32
                                     products, fp32 accumulation); what AUTO picks for the reference's own 16- / 32-wide nets
                                     (experiments/replicate_uci/mclmc.yaml [16,16,2], tabluar_classif/covertype.yaml [32,7]) */
  MILE_GRAD_LENET_BF16 = 9        /* MILE_MODEL_LENET, <= 4 image channels: the five convolution products as implicit GEMMs on
                                     v_mfma_f32_16x16x32_bf16 with bf16-ROUNDED operands (BASELINE config 5 names bf16), the rest
                                     as LENET_F32; explicit only */
} mile_grad_kernel;
/* Which network: the FCN (src/models/tabular/fcn.py:16-28) or LeNet (src/models/images/cnns.py:10-66). */
typedef enum mile_model { MILE_MODEL_FCN = 0, MILE_MODEL_LENET = 1 } mile_model;

/* FCNConfig (src/config/models/fcn.py:7-30) + PriorConfig (src/config/sampler.py:60-95)
 * + Task: everything log_unnormalized_posterior (src/training/probabilistic.py:115-138)
 * closes over, apart from the data. */
typedef struct mile_model_spec {
  int32_t in_features;               /* F: columns of X */
  int32_t n_layers;                  /* len(hidden_structure); last entry is the output layer */
  int32_t widths[MILE_MAX_LAYERS];   /* hidden_structure */
  int32_t activation;                /* mile_activation, between layers, none after the last */
  int32_t task;                      /* mile_task: regr -> output width 2 (mu, log sigma) */
  int32_t prior;                     /* mile_prior */
  float prior_loc;
  float prior_scale;
  int32_t use_bias;                  /* FCNConfig.use_bias; only 1 is supported */
  int32_t model;                     /* mile_model.  LENET: X rows are NCHW images, in_features = C*H*W,
                                      * n_layers = 1 and widths[0] = out_dim; parameter order is ravel_pytree's
                                      * (conv1, conv2, fc1, fc2, fc3; bias before kernel [kh,kw,in,out]) */
  int32_t img_c;                     /* LENET image geometry (ignored for the FCN) */
  int32_t img_h;
  int32_t img_w;
} mile_model_spec;

/* blackjax IntegratorState(position, momentum, logdensity, logdensity_grad) for an
 * ensemble (src/types.py:17-27 State is its `position` prefix). */
typedef struct mile_state {
  int32_t n_particles;       /* E */
  float *position;           /* [E, d] */
  float *momentum;           /* [E, d], unit rows */
  float *logdensity;         /* [E] */
  float *logdensity_grad;    /* [E, d] */
} mile_state;

/* Arguments of n_steps kernel steps == the lax.scan body of
 * src/training/sampling.py:140-178 run n_steps times. */
typedef struct mile_step_args {
  const float *step_size;        /* [E] per-chain step size (warmup_params.txt line 1) */
  const float *L;                /* [E] per-chain momentum decoherence length (line 2) */
  const float *sqrt_diag_cov;    /* [E, d] or NULL (== 1.0, what blackjax.mclmc defaults to) */
  const float *noise;            /* [n_steps, 2, E, d] N(0,1) draws (parity mode) or NULL */
  uint64_t seed;                 /* counter RNG (Philox4x32-10 + Box-Muller) when noise == NULL */
  const int32_t *particle_ids;   /* [E] GLOBAL chain ids keying the RNG streams, or NULL => 0..E-1 */
  int64_t step_offset;           /* index of the first step: RNG counter and thinning predicate */
  int32_t n_steps;
  int32_t n_thinning;            /* keep position when (step_offset+i) % n_thinning == 0; <=0: keep none */
  int32_t refresh;               /* mile_refresh */
  float *out_samples;            /* [n_kept, E, d] kept positions in step order, or NULL */
  float *out_info;               /* [n_steps, E, 3] (logdensity, kinetic_change, energy_change) or NULL */
} mile_step_args;

/* Arguments of n_steps warm-up steps with on-device step-size adaptation == the scan body `step`
 * of make_L_step_size_adaptation (src/training/warmup.py:271-363): kernel step, handle_nans
 * (:468-483), the energy-variance step-size predictor, and the streaming averages of x and x^2
 * that give L = sqrt(sum Var[x_i]) after tune2.  One tuner per chain; all arrays are device
 * memory owned by the caller and updated in place. */
typedef struct mile_tune_args {
  float *step_size;              /* [E] in/out */
  const float *L;                /* [E] */
  const float *sqrt_diag_cov;    /* [E, d] or NULL */
  float *step_size_max;          /* [E] in/out, start at +inf */
  float *time;                   /* [E] in/out, start at 0 */
  float *x_average;              /* [E] in/out, start at 0 */
  float *stream_weight;          /* [E] in/out, start at 0 */
  float *stream_average;         /* [E, 2, d] in/out, start at 0: weighted means of x and x^2 */
  const float *noise;            /* [n_steps, 2, E, d] or NULL (counter RNG) */
  uint64_t seed;
  const int32_t *particle_ids;   /* [E] or NULL */
  int64_t step_offset;           /* RNG step counter of the first step */
  int32_t n_steps;
  int32_t schedule_step0;        /* position of the first step in the schedule (0 .. tune1+tune2) */
  int32_t n_mask_steps;          /* schedule positions < n_mask_steps are tune1 (mask = 1: no averaging) */
  int32_t schedule_total;        /* tune1 + tune2 + 1 (warmup.py:251,259) */
  float desired_energy_var_start;
  float desired_energy_var_end;  /* linear decay, or exponential with tau = total/4 when start > 2 */
  float trust_in_estimate;
  float decay_rate;              /* (n_eff - 1) / (n_eff + 1) */
  int32_t refresh;               /* mile_refresh */
  float *out_info;               /* [n_steps, E, 3] or NULL */
} mile_tune_args;

/* Optimizer of the warm-start stage (src/config/warmstart.py: OptimizerConfig -> optax; src/training/trainer.py:390-538). */
typedef enum mile_optimizer { MILE_OPT_SGD = 0, MILE_OPT_ADAM = 1, MILE_OPT_ADAMW = 2 } mile_optimizer;

/* One optimizer step of the deep-ensemble members on the current row window (round 3).  Update rules as optax writes them:
 * m = b1 m + (1 - b1) g, v = b2 v + (1 - b2) g^2, bias-corrected with step count t, update = lr (m^ / (sqrt(v^) + eps)
 * [+ weight_decay theta for adamw]); sgd: lr g.  g is the gradient of the batch-MEAN negative log-likelihood (no prior:
 * src/training/trainer.py:729-737). */
typedef struct mile_optim_args {
  int32_t kind;                  /* mile_optimizer */
  float learning_rate, b1, b2, eps, weight_decay;
  int64_t t;                     /* step count including this step (bias correction) */
  float *m, *v;                  /* [E, d] moments, caller-owned, updated in place (unused for sgd: may be NULL) */
  const uint8_t *active;         /* [E] 1 = still training; 0 = early-stopped: parameters AND moments stay frozen.  NULL = all */
  float *out_nll;                /* [E] batch-mean negative log-likelihood at the parameters BEFORE the update, or NULL */
} mile_optim_args;

typedef struct mile_sampler mile_sampler;

const char *mile_last_error(void);
int32_t mile_abi_version(void);

/* Replaces: config.kernel(logdensity_fn, ...) construction of the target, i.e.
 * ProbabilisticModel.__init__ (src/training/probabilistic.py:19-47) + Prior.from_name
 * (src/training/priors.py:67-91).  `device` is the HIP device ordinal. */
int32_t mile_create(const mile_model_spec *spec, int32_t device, mile_sampler **out);
int32_t mile_destroy(mile_sampler *s);

/* pytree_size(position) (blackjax.util; src/training/warmup.py:203). */
int64_t mile_param_count(const mile_sampler *s);

/* Offsets of layer `layer`'s bias and kernel inside the raveled vector (ravel_pytree order). */
int32_t mile_param_offsets(const mile_sampler *s, int32_t layer, int64_t *bias_off, int64_t *kernel_off);

/* Replaces: partial(log_unnormalized_posterior, x=train_x, y=train_y)
 * (src/training/trainer.py:576-580).  X [N, F] fp32; y [N] fp32 (regr) or int32 (classification). */
int32_t mile_set_data(mile_sampler *s, const float *X, const void *y, int64_t N, void *stream);

/* Restrict the likelihood to rows [begin, begin + count) of the training set for the following mile_logpost_grad calls
 * (count = 0: all rows again).  Replaces the minibatches of the warm-start stage: loader.iter(split='train', batch_size=...)
 * (src/dataset/tabular.py:170-212) feeding single_step_regr / single_step_class (src/training/trainer.py:706-760).
 * Supported by MILE_GRAD_GENERIC, the MFMA_NARROW, MFMA_W64, MFMA_WIDE and LENET kernels (mile_logpost_grad fails with MILE_ERR_STATE on
 * MFMA_W128_BF16 / GEMM_F32 under a window); the MCLMC path itself is full-batch (n_batches = 1). */
int32_t mile_set_row_window(mile_sampler *s, int64_t begin, int64_t count);

/* Replaces: one `single_step_regr` / `single_step_class` + `optimizer.update` + `optax.apply_updates` of the warm-start loop
 * (src/training/trainer.py:706-760, 430-470) for all E members at once: the likelihood gradient of the rows selected by
 * mile_set_row_window (the minibatch; all rows without a window) from the grad kernel, then ONE fused launch that forms the
 * batch-mean NLL gradient from the partial-gradient slabs, updates the moments and the parameters in place.  theta [E, d].
 * Grad kernels with row windows only (see mile_set_row_window). */
int32_t mile_warmstart_step(mile_sampler *s, float *theta, int32_t E, const mile_optim_args *args, void *stream);

/* Size the internal workspace (partial-gradient slabs etc.) for ensembles of up to E
 * particles.  Allocation happens here, never inside a launch call. */
int32_t mile_reserve(mile_sampler *s, int32_t E);

/* Select the grad kernel (default AUTO). */
int32_t mile_set_grad_kernel(mile_sampler *s, int32_t which);
int32_t mile_get_grad_kernel(const mile_sampler *s);

/* Replaces: jax.value_and_grad(logdensity_fn)(position) as used inside blackjax
 * (integrators.py position update; mclmc.init).  theta [E, d] -> logp [E], grad [E, d]. */
int32_t mile_logpost_grad(mile_sampler *s, const float *theta, int32_t E, float *logp, float *grad,
                          void *stream);

/* Replaces: blackjax.mcmc.mclmc.init(position, logdensity_fn, rng_key)
 * (src/training/warmup.py:539-541).  Fills state->momentum = z/|z| with z = `noise` [E, d] if
 * non-NULL, else Philox(seed, particle id, step 0, stage 2); evaluates logdensity and its
 * gradient at state->position (which the caller has filled). */
int32_t mile_init(mile_sampler *s, mile_state *state, const float *noise, uint64_t seed,
                  const int32_t *particle_ids, void *stream);

/* Replaces: the scan of sampler.step(rng_key, state) at src/training/sampling.py:140-178
 * (blackjax.mclmc(...).step == build_kernel(...)(rng_key, state, L, step_size),
 * src/training/warmup.py:286-291,427-432).  Advances `state` in place by n_steps. */
int32_t mile_step(mile_sampler *s, mile_state *state, const mile_step_args *args, void *stream);

/* Replaces: the lax.scan over `step` in make_L_step_size_adaptation.run_steps
 * (src/training/warmup.py:352-363), i.e. phases 1+2 of mclmc_find_L_and_step_size, on the device.
 * Advances `state` in place.  d <= 16384: the tuner runs inside the record-point update kernel; beyond that each
 * step is an ordinary kernel step followed by one tuner launch (k_tune_post), same arithmetic. */
int32_t mile_tune(mile_sampler *s, mile_state *state, const mile_tune_args *args, void *stream);

/* Replaces: the per-sample forward pass + log_prob of the evaluation path, i.e. predict_from_samples /
 * pointwise_lppd (src/inference/evaluation.py:16-43, src/inference/metrics.py:247-294).
 * theta [S, d] (S posterior samples, any chains), X [N, F] and y [N] (fp32 or int32 labels) are device
 * pointers of a TEST set (independent of mile_set_data); out [S, N] = log p(y_n | x_n, theta_s). */
int32_t mile_pointwise_loglik(mile_sampler *s, const float *theta, int32_t S, const float *X, const void *y,
                              int64_t N, float *out, void *stream);

/* Counter-RNG words/normals exactly as the step kernels draw them (test hook). out [E, d]. */
int32_t mile_debug_noise(mile_sampler *s, uint64_t seed, const int32_t *particle_ids, int32_t E,
                         int64_t step, int32_t stage, float *out, void *stream);

/* Introspection for bench.py's roofline: name, workgroups and LDS bytes of the grad kernel
 * that the current configuration launches for E particles. */
int32_t mile_grad_launch_info(const mile_sampler *s, int32_t E, int32_t *grid_x, int32_t *grid_y,
                              int32_t *block, int32_t *lds_bytes, char *name, int32_t name_len);

/* HIP-event timing of the grad kernel launches only (on `stream`): call begin, run steps,
 * call end -> total milliseconds and number of grad launches in between. */
int32_t mile_grad_timing_begin(mile_sampler *s);
int32_t mile_grad_timing_end(mile_sampler *s, float *total_ms, int32_t *n_launches);

#ifdef __cplusplus
}
#endif
#endif /* MILE_HIP_H_ */
