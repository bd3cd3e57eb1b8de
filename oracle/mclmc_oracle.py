"""CPU oracle for the MILE MCLMC hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

PARITY UNPINNED: the reference (zhiyuan-yang/MILE) ships no tests, golden
vectors or fixtures for this path, and its numerical stack (jax 0.4.28,
blackjax 1.2.2, flax 0.8.5) is not installed in the build container, so this
restatement could not be checked against outputs of the reference itself.  It
is pinned only by (i) analytic known-answer tests, (ii) torch.autograd fp64 for
the gradient and (iii) algebraic invariants -- see tests/test_oracle_*.py.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.  The product path (``mile_amd``)
must never route through it.

What is restated, and from where (paths relative to /root/reference):

* Dense stack                     src/flax_building_blocks/basic.py:42-61
                                  src/models/tabular/fcn.py:16-28
* log-likelihood / posterior      src/training/probabilistic.py:68-138
* priors                          src/training/priors.py:101-128
* raveled parameter order         jax.flatten_util.ravel_pytree (sorted dict keys),
                                  used at src/training/priors.py:105,
                                  src/training/warmup.py:341,442
* MCLMC kernel                    blackjax==1.2.2 (THIRD PARTY, pinned in
                                  pyproject.toml:12 / poetry.lock:358-359; source not
                                  in the container).  Restated from its published
                                  algorithm: mcmc/mclmc.py (init, build_kernel),
                                  mcmc/integrators.py (isokinetic_mclachlan,
                                  esh_dynamics_momentum_update_one_step,
                                  partially_refresh_momentum,
                                  with_isokinetic_maruyama).  Call sites:
                                  src/training/warmup.py:286-291,427-432,524-531,539-541
                                  src/training/kernels/__init__.py:14-18
* warm-up tuner                   src/training/warmup.py:155-483 (in-repo, exact)
* streaming_average_update        blackjax.util (call site warmup.py:343-348)
* effective_sample_size           blackjax.diagnostics (call site warmup.py:458)
* sampling loop / thinning        src/training/sampling.py:140-196
* LPPD                            src/inference/metrics.py:247-312,428-446

Everything is vectorised over a leading ensemble axis E (one row per chain /
particle) and is dtype-generic: float64 is the checker, float32 is what
``bench.py`` times as the CPU baseline.

RNG: the reference draws noise with JAX threefry, which cannot be reproduced
without JAX.  All stochastic functions here therefore take the Gaussian noise
``z`` as an explicit input.  ``philox_normal`` is the counter-based generator
that the HIP path uses in throughput mode (Philox4x32-10 + Box-Muller), restated
so tests can check the device bits.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Sequence

import numpy as np

# --------------------------------------------------------------------------
# Model spec and the raveled parameter layout
# --------------------------------------------------------------------------

ACTIVATIONS = ('relu', 'tanh', 'sigmoid')  # subset of src/config/models/base.py:25-39
TASKS = ('regr', 'classification')         # src/config/data.py Task values
PRIORS = ('Normal', 'Laplace')             # src/training/priors.py:47-49


@dataclass(frozen=True)
class ModelSpec:
    """What crosses the native boundary instead of a Python ``logdensity_fn``.

    ``hidden_structure`` has the reference's meaning (src/config/models/fcn.py:
    13-20): one entry per Dense layer, the LAST entry is the output layer.
    """

    in_features: int
    hidden_structure: tuple[int, ...]
    activation: str = 'relu'
    task: str = 'regr'
    prior: str = 'Normal'
    prior_loc: float = 0.0
    prior_scale: float = 1.0
    use_bias: bool = True

    def __post_init__(self):
        assert self.activation in ACTIVATIONS, self.activation
        assert self.task in TASKS, self.task
        assert self.prior in PRIORS, self.prior
        assert len(self.hidden_structure) >= 1

    @property
    def layer_dims(self) -> list[tuple[int, int]]:
        dims, fin = [], self.in_features
        for w in self.hidden_structure:
            dims.append((fin, int(w)))
            fin = int(w)
        return dims

    @property
    def n_params(self) -> int:
        return sum(i * o + (o if self.use_bias else 0) for i, o in self.layer_dims)


def layer_order(n_layers: int) -> list[int]:
    """Layer visiting order of ravel_pytree: dict keys sorted as STRINGS.

    'layer10' < 'layer2' lexicographically, so for >= 11 layers the raveled
    order is not the natural one (param vector layout).
    """
    return sorted(range(n_layers), key=lambda i: f'layer{i}')


def param_slices(spec: ModelSpec) -> list[dict]:
    """Offsets of every leaf inside the raveled vector.

    Per layer (in ``layer_order``): bias[out] then kernel[in, out] row-major
    ('bias' < 'kernel').  Returned list is indexed by natural layer number.
    """
    out = [None] * len(spec.layer_dims)
    off = 0
    for li in layer_order(len(spec.layer_dims)):
        fin, fout = spec.layer_dims[li]
        ent = {'layer': li, 'in': fin, 'out': fout}
        if spec.use_bias:
            ent['bias'] = (off, off + fout)
            off += fout
        else:
            ent['bias'] = None
        ent['kernel'] = (off, off + fin * fout)
        off += fin * fout
        out[li] = ent
    assert off == spec.n_params
    return out


def flattened_keys(spec: ModelSpec, root: str = 'fcn') -> list[str]:
    """Dotted leaf names in pytree order (src/utils.py:50-70 on the FCN tree)."""
    keys = []
    for li in layer_order(len(spec.layer_dims)):
        if spec.use_bias:
            keys.append(f'{root}.layer{li}.bias')
        keys.append(f'{root}.layer{li}.kernel')
    return keys


def unravel(spec: ModelSpec, theta: np.ndarray) -> list[tuple[np.ndarray, np.ndarray | None]]:
    """theta[..., d] -> [(kernel[..., in, out], bias[..., out] | None)] per natural layer."""
    lead = theta.shape[:-1]
    res = []
    for ent in param_slices(spec):
        k0, k1 = ent['kernel']
        W = theta[..., k0:k1].reshape(*lead, ent['in'], ent['out'])
        b = None
        if ent['bias'] is not None:
            b0, b1 = ent['bias']
            b = theta[..., b0:b1]
        res.append((W, b))
    return res


# --------------------------------------------------------------------------
# Dense stack, likelihood, prior -- value and analytic gradient
# --------------------------------------------------------------------------

def _act(name: str, z: np.ndarray) -> np.ndarray:
    if name == 'relu':
        return np.maximum(z, 0)
    if name == 'tanh':
        return np.tanh(z)
    if name == 'sigmoid':
        return 1.0 / (1.0 + np.exp(-z))
    raise NotImplementedError(name)


def _act_grad(name: str, z: np.ndarray, h: np.ndarray) -> np.ndarray:
    """d act / d z given pre-activation z and output h.  ReLU'(0) = 0 as in JAX."""
    if name == 'relu':
        return (z > 0).astype(z.dtype)
    if name == 'tanh':
        return 1.0 - h * h
    if name == 'sigmoid':
        return h * (1.0 - h)
    raise NotImplementedError(name)


def mlp_forward(spec: ModelSpec, theta: np.ndarray, X: np.ndarray, keep: bool = False):
    """FullyConnected.__call__ (basic.py:42-61) for an ensemble.

    theta [E, d], X [N, F] -> out [E, N, out_width].  Activation between layers,
    none after the last (FCN passes last_layer_activation=None, fcn.py:22).
    """
    layers = unravel(spec, theta)
    h = np.broadcast_to(X.astype(theta.dtype), (theta.shape[0],) + X.shape)
    zs, hs = [], [h]
    for li, (W, b) in enumerate(layers):
        z = h @ W
        if b is not None:
            z = z + b[:, None, :]
        if li < len(layers) - 1:
            h = _act(spec.activation, z)
        else:
            h = z
        if keep:
            zs.append(z)
            hs.append(h)
    if keep:
        return h, zs, hs
    return h


_LOG_SQRT_2PI = 0.5 * math.log(2.0 * math.pi)


def pointwise_loglik(spec: ModelSpec, out: np.ndarray, y: np.ndarray):
    """Per-row log-likelihood and its gradient w.r.t. the network output.

    regr (probabilistic.py:92-100): norm.logpdf(y; loc=out[...,0],
    scale=clip(exp(out[...,1]), 1e-6, 1e6)).
    classification (probabilistic.py:101-109): log_softmax(out)[y].
    Returns (ll [E, N], dout [E, N, C]); NaN rows are zeroed in both (nansum).
    """
    dt = out.dtype
    if spec.task == 'regr':
        mu, s = out[..., 0], out[..., 1]
        with np.errstate(over='ignore', invalid='ignore'):
            es = np.exp(s)
            sigma = np.clip(es, 1e-6, 1e6).astype(dt)
            unclipped = ((es > 1e-6) & (es < 1e6)).astype(dt)
            r = (y.astype(dt)[None, :] - mu) / sigma
            ll = -0.5 * r * r - np.log(sigma) - dt.type(_LOG_SQRT_2PI)
            dmu = r / sigma
            ds = (r * r - 1.0) * unclipped
        dout = np.stack([dmu, ds], axis=-1)
    else:
        with np.errstate(over='ignore', invalid='ignore'):
            m = out.max(axis=-1, keepdims=True)
            e = np.exp(out - m)
            lse = m + np.log(e.sum(axis=-1, keepdims=True))
            logp = out - lse
            yi = y.astype(np.int64)
            ll = np.take_along_axis(logp, np.broadcast_to(yi[None, :, None], out.shape[:2] + (1,)), axis=-1)[..., 0]
            dout = -np.exp(logp)
            np.put_along_axis(dout, np.broadcast_to(yi[None, :, None], out.shape[:2] + (1,)),
                              np.take_along_axis(dout, np.broadcast_to(yi[None, :, None], out.shape[:2] + (1,)), axis=-1) + 1.0,
                              axis=-1)
    bad = np.isnan(ll)
    if bad.any():
        ll = np.where(bad, 0, ll)
        dout = np.where(bad[..., None], 0, dout)
    return ll.astype(dt), dout.astype(dt)


def log_prior(spec: ModelSpec, theta: np.ndarray):
    """priors.py:101-108 (Normal) / :121-128 (Laplace).  Returns (value[E], grad[E,d])."""
    dt = theta.dtype
    loc, scale = dt.type(spec.prior_loc), dt.type(spec.prior_scale)
    t = (theta - loc) / scale
    if spec.prior == 'Normal':
        val = (-0.5 * t * t - np.log(scale) - dt.type(_LOG_SQRT_2PI)).sum(axis=-1)
        grad = -t / scale
    else:
        val = (-np.abs(t) - np.log(2.0 * scale)).sum(axis=-1)
        grad = -np.sign(t) / scale
    return val.astype(dt), grad.astype(dt)


def logpost_and_grad(spec: ModelSpec, theta: np.ndarray, X: np.ndarray, y: np.ndarray,
                     n_batches: int = 1):
    """log_unnormalized_posterior (probabilistic.py:115-138) and its gradient.

    theta [E, d] -> (logp [E], grad [E, d]).  ``n_batches`` is hard-wired to 1
    by the reference (trainer.py:299-302) but kept for fidelity.
    """
    assert theta.ndim == 2 and theta.shape[1] == spec.n_params
    dt = theta.dtype
    out, zs, hs = mlp_forward(spec, theta, X, keep=True)
    ll, dout = pointwise_loglik(spec, out, y)
    lp, gp = log_prior(spec, theta)
    logp = lp + ll.sum(axis=-1) * dt.type(n_batches)

    grad = np.zeros_like(theta)
    ents = param_slices(spec)
    layers = unravel(spec, theta)
    dz = dout * dt.type(n_batches)
    for li in range(len(layers) - 1, -1, -1):
        W, _ = layers[li]
        ent = ents[li]
        hin = hs[li]                                   # [E, N, in]
        dW = np.swapaxes(hin, 1, 2) @ dz               # [E, in, out]
        k0, k1 = ent['kernel']
        grad[:, k0:k1] = dW.reshape(theta.shape[0], -1)
        if ent['bias'] is not None:
            b0, b1 = ent['bias']
            grad[:, b0:b1] = dz.sum(axis=1)
        if li > 0:
            dh = dz @ np.swapaxes(W, 1, 2)             # [E, N, in]
            dz = dh * _act_grad(spec.activation, zs[li - 1], hs[li])
    return logp.astype(dt), (grad + gp).astype(dt)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round to the nearest bfloat16 (8-bit significand, ties to even), returned in x's dtype."""
    f = np.ascontiguousarray(x, dtype=np.float32)
    u = f.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    r = u.astype(np.uint32).view(np.float32)
    return np.where(np.isfinite(f), r, f).astype(x.dtype)


def logpost_and_grad_bf16(spec: ModelSpec, theta: np.ndarray, X: np.ndarray, y: np.ndarray):
    """logpost_and_grad under the mixed-precision recipe of BASELINE config 3 ("bf16"): every operand of a
    matrix product (inputs, weights, activations, back-propagated signals) is rounded to bfloat16 where it
    enters the product, accumulation and everything elementwise (bias add, ReLU, likelihood, prior) stay in
    the working precision.  ReLU derivatives come from the un-rounded pre-activations.  This is the checker
    for the bf16-operand HIP kernel; same citations as logpost_and_grad."""
    assert spec.activation == 'relu'
    dt = theta.dtype
    q = bf16_round
    layers = unravel(spec, theta)
    E = theta.shape[0]
    h = np.broadcast_to(q(X.astype(dt)), (E,) + X.shape)
    zs, hs = [], [h]
    for li, (W, b) in enumerate(layers):
        z = h @ q(W) + b[:, None, :]
        zs.append(z)
        h = q(np.maximum(z, 0)) if li < len(layers) - 1 else z
        hs.append(h)
    ll, dout = pointwise_loglik(spec, h, y)
    lp, gp = log_prior(spec, theta)
    grad = np.zeros_like(theta)
    ents = param_slices(spec)
    dz = q(dout)
    for li in range(len(layers) - 1, -1, -1):
        W, _ = layers[li]
        k0, k1 = ents[li]['kernel']
        grad[:, k0:k1] = (np.swapaxes(hs[li], 1, 2) @ dz).reshape(E, -1)
        b0, b1 = ents[li]['bias']
        grad[:, b0:b1] = dz.sum(axis=1)
        if li > 0:
            dz = q((dz @ np.swapaxes(q(W), 1, 2)) * (zs[li - 1] > 0))
    return (lp + ll.sum(axis=-1)).astype(dt), (grad + gp).astype(dt)


# --------------------------------------------------------------------------
# MCLMC kernel (blackjax 1.2.2 semantics)
# --------------------------------------------------------------------------

MCLACHLAN_B1 = 0.1931833275037836
MCLACHLAN_COEFFS = (MCLACHLAN_B1, 0.5, 1.0 - 2.0 * MCLACHLAN_B1, 0.5, MCLACHLAN_B1)


@dataclass
class State:
    """blackjax IntegratorState for an ensemble: every field has a leading E axis."""

    position: np.ndarray        # [E, d]
    momentum: np.ndarray        # [E, d], unit rows
    logdensity: np.ndarray      # [E]
    logdensity_grad: np.ndarray  # [E, d]

    def copy(self):
        return State(self.position.copy(), self.momentum.copy(),
                     self.logdensity.copy(), self.logdensity_grad.copy())


@dataclass
class Info:
    """blackjax MCLMCInfo."""

    logdensity: np.ndarray      # [E]
    kinetic_change: np.ndarray  # [E]
    energy_change: np.ndarray   # [E]


def _normalize(x: np.ndarray, tol: float = 1e-13):
    """blackjax.mcmc.integrators.normalized_flatten_array: (x / |x| if |x| > tol else x, |x|).  NaN / inf norms behave as in
    jnp: a NaN norm fails the comparison (x is returned as it is), an infinite one divides (finite entries -> 0)."""
    with np.errstate(over='ignore', invalid='ignore', divide='ignore'):
        n = np.sqrt((x * x).sum(axis=-1, keepdims=True))
        ok = n > tol
        return np.where(ok, x / np.where(ok, n, 1), x), n


def esh_momentum_update(u, g, eps, coef, sqrt_diag_cov=1.0):
    """B-step (A.2).  u,g [E,d]; eps [E] -> (u' [E,d], velocity [E,d], dK [E]).

    As published (blackjax 1.2.x `esh_dynamics_momentum_update_one_step`) ONE helper, normalized_flatten_array with its
    `norm > 1e-13` guard, serves both the (preconditioned) gradient and the new momentum: a zero gradient -- e.g. one that
    handle_nans' nan_to_num produced from a NaN gradient (src/training/warmup.py:478-482) -- leaves e = g = 0, delta = 0 and the
    momentum unchanged instead of dividing 0 / 0 (VERDICT r2 weak #1b; rounds 1-2 divided unguarded)."""
    dt = u.dtype
    d = u.shape[-1]
    gs = g * sqrt_diag_cov
    with np.errstate(over='ignore', invalid='ignore', divide='ignore'):
        e, gnorm = _normalize(gs)
        ue = (u * e).sum(axis=-1, keepdims=True)
        delta = (eps[:, None] * dt.type(coef)) * gnorm / dt.type(d - 1)
        zeta = np.exp(-delta)
        uu = e * (1 - zeta) * (1 + zeta + ue * (1 - zeta)) + 2 * zeta * u
        un, _ = _normalize(uu)
        dK = (delta - dt.type(math.log(2.0)) + np.log(1 + ue + (1 - ue) * zeta * zeta)) * dt.type(d - 1)
    return un.astype(dt), (un * sqrt_diag_cov).astype(dt), dK[:, 0].astype(dt)


def partial_refresh(u, z, h, L):
    """O-step (A.5): u' = (u + nu z)/|u + nu z|, nu = sqrt((exp(2h/L) - 1)/d)."""
    dt = u.dtype
    d = u.shape[-1]
    nu = np.sqrt((np.exp(2.0 * h / L) - 1.0) / dt.type(d)).astype(dt)
    v = u + nu[:, None] * z
    return (v / np.sqrt((v * v).sum(axis=-1, keepdims=True))).astype(dt)


def mclachlan_step(logdensity_and_grad: Callable, state: State, eps, sqrt_diag_cov=1.0):
    """isokinetic_mclachlan (A.4): B(b1) A(1/2) B(1-2b1) A(1/2) B(b1)."""
    dt = state.position.dtype
    b1, a1, b2, a2, b3 = MCLACHLAN_COEFFS
    x, u, l, g = state.position, state.momentum, state.logdensity, state.logdensity_grad
    u, v, dK = esh_momentum_update(u, g, eps, b1, sqrt_diag_cov)
    x = x + (eps[:, None] * dt.type(a1)) * v
    l, g = logdensity_and_grad(x)
    u, v, dK2 = esh_momentum_update(u, g, eps, b2, sqrt_diag_cov)
    x = x + (eps[:, None] * dt.type(a2)) * v
    l, g = logdensity_and_grad(x)
    u, v, dK3 = esh_momentum_update(u, g, eps, b3, sqrt_diag_cov)
    return State(x, u, l, g), dK + dK2 + dK3


def mclmc_init(logdensity_and_grad: Callable, position: np.ndarray, z: np.ndarray) -> State:
    """blackjax.mcmc.mclmc.init (A.1): momentum = z/|z|."""
    if position.shape[-1] < 2:
        raise ValueError('The target distribution must have more than 1 dimension for MCLMC.')
    l, g = logdensity_and_grad(position)
    u = z / np.sqrt((z * z).sum(axis=-1, keepdims=True))
    return State(position.copy(), u.astype(position.dtype), l, g)


def mclmc_step(logdensity_and_grad: Callable, state: State, eps, L, z1, z2,
               sqrt_diag_cov=1.0, refresh: str = 'O-step-O') -> tuple[State, Info]:
    """One MCLMC kernel step (A.6).

    refresh='O-step-O' (default): O(eps/2; z1) . McLachlan . O(eps/2; z2), the
    with_isokinetic_maruyama form recalled for blackjax 1.2.2.
    refresh='step-O': McLachlan . O(eps; z2), the older 1.1.x form.  z1 unused.
    The placement could not be verified offline, hence the switch.
    """
    eps = np.asarray(eps, dtype=state.position.dtype)
    L = np.asarray(L, dtype=state.position.dtype)
    l_old = state.logdensity
    st = state
    if refresh == 'O-step-O':
        st = State(st.position, partial_refresh(st.momentum, z1, 0.5 * eps, L), st.logdensity, st.logdensity_grad)
        st, dK = mclachlan_step(logdensity_and_grad, st, eps, sqrt_diag_cov)
        st.momentum = partial_refresh(st.momentum, z2, 0.5 * eps, L)
    elif refresh == 'step-O':
        st, dK = mclachlan_step(logdensity_and_grad, st, eps, sqrt_diag_cov)
        st.momentum = partial_refresh(st.momentum, z2, eps, L)
    else:
        raise ValueError(refresh)
    info = Info(logdensity=st.logdensity, kinetic_change=dK, energy_change=dK - st.logdensity + l_old)
    return st, info


# --------------------------------------------------------------------------
# Counter-based noise used by the HIP path in throughput mode
# --------------------------------------------------------------------------

_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = np.uint32(0x9E3779B9)
_PHILOX_W1 = np.uint32(0xBB67AE85)


def philox4x32(counter: np.ndarray, key: np.ndarray, rounds: int = 10) -> np.ndarray:
    """Philox4x32-R (Salmon et al., SC'11).  counter [...,4] u32, key [...,2] u32."""
    c = np.array(counter, dtype=np.uint32, copy=True)
    k = np.array(np.broadcast_to(key, c.shape[:-1] + (2,)), dtype=np.uint32, copy=True)
    c0, c1, c2, c3 = (c[..., i].copy() for i in range(4))
    k0, k1 = k[..., 0].copy(), k[..., 1].copy()
    for _ in range(rounds):
        p0 = c0.astype(np.uint64) * _PHILOX_M0
        p1 = c2.astype(np.uint64) * _PHILOX_M1
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        with np.errstate(over='ignore'):
            k0 = (k0 + _PHILOX_W0).astype(np.uint32)
            k1 = (k1 + _PHILOX_W1).astype(np.uint32)
    return np.stack([c0, c1, c2, c3], axis=-1)


def philox_bits(seed: int, particle_ids: np.ndarray, step: int, stage: int, d: int) -> np.ndarray:
    """Raw 32-bit words for noise element (particle, i): counter=(i//4, particle,
    step, stage), key=(seed_lo, seed_hi), word i%4.  Returns [E, d] uint32."""
    nq = (d + 3) // 4
    E = len(particle_ids)
    ctr = np.zeros((E, nq, 4), dtype=np.uint32)
    ctr[..., 0] = np.arange(nq, dtype=np.uint32)[None, :]
    ctr[..., 1] = np.asarray(particle_ids, dtype=np.uint32)[:, None]
    ctr[..., 2] = np.uint32(step & 0xFFFFFFFF)
    ctr[..., 3] = np.uint32(stage)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    out = philox4x32(ctr, key)
    return out.reshape(E, nq * 4)[:, :d]


def philox_normal(seed: int, particle_ids: np.ndarray, step: int, stage: int, d: int,
                  dtype=np.float64) -> np.ndarray:
    """N(0,1) noise [E, d] from Philox words via Box-Muller on word pairs.

    Pair (w0, w1) of a quad -> (r cos t, r sin t); pair (w2, w3) likewise, with
    u1 = (w_even + 1) * 2^-32 in (0, 1], u2 = w_odd * 2^-32 in [0, 1),
    r = sqrt(-2 ln u1), t = 2 pi u2.  Depends only on (seed, GLOBAL particle id,
    step, stage, element), so results are independent of how particles are
    sharded over GPUs.
    """
    nq = (d + 3) // 4
    bits = philox_bits(seed, particle_ids, step, stage, nq * 4).reshape(len(particle_ids), nq, 2, 2)
    u1 = (bits[..., 0].astype(np.float64) + 1.0) * 2.0 ** -32
    u2 = bits[..., 1].astype(np.float64) * 2.0 ** -32
    if np.dtype(dtype) == np.float32:
        u1, u2 = u1.astype(np.float32), u2.astype(np.float32)
    r = np.sqrt(-2.0 * np.log(u1))
    t = (2.0 * math.pi) * u2
    z = np.stack([r * np.cos(t), r * np.sin(t)], axis=-1)
    return z.reshape(len(particle_ids), nq * 4)[:, :d].astype(dtype)


# --------------------------------------------------------------------------
# Sampling loop with thinning (sampling.py:140-196)
# --------------------------------------------------------------------------

def kept_indices(n_samples: int, n_thinning: int) -> np.ndarray:
    """Indices n for which sample_<n>.npz is written: idx % n_thinning == 0
    (sampling.py:163-165); n_thinning == 1 keeps every step (:112-121)."""
    idx = np.arange(n_samples, dtype=np.int32)
    return idx[(idx % np.int32(max(n_thinning, 1))) == 0]


def train_plan(n_chains: int, n_devices: int) -> list[np.ndarray]:
    """BDETrainer.train_plan (trainer.py:75-82): chain groups run sequentially."""
    if n_chains % n_devices:
        raise ValueError('n_chains must be divisible by the number of devices.'
                         f'{n_chains} % {n_devices} != 0.')
    return np.array_split(np.arange(n_chains), n_chains // n_devices)


def sample_chain(logdensity_and_grad: Callable, state: State, eps, L, noise_fn: Callable,
                 n_samples: int, n_thinning: int, refresh: str = 'O-step-O'):
    """scan of sampler.step with the thinning predicate.  noise_fn(idx) -> (z1, z2).
    Returns (final state, kept positions [K, E, d], kept idx [K] int32)."""
    kept, kept_idx = [], []
    for idx in range(n_samples):
        z1, z2 = noise_fn(idx)
        state, _ = mclmc_step(logdensity_and_grad, state, eps, L, z1, z2, refresh=refresh)
        if idx % max(n_thinning, 1) == 0:
            kept.append(state.position.copy())
            kept_idx.append(idx)
    return state, np.stack(kept), np.asarray(kept_idx, dtype=np.int32)


# --------------------------------------------------------------------------
# Warm-up tuner (warmup.py:155-483)
# --------------------------------------------------------------------------

def desired_energy_var(step, total_steps, start, end):
    """warmup.py:249-269.  total_steps = tune1 + tune2 + 1."""
    if start > 2.0:
        tau = total_steps / 4
        return start * np.exp(-step / tau) + end * (1 - np.exp(-step / tau))
    progress = min(step / total_steps, 1.0)
    return start - (start - end) * progress


def handle_nans(prev: State, nxt: State, step_size, step_size_max, energy_change):
    """warmup.py:468-483, per particle.  Returns (success[E], state, step_size_max, energy_change)."""
    ok = np.all(np.isfinite(nxt.position), axis=-1)
    fmax = np.finfo(nxt.position.dtype).max

    def sel(new, old):
        new = np.nan_to_num(new, nan=0.0, posinf=fmax, neginf=-fmax)
        m = ok.reshape((-1,) + (1,) * (new.ndim - 1))
        return np.where(m, new, old)

    st = State(sel(nxt.position, prev.position), sel(nxt.momentum, prev.momentum),
               sel(nxt.logdensity, prev.logdensity), sel(nxt.logdensity_grad, prev.logdensity_grad))
    return ok, st, sel(step_size_max, step_size * 0.8), sel(energy_change, np.zeros_like(energy_change))


def streaming_average_update(value, weight_and_avg, weight, zero_prevention):
    """blackjax.util.streaming_average_update (A.7), per particle.
    value/avg [E, ...]; weight, zero_prevention [E]."""
    W, avg = weight_and_avg
    Wn = W + weight
    sh = (-1,) + (1,) * (avg.ndim - 1)
    avg_n = (W.reshape(sh) * avg + weight.reshape(sh) * value) / (Wn + zero_prevention).reshape(sh)
    return Wn, avg_n


@dataclass
class TunerResult:
    state: State
    L: np.ndarray
    step_size: np.ndarray
    sqrt_diag_cov: np.ndarray
    trace: dict = field(default_factory=dict)


@dataclass
class AdaptiveState:
    """The scan carry of make_L_step_size_adaptation.step besides (state, params): adaptive_state = (time, x_average,
    step_size_max) and streaming_avg = (weight, [mean x, mean x^2]) (warmup.py:283,331,356-361), one per chain."""

    time: np.ndarray            # [E]
    x_average: np.ndarray       # [E]
    step_size_max: np.ndarray   # [E]
    W: np.ndarray               # [E]
    avg: np.ndarray             # [E, 2, d]

    @staticmethod
    def fresh(E: int, d: int, dt) -> 'AdaptiveState':
        return AdaptiveState(np.zeros(E, dtype=dt), np.zeros(E, dtype=dt), np.full(E, np.inf, dtype=dt),
                             np.zeros(E, dtype=dt), np.zeros((E, 2, d), dtype=dt))

    def copy(self):
        return AdaptiveState(*(np.array(v, copy=True) for v in (self.time, self.x_average, self.step_size_max, self.W, self.avg)))


def predictor_update(dE, eps, ad: AdaptiveState, *, dim, var, trust_in_estimate, decay):
    """The step-size arithmetic of `predictor` (warmup.py:301-326) on the handle_nans'ed energy change, in dE's dtype.
    Returns (new step size, xi, weight); updates ad.time / ad.x_average in place.  Written as the reference writes it,
    0 * inf corner and the overflow of xi / eps^6 in float32 included (profiles/r03/01_*: that overflow is what turns
    `step_size` into 0 for good once a chain has been thrown into a region with |grad| ~ 1e11)."""
    dt = dE.dtype
    with np.errstate(over='ignore', invalid='ignore', divide='ignore'):
        xi = dE * dE / (dt.type(dim) * dt.type(var)) + dt.type(1e-8)
        w = np.exp(-0.5 * np.square(np.log(xi) / dt.type(6.0 * trust_in_estimate)))
        ad.x_average = (dt.type(decay) * ad.x_average + w * (xi / np.power(eps, dt.type(6.0)))).astype(dt)
        ad.time = (dt.type(decay) * ad.time + w).astype(dt)
        new = np.power(ad.x_average / ad.time, dt.type(-1.0 / 6.0))
        new = ((new < ad.step_size_max) * new + (new > ad.step_size_max) * ad.step_size_max).astype(dt)
    return new, xi, w


def tuner_step(logdensity_and_grad, state: State, eps, L, sdc, z1, z2, ad: AdaptiveState, *, mask, var,
               trust_in_estimate, decay, refresh='O-step-O'):
    """One iteration of make_L_step_size_adaptation.step (warmup.py:328-350): kernel step, handle_nans, predictor,
    streaming averages.  Returns (state, new step size, success [E], info of the raw kernel step); `ad` is updated in place."""
    dt = state.position.dtype
    E, d = state.position.shape
    nxt, info = mclmc_step(logdensity_and_grad, state, eps, L, z1, z2, sdc, refresh)
    ok, state, ad.step_size_max, dE = handle_nans(state, nxt, eps, ad.step_size_max, info.energy_change)
    eps_new, _, _ = predictor_update(dE, eps, ad, dim=d, var=var, trust_in_estimate=trust_in_estimate, decay=decay)
    x = state.position
    ad.W, ad.avg = streaming_average_update(
        np.stack([x, x * x], axis=1), (ad.W, ad.avg),
        weight=((1 - mask) * ok * eps_new).astype(dt), zero_prevention=np.full(E, mask, dtype=dt))
    return state, eps_new, ok, info


def tune_phase12(logdensity_and_grad, state: State, noise_fn, tune1: int, tune2: int, *,
                 step_size_init, desired_energy_var_start, desired_energy_var_end,
                 trust_in_estimate, num_effective_samples, diagonal_preconditioning=False,
                 refresh='O-step-O', noise_offset=0, record=False) -> TunerResult:
    """make_L_step_size_adaptation (warmup.py:231-405), every chain tuned separately."""
    dt = state.position.dtype
    E, d = state.position.shape
    L = np.full(E, max(math.sqrt(d), 15.0), dtype=dt)              # warmup.py:205
    eps = np.full(E, step_size_init, dtype=dt)
    sdc = np.ones((E, d), dtype=dt)
    decay = dt.type((num_effective_samples - 1.0) / (num_effective_samples + 1.0))
    total = tune1 + tune2 + 1
    trace = {'step_size': [], 'energy_change': [], 'success': []} if record else {}

    def run_steps(state, eps, masks, offset):
        ad = AdaptiveState.fresh(E, d, dt)                         # fresh per run_steps call (warmup.py:352-363)
        for i, mask in enumerate(masks):
            z1, z2 = noise_fn(offset + i)
            var = desired_energy_var(i, total, desired_energy_var_start, desired_energy_var_end)
            state, eps, ok, info = tuner_step(logdensity_and_grad, state, eps, L_cur[0], sdc_cur[0], z1, z2, ad, mask=mask,
                                              var=var, trust_in_estimate=trust_in_estimate, decay=decay, refresh=refresh)
            if record:
                trace['step_size'].append(eps.copy())
                trace['energy_change'].append(info.energy_change.copy())
                trace['success'].append(ok.copy())
        return state, eps, ad.avg

    L_cur, sdc_cur = [L], [sdc]
    masks = [1.0] * tune1 + [0.0] * tune2
    state, eps, avg = run_steps(state, eps, masks, noise_offset)
    if tune2 != 0:
        var = avg[:, 1] - np.square(avg[:, 0])
        with np.errstate(invalid='ignore'):
            L = np.sqrt(var.sum(axis=-1)).astype(dt)
        if diagonal_preconditioning:
            with np.errstate(invalid='ignore'):
                sdc = np.sqrt(var).astype(dt)
            # warmup.py:389-401: only `params.sqrt_diag_cov` is replaced before the re-adjustment run, so its kernel steps
            # still use params.L = max(sqrt(d), 15) of phase 1; sqrt(d) is the RETURNED L only (differs for d < 225).
            L = np.full(E, math.sqrt(d), dtype=dt)
            sdc_cur[0] = sdc
            steps = tune2 // 3
            state, eps, _ = run_steps(state, eps, [1.0] * steps, noise_offset + tune1 + tune2)
    return TunerResult(state, L, eps, sdc, trace)


def effective_sample_size(x: np.ndarray) -> np.ndarray:
    """blackjax.diagnostics.effective_sample_size (A.8) for x[chains, samples, dims].

    Stan-style: FFT autocovariance, Geyer initial-positive then initial-monotone
    sequence on paired autocorrelations.  Restated from the published blackjax
    source (itself a port of the TFP/Stan estimator); call site warmup.py:458
    uses a single chain (C=1).
    """
    x = np.asarray(x)
    C, S = x.shape[0], x.shape[1]
    assert S > 1
    from scipy.fft import next_fast_len
    mean_chain = x.mean(axis=1, keepdims=True)
    xc = x - mean_chain
    m = next_fast_len(2 * S)
    f = np.fft.rfft(xc, n=m, axis=1)
    f = f * np.conjugate(f)
    acov = np.fft.irfft(f, n=m, axis=1)[:, :S] / S
    mean_acov = acov.mean(axis=0, keepdims=True)               # [1, S, ...]
    mean_var0 = mean_acov[:, :1] * S / (S - 1.0)
    weighted_var = mean_var0 * (S - 1.0) / S
    if C > 1:
        weighted_var = weighted_var + mean_chain.var(axis=0, ddof=1, keepdims=True)
    S_even = S - S % 2
    acov_tp1 = mean_acov[:, 1:S_even]
    rho = np.concatenate([np.ones_like(mean_var0), 1.0 - (mean_var0 - acov_tp1) / weighted_var], axis=1)
    rho = np.moveaxis(rho, 1, 0)                                # [S_even, 1, ...]
    rho_even, rho_odd = rho[0::2].copy(), rho[1::2].copy()
    T = rho_even.shape[0]

    mask0 = (rho_even + rho_odd) > 0.0
    mask = np.logical_and.accumulate(mask0, axis=0)            # carry_cond & mask_t
    tt = np.arange(T).reshape((-1,) + (1,) * (mask.ndim - 1))
    max_t = np.where(mask, tt, 0).max(axis=0)                   # last t with mask true (0 if none)
    rho_odd = np.where(mask, rho_odd, 0.0)
    # JAX semantics for index max_t+1 == T: gathers clamp to T-1, scatters are dropped.
    nxt = np.minimum(max_t + 1, T - 1)
    in_range = (max_t + 1) <= (T - 1)
    mask_even = mask.copy()
    take = np.take_along_axis(rho_even, nxt[None], axis=0)[0]
    upd = np.where(in_range, take > 0, np.take_along_axis(mask_even, nxt[None], axis=0)[0])
    np.put_along_axis(mask_even, nxt[None], upd[None], axis=0)
    rho_even = np.where(mask_even, rho_even, 0.0)

    rsum = rho_even + rho_odd
    upd_mask = np.zeros_like(rsum, dtype=bool)
    upd_val = np.zeros_like(rsum)
    prev = rsum[0]
    for t in range(T):
        um = rsum[t] > prev
        cur = np.where(um, prev, rsum[t])
        upd_mask[t], upd_val[t] = um, cur
        prev = cur
    rho_even_f = np.where(upd_mask, upd_val / 2.0, rho_even)
    rho_odd_f = np.where(upd_mask, upd_val / 2.0, rho_odd)

    ess_raw = C * S
    last = np.take_along_axis(rho_even_f, nxt[None], axis=0)[0]
    tau = -1.0 + 2.0 * (rho_even_f + rho_odd_f).sum(axis=0) - last
    tau = np.maximum(tau, 1.0 / np.log10(ess_raw))
    return np.squeeze(ess_raw / tau)


def tune_phase3(logdensity_and_grad, state: State, eps, L, noise_fn, tune3: int, *,
                sqrt_diag_cov=1.0, Lfactor=0.4, param_subset: Sequence[np.ndarray] | None = None,
                fft_params_limit=2000, fft_samples_limit=10000, refresh='O-step-O', noise_offset=0):
    """make_adaptation_L (warmup.py:408-465).  ``param_subset[e]`` replaces
    jax.random.permutation(key, d)[:2000] (JAX RNG is not reproducible here)."""
    E, d = state.position.shape
    samples = np.empty((tune3, E, d), dtype=state.position.dtype)
    for i in range(tune3):
        z1, z2 = noise_fn(noise_offset + i)
        state, _ = mclmc_step(logdensity_and_grad, state, eps, L, z1, z2, sqrt_diag_cov, refresh)
        samples[i] = state.position
    Lnew = np.empty(E, dtype=state.position.dtype)
    for e in range(E):
        flat = samples[:, e, :]
        if d > fft_params_limit:
            cols = param_subset[e] if param_subset is not None else np.arange(fft_params_limit)
            flat = flat[:, cols]
        if flat.shape[0] > fft_samples_limit:
            flat = flat[np.linspace(0, flat.shape[0] - 1, fft_samples_limit).astype(np.int32)]
        ess = effective_sample_size(flat[None].astype(np.float64))
        Lnew[e] = Lfactor * eps[e] * np.mean(tune3 / ess)
    return state, Lnew


def phase_steps(num_steps: int, ratio=(0.8, 0.1, 0.1)) -> tuple[int, int, int]:
    """int() truncation of warmup.py:555-557."""
    return int(num_steps * ratio[0]), int(num_steps * ratio[1]), int(num_steps * ratio[2])


# --------------------------------------------------------------------------
# LPPD (src/inference/metrics.py:247-312)
# --------------------------------------------------------------------------

def pointwise_lppd(spec: ModelSpec, lvals: np.ndarray, y: np.ndarray) -> np.ndarray:
    """lvals [C, S, N, out] -> [C, S, N] log predictive density of each sample."""
    C, S, N, O = lvals.shape
    ll, _ = pointwise_loglik_raw(spec, lvals.reshape(C * S, N, O), y)
    return ll.reshape(C, S, N)


def pointwise_loglik_raw(spec: ModelSpec, out: np.ndarray, y: np.ndarray):
    """As pointwise_loglik but WITHOUT the nansum zeroing (metrics.py uses
    numpyro Normal/Categorical.log_prob directly)."""
    dt = out.dtype
    if spec.task == 'regr':
        mu, s = out[..., 0], out[..., 1]
        sigma = np.clip(np.exp(s), 1e-6, 1e6).astype(dt)
        r = (y.astype(dt)[None, :] - mu) / sigma
        return (-0.5 * r * r - np.log(sigma) - dt.type(_LOG_SQRT_2PI)), None
    m = out.max(axis=-1, keepdims=True)
    logp = out - (m + np.log(np.exp(out - m).sum(axis=-1, keepdims=True)))
    yi = np.broadcast_to(y.astype(np.int64)[None, :, None], out.shape[:2] + (1,))
    return np.take_along_axis(logp, yi, axis=-1)[..., 0], None


def lppd(lppd_pointwise: np.ndarray) -> float:
    """metrics.py:297-312: mean_n( logsumexp_{c,s} l - log(C*S) )."""
    C, S, N = lppd_pointwise.shape
    flat = lppd_pointwise.reshape(C * S, N)
    m = flat.max(axis=0)
    lse = m + np.log(np.exp(flat - m).sum(axis=0))
    return float((lse - math.log(C * S)).mean())


# --------------------------------------------------------------------------
# Synthetic workloads of BASELINE.md section 3 (seeded, shared by tests/bench)
# --------------------------------------------------------------------------

def synthetic_problem(spec: ModelSpec, N: int, E: int, seed: int = 0, theta_scale: float = 0.1):
    """X ~ N(0,1); regr y from a fixed random 1-hidden-layer tanh teacher + 0.1 noise,
    z-scored; classification y ~ Categorical(softmax(X W*)); theta0 ~ N(0, scale^2);
    eps = 1e-2 (1 +- 5 %), L = sqrt(d) (1 +- 5 %) per particle."""
    rng = np.random.Generator(np.random.PCG64(seed))
    F = spec.in_features
    X = rng.standard_normal((N, F)).astype(np.float32)
    if spec.task == 'regr':
        Wt = rng.standard_normal((F, 16)) / math.sqrt(F)
        vt = rng.standard_normal(16) / 4.0
        y = np.tanh(X @ Wt) @ vt + 0.1 * rng.standard_normal(N)
        y = ((y - y.mean()) / y.std()).astype(np.float32)
    else:
        C = spec.hidden_structure[-1]
        Wt = rng.standard_normal((F, C))
        logits = X @ Wt
        g = rng.gumbel(size=logits.shape)
        y = np.argmax(logits + g, axis=-1).astype(np.int32)
    d = spec.n_params
    theta0 = (theta_scale * rng.standard_normal((E, d))).astype(np.float32)
    u0 = rng.standard_normal((E, d)).astype(np.float32)
    eps = (1e-2 * (1.0 + 0.05 * (2.0 * rng.random(E) - 1.0))).astype(np.float32)
    L = (math.sqrt(d) * (1.0 + 0.05 * (2.0 * rng.random(E) - 1.0))).astype(np.float32)
    return {'X': X, 'y': y, 'theta0': theta0, 'u0': u0, 'eps': eps, 'L': L}


CONFIGS = {
    # name: (in_features, hidden_structure, task, N, E)   -- SURVEY section 8 / Appendix C
    'stock': (5, (16, 16, 2), 'regr', 1052, 12),
    'B1': (5, (64, 64, 64, 2), 'regr', 1052, 16),
    'B2': (5, (64, 64, 64, 2), 'regr', 1052, 128),
    'B3': (9, (128, 128, 128, 2), 'regr', 36000, 512),
    'B4': (54, (256, 256, 256, 256, 7), 'classification', 232404, 128),
}


def config_spec(name: str) -> tuple[ModelSpec, int, int]:
    F, hs, task, N, E = CONFIGS[name]
    return ModelSpec(in_features=F, hidden_structure=hs, task=task), N, E
