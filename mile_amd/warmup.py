"""MCLMC warm-up tuner (mirror of src/training/warmup.py:155-568), one tuner per chain,
all chains advanced together on the device.

The arithmetic per chain is exactly the reference's (phase 1+2: step size from the energy
variance with a decaying target and the streaming x / x^2 averages -> L; phase 3: L from the
FFT effective sample size).  The kernel step is libmile_hip's; the per-chain scalar updates
run on the device too: phases 1+2 inside libmile_hip (mile_tune: the step-size predictor,
handle_nans and the streaming averages are part of the record-point update kernel); a host-driven
loop of torch ops on [E] tensors is kept for d > 16384.
"""
from __future__ import annotations

import math
from typing import Callable, NamedTuple

import torch

from mile_amd.diagnostics import effective_sample_size
from mile_amd.engine import IntegratorState
from mile_amd.probabilistic import resolve_target
from mile_amd.tree import as_key, ravel_tree


class MCLMCAdaptationState(NamedTuple):
    """blackjax.adaptation.mclmc_adaptation.MCLMCAdaptationState, per chain."""

    L: torch.Tensor              # [E]
    step_size: torch.Tensor      # [E]
    sqrt_diag_cov: torch.Tensor  # [E, d]


class AdaptationResults(NamedTuple):
    state: IntegratorState
    parameters: MCLMCAdaptationState


class AdaptationAlgorithm(NamedTuple):
    run: Callable


def desired_energy_var(step: int, total_steps: int, start: float, end: float) -> float:
    """warmup.py:249-269: linear decay, or exponential (tau = total/4) when start > 2.0."""
    if start > 2.0:
        tau = total_steps / 4
        return start * math.exp(-step / tau) + end * (1 - math.exp(-step / tau))
    progress = min(step / total_steps, 1.0)
    return start - (start - end) * progress


def predictor_update(energy_change, step_size, time, x_average, step_size_max, *, dim, desired_var,
                     trust_in_estimate, decay_rate):
    """The step-size update of ``predictor`` (warmup.py:301-326) on [E] tensors.
    Returns (step_size, time, x_average)."""
    xi = energy_change.square() / (dim * desired_var) + 1e-8
    weight = torch.exp(-0.5 * (torch.log(xi) / (6.0 * trust_in_estimate)).square())
    x_average = decay_rate * x_average + weight * (xi / step_size.pow(6.0))
    time = decay_rate * time + weight
    new = (x_average / time).pow(-1.0 / 6.0)
    new = (new < step_size_max) * new + (new > step_size_max) * step_size_max
    return new, time, x_average


def handle_nans(prev: IntegratorState, nxt: IntegratorState, step_size, step_size_max, energy_change):
    """warmup.py:468-483 per chain: reject a step whose position is non-finite and cap the
    step size at 0.8 x the offending one.

    Almost every step is finite everywhere, and then the reference's where/nan_to_num select exactly ``nxt``:
    that case is detected with one max-|.| reduction per tensor (NaN and inf both propagate through it) and
    returns ``nxt`` without the seven [E, d] passes of the general path -- they cost more than the update
    kernels of a step on the wide nets that take this host-driven loop."""
    inf = float('inf')
    ok = torch.isfinite(torch.linalg.vector_norm(nxt.position, ord=inf, dim=1))
    rest = torch.stack([torch.linalg.vector_norm(nxt.momentum, ord=inf), torch.linalg.vector_norm(nxt.logdensity_grad, ord=inf),
                        torch.linalg.vector_norm(nxt.logdensity, ord=inf)])
    if bool(ok.all() & torch.isfinite(rest).all()):
        return ok, nxt, torch.nan_to_num(step_size_max), torch.nan_to_num(energy_change)
    okc = ok[:, None]
    state = IntegratorState(
        torch.where(okc, torch.nan_to_num(nxt.position), prev.position),
        torch.where(okc, torch.nan_to_num(nxt.momentum), prev.momentum),
        torch.where(ok, torch.nan_to_num(nxt.logdensity), prev.logdensity),
        torch.where(okc, torch.nan_to_num(nxt.logdensity_grad), prev.logdensity_grad))
    step_size_max = torch.where(ok, torch.nan_to_num(step_size_max), step_size * 0.8)
    energy_change = torch.where(ok, torch.nan_to_num(energy_change), torch.zeros_like(energy_change))
    return ok, state, step_size_max, energy_change


def streaming_average_update(value, weight_and_avg, weight, zero_prevention):
    """blackjax.util.streaming_average_update (call site warmup.py:343-348)."""
    W, avg = weight_and_avg
    Wn = W + weight
    sh = (-1,) + (1,) * (avg.ndim - 1)
    return Wn, (W.reshape(sh) * avg + weight.reshape(sh) * value) / (Wn + zero_prevention).reshape(sh)


def mclmc_find_L_and_step_size(eng, state: IntegratorState, rng_key, *, tune1_steps, tune2_steps, tune3_steps,
                               step_size_init, desired_energy_var_start, desired_energy_var_end,
                               trust_in_estimate, num_effective_samples, diagonal_preconditioning,
                               chain_ids=None, refresh='O-step-O', noise_fn=None, param_subset=None, force_host_loop=False,
                               Lfactor=0.4, fft_params_limit=2000, fft_samples_limit=10000):
    """warmup.py:155-228 for an ensemble.  ``noise_fn(i) -> [2, E, d]`` switches to explicit noise
    (parity tests); ``param_subset`` [E, <=2000] replaces jax.random.permutation in phase 3."""
    key = as_key(rng_key)
    part1_key, part2_key = key.split(2)
    dev = eng.device
    E, d = state.position.shape
    f32 = dict(dtype=torch.float32, device=dev)
    L = torch.full((E,), max(math.sqrt(d), 15.0), **f32)                 # warmup.py:205
    eps = torch.full((E,), float(step_size_init), **f32)
    sdc = None                                                            # ones(d)
    decay_rate = (num_effective_samples - 1.0) / (num_effective_samples + 1.0)
    total = tune1_steps + tune2_steps + 1

    def run_steps_device(state, eps, masks, seed, offset, chunk=256):
        """The scan of `step` on the device (mile_tune): adaptation state stays in HBM, no host loop
        per step.  Fresh adaptive state per call, exactly as run_steps does (warmup.py:352-363)."""
        tuner = {'step_size': eps.clone(), 'step_size_max': torch.full((E,), float('inf'), **f32),
                 'time': torch.zeros(E, **f32), 'x_average': torch.zeros(E, **f32),
                 'stream_weight': torch.zeros(E, **f32), 'stream_average': torch.zeros((E, 2, d), **f32)}
        state = IntegratorState(*(t.clone() for t in state))
        n_mask = sum(1 for m in masks if m == 1.0)
        assert all(m == 1.0 for m in masks[:n_mask]) and all(m == 0.0 for m in masks[n_mask:])
        done = 0
        while done < len(masks):
            c = min(chunk, len(masks) - done)
            z = torch.stack([noise_fn(offset + done + i) for i in range(c)]) if noise_fn is not None else None
            eng.tune(state, tuner, L_cur[0], c, schedule_step0=done, n_mask_steps=n_mask, schedule_total=total,
                     desired_energy_var_start=desired_energy_var_start, desired_energy_var_end=desired_energy_var_end,
                     trust_in_estimate=trust_in_estimate, decay_rate=decay_rate, noise=z, seed=seed,
                     step_offset=offset + done, particle_ids=chain_ids, refresh=refresh, sqrt_diag_cov=sdc_cur[0])
            done += c
        return state, tuner['step_size'], tuner['stream_average']

    def run_steps_host(state, eps, masks, seed, offset):
        time = torch.zeros(E, **f32)
        x_avg = torch.zeros(E, **f32)
        eps_max = torch.full((E,), float('inf'), **f32)
        W = torch.zeros(E, **f32)
        avg = torch.zeros((E, 2, d), **f32)
        for i, mask in enumerate(masks):
            z = noise_fn(offset + i)[None] if noise_fn is not None else None
            nxt, info, _ = eng.step(state, eps, L_cur[0], n_steps=1, noise=z, seed=seed, step_offset=offset + i,
                                    particle_ids=chain_ids, refresh=refresh, sqrt_diag_cov=sdc_cur[0])
            ok, state, eps_max, dE = handle_nans(state, nxt, eps, eps_max, info.energy_change[0])
            var = desired_energy_var(i, total, desired_energy_var_start, desired_energy_var_end)
            eps, time, x_avg = predictor_update(dE, eps, time, x_avg, eps_max, dim=d, desired_var=var,
                                                trust_in_estimate=trust_in_estimate, decay_rate=decay_rate)
            if mask == 0.0:   # (1 - mask) * success * step_size is zero during tune1
                x = state.position
                W, avg = streaming_average_update(torch.stack([x, x * x], dim=1), (W, avg),
                                                  weight=ok.to(torch.float32) * eps,
                                                  zero_prevention=torch.zeros(E, **f32))
        return state, eps, avg

    fft_warm = None
    if tune3_steps > 1:                                                   # rocFFT's ~2 s first-call cost, under phases 1+2
        from mile_amd.diagnostics import warm_fft
        fft_warm = warm_fft(min(tune3_steps, fft_samples_limit), dev)
    run_steps = run_steps_device if (eng.supports_device_tuner and not force_host_loop) else run_steps_host
    L_cur, sdc_cur = [L], [sdc]
    masks = [1.0] * tune1_steps + [0.0] * tune2_steps
    state, eps, avg = run_steps(state, eps, masks, part1_key.seed, 0)
    sqrt_diag_cov = torch.ones((E, d), **f32)
    if tune2_steps != 0:
        variances = avg[:, 1] - avg[:, 0].square()
        L = variances.sum(dim=1).sqrt()
        if diagonal_preconditioning:
            sqrt_diag_cov = variances.sqrt()
            # the re-adjustment keeps running with the phase-1 L (params.L is not replaced before run_steps,
            # warmup.py:389-401); L = sqrt(dim) is what gets returned (:392,403)
            L = torch.full((E,), math.sqrt(d), **f32)
            sdc_cur[0] = sqrt_diag_cov
            steps = tune2_steps // 3
            state, eps, _ = run_steps(state, eps, [1.0] * steps, part1_key.fold_in(1).seed, tune1_steps + tune2_steps)
    L_cur[0] = L

    if tune3_steps != 0:   # make_adaptation_L (warmup.py:408-465)
        P = min(d, fft_params_limit)
        if param_subset is None:
            g = torch.Generator().manual_seed(part2_key.seed & 0x7FFFFFFFFFFFFFFF)
            param_subset = torch.stack([torch.randperm(d, generator=g)[:P] for _ in range(E)]) if d > fft_params_limit \
                else torch.arange(d).expand(E, d)
        cols = torch.as_tensor(param_subset).to(dev)
        keep = None
        if tune3_steps > fft_samples_limit:
            keep = set(torch.linspace(0, tune3_steps - 1, fft_samples_limit).to(torch.int32).tolist())
        trace = []
        sd = sdc_cur[0] if diagonal_preconditioning else None
        state = IntegratorState(*(t.clone() for t in state))
        chunk = max(1, min(64, (1 << 28) // max(E * d, 1)))              # <= 1 GiB of kept positions per call
        done = 0
        while done < tune3_steps:                                        # scan of kernel steps, all positions kept
            c = min(chunk, tune3_steps - done)
            z = torch.stack([noise_fn(10 ** 9 + done + i) for i in range(c)]) if noise_fn is not None else None
            state, _, kept_pos = eng.step(state, eps, L, n_steps=c, noise=z, seed=part2_key.seed, step_offset=done,
                                          n_thinning=1, particle_ids=chain_ids, refresh=refresh, sqrt_diag_cov=sd,
                                          want_info=False, inplace=True)
            sub = torch.gather(kept_pos, 2, cols[None].expand(c, -1, -1))         # [c, E, P]
            if keep is not None:
                sel = [i for i in range(c) if (done + i) in keep]
                sub = sub[sel]
            trace.append(sub)
            done += c
        flat = torch.cat(trace, dim=0).permute(1, 0, 2).contiguous()       # [E, S, P]
        if fft_warm is not None:
            fft_warm.join()
        # one ESS per chain and parameter.  With a single chain per call every (chain, parameter) column is
        # independent, so groups of chains go through the estimator as extra columns (bounded workspace)
        # instead of E python-level calls.
        mean_ratio = torch.empty(E, **f32)
        S_, P_ = flat.shape[1], flat.shape[2]
        group = max(1, min(E, (1 << 26) // max(S_ * P_, 1)))
        for e0 in range(0, E, group):
            blk = flat[e0:e0 + group]                                        # [G, S, P]
            G = blk.shape[0]
            ess = effective_sample_size(blk.permute(1, 0, 2).reshape(1, S_, G * P_)).reshape(G, P_)
            mean_ratio[e0:e0 + G] = (tune3_steps / ess).mean(dim=1)
        L = Lfactor * eps * mean_ratio
    return state, MCLMCAdaptationState(L, eps, sqrt_diag_cov)


def custom_mclmc_warmup(logdensity_fn, diagonal_preconditioning: bool = True, desired_energy_var_start: float = 5e-4,
                        desired_energy_var_end: float = 5e-4, trust_in_estimate: float = 1.5,
                        num_effective_samples: int = 100, step_size_init: float = 0.005,
                        chain_ids=None, refresh: str = 'O-step-O') -> AdaptationAlgorithm:
    """warmup.py:486-568.  run(rng_key, position, num_steps) -> (state, MCLMCAdaptationState)."""
    model, x, y = resolve_target(logdensity_fn)

    def run(rng_key, position, num_steps: int = 1000) -> AdaptationResults:
        eng = model.engine(x, y)
        key = as_key(rng_key)
        flat = position if torch.is_tensor(position) else ravel_tree(model.spec, position)
        if flat.ndim == 1:
            flat = flat[None]
        state = eng.init(flat, seed=key.seed, particle_ids=chain_ids)     # same key reused, warmup.py:539-552
        phase_ratio = (0.8, 0.1, 0.1)
        state, params = mclmc_find_L_and_step_size(
            eng, state, key,
            tune1_steps=int(num_steps * phase_ratio[0]), tune2_steps=int(num_steps * phase_ratio[1]),
            tune3_steps=int(num_steps * phase_ratio[2]), step_size_init=step_size_init,
            desired_energy_var_start=desired_energy_var_start, desired_energy_var_end=desired_energy_var_end,
            trust_in_estimate=trust_in_estimate, num_effective_samples=num_effective_samples,
            diagonal_preconditioning=diagonal_preconditioning, chain_ids=chain_ids, refresh=refresh)
        return AdaptationResults(state, params)

    return AdaptationAlgorithm(run)
