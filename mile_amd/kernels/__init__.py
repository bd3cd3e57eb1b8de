"""Sampler plugin registry (mirror of src/training/kernels/__init__.py:14-20).

``KERNELS['mclmc']`` has blackjax.mclmc's factory shape:
    sampler = KERNELS['mclmc'](logdensity_fn, L=..., step_size=...)
    state = sampler.init(position, rng_key);  state, info = sampler.step(rng_key, state)
with every array carrying a leading ensemble axis (one row per chain), backed by
libmile_hip.so.  'mclmc_hip' is an alias.  nuts/hmc are outside the hot path.
"""
from __future__ import annotations

from typing import Callable, NamedTuple

import torch

from mile_amd.engine import IntegratorState, MCLMCInfo
from mile_amd.probabilistic import resolve_target
from mile_amd.tree import as_key, ravel_tree

__all__ = ['mclmc', 'KERNELS', 'WARMUP_KERNELS', 'SamplingAlgorithm']


class SamplingAlgorithm(NamedTuple):
    """blackjax.base.SamplingAlgorithm."""

    init: Callable
    step: Callable


def _flat(spec, position, device):
    flat = position if torch.is_tensor(position) else ravel_tree(spec, position)
    if flat.ndim == 1:
        flat = flat[None]
    return flat.to(device=device, dtype=torch.float32).contiguous()


def mclmc(logdensity_fn, L, step_size, integrator: str = 'isokinetic_mclachlan', sqrt_diag_cov=1.0,
          chain_ids=None, refresh: str = 'O-step-O') -> SamplingAlgorithm:
    """blackjax.mclmc(logdensity_fn, L, step_size, integrator=isokinetic_mclachlan, sqrt_diag_cov=1.0)
    as called at src/training/sampling.py:133,173 (``config.kernel(unnorm_log_posterior, **parameters)``).

    L, step_size: scalars or [E] per-chain values (the reference tunes every chain separately,
    sampling.py:92-97).  chain_ids: global chain numbers keying the RNG streams.
    """
    if integrator != 'isokinetic_mclachlan':
        raise NotImplementedError('only the isokinetic McLachlan integrator is implemented')
    model, x, y = resolve_target(logdensity_fn)
    eng = model.engine(x, y)
    sdc = None if (not torch.is_tensor(sqrt_diag_cov) and float(sqrt_diag_cov) == 1.0) else sqrt_diag_cov

    def init(position, rng_key) -> IntegratorState:
        key = as_key(rng_key)
        return eng.init(_flat(model.spec, position, eng.device), seed=key.seed, particle_ids=chain_ids)

    def step(rng_key, state: IntegratorState, step_index: int = 0):
        """One kernel step.  The noise stream is Philox(rng_key.seed; chain id, step_index)."""
        key = as_key(rng_key)
        s = sdc
        if s is not None and torch.as_tensor(s).ndim < 2:
            s = torch.as_tensor(s, dtype=torch.float32, device=eng.device).expand(state.position.shape).contiguous()
        new, info, _ = eng.step(state, step_size, L, n_steps=1, seed=key.seed, step_offset=step_index,
                                particle_ids=chain_ids, refresh=refresh, sqrt_diag_cov=s)
        return new, MCLMCInfo(info.logdensity[0], info.kinetic_change[0], info.energy_change[0])

    return SamplingAlgorithm(init, step)


KERNELS: dict[str, Callable[..., SamplingAlgorithm]] = {
    'mclmc': mclmc,
    'mclmc_hip': mclmc,
}

WARMUP_KERNELS: dict[str, Callable[..., SamplingAlgorithm]] = {}
