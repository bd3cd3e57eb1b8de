"""Priors over the raveled parameter vector (mirror of src/training/priors.py).

The log-prior and its gradient are evaluated on the device inside the integrator kernel
(mile_update.h); the callables here exist for API parity and for host-side checks.
"""
from __future__ import annotations

import math
from typing import Callable, NamedTuple

import torch


class PriorDist:
    """src/training/priors.py:24-65."""

    NORMAL = 'Normal'
    StandardNormal = 'StandardNormal'
    LAPLACE = 'Laplace'
    ALL = (NORMAL, StandardNormal, LAPLACE)


class Prior(NamedTuple):
    """src/training/priors.py:60-91, plus the (loc, scale) the native boundary needs."""

    f_init: Callable
    log_prior: Callable
    name: str
    loc: float = 0.0
    scale: float = 1.0

    @classmethod
    def from_name(cls, name: str, **parameters):
        if name == PriorDist.StandardNormal:
            return cls(name=name, f_init=f_init_normal(), log_prior=log_prior_normal())
        if name == PriorDist.NORMAL:
            return cls(name=name, f_init=f_init_normal(**parameters), log_prior=log_prior_normal(**parameters),
                       loc=float(parameters.get('loc', 0.0)), scale=float(parameters.get('scale', 1.0)))
        if name == PriorDist.LAPLACE:
            return cls(name=name, f_init=f_init_laplace(**parameters), log_prior=log_prior_laplace(**parameters),
                       loc=float(parameters.get('loc', 0.0)), scale=float(parameters.get('scale', 1.0)))
        raise NotImplementedError(f'Prior Distribution for {name} is not yet implemented.')


def f_init_normal(loc: float = 0.0, scale: float = 1.0):
    """jinit.normal(stddev=scale); loc is ignored as in the reference (:97-99)."""
    def f_init(generator, shape, dtype=torch.float32):
        return torch.randn(shape, generator=generator, dtype=dtype) * scale
    return f_init


def log_prior_normal(loc: float = 0.0, scale: float = 1.0):
    def log_prior(flat: torch.Tensor):
        t = (flat - loc) / scale
        return (-0.5 * t * t - math.log(scale) - 0.5 * math.log(2 * math.pi)).sum(-1)
    return log_prior


def f_init_laplace(loc: float = 0.0, scale: float = 1.0):
    def f_init(generator, shape, dtype=torch.float32):
        u = torch.rand(shape, generator=generator, dtype=dtype) - 0.5
        return loc - scale * torch.sign(u) * torch.log1p(-2 * u.abs())
    return f_init


def log_prior_laplace(loc: float = 0.0, scale: float = 1.0):
    def log_prior(flat: torch.Tensor):
        return (-(flat - loc).abs() / scale - math.log(2 * scale)).sum(-1)
    return log_prior
