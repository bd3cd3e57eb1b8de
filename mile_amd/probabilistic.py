"""ProbabilisticModel: the BNN target density (mirror of src/training/probabilistic.py).

In the reference this object closes over a Flax module and is differentiated by JAX.
Here it carries the model SPEC; ``log_unnormalized_posterior`` bound to (x, y) is what the
kernel factory accepts as ``logdensity_fn`` -- it is evaluated by the HIP library, never in
Python, so a bound target can cross the C ABI (SURVEY 8b "Factory").
"""
from __future__ import annotations

import dataclasses
from functools import partial

import torch

from mile_amd.priors import Prior
from mile_amd.spec import LeNetSpec, ModelSpec

TASK_ALIASES = {'regr': 'regr', 'regression': 'regr', 'class': 'classification', 'classification': 'classification'}


class ProbabilisticModel:
    """src/training/probabilistic.py:16-138."""

    def __init__(self, module, params=None, prior: Prior | None = None, task: str = 'regr', n_batches: int = 1,
                 grad_kernel: str = 'auto'):
        """``module``: anything with in_features / hidden_structure / activation (an FCNConfig
        plus the feature count, or a ModelSpec)."""
        prior = prior or Prior.from_name('StandardNormal')
        task = TASK_ALIASES[str(task)]
        if n_batches != 1:
            raise NotImplementedError('Mini-Batch Sampling not yet implemented.')  # trainer.py:591-592
        if isinstance(module, LeNetSpec):
            self.spec = dataclasses.replace(module, task=task, prior=prior.name, prior_loc=prior.loc, prior_scale=prior.scale)
        if isinstance(module, (ModelSpec, LeNetSpec)):
            base = module
        else:
            base = ModelSpec(in_features=module.in_features, hidden_structure=tuple(module.hidden_structure),
                             activation=str(getattr(module.activation, 'value', module.activation)),
                             use_bias=getattr(module, 'use_bias', True))
        if not isinstance(module, LeNetSpec):
            self.spec = ModelSpec(in_features=base.in_features, hidden_structure=base.hidden_structure,
                                  activation=base.activation, task=task, prior=prior.name,
                                  prior_loc=prior.loc, prior_scale=prior.scale, use_bias=base.use_bias)
        self.task = task
        self.module = module
        self.n_params = self.spec.n_params
        self.n_batches = n_batches
        self.prior = prior
        self.grad_kernel = grad_kernel   # 'auto' | 'generic' | 'mfma_w64' | 'mfma_w128_bf16' (extension)
        self._engines: dict = {}

    def __str__(self):
        return (f'{self.__class__.__name__}:\n | Task: {self.task}\n | Params: {self.n_params}'
                f' | Batches: {self.n_batches}\n | Prior: {self.prior.name}')

    @property
    def minibatch(self):
        return self.n_batches > 1

    # -- device evaluation ---------------------------------------------------------
    def engine(self, x, y, device=None):
        """The HIP engine for this target and data (cached per data identity and device)."""
        from mile_amd.engine import Engine
        dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        key = (id(x), id(y), str(dev))
        if key not in self._engines:
            self._engines[key] = (Engine(self.spec, x, y, device=dev, grad_kernel=self.grad_kernel), x, y)   # keep x, y alive: id() stays unique
        return self._engines[key][0]

    def log_prior(self, params) -> torch.Tensor:
        from mile_amd.tree import ravel_tree
        flat = params if torch.is_tensor(params) else ravel_tree(self.spec, params)
        return self.prior.log_prior(flat)

    def log_unnormalized_posterior(self, position, x, y, **kwargs):
        """probabilistic.py:115-138, evaluated on the GPU.  position: param tree or [E, d]."""
        from mile_amd.tree import ravel_tree
        eng = self.engine(x, y)
        flat = position if torch.is_tensor(position) else ravel_tree(self.spec, position)
        squeeze = flat.ndim == 1
        logp, _ = eng.logpost_grad(flat.reshape(-1, self.n_params))
        return logp[0] if squeeze else logp

    def log_likelihood(self, params, x, y, **kwargs):
        return self.log_unnormalized_posterior(params, x, y) - self.log_prior(params).to(self.engine(x, y).device)

    def bind(self, x, y):
        """partial(self.log_unnormalized_posterior, x=x, y=y) as trainer.py:576-580 builds it."""
        return partial(self.log_unnormalized_posterior, x=x, y=y)


def resolve_target(logdensity_fn):
    """(ProbabilisticModel, x, y) behind a bound ``logdensity_fn``.

    An arbitrary Python callable cannot be differentiated on the device: only targets built
    from ProbabilisticModel (as BDETrainer.start_sampling builds them) are accepted.
    """
    fn = logdensity_fn
    if isinstance(fn, partial) and getattr(fn.func, '__self__', None) is not None \
            and isinstance(fn.func.__self__, ProbabilisticModel):
        kw = fn.keywords or {}
        if 'x' in kw and 'y' in kw:
            return fn.func.__self__, kw['x'], kw['y']
    raise TypeError('logdensity_fn must be partial(ProbabilisticModel.log_unnormalized_posterior, x=..., y=...) '
                    '(or ProbabilisticModel.bind(x, y)): the MI355X sampler evaluates the target natively and '
                    'cannot differentiate an arbitrary Python callable.')
