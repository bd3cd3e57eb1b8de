"""Warm-start training of the deep ensemble the chains start from (mirror of src/training/trainer.py:330-538).

The reference trains every ensemble member with optax (AdamW by default) on minibatches of the mean negative
log-likelihood, one epoch = ceil(N / batch_size) optimizer steps, validation after each epoch, per-member early
stopping (`earlystop`, trainer.py:920-939).  Here the gradient comes from the HIP engine, which evaluates the
FULL-batch log-posterior of all members in one launch: the loop keeps the reference's optimizer, its number of
optimizer steps per epoch, its validation schedule and its early-stopping rule, but every step sees the whole
training set (deterministic; documented deviation -- there is no minibatch kernel on the MI355X path).
"""
from __future__ import annotations

import logging
import math

import torch

logger = logging.getLogger(__name__)


def earlystop(losses: torch.Tensor, patience: int) -> torch.Tensor:
    """trainer.py:920-939: stop a member when none of its last `patience` validation losses is below the one before
    them.  losses [E, n_epochs] -> bool [E]."""
    if losses.shape[-1] < patience + 1:
        return torch.zeros(losses.shape[0], dtype=torch.bool, device=losses.device)
    ref = losses[:, -(patience + 1)][:, None]
    return (losses[:, -patience:] >= ref).all(dim=1)


def prior_value_and_grad(prior, theta: torch.Tensor):
    """log prior [E] and its gradient [E, d] (src/training/priors.py:101-128), to take the prior back out of the
    engine's log-posterior gradient: the warm-start loss is the likelihood alone (trainer.py:729-737)."""
    t = (theta - prior.loc) / prior.scale
    if prior.name == 'Laplace':
        return prior.log_prior(theta), -torch.sign(t) / prior.scale
    return prior.log_prior(theta), -t / prior.scale


class _Optimizer:
    """optax.adamw / adam / sgd on a flat [E, d] tensor (update rules as in optax: bias-corrected moments,
    decoupled weight decay scaled by the learning rate)."""

    def __init__(self, name: str, params: dict, like: torch.Tensor):
        self.name = name.lower()
        self.lr = float(params.get('learning_rate', 1e-3))
        self.b1 = float(params.get('b1', 0.9))
        self.b2 = float(params.get('b2', 0.999))
        self.eps = float(params.get('eps', 1e-8))
        self.wd = float(params.get('weight_decay', 1e-4 if self.name == 'adamw' else 0.0))
        if self.name not in ('adamw', 'adam', 'sgd'):
            raise NotImplementedError(f'optimizer {name!r} (available: adamw, adam, sgd)')
        self.m = torch.zeros_like(like)
        self.v = torch.zeros_like(like)
        self.t = 0

    def step(self, theta: torch.Tensor, grad: torch.Tensor, active: torch.Tensor):
        """theta <- theta - update for the members with active[e]; moments of stopped members stay frozen."""
        self.t += 1
        a = active[:, None]
        if self.name == 'sgd':
            upd = self.lr * grad
        else:
            self.m = torch.where(a, self.b1 * self.m + (1 - self.b1) * grad, self.m)
            self.v = torch.where(a, self.b2 * self.v + (1 - self.b2) * grad * grad, self.v)
            mh = self.m / (1 - self.b1 ** self.t)
            vh = self.v / (1 - self.b2 ** self.t)
            upd = self.lr * (mh / (vh.sqrt() + self.eps) + (self.wd * theta if self.name == 'adamw' else 0.0))
        return torch.where(a, theta - upd, theta)


def train_deep_ensemble(eng, prior, theta0: torch.Tensor, n_train: int, valid_x, valid_y, *, optimizer: str = 'adamw',
                        optimizer_parameters: dict | None = None, max_epochs: int = 100, batch_size: int | None = None,
                        patience: int | None = None) -> tuple[torch.Tensor, dict]:
    """Train E members in parallel on the engine's training set.  Returns (theta [E, d], history)."""
    dev = eng.device
    theta = theta0.to(dev, torch.float32).clone()
    E = theta.shape[0]
    opt = _Optimizer(optimizer, optimizer_parameters or {}, theta)
    steps_per_epoch = 1 if not batch_size else max(1, math.ceil(n_train / batch_size))
    has_valid = valid_x is not None and len(valid_x) > 0
    stopped = torch.zeros(E, dtype=torch.bool, device=dev)
    hist_valid = torch.empty((E, 0), device=dev)
    train_nll = None
    epoch = -1
    for epoch in range(max_epochs):
        if bool(stopped.all()):
            break
        for _ in range(steps_per_epoch):
            logp, g = eng.logpost_grad(theta)
            lp_prior, g_prior = prior_value_and_grad(prior, theta)
            grad_nll = -(g - g_prior) / n_train                         # gradient of the mean negative log-likelihood
            train_nll = -(logp - lp_prior) / n_train
            theta = opt.step(theta, grad_nll, ~stopped)
        if has_valid:
            v = -eng.pointwise_loglik(theta, valid_x, valid_y).mean(dim=-1)          # [E]
            hist_valid = torch.cat([hist_valid, v[:, None]], dim=1)
            if patience:
                stopped = stopped | earlystop(hist_valid, patience)
            logger.info(f'Epoch {epoch} | Validation Loss: {v.mean().item():.4f} | early stopped: {int(stopped.sum())}/{E}')
    hist = {'epochs': epoch + 1, 'valid_nll': hist_valid.cpu(), 'train_nll': None if train_nll is None else train_nll.cpu(),
            'stopped': stopped.cpu()}
    return theta, hist
