"""Warm-start training of the deep ensemble the chains start from (mirror of src/training/trainer.py:330-538).

The reference trains every ensemble member with optax (AdamW by default) on minibatches of the mean negative
log-likelihood: per epoch the training rows are reshuffled and cut into len // batch_size batches of exactly
batch_size rows (the remainder is dropped, src/dataset/tabular.py:170-212), every member sees the same batch,
validation after each epoch, per-member early stopping (`earlystop`, trainer.py:920-939).  Here the gradient comes
from the HIP engine: the epoch's shuffled copy of the training set is handed to it once (`set_data`) and each
optimizer step evaluates the members on one contiguous row window of it (`mile_set_row_window`: the generic, width-64
MFMA, layer-wise wide-net and LeNet kernels) -- the reference's batches, optimizer, validation schedule and
early-stopping rule; only the permutation differs (torch's generator instead of JAX's key).  The two kernels without a
row window (the rocBLAS cross-check path and the width-128 bf16 kernel) are swapped for the kernel AUTO picks for the
net while the members are trained (round 3; before: the same number of FULL-batch steps per epoch), and come back for
sampling.
"""
from __future__ import annotations

import logging

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
                        patience: int | None = None, train_x=None, train_y=None, seed: int = 0) -> tuple[torch.Tensor, dict]:
    """Train E members in parallel on the engine's training set.  Returns (theta [E, d], history).
    ``train_x`` / ``train_y`` (the tensors the engine was built on) enable true minibatches; without them, or on a
    grad kernel without row windows, every step sees the whole training set."""
    dev = eng.device
    theta = theta0.to(dev, torch.float32).clone().contiguous()
    E = theta.shape[0]
    # Round 3: the step itself runs in the library (mile_warmstart_step: grad kernel + one fused optimizer launch, theta and the
    # moments updated in place) -- the same update rules as _Optimizer below, which stays as their host-side restatement (CPU
    # test; MILE_WARMSTART_TORCH=1 runs it instead, on `grad(log posterior) - grad(log prior)`, whose rounding residue Adam
    # amplifies on units with an exactly-zero likelihood gradient -- a debugging path, not a second product path).  ~30 elementwise torch launches per step were 1.1 ms of a 1.2 ms step.
    import os
    fused = os.environ.get('MILE_WARMSTART_TORCH') is None
    opt = _Optimizer(optimizer, optimizer_parameters or {}, theta)
    ostate = {'name': opt.name, 'learning_rate': opt.lr, 'b1': opt.b1, 'b2': opt.b2, 'eps': opt.eps, 'weight_decay': opt.wd,
              't': 0, 'm': opt.m, 'v': opt.v}
    WINDOWED = ('generic', 'mfma_narrow_f32', 'mfma_w64', 'mfma_w64_bf16x3', 'mfma_wide_bf16x3', 'mfma_wide_bf16', 'lenet_f32', 'lenet_bf16')
    want_minibatch = bool(batch_size) and batch_size < n_train and train_x is not None and train_y is not None
    sampler_kernel = eng.grad_kernel
    if want_minibatch and sampler_kernel not in WINDOWED:
        # the sampler's kernel has no row window (width-128 bf16, rocBLAS cross-check): train on what AUTO picks for this net
        # (fp32-faithful, windowed) and hand the engine back as configured -- the reference's minibatches rather than the same
        # number of full-batch steps
        eng.set_grad_kernel('auto')
        if eng.grad_kernel not in WINDOWED:
            eng.set_grad_kernel(sampler_kernel)
    minibatch = want_minibatch and eng.grad_kernel in WINDOWED
    if minibatch:
        n_batches = n_train // batch_size                                    # drop last, tabular.py:190-191
        gen = torch.Generator().manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
        tx = torch.as_tensor(train_x).reshape(n_train, -1)
        ty = torch.as_tensor(train_y)
    steps_per_epoch = 1 if not batch_size else max(1, n_train // batch_size)      # len // batch_size, tabular.py:190-191
    has_valid = valid_x is not None and len(valid_x) > 0
    stopped = torch.zeros(E, dtype=torch.bool, device=dev)
    hist_valid = torch.empty((E, 0), device=dev)
    train_nll = None
    epoch = -1
    try:
        for epoch in range(max_epochs):
            if bool(stopped.all()):
                break
            if minibatch:
                perm = torch.randperm(n_train, generator=gen)                # loader.shuffle() between epochs
                eng.set_data(tx[perm], ty[perm])
                active = ~stopped
                for b in range(n_batches):
                    eng.set_row_window(b * batch_size, batch_size)
                    if fused:
                        train_nll = eng.warmstart_step(theta, ostate, active, want_nll=(b == n_batches - 1))
                        continue
                    logp, g = eng.logpost_grad(theta)
                    lp_prior, g_prior = prior_value_and_grad(prior, theta)
                    grad_nll = -(g - g_prior) / batch_size                   # gradient of the batch-mean negative log-likelihood
                    train_nll = -(logp - lp_prior) / batch_size
                    theta = opt.step(theta, grad_nll, active)
                eng.set_row_window(0, 0)
            else:
                active = ~stopped
                for k in range(steps_per_epoch):
                    if fused:
                        train_nll = eng.warmstart_step(theta, ostate, active, want_nll=(k == steps_per_epoch - 1))
                        continue
                    logp, g = eng.logpost_grad(theta)
                    lp_prior, g_prior = prior_value_and_grad(prior, theta)
                    grad_nll = -(g - g_prior) / n_train                      # gradient of the mean negative log-likelihood
                    train_nll = -(logp - lp_prior) / n_train
                    theta = opt.step(theta, grad_nll, active)
            if has_valid:
                v = -eng.pointwise_loglik(theta, valid_x, valid_y).mean(dim=-1)          # [E]
                hist_valid = torch.cat([hist_valid, v[:, None]], dim=1)
                if patience:
                    stopped = stopped | earlystop(hist_valid, patience)
                logger.info(f'Epoch {epoch} | Validation Loss: {v.mean().item():.4f} | early stopped: {int(stopped.sum())}/{E}')
    finally:
        if minibatch:                                                        # the sampler wants the full set, original order
            eng.set_data(tx, ty)
        if eng.grad_kernel != sampler_kernel:
            eng.set_grad_kernel(sampler_kernel)
    hist = {'epochs': epoch + 1, 'valid_nll': hist_valid.cpu(), 'train_nll': None if train_nll is None else train_nll.cpu(),
            'stopped': stopped.cpu(), 'minibatch': minibatch}
    return theta, hist
