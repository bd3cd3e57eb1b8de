"""Model spec and raveled-parameter layout of the FCN target.

Mirrors what the reference's closure ``partial(log_unnormalized_posterior, x=..., y=...)``
captures (src/training/trainer.py:576-580): FCNConfig (src/config/models/fcn.py:7-30),
the prior (src/training/priors.py:67-91) and the task (src/training/probabilistic.py:92-109).
"""
from __future__ import annotations

from dataclasses import dataclass

ACTIVATIONS = ('relu', 'tanh', 'sigmoid')
TASKS = ('regr', 'classification')
PRIORS = ('Normal', 'StandardNormal', 'Laplace')


@dataclass(frozen=True)
class ModelSpec:
    """``hidden_structure`` as in FCNConfig: one entry per Dense layer, last = output layer."""

    in_features: int
    hidden_structure: tuple
    activation: str = 'relu'
    task: str = 'regr'
    prior: str = 'StandardNormal'
    prior_loc: float = 0.0
    prior_scale: float = 1.0
    use_bias: bool = True
    root: str = 'fcn'   # name of the FullyConnected submodule inside FCN (src/models/tabular/fcn.py:18)

    def __post_init__(self):
        object.__setattr__(self, 'hidden_structure', tuple(int(w) for w in self.hidden_structure))
        if self.activation not in ACTIVATIONS:
            raise NotImplementedError(f'activation {self.activation!r} (supported: {ACTIVATIONS})')
        if self.task not in TASKS:
            raise NotImplementedError(f'Likelihood computation for {self.task} not implemented')
        if self.prior not in PRIORS:
            raise NotImplementedError(f'Prior Distribution for {self.prior} is not yet implemented.')
        if self.prior == 'StandardNormal':
            object.__setattr__(self, 'prior_loc', 0.0)
            object.__setattr__(self, 'prior_scale', 1.0)
        if self.task == 'regr' and self.hidden_structure[-1] != 2:
            raise ValueError('regression needs hidden_structure[-1] == 2 (mu, log sigma)')

    @property
    def layer_dims(self):
        dims, fin = [], self.in_features
        for w in self.hidden_structure:
            dims.append((fin, w))
            fin = w
        return dims

    @property
    def n_params(self) -> int:
        return sum(i * o + (o if self.use_bias else 0) for i, o in self.layer_dims)

    def layer_order(self):
        """ravel_pytree visits dict keys sorted as strings: 'layer10' < 'layer2'."""
        return sorted(range(len(self.hidden_structure)), key=lambda i: f'layer{i}')

    def leaves(self):
        """[(dotted name, offset, shape)] in pytree order == np.savez key order of
        save_position (src/training/callbacks.py:36-43, src/utils.py:50-70)."""
        out, off = [], 0
        for li in self.layer_order():
            fin, fout = self.layer_dims[li]
            if self.use_bias:
                out.append((f'{self.root}.layer{li}.bias', off, (fout,)))
                off += fout
            out.append((f'{self.root}.layer{li}.kernel', off, (fin, fout)))
            off += fin * fout
        return out


@dataclass(frozen=True)
class LeNetSpec:
    """LeNet (src/models/images/cnns.py:10-66, LeNetConfig src/config/models/cnns.py) on [N, C, H, W] images:
    Conv(6, 5x5, pad 2) - act - avg_pool 2 - Conv(16, 5x5) - act - avg_pool 2 - Dense 120 - Dense 84 - Dense out_dim.
    Duck-types ModelSpec where the host code needs it (in_features, n_params, leaves, task/prior fields)."""

    channels: int
    height: int
    width: int
    out_dim: int
    activation: str = 'relu'
    task: str = 'classification'
    prior: str = 'StandardNormal'
    prior_loc: float = 0.0
    prior_scale: float = 1.0
    use_bias: bool = True
    root: str = 'core'   # name of the LeNetCore submodule inside LeNet (cnns.py:21-25)

    def __post_init__(self):
        if self.activation not in ACTIVATIONS:
            raise NotImplementedError(f'activation {self.activation!r} (supported: {ACTIVATIONS})')
        if self.task not in TASKS:
            raise NotImplementedError(f'Likelihood computation for {self.task} not implemented')
        if self.prior not in PRIORS:
            raise NotImplementedError(f'Prior Distribution for {self.prior} is not yet implemented.')
        if self.prior == 'StandardNormal':
            object.__setattr__(self, 'prior_loc', 0.0)
            object.__setattr__(self, 'prior_scale', 1.0)
        if not self.use_bias:
            raise NotImplementedError('use_bias=False is not supported')
        if self.task == 'regr' and self.out_dim != 2:
            raise ValueError('regression needs out_dim == 2 (mu, log sigma)')
        if (self.height // 2 - 4) // 2 < 1 or (self.width // 2 - 4) // 2 < 1:
            raise ValueError('image too small for LeNet')

    @property
    def in_features(self) -> int:
        return self.channels * self.height * self.width

    @property
    def flat(self) -> int:
        return ((self.height // 2 - 4) // 2) * ((self.width // 2 - 4) // 2) * 16

    @property
    def hidden_structure(self):          # only its last entry (the output width) is meaningful for LeNet
        return (self.out_dim,)

    def leaves(self):
        """[(dotted name, offset, shape)] in ravel_pytree order: conv1, conv2, fc1, fc2, fc3; bias before kernel;
        conv kernels [kh, kw, in, out] as flax stores them."""
        shapes = [('conv1.bias', (6,)), ('conv1.kernel', (5, 5, self.channels, 6)),
                  ('conv2.bias', (16,)), ('conv2.kernel', (5, 5, 6, 16)),
                  ('fc1.bias', (120,)), ('fc1.kernel', (self.flat, 120)),
                  ('fc2.bias', (84,)), ('fc2.kernel', (120, 84)),
                  ('fc3.bias', (self.out_dim,)), ('fc3.kernel', (84, self.out_dim))]
        out, off = [], 0
        for name, sh in shapes:
            out.append((f'{self.root}.{name}', off, sh))
            n = 1
            for v in sh:
                n *= v
            off += n
        return out

    @property
    def n_params(self) -> int:
        name, off, sh = self.leaves()[-1]
        n = 1
        for v in sh:
            n *= v
        return off + n
