"""CPU restatement (NumPy, fp64 by default) of the LeNet target of BASELINE config 5 -- TEST
INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED against the reference (no tests, JAX/flax
absent: see the header of oracle/mclmc_oracle.py); pinned instead against torch.autograd in fp64
(tests/test_oracle.py), an independent implementation of the same convolutions.

Follows src/models/images/cnns.py:33-66 (LeNetCore):
    x NCHW -> NHWC; Conv(6, 5x5, stride 1, padding 2) -> act -> avg_pool 2x2/2 VALID
    -> Conv(16, 5x5, stride 1, padding 0) -> act -> avg_pool 2x2/2 VALID -> flatten (h, w, c)
    -> Dense(120) -> act -> Dense(84) -> act -> Dense(out_dim)
flax nn.Conv is a cross-correlation with kernel [kh, kw, in, out]; likelihood and prior as for the
FCN (src/training/probabilistic.py:92-138, src/training/priors.py:101-128).

Raveled parameter order = ravel_pytree's sorted keys of {'core': {conv1, conv2, fc1, fc2, fc3}}, bias
before kernel inside each.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from oracle import mclmc_oracle as M


@dataclass(frozen=True)
class LeNetSpec:
    channels: int
    height: int
    width: int
    out_dim: int
    activation: str = 'relu'
    task: str = 'classification'
    prior: str = 'Normal'
    prior_loc: float = 0.0
    prior_scale: float = 1.0

    def __post_init__(self):
        assert self.activation in M.ACTIVATIONS and self.task in M.TASKS and self.prior in M.PRIORS
        assert self.hp2 >= 1 and self.wp2 >= 1, 'image too small for LeNet'

    # stage geometry
    hp1 = property(lambda s: s.height // 2)
    wp1 = property(lambda s: s.width // 2)
    h2 = property(lambda s: s.hp1 - 4)
    w2 = property(lambda s: s.wp1 - 4)
    hp2 = property(lambda s: s.h2 // 2)
    wp2 = property(lambda s: s.w2 // 2)
    flat = property(lambda s: s.hp2 * s.wp2 * 16)
    in_features = property(lambda s: s.channels * s.height * s.width)

    def leaves(self):
        """[(dotted name, offset, shape)] in ravel_pytree order."""
        shapes = [('core.conv1.bias', (6,)), ('core.conv1.kernel', (5, 5, self.channels, 6)),
                  ('core.conv2.bias', (16,)), ('core.conv2.kernel', (5, 5, 6, 16)),
                  ('core.fc1.bias', (120,)), ('core.fc1.kernel', (self.flat, 120)),
                  ('core.fc2.bias', (84,)), ('core.fc2.kernel', (120, 84)),
                  ('core.fc3.bias', (self.out_dim,)), ('core.fc3.kernel', (84, self.out_dim))]
        out, off = [], 0
        for n, sh in shapes:
            out.append((n, off, sh))
            off += int(np.prod(sh))
        return out

    @property
    def n_params(self) -> int:
        n, o, sh = self.leaves()[-1]
        return o + int(np.prod(sh))


def _unravel(spec: LeNetSpec, theta: np.ndarray) -> dict:
    return {n: theta[:, o:o + int(np.prod(sh))].reshape((theta.shape[0],) + sh) for n, o, sh in spec.leaves()}


def _patches(x: np.ndarray, k: int) -> np.ndarray:
    """x [..., H, W, C] -> [..., H-k+1, W-k+1, k, k, C] (no copy until reshaped)."""
    v = np.lib.stride_tricks.sliding_window_view(x, (k, k), axis=(-3, -2))     # [..., H', W', C, k, k]
    return np.moveaxis(v, -3, -1)                                              # [..., H', W', k, k, C]


def _pool(a: np.ndarray) -> np.ndarray:
    """avg_pool 2x2 stride 2 VALID on [..., H, W, C]."""
    H, W = a.shape[-3] // 2 * 2, a.shape[-2] // 2 * 2
    a = a[..., :H, :W, :]
    return 0.25 * (a[..., 0::2, 0::2, :] + a[..., 1::2, 0::2, :] + a[..., 0::2, 1::2, :] + a[..., 1::2, 1::2, :])


def _unpool(g: np.ndarray, H: int, W: int) -> np.ndarray:
    out = np.zeros(g.shape[:-3] + (H, W, g.shape[-1]), dtype=g.dtype)
    for dy in (0, 1):
        for dx in (0, 1):
            out[..., dy:2 * g.shape[-3]:2, dx:2 * g.shape[-2]:2, :] = 0.25 * g
    return out


def _same(x):
    return x


def forward(spec: LeNetSpec, theta: np.ndarray, X: np.ndarray, keep: bool = False, q=_same):
    """theta [E, d], X [N, C, H, W] -> out [E, N, out_dim] (and the intermediates for the backward pass).
    q rounds the operands of the two convolutions (identity here; bfloat16 in logpost_and_grad_bf16)."""
    P = _unravel(spec, theta)
    E, N = theta.shape[0], X.shape[0]
    dt = theta.dtype
    x = np.transpose(X.astype(dt), (0, 2, 3, 1))                               # NHWC
    xp = np.pad(q(x), ((0, 0), (2, 2), (2, 2), (0, 0)))
    col1 = _patches(xp, 5).reshape(N * spec.height * spec.width, 25 * spec.channels)
    z1 = np.einsum('nk,eko->eno', col1, q(P['core.conv1.kernel']).reshape(E, -1, 6)) + P['core.conv1.bias'][:, None, :]
    a1 = M._act(spec.activation, z1).reshape(E, N, spec.height, spec.width, 6)
    p1 = _pool(a1)
    col2 = _patches(q(p1), 5).reshape(E, N * spec.h2 * spec.w2, 150)
    z2 = col2 @ q(P['core.conv2.kernel']).reshape(E, 150, 16) + P['core.conv2.bias'][:, None, :]
    a2 = M._act(spec.activation, z2).reshape(E, N, spec.h2, spec.w2, 16)
    p2 = _pool(a2).reshape(E, N, spec.flat)
    zf1 = p2 @ P['core.fc1.kernel'] + P['core.fc1.bias'][:, None, :]
    f1 = M._act(spec.activation, zf1)
    zf2 = f1 @ P['core.fc2.kernel'] + P['core.fc2.bias'][:, None, :]
    f2 = M._act(spec.activation, zf2)
    out = f2 @ P['core.fc3.kernel'] + P['core.fc3.bias'][:, None, :]
    if keep:
        return out, dict(P=P, col1=col1, z1=z1, a1=a1, col2=col2, z2=z2, a2=a2, p2=p2, zf1=zf1, f1=f1, zf2=zf2, f2=f2)
    return out


def logpost_and_grad_bf16(spec: LeNetSpec, theta: np.ndarray, X: np.ndarray, y: np.ndarray):
    """logpost_and_grad under the mixed-precision recipe of the MFMA convolution kernels (`lenet_bf16`, BASELINE config 5
    names bf16): every operand of a CONVOLUTION product -- the layer's input, its kernel, and in the backward pass the
    back-propagated dZ -- is rounded to bfloat16 where it enters the product; accumulation, bias, activation and its
    derivative (from the un-rounded activations), pooling, the three Dense layers, likelihood and prior stay in the working
    precision.  Checker for the bf16 HIP path; same citations as logpost_and_grad."""
    return logpost_and_grad(spec, theta, X, y, q=M.bf16_round)


def logpost_and_grad(spec: LeNetSpec, theta: np.ndarray, X: np.ndarray, y: np.ndarray, q=_same):
    """log_unnormalized_posterior and its gradient for an ensemble: theta [E, d] -> (logp [E], grad [E, d])."""
    E, N = theta.shape[0], X.shape[0]
    out, c = forward(spec, theta, X, keep=True, q=q)
    P = c['P']
    ll, dout = M.pointwise_loglik(spec, out, y)
    lp, gp = M.log_prior(spec, theta)
    g = {}
    g['core.fc3.kernel'] = np.swapaxes(c['f2'], 1, 2) @ dout
    g['core.fc3.bias'] = dout.sum(axis=1)
    d2 = (dout @ np.swapaxes(P['core.fc3.kernel'], 1, 2)) * M._act_grad(spec.activation, c['zf2'], c['f2'])
    g['core.fc2.kernel'] = np.swapaxes(c['f1'], 1, 2) @ d2
    g['core.fc2.bias'] = d2.sum(axis=1)
    d1 = (d2 @ np.swapaxes(P['core.fc2.kernel'], 1, 2)) * M._act_grad(spec.activation, c['zf1'], c['f1'])
    g['core.fc1.kernel'] = np.swapaxes(c['p2'], 1, 2) @ d1
    g['core.fc1.bias'] = d1.sum(axis=1)
    dp2 = (d1 @ np.swapaxes(P['core.fc1.kernel'], 1, 2)).reshape(E, N, spec.hp2, spec.wp2, 16)
    da2 = _unpool(dp2, spec.h2, spec.w2).reshape(E, N * spec.h2 * spec.w2, 16)
    dz2 = q(da2 * M._act_grad(spec.activation, c['z2'], c['a2'].reshape(E, -1, 16)))
    g['core.conv2.kernel'] = (np.swapaxes(c['col2'], 1, 2) @ dz2).reshape(E, 5, 5, 6, 16)
    g['core.conv2.bias'] = dz2.sum(axis=1)
    dcol2 = (dz2 @ np.swapaxes(q(P['core.conv2.kernel']).reshape(E, 150, 16), 1, 2)).reshape(E, N, spec.h2, spec.w2, 5, 5, 6)
    dp1 = np.zeros((E, N, spec.hp1, spec.wp1, 6), dtype=theta.dtype)
    for kh in range(5):
        for kw in range(5):
            dp1[:, :, kh:kh + spec.h2, kw:kw + spec.w2, :] += dcol2[:, :, :, :, kh, kw, :]
    da1 = _unpool(dp1, spec.height, spec.width).reshape(E, N * spec.height * spec.width, 6)
    dz1 = q(da1 * M._act_grad(spec.activation, c['z1'], c['a1'].reshape(E, -1, 6)))
    g['core.conv1.kernel'] = np.einsum('nk,eno->eko', c['col1'], dz1).reshape(E, 5, 5, spec.channels, 6)
    g['core.conv1.bias'] = dz1.sum(axis=1)
    grad = np.concatenate([g[n].reshape(E, -1) for n, _, _ in spec.leaves()], axis=1)
    return (lp + ll.sum(axis=-1)).astype(theta.dtype), (grad + gp).astype(theta.dtype)


def synthetic_problem(spec: LeNetSpec, N: int, E: int, seed: int = 0) -> dict:
    """Seeded synthetic images/labels and flax-style initial parameters (lecun-normal kernels, zero biases)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    X = rng.standard_normal((N, spec.channels, spec.height, spec.width)).astype(np.float32)
    if spec.task == 'classification':
        y = rng.integers(0, spec.out_dim, N).astype(np.int32)
    else:
        y = rng.standard_normal(N).astype(np.float32)
    theta = np.zeros((E, spec.n_params), dtype=np.float32)
    for n, o, sh in spec.leaves():
        if n.endswith('kernel'):
            fan_in = int(np.prod(sh[:-1]))
            theta[:, o:o + int(np.prod(sh))] = rng.standard_normal((E, int(np.prod(sh)))) / np.sqrt(fan_in)
        else:
            theta[:, o:o + int(np.prod(sh))] = 0.05 * rng.standard_normal((E, int(np.prod(sh))))
    d = spec.n_params
    return {'X': X, 'y': y, 'theta0': theta, 'u0': rng.standard_normal((E, d)).astype(np.float32),
            'eps': (1e-3 * (1 + 0.05 * rng.uniform(-1, 1, E))).astype(np.float32),
            'L': (np.sqrt(d) * 1e-2 * (1 + 0.05 * rng.uniform(-1, 1, E))).astype(np.float32)}
