"""Param trees <-> the raveled [E, d] layout, and the PRNG key stand-in.

The reference moves nested-dict pytrees of jax arrays around and ravels them with
jax.flatten_util.ravel_pytree (src/training/priors.py:105, src/training/warmup.py:341).
Here the tree is a nested dict of torch tensors / numpy arrays with the same names
({'fcn': {'layer0': {'bias', 'kernel'}, ...}}) and the flat layout is spec.leaves().
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from mile_amd.spec import ModelSpec

_MASK = (1 << 64) - 1


def _splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & _MASK
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
    return z ^ (z >> 31)


@dataclass(frozen=True)
class PRNGKey:
    """Stand-in for a jax PRNG key: a 64-bit seed for the device's Philox4x32-10 streams.

    The reference threads jax threefry keys through split() (src/training/sampling.py:65,
    173-174); its noise cannot be reproduced without JAX, so only the STRUCTURE is kept:
    split() derives independent child keys; the device draws noise from
    Philox(key.seed; particle id, step index, stage).
    """

    seed: int

    def __post_init__(self):
        object.__setattr__(self, 'seed', int(self.seed) & _MASK)

    def split(self, n: int = 2) -> list['PRNGKey']:
        return [PRNGKey(_splitmix64(self.seed ^ _splitmix64(i + 1))) for i in range(n)]

    def fold_in(self, data: int) -> 'PRNGKey':
        return PRNGKey(_splitmix64(self.seed ^ _splitmix64(int(data) + 0x5851F42D4C957F2D)))


def as_key(k) -> PRNGKey:
    if isinstance(k, PRNGKey):
        return k
    if isinstance(k, (int, np.integer)):
        return PRNGKey(int(k))
    raise TypeError(f'rng_key must be a mile_amd.tree.PRNGKey or an int seed, got {type(k)}')


from mile_amd.callbacks import get_flattened_keys  # noqa: E402,F401  (pure-Python helper lives with the I/O code)


def _get(tree: dict, dotted: str):
    node = tree
    for part in dotted.split('.'):
        node = node[part]
    return node


def ravel_tree(spec: ModelSpec, tree: dict, device=None) -> torch.Tensor:
    """Param tree -> [E, d] (leaves may carry a leading ensemble axis; [d] input -> [1, d])."""
    parts, E = [], None
    for name, _, shape in spec.leaves():
        leaf = torch.as_tensor(np.asarray(_get(tree, name)) if not torch.is_tensor(_get(tree, name)) else _get(tree, name))
        if leaf.ndim == len(shape):
            leaf = leaf[None]
        if tuple(leaf.shape[1:]) != tuple(shape):
            raise ValueError(f'{name}: expected [..., {shape}], got {tuple(leaf.shape)}')
        E = leaf.shape[0] if E is None else E
        if leaf.shape[0] != E:
            raise ValueError('inconsistent ensemble axis in the param tree')
        parts.append(leaf.reshape(E, -1).to(torch.float32))
    flat = torch.cat(parts, dim=1)
    return flat.to(device) if device is not None else flat


def unravel_tree(spec: ModelSpec, flat) -> dict:
    """[E, d] (or [d]) -> param tree with leaves [E, ...] (or unbatched), sorted-key order."""
    tree: dict = {}
    for name, off, shape in spec.leaves():
        n = int(np.prod(shape))
        leaf = flat[..., off:off + n].reshape(*flat.shape[:-1], *shape)
        node = tree
        parts = name.split('.')
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = leaf
    return tree
