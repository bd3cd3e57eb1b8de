"""Tabular data loading (host-side ETL; mirror of src/dataset/tabular.py:16-211).

Deviation: the reference shuffles with jax.random.permutation(key) (tabular.py:152-155),
which is not reproducible without JAX; here the permutation comes from
numpy.random.Generator(PCG64(seed)).  ``source: synthetic`` generates BASELINE.md's seeded
workload instead of reading a file (path = '<N>x<F>').
"""
from __future__ import annotations

import os

import numpy as np

from mile_amd.config import DataConfig


class TabularLoader:
    def __init__(self, config: DataConfig, rng: int, target_len: int = 1, shuffle: bool = True):
        assert config.data_type == 'tabular'
        self.config = config
        self.target_len = target_len
        self._rng = np.random.Generator(np.random.PCG64(rng))
        self.data = self.load_data(shuffle=shuffle, normalize=config.normalize)
        if config.datapoint_limit:
            self.data = self.data[: config.datapoint_limit]
        n = len(self.data)
        a, b = int(n * config.train_split), int(n * (config.train_split + config.valid_split))
        self.data_train, self.data_valid, self.data_test = self.data[:a], self.data[a:b], self.data[b:]

    def load_data(self, shuffle: bool, normalize: bool = True) -> np.ndarray:
        path = self.config.path
        if self.config.source != 'synthetic' and not os.path.exists(path):
            # relative paths of the shipped YAMLs ('data/airfoil.data') resolve against the repository root too
            alt = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path)
            if os.path.exists(alt):
                path = alt
        if self.config.source == 'synthetic':
            N, F = (int(v) for v in path.lower().split('x'))
            X = self._rng.standard_normal((N, F))
            if self.config.task == 'class':
                W = self._rng.standard_normal((F, 7))
                y = np.argmax(X @ W + self._rng.gumbel(size=(N, 7)), axis=1).astype(np.float64)
            else:
                Wt = self._rng.standard_normal((F, 16)) / np.sqrt(F)
                y = np.tanh(X @ Wt) @ (self._rng.standard_normal(16) / 4.0) + 0.1 * self._rng.standard_normal(N)
            data = np.concatenate([X, y[:, None]], axis=1)
        elif path.endswith('.npy'):
            data = np.load(path)
        elif path.endswith('.csv'):
            data = np.loadtxt(path, delimiter=',')
        elif path.endswith('.data'):
            data = np.genfromtxt(path, delimiter=' ')
        else:
            raise NotImplementedError('Only .npy and .csv files are supported at this time.')
        data = np.asarray(data, dtype=np.float32)      # jnp.array default dtype
        if normalize:
            if self.config.task == 'class':
                data = np.concatenate([(data[:, :-1] - data[:, :-1].mean(axis=0)) / data[:, :-1].std(axis=0),
                                       data[:, -1:]], axis=1)
            else:
                data = (data - data.mean(axis=0)) / data.std(axis=0)
        if shuffle:
            data = data[self._rng.permutation(len(data))]
        return data.astype(np.float32)

    def _x(self, d):
        return d[..., : -self.target_len]

    def _y(self, d):
        y = d[..., -self.target_len:].squeeze(-1) if self.target_len == 1 else d[..., -self.target_len:]
        return y.astype(np.int32) if self.config.task == 'class' else y

    train_x = property(lambda self: self._x(self.data_train))
    train_y = property(lambda self: self._y(self.data_train))
    valid_x = property(lambda self: self._x(self.data_valid))
    valid_y = property(lambda self: self._y(self.data_valid))
    test_x = property(lambda self: self._x(self.data_test))
    test_y = property(lambda self: self._y(self.data_test))

    def __len__(self):
        return len(self.data)


class ImageLoader:
    """Image data for the LeNet target (mirror of src/dataset/image.py's role: [N, C, H, W] float32 images and
    integer labels, split into train / valid / test).  Only ``source: synthetic`` (path = '<N>x<C>x<H>x<W>',
    10 classes) and local ``.npz`` files with arrays ``x`` [N, C, H, W] and ``y`` [N] are available: the
    reference's torchvision download (image.py:161-173) needs the network."""

    def __init__(self, config: DataConfig, rng: int, shuffle: bool = True):
        assert config.data_type == 'image'
        self.config = config
        g = np.random.Generator(np.random.PCG64(rng))
        if config.source == 'synthetic':
            N, C, H, W = (int(v) for v in config.path.lower().split('x'))
            x = g.standard_normal((N, C, H, W)).astype(np.float32)
            proto = g.standard_normal((10, C, H, W)).astype(np.float32)           # class prototypes
            y = g.integers(0, 10, N)
            x = (x + 0.5 * proto[y]).astype(np.float32)
        elif str(config.path).endswith('.npz'):
            z = np.load(config.path)
            x, y = np.asarray(z['x'], dtype=np.float32), np.asarray(z['y'])
        else:
            raise NotImplementedError('image data: only source "synthetic" or a local .npz with x [N,C,H,W], y [N]')
        if config.normalize:
            x = (x - x.mean()) / x.std()
        if shuffle:
            perm = g.permutation(len(x))
            x, y = x[perm], y[perm]
        if config.datapoint_limit:
            x, y = x[: config.datapoint_limit], y[: config.datapoint_limit]
        y = y.astype(np.int32) if config.task == 'class' else y.astype(np.float32)
        n = len(x)
        a, b = int(n * config.train_split), int(n * (config.train_split + config.valid_split))
        self.train_x, self.valid_x, self.test_x = x[:a], x[a:b], x[b:]
        self.train_y, self.valid_y, self.test_y = y[:a], y[a:b], y[b:]

    def __len__(self):
        return len(self.train_x) + len(self.valid_x) + len(self.test_x)
