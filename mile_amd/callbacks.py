"""Sample persistence (mirror of src/training/callbacks.py:17-44, src/training/utils.py:69-161)."""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np

from mile_amd.spec import ModelSpec


def get_flattened_keys(d: dict, sep: str = '.') -> list[str]:
    """src/utils.py:50-70: dotted paths of the leaves, dict order."""
    keys = []
    for k, v in d.items():
        if isinstance(v, dict):
            keys.extend([f'{k}{sep}{kk}' for kk in get_flattened_keys(v)])
        else:
            keys.append(k)
    return keys


def save_position(position: dict, base: Path, idx, n: int):
    """np.savez_compressed(base/<idx>/sample_<n>.npz, **{dotted name: leaf}) -- the reference's
    per-sample file (callbacks.py:36-43).  ``position`` is ONE chain's param tree."""
    names = get_flattened_keys(position)
    leaves = []

    def walk(d):
        for v in d.values():
            if isinstance(v, dict):
                walk(v)
            else:
                leaves.append(np.asarray(v))
    walk(position)
    path = Path(base) / f'{int(idx)}/sample_{int(n)}.npz'
    path.parent.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(path, **dict(zip(names, leaves)))
    return position


def save_flat_sample(spec: ModelSpec, flat: np.ndarray, base: Path, idx: int, n: int):
    """Same file as save_position, straight from one raveled [d] row (no tree round trip)."""
    path = Path(base) / f'{int(idx)}/sample_{int(n)}.npz'
    path.parent.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(path, **{name: flat[off:off + int(np.prod(shape))].reshape(shape)
                                 for name, off, shape in spec.leaves()})


def save_tree(dir: Path, spec: ModelSpec):
    """The reference pickles a JAX PyTreeDef to <exp>/tree (utils.py:90-93), unreadable without
    JAX.  Documented deviation: a JSON sidecar with leaf names and shapes in pytree order."""
    with open(Path(dir) / 'tree', 'w') as f:
        json.dump({'format': 'mile_amd.tree.v1',
                   'leaves': [{'name': n, 'offset': o, 'shape': list(s)} for n, o, s in spec.leaves()]}, f)


def save_params(dir: Path, spec: ModelSpec, flat: np.ndarray, idx: int | None = None):
    """utils.py:69-87: <dir>/params_<idx>.npz with dotted keys (+ the tree sidecar next to it)."""
    dir = Path(dir)
    dir.mkdir(parents=True, exist_ok=True)
    if not (dir.parent / 'tree').exists():
        save_tree(dir.parent, spec)
    name = f'params_{idx}.npz' if idx is not None else 'params.npz'
    np.savez_compressed(dir / name, **{n: flat[o:o + int(np.prod(s))].reshape(s) for n, o, s in spec.leaves()})


def load_params_batch(params_path: list, spec: ModelSpec) -> np.ndarray:
    """utils.py:111-128: files sorted by the integer suffix, stacked on axis 0 -> [n, d]."""
    paths = sorted((Path(p) for p in params_path), key=lambda x: int(x.stem.split('_')[-1]))
    rows = []
    for p in paths:
        with np.load(p) as z:
            rows.append(np.concatenate([np.asarray(z[n], dtype=np.float32).reshape(-1) for n, _, _ in spec.leaves()]))
    return np.stack(rows)


def load_samples_from_dir(dir: Path, spec: ModelSpec) -> np.ndarray:
    """utils.py:131-161: samples/<chain>/sample_<n>.npz -> [n_chains, n_saved, d] (chains and samples
    sorted by their integer suffix)."""
    dir = Path(dir)
    chain_dirs = sorted([d for d in dir.iterdir() if d.is_dir()], key=lambda x: int(x.stem.split('_')[-1]))
    out = []
    for cd in chain_dirs:
        files = sorted([p for p in cd.iterdir() if p.suffix == '.npz'], key=lambda x: int(x.stem.split('_')[-1]))
        if not files:
            raise ValueError('No samples found in the directory')
        out.append(load_params_batch_ordered(files, spec))
    return np.stack(out)


def load_params_batch_ordered(files, spec):
    rows = []
    for p in files:
        with np.load(p) as z:
            rows.append(np.concatenate([np.asarray(z[n], dtype=np.float32).reshape(-1) for n, _, _ in spec.leaves()]))
    return np.stack(rows)
