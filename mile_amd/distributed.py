"""Ensemble sharding across GPUs: one process per GPU, chains partitioned, no collective on
the data path; ONE all-gather at sample-collection time (RCCL over xGMI; gloo in CPU tests).

The reference runs chains as collective-free jax.pmap replicas (src/training/sampling.py:
180-188), so there is no reference communication pattern to mirror.
"""
from __future__ import annotations

import os

import numpy as np
import torch


def world() -> tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1-process default)."""
    return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))


def shard_chains(chain_ids, world_size: int, rank: int) -> np.ndarray:
    """Contiguous block partition of the chain ids: rank r owns [r n/G, (r+1) n/G).
    Chain ids (not local indices) key the RNG streams, so per-chain results do not depend on G."""
    ids = np.asarray(chain_ids)
    n = len(ids)
    lo, hi = (rank * n) // world_size, ((rank + 1) * n) // world_size
    return ids[lo:hi]


def init_process_group(backend: str | None = None, device: torch.device | None = None):
    import torch.distributed as dist
    if dist.is_initialized():
        return dist
    rank, ws, _ = world()
    if ws == 1:
        return None
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29511')
    backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
    kw = {'device_id': device} if (backend == 'nccl' and device is not None) else {}
    dist.init_process_group(backend, rank=rank, world_size=ws, **kw)
    return dist


def broadcast_object(obj_list: list, src: int = 0) -> list:
    """dist.broadcast_object_list when a group is up; identity for one process."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast_object_list(obj_list, src=src)
    return obj_list


def gather_objects(obj) -> list:
    """Every rank's object, rank order (all_gather_object); [obj] for one process."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, obj)
        return out
    return [obj]


def gather_samples(samples: torch.Tensor, async_op: bool = False):
    """All-gather kept positions [K, E_local, d] -> [K, E_total, d] (rank-major chain order ==
    shard_chains order).  Equal E_local on every rank.  Returns (tensor, work|None)."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return samples, None
    ws = dist.get_world_size()
    K, El, d = samples.shape
    out = torch.empty((ws * K, El, d), dtype=samples.dtype, device=samples.device)   # rank-concatenated
    work = dist.all_gather_into_tensor(out, samples.contiguous(), async_op=async_op)
    view = out.view(ws, K, El, d).permute(1, 0, 2, 3).reshape(K, ws * El, d) if not async_op else \
        _LazyGathered(out, ws, K, El, d)
    return view, work


class _LazyGathered:
    """The gathered buffer of an async all-gather; call .contiguous() after work.wait()."""

    def __init__(self, out, ws, K, El, d):
        self._out, self._shape = out, (ws, K, El, d)

    def contiguous(self):
        ws, K, El, d = self._shape
        return self._out.view(ws, K, El, d).permute(1, 0, 2, 3).reshape(K, ws * El, d)


def lppd_distributed(lppd_pointwise_local: torch.Tensor, total_chains: int) -> torch.Tensor:
    """LPPD over chains sharded across ranks without gathering samples: all-reduce (max, then sum)
    of the per-observation logsumexp partials -- [N] floats instead of [C, S, d]."""
    import math
    import torch.distributed as dist
    C, S = lppd_pointwise_local.shape[:2]
    flat = lppd_pointwise_local.reshape(C * S, -1)
    m = flat.max(dim=0).values
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        s = torch.exp(flat - m).sum(dim=0)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    else:
        s = torch.exp(flat - m).sum(dim=0)
    return (m + torch.log(s) - math.log(total_chains * S)).mean()
