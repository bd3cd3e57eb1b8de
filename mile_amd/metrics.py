"""LPPD, the parity metric (mirror of src/inference/metrics.py:247-312,428-446), and the
batched predictive forward pass that feeds it (src/inference/evaluation.py:16-43 does a Python
loop of module.apply per sample; here all C*S samples go through one batched matmul chain).
Evaluation is a consumer of the hot path, not part of it: plain torch on the device.
"""
from __future__ import annotations

import math

import torch

from mile_amd.spec import ModelSpec

_LOG_SQRT_2PI = 0.5 * math.log(2.0 * math.pi)


def predict(spec: ModelSpec, flat: torch.Tensor, X: torch.Tensor, batch: int = 256) -> torch.Tensor:
    """flat [..., d] samples -> network outputs [..., N, out] (FullyConnected.__call__)."""
    lead = flat.shape[:-1]
    th = flat.reshape(-1, flat.shape[-1])
    X = X.to(th.device, th.dtype)
    outs = []
    leaves = {n: (o, s) for n, o, s in spec.leaves()}
    act = {'relu': torch.relu, 'tanh': torch.tanh, 'sigmoid': torch.sigmoid}[spec.activation]
    nl = len(spec.hidden_structure)
    for b0 in range(0, th.shape[0], batch):
        t = th[b0:b0 + batch]
        h = X[None].expand(t.shape[0], -1, -1)
        for li in range(nl):
            ko, ks = leaves[f'{spec.root}.layer{li}.kernel']
            bo, bs = leaves[f'{spec.root}.layer{li}.bias']
            W = t[:, ko:ko + ks[0] * ks[1]].reshape(-1, *ks)
            b = t[:, bo:bo + bs[0]]
            h = torch.baddbmm(b[:, None, :], h, W)
            if li < nl - 1:
                h = act(h)
        outs.append(h)
    out = torch.cat(outs, dim=0)
    return out.reshape(*lead, *out.shape[1:])


def pointwise_lppd(lvals: torch.Tensor, y: torch.Tensor, task: str) -> torch.Tensor:
    """lvals [C, S, N, out] -> [C, S, N] (metrics.py:247-294)."""
    if lvals.ndim == 3:
        lvals = lvals[None]
    elif lvals.ndim == 2:
        lvals = lvals[None, None]
    y = y.to(lvals.device)
    if task in ('regr', 'regression'):
        sigma = torch.exp(lvals[..., 1]).clamp(min=1e-6, max=1e6)
        r = (y.to(lvals.dtype) - lvals[..., 0]) / sigma
        return -0.5 * r * r - torch.log(sigma) - _LOG_SQRT_2PI
    logp = torch.log_softmax(lvals, dim=-1)
    idx = y.to(torch.int64).expand(lvals.shape[:-1])[..., None]
    return torch.gather(logp, -1, idx)[..., 0]


def lppd(lppd_pointwise: torch.Tensor) -> torch.Tensor:
    """metrics.py:297-312: mean_n( logsumexp_{c,s} l - log(C*S) )."""
    C, S = lppd_pointwise.shape[:2]
    flat = lppd_pointwise.reshape(C * S, -1)
    return (torch.logsumexp(flat, dim=0) - math.log(C * S)).mean()


def running_lppd(lppd_pointwise: torch.Tensor) -> torch.Tensor:
    """metrics.py:428-446: running mean over the sample axis, averaged over obs and chains."""
    e = torch.exp(lppd_pointwise)
    cnt = torch.arange(1, e.shape[1] + 1, device=e.device, dtype=e.dtype)[None, :, None]
    return torch.log(torch.cumsum(e, dim=1) / cnt).mean(dim=-1).mean(dim=0)


def rank_normalize_array(samples: torch.Tensor) -> torch.Tensor:
    """metrics.py:226-244: overall ranks (average rank for ties) -> normal quantiles."""
    flat = samples.reshape(-1).to(torch.float64)
    n = flat.numel()
    order = torch.argsort(flat, stable=True)
    sv = flat[order]
    ranks_sorted = torch.arange(1, n + 1, dtype=torch.float64, device=flat.device)
    # average rank within runs of equal values (scipy.stats.rankdata method='average')
    new_run = torch.ones(n, dtype=torch.bool, device=flat.device)
    new_run[1:] = sv[1:] != sv[:-1]
    run_id = torch.cumsum(new_run.to(torch.int64), 0) - 1
    n_runs = int(run_id[-1].item()) + 1 if n else 0
    sums = torch.zeros(n_runs, dtype=torch.float64, device=flat.device).index_add_(0, run_id, ranks_sorted)
    cnts = torch.zeros(n_runs, dtype=torch.float64, device=flat.device).index_add_(0, run_id, torch.ones_like(ranks_sorted))
    ranks = torch.empty_like(flat)
    ranks[order] = (sums / cnts)[run_id]
    tmp = (ranks - 0.375) / (n + 0.25)
    return (math.sqrt(2.0) * torch.erfinv(2.0 * tmp - 1.0)).reshape(samples.shape).to(samples.dtype)


def effective_sample_size(x: torch.Tensor, rank_normalize: bool = True) -> torch.Tensor:
    """metrics.py:386-405: x [C, S, ...] -> per-chain ESS [C, ...].  Each trailing column is rank-normalised over
    the pooled C*S draws, then every chain goes through the single-chain FFT/Geyer estimator (the reference calls
    numpyro's; here the Stan-style estimator of mile_amd.diagnostics, the same one the tuner uses)."""
    from mile_amd.diagnostics import effective_sample_size as ess1
    C, S = x.shape[:2]
    cols = x.reshape(C, S, -1)
    if rank_normalize:
        cols = torch.stack([rank_normalize_array(cols[:, :, k]) for k in range(cols.shape[2])], dim=2)
    out = torch.stack([ess1(cols[c][None]) for c in range(C)])
    return out.reshape(C, *x.shape[2:])
