"""effective_sample_size on the device (torch.fft) -- the estimator blackjax.diagnostics
provides to make_adaptation_L (src/training/warmup.py:458): FFT autocovariance, Geyer's
initial positive and initial monotone sequences on paired autocorrelations (Stan style).
"""
from __future__ import annotations

import math

import torch


def _next_fast_len(n: int) -> int:
    """Smallest 5-smooth number >= n (scipy.fft.next_fast_len for real FFTs of this size)."""
    best = None
    p5 = 1
    while p5 < 2 * n:
        p35 = p5
        while p35 < 2 * n:
            q = -(-n // p35)
            p2 = 1 << max(0, (q - 1).bit_length())
            cand = p2 * p35
            if cand >= n and (best is None or cand < best):
                best = cand
            p35 *= 3
        p5 *= 5
    return best


def warm_fft(n_samples: int, device) -> 'threading.Thread | None':
    """The first torch.fft call of a process spends ~2 s inside rocFFT / hipFFT (library load and kernel cache, measured
    for every transform length and layout: tools/r03/fft_jit_probe.py) -- as much as 15 000 integrator steps of BASELINE's B2.
    Phase 3 of the warm-up only needs the FFT after phases 1+2 have run, so that one-off cost is paid on a helper thread and a
    side stream while the stepping launches go on (PyTorch releases the GIL inside the call).  Same transform length and
    layout as effective_sample_size will use, so the plan it builds is the one that gets reused."""
    import threading
    dev = torch.device(device)
    if dev.type != 'cuda' or n_samples < 2:
        return None

    def work():
        try:
            with torch.cuda.device(dev), torch.cuda.stream(torch.cuda.Stream(device=dev)):
                m = _next_fast_len(2 * n_samples)
                x = torch.zeros((1, n_samples, 8), dtype=torch.float32, device=dev)
                f = torch.fft.rfft(x, n=m, dim=1)
                torch.fft.irfft(f * f.conj(), n=m, dim=1)
                torch.cuda.current_stream(dev).synchronize()
        except Exception:                                  # noqa: BLE001 -- a failed warm-up only means the cost is paid later
            pass
    t = threading.Thread(target=work, name='mile-fft-warm', daemon=True)
    t.start()
    return t


def effective_sample_size(x: torch.Tensor) -> torch.Tensor:
    """x [chains, samples, dims] -> ess [dims]."""
    C, S = x.shape[0], x.shape[1]
    assert S > 1
    x = x.to(torch.float64) if x.device.type == 'cpu' else x.to(torch.float32)
    mean_chain = x.mean(dim=1, keepdim=True)
    xc = x - mean_chain
    m = _next_fast_len(2 * S)
    f = torch.fft.rfft(xc, n=m, dim=1)
    acov = torch.fft.irfft(f * f.conj(), n=m, dim=1)[:, :S] / S
    mean_acov = acov.mean(dim=0, keepdim=True)
    mean_var0 = mean_acov[:, :1] * S / (S - 1.0)
    weighted_var = mean_var0 * (S - 1.0) / S
    if C > 1:
        weighted_var = weighted_var + mean_chain.var(dim=0, unbiased=True, keepdim=True)
    S_even = S - S % 2
    rho = torch.cat([torch.ones_like(mean_var0), 1.0 - (mean_var0 - mean_acov[:, 1:S_even]) / weighted_var], dim=1)
    rho = rho.movedim(1, 0)                       # [S_even, 1, dims]
    rho_even, rho_odd = rho[0::2].clone(), rho[1::2].clone()
    T = rho_even.shape[0]
    mask = torch.cumprod(((rho_even + rho_odd) > 0).to(torch.int8), dim=0).bool()
    tt = torch.arange(T, device=x.device).reshape((-1,) + (1,) * (mask.ndim - 1))
    max_t = torch.where(mask, tt, torch.zeros_like(tt)).amax(dim=0)
    rho_odd = torch.where(mask, rho_odd, torch.zeros_like(rho_odd))
    nxt = torch.clamp(max_t + 1, max=T - 1)       # gathers clamp, out-of-range scatters are dropped
    in_range = (max_t + 1) <= (T - 1)
    take = torch.gather(rho_even, 0, nxt[None])[0]
    cur = torch.gather(mask, 0, nxt[None])[0]
    mask_even = mask.scatter(0, nxt[None], torch.where(in_range, take > 0, cur)[None])
    rho_even = torch.where(mask_even, rho_even, torch.zeros_like(rho_even))
    rsum = rho_even + rho_odd
    run_min = torch.cummin(rsum, dim=0).values    # initial monotone sequence
    prev = torch.cat([rsum[:1], run_min[:-1]], dim=0)
    upd = rsum > prev
    rho_even_f = torch.where(upd, run_min / 2.0, rho_even)
    rho_odd_f = torch.where(upd, run_min / 2.0, rho_odd)
    ess_raw = C * S
    last = torch.gather(rho_even_f, 0, nxt[None])[0]
    tau = -1.0 + 2.0 * (rho_even_f + rho_odd_f).sum(dim=0) - last
    tau = torch.clamp(tau, min=1.0 / math.log10(ess_raw))
    return (ess_raw / tau).squeeze(0)
