"""Full-batch MCLMC sampling driver (mirror of src/training/sampling.py:32-292).

inference_loop: warm-up -> n_samples kernel steps with the thinning predicate -> one
``samples/<chain>/sample_<idx>.npz`` per kept step, ``warmup_params.txt`` and ``info.pkl``.
The scan over steps runs inside libmile_hip (mile_step) in chunks; kept positions land in an
HBM buffer and are written by a background thread pool, off the stepping critical path (the
reference writes from an io_callback inside the scan body).
"""
from __future__ import annotations

import logging
import pickle
import time
import os
from pathlib import Path

import numpy as np
import torch

from mile_amd import distributed as mdist
from mile_amd.sample_writer import WriterPool
from mile_amd.kernels import KERNELS
from mile_amd.probabilistic import resolve_target
from mile_amd.tree import as_key, ravel_tree
from mile_amd.warmup import custom_mclmc_warmup

logger = logging.getLogger(__name__)


def kept_indices(n_samples: int, n_thinning: int) -> np.ndarray:
    """idx with idx % n_thinning == 0 (sampling.py:163-165); n_thinning == 1 keeps all (:112-121).
    int32 like jnp.arange."""
    idx = np.arange(n_samples, dtype=np.int32)
    return idx[(idx % np.int32(max(int(n_thinning), 1))) == 0]


def warmup_mclmc(config, rng_key, init_params, unnorm_log_posterior, n_devices: int, chain_ids=None):
    """sampling.py:258-292: returns (state, {'step_size', 'L'}) -- sqrt_diag_cov is dropped there."""
    warmup_algo = custom_mclmc_warmup(
        logdensity_fn=unnorm_log_posterior,
        diagonal_preconditioning=config.diagonal_preconditioning,
        desired_energy_var_start=config.desired_energy_var_start,
        desired_energy_var_end=config.desired_energy_var_end,
        trust_in_estimate=config.trust_in_estimate,
        num_effective_samples=config.num_effective_samples,
        step_size_init=config.step_size_init,
        chain_ids=chain_ids,
    )
    warmup_state, parameters = warmup_algo.run(rng_key, init_params, config.warmup_steps)
    return warmup_state, {'step_size': parameters.step_size, 'L': parameters.L}


def inference_loop(unnorm_log_posterior, config, rng_key, init_params, step_ids, saving_path: Path,
                   saving_path_warmup: Path | None = None, chunk_steps: int = 500, io_workers: int | None = None,
                   return_samples: bool = False):
    """Same arguments and side effects as the reference's inference_loop (sampling.py:32-40).

    init_params: param tree whose leaves have a leading ensemble axis (or an [E, d] tensor);
    step_ids: the chain ids of this group.  Returns None (files are the output) unless
    ``return_samples`` (then the kept positions [n_kept, E, d] as a CPU tensor).
    """
    info = {}
    step_ids = np.asarray(step_ids).reshape(-1)
    n_devices = len(step_ids)
    key = as_key(rng_key)
    rng_key, warmup_key, sample_key = key.split(3)
    assert config.warmup_steps > 0, 'Number of warmup steps must be greater than 0.'
    if config.name not in ('mclmc', 'mclmc_hip'):
        raise NotImplementedError(f'{config.name} does not have a warmup implemented.')
    saving_path = Path(saving_path)
    model, x, y = resolve_target(unnorm_log_posterior)
    eng = model.engine(x, y)
    chain_ids = torch.as_tensor(step_ids, dtype=torch.int32)
    flat0 = init_params if torch.is_tensor(init_params) else ravel_tree(model.spec, init_params)
    if flat0.ndim == 1:
        flat0 = flat0[None]
    if flat0.shape[0] != n_devices:
        raise ValueError(f'init_params has {flat0.shape[0]} chains but step_ids has {n_devices}')

    logger.info('> Starting Warmup sampling...')
    t_w0 = time.time()
    warmup_state, parameters = warmup_mclmc(config=config, rng_key=warmup_key, init_params=flat0,
                                            unnorm_log_posterior=unnorm_log_posterior, n_devices=n_devices,
                                            chain_ids=chain_ids)
    saving_path.mkdir(parents=True, exist_ok=True)
    eps_host = parameters['step_size'].detach().cpu().numpy().reshape(-1)
    L_host = parameters['L'].detach().cpu().numpy().reshape(-1)
    # one file for the whole chain group (sampling.py:92-97): ranks hold disjoint chains, rank 0 writes all of them
    # in chain order (shard_chains is a contiguous rank-major partition)
    parts = mdist.gather_objects((eps_host, L_host))
    if mdist.world()[0] == 0:
        with open(saving_path.parent / 'warmup_params.txt', 'w') as f:
            f.write(','.join(str(v) for p in parts for v in p[0]) + '\n')
            f.write(','.join(str(v) for p in parts for v in p[1]) + '\n')
    torch.cuda.synchronize(eng.device)
    logger.info(f'> Warmup sampling completed successfully. ({time.time() - t_w0:.2f} s)')

    # Sampling with the tuned parameters.  blackjax.mclmc(logdensity_fn, L, step_size): the tuned
    # preconditioner is NOT forwarded (sampling.py:291), sqrt_diag_cov stays 1.
    sampler = config.kernel(unnorm_log_posterior, chain_ids=chain_ids, **parameters)
    del sampler  # the factory validates the target; the scan itself runs inside mile_step
    state = warmup_state if config.use_warmup_as_init else eng.init(flat0, seed=sample_key.seed, particle_ids=chain_ids)
    logger.info(f'> Starting {config.name} Sampling...')
    kept_all = []
    t_s0 = time.time()
    # the reference writes one zlib-compressed .npz per kept sample and chain from inside the scan; here a pool of
    # writer SUBPROCESSES (numpy only, GPUs hidden, nothing forked from this GPU process) does it off the stepping path
    io_workers = io_workers or max(2, min(12, (os.cpu_count() or 4) - 2))
    pool = WriterPool(io_workers)
    leaves = [(n, o, tuple(sh)) for n, o, sh in model.spec.leaves()]
    done = 0
    n_thin = max(int(config.n_thinning), 1)
    # Kept samples leave the device through two pinned host buffers on a copy stream: the D2H of chunk i runs under the
    # steps of chunk i+1 and its files are handed to the writers one chunk later -- the stepping stream never waits for
    # a copy or for the host (the reference blocks the scan on an io_callback per kept sample, sampling.py:140-178).
    copy_stream = torch.cuda.Stream(device=eng.device)
    pinned = [None, None]
    in_flight = None                                   # (event, host buffer view, idxs) of the previous chunk

    def hand_over(job):
        ev, host_t, idxs = job
        ev.synchronize()
        host = host_t.numpy()
        if return_samples:
            kept_all.append(host_t.clone())
        # `host` is a view of a pinned buffer that the D2H copy two chunks later overwrites, and the pool only ENQUEUES
        # the array (its feeder thread pickles it later): every chain's slice must be a private copy.  host[:, e] is
        # already C-contiguous when the rank holds one chain or the chunk keeps one sample, so ascontiguousarray
        # would hand over the view itself.
        for e, cid in enumerate(step_ids):
            pool.submit(leaves, np.array(host[:, e], copy=True, order='C'), str(saving_path), int(cid), idxs)

    slot = 0
    while done < config.n_samples:
        c = min(chunk_steps, config.n_samples - done)
        state, _, samples = eng.step(state, parameters['step_size'], parameters['L'], n_steps=c,
                                     seed=sample_key.seed, step_offset=done, n_thinning=n_thin,
                                     particle_ids=chain_ids, want_info=False, inplace=True)
        if samples is not None:
            idxs = [done + i for i in range(c) if (done + i) % n_thin == 0]
            if pinned[slot] is None or pinned[slot].numel() < samples.numel():
                pinned[slot] = torch.empty(samples.numel(), dtype=torch.float32, pin_memory=True)
            host_t = pinned[slot][:samples.numel()].view(samples.shape)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(eng.device))
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(ready)
                host_t.copy_(samples, non_blocking=True)
                samples.record_stream(copy_stream)
                copied = torch.cuda.Event()
                copied.record(copy_stream)
            if in_flight is not None:
                hand_over(in_flight)                   # the previous chunk's copy finished long ago; frees the other buffer
            in_flight = (copied, host_t, idxs)
            slot ^= 1
        done += c
    if in_flight is not None:
        hand_over(in_flight)
    torch.cuda.synchronize(eng.device)
    t_s1 = time.time()
    n_files = pool.close()
    logger.info(f'> stepping {t_s1 - t_s0:.2f} s, waiting for sample files {time.time() - t_s1:.2f} s '
                f'({n_files} files, {io_workers} writer processes)')
    logger.info(f'> {config.name} Sampling completed successfully.')
    if mdist.world()[0] == 0:
        with open(saving_path / 'info.pkl', 'wb') as f:                   # sampling.py:212-216
            pickle.dump(info, f)
    if return_samples:
        return torch.cat(kept_all, dim=0) if kept_all else None
    return None
