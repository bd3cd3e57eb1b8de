#!/usr/bin/env python3
"""Headline benchmark: MCLMC particle-steps/s on BASELINE config B2.

Workload (BASELINE.json configs[1]): airfoil-shaped synthetic data
N=1052, F=5; FCN hidden_structure [64,64,64,2] (ReLU, Gaussian head, StandardNormal
prior), d=8834; E=128 particles PER GPU (weak scaling); one "step" = one full MCLMC
kernel step (O.B.A.B.A.B.O, two full-batch gradients) of every particle on the rank;
counter-RNG noise; position kept every 10th step (stock n_thinning) into an HBM buffer;
for N>1 the kept samples of each chunk are all-gathered over RCCL asynchronously
(sample collection), overlapped with the next chunk's steps.

Prints ONE JSON line on rank 0.  `value` = particles x steps / wall seconds over all ranks.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOAD = 'B2'
E_PER_GPU = 128
N_THINNING = 10
CHUNK = 50                       # steps per sample-collection chunk
PEAK_FP32_MFMA_TFLOPS = 157.3    # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16 (same guide; never the 2:1-sparsity figure)

# The driver's bench line is B2 (the default).  --workload B3 reports BASELINE configs[2] in the same format.
WORKLOADS = {
    'B2': dict(ensemble=128, kernel='auto', dtype='f32', peak=PEAK_FP32_MFMA_TFLOPS, steps=400, warmup=50, cpu_particles=None,
               text='B2: airfoil-shaped N=1052 F=5, FCN hidden_structure [64,64,64,2] relu, Gaussian head, '
                    'StandardNormal prior, d=8834'),
    'B3': dict(ensemble=512, kernel='mfma_w128_bf16', dtype='bf16', peak=PEAK_BF16_MFMA_TFLOPS, steps=40, warmup=5,
               cpu_particles=16,
               text='B3: protein-shaped N=36000 F=9, FCN hidden_structure [128,128,128,2] relu, Gaussian head, '
                    'StandardNormal prior, d=34562; bf16 matrix operands, fp32 accumulation/parameters/integrator'),
}


def grad_flops_per_particle(F, hs, N):
    """Algorithmic FLOPs of ONE gradient evaluation: fwd 2NW + bwd 4NW - 2NW0."""
    dims, fin = [], F
    for w in hs:
        dims.append((fin, w))
        fin = w
    W = sum(i * o for i, o in dims)
    W0 = dims[0][0] * dims[0][1]
    return 2 * N * W + 4 * N * W - 2 * N * W0


def cpu_baseline(spec_o, prob, oracle, seconds_target=15.0):
    """The oracle's C/OpenMP restatement (oracle/cpu_mclmc.c: fp32, one particle per core, the shape
    of the reference's own CPU run) on the host cores, on a bounded sample of the same workload: all E
    particles, a few steps.  Falls back to the NumPy oracle if the C library cannot be built."""
    dt = np.float32
    E, d = prob['theta0'].shape
    rng = np.random.default_rng(0)
    try:
        from oracle.cpu_c import CpuPort
        port = CpuPort(spec_o, prob['X'], prob['y'])
        x = prob['theta0'].astype(dt).copy()
        u = (prob['u0'] / np.linalg.norm(prob['u0'], axis=1, keepdims=True)).astype(dt)
        logp, g = port.logpost_grad(x)
        T = 2
        noise = rng.standard_normal((T, 2, E, d), dtype=dt)
        port.steps(x, u, logp, g, prob['eps'], prob['L'], noise)          # warm-up (threads, caches)
        n, t0 = 0, time.perf_counter()
        while True:
            port.steps(x, u, logp, g, prob['eps'], prob['L'], noise)
            n += T
            el = time.perf_counter() - t0
            if el > seconds_target or n >= 400:
                break
        return {'value': E * n / el, 'unit': 'particle-steps/s', 'cores': int(port.threads), 'kind': 'port',
                'sample': f'{n} MCLMC steps of {E} particles of the workload (oracle/cpu_mclmc.c, fp32, OpenMP threads='
                          f'{port.threads}, one particle per thread), {el:.1f} s'}
    except Exception as exc:                                                # noqa: BLE001
        note = f' [C port unavailable: {type(exc).__name__}]'
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    f = lambda th: oracle.logpost_and_grad(spec_o, th, prob['X'], prob['y'])
    st = oracle.mclmc_init(f, prob['theta0'].astype(dt), prob['u0'].astype(dt))
    eps, L = prob['eps'].astype(dt), prob['L'].astype(dt)
    n, t0 = 0, time.perf_counter()
    while True:
        z1 = rng.standard_normal((E, d), dtype=dt)
        z2 = rng.standard_normal((E, d), dtype=dt)
        st, _ = oracle.mclmc_step(f, st, eps, L, z1, z2)
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_target or n >= 64:
            break
    return {'value': E * n / el, 'unit': 'particle-steps/s', 'cores': int(cores), 'kind': 'port',
            'sample': f'{n} MCLMC steps of all {E} particles (oracle, NumPy fp32, OpenBLAS threads={cores}), '
                      f'{el:.1f} s' + note}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--workload', default=WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--ensemble', type=int, default=None, help='particles per GPU')
    ap.add_argument('--grad-kernel', default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--force-dist', action='store_true', help='init the process group even for 1 rank (rehearsal)')
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    args.steps = wl['steps'] if args.steps is None else args.steps
    args.warmup = wl['warmup'] if args.warmup is None else args.warmup
    args.ensemble = wl['ensemble'] if args.ensemble is None else args.ensemble
    args.grad_kernel = wl['kernel'] if args.grad_kernel is None else args.grad_kernel

    # stdout carries exactly one JSON line: anything native libraries write to fd 1 (RCCL prints its version
    # banner there) goes to stderr instead, and the result is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)')
    torch.cuda.set_device(local_rank)
    dev = torch.device(f'cuda:{local_rank}')
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    # the oracle is imported here ONLY as workload generator (synthetic_problem) and for the
    # cpu_baseline leg; the timed path below never touches it
    from oracle import mclmc_oracle as oracle
    from mile_amd import ModelSpec
    from mile_amd.engine import Engine

    spec_o, N, _ = oracle.config_spec(args.workload)
    E = args.ensemble
    prob = oracle.synthetic_problem(spec_o, N, E * world, seed=0)
    lo, hi = rank * E, (rank + 1) * E
    spec = ModelSpec(spec_o.in_features, spec_o.hidden_structure, activation='relu', task='regr',
                     prior='StandardNormal')
    eng = Engine(spec, torch.from_numpy(prob['X']), torch.from_numpy(prob['y']), device=dev,
                 grad_kernel=args.grad_kernel)
    ids = torch.arange(lo, hi, dtype=torch.int32, device=dev)
    eps = torch.from_numpy(prob['eps'][lo:hi]).to(dev)
    L = torch.from_numpy(prob['L'][lo:hi]).to(dev)
    state = eng.init(torch.from_numpy(prob['theta0'][lo:hi]), seed=1234, particle_ids=ids)

    def run(n_steps, offset, state, collect):
        """n_steps steps in chunks; returns (state, list of pending all-gathers)."""
        pending, done = [], 0
        while done < n_steps:
            c = min(CHUNK, n_steps - done)
            state, _, samples = eng.step(state, eps, L, n_steps=c, seed=1234, step_offset=offset + done,
                                         n_thinning=N_THINNING, particle_ids=ids, want_info=False, inplace=True)
            if collect and dist is not None and samples is not None:
                out = torch.empty((world * samples.shape[0],) + tuple(samples.shape[1:]), dtype=samples.dtype, device=dev)
                pending.append((dist.all_gather_into_tensor(out, samples, async_op=True), out, samples))
            done += c
        return state, pending

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    state, pend = run(args.warmup, 0, state, True)
    for w, _, _ in pend:
        w.wait()
    barrier()
    t0 = time.perf_counter()
    state, pend = run(args.steps, args.warmup, state, True)
    for w, _, _ in pend:
        w.wait()
    barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    finite = bool(torch.isfinite(state.position).all().item())

    # ---- dominant kernel: HIP events around every grad launch, same workload ----------
    roof = None
    if rank == 0 and not args.no_kernel_timing:
        k_steps = min(args.steps, 200)
        eng.grad_timing_begin()
        state2, _ = run(k_steps, args.warmup + args.steps, IntegratorStateClone(state), False)
        torch.cuda.synchronize()
        ms, n_launch = eng.grad_timing_end()
        flops = grad_flops_per_particle(spec.in_features, spec.hidden_structure, N) * E
        avg_s = ms * 1e-3 / max(n_launch, 1)
        achieved = flops / avg_s / 1e12
        info = eng.grad_launch_info(E)
        traffic = None   # PMC counters cannot be read from inside the run: last committed measurement
        tj = ROOT / 'profiles' / 'r01' / 'traffic.json'
        if tj.exists() and args.workload == 'B2' and E == E_PER_GPU and info['kernel'] == 'k_grad_w64':
            traffic = json.loads(tj.read_text())['hbm_bytes_per_launch']
        peak = wl['peak'] if eng.grad_kernel == 'mfma_w128_bf16' or args.workload == 'B2' else PEAK_FP32_MFMA_TFLOPS
        roof = {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': peak,
                'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4), 'traffic': traffic,
                'kernel': info['kernel'], 'grid': list(info['grid']), 'lds_bytes': info['lds_bytes'],
                'avg_launch_us': round(avg_s * 1e6, 2), 'launches_timed': n_launch,
                'flop_per_launch': flops}
        if eng.grad_kernel == 'mfma_w64_bf16x3':
            # `peak` stays the fp32 MFMA peak (the arithmetic is fp32: exact products of 3-term bf16 splits, fp32
            # accumulation).  The kernel's own MFMA-pipe floor is lower than an all-fp32 kernel's: hidden forward /
            # dH / dW tiles cost 6 bf16 MFMAs x 32 clk per 16-deep chunk instead of 8 fp32 MFMAs x 64 clk (only
            # the first layer stays on the fp32 MFMA).  `frac_of_mix_bound` prices `achieved` against that mix.
            nh, fq = len(spec.hidden_structure) - 1, (spec.in_features + 7) // 8
            clk_fp32 = (8 * fq + 192 * (nh - 1)) * 64
            clk_mix = 8 * fq * 64 + (nh - 1) * 144 * 32
            roof['mix'] = 'hidden-layer products as 6 bf16 MFMA products of exact 3-term bf16 splits (fp32-faithful)'
            roof['peak_mix_bound'] = round(peak * clk_fp32 / clk_mix, 1)
            roof['frac_of_mix_bound'] = round(achieved / (peak * clk_fp32 / clk_mix), 4)
    if dist is not None:
        dist.barrier()

    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline:
            Ec = E if wl['cpu_particles'] is None else min(E, wl['cpu_particles'])   # bounded sample
            prob1 = {k: (v[:Ec] if k in ('theta0', 'u0', 'eps', 'L') else v) for k, v in prob.items()}
            cpu = cpu_baseline(spec_o, prob1, oracle)
        value = E * world * args.steps / el
        out = {
            'metric': 'MCLMC integrator particle-steps/s (integrator-steps/s x ensemble size)',
            'value': round(value, 1),
            'unit': 'particle-steps/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(el / args.steps * 1e3, 5),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': wl['dtype'] if eng.grad_kernel == 'mfma_w128_bf16' or args.workload == 'B2' else 'f32',
            'data': 'synthetic',
            'config': {'workload': wl['text'],
                       'ensemble_per_gpu': E, 'ensemble_total': E * world, 'n_thinning': N_THINNING,
                       'integrator': 'isokinetic McLachlan, O-step-O refresh, 2 full-batch gradients/step',
                       'noise': 'Philox4x32-10 counter RNG', 'grad_kernel': eng.grad_kernel,
                       'parallelism': f'particles sharded {E}/GPU x {world}, async RCCL all-gather of kept samples',
                       'finite': finite},
            'roofline': roof,
            'cpu_baseline': cpu,
        }
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def IntegratorStateClone(st):
    from mile_amd.engine import IntegratorState
    return IntegratorState(*(t.clone() for t in st))


if __name__ == '__main__':
    main()
