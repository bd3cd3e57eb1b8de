#!/usr/bin/env python3
"""Headline benchmark: MCLMC particle-steps/s on BASELINE config B2.

Workload (BASELINE.json configs[1]): airfoil-shaped synthetic data
N=1052, F=5; FCN hidden_structure [64,64,64,2] (ReLU, Gaussian head, StandardNormal
prior), d=8834; E=128 particles PER GPU (weak scaling); one "step" = one full MCLMC
kernel step (O.B.A.B.A.B.O, two full-batch gradients) of every particle on the rank;
counter-RNG noise; position kept every 10th step (stock n_thinning) into an HBM buffer;
for N>1 the kept samples of each chunk are all-gathered over RCCL (sample collection).

`python bench.py --gpus N` with N > 1 and no torchrun environment starts its own N ranks
(child processes, one per GPU; the parent never touches the GPU) and relays rank 0's line.
Under `python -m torch.distributed.run ... bench.py --gpus N` it is one of the N ranks.

Timing: W untimed warm-up steps, a clock warm-up, then the timed region -- EXACTLY K steps
between barrier + synchronize on both sides, max over ranks -- is repeated (>= 5 times,
about a second in total) and the MEDIAN repetition is reported, so a short region (the
driver passes --steps 20: 3 ms) is not dominated by clock ramp and first-launch effects.

Prints ONE JSON line on rank 0.  `value` = particles x steps / wall seconds over all ranks.
The default run adds BASELINE configs[2..4] (B3, B4, B5) as bounded secondary legs inside the same line.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

WORKLOAD = 'B2'
N_THINNING = 10
CHUNK = 50                       # steps per sample-collection chunk
PEAK_FP32_MFMA_TFLOPS = 157.3    # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16 (same guide; never the 2:1-sparsity figure)

# The driver's bench line is B2 (the default).  --workload B3 reports BASELINE configs[2] in the same format.
WORKLOADS = {
    'B2': dict(ensemble=128, kernel='auto', dtype='f32', peak=PEAK_FP32_MFMA_TFLOPS, steps=400, warmup=50, cpu_particles=None,
               cpu_seconds=15.0,
               text='B2: airfoil-shaped N=1052 F=5, FCN hidden_structure [64,64,64,2] relu, Gaussian head, '
                    'StandardNormal prior, d=8834'),
    'B3': dict(ensemble=512, kernel='mfma_w128_bf16', dtype='bf16', peak=PEAK_BF16_MFMA_TFLOPS, steps=40, warmup=5,
               cpu_particles=16, cpu_seconds=8.0,
               text='B3: protein-shaped N=36000 F=9, FCN hidden_structure [128,128,128,2] relu, Gaussian head, '
                    'StandardNormal prior, d=34562; bf16 matrix operands, fp32 accumulation/parameters/integrator'),
    'B4': dict(ensemble=128, kernel='auto', dtype='f32', peak=PEAK_FP32_MFMA_TFLOPS, steps=3, warmup=1, cpu_particles=2, cpu_seconds=5.0,
               cpu_steps_per_call=1, cpu_warmup_call=False,   # one MCLMC step of two particles is 1.2 TFLOP on the host
               text='B4: covertype-shaped N=232404 F=54, FCN hidden_structure [256,256,256,256,7] relu, softmax head, '
                    'StandardNormal prior, d=213255, 128 particles per GPU (of 1024 over 8); layer-wise MFMA GEMMs (k_mm3), '
                    'fp32-faithful three-term bf16 products'),
    # BASELINE configs[4]: CIFAR-10-shaped LeNet, 256 particles over 8 GPUs = 32 per GPU, bf16; 4000 training images as
    # experiments/mclmc_cifar_lenet_b5.yaml (5000 synthetic images, 0.8 train split)
    'B5': dict(ensemble=32, kernel='lenet_bf16', dtype='bf16', peak=PEAK_BF16_MFMA_TFLOPS, steps=20, warmup=3, cpu_particles=1, cpu_seconds=8.0,
               model='lenet', image=(3, 32, 32), classes=10, rows=4000,
               text='B5: CIFAR-10-shaped N=4000 images 3x32x32, LeNet (conv 6@5x5 pad 2, pool, conv 16@5x5, pool, 120, 84, 10) relu, '
                    'softmax head, StandardNormal prior, d=83126, 32 particles per GPU (of 256 over 8); convolution products as '
                    'implicit GEMMs on bf16 MFMA (bf16 operands, fp32 accumulation), Dense layers / integrator fp32'),
}
# bounded secondary legs of the default run: (workload, steps per repetition, warm-up steps)
SECONDARY = (('B3', 10, 3), ('B4', 2, 1), ('B5', 10, 2))


def grad_flops_per_particle(F, hs, N):
    """Algorithmic FLOPs of ONE gradient evaluation: fwd 2NW + bwd 4NW - 2NW0."""
    dims, fin = [], F
    for w in hs:
        dims.append((fin, w))
        fin = w
    W = sum(i * o for i, o in dims)
    W0 = dims[0][0] * dims[0][1]
    return 2 * N * W + 4 * N * W - 2 * N * W0


def lenet_grad_flops_per_particle(C, H, W, K, N):
    """Algorithmic FLOPs of one LeNet gradient evaluation: forward 2 N M, backward 4 N M minus the first layer's input gradient."""
    h2, w2 = H // 2 - 4, W // 2 - 4
    m1 = H * W * 25 * C * 6
    m = m1 + h2 * w2 * 150 * 16 + (h2 // 2) * (w2 // 2) * 16 * 120 + 120 * 84 + 84 * K
    return (6 * m - 2 * m1) * N


def cpu_baseline(spec_o, prob, oracle, seconds_target=15.0, logpost_and_grad=None, steps_per_call=2, warmup_call=True):
    """The oracle's C/OpenMP restatement (oracle/cpu_mclmc.c: fp32; one particle per core -- the shape of the reference's
    own CPU run -- or, with fewer particles than cores, a particle's rows over the cores) on the host cores, on a bounded
    sample of the same workload: E particles, a few steps.  Any FCN (ReLU / tanh / sigmoid, Gaussian or softmax head);
    LeNet, which the C port does not restate, takes the NumPy oracle.  `cores` is what this process may really use
    (affinity mask and cgroup quota), in both forms."""
    import numpy as np
    from oracle.cpu_c import effective_cpus
    dt = np.float32
    E, d = prob['theta0'].shape
    rng = np.random.default_rng(0)
    try:
        if logpost_and_grad is not None:
            raise NotImplementedError('the C port restates the FCN only')
        from oracle.cpu_c import CpuPort
        port = CpuPort(spec_o, prob['X'], prob['y'])
        x = prob['theta0'].astype(dt).copy()
        u = (prob['u0'] / np.linalg.norm(prob['u0'], axis=1, keepdims=True)).astype(dt)
        logp, g = port.logpost_grad(x)
        T = steps_per_call
        noise = rng.standard_normal((T, 2, E, d), dtype=dt)
        if warmup_call:
            port.steps(x, u, logp, g, prob['eps'], prob['L'], noise)      # warm-up (threads, caches)
        n, t0 = 0, time.perf_counter()
        while True:
            port.steps(x, u, logp, g, prob['eps'], prob['L'], noise)
            n += T
            el = time.perf_counter() - t0
            if el > seconds_target or n >= 400:
                break
        how = 'one particle per thread' if E >= port.threads else "each particle's rows split over the threads"
        return {'value': E * n / el, 'unit': 'particle-steps/s', 'cores': int(port.threads), 'kind': 'port',
                'sample': f'{n} MCLMC steps of {E} particles of the workload (oracle/cpu_mclmc.c, fp32, OpenMP threads='
                          f'{port.threads}, {how}), {el:.1f} s'}
    except Exception as exc:                                                # noqa: BLE001
        note = f' [C port not used: {type(exc).__name__}]'
    cores = effective_cpus()                                                # affinity mask / cgroup quota, not the machine's CPU count
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=cores)                                     # OpenBLAS would otherwise start one thread per visible CPU
    except Exception:                                                       # noqa: BLE001
        pass
    lg = logpost_and_grad or oracle.logpost_and_grad
    f = lambda th: lg(spec_o, th, prob['X'], prob['y'])
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


# ----------------------------------------------------------------------------------------------
# N > 1 without a launcher: the parent starts one child per GPU and never initialises the GPU itself
# ----------------------------------------------------------------------------------------------
def spawn_ranks(n: int, argv: list[str]) -> int:
    with socket.socket() as s:                      # a free rendezvous port on the loopback interface
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), MILE_BENCH_CHILD='1')
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    print(f'bench.py: started {n} ranks (pids {[p.pid for p in procs]}), rendezvous 127.0.0.1:{port}', file=sys.stderr)
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:               # one rank failed: the others would wait in a collective forever
                rc = code
                print(f'bench.py: rank {r} exited with code {code}; stopping the other ranks', file=sys.stderr)
                for q in sorted(alive):
                    procs[q].terminate()            # exact pids of our own children
        time.sleep(0.05)
    out = procs[0].stdout.read().decode() if procs[0].stdout else ''
    if rc == 0:
        lines = [ln for ln in out.splitlines() if ln.startswith('{')]
        if len(lines) != 1:
            print(f'bench.py: expected one JSON line from rank 0, got {len(lines)}', file=sys.stderr)
            return 1
        sys.stdout.write(lines[0] + '\n')
        sys.stdout.flush()
    return rc if rc else 0


class Leg:
    """One workload on this rank's GPU: engine, state and the chunked stepping loop."""

    def __init__(self, name, args, dist, rank, world, dev, gather_cpu):
        import torch
        from oracle import mclmc_oracle as oracle      # workload generator + cpu_baseline leg only; never in the timed path
        from mile_amd import ModelSpec
        from mile_amd.engine import Engine
        self.torch, self.oracle, self.dist, self.rank, self.world, self.dev = torch, oracle, dist, rank, world, dev
        self.name, self.wl = name, WORKLOADS[name]
        self.gather_cpu, self.async_gather = gather_cpu, args.async_gather
        self.E = E = args.ensemble if (args.ensemble and name == args.workload) else self.wl['ensemble']
        kernel = args.grad_kernel if (args.grad_kernel and name == args.workload) else self.wl['kernel']
        self.lenet = self.wl.get('model') == 'lenet'
        if self.lenet:
            from oracle import lenet_oracle
            from mile_amd import LeNetSpec
            (C, H, W), K = self.wl['image'], self.wl['classes']
            self.lenet_oracle = lenet_oracle
            self.spec_o, self.N = lenet_oracle.LeNetSpec(C, H, W, K), self.wl['rows']
            self.prob = prob = lenet_oracle.synthetic_problem(self.spec_o, self.N, E * world, seed=0)
            self.spec = LeNetSpec(C, H, W, K)
            self.flops_per_particle = lenet_grad_flops_per_particle(C, H, W, K, self.N)
        else:
            self.spec_o, self.N, _ = oracle.config_spec(name)
            self.prob = prob = oracle.synthetic_problem(self.spec_o, self.N, E * world, seed=0)
            self.spec = ModelSpec(self.spec_o.in_features, self.spec_o.hidden_structure, activation='relu', task=self.spec_o.task,
                                  prior='StandardNormal')
            self.flops_per_particle = grad_flops_per_particle(self.spec.in_features, self.spec.hidden_structure, self.N)
        lo, hi = rank * E, (rank + 1) * E
        self.eng = Engine(self.spec, torch.from_numpy(prob['X']), torch.from_numpy(prob['y']), device=dev, grad_kernel=kernel)
        self.ids = torch.arange(lo, hi, dtype=torch.int32, device=dev)
        self.eps = torch.from_numpy(prob['eps'][lo:hi]).to(dev)
        self.L = torch.from_numpy(prob['L'][lo:hi]).to(dev)
        self.state = self.eng.init(torch.from_numpy(prob['theta0'][lo:hi]), seed=1234, particle_ids=self.ids)
        self.offset = 0

    def run(self, n_steps, state=None, collect=True):
        """n_steps steps in chunks of CHUNK; kept samples of every chunk are all-gathered (N > 1)."""
        torch, dist = self.torch, self.dist
        own = state is None
        state = self.state if own else state
        pending, done = [], 0
        while done < n_steps:
            c = min(CHUNK, n_steps - done)
            state, _, samples = self.eng.step(state, self.eps, self.L, n_steps=c, seed=1234, step_offset=self.offset + done,
                                              n_thinning=N_THINNING, particle_ids=self.ids, want_info=False, inplace=True)
            if collect and dist is not None and samples is not None:
                if self.gather_cpu:                 # single-device rehearsal under gloo: stage through the host
                    loc = samples.cpu()
                    out = torch.empty((self.world * loc.shape[0],) + tuple(loc.shape[1:]), dtype=loc.dtype)
                    dist.all_gather_into_tensor(out, loc)
                else:
                    out = torch.empty((self.world * samples.shape[0],) + tuple(samples.shape[1:]), dtype=samples.dtype, device=self.dev)
                    # default: the gather is ordered on the compute stream (no RCCL kernel competes with a grad launch that
                    # wants every CU); --async-gather overlaps it with the next chunk instead
                    w = dist.all_gather_into_tensor(out, samples, async_op=self.async_gather)
                    if self.async_gather:
                        pending.append((w, out, samples))
            done += c
        for w, _, _ in pending:
            w.wait()
        self.offset += n_steps
        if own:
            self.state = state
        return state

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def timed(self, n_steps):
        """EXACTLY n_steps steps between barrier + synchronize on both sides; max over ranks (seconds)."""
        torch, dist = self.torch, self.dist
        self.barrier()
        t0 = time.perf_counter()
        self.run(n_steps)
        self.barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device='cpu' if self.gather_cpu else self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    def measure(self, steps, warmup, budget_s=1.0, min_reps=5, max_reps=200):
        self.run(warmup)
        self.barrier()
        first = self.timed(steps)                                   # also the estimate that sizes the repetitions
        n_warm = int(min(max(0.3 / max(first / steps, 1e-9), 0), 20 * max(steps, 1)))
        if n_warm > 0:
            self.run(n_warm)                                        # clock warm-up, untimed
        reps = int(min(max(min_reps, math.ceil(budget_s / max(first, 1e-9))), max_reps))
        if self.dist is not None:                                   # every rank must run the same number of repetitions
            t = self.torch.tensor([reps], dtype=self.torch.int64, device='cpu' if self.gather_cpu else self.dev)
            self.dist.broadcast(t, 0)
            reps = int(t.item())
        times = [self.timed(steps) for _ in range(reps)]
        return {'median': statistics.median(times), 'min': min(times), 'max': max(times), 'first': first, 'reps': reps}

    def roofline(self, k_steps, args):
        """HIP events around every grad launch on the launch stream, same workload, separate pass (events inside the
        timed pass would perturb `value`)."""
        from mile_amd.engine import IntegratorState
        eng, spec, E = self.eng, self.spec, self.E
        eng.grad_timing_begin()
        self.run(k_steps, IntegratorState(*(t.clone() for t in self.state)), collect=False)
        self.torch.cuda.synchronize()
        ms, n_launch = eng.grad_timing_end()
        flops = self.flops_per_particle * E
        avg_s = ms * 1e-3 / max(n_launch, 1)
        achieved = flops / avg_s / 1e12
        info = eng.grad_launch_info(E)
        traffic = None   # PMC counters cannot be read from inside the run: last committed measurement of this kernel
        for rdir in ('r03', 'r02', 'r01'):
            tj = ROOT / 'profiles' / rdir / 'traffic.json'
            if tj.exists() and self.name == 'B2' and E == WORKLOADS['B2']['ensemble'] and info['kernel'] == 'k_grad_w64':
                traffic = json.loads(tj.read_text())['hbm_bytes_per_launch']
                break
            tj = ROOT / 'profiles' / rdir / 'traffic_b3.json'
            if tj.exists() and self.name == 'B3' and E == WORKLOADS['B3']['ensemble'] and info['kernel'] == 'k_grad_w128b':
                traffic = json.loads(tj.read_text())['hbm_bytes_per_launch']
                break
        peak = self.wl['peak'] if eng.grad_kernel in ('mfma_w128_bf16', 'lenet_bf16') or self.name in ('B2', 'B4') else PEAK_FP32_MFMA_TFLOPS
        roof = {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': peak,
                'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4), 'traffic': traffic,
                'kernel': info['kernel'], 'grid': list(info['grid']), 'lds_bytes': info['lds_bytes'],
                'avg_launch_us': round(avg_s * 1e6, 2), 'launches_timed': n_launch,
                'flop_per_launch': flops}
        if self.lenet:
            roof['kernel_note'] = ('avg_launch_us is one whole gradient: five convolution launches (implicit GEMMs on bf16 MFMA, pooling '
                                   'fused), the three Dense layers as hand-written k_mm3 GEMMs with fp32-faithful three-term bf16 products '
                                   '(lenet_bf16 uses no library kernel), head; priced whole against the bf16 dense peak although the Dense '
                                   'share runs six products per fp32 product')
        # The *_bf16x3 kernels do fp32 ARITHMETIC (exact products of three-term bf16 splits, fp32 accumulation) on the bf16
        # matrix pipe: six bf16 MFMA products of 32 clk per 16-deep chunk instead of eight fp32 MFMAs of 64 clk.  The roof that
        # bounds them is therefore their own instruction mix, not the fp32-MFMA peak -- pricing six-product GEMMs against the
        # fp32 peak flatters them.  `peak` / `frac` are the mix bound; the fp32-peak figures stay alongside.
        mix_scale = None
        if eng.grad_kernel == 'mfma_wide_bf16x3':
            roof['mix'] = 'all Dense products as 6 bf16 MFMA products of exact 3-term bf16 splits (fp32-faithful), layer-wise GEMMs'
            mix_scale = 8 * 64 / (6 * 32)
        if eng.grad_kernel == 'mfma_w64_bf16x3':
            # only the first layer (F <= 16 inputs) stays on the fp32 MFMA: per 32-row block 8 fq fp32 MFMAs of 64 clk and
            # 144 (nh - 1) bf16 MFMAs of 32 clk against 8 fq + 192 (nh - 1) fp32 MFMAs for the all-fp32 kernel
            nh, fq = len(spec.hidden_structure) - 1, (spec.in_features + 7) // 8
            clk_fp32 = (8 * fq + 192 * (nh - 1)) * 64
            clk_mix = 8 * fq * 64 + (nh - 1) * 144 * 32
            roof['mix'] = 'hidden-layer products as 6 bf16 MFMA products of exact 3-term bf16 splits (fp32-faithful)'
            mix_scale = clk_fp32 / clk_mix
        if mix_scale is not None:
            roof['peak_fp32_mfma'] = peak
            roof['frac_of_fp32_peak'] = round(achieved / peak, 4)
            roof['peak'] = round(peak * mix_scale, 1)
            roof['frac'] = round(achieved / (peak * mix_scale), 4)
            roof['peak_note'] = ('peak = the MFMA-pipe bound of this kernel\'s own instruction mix (fp32-exact three-term bf16 products); '
                                 'frac_of_fp32_peak prices the same algorithmic fp32 FLOPs against the 157.3 TFLOP/s fp32-MFMA peak')
        return roof

    def cpu(self):
        Ec = self.E if self.wl['cpu_particles'] is None else min(self.E, self.wl['cpu_particles'])   # bounded sample
        prob1 = {k: (v[:Ec] if k in ('theta0', 'u0', 'eps', 'L') else v) for k, v in self.prob.items()}
        if self.lenet:      # NumPy oracle of the same recipe (no C port for the convolutions)
            return cpu_baseline(self.spec_o, prob1, self.oracle, self.wl['cpu_seconds'], logpost_and_grad=self.lenet_oracle.logpost_and_grad_bf16)
        return cpu_baseline(self.spec_o, prob1, self.oracle, self.wl['cpu_seconds'], steps_per_call=self.wl.get('cpu_steps_per_call', 2),
                            warmup_call=self.wl.get('cpu_warmup_call', True))

    def dtype(self):
        return self.wl['dtype'] if self.eng.grad_kernel in ('mfma_w128_bf16', 'lenet_bf16') or self.name in ('B2', 'B4') else 'f32'


def secondary_leg(name, k2, w2, args, dev, torch):
    """One bounded secondary leg of the default run: the same measurement protocol, a few steps per repetition."""
    leg2 = Leg(name, args, None, 0, 1, dev, gather_cpu=False)
    tm2 = leg2.measure(k2, w2, budget_s=0.5, min_reps=3, max_reps=5)
    return {
        'metric': 'MCLMC integrator particle-steps/s', 'value': round(leg2.E * k2 / tm2['median'], 1),
        'unit': 'particle-steps/s', 'ms_per_step': round(tm2['median'] / k2 * 1e3, 4), 'steps': k2, 'reps': tm2['reps'],
        'dtype': leg2.dtype(), 'config': {'workload': leg2.wl['text'], 'ensemble_per_gpu': leg2.E,
                                          'grad_kernel': leg2.eng.grad_kernel,
                                          'finite': bool(torch.isfinite(leg2.state.position).all().item())},
        'roofline': None if args.no_kernel_timing else leg2.roofline(k2, args),
        'cpu_baseline': None if args.no_cpu_baseline else leg2.cpu()}


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
    ap.add_argument('--no-secondary', action='store_true', help='skip the bounded B3 / B4 / B5 legs of the default run')
    ap.add_argument('--async-gather', action='store_true',
                    help='N > 1: overlap each chunk\'s RCCL all-gather with the next chunk (default: ordered on the compute stream)')
    ap.add_argument('--force-dist', action='store_true', help='init the process group even for 1 rank (rehearsal)')
    ap.add_argument('--rehearse-one-gpu', action='store_true',
                    help='all ranks share cuda:0 and talk over gloo (rehearsal of the N-rank control flow on a 1-GPU box)')
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')

    # ---- N > 1 and no launcher environment: become the launcher (before anything touches the GPU) ----
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    wl = WORKLOADS[args.workload]
    args.steps = wl['steps'] if args.steps is None else args.steps
    args.warmup = wl['warmup'] if args.warmup is None else args.warmup

    # stdout carries exactly one JSON line: anything native libraries write to fd 1 (RCCL prints its version
    # banner there) goes to stderr instead, and the result is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} '
                         f'or run `python bench.py --gpus {args.gpus}` without a launcher environment')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)')
    if args.rehearse_one_gpu:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f'rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible')
    torch.cuda.set_device(local_rank)
    dev = torch.device(f'cuda:{local_rank}')
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if args.rehearse_one_gpu:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        print(f'bench.py: rank {rank}/{world} on {dev} ({dist.get_backend()})', file=sys.stderr)

    leg = Leg(args.workload, args, dist, rank, world, dev, gather_cpu=args.rehearse_one_gpu)
    tm = leg.measure(args.steps, args.warmup)
    el = tm['median']
    finite = bool(torch.isfinite(leg.state.position).all().item())

    roof = None
    if rank == 0 and not args.no_kernel_timing:
        # a second pass of the same workload with HIP events around every grad launch: ~100-200 steps where a step is
        # microseconds (B2), the timed step count where it is tens of milliseconds or more (B3, B4)
        roof = leg.roofline(min(max(args.steps, 100), 200) if el / args.steps < 2e-3 else args.steps, args)
    if dist is not None:
        dist.barrier()

    secondary = None
    if world == 1 and args.workload == 'B2' and not args.no_secondary and args.ensemble is None and args.grad_kernel is None:
        # BASELINE configs[2..4] (B3, B4 per GPU, B5 per GPU), bounded: a few steps per repetition, 1-16 particles on the CPU side
        secondary = {}
        for name, k2, w2 in SECONDARY:
            try:
                secondary[name] = secondary_leg(name, k2, w2, args, dev, torch)
            except Exception as exc:                                # noqa: BLE001 -- a secondary leg must never cost the headline line
                secondary[name] = {'error': f'{type(exc).__name__}: {exc}'[:300]}
            torch.cuda.empty_cache()

    if rank == 0:
        cpu = None if args.no_cpu_baseline else leg.cpu()
        E = leg.E
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
            'dtype': leg.dtype(),
            'data': 'synthetic',
            'config': {'workload': leg.wl['text'],
                       'ensemble_per_gpu': E, 'ensemble_total': E * world, 'n_thinning': N_THINNING,
                       'integrator': 'isokinetic McLachlan, O-step-O refresh, 2 full-batch gradients/step',
                       'noise': 'Philox4x32-10 counter RNG', 'grad_kernel': leg.eng.grad_kernel,
                       'parallelism': f'particles sharded {E}/GPU x {world}, '
                                      + ('no collective (one rank, no process group)' if dist is None else
                                         (f'{dist.get_backend()} all-gather of kept samples per {CHUNK}-step chunk '
                                          + ('-- a ONE-GPU REHEARSAL of the N-rank control flow over gloo through host memory, not RCCL '
                                             if args.rehearse_one_gpu else '(RCCL over xGMI) ')
                                          + ('(async, overlapping the next chunk)' if args.async_gather else '(ordered on the compute stream)'))),
                       'collective_backend': None if dist is None else dist.get_backend(),
                       'finite': finite},
            'timing': {'repetitions': tm['reps'], 'reported': 'median repetition of the K-step timed region',
                       'ms_per_step_min': round(tm['min'] / args.steps * 1e3, 5),
                       'ms_per_step_max': round(tm['max'] / args.steps * 1e3, 5),
                       'ms_per_step_first': round(tm['first'] / args.steps * 1e3, 5)},
            'roofline': roof,
            'cpu_baseline': cpu,
        }
        if secondary is not None:
            out['secondary'] = secondary
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
