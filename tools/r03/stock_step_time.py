"""Stock-net shape ([5,16,16,2], N = 1052): wall time per mile_step / mile_tune step for a dozen chains, to be set against the
kernel durations of a rocprofv3 trace of the same run (is a step launch-bound?).  Dev tool.  usage: stock_step_time.py [E] [n]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from mile_amd import ModelSpec
from mile_amd.engine import Engine, IntegratorState
E = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
N = 1052
spec = ModelSpec(5, (16, 16, 2)); d = spec.n_params
rng = np.random.default_rng(0)
X = torch.from_numpy(rng.standard_normal((N, 5)).astype(np.float32)); y = torch.from_numpy(rng.standard_normal(N).astype(np.float32))
th = torch.from_numpy((0.3 * rng.standard_normal((E, d))).astype(np.float32)).cuda()
eng = Engine(spec, X, y, device='cuda:0')
st = eng.init(th, seed=1)
f32 = dict(dtype=torch.float32, device='cuda')
eps = torch.full((E,), 2e-2, **f32); L = torch.full((E,), 8.0, **f32)
def run_step(chunk):
    s = IntegratorState(*(t.clone() for t in st)); done = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    while done < n:
        c = min(chunk, n - done)
        eng.step(s, eps, L, n_steps=c, seed=3, step_offset=done, n_thinning=10, want_info=False, inplace=True); done += c
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
def run_tune(chunk):
    s = IntegratorState(*(t.clone() for t in st)); done = 0
    t = {'step_size': torch.full((E,), 1e-3, **f32), 'step_size_max': torch.full((E,), float('inf'), **f32),
         'time': torch.zeros(E, **f32), 'x_average': torch.zeros(E, **f32), 'stream_weight': torch.zeros(E, **f32),
         'stream_average': torch.zeros((E, 2, d), **f32)}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    while done < n:
        c = min(chunk, n - done)
        eng.tune(s, t, L, c, schedule_step0=done, n_mask_steps=n, schedule_total=2 * n + 1, desired_energy_var_start=0.5,
                 desired_energy_var_end=0.1, trust_in_estimate=1.5, decay_rate=99 / 101, seed=3, step_offset=done); done += c
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
run_step(500); run_tune(256)
print(f'E={E} mile_step  %.1f us/step (chunk 500)' % run_step(500))
print(f'E={E} mile_tune  %.1f us/step (chunk 256)' % run_tune(256))
