"""B2 shape: cost of a warm-up (mile_tune) step against a sampling (mile_step) step.  Dev tool."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from mile_amd import ModelSpec
from mile_amd.engine import Engine, IntegratorState
E, N = 128, 1052
spec = ModelSpec(5, (64, 64, 64, 2))
d = spec.n_params
rng = np.random.default_rng(0)
X = torch.from_numpy(rng.standard_normal((N, 5)).astype(np.float32)); y = torch.from_numpy(rng.standard_normal(N).astype(np.float32))
th = torch.from_numpy((0.1 * rng.standard_normal((E, d))).astype(np.float32)).cuda()
eng = Engine(spec, X, y, device='cuda:0')
st = eng.init(th, seed=1)
f32 = dict(dtype=torch.float32, device='cuda')
eps = torch.full((E,), 1e-2, **f32); L = torch.full((E,), 94.0, **f32)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
def tuner():
    return {'step_size': torch.full((E,), 1e-3, **f32), 'step_size_max': torch.full((E,), float('inf'), **f32),
            'time': torch.zeros(E, **f32), 'x_average': torch.zeros(E, **f32), 'stream_weight': torch.zeros(E, **f32),
            'stream_average': torch.zeros((E, 2, d), **f32)}
kw = dict(schedule_total=2 * n + 1, desired_energy_var_start=5e-4, desired_energy_var_end=1e-4, trust_in_estimate=1.5,
          decay_rate=99 / 101, seed=3)
def run_tune(mask_steps, chunk=256):
    s = IntegratorState(*(t.clone() for t in st)); t = tuner(); done = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    while done < n:
        c = min(chunk, n - done)
        eng.tune(s, t, L, c, schedule_step0=done, n_mask_steps=mask_steps, step_offset=done, **kw); done += c
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
def run_step(chunk=256):
    s = IntegratorState(*(t.clone() for t in st)); done = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    while done < n:
        c = min(chunk, n - done)
        eng.step(s, eps, L, n_steps=c, seed=3, step_offset=done, want_info=False, inplace=True); done += c
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
run_step(); run_tune(n)
print('mile_step            %.1f us/step' % run_step())
print('mile_tune (tune1)    %.1f us/step' % run_tune(n))
print('mile_tune (tune2)    %.1f us/step' % run_tune(0))
print('mile_tune chunk 1    %.1f us/step' % run_tune(n, chunk=1))
