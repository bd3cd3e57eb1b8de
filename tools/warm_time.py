"""Where does a host-driven warm-up step go?  B3 shape.  Dev tool."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from mile_amd import ModelSpec
from mile_amd.engine import Engine
from mile_amd import warmup as W
from mile_amd.tree import PRNGKey
E, N = 512, 36000
spec = ModelSpec(9, (128, 128, 128, 2))
d = spec.n_params
rng = np.random.default_rng(0)
X = torch.from_numpy(rng.standard_normal((N, 9)).astype(np.float32)); y = torch.from_numpy(rng.standard_normal(N).astype(np.float32))
th = torch.from_numpy((0.05 * rng.standard_normal((E, d))).astype(np.float32)).cuda()
eng = Engine(spec, X, y, device='cuda:0', grad_kernel='mfma_w128_bf16')
st = eng.init(th, seed=1)
eps = torch.full((E,), 1e-3, device='cuda'); L = torch.full((E,), 1.0, device='cuda')
def t(f, n=20):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('step x1 (pure)    %.2f ms' % t(lambda: eng.step(st, eps, L, n_steps=1, seed=3)))
print('step x1 (inplace) %.2f ms' % t(lambda: eng.step(st, eps, L, n_steps=1, seed=3, inplace=True)))
print('step x10 /10      %.2f ms' % (t(lambda: eng.step(st, eps, L, n_steps=10, seed=3, inplace=True), 4) / 10))
nxt, info, _ = eng.step(st, eps, L, n_steps=1, seed=3)
em = torch.full((E,), float('inf'), device='cuda')
print('handle_nans       %.2f ms' % t(lambda: W.handle_nans(st, nxt, eps, em, info.energy_change[0])))
t0 = time.perf_counter()
s2, p = W.mclmc_find_L_and_step_size(eng, st, PRNGKey(3), tune1_steps=100, tune2_steps=0, tune3_steps=0, step_size_init=1e-3,
    desired_energy_var_start=5e-4, desired_energy_var_end=1e-4, trust_in_estimate=1.5, num_effective_samples=100, diagonal_preconditioning=False)
torch.cuda.synchronize(); print('tune1 x100        %.2f ms/step' % ((time.perf_counter() - t0) * 10))
t0 = time.perf_counter()
s2, p = W.mclmc_find_L_and_step_size(eng, st, PRNGKey(3), tune1_steps=0, tune2_steps=50, tune3_steps=0, step_size_init=1e-3,
    desired_energy_var_start=5e-4, desired_energy_var_end=1e-4, trust_in_estimate=1.5, num_effective_samples=100, diagonal_preconditioning=False)
torch.cuda.synchronize(); print('tune2 x50         %.2f ms/step' % ((time.perf_counter() - t0) * 20))
t0 = time.perf_counter()
s2, p = W.mclmc_find_L_and_step_size(eng, st, PRNGKey(3), tune1_steps=0, tune2_steps=0, tune3_steps=50, step_size_init=1e-3,
    desired_energy_var_start=5e-4, desired_energy_var_end=1e-4, trust_in_estimate=1.5, num_effective_samples=100, diagonal_preconditioning=False)
torch.cuda.synchronize(); print('tune3 x50         %.2f ms/step' % ((time.perf_counter() - t0) * 20))
