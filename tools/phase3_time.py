"""Where does warm-up phase 3 (make_adaptation_L) go?  B2 shape.  Dev tool."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from mile_amd import ModelSpec
from mile_amd.engine import Engine
from mile_amd import warmup as W
from mile_amd.diagnostics import effective_sample_size
from mile_amd.tree import PRNGKey
E, N = 128, 1052
spec = ModelSpec(5, (64, 64, 64, 2)); d = spec.n_params
rng = np.random.default_rng(0)
X = torch.from_numpy(rng.standard_normal((N, 5)).astype(np.float32)); y = torch.from_numpy(rng.standard_normal(N).astype(np.float32))
th = torch.from_numpy((0.05 * rng.standard_normal((E, d))).astype(np.float32)).cuda()
eng = Engine(spec, X, y, device='cuda:0')
st = eng.init(th, seed=1)
kw = dict(step_size_init=1e-3, desired_energy_var_start=5e-4, desired_energy_var_end=1e-4, trust_in_estimate=1.5,
          num_effective_samples=100, diagonal_preconditioning=False)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    W.mclmc_find_L_and_step_size(eng, st, PRNGKey(3), tune1_steps=0, tune2_steps=0, tune3_steps=200, **kw)
    torch.cuda.synchronize(); print('phase 3, 200 steps: %.3f s' % (time.perf_counter() - t0))
t0 = time.perf_counter()
g = torch.Generator().manual_seed(1)
cols = torch.stack([torch.randperm(d, generator=g)[:2000] for _ in range(E)])
print('randperm x E: %.3f s' % (time.perf_counter() - t0))
x = torch.randn(E, 200, 2000, device='cuda')
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e = effective_sample_size(x.permute(1, 0, 2).reshape(1, 200, E * 2000))
    torch.cuda.synchronize(); print('ESS [1,200,256000]: %.3f s' % (time.perf_counter() - t0))
