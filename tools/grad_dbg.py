import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from mile_amd import ModelSpec
from mile_amd.engine import Engine
E, N = 128, 1052
spec = ModelSpec(5, (64, 64, 64, 2))
rng = np.random.default_rng(0)
X = torch.from_numpy(rng.standard_normal((N, 5)).astype(np.float32))
y = torch.from_numpy(rng.standard_normal(N).astype(np.float32))
eng = Engine(spec, X, y, device='cuda:0')
th = torch.from_numpy((0.1 * rng.standard_normal((E, spec.n_params))).astype(np.float32)).cuda()
for dbg in (0, 1, 3, 5, 7):
    os.environ['MILE_DEBUG'] = str(dbg)
    for _ in range(5):
        eng.logpost_grad(th)
    torch.cuda.synchronize()
    eng.grad_timing_begin()
    for _ in range(50):
        eng.logpost_grad(th)
    torch.cuda.synchronize()
    ms, n = eng.grad_timing_end()
    print(f'dbg={dbg} (skip blocks={dbg&1}, staging={(dbg>>1)&1}, reduction={(dbg>>2)&1}): {ms/n*1e3:.2f} us')
