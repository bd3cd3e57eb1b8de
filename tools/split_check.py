"""fp32-MFMA vs split-bf16 (bf16x3, six products) width-64 grad kernel: time and error of BOTH against the fp64 oracle.
Dev tool.  usage: python tools/split_check.py [N] [E] [nh] [F]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from mile_amd import ModelSpec
from mile_amd.engine import Engine
from oracle import mclmc_oracle as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1052
E = int(sys.argv[2]) if len(sys.argv) > 2 else 128
nh = int(sys.argv[3]) if len(sys.argv) > 3 else 3
F = int(sys.argv[4]) if len(sys.argv) > 4 else 5
hs = (64,) * nh + (2,)
spec = ModelSpec(F, hs)
so = O.ModelSpec(in_features=F, hidden_structure=hs)
prob = O.synthetic_problem(so, N, E, seed=1)
X, y, th = prob['X'], prob['y'], prob['theta0']
Eo = min(E, 8)
lp64, g64 = O.logpost_and_grad(so, th[:Eo].astype(np.float64), X.astype(np.float64), y.astype(np.float64))
thc = torch.from_numpy(th).cuda()
for k in ('mfma_w64', 'mfma_w64_bf16x3', 'generic'):
    eng = Engine(spec, torch.from_numpy(X), torch.from_numpy(y), device='cuda:0', grad_kernel=k)
    lp, g = eng.logpost_grad(thc)
    torch.cuda.synchronize()
    eng.grad_timing_begin()
    for _ in range(20):
        eng.logpost_grad(thc)
    torch.cuda.synchronize()
    ms, n = eng.grad_timing_end()
    gd = g[:Eo].double().cpu().numpy()
    scale = np.abs(g64).max(axis=1, keepdims=True)
    err = np.abs(gd - g64) / scale
    lpe = np.abs(lp[:Eo].double().cpu().numpy() - lp64) / np.abs(lp64)
    print(f'{k:18s} us/grad={1e3 * ms / n:8.1f}  grad err vs fp64 (rel. to max|g|): max {err.max():.2e} mean {err.mean():.2e}   logp rel err max {lpe.max():.2e}', flush=True)
