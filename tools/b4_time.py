"""dev: time one ensemble gradient at BASELINE config B4's full size on a grad kernel.  usage: b4_time.py [kernel] [rows] [E]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import mclmc_oracle as O
from mile_amd import ModelSpec
from mile_amd.engine import Engine
kernel = sys.argv[1] if len(sys.argv) > 1 else 'auto'
ospec, N, E = O.config_spec('B4')
N = int(sys.argv[2]) if len(sys.argv) > 2 else N
E = int(sys.argv[3]) if len(sys.argv) > 3 else E
prob = O.synthetic_problem(ospec, N, E, seed=0)
spec = ModelSpec(ospec.in_features, ospec.hidden_structure, activation='relu', task='classification')
eng = Engine(spec, torch.from_numpy(prob['X']), torch.from_numpy(prob['y']), device='cuda:0', grad_kernel=kernel)
th = torch.from_numpy(prob['theta0']).cuda()
lp, g = eng.logpost_grad(th)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 3
for _ in range(n):
    lp, g = eng.logpost_grad(th)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
W = sum(a * b for a, b in zip((54, 256, 256, 256, 256), (256, 256, 256, 256, 7)))
fl = E * N * (6 * W - 2 * 54 * 256)
print(f'{eng.grad_kernel}: N={N} E={E}: {dt * 1e3:.1f} ms per ensemble gradient = {fl / dt / 1e12:.1f} TFLOP/s algorithmic; finite={bool(torch.isfinite(g).all())}')
