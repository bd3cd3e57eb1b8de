"""dev: time one ensemble gradient of LeNet at BASELINE config 5's shape (3 x 32 x 32, 10 classes) on a grad kernel.
usage: lenet_time.py [kernel=lenet_f32|lenet_bf16] [N=4000] [E=256]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import lenet_oracle as LN
from mile_amd import LeNetSpec
from mile_amd.engine import Engine
kernel = sys.argv[1] if len(sys.argv) > 1 else 'lenet_f32'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
E = int(sys.argv[3]) if len(sys.argv) > 3 else 256
ospec = LN.LeNetSpec(3, 32, 32, 10)
prob = LN.synthetic_problem(ospec, N, E, seed=0)
spec = LeNetSpec(3, 32, 32, 10)
eng = Engine(spec, torch.from_numpy(prob['X']), torch.from_numpy(prob['y']), device='cuda:0', grad_kernel=kernel)
th = torch.from_numpy(prob['theta0']).cuda()
lp, g = eng.logpost_grad(th)
torch.cuda.synchronize()
n = 3
t0 = time.perf_counter()
for _ in range(n):
    lp, g = eng.logpost_grad(th)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
flop = 3 * 2 * (1024 * 75 * 6 + 144 * 150 * 16 + 576 * 120 + 120 * 84 + 84 * 10) * N * E
print(f'{eng.grad_kernel}: N={N} E={E} d={spec.n_params}  {dt * 1e3:.1f} ms per ensemble gradient  {flop / dt / 1e12:.1f} TFLOP/s (algorithmic)  '
      f'finite={bool(torch.isfinite(g).all())}  |g|={g.norm().item():.6g} lp0={lp[0].item():.6g}')
