"""dev: time the LeNet evaluation forward (pointwise_loglik: conv kernels without the full-size stores when lenet_bf16) and,
with MILE_CM_SKIP knobs, what the parts of k_conv5m_fwd cost.  usage: lenet_fwd_time.py [N=4000] [E=256]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import lenet_oracle as LN
from mile_amd import LeNetSpec
from mile_amd.engine import Engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ospec = LN.LeNetSpec(3, 32, 32, 10)
prob = LN.synthetic_problem(ospec, N, E, seed=0)
eng = Engine(LeNetSpec(3, 32, 32, 10), torch.from_numpy(prob['X']), torch.from_numpy(prob['y']), device='cuda:0', grad_kernel='lenet_bf16')
th = torch.from_numpy(prob['theta0']).cuda()
for skip in (0, 1, 2, 3, 4, 8, 12, 15):
    os.environ['MILE_CM_SKIP'] = str(skip)
    eng.logpost_grad(th); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        eng.logpost_grad(th)
    torch.cuda.synchronize()
    print(f'MILE_CM_SKIP={skip:2d}: {(time.perf_counter() - t0) / 2 * 1e3:7.1f} ms per gradient', flush=True)
