"""Time mile_logpost_grad for the LeNet target.  usage: python tools/lenet_time.py C H W out_dim N E [reps]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from mile_amd import LeNetSpec
from mile_amd.engine import Engine

C, H, W, K, N, E = (int(v) for v in sys.argv[1:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 3
spec = LeNetSpec(C, H, W, K)
d = spec.n_params
hp1, wp1 = H // 2, W // 2
h2, w2 = hp1 - 4, wp1 - 4
fwd = 2 * (H * W * 25 * C * 6 + h2 * w2 * 150 * 16 + spec.flat * 120 + 120 * 84 + 84 * K)
flop = E * N * (3 * fwd - 2 * H * W * 25 * C * 6)
rng = np.random.default_rng(0)
X = torch.from_numpy(rng.standard_normal((N, C, H, W)).astype(np.float32))
y = torch.from_numpy(rng.integers(0, K, N).astype(np.int32))
th = torch.from_numpy((0.05 * rng.standard_normal((E, d))).astype(np.float32)).cuda()
eng = Engine(spec, X, y, device='cuda:0')
eng.logpost_grad(th)
torch.cuda.synchronize()
eng.grad_timing_begin()
for _ in range(reps):
    eng.logpost_grad(th)
torch.cuda.synchronize()
ms, n = eng.grad_timing_end()
print(f'LeNet {C}x{H}x{W}->{K} d={d} N={N} E={E}: ms/grad={ms / n:10.2f}  {flop / (ms / n * 1e-3) / 1e12:7.2f} TFLOP/s', flush=True)
