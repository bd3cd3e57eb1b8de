"""Time mile_logpost_grad on an arbitrary FCN shape.  Dev tool.
usage: python tools/shape_time.py F h1,h2,..,out task N E kernel[,kernel...] [reps] [activation]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from mile_amd import ModelSpec
from mile_amd.engine import Engine

F = int(sys.argv[1]); hs = tuple(int(v) for v in sys.argv[2].split(',')); task = sys.argv[3]
N = int(sys.argv[4]); E = int(sys.argv[5]); kernels = sys.argv[6].split(','); reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
act = sys.argv[8] if len(sys.argv) > 8 else 'relu'
spec = ModelSpec(F, hs, task=task, activation=act)
d = spec.n_params
fin, W = F, 0
for w in hs:
    W += fin * w; fin = w
flop = E * N * (6 * W - 2 * F * hs[0])
rng = np.random.default_rng(0)
X = torch.from_numpy(rng.standard_normal((N, F)).astype(np.float32))
y = torch.from_numpy(rng.standard_normal(N).astype(np.float32)) if task == 'regr' else \
    torch.from_numpy(rng.integers(0, hs[-1], N).astype(np.int32))
th = torch.from_numpy((0.1 * rng.standard_normal((E, d))).astype(np.float32)).cuda()
ref = None
for k in kernels:
    eng = Engine(spec, X, y, device='cuda:0', grad_kernel=k)
    lp, g = eng.logpost_grad(th)
    torch.cuda.synchronize()
    eng.grad_timing_begin()
    for _ in range(reps):
        eng.logpost_grad(th)
    torch.cuda.synchronize()
    ms, n = eng.grad_timing_end()
    err = '' if ref is None else f' max rel diff vs {kernels[0]}: {float((g - ref).abs().max() / ref.abs().max()):.2e}'
    if ref is None:
        ref = g.clone()
    print(f'{k:16s} d={d} ms/grad={ms / n:10.3f}  {flop / (ms / n * 1e-3) / 1e12:8.1f} TFLOP/s{err}', flush=True)
