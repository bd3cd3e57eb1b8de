"""Time the grad kernels and a few MCLMC steps on the B3 shape (protein-like: F=9, [128,128,128,2],
N=36000, E=512).  Dev tool.  usage: python tools/b3_time.py [E] [N] [kernels comma-separated]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from mile_amd import ModelSpec
from mile_amd.engine import Engine

E = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 36000
kernels = sys.argv[3].split(',') if len(sys.argv) > 3 else ['mfma_w128_bf16', 'generic']
spec = ModelSpec(9, (128, 128, 128, 2))
d = spec.n_params
W = 9 * 128 + 2 * 128 * 128 + 128 * 2
flop = E * N * (6 * W - 2 * 9 * 128)
rng = np.random.default_rng(0)
X = torch.from_numpy(rng.standard_normal((N, 9)).astype(np.float32))
y = torch.from_numpy(rng.standard_normal(N).astype(np.float32))
th = torch.from_numpy((0.1 * rng.standard_normal((E, d))).astype(np.float32)).cuda()
for k in kernels:
    eng = Engine(spec, X, y, device='cuda:0', grad_kernel=k)
    reps = 20 if k != 'generic' else 2
    eng.logpost_grad(th)
    torch.cuda.synchronize()
    eng.grad_timing_begin()
    for _ in range(reps):
        eng.logpost_grad(th)
    torch.cuda.synchronize()
    ms, n = eng.grad_timing_end()
    info = eng.grad_launch_info(E)
    print(f'{k:16s} grid={info["grid"]} lds={info["lds_bytes"]} ms/launch={ms / n:9.3f}  {flop / (ms / n * 1e-3) / 1e12:8.1f} TFLOP/s',
          flush=True)
    if k != 'generic':
        st = eng.init(th, seed=1)
        eps = torch.full((E,), 1e-3, device='cuda'); L = torch.full((E,), float(np.sqrt(d)) * 1e-2, device='cuda')
        st, info_, _ = eng.step(st, eps, L, n_steps=3, seed=2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st, info_, _ = eng.step(st, eps, L, n_steps=20, seed=3, step_offset=3)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f'{k:16s} MCLMC step {dt * 1e3:8.3f} ms -> {E / dt:10.0f} particle-steps/s; finite={bool(torch.isfinite(st.position).all())}',
              flush=True)
