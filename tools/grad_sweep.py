"""Time mile_logpost_grad's grad kernel (HIP events inside the library) vs N to separate the
fixed cost per launch from the cost per 32-row block.  Dev tool, not part of the product."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from mile_amd import ModelSpec
from mile_amd.engine import Engine

E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
spec = ModelSpec(5, (64, 64, 64, 2))
rng = np.random.default_rng(0)
res = []
for N in (128, 256, 512, 1024, 1052, 1280, 1536, 2048, 4096):
    X = torch.from_numpy(rng.standard_normal((N, 5)).astype(np.float32))
    y = torch.from_numpy(rng.standard_normal(N).astype(np.float32))
    eng = Engine(spec, X, y, device='cuda:0')
    th = torch.from_numpy((0.1 * rng.standard_normal((E, spec.n_params))).astype(np.float32)).cuda()
    for _ in range(5):
        eng.logpost_grad(th)
    torch.cuda.synchronize()
    eng.grad_timing_begin()
    for _ in range(50):
        eng.logpost_grad(th)
    torch.cuda.synchronize()
    ms, n = eng.grad_timing_end()
    info = eng.grad_launch_info(E)
    nb = (N + 31) // 32
    res.append((N, nb, info['grid'], ms / n * 1e3))
    print(f'N={N:5d} blocks={nb:4d} grid={info["grid"]} us/launch={ms / n * 1e3:8.2f}', flush=True)
