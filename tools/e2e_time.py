"""End-to-end timing of inference_loop on a B2-shaped synthetic problem (dev tool)."""
import sys, time, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch, logging
logging.basicConfig(level=logging.INFO)
from mile_amd import ModelSpec
from mile_amd.config import SamplerConfig
from mile_amd.probabilistic import ProbabilisticModel
from mile_amd.sampling import inference_loop
from mile_amd.tree import PRNGKey
E, N = 128, 1052
rng = np.random.default_rng(0)
X = torch.from_numpy(rng.standard_normal((N, 5)).astype(np.float32)); y = torch.from_numpy(rng.standard_normal(N).astype(np.float32))
pm = ProbabilisticModel(ModelSpec(5, (64, 64, 64, 2)), task='regr')
cfg = SamplerConfig(name='mclmc', warmup_steps=int(sys.argv[1]) if len(sys.argv) > 1 else 2000, n_chains=E, n_samples=int(sys.argv[2]) if len(sys.argv) > 2 else 1000,
                    n_thinning=10, desired_energy_var_start=0.5, desired_energy_var_end=0.1, step_size_init=0.001)
th = torch.from_numpy((0.1 * rng.standard_normal((E, 8834))).astype(np.float32))
with tempfile.TemporaryDirectory() as td:
    t0 = time.time()
    inference_loop(pm.bind(X, y), cfg, PRNGKey(4), th, np.arange(E), Path(td) / 'samples')
    torch.cuda.synchronize()
    t1 = time.time()
    print(f'total {t1 - t0:.2f}s for {cfg.warmup_steps} warmup + {cfg.n_samples} sampling steps x {E} chains')
    print(open(Path(td) / 'warmup_params.txt').read()[:200])
