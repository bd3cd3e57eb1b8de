#!/usr/bin/env python3
"""BASELINE config 3 on the reference's protein table: the bf16-operand kernel against the fp32-faithful path after an EQUAL
step count from the SAME state with the SAME tuned (step_size, L) and the SAME Philox streams (the +-1 %
LPPD statement of tests/test_gpu_e2e.py::test_bf16_kernel_lppd_within_one_percent_of_fp32 on real rows).

Warm-start training and warm-up run once (fp32-faithful kernel, the YAML's schedule); the sampling phase then runs twice from
the warm-up's final state.  usage: protein_equal_steps.py <out.json> [var_start var_end]
"""
import json
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from mile_amd.callbacks import load_params_batch            # noqa: E402
from mile_amd.config import Config                            # noqa: E402
from mile_amd.engine import IntegratorState                   # noqa: E402
from mile_amd.metrics import lppd                             # noqa: E402
from mile_amd.trainer import BDETrainer                       # noqa: E402
from mile_amd.warmup import custom_mclmc_warmup               # noqa: E402


def main():
    out_path = Path(sys.argv[1])
    cfg = Config.from_file(ROOT / 'experiments' / 'mclmc_protein_b3.yaml')
    cfg = cfg.replace(saving_dir=tempfile.mkdtemp(prefix='p3eq_'))
    sc = cfg.training.sampler
    vs = float(sys.argv[2]) if len(sys.argv) > 2 else sc.desired_energy_var_start
    ve = float(sys.argv[3]) if len(sys.argv) > 3 else sc.desired_energy_var_end
    tr = BDETrainer(cfg)
    t0 = time.time()
    tr.train_warmstart()
    t_ws = time.time() - t0
    warm = Path(tr.exp_dir) / cfg.training.warmstart._dir_name
    chains = sorted((p for p in warm.iterdir() if p.name.startswith('params')), key=lambda p: int(p.stem.split('_')[-1]))
    params = torch.from_numpy(load_params_batch(chains, tr.prob_model.spec))
    x, y = tr._engine_inputs
    eng = tr.prob_model.engine(x, y)
    E = params.shape[0]
    ids = torch.arange(E, dtype=torch.int32)
    eng.set_grad_kernel('auto')
    assert eng.grad_kernel == 'mfma_wide_bf16x3'
    key = tr.key
    _, warmup_key, sample_key = key.split(3)
    algo = custom_mclmc_warmup(tr.prob_model.bind(x, y), diagonal_preconditioning=sc.diagonal_preconditioning,
                               desired_energy_var_start=vs, desired_energy_var_end=ve, trust_in_estimate=sc.trust_in_estimate,
                               num_effective_samples=sc.num_effective_samples, step_size_init=sc.step_size_init, chain_ids=ids)
    t0 = time.time()
    state0, prm = algo.run(warmup_key, params, sc.warmup_steps)
    torch.cuda.synchronize()
    t_wu = time.time() - t0
    eps, L = prm.step_size, prm.L
    alive = torch.isfinite(eps) & (eps > 0) & torch.isfinite(L) & (L > 0) & torch.isfinite(state0.position).all(dim=1)
    tx = torch.from_numpy(np.ascontiguousarray(tr.loader.test_x)).float()
    ty = torch.from_numpy(np.ascontiguousarray(tr.loader.test_y)).float()
    res = {'var_targets': [vs, ve], 'warmstart_s': round(t_ws, 1), 'warmup_s': round(t_wu, 1), 'n_chains': E,
           'alive_after_warmup': int(alive.sum()), 'eps_median': float(eps[alive].median()), 'L_median': float(L[alive].median())}
    pw_all = {}
    for k in ('mfma_wide_bf16x3', 'mfma_w128_bf16'):
        eng.set_grad_kernel(k)
        st = IntegratorState(*(t.clone() for t in state0))
        kept = []
        torch.cuda.synchronize(); t0 = time.time()
        done = 0
        while done < sc.n_samples:
            c = min(100, sc.n_samples - done)
            st, _, smp = eng.step(st, eps, L, n_steps=c, seed=sample_key.seed, step_offset=done, n_thinning=sc.n_thinning,
                                  particle_ids=ids, want_info=False, inplace=True)
            kept.append(smp)
            done += c
        torch.cuda.synchronize()
        t_s = time.time() - t0
        smp = torch.cat(kept, dim=0).permute(1, 0, 2).contiguous()            # [C, S, d]
        fin = torch.isfinite(smp).all(dim=2).all(dim=1) & alive.to(smp.device)
        eng.set_grad_kernel('auto')                                         # evaluation in fp32 for both
        pw = eng.pointwise_loglik(smp, tx, ty)                              # [C, S, Nt]
        pw_all[k] = (pw, fin)
        pc = torch.stack([lppd(pw[c:c + 1]) for c in range(E)])
        res[k] = {'sampling_s': round(t_s, 2), 'finite_chains': int(fin.sum()), 'lppd_all_finite': float(lppd(pw[fin]).item()),
                  'per_chain_lppd_median': float(pc[fin].median().item())}
    both = pw_all['mfma_wide_bf16x3'][1] & pw_all['mfma_w128_bf16'][1]
    a = lppd(pw_all['mfma_wide_bf16x3'][0][both]).item()
    b = lppd(pw_all['mfma_w128_bf16'][0][both]).item()
    pca = torch.stack([lppd(pw_all['mfma_wide_bf16x3'][0][c:c + 1]) for c in range(E)])[both]
    pcb = torch.stack([lppd(pw_all['mfma_w128_bf16'][0][c:c + 1]) for c in range(E)])[both]
    healthy = (pca > -2) & (pcb > -2)
    res['common_finite_chains'] = int(both.sum())
    res['lppd_fp32_faithful'] = a
    res['lppd_bf16'] = b
    res['lppd_rel_diff'] = abs(a - b) / abs(a)
    res['healthy_in_both'] = int(healthy.sum())
    res['per_chain_lppd_mean_fp32_faithful'] = float(pca[healthy].mean().item())
    res['per_chain_lppd_mean_bf16'] = float(pcb[healthy].mean().item())
    res['per_chain_abs_diff_median'] = float((pca - pcb)[healthy].abs().median().item())
    out_path.parent.mkdir(parents=True, exist_ok=True)
    out_path.write_text(json.dumps(res, indent=1))
    print(json.dumps(res))


if __name__ == '__main__':
    main()
