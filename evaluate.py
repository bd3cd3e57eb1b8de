#!/usr/bin/env python3
"""LPPD of a finished experiment (the consumer of mile_pointwise_loglik; mirrors what the reference's
report notebook does with src/inference/evaluation.py:409-544 + src/inference/metrics.py:247-312 for the
log-score part):

    python evaluate.py -e results/mile_amd/<experiment> [--split test]

Reloads config.yaml and the samples/<chain>/sample_<n>.npz files, rebuilds the data split with the same
seed, evaluates all C x S samples on the split in one device pass and writes metrics.json next to them.
"""
import argparse
import json
from pathlib import Path

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser(description='LPPD / NLL of the samples of an experiment directory')
    ap.add_argument('--exp', '-e', required=True, help='experiment directory (holds config.yaml and samples/)')
    ap.add_argument('--split', default='test', choices=['train', 'valid', 'test'])
    ap.add_argument('--device', default='cuda:0')
    ap.add_argument('--ess-params', type=int, default=256, help='size of the random parameter subset ESS is evaluated on')
    ap.add_argument('--drop-nonfinite', action='store_true',
                    help='leave out chains with non-finite samples (the reference would report NaN; default: keep them)')
    args = ap.parse_args()
    exp = Path(args.exp)
    from mile_amd.callbacks import load_samples_from_dir
    from mile_amd.config import Config
    from mile_amd.metrics import lppd, running_lppd
    from mile_amd.trainer import BDETrainer
    cfg = Config.from_file(exp / 'config.yaml').replace(logging=False)
    tr = BDETrainer.__new__(BDETrainer)            # data + model spec only: no new experiment directory
    tr.build_model(cfg)
    spec = tr.prob_model.spec
    samples = load_samples_from_dir(exp / cfg.training.sampler._dir_name, spec)       # [C, S, d]
    bad_chains = ~np.isfinite(samples).all(axis=(1, 2))
    if args.drop_nonfinite and bad_chains.any() and not bad_chains.all():
        samples = samples[~bad_chains]
    x = getattr(tr.loader, f'{args.split}_x')
    y = getattr(tr.loader, f'{args.split}_y')
    x = np.ascontiguousarray(x).reshape(len(x), -1)
    eng = tr.prob_model.engine(torch.from_numpy(np.ascontiguousarray(tr.loader.train_x).reshape(len(tr.loader.train_x), -1)),
                               torch.from_numpy(np.ascontiguousarray(tr.loader.train_y)), device=args.device)
    pw = eng.pointwise_loglik(torch.from_numpy(samples), torch.from_numpy(x), torch.from_numpy(np.ascontiguousarray(y)))
    out = {'experiment': cfg.experiment_name, 'split': args.split, 'n_chains': int(samples.shape[0]),
           'n_samples': int(samples.shape[1]), 'n_points': int(x.shape[0]),
           'lppd': float(lppd(pw).item()), 'nll_mean': float(-pw.mean().item()),
           'running_lppd_last': float(running_lppd(pw)[-1].item()),
           'nonfinite_chains_total': int(bad_chains.sum()), 'nonfinite_samples': int((~torch.isfinite(torch.from_numpy(samples)).all(dim=-1)).sum().item())}
    # per chain (src/inference/evaluation.py:520-529 prints the same per-chain LPPD): the ensemble figure above is a logsumexp
    # over all chains, which a few chains thrown into a bad region (profiles/r03/01_*) barely move -- their own LPPD shows them
    kept_ids = np.arange(len(bad_chains))[~bad_chains] if (args.drop_nonfinite and bad_chains.any() and not bad_chains.all()) \
        else np.arange(len(bad_chains))
    pc = [float(lppd(pw[c:c + 1]).item()) for c in range(pw.shape[0])]
    out['chain_ids'] = [int(c) for c in kept_ids]
    out['per_chain_lppd'] = pc
    out['per_chain_lppd_median'] = float(np.nanmedian(pc))
    # RMSE of the posterior-mean prediction (src/inference/evaluation.py:509-518: mean over (chain, sample) of the predicted
    # mean, regression FCNs): a plain torch forward of the Dense stack on the device -- evaluation tooling, not the hot path
    if cfg.data.task == 'regr' and cfg.model.model == 'FCN':
        dev = torch.device(args.device)
        xt = torch.from_numpy(x).to(dev)
        flat = torch.from_numpy(samples.reshape(-1, samples.shape[-1]))
        act = {'relu': torch.relu, 'tanh': torch.tanh, 'sigmoid': torch.sigmoid}[cfg.model.activation]
        leaves = {n: (o, sh) for n, o, sh in spec.leaves()}
        nl = len(spec.hidden_structure)
        mu_sum = torch.zeros(x.shape[0], dtype=torch.float64, device=dev)
        n_ok = 0
        S_ = samples.shape[1]
        mu_chain = torch.zeros((samples.shape[0], x.shape[0]), dtype=torch.float64, device=dev)     # per-chain mean prediction
        for c0 in range(0, flat.shape[0], 512):
            th = flat[c0:c0 + 512].to(dev)
            h = xt[None].expand(th.shape[0], -1, -1)
            for li in range(nl):
                ob, shb = leaves[f'fcn.layer{li}.bias']
                ok_, shk = leaves[f'fcn.layer{li}.kernel']
                h = torch.baddbmm(th[:, None, ob:ob + shb[0]], h, th[:, ok_:ok_ + shk[0] * shk[1]].reshape(-1, shk[0], shk[1]))
                if li < nl - 1:
                    h = act(h)
            mu = h[..., 0]
            fin_rows = torch.isfinite(mu).all(dim=1)
            mu_sum += mu[fin_rows].double().sum(dim=0)
            n_ok += int(fin_rows.sum())
            idx = torch.arange(c0, c0 + th.shape[0], device=dev) // S_
            mu_chain.index_add_(0, idx, torch.nan_to_num(mu.double(), nan=0.0, posinf=0.0, neginf=0.0))
        if n_ok:
            yt = torch.from_numpy(np.ascontiguousarray(y)).to(dev).double().reshape(-1)
            out['rmse'] = float(torch.sqrt(((yt - mu_sum / n_ok) ** 2).mean()).item())
            rc = torch.sqrt(((yt[None] - mu_chain / S_) ** 2).mean(dim=1))
            out['per_chain_rmse'] = [float(v) for v in rc.cpu()]
            out['per_chain_rmse_median'] = float(rc.median().item())
    # dead chains: a tuned step size of 0 / NaN in warmup_params.txt (profiles/r03/01_dead_chains_mechanism.md)
    wp = exp / 'warmup_params.txt'
    if wp.exists():
        eps = np.array([float(v) for v in wp.read_text().split('\n')[0].split(',')])
        out['dead_chains_step_size_zero_or_nan'] = int((~(np.isfinite(eps) & (eps > 0))).sum())
    # ESS and ESS/s (BASELINE.json's metric; src/inference/metrics.py:386-405 on a parameter subset): per-chain ESS of
    # each selected parameter, summed over chains, min / median over the subset, per second of `time.sampling`
    from mile_amd.metrics import effective_sample_size
    fin = np.isfinite(samples).all(axis=(1, 2))             # (samples may already have been filtered by --drop-nonfinite)
    ok = samples[fin] if (fin.any() and not fin.all()) else samples
    rng = np.random.default_rng(0)
    cols = np.sort(rng.choice(samples.shape[2], size=min(args.ess_params, samples.shape[2]), replace=False))
    if ok.shape[1] >= 8:
        ess = effective_sample_size(torch.from_numpy(np.ascontiguousarray(ok[:, :, cols])).to(args.device)).sum(dim=0).cpu().numpy()
        t_sampling = None
        log = exp / 'training.log'
        if log.exists():
            import re
            m = re.findall(r'time\.sampling took ([0-9.]+) seconds', log.read_text())
            t_sampling = float(m[-1]) if m else None
        out.update({'ess_params': int(len(cols)), 'ess_min': float(np.nanmin(ess)), 'ess_median': float(np.nanmedian(ess)),
                    'time_sampling_s': t_sampling,
                    'ess_per_s_min': float(np.nanmin(ess) / t_sampling) if t_sampling else None,
                    'ess_per_s_median': float(np.nanmedian(ess) / t_sampling) if t_sampling else None})
    (exp / 'metrics.json').write_text(json.dumps(out, indent=1) + '\n')
    print(json.dumps({k: v for k, v in out.items() if not isinstance(v, list)}))        # the per-chain arrays stay in metrics.json


if __name__ == '__main__':
    main()
