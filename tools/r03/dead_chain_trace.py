#!/usr/bin/env python3
"""why do chains die under the reference's stock tuner targets (desired_energy_var 0.5 -> 0.1)?

Re-runs the warm-up of a YAML experiment (warm-start training included, same seeds as train.py) with the device tuner in
chunks, recording per chunk the adaptive state of every chain and per step the MCLMCInfo triple.  For the first chains whose
step size leaves (0, inf) it replays the offending chunk from the snapshot taken at its start, ONE step per mile_tune call,
and dumps per step: step index, eps, step_size_max, time, x_average, logdensity, kinetic / energy change, |g|, min / max
sigma of the Gaussian head, number of clipped rows, finiteness of position / momentum / gradient.

    python tools/r03/dead_chain_trace.py experiments/mclmc_airfoil_b2.yaml gpurun_out/r3a/trace_b2.json [max_dead]
"""
import json
import math
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from mile_amd.config import Config                      # noqa: E402
from mile_amd.engine import IntegratorState             # noqa: E402
from mile_amd.trainer import BDETrainer                 # noqa: E402
from mile_amd.callbacks import load_params_batch        # noqa: E402


def head_sigma(spec, theta, X):
    """min / max sigma = clip(exp(out[:, 1])) and number of clipped rows per chain (torch fp32 forward of the FCN)."""
    E = theta.shape[0]
    h = X[None].expand(E, -1, -1)
    leaves = {n: (o, sh) for n, o, sh in spec.leaves()}
    nl = len(spec.hidden_structure)
    for li in range(nl):
        ob, shb = leaves[f'fcn.layer{li}.bias']
        ok, shk = leaves[f'fcn.layer{li}.kernel']
        b = theta[:, ob:ob + shb[0]]
        W = theta[:, ok:ok + shk[0] * shk[1]].reshape(E, shk[0], shk[1])
        h = torch.baddbmm(b[:, None, :], h, W)
        if li < nl - 1:
            h = torch.relu(h)
    s = h[..., 1]
    es = torch.exp(s)
    clipped = ((es <= 1e-6) | (es >= 1e6)).sum(dim=1)
    sig = es.clamp(1e-6, 1e6)
    return sig.min(dim=1).values, sig.max(dim=1).values, clipped, h[..., 0]


def main():
    yaml_path, out_path = sys.argv[1], Path(sys.argv[2])
    max_dead = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    chunk = 100
    cfg = Config.from_file(yaml_path)
    tmp = tempfile.mkdtemp(prefix='mile_trace_')
    cfg = cfg.replace(saving_dir=tmp)
    tr = BDETrainer(cfg)
    tr.train_warmstart()
    sc = cfg.training.sampler
    ws = cfg.training.warmstart
    warm = Path(tr.exp_dir) / ws._dir_name
    chains = sorted((p for p in warm.iterdir() if p.name.startswith('params')), key=lambda p: int(p.stem.split('_')[-1]))
    E = len(chains)
    params = torch.from_numpy(load_params_batch(chains, tr.prob_model.spec))
    x = torch.from_numpy(np.ascontiguousarray(tr.loader.train_x).reshape(len(tr.loader.train_x), -1))
    y = torch.from_numpy(np.ascontiguousarray(tr.loader.train_y))
    if getattr(tr, '_engine_inputs', None) is not None:
        x, y = tr._engine_inputs
    eng = tr.prob_model.engine(x, y)
    dev = eng.device
    Xd, yd = x.to(dev).float(), y.to(dev).float().reshape(-1)
    spec = tr.prob_model.spec
    d = eng.d
    key = tr.key
    _, warmup_key, _ = key.split(3)
    chain_ids = torch.arange(E, dtype=torch.int32)
    state = eng.init(params, seed=warmup_key.seed, particle_ids=chain_ids)
    part1_key, _ = warmup_key.split(2)
    n = sc.warmup_steps
    tune1, tune2 = int(n * 0.8), int(n * 0.1)
    total = tune1 + tune2 + 1
    f32 = dict(dtype=torch.float32, device=dev)
    L = torch.full((E,), max(math.sqrt(d), 15.0), **f32)
    tuner = {'step_size': torch.full((E,), float(sc.step_size_init), **f32),
             'step_size_max': torch.full((E,), float('inf'), **f32), 'time': torch.zeros(E, **f32),
             'x_average': torch.zeros(E, **f32), 'stream_weight': torch.zeros(E, **f32),
             'stream_average': torch.zeros((E, 2, d), **f32)}
    decay = (sc.num_effective_samples - 1.0) / (sc.num_effective_samples + 1.0)
    kw = dict(n_mask_steps=tune1, schedule_total=total, desired_energy_var_start=sc.desired_energy_var_start,
              desired_energy_var_end=sc.desired_energy_var_end, trust_in_estimate=sc.trust_in_estimate, decay_rate=decay,
              seed=part1_key.seed, particle_ids=chain_ids)

    def bad_eps(t):
        e = t['step_size']
        return ~(torch.isfinite(e) & (e > 0))

    fixtures = {}
    dead_first = {}                     # chain -> chunk start step
    replays = []
    summary = []                        # per chunk: step, median eps, n_dead, max |dE|
    done = 0
    while done < tune1 + tune2:
        c = min(chunk, tune1 + tune2 - done)
        snap_state = IntegratorState(*(t.clone() for t in state))
        snap_tuner = {k: v.clone() for k, v in tuner.items()}
        was_bad = bad_eps(tuner)
        info = eng.tune(state, tuner, L, c, schedule_step0=done, step_offset=done, want_info=True, **kw)
        now_bad = bad_eps(tuner)
        new = torch.nonzero(now_bad & ~was_bad).flatten().tolist()
        dE = info.energy_change
        summary.append({'step': done, 'eps_median': float(tuner['step_size'].median()), 'eps_min': float(tuner['step_size'].min()),
                        'eps_max': float(tuner['step_size'][torch.isfinite(tuner['step_size'])].max()),
                        'n_bad_eps': int(now_bad.sum()), 'absdE_max_finite': float(dE[torch.isfinite(dE)].abs().max()),
                        'n_nonfinite_dE': int((~torch.isfinite(dE)).sum())})
        for ch in new:
            dead_first[ch] = done
        if new and len(replays) < max_dead:
            take = new[:max_dead - len(replays)]
            rs = IntegratorState(*(t.clone() for t in snap_state))
            rt = {k: v.clone() for k, v in snap_tuner.items()}
            rows = {ch: [] for ch in take}
            for i in range(c):
                pre = {k: rt[k][take].tolist() for k in ('step_size', 'step_size_max', 'time', 'x_average')}
                gpre = torch.linalg.vector_norm(rs.logdensity_grad[take], dim=1).tolist()
                inf1 = eng.tune(rs, rt, L, 1, schedule_step0=done + i, step_offset=done + i, want_info=True, **kw)
                smin, smax, ncl, mu = head_sigma(spec, rs.position[take], Xd)
                resid = (yd[None] - mu).abs().max(dim=1).values
                for j, ch in enumerate(take):
                    rows[ch].append({
                        'step': done + i, 'eps_in': pre['step_size'][j], 'eps_max_in': pre['step_size_max'][j],
                        'time_in': pre['time'][j], 'x_average_in': pre['x_average'][j], 'gnorm_in': gpre[j],
                        'logdensity': float(inf1.logdensity[0, ch]), 'kinetic_change': float(inf1.kinetic_change[0, ch]),
                        'energy_change': float(inf1.energy_change[0, ch]),
                        'eps_out': float(rt['step_size'][ch]), 'eps_max_out': float(rt['step_size_max'][ch]),
                        'time_out': float(rt['time'][ch]), 'x_average_out': float(rt['x_average'][ch]),
                        'gnorm_out': float(torch.linalg.vector_norm(rs.logdensity_grad[ch])),
                        'finite_x': bool(torch.isfinite(rs.position[ch]).all()),
                        'finite_u': bool(torch.isfinite(rs.momentum[ch]).all()),
                        'finite_g': bool(torch.isfinite(rs.logdensity_grad[ch]).all()),
                        'sigma_min': float(smin[j]), 'sigma_max': float(smax[j]), 'rows_clipped': int(ncl[j]),
                        'max_abs_resid': float(resid[j])})
            # keep the steps around the transition only (the 12 before the first bad eps_out and 3 after)
            for ch in take:
                r = rows[ch]
                # state + adaptive state one step BEFORE the first catastrophic step (|dE| > 1e6), for the oracle replays
                kc = next((i for i, row in enumerate(r) if not abs(row['energy_change']) < 1e6), None)
                if kc is not None and kc >= 1:
                    s2 = IntegratorState(*(t.clone() for t in snap_state))
                    t2 = {k_: v.clone() for k_, v in snap_tuner.items()}
                    if kc - 1 > 0:
                        eng.tune(s2, t2, L, kc - 1, schedule_step0=done, step_offset=done, **kw)
                    fixtures[ch] = {'step0': done + kc - 1, 'x': s2.position[ch].cpu().numpy(), 'u': s2.momentum[ch].cpu().numpy(),
                                    'g': s2.logdensity_grad[ch].cpu().numpy(), 'logp': float(s2.logdensity[ch]),
                                    **{k_: float(t2[k_][ch]) for k_ in ('step_size', 'step_size_max', 'time', 'x_average')},
                                    'device_rows': r[kc - 1:kc + 14]}
                k = next((i for i, row in enumerate(r) if not (math.isfinite(row['eps_out']) and row['eps_out'] > 0)), len(r) - 1)
                replays.append({'chain': ch, 'chunk_start': done, 'first_bad_step': r[k]['step'],
                                'replay_matches_chunk': bool(torch.equal(bad_eps(rt)[take], now_bad[take])),
                                'rows': r[max(0, k - 12):k + 4],
                                'eps_history_chunk': [row['eps_out'] for row in r],
                                'dE_history_chunk': [row['energy_change'] for row in r]})
        done += c
    final_bad = bad_eps(tuner)
    out = {'yaml': yaml_path, 'E': E, 'd': d, 'tune1': tune1, 'tune2': tune2, 'chunk': chunk,
           'n_dead_after_phase12': int(final_bad.sum()), 'dead_chains': sorted(dead_first),
           'dead_first_chunk': {str(k): v for k, v in sorted(dead_first.items())},
           'final_step_size': tuner['step_size'].tolist(), 'summary_every_50_chunks': summary[::50], 'replays': replays}
    out_path.parent.mkdir(parents=True, exist_ok=True)
    out_path.write_text(json.dumps(out, indent=1))
    if fixtures:
        ch = sorted(fixtures)[0]
        fx = fixtures[ch]
        np.savez_compressed(out_path.with_suffix('.fixture.npz'), X=Xd.cpu().numpy(), y=yd.cpu().numpy(), chain=np.int32(ch),
                            seed=np.uint64(part1_key.seed), step0=np.int64(fx['step0']), x=fx['x'], u=fx['u'], g=fx['g'],
                            logp=np.float32(fx['logp']), L=np.float32(L[0].item()),
                            tuner=np.array([fx['step_size'], fx['step_size_max'], fx['time'], fx['x_average']], np.float32),
                            schedule=np.array([tune1, tune2, total], np.int64),
                            targets=np.array([sc.desired_energy_var_start, sc.desired_energy_var_end, sc.trust_in_estimate, decay], np.float64),
                            hidden=np.array(spec.hidden_structure, np.int32),
                            device_rows=json.dumps(fx['device_rows']))
    print(json.dumps({k: out[k] for k in ('E', 'd', 'n_dead_after_phase12', 'dead_first_chunk')}))
    for r in replays:
        print(f"--- chain {r['chain']}: first bad step {r['first_bad_step']} (replay matches chunk: {r['replay_matches_chunk']})")
        for row in r['rows']:
            print({k: (f'{v:.4g}' if isinstance(v, float) else v) for k, v in row.items()})


if __name__ == '__main__':
    main()
