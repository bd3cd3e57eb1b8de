#!/usr/bin/env python3
"""Replay the dead-chain fixture of tools/r03/dead_chain_trace.py in the CPU oracle, in float32 AND float64, on the noise the
device drew (Philox, keyed by seed / chain id / step / stage), next to the device's own record of the same steps.

    python tools/r03/replay_fixture.py tests/golden/dead_chain_b2.npz [n_steps]
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import mclmc_oracle as O   # noqa: E402


def replay(fx, dt, n_steps):
    spec = O.ModelSpec(fx['X'].shape[1], tuple(int(v) for v in fx['hidden']))
    X, y = fx['X'].astype(dt), fx['y'].astype(dt)
    f = lambda th: O.logpost_and_grad(spec, th, X, y)
    d = spec.n_params
    st = O.State(fx['x'][None].astype(dt), fx['u'][None].astype(dt), np.array([fx['logp']], dt), fx['g'][None].astype(dt))
    # the gradient / logdensity leaves of the state are the device's; re-evaluate them in this precision for the fp64 run
    if dt == np.float64:
        l, g = f(st.position)
        st.logdensity, st.logdensity_grad = l, g
    eps0, eps_max, time, xavg = (dt(v) for v in fx['tuner'])
    ad = O.AdaptiveState(np.array([time], dt), np.array([xavg], dt), np.array([eps_max], dt), np.zeros(1, dt), np.zeros((1, 2, d), dt))
    eps = np.array([eps0], dt)
    L = np.array([fx['L']], dt)
    tune1, tune2, total = (int(v) for v in fx['schedule'])
    v0, v1, trust, decay = (float(v) for v in fx['targets'])
    ids = np.array([int(fx['chain'])])
    seed, step0 = int(fx['seed']), int(fx['step0'])
    rows = []
    for i in range(n_steps):
        k = step0 + i
        z1 = O.philox_normal(seed, ids, k, 0, d, dtype=np.float32).astype(dt)
        z2 = O.philox_normal(seed, ids, k, 1, d, dtype=np.float32).astype(dt)
        var = O.desired_energy_var(k, total, v0, v1)
        eps_in = eps.copy()
        st, eps, ok, info = O.tuner_step(f, st, eps, L, np.ones((1, d), dt), z1, z2, ad, mask=1.0 if k < tune1 else 0.0, var=var,
                                         trust_in_estimate=trust, decay=decay)
        rows.append({'step': k, 'eps_in': float(eps_in[0]), 'logdensity': float(info.logdensity[0]),
                     'kinetic_change': float(info.kinetic_change[0]), 'energy_change': float(info.energy_change[0]),
                     'eps_out': float(eps[0]), 'x_average_out': float(ad.x_average[0]), 'time_out': float(ad.time[0]),
                     'gnorm_out': float(np.linalg.norm(st.logdensity_grad[0])), 'accepted': bool(ok[0]),
                     'finite_g': bool(np.isfinite(st.logdensity_grad).all())})
    return rows


def main():
    fx = dict(np.load(sys.argv[1], allow_pickle=False))
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    dev = json.loads(str(fx['device_rows']))[:n]
    r32 = replay(fx, np.float32, n)
    r64 = replay(fx, np.float64, n)
    keys = ('eps_in', 'logdensity', 'kinetic_change', 'energy_change', 'eps_out', 'x_average_out', 'gnorm_out')
    print(f"chain {int(fx['chain'])}, first replayed step {int(fx['step0'])}")
    for i in range(n):
        print(f"step {r32[i]['step']}")
        for name, r in (('device ', dev[i] if i < len(dev) else None), ('orc f32', r32[i]), ('orc f64', r64[i])):
            if r is None:
                continue
            print('   ' + name + ' ' + '  '.join(f"{k}={r[k]:.6g}" for k in keys))
    out = {'chain': int(fx['chain']), 'step0': int(fx['step0']), 'device': dev, 'oracle_f32': r32, 'oracle_f64': r64}
    if len(sys.argv) > 3:
        Path(sys.argv[3]).write_text(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
