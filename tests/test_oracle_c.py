"""The C/OpenMP restatement (oracle/cpu_mclmc.c, bench.py's cpu_baseline) against the NumPy oracle."""
import shutil

import numpy as np
import pytest

from oracle import mclmc_oracle as O

pytestmark = pytest.mark.skipif(shutil.which('gcc') is None, reason='needs gcc')


@pytest.mark.parametrize('F,hs,N,E,kw', [
    (5, (64, 64, 64, 2), 150, 5, {}), (5, (16, 16, 2), 77, 3, {}), (9, (24, 2), 64, 2, {}),
    # round 3: the heads / activations / priors of BASELINE config 4 and of the reference's classification YAMLs, and the
    # rows-over-threads form a bounded B4 sample takes (fewer particles than threads: E = 1 whenever the host has > 1 CPU)
    (54, (32, 32, 7), 300, 1, dict(task='classification')),
    (11, (32, 7), 130, 4, dict(activation='sigmoid', task='classification')),
    (7, (12, 9, 3), 90, 3, dict(activation='tanh', task='classification', prior='Laplace', prior_scale=0.5)),
    (5, (20, 2), 200, 1, dict(activation='tanh')),
])
def test_c_port_matches_numpy_oracle(F, hs, N, E, kw):
    from oracle.cpu_c import CpuPort
    spec = O.ModelSpec(F, hs, **kw)
    pr = O.synthetic_problem(spec, N, E, seed=2)
    port = CpuPort(spec, pr['X'], pr['y'])
    lp, g = port.logpost_grad(pr['theta0'])
    lp_ref, g_ref = O.logpost_and_grad(spec, pr['theta0'].astype(np.float64), pr['X'], pr['y'])
    assert np.abs(lp - lp_ref).max() / np.abs(lp_ref).max() < 2e-6
    assert np.abs(g - g_ref).max() / np.abs(g_ref).max() < 2e-5
    # three kernel steps with explicit noise
    rng = np.random.default_rng(1)
    d = spec.n_params
    T = 3
    z0 = rng.standard_normal((E, d)).astype(np.float32)
    noise = rng.standard_normal((T, 2, E, d)).astype(np.float32)
    f = lambda th: O.logpost_and_grad(spec, th, pr['X'], pr['y'])
    st = O.mclmc_init(f, pr['theta0'].astype(np.float64), z0.astype(np.float64))
    x, u = pr['theta0'].copy(), st.momentum.astype(np.float32)
    logp, grad = port.logpost_grad(x)
    infos = []
    for i in range(T):
        st, info = O.mclmc_step(f, st, pr['eps'].astype(np.float64), pr['L'].astype(np.float64),
                                noise[i, 0].astype(np.float64), noise[i, 1].astype(np.float64))
        infos.append(info.energy_change)
    got = port.steps(x, u, logp, grad, pr['eps'], pr['L'], noise, want_info=True)
    assert np.abs(x - st.position).max() / np.abs(st.position).max() < 1e-5
    assert np.abs(u - st.momentum).max() < 1e-4
    assert np.abs(logp - st.logdensity).max() / np.abs(st.logdensity).max() < 1e-5
    assert np.abs(got[..., 2] - np.stack(infos)).max() < 2e-2
    assert port.threads >= 1
