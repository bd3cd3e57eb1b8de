"""Generate the committed golden fixtures FROM THE ORACLE (run: python tests/golden/make_golden.py).

The reference ships no golden vectors and cannot run offline (no JAX), so these vectors pin the
oracle against regressions and give the GPU tests a fixed target; they do NOT pin parity with
the reference ("parity unpinned", see oracle/mclmc_oracle.py header).
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import mclmc_oracle as O  # noqa: E402

OUT = Path(__file__).resolve().parent


def small_case(name, spec, N, E, T, seed, refresh='O-step-O'):
    prob = O.synthetic_problem(spec, N, E, seed=seed)
    rng = np.random.default_rng(seed + 100)
    d = spec.n_params
    z0 = rng.standard_normal((E, d)).astype(np.float32)
    noise = rng.standard_normal((T, 2, E, d)).astype(np.float32)
    f = lambda th: O.logpost_and_grad(spec, th, prob['X'], prob['y'])
    st = O.mclmc_init(f, prob['theta0'].astype(np.float64), z0.astype(np.float64))
    out = {'X': prob['X'], 'y': prob['y'], 'theta0': prob['theta0'], 'z0': z0, 'noise': noise,
           'eps': prob['eps'], 'L': prob['L'],
           'logp0': st.logdensity, 'grad0': st.logdensity_grad, 'u0': st.momentum}
    infos = []
    for i in range(T):
        st, info = O.mclmc_step(f, st, prob['eps'].astype(np.float64), prob['L'].astype(np.float64),
                                noise[i, 0].astype(np.float64), noise[i, 1].astype(np.float64), refresh=refresh)
        infos.append(np.stack([info.logdensity, info.kinetic_change, info.energy_change], axis=-1))
        if i + 1 in (1, T):
            out[f'x_{i + 1}'] = st.position.copy()
            out[f'u_{i + 1}'] = st.momentum.copy()
            out[f'logp_{i + 1}'] = st.logdensity.copy()
            out[f'grad_{i + 1}'] = st.logdensity_grad.copy()
    out['info'] = np.stack(infos)
    out['meta'] = np.array([spec.in_features, *spec.hidden_structure, N, E, T])
    np.savez_compressed(OUT / f'{name}.npz', **out)
    print(name, {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    small_case('regr_relu_8x8', O.ModelSpec(5, (8, 8, 2)), N=40, E=3, T=10, seed=1)
    small_case('regr_relu_8x8_stepO', O.ModelSpec(5, (8, 8, 2)), N=40, E=3, T=10, seed=1, refresh='step-O')
    small_case('class_tanh_6x4', O.ModelSpec(7, (6, 4), activation='tanh', task='classification', prior='Laplace',
                                             prior_scale=0.5), N=30, E=2, T=5, seed=2)
    small_case('regr_relu_64x3', O.ModelSpec(5, (64, 64, 64, 2)), N=70, E=2, T=2, seed=3)
    # counter RNG and diagnostics
    ids = np.array([0, 1, 77, 2**31 - 1])
    bits = O.philox_bits(0x1234ABCD5678EF, ids, 9, 1, 37)
    nrm = O.philox_normal(0x1234ABCD5678EF, ids, 9, 1, 37)
    rng = np.random.default_rng(5)
    x = np.zeros((1, 400, 3))
    e = rng.standard_normal((400, 3))
    for t in range(1, 400):
        x[0, t] = 0.8 * x[0, t - 1] + e[t]
    lp = rng.standard_normal((2, 5, 11))
    np.savez_compressed(OUT / 'misc.npz', philox_ids=ids, philox_bits=bits, philox_normal=nrm,
                        ar1=x, ar1_ess=O.effective_sample_size(x), lppd_in=lp, lppd_out=np.array(O.lppd(lp)))
    print('misc ok')
