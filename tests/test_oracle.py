"""CPU tests of the oracle: known answers, invariants and the committed golden vectors."""
import math
from pathlib import Path

import numpy as np
import pytest

from oracle import mclmc_oracle as O

GOLD = Path(__file__).parent / 'golden'


def test_philox_known_answer_vectors():
    # Random123 kat_vectors for philox4x32-10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, exp in kat:
        got = O.philox4x32(np.array(ctr, np.uint32), np.array(key, np.uint32))
        assert tuple(int(v) for v in got) == exp


def test_philox_normal_moments_and_sharding_independence():
    ids = np.arange(6)
    z = O.philox_normal(42, ids, 3, 1, 4001)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02
    # a particle's stream depends on its GLOBAL id only
    z2 = O.philox_normal(42, ids[3:], 3, 1, 4001)
    assert np.array_equal(z[3:], z2)


@pytest.mark.parametrize('act,task,prior', [('relu', 'regr', 'Normal'), ('tanh', 'regr', 'Laplace'),
                                            ('sigmoid', 'classification', 'Normal'), ('relu', 'classification', 'Laplace')])
def test_gradient_matches_torch_autograd_fp64(act, task, prior):
    torch = pytest.importorskip('torch')
    hs = (9, 6, 2) if task == 'regr' else (9, 6, 4)
    spec = O.ModelSpec(5, hs, activation=act, task=task, prior=prior, prior_loc=0.1, prior_scale=0.8)
    pr = O.synthetic_problem(spec, 37, 3, seed=4)
    th = pr['theta0'].astype(np.float64)
    lp, g = O.logpost_and_grad(spec, th, pr['X'], pr['y'])
    t = torch.tensor(th, requires_grad=True)
    X = torch.tensor(pr['X'], dtype=torch.float64)
    y = torch.tensor(pr['y'])
    actf = {'relu': torch.relu, 'tanh': torch.tanh, 'sigmoid': torch.sigmoid}[act]
    vals = []
    for e in range(3):
        h, off, fin = X, 0, 5
        for li, w in enumerate(hs):
            b = t[e, off:off + w]; off += w
            W = t[e, off:off + fin * w].reshape(fin, w); off += fin * w
            h = h @ W + b
            if li < len(hs) - 1:
                h = actf(h)
            fin = w
        if task == 'regr':
            ll = torch.distributions.Normal(h[:, 0], torch.exp(h[:, 1]).clamp(1e-6, 1e6)).log_prob(y.double()).sum()
        else:
            ll = torch.distributions.Categorical(logits=h).log_prob(y.long()).sum()
        loc, sc = torch.tensor(0.1, dtype=torch.float64), torch.tensor(0.8, dtype=torch.float64)
        pd = torch.distributions.Normal(loc, sc) if prior == 'Normal' else torch.distributions.Laplace(loc, sc)
        vals.append(ll + pd.log_prob(t[e]).sum())
    v = torch.stack(vals)
    v.sum().backward()
    assert np.abs(v.detach().numpy() - lp).max() < 1e-10
    assert np.abs(t.grad.numpy() - g).max() < 1e-10


def test_param_layout_is_ravel_pytree_order():
    spec = O.ModelSpec(3, (4,) * 11 + (2,))
    order = O.layer_order(12)
    assert order[:4] == [0, 1, 10, 11] and order[4] == 2          # 'layer10' < 'layer2'
    keys = O.flattened_keys(spec)
    assert keys[0] == 'fcn.layer0.bias' and keys[1] == 'fcn.layer0.kernel' and keys[4] == 'fcn.layer10.bias'
    ents = O.param_slices(spec)
    assert ents[0]['bias'] == (0, 4) and ents[0]['kernel'] == (4, 16)
    assert ents[10]['bias'][0] == ents[1]['kernel'][1]             # layer10 right after layer1
    assert sum(e['kernel'][1] - e['kernel'][0] + 4 - (2 if e['layer'] == 11 else 0) for e in ents) == spec.n_params


def _rand_unit(rng, E, d):
    u = rng.standard_normal((E, d))
    return u / np.linalg.norm(u, axis=1, keepdims=True)


def test_b_step_invariants_and_closed_form():
    rng = np.random.default_rng(0)
    E, d = 4, 50
    u, g = _rand_unit(rng, E, d), 30 * rng.standard_normal((E, d))
    eps = np.full(E, 0.3)
    un, v, dK = O.esh_momentum_update(u, g, eps, 0.7)
    assert np.abs(np.linalg.norm(un, axis=1) - 1).max() < 1e-14
    un0, _, dK0 = O.esh_momentum_update(u, g, np.full(E, 1e-14), 0.7)
    assert np.abs(un0 - u).max() < 1e-12 and np.abs(dK0).max() < 1e-10      # eps -> 0 is the identity
    # u orthogonal to e: u' = (e (1 - zeta^2) + 2 zeta u)/(1 + zeta^2), dK = (d-1)(delta - ln2 + ln(1 + zeta^2))
    e = g / np.linalg.norm(g, axis=1, keepdims=True)
    uo = u - (u * e).sum(1, keepdims=True) * e
    uo /= np.linalg.norm(uo, axis=1, keepdims=True)
    un, _, dK = O.esh_momentum_update(uo, g, eps, 0.7)
    delta = 0.3 * 0.7 * np.linalg.norm(g, axis=1) / (d - 1)
    zeta = np.exp(-delta)
    ref = (e * (1 - zeta**2)[:, None] + 2 * zeta[:, None] * uo) / (1 + zeta**2)[:, None]
    assert np.abs(un - ref).max() < 1e-13
    assert np.abs(dK - (d - 1) * (delta - math.log(2) + np.log(1 + zeta**2))).max() < 1e-10


def test_b_step_is_the_exact_flow_of_the_esh_momentum_equation():
    """A check of the restated B-step that does not lean on the recollection of blackjax's text.  For a frozen gradient g the
    ESH / MCLMC momentum obeys du/dt = (|g| / (d - 1)) (e - (u.e) u), e = g / |g| (Robnik et al., Microcanonical HMC; the
    equation blackjax integrates), whose solution is what `esh_dynamics_momentum_update_one_step` evaluates in closed form.  An
    exact flow map must (i) compose: B(t1) then B(t2) == B(t1 + t2), for the momentum AND the accumulated kinetic-energy change
    (a wrong coefficient anywhere in the formula breaks this); (ii) have the equation's right-hand side as its derivative at
    t = 0; (iii) change the kinetic energy at the rate of the work done along the velocity, d(dK)/dt = g.u -- the term that
    cancels d(log density)/dt = g.u of the position update, i.e. energy conservation to first order."""
    rng = np.random.default_rng(5)
    E, d = 5, 37
    u, g = _rand_unit(rng, E, d), 12.0 * rng.standard_normal((E, d))
    one = np.ones(E)
    t1, t2 = 0.23, 0.41
    u1, _, k1 = O.esh_momentum_update(u, g, t1 * one, 1.0)
    u12, _, k2 = O.esh_momentum_update(u1, g, t2 * one, 1.0)
    ud, _, kd = O.esh_momentum_update(u, g, (t1 + t2) * one, 1.0)
    assert np.abs(u12 - ud).max() < 1e-13 and np.abs(k1 + k2 - kd).max() < 1e-11
    # the coefficient multiplies the time: B(eps, c) == B(eps c, 1)
    uc, _, kc = O.esh_momentum_update(u, g, one * 0.9, 0.37)
    ue_, _, ke_ = O.esh_momentum_update(u, g, one * 0.9 * 0.37, 1.0)
    assert np.abs(uc - ue_).max() < 1e-15 and np.abs(kc - ke_).max() < 1e-12
    # derivative at 0 (central difference of the flow at +-h; the flow at -h is the inverse map)
    h = 1e-5
    up, _, kp = O.esh_momentum_update(u, g, h * one, 1.0)
    um, _, km = O.esh_momentum_update(u, g, -h * one, 1.0)
    gn = np.linalg.norm(g, axis=1, keepdims=True)
    e = g / gn
    rhs = gn / (d - 1) * (e - (u * e).sum(1, keepdims=True) * u)
    assert np.abs((up - um) / (2 * h) - rhs).max() < 1e-7 * np.abs(rhs).max()
    assert np.abs((kp - km) / (2 * h) - (g * u).sum(1)).max() < 1e-6 * np.abs((g * u).sum(1)).max()
    # and the O-step leaves the momentum on the unit sphere while nu -> sqrt(2 h / (L d)) for h << L (the Langevin limit)
    z = rng.standard_normal((E, d))
    uo = O.partial_refresh(u, z, 1e-6 * one, 3.0 * one)
    assert np.abs(np.linalg.norm(uo, axis=1) - 1).max() < 1e-14
    nu = math.sqrt(2e-6 / (3.0 * d))
    assert np.abs((uo - u) - nu * (z - (u * z).sum(1, keepdims=True) * u)).max() < 5e-3 * nu


def _gauss_target(th):
    return -0.5 * (th * th).sum(axis=1), -th


def test_mclachlan_is_time_reversible():
    rng = np.random.default_rng(1)
    E, d = 3, 20
    x, u = rng.standard_normal((E, d)), _rand_unit(rng, E, d)
    l, g = _gauss_target(x)
    st0 = O.State(x, u, l, g)
    eps = np.full(E, 0.2)
    st1, dK = O.mclachlan_step(_gauss_target, st0, eps)
    back, dKb = O.mclachlan_step(_gauss_target, O.State(st1.position, -st1.momentum, st1.logdensity, st1.logdensity_grad), eps)
    assert np.abs(back.position - x).max() < 1e-12
    assert np.abs(-back.momentum - u).max() < 1e-12
    assert np.abs(dK + dKb).max() < 1e-10


def test_partial_refresh_limits():
    rng = np.random.default_rng(2)
    u, z = _rand_unit(rng, 2, 30), rng.standard_normal((2, 30))
    out = O.partial_refresh(u, z, np.full(2, 0.01), np.full(2, 1e12))
    assert np.abs(out - u).max() < 1e-6                         # nu -> 0 as L -> inf
    out = O.partial_refresh(u, z, np.full(2, 0.5), np.full(2, 3.0))
    assert np.abs(np.linalg.norm(out, axis=1) - 1).max() < 1e-14


def test_energy_error_order_on_gaussian():
    """isokinetic McLachlan is 2nd order: the one-step energy error scales like eps^3."""
    rng = np.random.default_rng(3)
    E, d = 64, 100
    x, u = rng.standard_normal((E, d)), _rand_unit(rng, E, d)
    l, g = _gauss_target(x)
    errs = []
    for eps in (0.4, 0.2, 0.1):
        st, dK = O.mclachlan_step(_gauss_target, O.State(x, u, l, g), np.full(E, eps))
        errs.append(np.sqrt(np.mean((dK - st.logdensity + l) ** 2)))
    assert 5.0 < errs[0] / errs[1] < 12.0 and 5.0 < errs[1] / errs[2] < 12.0


def test_standard_gaussian_stationary_moments():
    rng = np.random.default_rng(4)
    E, d, T = 8, 50, 3000
    st = O.mclmc_init(_gauss_target, rng.standard_normal((E, d)), rng.standard_normal((E, d)))
    eps, L = np.full(E, 0.5), np.full(E, math.sqrt(d))
    acc = []
    for i in range(T):
        st, info = O.mclmc_step(_gauss_target, st, eps, L, rng.standard_normal((E, d)), rng.standard_normal((E, d)))
        if i >= 500:
            acc.append(st.position.copy())
    xs = np.concatenate(acc)
    assert abs(xs.mean()) < 0.05 and abs(xs.var() - 1.0) < 0.1


def test_refresh_switch_differs_only_in_placement():
    rng = np.random.default_rng(5)
    E, d = 2, 10
    st = O.mclmc_init(_gauss_target, rng.standard_normal((E, d)), rng.standard_normal((E, d)))
    z1, z2 = rng.standard_normal((E, d)), rng.standard_normal((E, d))
    a, _ = O.mclmc_step(_gauss_target, st, np.full(E, 0.1), np.full(E, 3.0), z1, z2, refresh='O-step-O')
    b, _ = O.mclmc_step(_gauss_target, st, np.full(E, 0.1), np.full(E, 3.0), z1, z2, refresh='step-O')
    det, _ = O.mclachlan_step(_gauss_target, st, np.full(E, 0.1))
    assert np.abs(b.position - det.position).max() == 0.0      # step-O does not touch positions
    assert np.abs(a.position - det.position).max() > 0.0


def test_integer_bookkeeping():
    assert O.kept_indices(25, 10).tolist() == [0, 10, 20] and O.kept_indices(25, 10).dtype == np.int32
    assert O.kept_indices(3, 1).tolist() == [0, 1, 2]
    assert O.kept_indices(10000, 10)[-1] == 9990 and len(O.kept_indices(10000, 10)) == 1000
    assert O.phase_steps(50000) == (40000, 5000, 5000) and O.phase_steps(57) == (45, 5, 5)
    plan = O.train_plan(12, 4)
    assert [p.tolist() for p in plan] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11]]
    with pytest.raises(ValueError):
        O.train_plan(12, 5)


def test_desired_energy_var_schedules():
    assert O.desired_energy_var(0, 101, 0.5, 0.1) == 0.5
    assert abs(O.desired_energy_var(101, 101, 0.5, 0.1) - 0.1) < 1e-12
    assert abs(O.desired_energy_var(500, 101, 0.5, 0.1) - 0.1) < 1e-12     # clamped progress
    v = O.desired_energy_var(25, 101, 5.0, 0.1)                              # start > 2: exponential
    assert abs(v - (5.0 * math.exp(-25 / 25.25) + 0.1 * (1 - math.exp(-25 / 25.25)))) < 1e-12


def test_ess_on_ar1_and_white_noise():
    rng = np.random.default_rng(6)
    S = 20000
    for phi in (0.0, 0.9):
        x = np.zeros((1, S, 4))
        e = rng.standard_normal((S, 4))
        for t in range(1, S):
            x[0, t] = phi * x[0, t - 1] + e[t]
        ess = O.effective_sample_size(x)
        assert np.abs(ess / (S * (1 - phi) / (1 + phi)) - 1).max() < 0.2


def test_lppd_single_sample_is_mean_logpdf():
    spec = O.ModelSpec(3, (4, 2))
    rng = np.random.default_rng(7)
    out = rng.standard_normal((1, 1, 9, 2))
    y = rng.standard_normal(9)
    pw = O.pointwise_lppd(spec, out, y)
    sig = np.exp(out[0, 0, :, 1])
    ref = (-0.5 * ((y - out[0, 0, :, 0]) / sig) ** 2 - np.log(sig) - 0.5 * math.log(2 * math.pi)).mean()
    assert abs(O.lppd(pw) - ref) < 1e-12


def test_tuner_runs_and_step_size_adapts_on_gaussian():
    rng = np.random.default_rng(8)
    E, d = 3, 30
    st = O.mclmc_init(_gauss_target, rng.standard_normal((E, d)), rng.standard_normal((E, d)))
    noise = lambda i: (rng.standard_normal((E, d)), rng.standard_normal((E, d)))
    res = O.tune_phase12(_gauss_target, st, noise, 300, 60, step_size_init=0.01, desired_energy_var_start=0.5,
                         desired_energy_var_end=0.1, trust_in_estimate=1.5, num_effective_samples=100)
    assert np.all(res.step_size > 0.05) and np.all(np.isfinite(res.L))
    assert np.abs(res.L / math.sqrt(d) - 1).max() < 0.5           # L = sqrt(sum Var) ~ sqrt(d) on N(0, I)
    st2, L3 = O.tune_phase3(_gauss_target, res.state, res.step_size, res.L, noise, 200)
    assert np.all(L3 > 0)


@pytest.mark.parametrize('name,spec,refresh', [
    ('regr_relu_8x8', O.ModelSpec(5, (8, 8, 2)), 'O-step-O'),
    ('regr_relu_8x8_stepO', O.ModelSpec(5, (8, 8, 2)), 'step-O'),
    ('class_tanh_6x4', O.ModelSpec(7, (6, 4), activation='tanh', task='classification', prior='Laplace', prior_scale=0.5), 'O-step-O'),
    ('regr_relu_64x3', O.ModelSpec(5, (64, 64, 64, 2)), 'O-step-O'),
])
def test_oracle_reproduces_golden_vectors(name, spec, refresh):
    z = np.load(GOLD / f'{name}.npz')
    f = lambda th: O.logpost_and_grad(spec, th, z['X'], z['y'])
    st = O.mclmc_init(f, z['theta0'].astype(np.float64), z['z0'].astype(np.float64))
    assert np.allclose(st.logdensity, z['logp0'], rtol=1e-12, atol=0) and np.allclose(st.logdensity_grad, z['grad0'], rtol=1e-10, atol=1e-12)
    T = z['noise'].shape[0]
    for i in range(T):
        st, info = O.mclmc_step(f, st, z['eps'].astype(np.float64), z['L'].astype(np.float64),
                                z['noise'][i, 0].astype(np.float64), z['noise'][i, 1].astype(np.float64), refresh=refresh)
        assert np.allclose(np.stack([info.logdensity, info.kinetic_change, info.energy_change], -1), z['info'][i], rtol=1e-9, atol=1e-9)
    assert np.allclose(st.position, z[f'x_{T}'], rtol=1e-10, atol=1e-12)
    assert np.allclose(st.momentum, z[f'u_{T}'], rtol=1e-9, atol=1e-12)


def test_misc_golden():
    z = np.load(GOLD / 'misc.npz')
    assert np.array_equal(O.philox_bits(0x1234ABCD5678EF, z['philox_ids'], 9, 1, 37), z['philox_bits'])   # bit-exact
    assert np.allclose(O.philox_normal(0x1234ABCD5678EF, z['philox_ids'], 9, 1, 37), z['philox_normal'], rtol=1e-13)
    assert np.allclose(O.effective_sample_size(z['ar1']), z['ar1_ess'], rtol=1e-10)
    assert abs(O.lppd(z['lppd_in']) - float(z['lppd_out'])) < 1e-13


def test_bf16_round_matches_torch_and_recipe_is_close(oracle):
    """bf16_round is round-to-nearest-even to 8 significand bits (checked against torch's conversion); the
    bf16-operand recipe stays within a few percent of the full-precision gradient."""
    torch = pytest.importorskip('torch')
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(4096) * 10.0 ** rng.integers(-20, 20, 4096),
                        [0.0, -0.0, 1.0, 1.00390625, 1.01171875, 3.3e38, 1e-40]])
    want = torch.tensor(x, dtype=torch.float32).to(torch.bfloat16).to(torch.float64).numpy()
    got = oracle.bf16_round(x)
    fin = np.isfinite(want)                  # values that round up to inf in bf16 are left to the caller
    np.testing.assert_array_equal(got[fin], want[fin])
    sp = oracle.ModelSpec(9, (128, 128, 2))
    pr = oracle.synthetic_problem(sp, 200, 3, seed=3)
    th = pr['theta0'].astype(np.float64)
    lp, g = oracle.logpost_and_grad(sp, th, pr['X'], pr['y'])
    lpb, gb = oracle.logpost_and_grad_bf16(sp, th, pr['X'], pr['y'])
    assert np.abs(lpb - lp).max() < 5e-3 * np.abs(lp).max()
    assert (np.linalg.norm(gb - g, axis=1) / np.linalg.norm(g, axis=1)).max() < 5e-2


@pytest.mark.parametrize('C,H,W,K,act,task', [(1, 28, 28, 10, 'relu', 'classification'), (3, 32, 32, 10, 'tanh', 'classification'),
                                               (2, 13, 17, 2, 'sigmoid', 'regr')])
def test_lenet_oracle_matches_torch_autograd(C, H, W, K, act, task):
    """The LeNet restatement (oracle/lenet_oracle.py, src/models/images/cnns.py:33-66) against an independent
    fp64 implementation: torch conv2d / avg_pool2d / autograd."""
    torch = pytest.importorskip('torch')
    from oracle import lenet_oracle as LN
    spec = LN.LeNetSpec(C, H, W, K, activation=act, task=task, prior='Laplace' if act == 'tanh' else 'Normal', prior_scale=0.8)
    prob = LN.synthetic_problem(spec, 7, 2, seed=1)
    th = prob['theta0'].astype(np.float64)
    lp, g = LN.logpost_and_grad(spec, th, prob['X'], prob['y'])
    F = torch.nn.functional
    a = {'relu': torch.relu, 'tanh': torch.tanh, 'sigmoid': torch.sigmoid}[act]
    X = torch.tensor(prob['X'], dtype=torch.float64)
    for e in range(2):
        t = torch.tensor(th[e], requires_grad=True)
        P = {n: t[o:o + int(np.prod(sh))].reshape(sh) for n, o, sh in spec.leaves()}
        x = F.conv2d(X, P['core.conv1.kernel'].permute(3, 2, 0, 1), P['core.conv1.bias'], padding=2)
        x = F.avg_pool2d(a(x), 2)
        x = F.conv2d(x, P['core.conv2.kernel'].permute(3, 2, 0, 1), P['core.conv2.bias'])
        x = F.avg_pool2d(a(x), 2).permute(0, 2, 3, 1).reshape(7, -1)
        x = a(x @ P['core.fc1.kernel'] + P['core.fc1.bias'])
        x = a(x @ P['core.fc2.kernel'] + P['core.fc2.bias'])
        out = x @ P['core.fc3.kernel'] + P['core.fc3.bias']
        if task == 'regr':
            sig = torch.exp(out[:, 1]).clamp(1e-6, 1e6)
            ll = torch.distributions.Normal(out[:, 0], sig).log_prob(torch.tensor(prob['y'], dtype=torch.float64)).sum()
        else:
            ll = -F.cross_entropy(out, torch.tensor(prob['y'], dtype=torch.long), reduction='sum')
        loc, sc = torch.zeros((), dtype=torch.float64), torch.tensor(0.8, dtype=torch.float64)
        pr = (torch.distributions.Normal(loc, sc) if spec.prior == 'Normal' else torch.distributions.Laplace(loc, sc)).log_prob(t).sum()
        tot = ll + pr
        tot.backward()
        assert abs(tot.item() - lp[e]) < 1e-9 * abs(lp[e])
        assert np.abs(t.grad.numpy() - g[e]).max() < 1e-9 * np.abs(g[e]).max()
    assert spec.n_params == {(1, 28, 28, 10): 61706, (3, 32, 32, 10): 83126}.get((C, H, W, K), spec.n_params)


def test_golden_bf16_recipe_and_lenet():
    """Committed outputs of the two later restatements (tests/golden/make_golden_r01b.py)."""
    from oracle import lenet_oracle as LN
    z = np.load(GOLD / 'bf16_recipe_128x2.npz')
    spec = O.ModelSpec(9, (128, 128, 2))
    lp, g = O.logpost_and_grad_bf16(spec, z['theta0'].astype(np.float64), z['X'], z['y'])
    assert np.allclose(lp, z['logp'], rtol=1e-12, atol=0)
    assert np.abs(g - z['grad']).max() <= 1e-6 * np.abs(z['grad']).max()        # grad stored as float32
    assert np.allclose(np.linalg.norm(g, axis=1), z['grad_norm'], rtol=1e-12)
    z = np.load(GOLD / 'lenet_2x12x14.npz')
    ls = LN.LeNetSpec(2, 12, 14, 3, activation='tanh')
    lp, g = LN.logpost_and_grad(ls, z['theta0'].astype(np.float64), z['X'], z['y'])
    assert np.allclose(lp, z['logp'], rtol=1e-12, atol=0) and np.allclose(g, z['grad'], rtol=1e-10, atol=1e-12)


def test_tuner_readjustment_keeps_phase1_L(monkeypatch):
    """src/training/warmup.py:389-403: with diagonal_preconditioning the extra tune2 // 3 steps run with the L of
    phase 1, max(sqrt(d), 15) -- `params` only had sqrt_diag_cov replaced -- and sqrt(d) is what is returned.
    d = 10 < 225, so the two differ."""
    d, E, t1, t2 = 10, 3, 12, 9
    rng = np.random.default_rng(0)
    x0 = rng.standard_normal((E, d))
    st = O.mclmc_init(_gauss_target, x0, rng.standard_normal((E, d)))
    seen = []
    real = O.mclmc_step

    def spy(f, state, eps, L, z1, z2, sdc=None, refresh='O-step-O'):
        seen.append((np.array(L, copy=True), None if sdc is None else np.array(sdc, copy=True)))
        return real(f, state, eps, L, z1, z2, sdc, refresh)
    monkeypatch.setattr(O, 'mclmc_step', spy)
    noise = lambda i: (np.random.default_rng(100 + i).standard_normal((E, d)), np.random.default_rng(500 + i).standard_normal((E, d)))
    res = O.tune_phase12(_gauss_target, st, noise, t1, t2, step_size_init=0.05, desired_energy_var_start=0.5,
                         desired_energy_var_end=0.1, trust_in_estimate=1.5, num_effective_samples=100,
                         diagonal_preconditioning=True)
    assert len(seen) == t1 + t2 + t2 // 3
    assert all(np.all(L == 15.0) for L, _ in seen)                       # every kernel step, re-adjustment included
    assert all(np.all(s == 1.0) for _, s in seen[:t1 + t2])
    assert all(np.allclose(s, res.sqrt_diag_cov) for _, s in seen[t1 + t2:]) and not np.allclose(res.sqrt_diag_cov, 1.0)
    assert np.allclose(res.L, math.sqrt(d))


def test_dead_chain_fixture_float32_overflow_kills_the_step_size_float64_does_not():
    """profiles/r03/01_*: under the reference's stock tuner targets (desired_energy_var 0.5 -> 0.1) on the 3x64 airfoil net a
    chain is thrown, by one step at eps = 0.21, into a region with log-density -7e8 and |grad| 1.6e12.  tests/golden/
    dead_chain_b2.npz holds that chain's state one step earlier as the DEVICE had it (written by tools/r03/dead_chain_trace.py
    on an MI355X; X, y = the airfoil training split).  Replayed here through the oracle's tuner_step on the same Philox noise:
      * float32 (the arithmetic of the reference: JAX default precision): the catastrophic step is reproduced (log-density,
        kinetic / energy change within 2e-3, next step size within 1e-3 of the device's record); after it energy_change = dK - l' + l is a
        difference of 1e7-sized float32 numbers (noise of +-1e3), xi is inflated, eps shrinks each step and
        xi / eps^6 (warmup.py:311-313) overflows float32: x_average = inf, step_size = inf^(-1/6) = 0, for good;
      * float64: same catastrophic step, then eps settles near 4.5e-5, x_average stays finite, the chain climbs back.
    So the dead chains of that run are float32 arithmetic of the reference's own expressions, not a deviation of the kernels."""
    import json
    fx = dict(np.load(Path(__file__).parent / 'golden' / 'dead_chain_b2.npz', allow_pickle=False))
    dev = json.loads(str(fx['device_rows']))
    spec = O.ModelSpec(5, tuple(int(v) for v in fx['hidden']))
    d = spec.n_params
    tune1, tune2, total = (int(v) for v in fx['schedule'])
    v0, v1, trust, decay = (float(v) for v in fx['targets'])
    ids, seed, step0 = np.array([int(fx['chain'])]), int(fx['seed']), int(fx['step0'])
    out = {}
    for dt in (np.float32, np.float64):
        X, y = fx['X'].astype(dt), fx['y'].astype(dt)
        f = lambda th: O.logpost_and_grad(spec, th, X, y)
        st = O.State(fx['x'][None].astype(dt), fx['u'][None].astype(dt), np.array([fx['logp']], dt), fx['g'][None].astype(dt))
        ad = O.AdaptiveState(np.array([fx['tuner'][2]], dt), np.array([fx['tuner'][3]], dt), np.array([fx['tuner'][1]], dt),
                             np.zeros(1, dt), np.zeros((1, 2, d), dt))
        eps, L = np.array([fx['tuner'][0]], dt), np.array([fx['L']], dt)
        rows = []
        for i in range(14):
            k = step0 + i
            z1 = O.philox_normal(seed, ids, k, 0, d, dtype=np.float32).astype(dt)
            z2 = O.philox_normal(seed, ids, k, 1, d, dtype=np.float32).astype(dt)
            st, eps, ok, info = O.tuner_step(f, st, eps, L, np.ones((1, d), dt), z1, z2, ad, mask=1.0 if k < tune1 else 0.0,
                                             var=O.desired_energy_var(k, total, v0, v1), trust_in_estimate=trust, decay=dt(decay))
            assert ok[0] and np.isfinite(st.position).all()          # never rejected: every position stays finite
            rows.append((float(info.logdensity[0]), float(info.kinetic_change[0]), float(info.energy_change[0]), float(eps[0]),
                         float(ad.x_average[0])))
        out[dt] = rows
    r32, r64 = out[np.float32], out[np.float64]
    # the accepted step before, and the catastrophic step: oracle (both precisions) == the device's record
    for i in (0, 1):
        for r in (r32, r64):
            # 2e-3: the landing point has sigma_min = 3e-4 and the log-density goes with 1 / sigma^2
            assert abs(r[i][0] - dev[i]['logdensity']) <= 2e-3 * abs(dev[i]['logdensity'])
            assert abs(r[i][3] - dev[i]['eps_out']) <= 1e-3 * dev[i]['eps_out']
    assert dev[1]['logdensity'] < -1e8 and dev[1]['energy_change'] > 1e10 and dev[1]['finite_x'] and dev[1]['finite_g']
    assert abs(r32[1][2] - dev[1]['energy_change']) <= 2e-3 * dev[1]['energy_change']
    # float32: x_average overflows, the step size is 0 from then on (the device: 6 steps after the catastrophic one)
    dead32 = [i for i, r in enumerate(r32) if r[3] == 0.0]
    dead_dev = [i for i, r in enumerate(dev) if r['eps_out'] == 0.0]
    assert dead32 and dead_dev and abs(dead32[0] - dead_dev[0]) <= 4
    assert np.isinf(r32[dead32[0]][4]) and all(r[3] == 0.0 for r in r32[dead32[0]:])
    # float64: no overflow, eps stays positive and the log-density recovers
    assert all(np.isfinite(r[4]) and r[3] > 1e-6 for r in r64)
    assert r64[-1][0] > 0.5 * r64[3][0]
    # and what separates them: the float32 energy changes in the bad region are rounding noise of the 1e7-sized log-densities
    assert max(abs(r[2]) for r in r64[3:]) < 500 and max(abs(r[2]) for r in r32[3:8]) > 500
