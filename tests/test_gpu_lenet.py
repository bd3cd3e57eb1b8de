"""LeNet target (BASELINE config 5) on the HIP path vs the NumPy oracle (-m gpu)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def LN():
    from oracle import lenet_oracle
    return lenet_oracle


def _engine(ospec, prob, kernel='auto'):
    from mile_amd import LeNetSpec
    from mile_amd.engine import Engine
    spec = LeNetSpec(ospec.channels, ospec.height, ospec.width, ospec.out_dim, activation=ospec.activation, task=ospec.task,
                     prior=ospec.prior, prior_loc=ospec.prior_loc, prior_scale=ospec.prior_scale)
    assert spec.n_params == ospec.n_params
    assert [(n, o, tuple(s)) for n, o, s in spec.leaves()] == [(n, o, tuple(s)) for n, o, s in ospec.leaves()]
    return Engine(spec, torch.from_numpy(prob['X']), torch.from_numpy(prob['y']), device='cuda:0', grad_kernel=kernel)


def _relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


CASES = [
    # C, H, W, out_dim, activation, task, prior, N, E
    (1, 28, 28, 10, 'relu', 'classification', 'Normal', 37, 3),      # MNIST-shaped
    (3, 32, 32, 10, 'relu', 'classification', 'Normal', 20, 4),      # CIFAR-shaped (config 5)
    (2, 13, 17, 2, 'tanh', 'regr', 'Laplace', 9, 2),                 # odd sizes: pooling crops a row and a column
    (1, 12, 12, 3, 'sigmoid', 'classification', 'Normal', 1, 1),     # smallest image, one row, one particle
]


@pytest.mark.parametrize('C,H,W,K,act,task,prior,N,E', CASES)
def test_lenet_logpost_grad_matches_oracle(LN, C, H, W, K, act, task, prior, N, E):
    ospec = LN.LeNetSpec(C, H, W, K, activation=act, task=task, prior=prior, prior_scale=0.7 if prior == 'Laplace' else 1.0)
    prob = LN.synthetic_problem(ospec, N, E, seed=3)
    lp_ref, g_ref = LN.logpost_and_grad(ospec, prob['theta0'].astype(np.float64), prob['X'], prob['y'])
    eng = _engine(ospec, prob)
    assert eng.grad_kernel == 'lenet_f32'
    lp, g = eng.logpost_grad(torch.from_numpy(prob['theta0']))
    torch.cuda.synchronize()
    # fp32 accumulation vs fp64: 2e-5 relative to the largest entry, per parameter leaf
    assert _relerr(lp.cpu().numpy(), lp_ref) < 2e-5
    g = g.cpu().numpy()
    for name, off, shape in ospec.leaves():
        n = int(np.prod(shape))
        assert _relerr(g[:, off:off + n], g_ref[:, off:off + n]) < 5e-5, name


@pytest.mark.parametrize('C,H,W,K,act,task,prior,N,E', CASES + [
    (4, 20, 24, 4, 'relu', 'classification', 'Normal', 19, 2),       # four input channels: a full slot
    (3, 18, 22, 5, 'relu', 'classification', 'Normal', 7, 2),        # conv2's input is 9 x 11: odd width in the pair form of the input gradient
])
def test_lenet_mfma_convolutions_match_the_bf16_recipe(LN, C, H, W, K, act, task, prior, N, E):
    """`lenet_bf16`: the five convolution products as implicit GEMMs on v_mfma_f32_16x16x32_bf16 with bf16-rounded operands,
    against the oracle's restatement of that recipe (`logpost_and_grad_bf16`, fp64 accumulation).  The two round fp32- vs
    fp64-computed activations, so a value on a rounding boundary may flip (2^-8 of ONE operand): 3e-3 per leaf, 1e-3 in norm.
    Against the full-precision oracle the distance is what bf16 operands cost (a few per cent at these tiny N), bounded loosely here."""
    ospec = LN.LeNetSpec(C, H, W, K, activation=act, task=task, prior=prior, prior_scale=0.7 if prior == 'Laplace' else 1.0)
    prob = LN.synthetic_problem(ospec, N, E, seed=3)
    th64 = prob['theta0'].astype(np.float64)
    lp_ref, g_ref = LN.logpost_and_grad_bf16(ospec, th64, prob['X'], prob['y'])
    lp_full, g_full = LN.logpost_and_grad(ospec, th64, prob['X'], prob['y'])
    eng = _engine(ospec, prob, 'lenet_bf16')
    assert eng.grad_kernel == 'lenet_bf16'
    lp, g = eng.logpost_grad(torch.from_numpy(prob['theta0']))
    torch.cuda.synchronize()
    g = g.cpu().numpy()
    assert _relerr(lp.cpu().numpy(), lp_ref) < 1e-4
    for name, off, shape in ospec.leaves():
        n = int(np.prod(shape))
        assert _relerr(g[:, off:off + n], g_ref[:, off:off + n]) < 3e-3, name
    nrm = np.linalg.norm(g.astype(np.float64) - g_ref, axis=1) / np.linalg.norm(g_ref, axis=1)
    assert nrm.max() < 1e-3, nrm
    full = np.linalg.norm(g.astype(np.float64) - g_full, axis=1) / np.linalg.norm(g_full, axis=1)
    assert full.max() < 1e-1, full
    # image chunks accumulate (forced to 4 images per chunk), and evaluation runs the same forward kernels
    import os
    os.environ['MILE_GEMM_ROWS'] = '4'
    try:
        lp2, g2 = _engine(ospec, prob, 'lenet_bf16').logpost_grad(torch.from_numpy(prob['theta0']))
    finally:
        del os.environ['MILE_GEMM_ROWS']
    assert _relerr(g2.cpu().numpy(), g) < 2e-5 and _relerr(lp2.cpu().numpy(), lp.cpu().numpy()) < 1e-5
    pw = eng.pointwise_loglik(torch.from_numpy(prob['theta0']), torch.from_numpy(prob['X']), torch.from_numpy(prob['y']))
    ll_rows = M_pointwise(LN, ospec, th64, prob)
    assert np.abs(pw.cpu().numpy() - ll_rows).max() < 2e-3 * max(1.0, np.abs(ll_rows).max())


@pytest.mark.parametrize('C,H,W', [(1, 12, 13), (2, 15, 12), (3, 21, 19), (4, 14, 30), (1, 33, 12), (3, 12, 37), (2, 26, 26)])
def test_lenet_mfma_geometry_sweep(LN, C, H, W):
    """Image sizes around the tile / pair / chunk boundaries of the MFMA convolution kernels (odd and even widths of both
    convolutions, one and several tiles per row, pooling that crops a row or a column), small N and E: against the bf16 recipe."""
    ospec = LN.LeNetSpec(C, H, W, 3)
    prob = LN.synthetic_problem(ospec, 5, 2, seed=11)
    lp_ref, g_ref = LN.logpost_and_grad_bf16(ospec, prob['theta0'].astype(np.float64), prob['X'], prob['y'])
    lp, g = _engine(ospec, prob, 'lenet_bf16').logpost_grad(torch.from_numpy(prob['theta0']))
    g = g.cpu().numpy()
    assert _relerr(lp.cpu().numpy(), lp_ref) < 1e-4
    for name, off, shape in ospec.leaves():
        n = int(np.prod(shape))
        assert _relerr(g[:, off:off + n], g_ref[:, off:off + n]) < 3e-3, (name, C, H, W)


def M_pointwise(LN, ospec, th64, prob):
    out = LN.forward(ospec, th64, prob['X'], q=LN.M.bf16_round)
    return LN.M.pointwise_loglik(ospec, out, prob['y'])[0]


@pytest.mark.parametrize('gemm_form', [False, True])
def test_lenet_image_chunks_accumulate(LN, monkeypatch, gemm_form):
    """Images are walked in chunks (forced to 4 here); both forms of the convolution half: the direct LDS-tile kernels
    and the im2col + strided-batched SGEMM form kept as fallback for images too large for the tiles."""
    monkeypatch.setenv('MILE_GEMM_ROWS', '4')
    if gemm_form:
        monkeypatch.setenv('MILE_LENET_GEMM', '1')
    ospec = LN.LeNetSpec(3, 16, 20, 5)
    prob = LN.synthetic_problem(ospec, 11, 3, seed=5)
    lp_ref, g_ref = LN.logpost_and_grad(ospec, prob['theta0'].astype(np.float64), prob['X'], prob['y'])
    lp, g = _engine(ospec, prob).logpost_grad(torch.from_numpy(prob['theta0']))
    assert _relerr(lp.cpu().numpy(), lp_ref) < 2e-5
    assert _relerr(g.cpu().numpy(), g_ref) < 5e-5


def test_lenet_steps_and_pointwise_loglik_match_oracle(LN, oracle):
    ospec = LN.LeNetSpec(1, 14, 14, 4)
    N, E, T = 25, 3, 4
    prob = LN.synthetic_problem(ospec, N, E, seed=9)
    rng = np.random.default_rng(4)
    d = ospec.n_params
    z0 = rng.standard_normal((E, d)).astype(np.float32)
    noise = rng.standard_normal((T, 2, E, d)).astype(np.float32)
    f = lambda th: LN.logpost_and_grad(ospec, th, prob['X'], prob['y'])
    st = oracle.mclmc_init(f, prob['theta0'].astype(np.float64), z0.astype(np.float64))
    for i in range(T):
        st, info = oracle.mclmc_step(f, st, prob['eps'].astype(np.float64), prob['L'].astype(np.float64),
                                     noise[i, 0].astype(np.float64), noise[i, 1].astype(np.float64))
    eng = _engine(ospec, prob)
    s = eng.init(torch.from_numpy(prob['theta0']), noise=torch.from_numpy(z0))
    s, info_g, _ = eng.step(s, torch.from_numpy(prob['eps']), torch.from_numpy(prob['L']), n_steps=T, noise=torch.from_numpy(noise))
    torch.cuda.synchronize()
    assert _relerr(s.position.cpu().numpy(), st.position) < 1e-4
    assert _relerr(s.logdensity.cpu().numpy(), st.logdensity) < 1e-5
    assert _relerr(s.logdensity_grad.cpu().numpy(), st.logdensity_grad) < 1e-3
    assert np.abs(info_g.energy_change[-1].cpu().numpy() - info.energy_change).max() < 5e-3
    # evaluation path: per-row log-likelihood of held-out images
    test = LN.synthetic_problem(ospec, 13, 1, seed=10)
    out = LN.forward(ospec, prob['theta0'].astype(np.float64), test['X'])
    ref, _ = oracle.pointwise_loglik_raw(ospec, out, test['y'])
    pw = eng.pointwise_loglik(torch.from_numpy(prob['theta0']), torch.from_numpy(test['X']), torch.from_numpy(test['y']))
    assert pw.shape == (E, 13)
    assert np.abs(pw.cpu().numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


def test_full_size_properties_config5_lenet(LN):
    """BASELINE config 5 at its full ensemble (CIFAR-shaped LeNet, E = 256 particles; 1 000 images keep the test in seconds) on the
    MFMA convolution path: agreement with the fp32 direct-convolution path at the bf16 recipe's cost, additivity over images,
    permutation equivariance over particles (bit-exact), determinism and unit momentum through MCLMC steps (d = 83 126 takes the
    segmented update `k_update_seg`: segments of 8192 elements on all CUs)."""
    ospec = LN.LeNetSpec(3, 32, 32, 10)
    N, E = 1000, 256
    prob = LN.synthetic_problem(ospec, N, E, seed=0)
    th = torch.from_numpy(prob['theta0'])
    bf = _engine(ospec, prob, 'lenet_bf16')
    lp1, g1 = bf.logpost_grad(th)
    f32 = _engine(ospec, prob, 'lenet_f32')
    lp2, g2 = f32.logpost_grad(th[:16])
    assert _relerr(lp1[:16].cpu().numpy(), lp2.cpu().numpy()) < 2e-3
    rel = ((g1[:16] - g2).norm(dim=1) / g2.norm(dim=1)).max().item()
    assert rel < 1e-1, rel     # what bf16 operands cost at a random initialisation (4.5 % here; B3: 2-5 %)
    # two particles against the oracle's restatement of the recipe
    lo, go = LN.logpost_and_grad_bf16(ospec, prob['theta0'][:2].astype(np.float64), prob['X'], prob['y'])
    assert _relerr(lp1[:2].cpu().numpy(), lo) < 1e-4
    assert ((g1[:2].cpu().double() - torch.from_numpy(go)).norm(dim=1) / torch.from_numpy(go).norm(dim=1)).max().item() < 2e-3
    # additivity over images (every image is rounded on its own; halves add up to fp32 summation order)
    h = 504
    half = lambda sl: _engine(ospec, dict(prob, X=prob['X'][sl], y=prob['y'][sl]), 'lenet_bf16').logpost_grad(th)
    a, b = half(slice(0, h)), half(slice(h, N))
    prior_g = -th.cuda()
    assert _relerr((a[1] + b[1] - prior_g).cpu().numpy(), g1.cpu().numpy()) < 5e-5
    # particles are independent: permuting theta's rows permutes the outputs bit for bit
    perm = torch.randperm(E, generator=torch.Generator().manual_seed(1))
    lp3, g3 = bf.logpost_grad(th[perm])
    assert torch.equal(lp3.cpu(), lp1.cpu()[perm]) and torch.equal(g3.cpu(), g1.cpu()[perm])
    # steps
    ids = torch.arange(E, dtype=torch.int32)
    eps, L = torch.full((E,), 1e-3), torch.full((E,), 1.0)
    s0 = bf.init(th, seed=5, particle_ids=ids)
    s1, info, kept = bf.step(s0, eps, L, n_steps=4, seed=5, n_thinning=2, particle_ids=ids)
    s1b, info_b, _ = bf.step(s0, eps, L, n_steps=4, seed=5, n_thinning=2, particle_ids=ids)
    assert torch.equal(s1.position, s1b.position) and torch.equal(info.energy_change, info_b.energy_change)
    assert (s1.momentum.double().norm(dim=1) - 1).abs().max().item() < 1e-5
    assert kept.shape == (2, E, ospec.n_params) and torch.isfinite(kept).all() and torch.isfinite(info.energy_change).all()


def test_train_cli_lenet_yaml(tmp_path):
    """The YAML surface with `model: LeNet` and image data (config 5 shape, scaled down)."""
    import subprocess, sys, yaml
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    cfg = yaml.safe_load((root / 'experiments' / 'mclmc_cifar_lenet_b5.yaml').read_text())
    cfg['saving_dir'] = str(tmp_path)
    cfg['experiment_name'] = 'lenet_small'
    cfg['data']['path'] = '300x3x16x16'
    cfg['training']['sampler'].update(warmup_steps=200, n_samples=20, n_chains=3, desired_energy_var_start=5e-4,
                                      desired_energy_var_end=1e-4)
    # warm-start training on minibatches of 32 images (row windows of the bf16 kernel, trainer.py:330-538)
    cfg['training']['warmstart'] = {'include': True, 'optimizer_config': {'name': 'adamw', 'parameters': {'learning_rate': 0.003}},
                                    'max_epochs': 3, 'batch_size': 32, 'patience': 2}
    (tmp_path / 'cfg.yaml').write_text(yaml.safe_dump(cfg))
    r = subprocess.run([sys.executable, str(root / 'train.py'), '-c', str(tmp_path / 'cfg.yaml'), '-d', '1'],
                       capture_output=True, text=True, cwd=root, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    exp = tmp_path / 'lenet_small'
    log = (exp / 'training.log').read_text()
    assert 'Warmstart Training completed' in log
    w = np.load(exp / 'warmstart' / 'params_1.npz')
    assert np.abs(w['core.conv1.bias']).max() > 0 and all(np.isfinite(w[k]).all() for k in w.files)    # trained from zero biases
    assert sorted(p.name for p in (exp / 'samples').iterdir() if p.is_dir()) == ['0', '1', '2']
    z = np.load(exp / 'samples' / '2' / 'sample_10.npz')
    assert z.files == ['core.conv1.bias', 'core.conv1.kernel', 'core.conv2.bias', 'core.conv2.kernel', 'core.fc1.bias',
                       'core.fc1.kernel', 'core.fc2.bias', 'core.fc2.kernel', 'core.fc3.bias', 'core.fc3.kernel']
    assert z['core.conv1.kernel'].shape == (5, 5, 3, 6) and z['core.conv2.kernel'].shape == (5, 5, 6, 16)
    assert z['core.fc1.kernel'].shape == (2 * 2 * 16, 120) and z['core.fc3.bias'].shape == (10,)
    assert all(np.isfinite(z[k]).all() for k in z.files)
    r = subprocess.run([sys.executable, str(root / 'evaluate.py'), '-e', str(exp), '--split', 'valid'], capture_output=True,
                       text=True, cwd=root, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    import json
    m = json.loads((exp / 'metrics.json').read_text())
    assert m['split'] == 'valid' and m['n_points'] == 30 and np.isfinite(m['lppd'])


def test_hip_reproduces_committed_golden_fixtures(LN, oracle):
    """The committed fixtures of tests/golden/make_golden_r01b.py: LeNet gradient (fp32 vs fp64 fixture) and the
    bf16-operand recipe of k_grad_w128b."""
    from pathlib import Path
    from mile_amd import ModelSpec
    from mile_amd.engine import Engine
    gold = Path(__file__).parent / 'golden'
    z = np.load(gold / 'lenet_2x12x14.npz')
    ls = LN.LeNetSpec(2, 12, 14, 3, activation='tanh')
    eng = _engine(ls, {'X': z['X'], 'y': z['y']})
    lp, g = eng.logpost_grad(torch.from_numpy(z['theta0']))
    assert _relerr(lp.cpu().numpy(), z['logp']) < 2e-5 and _relerr(g.cpu().numpy(), z['grad']) < 5e-5
    z = np.load(gold / 'bf16_recipe_128x2.npz')
    spec = ModelSpec(9, (128, 128, 2))
    eng = Engine(spec, torch.from_numpy(z['X']), torch.from_numpy(z['y']), device='cuda:0', grad_kernel='mfma_w128_bf16')
    lp, g = eng.logpost_grad(torch.from_numpy(z['theta0']))
    assert _relerr(lp.cpu().numpy(), z['logp']) < 1e-4
    err = np.linalg.norm(g.cpu().numpy().astype(np.float64) - z['grad'], axis=1) / z['grad_norm']
    assert err.max() < 5e-3, err
