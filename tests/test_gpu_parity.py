"""HIP path vs the CPU oracle, through the C ABI, on the same seeded inputs (-m gpu)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


def _engine(oracle, ospec, prob, kernel='auto'):
    from mile_amd import ModelSpec
    from mile_amd.engine import Engine
    spec = ModelSpec(in_features=ospec.in_features, hidden_structure=ospec.hidden_structure,
                     activation=ospec.activation, task=ospec.task, prior=ospec.prior,
                     prior_loc=ospec.prior_loc, prior_scale=ospec.prior_scale)
    return Engine(spec, torch.from_numpy(prob['X']), torch.from_numpy(prob['y']), device='cuda:0',
                  grad_kernel=kernel)


def _relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


CASES = [
    # in_features, hidden, activation, task, prior, N, E, kernels
    (5, (64, 64, 64, 2), 'relu', 'regr', 'Normal', 1052, 16, ('generic', 'mfma_w64', 'mfma_w64_bf16x3')),
    (5, (64, 64, 64, 2), 'relu', 'regr', 'Normal', 100, 3, ('generic', 'mfma_w64', 'mfma_w64_bf16x3')),
    (5, (64, 64, 2), 'relu', 'regr', 'Normal', 333, 5, ('generic', 'mfma_w64', 'mfma_w64_bf16x3', 'gemm_f32', 'mfma_wide_bf16x3')),
    (8, (64, 2), 'relu', 'regr', 'Laplace', 64, 2, ('generic', 'mfma_w64')),
    (5, (16, 16, 2), 'relu', 'regr', 'Normal', 1052, 12, ('generic', 'mfma_narrow_f32', 'auto')),      # the reference's stock net
    (9, (24, 17, 2), 'tanh', 'regr', 'Normal', 257, 4, ('generic', 'gemm_f32', 'mfma_wide_bf16x3', 'mfma_narrow_f32')),
    (11, (32, 7), 'sigmoid', 'classification', 'Normal', 500, 6, ('generic', 'gemm_f32', 'mfma_wide_bf16x3', 'mfma_narrow_f32')),
    # round 3: k_grad_narrow (fp32 MFMA 16x16x4, what AUTO picks for hidden widths <= 32): the reference's real nets --
    # covertype [54 -> 32 -> 7] sigmoid, README / protein [16,16,16,2] -- every template form (1-3 hidden layers, one or two
    # 16-wide tiles per hidden layer, F <= 16 or <= 64), ragged widths, tanh, both heads, 16 classes, tiles vs workgroups
    (54, (32, 7), 'sigmoid', 'classification', 'Normal', 700, 5, ('generic', 'mfma_narrow_f32', 'auto')),
    (5, (16, 16, 16, 2), 'relu', 'regr', 'Normal', 1052, 12, ('generic', 'mfma_narrow_f32')),
    (9, (16, 16, 16, 2), 'relu', 'regr', 'Laplace', 333, 3, ('generic', 'mfma_narrow_f32')),
    (20, (32, 32, 32, 5), 'tanh', 'classification', 'Normal', 130, 2, ('generic', 'mfma_narrow_f32')),
    (64, (9, 30, 16), 'relu', 'classification', 'Normal', 97, 3, ('generic', 'mfma_narrow_f32')),
    (17, (5, 2), 'sigmoid', 'regr', 'Normal', 40, 1, ('generic', 'mfma_narrow_f32')),
    (3, (1, 1, 2), 'tanh', 'regr', 'Normal', 33, 2, ('generic', 'mfma_narrow_f32')),
    (5, (16, 16, 2), 'relu', 'regr', 'Normal', 2, 1, ('generic', 'mfma_narrow_f32')),
    (5, (16, 16, 2), 'relu', 'regr', 'Normal', 17, 300, ('generic', 'mfma_narrow_f32')),
    (16, (32, 16, 3), 'relu', 'classification', 'Normal', 4099, 2, ('generic', 'mfma_narrow_f32')),
    # the depth ablations among the reference's YAMLs: [16]*4 + [2], [8]*6 + [2], [16]*9 + [2]
    (5, (16, 16, 16, 16, 2), 'relu', 'regr', 'Normal', 400, 3, ('generic', 'mfma_narrow_f32', 'auto')),
    (8, (8, 8, 8, 8, 8, 8, 2), 'relu', 'regr', 'Normal', 203, 2, ('generic', 'mfma_narrow_f32')),
    (13, (16,) * 9 + (2,), 'tanh', 'regr', 'Normal', 150, 2, ('generic', 'mfma_narrow_f32')),
    (6, (12,) * 10 + (4,), 'sigmoid', 'classification', 'Normal', 90, 2, ('generic', 'mfma_narrow_f32')),
    (54, (40, 40, 7), 'relu', 'classification', 'Laplace', 130, 3, ('generic', 'gemm_f32', 'mfma_wide_bf16x3', 'mfma_narrow_f32', 'auto')),
    # hidden widths 33..64 with any activation / head: k_grad_narrow with its weights in LDS (VERDICT r2 missing #3: a 64-wide
    # softmax or tanh net used to get the VALU kernel)
    (5, (64, 64, 7), 'tanh', 'classification', 'Normal', 333, 3, ('generic', 'mfma_narrow_f32', 'auto')),
    (20, (48, 2), 'sigmoid', 'regr', 'Normal', 257, 2, ('generic', 'mfma_narrow_f32')),
    (54, (33, 64, 50, 5), 'relu', 'classification', 'Normal', 200, 2, ('generic', 'mfma_narrow_f32')),
    (3, (64, 16), 'relu', 'classification', 'Normal', 100, 1, ('generic', 'mfma_narrow_f32')),
    (9, (64, 64, 64, 2), 'sigmoid', 'regr', 'Normal', 1052, 4, ('generic', 'mfma_narrow_f32', 'auto')),
    # wide nets: the layer-wise paths -- hand-written MFMA GEMMs (what AUTO picks there) and rocBLAS (the cross-check) --
    # B3- and B4-shaped, and shapes that leave ragged 128 x 128 x 64 tiles in every dimension
    (9, (128, 128, 128, 2), 'relu', 'regr', 'Normal', 700, 6, ('gemm_f32', 'auto')),
    (54, (256, 256, 256, 256, 7), 'relu', 'classification', 'Normal', 300, 3, ('gemm_f32', 'generic', 'auto')),
    (54, (256, 256, 256, 7), 'relu', 'classification', 'Normal', 384, 2, ('mfma_wide_bf16x3',)),   # whole 128-tiles: k_mm3's predicate-free form
    (13, (200, 136, 3), 'tanh', 'classification', 'Normal', 1000, 2, ('generic', 'mfma_wide_bf16x3')),
    (5, (96, 2), 'sigmoid', 'regr', 'Laplace', 129, 1, ('generic', 'mfma_wide_bf16x3')),
    # edge cases: one particle, fewer rows than one MFMA block, ragged last block, one hidden layer,
    # more workgroups than row blocks, F at the padding boundary
    (5, (64, 64, 64, 2), 'relu', 'regr', 'Normal', 2, 1, ('generic', 'mfma_w64', 'mfma_w64_bf16x3')),
    (5, (64, 64, 64, 2), 'relu', 'regr', 'Normal', 31, 2, ('generic', 'mfma_w64', 'mfma_w64_bf16x3')),
    (5, (64, 64, 64, 2), 'relu', 'regr', 'Normal', 33, 300, ('generic', 'mfma_w64', 'mfma_w64_bf16x3')),
    (8, (64, 64, 2), 'relu', 'regr', 'Normal', 1057, 7, ('generic', 'mfma_w64', 'mfma_w64_bf16x3')),
    (16, (64, 2), 'relu', 'regr', 'Normal', 129, 2, ('generic', 'mfma_w64')),
    (9, (64, 64, 2), 'relu', 'regr', 'Normal', 64, 5, ('generic', 'mfma_w64', 'mfma_w64_bf16x3')),
    (3, (2,), 'relu', 'regr', 'Normal', 40, 3, ('generic',)),
    (6, (5,), 'relu', 'classification', 'Normal', 40, 3, ('generic',)),
]


@pytest.mark.parametrize('F,hs,act,task,prior,N,E,kernels', CASES)
def test_logpost_grad_matches_oracle(oracle, F, hs, act, task, prior, N, E, kernels):
    ospec = oracle.ModelSpec(F, hs, activation=act, task=task, prior=prior, prior_scale=0.7 if prior == 'Laplace' else 1.0)
    prob = oracle.synthetic_problem(ospec, N, E, seed=3)
    if act == 'relu' and len(hs) > 1:
        # ReLU'(z) is discontinuous at 0: a pre-activation within fp32 rounding of the kink (|z| < 3e-7 of the layer's
        # largest) may legitimately fall on either side in fp32 and in the fp64 oracle -- seed 3 has one in the
        # (F=8, N=1057) case, where both HIP kernels agree bit for bit with each other but differ from fp64 by 1.3e-4.
        # Such rows are left out of the comparison (for the oracle and for the device alike), not reseeded around.
        _, zs, _ = oracle.mlp_forward(ospec, prob['theta0'].astype(np.float64), prob['X'], keep=True)
        near = np.zeros(N, dtype=bool)
        for z in zs[:-1]:
            near |= (np.abs(z) < 3e-7 * np.abs(z).max()).any(axis=(0, 2))
        assert near.sum() <= 17, near.sum()            # what the worst case in CASES needs (measured: 5, (F=8, N=1057))
        if (F, N) == (8, 1057):
            assert near.any()                        # the case this mask exists for
        if near.any() and near.sum() < N:
            prob = dict(prob, X=np.ascontiguousarray(prob['X'][~near]), y=np.ascontiguousarray(prob['y'][~near]))
    lp_ref, g_ref = oracle.logpost_and_grad(ospec, prob['theta0'].astype(np.float64), prob['X'], prob['y'])
    for k in kernels:
        eng = _engine(oracle, ospec, prob, k)
        wide = max(hs[:-1], default=0) >= 96
        assert eng.grad_kernel == (k if k != 'auto' else ('mfma_wide_bf16x3' if wide else 'mfma_narrow_f32'))
        lp, g = eng.logpost_grad(torch.from_numpy(prob['theta0']))
        torch.cuda.synchronize()
        # fp32 accumulation over N rows vs fp64: tolerance 2e-5 relative to the largest entry
        assert _relerr(lp.cpu().numpy(), lp_ref) < 2e-5, k
        assert _relerr(g.cpu().numpy(), g_ref) < 2e-5, k


def test_narrow_kernel_on_random_specs(oracle):
    """k_grad_narrow over 30 random FCN specs inside its support (1-3 hidden layers of width 1..32 with F <= 64, or 4-10 of
    width 1..16 with F <= 16; 1..16 classes or the (mu, log sigma) head; relu / tanh / sigmoid; N from below one tile to a
    few hundred rows; 1..7 particles) against the fp64 oracle at the tolerance of the parity table, and against the generic
    kernel; AUTO must pick it for every one of them."""
    rng = np.random.default_rng(2024)
    worst = 0.0
    for it in range(30):
        deep = it % 3 == 2
        nh = int(rng.integers(4, 11)) if deep else int(rng.integers(1, 4))
        wmax = 16 if deep else 32
        F = int(rng.integers(1, 17 if deep else 65))
        hidden = tuple(int(rng.integers(1, wmax + 1)) for _ in range(nh))
        task = 'regr' if rng.random() < 0.5 else 'classification'
        K = 2 if task == 'regr' else int(rng.integers(2, 17))
        act = ('relu', 'tanh', 'sigmoid')[int(rng.integers(0, 3))]
        N, E = int(rng.integers(1, 400)), int(rng.integers(1, 8))
        ospec = oracle.ModelSpec(F, hidden + (K,), activation=act, task=task)
        prob = oracle.synthetic_problem(ospec, N, E, seed=100 + it)
        if act == 'relu':
            _, zs, _ = oracle.mlp_forward(ospec, prob['theta0'].astype(np.float64), prob['X'], keep=True)
            near = np.zeros(N, dtype=bool)
            for z in zs[:-1]:
                near |= (np.abs(z) < 3e-7 * max(np.abs(z).max(), 1e-30)).any(axis=(0, 2))
            if near.any() and near.sum() < N:
                prob = dict(prob, X=np.ascontiguousarray(prob['X'][~near]), y=np.ascontiguousarray(prob['y'][~near]))
        lp_ref, g_ref = oracle.logpost_and_grad(ospec, prob['theta0'].astype(np.float64), prob['X'], prob['y'])
        eng = _engine(oracle, ospec, prob, 'auto')
        assert eng.grad_kernel == 'mfma_narrow_f32', (F, hidden, K)
        lp, g = eng.logpost_grad(torch.from_numpy(prob['theta0']))
        lpg, gg = _engine(oracle, ospec, prob, 'generic').logpost_grad(torch.from_numpy(prob['theta0']))
        torch.cuda.synchronize()
        tag = (it, F, hidden, K, act, task, N, E)
        assert _relerr(lp.cpu().numpy(), lp_ref) < 2e-5, tag
        assert _relerr(g.cpu().numpy(), g_ref) < 2e-5, tag
        assert _relerr(g.cpu().numpy(), gg.cpu().numpy()) < 2e-5, tag
        worst = max(worst, _relerr(g.cpu().numpy(), g_ref))
    print('narrow kernel, 30 random specs: worst gradient error vs fp64 %.2e of max |g|' % worst)


def test_wide_path_is_not_poisoned_by_a_previous_non_finite_call(oracle):
    """launch_grad_wide shares two dZ buffers between layers of different padded widths (hidden widths that are
    not multiples of 8 and differ: 100 / 50 / 20 -> leading dimensions 104 / 56 / 24).  A layer's padding columns are not
    written by the dH epilogue, k_mm3 reads operands in 4- / 8-element granules, and a NaN a diverged call left in what is
    now padding would turn into NaN x 0 = NaN.  Two consecutive gradients on one engine, the first at a non-finite theta:
    the second must equal a fresh engine's (and the fp64 oracle's)."""
    ospec = oracle.ModelSpec(13, (100, 50, 20, 3), activation='tanh', task='classification')
    prob = oracle.synthetic_problem(ospec, 300, 3, seed=5)
    th = torch.from_numpy(prob['theta0'])
    bad = th.clone()
    bad[:] = float('nan')
    eng = _engine(oracle, ospec, prob, 'mfma_wide_bf16x3')
    lp_bad, g_bad = eng.logpost_grad(bad)
    assert not torch.isfinite(g_bad).any()
    lp, g = eng.logpost_grad(th)
    torch.cuda.synchronize()
    lp_ref, g_ref = oracle.logpost_and_grad(ospec, prob['theta0'].astype(np.float64), prob['X'], prob['y'])
    assert torch.isfinite(g).all() and torch.isfinite(lp).all()
    assert _relerr(lp.cpu().numpy(), lp_ref) < 2e-5 and _relerr(g.cpu().numpy(), g_ref) < 2e-5
    fresh = _engine(oracle, ospec, prob, 'mfma_wide_bf16x3').logpost_grad(th)
    assert torch.equal(fresh[1], g) and torch.equal(fresh[0], lp)


@pytest.mark.parametrize('F,hs,N,E', [(5, (64, 64, 64, 2), 1052, 32), (12, (64, 64, 2), 777, 9), (5, (64, 64, 2), 64, 300)])
def test_split_bf16_w64_kernel_is_fp32_faithful(oracle, F, hs, N, E):
    """mfma_w64_bf16x3 (what AUTO picks for 2-3 hidden layers of width 64) forms each fp32 product exactly from
    three-term bf16 splits and drops only terms below 2^-23 of the leading one, so its error against the fp64 oracle
    must be fp32-rounding-sized: worst particle and ensemble mean no worse than 2x those of the fp32-MFMA kernel (+1e-7
    of the largest entry), and an order of magnitude inside the 2e-5 tolerance the other tests use."""
    ospec = oracle.ModelSpec(F, hs)
    prob = oracle.synthetic_problem(ospec, N, E, seed=21)
    lp_ref, g_ref = oracle.logpost_and_grad(ospec, prob['theta0'].astype(np.float64), prob['X'], prob['y'])
    scale = np.abs(g_ref).max(axis=1, keepdims=True)
    err = {}
    for k in ('mfma_w64', 'mfma_w64_bf16x3'):
        eng = _engine(oracle, ospec, prob, k)
        assert eng.grad_kernel == k
        lp, g = eng.logpost_grad(torch.from_numpy(prob['theta0']))
        torch.cuda.synchronize()
        err[k] = (np.abs(g.cpu().numpy().astype(np.float64) - g_ref) / scale).max(axis=1)     # per particle
        assert _relerr(lp.cpu().numpy(), lp_ref) < 1e-6, k
    assert err['mfma_w64_bf16x3'].max() <= 2.0 * err['mfma_w64'].max() + 1e-7, (err['mfma_w64_bf16x3'].max(), err['mfma_w64'].max())
    assert err['mfma_w64_bf16x3'].mean() <= 2.0 * err['mfma_w64'].mean() + 1e-8, (err['mfma_w64_bf16x3'].mean(), err['mfma_w64'].mean())
    assert err['mfma_w64_bf16x3'].max() < 2e-6
    assert _engine(oracle, ospec, prob, 'auto').grad_kernel == 'mfma_w64_bf16x3'


BF16_CASES = [
    # in_features, hidden, N, E   (ReLU regression, width 128: what k_grad_w128b supports)
    (9, (128, 128, 128, 2), 1000, 5),
    (9, (128, 128, 128, 2), 4096, 300),     # more particles than CUs, several row ranges
    (5, (128, 128, 2), 33, 3),              # ragged second tile
    (16, (128, 2), 257, 2),                 # F at the padding boundary, one hidden layer
    (3, (128, 128, 128, 2), 2, 1),          # fewer rows than one tile
    # round 3 (the next tile pair's first layer runs inside the previous pair's last backward phase, the first-layer weight
    # gradient one pair late): exactly one pair, one pair + padding, few particles (many row ranges per particle, some of
    # one pair or none), two and one hidden layers
    (9, (128, 128, 128, 2), 64, 1),
    (9, (128, 128, 128, 2), 65, 2),
    (9, (128, 128, 128, 2), 700, 1),
    (7, (128, 128, 2), 130, 2),
    (4, (128, 2), 64, 1),
]


@pytest.mark.parametrize('F,hs,N,E', BF16_CASES)
def test_bf16_w128_grad_matches_oracle(oracle, F, hs, N, E):
    """bf16-operand MFMA kernel (config B3's precision) vs the oracle.  Operands are rounded to bf16 (8-bit
    significand) where they enter a matrix product, accumulation is fp32.  Two checks: (1) against the
    oracle's restatement of that same recipe in fp64 (logpost_and_grad_bf16): what is left is fp32
    accumulation order (typically 1e-5) plus the rare activation that lands on the other side of a bf16
    rounding boundary or of the ReLU kink (one such flip moves a leaf by ~2e-3)
    -> 5e-3 of each parameter leaf's gradient norm, 1e-4 on the log-posterior; (2) against the
    full-precision fp64 oracle, the cost of the bf16 operands themselves -> 5e-2 of the whole gradient's
    norm, 5e-3 on the log-posterior."""
    ospec = oracle.ModelSpec(F, hs)
    prob = oracle.synthetic_problem(ospec, N, E, seed=3)
    lp_ref, g_ref = oracle.logpost_and_grad_bf16(ospec, prob['theta0'].astype(np.float64), prob['X'], prob['y'])
    lp_full, g_full = oracle.logpost_and_grad(ospec, prob['theta0'].astype(np.float64), prob['X'], prob['y'])
    eng = _engine(oracle, ospec, prob, 'mfma_w128_bf16')
    assert eng.grad_kernel == 'mfma_w128_bf16'
    lp, g = eng.logpost_grad(torch.from_numpy(prob['theta0']))
    torch.cuda.synchronize()
    lp, g = lp.cpu().numpy(), g.cpu().numpy().astype(np.float64)
    assert np.isfinite(g).all()
    assert _relerr(lp, lp_ref) < 1e-4
    assert _relerr(lp, lp_full) < 5e-3
    off = 0
    fin = F
    for li, wd in enumerate(hs):               # ravel order: bias, kernel per layer
        for name, n in (('bias', wd), ('kernel', fin * wd)):
            a, b = g[:, off:off + n], g_ref[:, off:off + n]
            err = np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-30)
            assert err.max() < 5e-3, (li, name, err.max())
            off += n
        fin = wd
    assert off == ospec.n_params
    tot = np.linalg.norm(g - g_full, axis=1) / np.linalg.norm(g_full, axis=1)
    assert tot.max() < 5e-2, tot.max()


@pytest.mark.parametrize('kernel', ['gemm_f32', 'mfma_wide_bf16x3'])
def test_gemm_path_row_chunks_accumulate(oracle, monkeypatch, kernel):
    """The layer-wise paths walk the data in row chunks when the activation workspace would not hold all of
    it; force 3 ragged chunks and compare with the oracle."""
    monkeypatch.setenv('MILE_GEMM_ROWS', '100')
    ospec = oracle.ModelSpec(7, (96, 50, 2), activation='tanh')
    prob = oracle.synthetic_problem(ospec, 257, 4, seed=8)
    lp_ref, g_ref = oracle.logpost_and_grad(ospec, prob['theta0'].astype(np.float64), prob['X'], prob['y'])
    eng = _engine(oracle, ospec, prob, kernel)
    lp, g = eng.logpost_grad(torch.from_numpy(prob['theta0']))
    assert _relerr(lp.cpu().numpy(), lp_ref) < 2e-5
    assert _relerr(g.cpu().numpy(), g_ref) < 2e-5


def test_wide_mfma_path_is_fp32_faithful_and_bf16_form_is_close(oracle):
    """mfma_wide_bf16x3 forms every product from exact three-term bf16 splits (six MFMA products, fp32 accumulation): its
    error against the fp64 oracle must be fp32-rounding-sized -- no worse than 2x the rocBLAS SGEMM path's (+1e-7 of the
    largest entry) and an order of magnitude inside the 2e-5 tolerance.  mfma_wide_bf16 (operands rounded to bf16, one
    product) is the explicit reduced-precision form: a few % of the gradient norm, as for mfma_w128_bf16."""
    ospec = oracle.ModelSpec(54, (256, 256, 7), task='classification')
    prob = oracle.synthetic_problem(ospec, 1500, 4, seed=5)
    lp_ref, g_ref = oracle.logpost_and_grad(ospec, prob['theta0'].astype(np.float64), prob['X'], prob['y'])
    scale = np.abs(g_ref).max(axis=1, keepdims=True)
    err = {}
    for k in ('gemm_f32', 'mfma_wide_bf16x3', 'mfma_wide_bf16'):
        eng = _engine(oracle, ospec, prob, k)
        assert eng.grad_kernel == k
        lp, g = eng.logpost_grad(torch.from_numpy(prob['theta0']))
        torch.cuda.synchronize()
        g = g.cpu().numpy().astype(np.float64)
        err[k] = (np.abs(g - g_ref) / scale).max()
        if k == 'mfma_wide_bf16':
            rel = np.linalg.norm(g - g_ref, axis=1) / np.linalg.norm(g_ref, axis=1)
            assert rel.max() < 5e-2 and _relerr(lp.cpu().numpy(), lp_ref) < 5e-3, (rel.max(),)
        else:
            assert _relerr(lp.cpu().numpy(), lp_ref) < 2e-6, k
    assert err['mfma_wide_bf16x3'] <= 2.0 * err['gemm_f32'] + 1e-7, err
    assert err['mfma_wide_bf16x3'] < 2e-6, err
    assert err['mfma_wide_bf16'] > 10 * err['mfma_wide_bf16x3']          # the two forms really differ


def test_philox_noise_bits_match_oracle(oracle):
    ospec = oracle.ModelSpec(5, (16, 16, 2))
    prob = oracle.synthetic_problem(ospec, 64, 4)
    eng = _engine(oracle, ospec, prob)
    ids = np.array([7, 0, 123456, 2**31 - 1], dtype=np.int32)
    for step, stage in ((0, 0), (5, 1), (2**32 - 1, 2)):
        z = eng.debug_noise(seed=0xDEADBEEFCAFE, E=4, step=step, stage=stage, particle_ids=ids).cpu().numpy()
        ref = oracle.philox_normal(0xDEADBEEFCAFE, ids, step, stage, eng.d, dtype=np.float32)
        # integer Philox words are bit-exact; the fp32 Box-Muller differs by libm rounding only
        assert np.abs(z - ref).max() < 2e-5
        assert abs(z.mean()) < 0.1 and abs(z.std() - 1) < 0.1


@pytest.mark.parametrize('kernel', ['generic', 'mfma_w64', 'mfma_w64_bf16x3'])
@pytest.mark.parametrize('refresh', ['O-step-O', 'step-O'])
def test_steps_match_oracle_explicit_noise(oracle, kernel, refresh):
    ospec = oracle.ModelSpec(5, (64, 64, 64, 2))
    N, E, T = 300, 6, 10
    prob = oracle.synthetic_problem(ospec, N, E, seed=11)
    rng = np.random.default_rng(5)
    d = ospec.n_params
    z0 = rng.standard_normal((E, d)).astype(np.float32)
    noise = rng.standard_normal((T, 2, E, d)).astype(np.float32)
    f = lambda th: oracle.logpost_and_grad(ospec, th, prob['X'], prob['y'])
    st = oracle.mclmc_init(f, prob['theta0'].astype(np.float64), z0.astype(np.float64))
    infos = []
    kept = []
    for i in range(T):
        st, info = oracle.mclmc_step(f, st, prob['eps'].astype(np.float64), prob['L'].astype(np.float64),
                                     noise[i, 0].astype(np.float64), noise[i, 1].astype(np.float64), refresh=refresh)
        infos.append(info)
        if i % 3 == 0:
            kept.append(st.position.copy())

    eng = _engine(oracle, ospec, prob, kernel)
    s0 = eng.init(torch.from_numpy(prob['theta0']), noise=torch.from_numpy(z0))
    s1, info, samples = eng.step(s0, torch.from_numpy(prob['eps']), torch.from_numpy(prob['L']), n_steps=T,
                                 noise=torch.from_numpy(noise), n_thinning=3, refresh=refresh)
    torch.cuda.synchronize()
    # fp32 state vs fp64 oracle after 10 steps (20 gradients): 1e-4 relative
    assert _relerr(s1.position.cpu().numpy(), st.position) < 1e-4
    assert np.abs(s1.momentum.cpu().numpy() - st.momentum).max() < 1e-4 * np.abs(st.momentum).max() + 1e-6
    assert _relerr(s1.logdensity.cpu().numpy(), st.logdensity) < 1e-5
    assert _relerr(s1.logdensity_grad.cpu().numpy(), st.logdensity_grad) < 1e-3
    # unit momentum
    assert np.abs(np.linalg.norm(s1.momentum.cpu().numpy().astype(np.float64), axis=1) - 1).max() < 1e-5
    # info: energy_change is a difference of O(1e4) log-densities in fp32
    dE = np.stack([i.energy_change for i in infos])
    dK = np.stack([i.kinetic_change for i in infos])
    assert np.abs(info.kinetic_change.cpu().numpy() - dK).max() < 1e-3 + 1e-3 * np.abs(dK).max()
    assert np.abs(info.energy_change.cpu().numpy() - dE).max() < 5e-2
    # thinning: integer indexing exact, kept positions are those after steps 0,3,6,9
    assert samples.shape == (4, E, d)
    assert _relerr(samples.cpu().numpy(), np.stack(kept)) < 1e-4
    # the input state was not modified (functional step)
    assert torch.equal(s0.position.cpu(), torch.from_numpy(prob['theta0']))


@pytest.mark.parametrize('per_call', [1, 3])
def test_steps_match_oracle_beyond_the_register_cached_update(oracle, per_call):
    """d > 16384 floats takes the two-pass update kernel (k_update) instead of k_update_fast; one-step calls
    are what the host-driven tuner issues for such nets."""
    ospec = oracle.ModelSpec(9, (128, 128, 2))
    assert ospec.n_params > 16384
    N, E, T = 80, 3, 6
    prob = oracle.synthetic_problem(ospec, N, E, seed=21)
    rng = np.random.default_rng(6)
    d = ospec.n_params
    z0 = rng.standard_normal((E, d)).astype(np.float32)
    noise = rng.standard_normal((T, 2, E, d)).astype(np.float32)
    f = lambda th: oracle.logpost_and_grad(ospec, th, prob['X'], prob['y'])
    st = oracle.mclmc_init(f, prob['theta0'].astype(np.float64), z0.astype(np.float64))
    infos = []
    for i in range(T):
        st, info = oracle.mclmc_step(f, st, prob['eps'].astype(np.float64), prob['L'].astype(np.float64),
                                     noise[i, 0].astype(np.float64), noise[i, 1].astype(np.float64))
        infos.append(info.energy_change)
    eng = _engine(oracle, ospec, prob, 'gemm_f32')
    s = eng.init(torch.from_numpy(prob['theta0']), noise=torch.from_numpy(z0))
    got = []
    for i in range(0, T, per_call):
        s, info, _ = eng.step(s, torch.from_numpy(prob['eps']), torch.from_numpy(prob['L']), n_steps=per_call,
                              noise=torch.from_numpy(noise[i:i + per_call]), step_offset=i)
        got.append(info.energy_change.cpu().numpy())
    torch.cuda.synchronize()
    assert _relerr(s.position.cpu().numpy(), st.position) < 1e-4
    assert np.abs(s.momentum.cpu().numpy() - st.momentum).max() < 1e-4 * np.abs(st.momentum).max() + 1e-6
    assert _relerr(s.logdensity.cpu().numpy(), st.logdensity) < 1e-5
    assert _relerr(s.logdensity_grad.cpu().numpy(), st.logdensity_grad) < 1e-3
    assert np.abs(np.concatenate(got) - np.stack(infos)).max() < 5e-3


def test_layerwise_path_agrees_with_generic_kernel_on_random_shapes(oracle):
    """Two independent HIP implementations of the same gradient (single-launch VALU kernel vs rocBLAS SGEMMs +
    elementwise kernels) on a dozen random FCN shapes: widths, depth, activation, head, prior, ragged N, E = 1."""
    rng = np.random.default_rng(123)
    for trial in range(12):
        depth = int(rng.integers(1, 5))
        task = 'regr' if trial % 2 == 0 else 'classification'
        hs = tuple(int(v) for v in rng.integers(3, 90, depth)) + ((2,) if task == 'regr' else (int(rng.integers(2, 9)),))
        F = int(rng.integers(1, 40))
        act = ('relu', 'tanh', 'sigmoid')[trial % 3]
        prior = 'Laplace' if trial % 4 == 3 else 'Normal'
        N, E = int(rng.integers(1, 400)), int(rng.integers(1, 7))
        ospec = oracle.ModelSpec(F, hs, activation=act, task=task, prior=prior, prior_scale=0.9)
        prob = oracle.synthetic_problem(ospec, max(N, 2), E, seed=100 + trial)
        th = torch.from_numpy(prob['theta0'])
        a = _engine(oracle, ospec, prob, 'generic').logpost_grad(th)
        b = _engine(oracle, ospec, prob, 'gemm_f32').logpost_grad(th)
        assert _relerr(b[0].cpu().numpy(), a[0].cpu().numpy()) < 2e-5, (trial, F, hs, act, task)
        assert _relerr(b[1].cpu().numpy(), a[1].cpu().numpy()) < 5e-5, (trial, F, hs, act, task)


@pytest.mark.parametrize('kernel', ['generic', 'mfma_w64', 'mfma_w64_bf16x3'])
def test_row_window_is_the_minibatch_gradient(oracle, kernel):
    """mile_set_row_window (the minibatches of the warm-start stage, src/dataset/tabular.py:170-212 +
    src/training/trainer.py:706-760): the gradient over rows [begin, begin + count) equals the oracle's on those rows --
    windows at the start, in the middle (not 32-aligned), and ending at the last row; count = 0 restores the full set."""
    ospec = oracle.ModelSpec(5, (64, 64, 64, 2))
    N, E = 301, 5
    prob = oracle.synthetic_problem(ospec, N, E, seed=17)
    eng = _engine(oracle, ospec, prob, kernel)
    th = torch.from_numpy(prob['theta0'])
    full = eng.logpost_grad(th)
    for begin, count in ((0, 32), (45, 32), (64, 100), (N - 32, 32), (N - 7, 7)):
        eng.set_row_window(begin, count)
        lp, g = eng.logpost_grad(th)
        lp_ref, g_ref = oracle.logpost_and_grad(ospec, prob['theta0'].astype(np.float64), prob['X'][begin:begin + count],
                                                prob['y'][begin:begin + count])
        assert _relerr(lp.cpu().numpy(), lp_ref) < 2e-5, (kernel, begin, count)
        assert _relerr(g.cpu().numpy(), g_ref) < 2e-5, (kernel, begin, count)
    eng.set_row_window(0, 0)
    lp, g = eng.logpost_grad(th)
    assert torch.equal(lp, full[0]) and torch.equal(g, full[1])
    with pytest.raises(Exception, match='window'):
        eng.set_row_window(N - 3, 7)


def test_row_window_on_the_chunked_kernels(oracle):
    """The layer-wise wide-net kernels and the LeNet kernels walk the rows in chunks anyway: a window is a shorter walk (first
    call under a window, then the full set again); the width-128 bf16 kernel refuses a window."""
    ospec = oracle.ModelSpec(7, (96, 96, 3), task='classification')
    N, E = 203, 3
    prob = oracle.synthetic_problem(ospec, N, E, seed=18)
    th = torch.from_numpy(prob['theta0'])
    eng = _engine(oracle, ospec, prob, 'mfma_wide_bf16x3')
    for begin, count in ((40, 32), (0, 0), (N - 9, 9)):
        eng.set_row_window(begin, count)
        sl = slice(begin, begin + count) if count else slice(None)
        lp, g = eng.logpost_grad(th)
        lp_ref, g_ref = oracle.logpost_and_grad(ospec, prob['theta0'].astype(np.float64), prob['X'][sl], prob['y'][sl])
        assert _relerr(lp.cpu().numpy(), lp_ref) < 2e-5 and _relerr(g.cpu().numpy(), g_ref) < 2e-5, (begin, count)
    from oracle import lenet_oracle as LN
    from mile_amd import LeNetSpec
    from mile_amd.engine import Engine
    lspec = LN.LeNetSpec(3, 16, 16, 4)
    lprob = LN.synthetic_problem(lspec, 41, 2, seed=19)
    lth = torch.from_numpy(lprob['theta0'])
    for kernel, tol in (('lenet_f32', 5e-5), ('lenet_bf16', 3e-3)):
        le = Engine(LeNetSpec(3, 16, 16, 4), torch.from_numpy(lprob['X']), torch.from_numpy(lprob['y']), device='cuda:0', grad_kernel=kernel)
        f = LN.logpost_and_grad if kernel == 'lenet_f32' else LN.logpost_and_grad_bf16
        for begin, count in ((10, 16), (0, 0)):
            le.set_row_window(begin, count)
            sl = slice(begin, begin + count) if count else slice(None)
            lp, g = le.logpost_grad(lth)
            lp_ref, g_ref = f(lspec, lprob['theta0'].astype(np.float64), lprob['X'][sl], lprob['y'][sl])
            assert _relerr(lp.cpu().numpy(), lp_ref) < 1e-4 and _relerr(g.cpu().numpy(), g_ref) < tol, (kernel, begin, count)
    bspec = oracle.ModelSpec(9, (128, 128, 2))
    bprob = oracle.synthetic_problem(bspec, 128, 2, seed=20)
    be = _engine(oracle, bspec, bprob, 'mfma_w128_bf16')
    be.set_row_window(0, 64)
    with pytest.raises(Exception, match='window'):
        be.logpost_grad(torch.from_numpy(bprob['theta0']))
