"""dev: specialised (compile-time flags) vs general update kernels on the same inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import mclmc_oracle as O
from mile_amd import ModelSpec
from mile_amd.engine import Engine
hs = (8, 8, 2) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1].split(','))
ospec = O.ModelSpec(5, hs)
E, N, T = 3, 60, 3
prob = O.synthetic_problem(ospec, N, E, seed=3)
d = ospec.n_params
rng = np.random.default_rng(0)
z0 = torch.from_numpy(rng.standard_normal((E, d)).astype(np.float32))
noise = torch.from_numpy(rng.standard_normal((T, 2, E, d)).astype(np.float32))
out = {}
for mode in ('general', 'kinds'):
    if mode == 'general':
        os.environ['MILE_NO_UPD_KIND'] = '1'
    else:
        os.environ.pop('MILE_NO_UPD_KIND', None)
    os.environ['MILE_NO_FUSE'] = '1'
    eng = Engine(ModelSpec(5, hs), torch.from_numpy(prob['X']), torch.from_numpy(prob['y']), device='cuda:0')
    st = eng.init(torch.from_numpy(prob['theta0']), noise=z0)
    for n in (1, T):
        s, info, _ = eng.step(st, torch.from_numpy(prob['eps']), torch.from_numpy(prob['L']), n_steps=n, noise=noise[:n])
        out[mode, n] = (s, info)
for n in (1, T):
    a, b = out['general', n], out['kinds', n]
    for name in ('position', 'momentum', 'logdensity', 'logdensity_grad'):
        x, y = getattr(a[0], name), getattr(b[0], name)
        print(n, name, float((x - y).abs().max() / x.abs().max()))
    print(n, 'info', [float((x - y).abs().max()) for x, y in zip(a[1], b[1])])
a, b = out['general', 1][0].position.cpu().numpy(), out['kinds', 1][0].position.cpu().numpy()
bad = np.argwhere(np.abs(a - b) > 1e-6 * np.abs(a).max())
print('differing (particle, index):', bad[:40].tolist(), 'of d =', d, 'count', len(bad))
a, b = out['general', 1][0].momentum.cpu().numpy(), out['kinds', 1][0].momentum.cpu().numpy()
bad = np.argwhere(np.abs(a - b) > 1e-6 * np.abs(a).max())
print('momentum differing:', bad[:40].tolist(), 'count', len(bad))
