import sys, numpy as np
sys.path.insert(0,'/root/repo')
from oracle import mclmc_oracle as O
def run(dt, pert, hs=(8,8,2), N=60, E=3, t1=45, t2=15, seed=21, init=0.01, v0=0.5, v1=0.1, diag=False):
    ospec = O.ModelSpec(5, hs); d = ospec.n_params
    prob = O.synthetic_problem(ospec, N, E, seed=seed)
    rng = np.random.default_rng(2)
    z0 = rng.standard_normal((E, d)).astype(np.float32)
    nz = rng.standard_normal((t1+t2+t2//3, 2, E, d)).astype(np.float32)
    X, y = prob['X'].astype(dt), prob['y'].astype(dt)
    f = lambda th: O.logpost_and_grad(ospec, th, X, y)
    th0 = prob['theta0'].astype(dt) * dt(1 + pert)
    st = O.mclmc_init(f, th0, z0.astype(dt))
    res = O.tune_phase12(f, st, lambda i: (nz[i,0].astype(dt), nz[i,1].astype(dt)), t1, t2, step_size_init=init,
        desired_energy_var_start=v0, desired_energy_var_end=v1, trust_in_estimate=1.5, num_effective_samples=100, record=True,
        diagonal_preconditioning=diag)
    return res
def rel(a,b): return float(np.abs(np.asarray(a,np.float64)-np.asarray(b,np.float64)).max()/np.abs(np.asarray(b,np.float64)).max())
for kw in [dict(), dict(init=0.05), dict(hs=(16,16,2),N=150), dict(hs=(16,16,2),N=150,init=0.03), dict(v0=5e-4,v1=1e-4,init=0.02,diag=True,t1=30,t2=24)]:
    a=run(np.float32,0,**kw); b=run(np.float32,1e-7,**kw); c=run(np.float64,0,**kw)
    print(kw, 'eps', a.step_size, 'f32 vs f32pert eps', rel(a.step_size,b.step_size), 'L', rel(a.L,b.L), 'x', rel(a.state.position,b.state.position),
          '| f32 vs f64 eps', rel(a.step_size,c.step_size), 'L', rel(a.L,c.L), 'x', rel(a.state.position, c.state.position), 'dE last', a.trace['energy_change'][-1])
print('---- prior-dominated')
for kw in [dict(N=8), dict(N=8, init=0.05), dict(N=4,hs=(16,16,2)), dict(N=8,t1=80,t2=30), dict(N=8, v0=5e-4,v1=1e-4,init=0.02,diag=True,t1=30,t2=24), dict(N=8, v0=0.05,v1=0.01,init=0.02,diag=True,t1=40,t2=30)]:
    a=run(np.float32,0,**kw); b=run(np.float32,1e-7,**kw); c=run(np.float64,0,**kw)
    print(kw, 'eps', a.step_size, 'f32 vs f32pert eps', rel(a.step_size,b.step_size), 'L', rel(a.L,b.L), 'x', rel(a.state.position,b.state.position),
          '| f32 vs f64 eps', rel(a.step_size,c.step_size), 'L', rel(a.L,c.L), 'x', rel(a.state.position, c.state.position), 'dE last', a.trace['energy_change'][-1], 'sdc', rel(a.sqrt_diag_cov, c.sqrt_diag_cov))
