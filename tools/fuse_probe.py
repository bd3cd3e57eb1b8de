"""dev: a few B2 steps with MILE_DEBUG=32 (phase timestamps of k_grad_w64) -- run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import mclmc_oracle as O
from mile_amd import ModelSpec
from mile_amd.engine import Engine
ospec, N, E = O.config_spec('B2')
prob = O.synthetic_problem(ospec, N, E, seed=0)
eng = Engine(ModelSpec(5, (64, 64, 64, 2)), torch.from_numpy(prob['X']), torch.from_numpy(prob['y']), device='cuda:0')
st = eng.init(torch.from_numpy(prob['theta0']), seed=1)
eps, L = torch.from_numpy(prob['eps']).cuda(), torch.from_numpy(prob['L']).cuda()
os.environ.pop('MILE_DEBUG', None)
st, _, _ = eng.step(st, eps, L, n_steps=20, seed=1, inplace=True, want_info=False)
torch.cuda.synchronize()
os.environ['MILE_DEBUG'] = '32'
st, _, _ = eng.step(st, eps, L, n_steps=int(sys.argv[1]) if len(sys.argv) > 1 else 3, seed=1, step_offset=20, inplace=True, want_info=False)
torch.cuda.synchronize()
