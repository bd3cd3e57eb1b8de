"""dev: warm-start training trace on the airfoil table (train / validation NLL per epoch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, logging
from mile_amd.config import Config
from mile_amd.trainer import BDETrainer
from mile_amd.warmstart import train_deep_ensemble, prior_value_and_grad
cfg = Config.from_file('experiments/mclmc_airfoil_b2.yaml').replace(logging=False)
tr = BDETrainer.__new__(BDETrainer); tr.config = cfg; tr.build_model(cfg)
ld = tr.loader
x, y = torch.from_numpy(np.ascontiguousarray(ld.train_x)), torch.from_numpy(np.ascontiguousarray(ld.train_y))
vx, vy = torch.from_numpy(np.ascontiguousarray(ld.valid_x)), torch.from_numpy(np.ascontiguousarray(ld.valid_y))
print('train', x.shape, 'y mean/std', float(y.mean()), float(y.std()), 'valid y mean/std', float(vy.mean()), float(vy.std()))
eng = tr.prob_model.engine(x, y)
tr.rank, tr.world_size = 0, 1
tr.n_chains = 8
params = torch.from_numpy(BDETrainer.init_module_params(tr, range(8)))
prior = tr.prob_model.prior
theta = params.cuda()
for ep in range(6):
    logp, g = eng.logpost_grad(theta)
    lp_prior, _ = prior_value_and_grad(prior, theta)
    tr_nll = -(logp - lp_prior) / len(x)
    v = -eng.pointwise_loglik(theta, vx, vy).mean(dim=-1)
    t2 = -eng.pointwise_loglik(theta, x, y).mean(dim=-1)
    print(f'epoch {ep}: train nll (engine) {tr_nll.mean().item():.4f}  train nll (pointwise) {t2.mean().item():.4f}  valid nll {v.mean().item():.4f}')
    theta, hist = train_deep_ensemble(eng, prior, theta, n_train=len(x), valid_x=None, valid_y=None, optimizer='adamw',
                                      optimizer_parameters={'learning_rate': 0.001, 'b1': 0.9, 'b2': 0.999, 'weight_decay': 0.001},
                                      max_epochs=1, batch_size=32, patience=None)
