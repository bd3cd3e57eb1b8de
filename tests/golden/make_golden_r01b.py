"""Golden fixtures of the two oracle restatements added later in round 1 (run: python tests/golden/make_golden_r01b.py).
Generated FROM THE ORACLE, like make_golden.py: they pin the oracle against regressions, not parity with the reference."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import lenet_oracle as LN  # noqa: E402
from oracle import mclmc_oracle as O  # noqa: E402

OUT = Path(__file__).resolve().parent

if __name__ == '__main__':
    # bf16-operand recipe (the checker of k_grad_w128b) on a width-128 net
    spec = O.ModelSpec(9, (128, 128, 2))
    prob = O.synthetic_problem(spec, 48, 2, seed=21)
    lp, g = O.logpost_and_grad_bf16(spec, prob['theta0'].astype(np.float64), prob['X'], prob['y'])
    np.savez_compressed(OUT / 'bf16_recipe_128x2.npz', X=prob['X'], y=prob['y'], theta0=prob['theta0'], logp=lp, grad=g.astype(np.float32),
                        grad_norm=np.linalg.norm(g, axis=1))
    # LeNet target
    ls = LN.LeNetSpec(2, 12, 14, 3, activation='tanh')
    pl = LN.synthetic_problem(ls, 5, 2, seed=22)
    lp, g = LN.logpost_and_grad(ls, pl['theta0'].astype(np.float64), pl['X'], pl['y'])
    np.savez_compressed(OUT / 'lenet_2x12x14.npz', X=pl['X'], y=pl['y'], theta0=pl['theta0'], logp=lp, grad=g)
    print('written')
