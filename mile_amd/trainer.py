"""BDETrainer: the caller of the hot path (mirror of src/training/trainer.py:60-177,543-607).

In scope: directory / logging setup, data loading, the probabilistic model, ``train_plan``
chain grouping and ``start_sampling``.  Out of scope this round: the optax
warm-start training loop -- chains start from ``warmstart_exp_dir`` params if given, else from
``module.init``'s distribution (what the reference does when warm-start is disabled, trainer.py:569-573).
"""
from __future__ import annotations

import logging
import math
import os
import time
from contextlib import contextmanager
from pathlib import Path

import numpy as np
import torch

from mile_amd import distributed as mdist
from mile_amd.callbacks import load_params_batch, save_params, save_tree
from mile_amd.config import Config
from mile_amd.dataset import ImageLoader, TabularLoader
from mile_amd.probabilistic import ProbabilisticModel
from mile_amd.sampling import inference_loop
from mile_amd.spec import LeNetSpec, ModelSpec
from mile_amd.tree import PRNGKey

logger = logging.getLogger(__name__)


@contextmanager
def measure_time(name: str):
    """src/utils.py:25-31: the report notebook greps '<name> took X seconds' from training.log."""
    t0 = time.time()
    yield
    logger.info(f'{name} took {time.time() - t0:.2f} seconds')


def train_plan(n_chains: int, n_devices: int) -> list[np.ndarray]:
    """trainer.py:75-82: array_split(arange(n_chains), n_chains / n_devices); groups run sequentially."""
    if n_chains % n_devices:
        raise ValueError('n_chains must be divisible by the number of devices.'
                         f'{n_chains} % {n_devices} != 0.')
    return np.array_split(np.arange(n_chains), n_chains // n_devices)


class BDETrainer:
    def __init__(self, config: Config, chains_per_group: int | None = None):
        assert isinstance(config, Config)
        self.rank, self.world_size, self.local_rank = mdist.world()
        # Rank 0 resolves the experiment directory (setup_dir renames the experiment when the directory already
        # exists); every other rank must derive its paths from THAT name, or it writes into the previous run.
        if self.world_size > 1:
            mdist.init_process_group()
            # rank 0 broadcasts (name, error): if its setup_dir() fails the other ranks must not wait in the broadcast forever
            msg = [None]
            if self.rank == 0:
                try:
                    msg = [(config.setup_dir().experiment_name, None)]
                except Exception as exc:                      # noqa: BLE001 -- re-raised below, on every rank
                    msg = [(None, f'{type(exc).__name__}: {exc}')]
            mdist.broadcast_object(msg)
            name, err = msg[0]
            if err is not None:
                raise RuntimeError(f'rank 0 could not set up the experiment directory: {err}')
            self.config = config.replace(experiment_name=name)
        else:
            self.config = config.setup_dir()
        self._key = PRNGKey(config.rng)
        self.n_chains = config.n_chains
        # all chains of a rank advance together in one ensemble launch; `chains_per_group`
        # reproduces the reference's sequential chain groups when set
        self.n_devices = chains_per_group or self.n_chains
        self.train_plan = train_plan(self.n_chains, self.n_devices)
        self.build_model(config)
        self.exp_dir = self.config.experiment_dir
        if self.rank == 0:
            save_tree(self.exp_dir, self.prob_model.spec)
        logger.info(f'> Trainer has been successfully initialized\n{self.prob_model}')

    def build_model(self, config: Config):
        """Data loader, model spec and probabilistic model of a config (also used by evaluate.py)."""
        task = 'regr' if config.data.task == 'regr' else 'classification'
        if config.model.model == 'LeNet':
            if config.data.data_type != 'image':
                raise ValueError('model LeNet needs data_type: image')
            self.loader = ImageLoader(config.data, rng=config.rng)
            _, C, H, W = self.loader.train_x.shape
            self.spec_model = LeNetSpec(channels=C, height=H, width=W, out_dim=config.model.out_dim,
                                        activation=config.model.activation, task=task)
        else:
            if config.data.data_type != 'tabular':
                raise NotImplementedError('the FCN runs on tabular data; image data goes with model LeNet')
            self.loader = TabularLoader(config.data, rng=config.rng, target_len=config.data.target_len)
            F = self.loader.train_x.shape[-1]
            self.spec_model = ModelSpec(in_features=F, hidden_structure=tuple(config.model.hidden_structure),
                                        activation=config.model.activation, task=task)
        self.prob_model = ProbabilisticModel(module=self.spec_model, prior=config.training.sampler.prior,
                                             task=config.data.task, n_batches=1,
                                             grad_kernel=config.training.sampler.grad_kernel)

    @property
    def key(self) -> PRNGKey:
        self._key, k = self._key.split(2)
        return k

    def init_module_params(self, chain_ids) -> np.ndarray:
        """Random parameters as `module.init` gives them (trainer.py:206-228, 904-917): flax Dense defaults --
        kernel lecun_normal (truncated normal on [-2, 2] standard deviations, variance 1/fan_in), bias zeros --
        one stream per GLOBAL chain id.  (JAX's PRNG is not reproducible here; the distribution is.)"""
        spec = self.prob_model.spec
        rows = []
        for cid in chain_ids:
            g = torch.Generator().manual_seed((self.config.rng * 1000003 + int(cid)) & 0x7FFFFFFFFFFFFFFF)
            flat = torch.zeros(spec.n_params, dtype=torch.float32)
            for name, off, shape in spec.leaves():
                if name.endswith('kernel'):
                    w = torch.empty(shape, dtype=torch.float32)
                    torch.nn.init.trunc_normal_(w, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=g)
                    # variance_scaling(1.0, 'fan_in', 'truncated_normal'): std = sqrt(1/fan_in) / 0.87962566...
                    fan_in = int(np.prod(shape[:-1]))             # Dense: in; Conv: kh * kw * in
                    flat[off:off + w.numel()] = (w * (math.sqrt(1.0 / fan_in) / 0.87962566103423978)).reshape(-1)
            rows.append(flat.numpy())
        return np.stack(rows).astype(np.float32)

    def train_bde(self):
        with measure_time('time.warmstart'):
            self.train_warmstart()
        if self.world_size > 1:                       # rank 0 wrote the members every rank starts from
            import torch.distributed as dist
            if not dist.is_initialized():
                mdist.init_process_group()
            dist.barrier()
        self.start_sampling()

    def train_warmstart(self):
        """trainer.py:330-392: train the deep-ensemble members the chains start from (mile_amd/warmstart.py: the
        reference's optimizer, epoch / validation / early-stopping schedule, full-batch gradients from the HIP engine),
        or reuse the members of `warmstart_exp_dir`; with `include: false` the chains start from module.init-style
        random parameters."""
        ws = self.config.training.warmstart
        if ws.warmstart_exp_dir or self.rank != 0:
            return
        wdir = self.exp_dir / ws._dir_name
        params = self.init_module_params(range(self.n_chains))
        if ws.include:
            from mile_amd.warmstart import train_deep_ensemble
            tx = np.ascontiguousarray(self.loader.train_x).reshape(len(self.loader.train_x), -1)
            x, y = torch.from_numpy(tx), torch.from_numpy(np.ascontiguousarray(self.loader.train_y))
            eng = self.prob_model.engine(x, y)
            vx = np.ascontiguousarray(self.loader.valid_x).reshape(len(self.loader.valid_x), -1)
            logger.info(f'\t| Starting Training Warmstart for chains {list(range(self.n_chains))}')
            theta, hist = train_deep_ensemble(
                eng, self.prob_model.prior, torch.from_numpy(params), n_train=len(tx),
                valid_x=torch.from_numpy(vx) if len(vx) else None,
                valid_y=torch.from_numpy(np.ascontiguousarray(self.loader.valid_y)) if len(vx) else None,
                optimizer=ws.optimizer_config.name, optimizer_parameters=ws.optimizer_config.parameters,
                max_epochs=ws.max_epochs, batch_size=ws.batch_size, patience=ws.patience, train_x=x, train_y=y,
                seed=self.config.rng)
            params = theta.cpu().numpy()
            self._engine_inputs = (x, y)               # the sampler reuses the same engine (keyed by tensor identity)
            logger.info(f"\t| Warmstart Training completed after {hist['epochs']} epochs "
                        f"({int(hist['stopped'].sum())}/{self.n_chains} members stopped early)")
        for i in range(self.n_chains):
            save_params(wdir, self.prob_model.spec, params[i], i)

    def start_sampling(self):
        """trainer.py:543-607 (full-batch, non-partition branch)."""
        with measure_time('time.sampling'):
            cfgs = self.config.training.sampler
            ws = self.config.training.warmstart
            warm_exp = ws.warmstart_exp_dir or str(self.exp_dir)
            warm_path = Path(warm_exp) / ws._dir_name
            chains = []
            if warm_path.exists():
                chains = sorted((warm_path / i for i in os.listdir(warm_path) if i.startswith('params')),
                                key=lambda p: int(p.stem.split('_')[-1]))
            if getattr(self, '_engine_inputs', None) is not None:
                x, y = self._engine_inputs
            else:
                x = torch.from_numpy(np.ascontiguousarray(self.loader.train_x).reshape(len(self.loader.train_x), -1))
                y = torch.from_numpy(np.ascontiguousarray(self.loader.train_y))
            log_post = self.prob_model.bind(x, y)
            for step in self.train_plan:
                mine = mdist.shard_chains(step, self.world_size, self.rank)
                if len(mine) == 0:
                    # fewer chains in the group than ranks (4 chains on 8 GPUs): this rank has nothing to sample but still
                    # joins the group's one collective -- the gather of the tuned (step_size, L) for warmup_params.txt
                    logger.info('\t| No chain of this group on this rank')
                    mdist.gather_objects((np.zeros(0, np.float32), np.zeros(0, np.float32)))
                    continue
                logger.info(f'\t| Starting Sampling for chains {mine}')
                if chains:
                    params = load_params_batch([chains[i] for i in mine], self.prob_model.spec)
                else:
                    logger.warning('\t| No warmstart path found, starting sampling from random params.')
                    params = self.init_module_params(mine)
                inference_loop(unnorm_log_posterior=log_post, config=cfgs, rng_key=self.key,
                               init_params=torch.from_numpy(params), step_ids=mine,
                               saving_path=self.exp_dir / cfgs._dir_name,
                               saving_path_warmup=self.exp_dir / cfgs._warmup_dir_name)
