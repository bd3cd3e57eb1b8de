"""YAML/JSON experiment configuration: the subset of src/config/* on the MCLMC path, with
the SAME keys as experiments/replicate_uci/mclmc.yaml and the same strictness (unknown keys
are rejected, src/config/base.py:740-753).
"""
from __future__ import annotations

import dataclasses
import logging
import sys
import time
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any

import yaml

from mile_amd.priors import Prior, PriorDist


class ConfigError(ValueError):
    pass


def _from_dict(cls, data: dict, path: str = ''):
    if not isinstance(data, dict):
        raise ConfigError(f'{path or cls.__name__}: expected a mapping, got {type(data).__name__}')
    names = {f.name: f for f in dataclasses.fields(cls)}
    unknown = set(data) - set(names)
    if unknown:
        raise ConfigError(f'{path or cls.__name__}: unknown field(s) {sorted(unknown)}')
    kw = {}
    for k, v in data.items():
        f = names[k]
        sub = _NESTED.get((cls.__name__, k))
        if sub is not None and not isinstance(sub, type):
            sub = sub(v)                      # dispatch on the mapping's content (model family)
        kw[k] = _from_dict(sub, v, f'{path}.{k}' if path else k) if sub is not None and v is not None else v
    missing = [n for n, f in names.items() if n not in kw and f.default is dataclasses.MISSING
               and f.default_factory is dataclasses.MISSING]
    if missing:
        raise ConfigError(f'{path or cls.__name__}: missing required field(s) {missing}')
    return cls(**kw)


def _check(cond, msg):
    if not cond:
        raise ConfigError(msg)


@dataclass(frozen=True)
class DataConfig:
    """src/config/data.py:51-131."""

    path: str
    source: str
    data_type: str
    task: str
    target_column: Any = None
    target_len: int = 1
    features: Any = None
    datapoint_limit: int | None = None
    normalize: bool = False
    train_split: float = 0.8
    valid_split: float = 0.1
    test_split: float = 0.1

    def __post_init__(self):
        _check(self.task in ('regr', 'class'), f'data.task must be regr or class, got {self.task!r}')
        _check(self.data_type in ('tabular', 'image', 'text'), f'unknown data_type {self.data_type!r}')
        _check(self.source in ('local', 'url', 'huggingface', 'torchvision', 'synthetic'), f'unknown source {self.source!r}')
        _check((self.train_split + self.valid_split + self.test_split - 1.0) < 1e-6,
               'Train, Validation, and Test Split should sum to 1.0')


@dataclass(frozen=True)
class FCNConfig:
    """src/config/models/fcn.py:7-30."""

    model: str = 'FCN'
    hidden_structure: list = field(default_factory=lambda: [10, 10])
    activation: str = 'relu'
    use_bias: bool = True

    def __post_init__(self):
        _check(self.model == 'FCN', f'Could not find model {self.model}. Avaliable models: [\'FCN\'] '
                                   '(only the FCN is on the MI355X hot path)')
        _check(isinstance(self.hidden_structure, (list, tuple)) and all(isinstance(w, int) and w > 0 for w in self.hidden_structure),
               'hidden_structure must be a list of positive ints')
        _check(self.activation in ('sigmoid', 'relu', 'gelu', 'tanh', 'softmax', 'leaky_relu'),
               f'unknown activation {self.activation!r}')
        # the reference's enum (config/models/base.py:25-39) has three more; the HIP kernels implement these
        _check(self.activation in ('sigmoid', 'relu', 'tanh'),
               f'activation {self.activation!r} is not implemented on the MI355X path (supported: relu, tanh, sigmoid)')


@dataclass(frozen=True)
class LeNetConfig:
    """src/config/models/cnns.py:7-24."""

    model: str = 'LeNet'
    activation: str = 'sigmoid'
    out_dim: int = 10
    use_bias: bool = True

    def __post_init__(self):
        _check(self.model == 'LeNet', f'Could not find model {self.model}.')
        _check(self.activation in ('sigmoid', 'relu', 'gelu', 'tanh', 'softmax', 'leaky_relu'),
               f'unknown activation {self.activation!r}')
        # the reference's enum (config/models/base.py:25-39) has three more; the HIP kernels implement these
        _check(self.activation in ('sigmoid', 'relu', 'tanh'),
               f'activation {self.activation!r} is not implemented on the MI355X path (supported: relu, tanh, sigmoid)')
        _check(isinstance(self.out_dim, int) and self.out_dim > 0, 'out_dim must be a positive int')


def _model_config(data):
    """ModelConfig.from_dict dispatch on the `model` key (src/config/models/__init__.py)."""
    name = data.get('model', 'FCN') if isinstance(data, dict) else 'FCN'
    if name == 'LeNet':
        return LeNetConfig
    if name == 'FCN':
        return FCNConfig
    raise ConfigError(f"Could not find model {name}. Available on the MI355X hot path: ['FCN', 'LeNet']")


@dataclass(frozen=True)
class PriorConfig:
    """src/config/sampler.py:60-95."""

    name: str = PriorDist.StandardNormal
    parameters: dict = field(default_factory=dict)

    def __post_init__(self):
        _check(self.name in PriorDist.ALL, f'unknown prior {self.name!r}')

    def get_prior(self) -> Prior:
        return Prior.from_name(self.name, **self.parameters)


@dataclass(frozen=True)
class SamplerConfig:
    """src/config/sampler.py:97-216 (same fields, same defaults)."""

    name: str = 'nuts'
    epoch_wise_sampling: bool = False
    params_frozen: list = field(default_factory=list)
    warmup_steps: int = 50
    n_chains: int = 2
    n_samples: int = 1000
    use_warmup_as_init: bool = True
    n_thinning: int = 1
    diagonal_preconditioning: bool = False
    desired_energy_var_start: float = 5e-4
    desired_energy_var_end: float = 1e-4
    trust_in_estimate: float = 1.5
    num_effective_samples: int = 100
    step_size_init: float = 0.005
    keep_warmup: bool = False
    prior_config: PriorConfig = field(default_factory=PriorConfig)
    partition_sampling: bool = False
    # extension (not in the reference schema, optional): which grad kernel the HIP sampler uses.
    # 'auto' = fastest fp32 kernel; 'mfma_w128_bf16' = bf16 matrix operands (BASELINE config 3)
    grad_kernel: str = 'auto'

    def __post_init__(self):
        _check(self.name in ('nuts', 'mclmc', 'hmc', 'mclmc_hip'), f'unknown sampler {self.name!r}')
        _check(self.grad_kernel in ('auto', 'generic', 'mfma_w64', 'mfma_w64_bf16x3', 'mfma_w128_bf16', 'gemm_f32', 'mfma_wide_bf16x3',
                                    'mfma_wide_bf16', 'lenet_f32', 'lenet_bf16', 'mfma_narrow_f32'), f'unknown grad_kernel {self.grad_kernel!r}')

    @property
    def prior(self) -> Prior:
        return self.prior_config.get_prior()

    @property
    def kernel(self):
        """Sampler.get_kernel (src/config/sampler.py:29-42)."""
        from mile_amd.kernels import KERNELS
        if self.name not in KERNELS:
            raise NotImplementedError(f'Sampler for {self.name} is not yet implemented.')
        return KERNELS[self.name]

    @property
    def _warmup_dir_name(self):
        return 'sampling_warmup'

    @property
    def _dir_name(self):
        return 'samples'


@dataclass(frozen=True)
class OptimizerConfig:
    name: str = 'adamw'
    parameters: dict = field(default_factory=dict)


@dataclass(frozen=True)
class WarmStartConfig:
    """src/config/warmstart.py:45-75.  Parsed for schema compatibility; the optax warm-start
    training loop is outside the hot path -- see mile_amd.trainer."""

    include: bool = True
    optimizer_config: OptimizerConfig = field(default_factory=OptimizerConfig)
    warmstart_exp_dir: str | None = None
    max_epochs: int = 100
    batch_size: int | None = None
    patience: int | None = None
    partition_warmstart: bool = False

    @property
    def _dir_name(self):
        return 'warmstart'


@dataclass(frozen=True)
class TrainingConfig:
    warmstart: WarmStartConfig = field(default_factory=WarmStartConfig)
    sampler: SamplerConfig = field(default_factory=SamplerConfig)
    tokenizer: Any = None


@dataclass(frozen=True)
class Config:
    """src/config/core.py:25-70."""

    experiment_name: str
    data: DataConfig
    model: Any
    training: TrainingConfig = field(default_factory=TrainingConfig)
    saving_dir: str = 'results/'
    rng: int = 42
    logging: bool = True

    @classmethod
    def from_dict(cls, d: dict) -> 'Config':
        return _from_dict(cls, d)

    @classmethod
    def from_yaml(cls, path) -> 'Config':
        with open(path) as f:
            return cls.from_dict(yaml.safe_load(f))

    @classmethod
    def from_file(cls, path) -> 'Config':
        path = Path(path)
        if path.suffix in ('.yaml', '.yml'):
            return cls.from_yaml(path)
        if path.suffix == '.json':
            import json
            with open(path) as f:
                return cls.from_dict(json.load(f))
        raise ConfigError(f'unsupported config format: {path.suffix}')

    @classmethod
    def from_dir(cls, path) -> list['Config']:
        return [cls.from_file(p) for p in sorted(Path(path).iterdir()) if p.suffix in ('.yaml', '.yml', '.json')]

    def to_dict(self) -> dict:
        return dataclasses.asdict(self)

    def to_yaml(self, path):
        with open(path, 'w') as f:
            yaml.safe_dump(self.to_dict(), f, sort_keys=False)

    def replace(self, **kw) -> 'Config':
        return dataclasses.replace(self, **kw)

    @property
    def n_chains(self):
        return self.training.sampler.n_chains

    @property
    def experiment_dir(self) -> Path:
        return Path(self.saving_dir) / self.experiment_name

    def setup_dir(self) -> 'Config':
        """core.py:231-247: unique experiment dir, config.yaml dump, training.log."""
        cfg = self
        if cfg.experiment_dir.exists():
            cfg = cfg.replace(experiment_name=f"{cfg.experiment_name}_{time.strftime('%Y%m%d-%H%M%S')}")
        cfg.experiment_dir.mkdir(parents=True)
        cfg.to_yaml(cfg.experiment_dir / 'config.yaml')
        handlers = [logging.FileHandler(cfg.experiment_dir / 'training.log')]
        if cfg.logging:
            handlers.append(logging.StreamHandler(sys.stdout))
        logging.basicConfig(handlers=handlers, level=logging.INFO, force=True,
                            format='%(asctime)s - %(name)s - %(levelname)s - %(message)s')
        logging.getLogger(__name__).info('Logging successfully setup.')
        return cfg


_NESTED = {
    ('Config', 'data'): DataConfig,
    ('Config', 'model'): _model_config,
    ('Config', 'training'): TrainingConfig,
    ('TrainingConfig', 'warmstart'): WarmStartConfig,
    ('TrainingConfig', 'sampler'): SamplerConfig,
    ('WarmStartConfig', 'optimizer_config'): OptimizerConfig,
    ('SamplerConfig', 'prior_config'): PriorConfig,
}
