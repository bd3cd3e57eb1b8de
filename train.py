#!/usr/bin/env python3
"""Train Script (same surface as the reference's train.py:22-133):

    python train.py -c experiments/mclmc_airfoil.yaml [-d N] [--silent]

``-d`` is the number of MI355X GPUs; for N > 1 launch under torchrun
(`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 train.py -c ...`):
each rank samples its block of the chains and writes its own samples/<chain>/ files.
"""
import argparse
import logging
import os
import warnings
from pathlib import Path

logger = logging.getLogger(__name__)


def train_bde(config, n_devices: int):
    from mile_amd.trainer import BDETrainer
    logger.info(f'> Running experiment: {config.experiment_name}')
    trainer = BDETrainer(config=config)
    trainer.train_bde()


if __name__ == '__main__':
    parser = argparse.ArgumentParser(prog='train.py', description='Train a BDE model for a given configuration.',
                                     epilog='Example usage: python3 train.py -c experiments/mclmc_airfoil.yaml')
    parser.add_argument('--config', '-c', type=str, required=True, metavar='',
                        help='Path to the configuration file or directory.')
    parser.add_argument('--search_tree', '-s', type=str, required=False, metavar='',
                        help='Path to the search tree file (not supported on the MI355X path).')
    parser.add_argument('--devices', '-d', type=int, default=1, metavar='', help='Number of GPUs.')
    parser.add_argument('--device_limit', type=int, default=1, metavar='', help='Accepted for compatibility.')
    parser.add_argument('--silent', action='store_true', help='Disable logging to console.')
    parser.add_argument('--outer_parallel', action='store_true', default=False,
                        help='Accepted for compatibility (experiments run sequentially).')
    args = parser.parse_args()
    config_path = Path(args.config)
    if args.silent:
        logging.basicConfig(level=logging.WARNING)
    from mile_amd.config import Config
    if not config_path.exists():
        raise FileNotFoundError(f'Configuration file or directory not found: {config_path}')
    if config_path.is_dir():
        if args.search_tree:
            warnings.warn('Ignoring search tree file when loading directory of configs.', category=UserWarning)
        configs = Config.from_dir(config_path)
    else:
        if args.search_tree:
            raise NotImplementedError('grid search over a search tree is outside the MI355X hot path')
        configs = [Config.from_file(config_path)]
    ws = int(os.environ.get('WORLD_SIZE', 1))
    if args.devices > 1 and ws == 1:
        raise SystemExit(f'-d {args.devices}: launch under torchrun with --nproc-per-node {args.devices} '
                         '(one process per GPU)')
    if args.silent:
        configs = [c.replace(logging=False) for c in configs]
    logger.info(f'Loaded {len(configs)} Experiment(s)')
    import torch
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
    for cfg in configs:
        train_bde(cfg, args.devices)
