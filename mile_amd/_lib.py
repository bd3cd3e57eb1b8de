"""ctypes binding of include/mile_hip.h.  Fails loudly if the HIP library is missing."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

from mile_amd._build import LIB_PATH

MILE_MAX_LAYERS = 16
ABI_VERSION = 4

ACTIVATION_IDS = {'relu': 0, 'tanh': 1, 'sigmoid': 2}
TASK_IDS = {'regr': 0, 'regression': 0, 'classification': 1, 'class': 1}
PRIOR_IDS = {'Normal': 0, 'StandardNormal': 0, 'Laplace': 1}
REFRESH_IDS = {'O-step-O': 0, 'step-O': 1}
GRAD_KERNEL_IDS = {'auto': 0, 'generic': 1, 'mfma_w64': 2, 'mfma_w128_bf16': 3, 'gemm_f32': 4, 'lenet_f32': 5, 'mfma_w64_bf16x3': 6,
                   'mfma_wide_bf16x3': 7, 'mfma_wide_bf16': 8, 'lenet_bf16': 9, 'mfma_narrow_f32': 10}


class ModelSpecC(C.Structure):
    _fields_ = [
        ('in_features', C.c_int32),
        ('n_layers', C.c_int32),
        ('widths', C.c_int32 * MILE_MAX_LAYERS),
        ('activation', C.c_int32),
        ('task', C.c_int32),
        ('prior', C.c_int32),
        ('prior_loc', C.c_float),
        ('prior_scale', C.c_float),
        ('use_bias', C.c_int32),
        ('model', C.c_int32),
        ('img_c', C.c_int32),
        ('img_h', C.c_int32),
        ('img_w', C.c_int32),
    ]


class StateC(C.Structure):
    _fields_ = [
        ('n_particles', C.c_int32),
        ('position', C.c_void_p),
        ('momentum', C.c_void_p),
        ('logdensity', C.c_void_p),
        ('logdensity_grad', C.c_void_p),
    ]


class StepArgsC(C.Structure):
    _fields_ = [
        ('step_size', C.c_void_p),
        ('L', C.c_void_p),
        ('sqrt_diag_cov', C.c_void_p),
        ('noise', C.c_void_p),
        ('seed', C.c_uint64),
        ('particle_ids', C.c_void_p),
        ('step_offset', C.c_int64),
        ('n_steps', C.c_int32),
        ('n_thinning', C.c_int32),
        ('refresh', C.c_int32),
        ('out_samples', C.c_void_p),
        ('out_info', C.c_void_p),
    ]


class TuneArgsC(C.Structure):
    _fields_ = [
        ('step_size', C.c_void_p), ('L', C.c_void_p), ('sqrt_diag_cov', C.c_void_p),
        ('step_size_max', C.c_void_p), ('time', C.c_void_p), ('x_average', C.c_void_p),
        ('stream_weight', C.c_void_p), ('stream_average', C.c_void_p), ('noise', C.c_void_p),
        ('seed', C.c_uint64), ('particle_ids', C.c_void_p), ('step_offset', C.c_int64),
        ('n_steps', C.c_int32), ('schedule_step0', C.c_int32), ('n_mask_steps', C.c_int32),
        ('schedule_total', C.c_int32), ('desired_energy_var_start', C.c_float),
        ('desired_energy_var_end', C.c_float), ('trust_in_estimate', C.c_float), ('decay_rate', C.c_float),
        ('refresh', C.c_int32), ('out_info', C.c_void_p),
    ]


class OptimArgsC(C.Structure):
    _fields_ = [
        ('kind', C.c_int32), ('learning_rate', C.c_float), ('b1', C.c_float), ('b2', C.c_float), ('eps', C.c_float),
        ('weight_decay', C.c_float), ('t', C.c_int64), ('m', C.c_void_p), ('v', C.c_void_p), ('active', C.c_void_p),
        ('out_nll', C.c_void_p),
    ]


OPTIMIZER_IDS = {'sgd': 0, 'adam': 1, 'adamw': 2}

# name -> (restype, argtypes): every symbol include/mile_hip.h declares
SIGNATURES = {
    'mile_last_error': (C.c_char_p, []),
    'mile_abi_version': (C.c_int32, []),
    'mile_create': (C.c_int32, [C.POINTER(ModelSpecC), C.c_int32, C.POINTER(C.c_void_p)]),
    'mile_destroy': (C.c_int32, [C.c_void_p]),
    'mile_param_count': (C.c_int64, [C.c_void_p]),
    'mile_param_offsets': (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'mile_set_data': (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'mile_set_row_window': (C.c_int32, [C.c_void_p, C.c_int64, C.c_int64]),
    'mile_reserve': (C.c_int32, [C.c_void_p, C.c_int32]),
    'mile_warmstart_step': (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(OptimArgsC), C.c_void_p]),
    'mile_set_grad_kernel': (C.c_int32, [C.c_void_p, C.c_int32]),
    'mile_get_grad_kernel': (C.c_int32, [C.c_void_p]),
    'mile_logpost_grad': (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    'mile_init': (C.c_int32, [C.c_void_p, C.POINTER(StateC), C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    'mile_step': (C.c_int32, [C.c_void_p, C.POINTER(StateC), C.POINTER(StepArgsC), C.c_void_p]),
    'mile_tune': (C.c_int32, [C.c_void_p, C.POINTER(StateC), C.POINTER(TuneArgsC), C.c_void_p]),
    'mile_pointwise_loglik': (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                          C.c_void_p, C.c_void_p]),
    'mile_debug_noise': (C.c_int32, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32, C.c_int64, C.c_int32,
                                     C.c_void_p, C.c_void_p]),
    'mile_grad_launch_info': (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                          C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_char_p, C.c_int32]),
    'mile_grad_timing_begin': (C.c_int32, [C.c_void_p]),
    'mile_grad_timing_end': (C.c_int32, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
}

_lib = None


class MileHipError(RuntimeError):
    pass


def load_library(path: str | Path | None = None) -> C.CDLL:
    """dlopen libmile_hip.so.  There is NO fallback: without the HIP library the product
    path cannot run, and says so."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else LIB_PATH
    if not p.exists():
        raise MileHipError(
            f'{p} not found: the MI355X HIP extension is not built. Run '
            '`python -c "import __graft_entry__ as g; g.build()"` (needs hipcc). '
            'mile_amd has no CPU fallback.')
    try:
        lib = C.CDLL(str(p))
    except OSError as exc:
        raise MileHipError(f'cannot load {p}: {exc}') from exc
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise MileHipError(f'{p} does not export {name}') from exc
        fn.restype = res
        fn.argtypes = args
    if lib.mile_abi_version() != ABI_VERSION:
        raise MileHipError(f'ABI mismatch: library {lib.mile_abi_version()} != binding {ABI_VERSION}')
    if path is None:
        _lib = lib
    return lib


def check(rc: int, lib: C.CDLL | None = None):
    if rc != 0:
        lib = lib or load_library()
        msg = lib.mile_last_error()
        raise MileHipError(f'libmile_hip error {rc}: {msg.decode() if msg else "?"}')
