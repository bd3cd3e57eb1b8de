"""ctypes access to oracle/libcpu_mclmc.so (the C/OpenMP restatement) -- test and baseline
infrastructure only, like the rest of oracle/."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / 'libcpu_mclmc.so'


class CpuSpec(C.Structure):
    _fields_ = [('n_layers', C.c_int), ('in_features', C.c_int), ('widths', C.c_int * 16), ('w_off', C.c_int * 16),
                ('b_off', C.c_int * 16), ('d', C.c_int), ('prior_loc', C.c_float), ('prior_scale', C.c_float),
                ('activation', C.c_int), ('task', C.c_int), ('prior', C.c_int)]


def effective_cpus() -> int:
    """CPUs this process may really use: affinity mask and cgroup quota (a container can see 128 CPUs and own 16)."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, math.ceil(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                    n = min(n, max(1, math.ceil(q / per)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def load(build: bool = True):
    if build and (not LIB.exists() or LIB.stat().st_mtime < (HERE / 'cpu_mclmc.c').stat().st_mtime):
        subprocess.run(['make', '-C', str(HERE)], check=True, capture_output=True)
    lib = C.CDLL(str(LIB))
    lib.cpu_threads.restype = C.c_int
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class CpuPort:
    """Any FCN of the oracle's ModelSpec with < 11 layers: relu / tanh / sigmoid, regression or classification head, Normal
    or Laplace prior (BASELINE configs B1-B4)."""

    ACT = {'relu': 0, 'tanh': 1, 'sigmoid': 2}
    TASK = {'regr': 0, 'classification': 1}
    PRIOR = {'Normal': 0, 'Laplace': 1}

    def __init__(self, spec, X, y):
        assert len(spec.hidden_structure) < 11        # ravel_pytree order == layer order only below 'layer10'
        self.lib = load()
        self.cs = CpuSpec()
        w = (C.c_int * len(spec.hidden_structure))(*spec.hidden_structure)
        self.lib.cpu_spec_init(C.byref(self.cs), spec.in_features, len(spec.hidden_structure), w,
                               C.c_float(spec.prior_loc), C.c_float(spec.prior_scale), self.ACT[spec.activation],
                               self.TASK[spec.task], self.PRIOR[spec.prior])
        assert self.cs.d == spec.n_params
        self.X = np.ascontiguousarray(X, np.float32)
        self.y = np.ascontiguousarray(y, np.float32 if spec.task == 'regr' else np.int32)
        self.lib.cpu_set_threads(effective_cpus())
        self.threads = int(self.lib.cpu_threads())

    def logpost_grad(self, theta):
        theta = np.ascontiguousarray(theta, np.float32)
        E = theta.shape[0]
        logp, grad = np.empty(E, np.float32), np.empty_like(theta)
        self.lib.cpu_logpost_grad(C.byref(self.cs), _p(theta), E, _p(self.X), _p(self.y), len(self.y), _p(logp), _p(grad))
        return logp, grad

    def steps(self, x, u, logp, g, eps, L, noise, want_info=False):
        """In place on x, u, logp, g (fp32 contiguous).  noise [T, 2, E, d]."""
        T, E = noise.shape[0], x.shape[0]
        info = np.empty((T, E, 3), np.float32) if want_info else None
        self.lib.cpu_mclmc_steps(C.byref(self.cs), _p(x), _p(u), _p(logp), _p(g), E,
                                 _p(np.ascontiguousarray(eps, np.float32)), _p(np.ascontiguousarray(L, np.float32)),
                                 _p(np.ascontiguousarray(noise, np.float32)), T, _p(self.X), _p(self.y), len(self.y),
                                 _p(info) if info is not None else None)
        return info
