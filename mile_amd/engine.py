"""Thin torch <-> C-ABI shim around libmile_hip.so.

PyTorch-ROCm is plumbing here: it owns device memory and the stream; every compute
call goes through the C ABI of include/mile_hip.h on raw device pointers.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple

import torch

from mile_amd import _lib
from mile_amd.spec import LeNetSpec, ModelSpec


class IntegratorState(NamedTuple):
    """blackjax IntegratorState with a leading ensemble axis (one row per chain)."""

    position: torch.Tensor         # [E, d]
    momentum: torch.Tensor         # [E, d]
    logdensity: torch.Tensor       # [E]
    logdensity_grad: torch.Tensor  # [E, d]


class MCLMCInfo(NamedTuple):
    """blackjax MCLMCInfo, [n_steps, E] each."""

    logdensity: torch.Tensor
    kinetic_change: torch.Tensor
    energy_change: torch.Tensor


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _f32(t, device, shape=None, name='tensor'):
    t = torch.as_tensor(t, device=device)
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    t = t.contiguous()
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f'{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}')
    return t


class Engine:
    """One handle == one device, one model spec, one training set."""

    def __init__(self, spec: ModelSpec, X, y, device=None, grad_kernel: str = 'auto'):
        if not torch.cuda.is_available():
            raise _lib.MileHipError('mile_amd needs an MI355X (torch.cuda.is_available() is False); '
                                    'there is no CPU fallback.')
        self.lib = _lib.load_library()
        self.spec = spec
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        cs = _lib.ModelSpecC()
        cs.in_features = spec.in_features
        if isinstance(spec, LeNetSpec):
            cs.model, cs.img_c, cs.img_h, cs.img_w = 1, spec.channels, spec.height, spec.width
        cs.n_layers = len(spec.hidden_structure)
        if cs.n_layers > _lib.MILE_MAX_LAYERS:
            raise ValueError(f'at most {_lib.MILE_MAX_LAYERS} layers')
        for i, w in enumerate(spec.hidden_structure):
            cs.widths[i] = w
        cs.activation = _lib.ACTIVATION_IDS[spec.activation]
        cs.task = _lib.TASK_IDS[spec.task]
        cs.prior = _lib.PRIOR_IDS[spec.prior]
        cs.prior_loc = spec.prior_loc
        cs.prior_scale = spec.prior_scale
        cs.use_bias = int(spec.use_bias)
        h = C.c_void_p()
        _lib.check(self.lib.mile_create(C.byref(cs), self.device.index or 0, C.byref(h)), self.lib)
        self._h = h
        self.d = int(self.lib.mile_param_count(h))
        assert self.d == spec.n_params
        self._E_reserved = 0
        self.set_data(X, y)
        if grad_kernel != 'auto':
            self.set_grad_kernel(grad_kernel)

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            self.lib.mile_destroy(h)
            self._h = None

    # ------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_data(self, X, y):
        X = _f32(X, self.device, name='X')
        if isinstance(self.spec, LeNetSpec) and X.ndim == 4:      # [N, C, H, W] images -> rows of C*H*W
            if tuple(X.shape[1:]) != (self.spec.channels, self.spec.height, self.spec.width):
                raise ValueError(f'X must be [N, {self.spec.channels}, {self.spec.height}, {self.spec.width}], got {tuple(X.shape)}')
            X = X.reshape(X.shape[0], -1)
        if X.ndim != 2 or X.shape[1] != self.spec.in_features:
            raise ValueError(f'X must be [N, {self.spec.in_features}], got {tuple(X.shape)}')
        y = torch.as_tensor(y, device=self.device)
        if y.ndim == 2 and y.shape[1] == 1:
            y = y[:, 0]
        if y.shape != (X.shape[0],):
            raise ValueError(f'y must be [N], got {tuple(y.shape)}')
        if self.spec.task == 'regr':
            y = y.to(torch.float32).contiguous()
        else:
            y = y.to(torch.int32).contiguous()
            n_classes = self.spec.hidden_structure[-1]
            if int(y.min()) < 0 or int(y.max()) >= n_classes:
                raise ValueError('class labels out of range')
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mile_set_data(self._h, _ptr(X), _ptr(y), X.shape[0], self._stream()), self.lib)
            torch.cuda.current_stream(self.device).synchronize()   # X, y may be temporaries
        if int(X.shape[0]) != getattr(self, 'N', None):
            self._E_reserved = 0            # (same N: the library keeps its buffers and workspace)
        self.N = int(X.shape[0])

    def set_row_window(self, begin: int = 0, count: int = 0):
        """Likelihood over rows [begin, begin + count) of the training set for the following logpost_grad calls
        (count = 0: all rows).  The minibatches of the warm-start stage."""
        _lib.check(self.lib.mile_set_row_window(self._h, int(begin), int(count)), self.lib)

    def reserve(self, E: int):
        # always asked: a SMALLER ensemble splits each particle's rows over more workgroups and may need more slab rows than
        # the larger one reserved (mile_reserve returns at once when the workspace already fits, and never shrinks it)
        _lib.check(self.lib.mile_reserve(self._h, int(E)), self.lib)
        self._E_reserved = max(self._E_reserved, int(E))

    def set_grad_kernel(self, name: str):
        _lib.check(self.lib.mile_set_grad_kernel(self._h, _lib.GRAD_KERNEL_IDS[name]), self.lib)

    @property
    def grad_kernel(self) -> str:
        k = self.lib.mile_get_grad_kernel(self._h)
        return {v: n for n, v in _lib.GRAD_KERNEL_IDS.items()}[k]

    def grad_launch_info(self, E: int) -> dict:
        gx, gy, blk, lds = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        name = C.create_string_buffer(64)
        _lib.check(self.lib.mile_grad_launch_info(self._h, E, C.byref(gx), C.byref(gy), C.byref(blk),
                                                  C.byref(lds), name, 64), self.lib)
        return {'kernel': name.value.decode(), 'grid': (gx.value, gy.value), 'block': blk.value,
                'lds_bytes': lds.value}

    def grad_timing_begin(self):
        _lib.check(self.lib.mile_grad_timing_begin(self._h), self.lib)

    def grad_timing_end(self):
        ms, n = C.c_float(), C.c_int32()
        _lib.check(self.lib.mile_grad_timing_end(self._h, C.byref(ms), C.byref(n)), self.lib)
        return float(ms.value), int(n.value)

    # ------------------------------------------------------------------
    def logpost_grad(self, theta):
        """jax.value_and_grad(logdensity_fn) for an ensemble: theta [E, d] -> (logp [E], grad [E, d])."""
        theta = _f32(theta, self.device, name='theta')
        if theta.ndim != 2 or theta.shape[1] != self.d:
            raise ValueError(f'theta must be [E, {self.d}], got {tuple(theta.shape)}')
        E = theta.shape[0]
        self.reserve(E)
        logp = torch.empty(E, dtype=torch.float32, device=self.device)
        grad = torch.empty_like(theta)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mile_logpost_grad(self._h, _ptr(theta), E, _ptr(logp), _ptr(grad),
                                                  self._stream()), self.lib)
        return logp, grad

    def warmstart_step(self, theta: torch.Tensor, optim: dict, active: torch.Tensor | None = None,
                       want_nll: bool = False) -> torch.Tensor | None:
        """One optimizer step of all members on the current row window (mile_warmstart_step): likelihood gradient from the
        grad kernel + ONE fused optimizer launch.  ``theta`` [E, d] and the moments ``optim['m']``, ``optim['v']`` are
        updated IN PLACE; ``optim`` also carries name, learning_rate, b1, b2, eps, weight_decay and the step count ``t``
        (incremented here).  ``active`` [E] uint8 / bool: 0 freezes a member (early stopping)."""
        if theta.dtype != torch.float32 or not theta.is_contiguous() or theta.device != self.device or theta.shape[1] != self.d:
            raise ValueError(f'theta must be a contiguous fp32 [E, {self.d}] tensor on the engine device')
        E = theta.shape[0]
        self.reserve(E)
        a = _lib.OptimArgsC()
        a.kind = _lib.OPTIMIZER_IDS[optim['name']]
        a.learning_rate, a.b1, a.b2 = optim['learning_rate'], optim['b1'], optim['b2']
        a.eps, a.weight_decay = optim['eps'], optim['weight_decay']
        optim['t'] += 1
        a.t = optim['t']
        if a.kind != 0:
            for k in ('m', 'v'):
                t = optim[k]
                if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != theta.shape or t.device != self.device:
                    raise ValueError('optimizer moments must match theta')
            a.m, a.v = optim['m'].data_ptr(), optim['v'].data_ptr()
        act = None
        if active is not None:
            act = active.to(device=self.device, dtype=torch.uint8).contiguous()
            a.active = act.data_ptr()
        nll = torch.empty(E, dtype=torch.float32, device=self.device) if want_nll else None
        a.out_nll = nll.data_ptr() if nll is not None else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mile_warmstart_step(self._h, _ptr(theta), E, C.byref(a), self._stream()), self.lib)
        return nll

    def _state_c(self, st: IntegratorState):
        sc = _lib.StateC()
        sc.n_particles = st.position.shape[0]
        sc.position = st.position.data_ptr()
        sc.momentum = st.momentum.data_ptr()
        sc.logdensity = st.logdensity.data_ptr()
        sc.logdensity_grad = st.logdensity_grad.data_ptr()
        return sc

    def _ids(self, particle_ids, E):
        if particle_ids is None:
            return None
        ids = torch.as_tensor(particle_ids, device=self.device).to(torch.int32).contiguous()
        if ids.shape != (E,):
            raise ValueError(f'particle_ids must be [{E}]')
        return ids

    def init(self, position, noise=None, seed: int = 0, particle_ids=None) -> IntegratorState:
        """blackjax.mcmc.mclmc.init for an ensemble.  ``noise`` [E, d] (explicit N(0,1) draws)
        or the counter RNG keyed by (seed, particle id)."""
        position = _f32(position, self.device, name='position').clone()
        if position.ndim != 2 or position.shape[1] != self.d:
            raise ValueError(f'position must be [E, {self.d}], got {tuple(position.shape)}')
        if self.d < 2:
            raise ValueError('The target distribution must have more than 1 dimension for MCLMC.')
        E = position.shape[0]
        self.reserve(E)
        st = IntegratorState(position, torch.empty_like(position),
                             torch.empty(E, dtype=torch.float32, device=self.device),
                             torch.empty_like(position))
        z = _f32(noise, self.device, (E, self.d), 'noise') if noise is not None else None
        ids = self._ids(particle_ids, E)
        sc = self._state_c(st)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mile_init(self._h, C.byref(sc), _ptr(z), C.c_uint64(seed), _ptr(ids),
                                          self._stream()), self.lib)
        return st

    def step(self, state: IntegratorState, step_size, L, n_steps: int = 1, *, noise=None, seed: int = 0,
             step_offset: int = 0, n_thinning: int = 0, particle_ids=None, refresh: str = 'O-step-O',
             sqrt_diag_cov=None, want_info: bool = True, inplace: bool = False):
        """n_steps kernel steps.  Returns (state, MCLMCInfo | None, samples [n_kept, E, d] | None).

        Pure by default (the input state is cloned, as blackjax's step is functional);
        ``inplace=True`` advances the given tensors.
        """
        E = state.position.shape[0]
        dev = self.device
        if not inplace:
            state = IntegratorState(*(t.clone() for t in state))
        for t in state:
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
                raise ValueError('state tensors must be contiguous fp32 on the engine device')
        self.reserve(E)
        eps = _f32(step_size, dev).expand(E).contiguous() if torch.as_tensor(step_size).ndim == 0 \
            else _f32(step_size, dev, (E,), 'step_size')
        Lt = _f32(L, dev).expand(E).contiguous() if torch.as_tensor(L).ndim == 0 else _f32(L, dev, (E,), 'L')
        z = _f32(noise, dev, (n_steps, 2, E, self.d), 'noise') if noise is not None else None
        sdc = _f32(sqrt_diag_cov, dev, (E, self.d), 'sqrt_diag_cov') if sqrt_diag_cov is not None else None
        ids = self._ids(particle_ids, E)
        n_kept = 0
        if n_thinning > 0:
            n_kept = sum(1 for i in range(n_steps) if (step_offset + i) % n_thinning == 0)
        samples = torch.empty((n_kept, E, self.d), dtype=torch.float32, device=dev) if n_kept else None
        info = torch.empty((n_steps, E, 3), dtype=torch.float32, device=dev) if want_info else None
        a = _lib.StepArgsC()
        a.step_size = eps.data_ptr()
        a.L = Lt.data_ptr()
        a.sqrt_diag_cov = sdc.data_ptr() if sdc is not None else None
        a.noise = z.data_ptr() if z is not None else None
        a.seed = seed
        a.particle_ids = ids.data_ptr() if ids is not None else None
        a.step_offset = step_offset
        a.n_steps = n_steps
        a.n_thinning = n_thinning
        a.refresh = _lib.REFRESH_IDS[refresh]
        a.out_samples = samples.data_ptr() if samples is not None else None
        a.out_info = info.data_ptr() if info is not None else None
        sc = self._state_c(state)
        with torch.cuda.device(dev):
            _lib.check(self.lib.mile_step(self._h, C.byref(sc), C.byref(a), self._stream()), self.lib)
        inf = MCLMCInfo(info[..., 0], info[..., 1], info[..., 2]) if info is not None else None
        return state, inf, samples

    def tune(self, state: IntegratorState, tuner: dict, L, n_steps: int, *, schedule_step0: int, n_mask_steps: int,
             schedule_total: int, desired_energy_var_start: float, desired_energy_var_end: float,
             trust_in_estimate: float, decay_rate: float, noise=None, seed: int = 0, step_offset: int = 0,
             particle_ids=None, refresh: str = 'O-step-O', sqrt_diag_cov=None, want_info: bool = False):
        """n_steps warm-up steps with on-device step-size adaptation (mile_tune).  ``state`` and the
        ``tuner`` tensors (step_size, step_size_max, time, x_average, stream_weight [E];
        stream_average [E, 2, d]) are advanced IN PLACE.  Returns MCLMCInfo or None."""
        E, dev = state.position.shape[0], self.device
        for t in tuple(state) + tuple(tuner.values()):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
                raise ValueError('state / tuner tensors must be contiguous fp32 on the engine device')
        if tuner['stream_average'].shape != (E, 2, self.d):
            raise ValueError('stream_average must be [E, 2, d]')
        self.reserve(E)
        Lt = _f32(L, dev).expand(E).contiguous() if torch.as_tensor(L).ndim == 0 else _f32(L, dev, (E,), 'L')
        z = _f32(noise, dev, (n_steps, 2, E, self.d), 'noise') if noise is not None else None
        sdc = _f32(sqrt_diag_cov, dev, (E, self.d), 'sqrt_diag_cov') if sqrt_diag_cov is not None else None
        ids = self._ids(particle_ids, E)
        info = torch.empty((n_steps, E, 3), dtype=torch.float32, device=dev) if want_info else None
        a = _lib.TuneArgsC()
        a.step_size = tuner['step_size'].data_ptr()
        a.L = Lt.data_ptr()
        a.sqrt_diag_cov = sdc.data_ptr() if sdc is not None else None
        a.step_size_max = tuner['step_size_max'].data_ptr()
        a.time = tuner['time'].data_ptr()
        a.x_average = tuner['x_average'].data_ptr()
        a.stream_weight = tuner['stream_weight'].data_ptr()
        a.stream_average = tuner['stream_average'].data_ptr()
        a.noise = z.data_ptr() if z is not None else None
        a.seed = seed
        a.particle_ids = ids.data_ptr() if ids is not None else None
        a.step_offset = step_offset
        a.n_steps = n_steps
        a.schedule_step0 = schedule_step0
        a.n_mask_steps = n_mask_steps
        a.schedule_total = schedule_total
        a.desired_energy_var_start = desired_energy_var_start
        a.desired_energy_var_end = desired_energy_var_end
        a.trust_in_estimate = trust_in_estimate
        a.decay_rate = decay_rate
        a.refresh = _lib.REFRESH_IDS[refresh]
        a.out_info = info.data_ptr() if info is not None else None
        sc = self._state_c(state)
        with torch.cuda.device(dev):
            _lib.check(self.lib.mile_tune(self._h, C.byref(sc), C.byref(a), self._stream()), self.lib)
        return MCLMCInfo(info[..., 0], info[..., 1], info[..., 2]) if info is not None else None

    def pointwise_loglik(self, theta, X, y) -> torch.Tensor:
        """log p(y_n | x_n, theta_s) for every sample and test row: theta [..., d] -> [..., N]
        (pointwise_lppd's input, src/inference/metrics.py:247-294), computed by the HIP forward kernels."""
        theta = _f32(theta, self.device, name='theta')
        lead = theta.shape[:-1]
        th = theta.reshape(-1, self.d).contiguous()
        X = _f32(X, self.device, name='X')
        y = torch.as_tensor(y, device=self.device)
        y = (y.to(torch.float32) if self.spec.task == 'regr' else y.to(torch.int32)).contiguous()
        if isinstance(self.spec, LeNetSpec) and X.ndim == 4:
            X = X.reshape(X.shape[0], -1).contiguous()
        if X.ndim != 2 or X.shape[1] != self.spec.in_features or y.shape != (X.shape[0],):
            raise ValueError('X must be [N, F] and y [N]')
        if self.spec.task != 'regr' and y.numel():       # the kernels index the logits with the raw label
            if int(y.min()) < 0 or int(y.max()) >= self.spec.hidden_structure[-1]:
                raise ValueError('class labels out of range')
        out = torch.empty((th.shape[0], X.shape[0]), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mile_pointwise_loglik(self._h, _ptr(th), th.shape[0], _ptr(X), _ptr(y), X.shape[0],
                                                      _ptr(out), self._stream()), self.lib)
        return out.reshape(*lead, X.shape[0])

    @property
    def supports_device_tuner(self) -> bool:
        return self.d >= 4

    def debug_noise(self, seed: int, E: int, step: int, stage: int, particle_ids=None):
        ids = self._ids(particle_ids, E)
        out = torch.empty((E, self.d), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mile_debug_noise(self._h, C.c_uint64(seed), _ptr(ids), E, step, stage, _ptr(out),
                                                 self._stream()), self.lib)
        return out
