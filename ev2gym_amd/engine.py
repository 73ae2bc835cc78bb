"""ctypes host binding of libev2g_hip.so (include/ev2g.h).  The thin layer between the Python surface
(`EV2GymVec`, `EV2Gym` facade) and the HIP kernels; there is no CPU fallback: importing works anywhere,
creating an engine without the built library or without a GPU raises.

Reference boundary being replaced: ev2gym.models.ev2gym_env.EV2Gym (reset :243-331, step :333-447).
"""
from __future__ import annotations

import ctypes as C
import os
import warnings
from typing import Optional

import numpy as np

from . import _abi
from .scenario import ScenarioBatch

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libev2g_hip.so")
_lib = None


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[ev2g {code}] {msg}")
        self.code = code


def load_library(path: Optional[str] = None):
    """dlopen libev2g_hip.so and declare the prototypes.  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("EV2G_LIB") or _LIB_PATH   # EV2G_LIB: A/B builds of the library (tools/ab_bench.py)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME).  If ours were loaded
    # first from /opt/rocm, torch would later bring up a second runtime and fail with "No HIP GPUs are available";
    # importing torch first makes the dynamic linker resolve our DT_NEEDED libamdhip64.so.7 to torch's copy.
    if os.environ.get("EV2G_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(path):
        raise EngineError(-2, f"{path} not found: build it with `python -m ev2gym_amd.build` "
                              "(hipcc --offload-arch=gfx950); the step engine has no CPU fallback")
    L = C.CDLL(path)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    protos = {
        "ev2g_abi_version": (C.c_int, []),
        "ev2g_create": (C.c_int, [C.POINTER(_abi.ConfigC), C.POINTER(vp)]),
        "ev2g_destroy": (None, [vp]),
        "ev2g_last_error": (C.c_char_p, [vp]),
        "ev2g_load_scenarios": (C.c_int, [vp, C.POINTER(_abi.ScenarioBatchC)]),
        "ev2g_n_envs": (C.c_int, [vp]), "ev2g_n_scenarios": (C.c_int, [vp]), "ev2g_n_ports": (C.c_int, [vp]), "ev2g_obs_dim": (C.c_int, [vp]),
        "ev2g_n_steps": (C.c_int, [vp]), "ev2g_current_step": (C.c_int, [vp]),
        "ev2g_reset": (C.c_int, [vp, vp]),
        "ev2g_reset_ex": (C.c_int, [vp, vp, i64]),
        "ev2g_scenario_offset": (i64, [vp]),
        "ev2g_set_step_extras": (C.c_int, [vp, C.POINTER(_abi.StepExtrasC)]),
        "ev2g_kernel_name": (C.c_char_p, [vp]),
        "ev2g_last_launch_specialisation": (C.c_int, [vp]),
        "ev2g_last_launch_general_reason": (C.c_char_p, [vp]),
        "ev2g_fallback_reason": (C.c_char_p, [vp]),
        "ev2g_big_kernel_reason": (C.c_char_p, [vp]),
        "ev2g_step": (C.c_int, [vp, vp, vp, vp, vp, vp]),
        "ev2g_step_n": (C.c_int, [vp, C.c_int, C.c_int, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, C.c_int]),
        "ev2g_check_faults": (C.c_int, [vp, C.POINTER(i32)]),
        "ev2g_get_stats": (C.c_int, [vp, vp]),
        "ev2g_get_stats_reset": (C.c_int, [vp, vp, vp, C.c_int64]),
        "ev2g_get_stats_reset_f32": (C.c_int, [vp, vp, vp, C.c_int64]),
        "ev2g_reset_f32": (C.c_int, [vp, vp, C.c_int64]),
        "ev2g_collect": (C.c_int, [vp, vp, C.c_int, vp]),
        "ev2g_stat_name": (C.c_char_p, [C.c_int]),
        "ev2g_peek": (C.c_int, [vp, C.c_int, C.POINTER(_abi.EnvViewC)]),
        "ev2g_malloc": (vp, [vp, C.c_size_t]),
        "ev2g_free": (None, [vp, vp]),
        "ev2g_memcpy_h2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
        "ev2g_memcpy_d2h": (C.c_int, [vp, vp, vp, C.c_size_t]),
        "ev2g_host_malloc": (vp, [vp, C.c_size_t]),
        "ev2g_host_free": (None, [vp, vp]),
        "ev2g_synchronize": (C.c_int, [vp]),
        "ev2g_fill_uniform": (C.c_int, [vp, vp, i64, C.c_uint64, dbl, dbl]),
        "ev2g_host_uniform": (None, [vp, i64, C.c_uint64, dbl, dbl]),
        "ev2g_last_step_n_kernel_ms": (dbl, [vp]),
        "ev2g_mlp_create": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.c_float, C.POINTER(vp)]),
        "ev2g_mlp_create_ex": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.c_float, C.c_int, C.POINTER(vp)]),
        "ev2g_mlp_destroy": (None, [vp, vp]),
        "ev2g_mlp_forward": (C.c_int, [vp, vp, vp, vp, C.c_int]),
        "ev2g_rollout": (C.c_int, [vp, vp, C.c_int, vp, i64, vp, i64, vp, i64, C.c_int]),
        "ev2g_rollout_graph_launches": (C.c_longlong, [vp]),
        "ev2g_comm_get_unique_id": (C.c_int, [vp]),
        "ev2g_comm_init": (C.c_int, [vp, vp, C.c_int, C.c_int]),
        "ev2g_comm_destroy": (None, [vp]),
        "ev2g_step_n_kernel_ms_back": (dbl, [vp, C.c_int]),
        "ev2g_comm_world_size": (C.c_int, [vp]),
        "ev2g_comm_gathers": (C.c_longlong, [vp]),
        "ev2g_gather_stats": (C.c_int, [vp, vp]),
        "ev2g_pool_refill": (C.c_int, [vp, C.POINTER(_abi.GenConfigC), C.c_uint64, i64, i32, i32]),
        "ev2g_pool_refill_overflows": (C.c_longlong, [vp]),
        "ev2g_pool_session_capacity": (C.c_int, [vp]),
        "ev2g_gen_default_config": (C.c_int, [C.c_int, C.POINTER(_abi.GenConfigC)]),
        "ev2g_generate": (C.c_int, [C.POINTER(_abi.GenConfigC), i32, C.c_uint64, i32, C.POINTER(vp)]),
        "ev2g_gen_batch": (C.POINTER(_abi.ScenarioBatchC), [vp]),
        "ev2g_gen_free": (None, [vp]),
        "ev2g_gen_table": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(L, name)  # AttributeError here = the .so does not export what include/ev2g.h declares
        fn.restype = res
        fn.argtypes = args
    if L.ev2g_abi_version() != _abi.ABI_VERSION:
        raise EngineError(-1, "libev2g_hip.so ABI version mismatch")
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    "ev2g_abi_version", "ev2g_create", "ev2g_destroy", "ev2g_last_error", "ev2g_load_scenarios", "ev2g_n_envs",
    "ev2g_n_scenarios", "ev2g_n_ports", "ev2g_obs_dim", "ev2g_n_steps", "ev2g_current_step", "ev2g_reset", "ev2g_reset_ex",
    "ev2g_scenario_offset", "ev2g_set_step_extras", "ev2g_kernel_name", "ev2g_last_launch_specialisation", "ev2g_last_launch_general_reason", "ev2g_fallback_reason", "ev2g_big_kernel_reason", "ev2g_step", "ev2g_step_n",
    "ev2g_check_faults", "ev2g_get_stats", "ev2g_get_stats_reset", "ev2g_get_stats_reset_f32", "ev2g_reset_f32", "ev2g_collect", "ev2g_stat_name", "ev2g_peek", "ev2g_malloc", "ev2g_free",
    "ev2g_memcpy_h2d", "ev2g_memcpy_d2h", "ev2g_host_malloc", "ev2g_host_free", "ev2g_synchronize", "ev2g_fill_uniform", "ev2g_host_uniform",
    "ev2g_last_step_n_kernel_ms", "ev2g_step_n_kernel_ms_back", "ev2g_mlp_create", "ev2g_mlp_create_ex", "ev2g_mlp_destroy", "ev2g_mlp_forward", "ev2g_rollout",
    "ev2g_rollout_graph_launches", "ev2g_comm_get_unique_id", "ev2g_comm_init", "ev2g_comm_destroy", "ev2g_comm_world_size", "ev2g_comm_gathers", "ev2g_gather_stats",
    "ev2g_pool_refill", "ev2g_pool_refill_overflows", "ev2g_pool_session_capacity", "ev2g_gen_default_config", "ev2g_generate", "ev2g_gen_batch", "ev2g_gen_free", "ev2g_gen_table"]


def _ptr(x):
    """Device address of a DeviceBuffer / torch tensor / raw int (None -> NULL)."""
    if x is None:
        return None
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if hasattr(x, "data_ptr"):
        return int(x.data_ptr())
    return int(x)


class DeviceBuffer:
    """A hipMalloc'd array owned by an engine handle (for hosts that do not use torch)."""

    def __init__(self, engine: "Engine", shape, dtype):
        self.engine = engine
        self.shape = tuple(int(s) for s in np.atleast_1d(shape))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = engine._lib.ev2g_malloc(engine._h, self.nbytes)
        if not self.ptr:
            raise EngineError(-2, engine.last_error())

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, self.dtype)
        assert arr.nbytes == self.nbytes, (arr.shape, self.shape)
        self.engine._check(self.engine._lib.ev2g_memcpy_h2d(self.engine._h, self.ptr, arr.ctypes.data, self.nbytes))
        return self

    def to_host(self, out=None):
        """Copy to the host: into a new array, or into `out` (same shape / dtype, C-contiguous) so that views of it stay valid."""
        if out is None:
            out = np.empty(self.shape, self.dtype)
        else:
            assert out.dtype == self.dtype and out.nbytes == self.nbytes and out.flags.c_contiguous
        self.engine._check(self.engine._lib.ev2g_memcpy_d2h(self.engine._h, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def at(self, index_elems: int):
        return self.ptr + int(index_elems) * self.dtype.itemsize

    def free(self):
        if self.ptr and self.engine._h:
            self.engine._lib.ev2g_free(self.engine._h, self.ptr)
        self.ptr = None


class Engine:
    """One handle = one GPU = one HIP stream; E envs resident in HBM."""

    def __init__(self, batch: ScenarioBatch, reward_kind: int, state_kind: int, device: int = 0, flags: int = 0,
                 stream: Optional[int] = None, cost_kind: int = 0, n_active_envs: int = 0):
        """`batch` is the resident scenario pool (M scenarios); `n_active_envs` (default: all of them) envs are stepped
        per call, each reset choosing which window of the pool they run (`reset(offset=...)`)."""
        self._lib = load_library()
        self._h = None
        self._spec_checks_left = 4   # launches still examined by _warn_if_general
        cfg = _abi.ConfigC(int(device), int(reward_kind), int(state_kind), int(flags), stream, int(cost_kind), int(n_active_envs))
        h = C.c_void_p()
        rc = self._lib.ev2g_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise EngineError(rc, (self._lib.ev2g_last_error(None) or b"").decode())
        self._h = h
        self.reward_kind, self.state_kind, self.flags, self.device = reward_kind, state_kind, flags, device
        self.load(batch)

    # ---- scenario ----------------------------------------------------------------------------
    def load(self, batch: ScenarioBatch):
        cb = batch.to_c()
        self._check(self._lib.ev2g_load_scenarios(self._h, C.byref(cb)))
        self.batch = batch
        self.E = self._lib.ev2g_n_envs(self._h)
        self.M = self._lib.ev2g_n_scenarios(self._h)
        self.P = self._lib.ev2g_n_ports(self._h)
        self.D = self._lib.ev2g_obs_dim(self._h)
        self.T = self._lib.ev2g_n_steps(self._h)
        self.C, self.R = batch.n_chargers, batch.n_transformers
        why = self.fallback_reason
        if why and self.P <= 64:   # a shape of the common size that did not get the fast-path kernel: say so, once per cause
            import warnings
            warnings.warn(f"ev2gym_amd: step kernel {self.kernel_name} selected instead of the fast path ({why})", stacklevel=2)

    @property
    def kernel_name(self) -> str:
        """The step kernel ev2g_load_scenarios selected for the loaded shape (routing is never silent)."""
        return (self._lib.ev2g_kernel_name(self._h) or b"").decode()

    @property
    def last_launch_specialisation(self) -> int:
        """0 general / 1 full / 2 full+wide instantiation of the fast-path kernel used by the last launch (-1: none, or another kernel)."""
        return int(self._lib.ev2g_last_launch_specialisation(self._h))

    @property
    def fallback_reason(self) -> str:
        return (self._lib.ev2g_fallback_reason(self._h) or b"").decode()

    @property
    def big_kernel_reason(self) -> str:
        """Why a big env (512 < ports <= 1024) does NOT get `ev2g_step_big` for its specialised launches ("" when it does / not a big env)."""
        return (self._lib.ev2g_big_kernel_reason(self._h) or b"").decode()

    @property
    def scenario_offset(self) -> int:
        return int(self._lib.ev2g_scenario_offset(self._h))

    def set_extras(self, cost=None, cost_stride=0, obs_f32=None, obs_f32_stride=0, actions_f32=None):
        """Optional sticky step outputs / inputs (include/ev2g.h ev2g_step_extras); all None clears them."""
        self._extras_keep = (cost, obs_f32, actions_f32)   # keep the buffers alive
        x = _abi.StepExtrasC(_ptr(cost), int(cost_stride), _ptr(obs_f32), int(obs_f32_stride), _ptr(actions_f32))
        self._check(self._lib.ev2g_set_step_extras(self._h, C.byref(x)))

    # ---- plumbing ------------------------------------------------------------------------------
    def last_error(self) -> str:
        return (self._lib.ev2g_last_error(self._h) or b"").decode()

    def _check(self, rc):
        if rc != 0:
            raise EngineError(rc, self.last_error())

    def empty(self, shape, dtype=np.float64) -> DeviceBuffer:
        return DeviceBuffer(self, shape, dtype)

    def pinned(self, shape, dtype=np.float64) -> np.ndarray:
        """A numpy array over page-locked host memory (ev2g_host_malloc): the destination / source of per-step copies (the SB3 VecEnv
        hand-over).  The memory belongs to the handle: the array must not be used after `close()`."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        p = self._lib.ev2g_host_malloc(self._h, max(n, 1))
        if not p:
            raise EngineError("ev2g_host_malloc failed")
        buf = (C.c_char * max(n, 1)).from_address(p)
        return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)

    def memcpy_d2h(self, host_array: np.ndarray, dev_ptr, nbytes: int):
        """One device -> host copy of `nbytes` from a raw device address into a (pinned or ordinary) C-contiguous host array."""
        assert host_array.flags.c_contiguous and host_array.nbytes >= nbytes
        self._check(self._lib.ev2g_memcpy_d2h(self._h, host_array.ctypes.data, _ptr(dev_ptr), int(nbytes)))

    def synchronize(self):
        self._check(self._lib.ev2g_synchronize(self._h))

    @property
    def current_step(self) -> int:
        return self._lib.ev2g_current_step(self._h)

    # ---- hot path ------------------------------------------------------------------------------
    def reset(self, obs=None, offset: Optional[int] = None):
        """Re-arm every env; `offset` first draws the scenarios of the coming episode: env e runs scenario
        (e + offset) mod M of the resident pool (None: the same scenarios as before)."""
        if offset is None:
            self._check(self._lib.ev2g_reset(self._h, _ptr(obs)))
        else:
            self._check(self._lib.ev2g_reset_ex(self._h, _ptr(obs), int(offset)))

    def _warn_if_general(self):
        """The fast path has a specialised instantiation (~20 % faster) that a launch gets only when it passes every output with step
        stride 0 and no extras (include/ev2g.h).  Falling off it is legal but silent: the first launches of an engine are checked and the
        caller is told, once, which argument did it (like the kernel-routing warning at load)."""
        if self._spec_checks_left <= 0:
            return
        self._spec_checks_left -= 1
        if self._lib.ev2g_last_launch_specialisation(self._h) == 0:
            why = (self._lib.ev2g_last_launch_general_reason(self._h) or b"").decode()
            # only for arguments a caller would change to get the fast instantiation back (a missing output, mismatched buffers, strides
            # a narrow env cannot specialise on): modes that are legitimate API use -- auto_reset (the gym / vec-env default), a registered
            # cost buffer, charger histories, a run-time reward -- are not performance mistakes and stay silent
            quiet = ("EV2G_NO_FULL", "auto_reset", "a cost buffer", "EV2G_FLAG_LOG_CS_HISTORY", "the reward function")
            if why and not why.startswith(quiet):
                self._spec_checks_left = 0
                warnings.warn(f"ev2gym_amd: this launch ran the GENERAL instantiation of {self.kernel_name} (slower than the full one): {why}", stacklevel=3)

    def step(self, actions, obs=None, reward=None, done=None, mask=None):
        self._check(self._lib.ev2g_step(self._h, _ptr(actions), _ptr(obs), _ptr(reward), _ptr(done), _ptr(mask)))
        if self._spec_checks_left > 0:
            self._warn_if_general()

    def step_n(self, k, actions, a_stride, obs=None, o_stride=0, reward=None, r_stride=0, done=None, d_stride=0,
               mask=None, m_stride=0, auto_reset=True, persistent=False):
        rc = self._lib.ev2g_step_n(self._h, int(k), 1 if persistent else 0, _ptr(actions), int(a_stride), _ptr(obs),
                                   int(o_stride), _ptr(reward), int(r_stride), _ptr(done), int(d_stride), _ptr(mask),
                                   int(m_stride), int(auto_reset))   # 0 / AUTO_RESET_SAME (True) / AUTO_RESET_NEXT
        self._check(rc)
        if self._spec_checks_left > 0:
            self._warn_if_general()

    # ---- policy in the loop ----------------------------------------------------------------------
    def mlp_create(self, W1, b1, W2, b2, W3, b3, out_lo=-1.0, precision="bf16"):
        """Three-layer actor (torch.nn.Linear layout: W[out,in], b[out]; host float32 arrays) evaluated by one fused kernel.
        precision: "bf16" (bf16 operands, fp32 accumulation: fastest), "fp32" (float32 weights as two bf16 terms, five products per
        k-step: within 1e-5 of a float64 forward -- what a float32-trained policy, e.g. SB3's, computes -- at twice the bf16 time) or
        "fp32x3" (three terms, all 24 bits: 1e-7 level, 2.6x the bf16 time); see include/ev2g.h."""
        arrs = [np.ascontiguousarray(a, np.float32) for a in (W1, b1, W2, b2, W3, b3)]
        h1, d_in = arrs[0].shape
        h2, d_out = arrs[2].shape[0], arrs[4].shape[0]
        assert arrs[2].shape == (h2, h1) and arrs[4].shape == (d_out, h2) and arrs[1].shape == (h1,) and arrs[3].shape == (h2,) and arrs[5].shape == (d_out,)
        m = C.c_void_p()
        prec = {"bf16": 0, "fp32": 1, "f32": 1, "fp32x3": 2, "f32x3": 2}[precision]
        self._check(self._lib.ev2g_mlp_create_ex(self._h, d_in, h1, h2, d_out, *[a.ctypes.data for a in arrs], float(out_lo), prec, C.byref(m)))
        return m

    def mlp_destroy(self, m):
        if self._h:
            self._lib.ev2g_mlp_destroy(self._h, m)

    def mlp_forward(self, m, x, y, n_rows):
        self._check(self._lib.ev2g_mlp_forward(self._h, m, _ptr(x), _ptr(y), int(n_rows)))

    def rollout(self, m, k, reward=None, r_stride=0, done=None, d_stride=0, mask=None, m_stride=0, auto_reset=0):
        """k x (actor forward on the registered float32 observation -> float32 actions -> env step), one C call."""
        self._check(self._lib.ev2g_rollout(self._h, m, int(k), _ptr(reward), int(r_stride), _ptr(done), int(d_stride), _ptr(mask),
                                           int(m_stride), int(auto_reset)))

    @property
    def rollout_graph_launches(self) -> int:
        return int(self._lib.ev2g_rollout_graph_launches(self._h))

    def last_step_n_kernel_ms(self) -> float:
        return float(self._lib.ev2g_last_step_n_kernel_ms(self._h))

    def step_n_kernel_ms_back(self, back: int) -> float:
        """HIP-event duration of the timed call `back` calls before the last one (the handle keeps 32): queue launches, read them afterwards."""
        return float(self._lib.ev2g_step_n_kernel_ms_back(self._h, int(back)))

    def check_faults(self):
        bad = C.c_int32(-1)
        rc = self._lib.ev2g_check_faults(self._h, C.byref(bad))
        if rc != 0:
            raise EngineError(rc, f"env {bad.value}: " + self.last_error())

    def fill_uniform(self, dst, n, seed, lo, hi):
        self._check(self._lib.ev2g_fill_uniform(self._h, _ptr(dst), int(n), int(seed), float(lo), float(hi)))

    # ---- statistics / inspection ---------------------------------------------------------------
    def stats(self, out=None) -> np.ndarray:
        """[E,17] get_statistics() scalars (utils.py:84-101) as a host array (or into a device `out`)."""
        if out is not None:
            self._check(self._lib.ev2g_get_stats(self._h, _ptr(out)))
            return out
        buf = self.empty((self.E, _abi.N_STATS))
        try:
            self._check(self._lib.ev2g_get_stats(self._h, buf.ptr))
            return buf.to_host()
        finally:
            buf.free()

    def collect(self, m, k, obs, actions, reward, done, mask):
        """k x (actor forward -> env step) with the transitions written straight into the caller's DEVICE arrays (ev2g_collect):
        obs float32 [k + 1, E, D] (row 0 is the input observation), actions float32 [k, E, P], reward float64 [k, E], done / mask uint8."""
        class _Tr(C.Structure):
            _fields_ = [("obs", C.c_void_p), ("actions", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p), ("mask", C.c_void_p)]
        tr = _Tr(_ptr(obs), _ptr(actions), _ptr(reward), _ptr(done), _ptr(mask))
        self._check(self._lib.ev2g_collect(self._h, m, int(k), C.byref(tr)))

    def reset_f32(self, obs32=None, offset: int = 0):
        self._check(self._lib.ev2g_reset_f32(self._h, _ptr(obs32), int(offset)))

    def stats_reset_f32(self, out, obs32=None, offset: int = 0):
        self._check(self._lib.ev2g_get_stats_reset_f32(self._h, _ptr(out), _ptr(obs32), int(offset)))
        return out

    def stats_reset(self, out, obs=None, offset: int = 0):
        """Episode end in one launch: get_statistics() of the finished episode into the device buffer `out`, then reset() onto the pool
        window `offset` (reset observation into `obs`) -- ev2g_get_stats_reset."""
        self._check(self._lib.ev2g_get_stats_reset(self._h, _ptr(out), _ptr(obs), int(offset)))
        return out

    # ---- multi-GPU statistics exchange over RCCL (include/ev2g.h: ev2g_comm_*, ev2g_gather_stats) ----
    @staticmethod
    def comm_unique_id() -> bytes:
        """Rank 0: the communicator id to hand to every rank's comm_init (128 bytes; ship them by any host-side channel)."""
        buf = C.create_string_buffer(_abi.COMM_ID_BYTES)
        rc = load_library().ev2g_comm_get_unique_id(C.cast(buf, C.c_void_p))
        if rc:
            raise EngineError(rc, (load_library().ev2g_last_error(None) or b"").decode())
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world_size: int):
        assert len(unique_id) == _abi.COMM_ID_BYTES
        buf = C.create_string_buffer(unique_id, _abi.COMM_ID_BYTES)
        self._check(self._lib.ev2g_comm_init(self._h, C.cast(buf, C.c_void_p), int(rank), int(world_size)))

    @property
    def comm_world_size(self) -> int:
        return int(self._lib.ev2g_comm_world_size(self._h))

    @property
    def comm_gathers(self) -> int:
        return int(self._lib.ev2g_comm_gathers(self._h))

    def gather_stats(self, out=None) -> np.ndarray:
        """Episode statistics of every rank, [world*E,17] rank-major: this rank's statistics kernel followed by ncclAllGather on the
        engine's stream.  Into the device array `out` (asynchronous), or returned as a host array."""
        if out is not None:
            self._check(self._lib.ev2g_gather_stats(self._h, _ptr(out)))
            return out
        buf = self.empty((self.comm_world_size * self.E, _abi.N_STATS))
        try:
            self._check(self._lib.ev2g_gather_stats(self._h, buf.ptr))
            return buf.to_host()
        finally:
            buf.free()

    # ---- scenario generation on the device ---------------------------------------------------------
    def pool_refill(self, gen_cfg, seed: int, first_index: int, first_slot: int, n: int):
        """Re-draw pool slots [first_slot, first_slot + n) ON THE DEVICE as scenarios first_index.. of the stream (gen_cfg, seed): bit for
        bit what `generate_native(gen_cfg with that seed)` yields at those indices (include/ev2g.h: ev2g_pool_refill).  The engine must have
        been created with FLAG_REFILLABLE from a batch drawn with the same config.  Asynchronous; refill slots no env is stepping."""
        from .scenario_gen import gen_config_c
        c, keep = gen_config_c(gen_cfg)
        self._check(self._lib.ev2g_pool_refill(self._h, C.byref(c), int(seed) & (2 ** 64 - 1), int(first_index), int(first_slot), int(n)))
        self.batch_is_stale = True
        del keep

    @property
    def pool_refill_overflows(self) -> int:
        return int(self._lib.ev2g_pool_refill_overflows(self._h))

    @property
    def pool_session_capacity(self) -> int:
        return int(self._lib.ev2g_pool_session_capacity(self._h))

    def peek(self, env: int = 0) -> dict:
        """Host copy of one env's state in the reference's port order (feeds the EV2Gym facade)."""
        P, Cn, R, T = self.P, self.C, self.R, self.T
        st = self.batch.arrays["env_session_start"]
        scn = (env + self.scenario_offset) % self.M   # the scenario this env is running
        S = int(st[scn + 1] - st[scn])
        f8 = lambda *s: np.empty(s, np.float64)  # noqa: E731
        i4 = lambda *s: np.empty(s, np.int32)  # noqa: E731
        d = dict(port_capacity=f8(P), port_energy=f8(P), port_current=f8(P), port_total_energy=f8(P),
                 port_required_energy=f8(P), port_prev_power=f8(P), port_cycles=i4(P), port_session=i4(P),
                 cs_power=f8(Cn), cs_amps=f8(Cn), cs_profits=f8(Cn), cs_energy_charged=f8(Cn),
                 cs_energy_discharged=f8(Cn), tr_power=f8(R), tr_overload=f8(R, T), power_usage=f8(T),
                 power_potential=f8(T), session_port=i4(max(S, 1)), session_afap=f8(max(S, 1)),
                 session_final_cap=f8(max(S, 1)))
        v = _abi.EnvViewC()
        for k, a in d.items():
            ct = C.c_double if a.dtype == np.float64 else C.c_int32
            setattr(v, k, a.ctypes.data_as(C.POINTER(ct)))
        self._check(self._lib.ev2g_peek(self._h, int(env), C.byref(v)))
        d["session_port"] = d["session_port"][:S]
        d["session_afap"] = d["session_afap"][:S]
        d["session_final_cap"] = d["session_final_cap"][:S]
        d["current_step"] = int(v.current_step)
        return d

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ev2g_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def host_uniform(n, seed, lo, hi) -> np.ndarray:
    """Host twin of Engine.fill_uniform (same counter-based generator)."""
    out = np.empty(int(n))
    load_library().ev2g_host_uniform(out.ctypes.data, int(n), int(seed), float(lo), float(hi))
    return out
