"""ctypes binding of the CPU oracle (oracle/ev2g_oracle.c).  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by
anything under ev2gym_amd/ (the product path fails loudly without the HIP extension instead).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libev2g_oracle.so")


def build(force=False):
    src = os.path.join(HERE, "ev2g_oracle.c")
    hdr = os.path.join(HERE, "..", "include", "ev2g.h")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", HERE, "-B", "libev2g_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.ev2g_oracle_create.restype = C.c_void_p
        L.ev2g_oracle_create.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ev2g_oracle_obs_dim.argtypes = [C.c_void_p]
        L.ev2g_oracle_reset.argtypes = [C.c_void_p, C.c_void_p]
        L.ev2g_oracle_step.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.ev2g_oracle_step_range.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
        L.ev2g_oracle_run_range.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_longlong] + [C.c_void_p] * 4
        L.ev2g_oracle_peek.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 21
        L.ev2g_oracle_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.ev2g_oracle_destroy.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Batched CPU oracle with the same reset()/step() shape as the HIP engine."""

    def __init__(self, batch, reward_kind, state_kind):
        self.batch = batch
        self._c = batch.to_c()
        self.h = lib().ev2g_oracle_create(C.byref(self._c), int(reward_kind), int(state_kind))
        self.E, self.P, self.T = batch.n_envs, batch.n_ports, batch.n_steps
        self.C, self.R = batch.n_chargers, batch.n_transformers
        self.D = lib().ev2g_oracle_obs_dim(self.h)

    def reset(self):
        obs = np.empty((self.E, self.D))
        lib().ev2g_oracle_reset(self.h, _p(obs))
        return obs

    def step(self, actions):
        """actions [E,P] float64; mutated in place (empty ports zeroed) like the reference."""
        assert actions.dtype == np.float64 and actions.flags.c_contiguous and actions.shape == (self.E, self.P)
        obs = np.empty((self.E, self.D))
        rew = np.empty(self.E)
        done = np.empty(self.E, np.uint8)
        mask = np.empty((self.E, self.P), np.uint8)
        rc = lib().ev2g_oracle_step(self.h, _p(actions), _p(obs), _p(rew), _p(done), _p(mask))
        return obs, rew, done, mask, rc

    def run_range_nocopy(self, e0, e1, k, actions, a_stride, obs, rew, done, mask):
        """k steps of envs [e0,e1) in one C call (actions[step] at actions + step*a_stride doubles)."""
        return lib().ev2g_oracle_run_range(self.h, e0, e1, k, _p(actions), a_stride, _p(obs), _p(rew), _p(done), _p(mask))

    def step_range_nocopy(self, e0, e1, actions, obs, rew, done, mask):
        return lib().ev2g_oracle_step_range(self.h, e0, e1, _p(actions), _p(obs), _p(rew), _p(done), _p(mask))

    def peek(self, e=0):
        P, Cn, R, T = self.P, self.C, self.R, self.T
        st = self.batch.arrays["env_session_start"]
        S = int(st[e + 1] - st[e])
        d = dict(cap=np.empty(P), energy=np.empty(P), current=np.empty(P), tot_e=np.empty(P), req_e=np.empty(P),
                 prev_power=np.empty(P), cycles=np.empty(P, np.int32), session=np.empty(P, np.int32),
                 cs_power=np.empty(Cn), cs_amps=np.empty(Cn), cs_profits=np.empty(Cn), cs_e_ch=np.empty(Cn),
                 cs_e_dis=np.empty(Cn), tr_power=np.empty(R), tr_amps=np.empty(R), tr_overload=np.empty((R, T)),
                 usage=np.empty(T), potential=np.empty(T), session_port=np.empty(S, np.int32),
                 session_afap=np.empty(S), session_cap=np.empty(S))
        lib().ev2g_oracle_peek(self.h, e, *[_p(v) for v in d.values()])
        return d

    def stats(self):
        out = np.empty((self.E, 17))
        lib().ev2g_oracle_stats(self.h, _p(out))
        return out

    def close(self):
        if self.h:
            lib().ev2g_oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
