"""Stand-in for ev2gym_amd.engine.Engine on a box without a GPU (tests only): the calls bench.py makes, served by the CPU
oracle on the current window of the scenario pool.  `bench.py --backend gloo --engine tests.bench_stub_engine:StubEngine` runs
the benchmark's launcher and its whole multi-rank control flow here; the line it prints is marked as a stub run."""
import time

import numpy as np

from ev2gym_amd import _abi
from ev2gym_amd.engine import host_uniform
from oracle.oracle import Oracle


class StubEngine:
    kernel_name = "cpu-oracle-stub"

    def __init__(self, pool, rk, sk, device=0, stream=None, flags=0, n_active_envs=0):
        self.pool, self.rk, self.sk = pool, rk, sk
        self.M = pool.n_envs
        self.E = n_active_envs or self.M
        self.T, self.P = pool.n_steps, pool.n_ports
        self.ora = Oracle(pool.select(np.arange(self.E)), rk, sk)
        self.D = self.ora.D
        self.current_step, self._ms, self.offsets = 0, 0.0, []

    def fill_uniform(self, dst, n, seed, lo, hi):
        dst.view(-1)[:n] = __import__("torch").from_numpy(host_uniform(n, seed, lo, hi))

    def reset(self, obs=None, offset=None):
        if offset is not None:
            self.offsets.append(offset)
            self.ora.close()
            self.ora = Oracle(self.pool.select((np.arange(self.E) + offset) % self.M), self.rk, self.sk)
        self.ora.reset()
        self.current_step = 0

    def step_n(self, k, acts, a_stride, obs=None, o_stride=0, rew=None, r_stride=0, done=None, d_stride=0, mask=None, m_stride=0,
               auto_reset=False, persistent=False):
        t0 = time.perf_counter()
        torch = __import__("torch")   # bench passes acts[t]: to the engine an address, here the storage from that offset on
        a = torch.empty(0, dtype=acts.dtype).set_(acts.untyped_storage()).numpy()[acts.storage_offset():]
        for i in range(k):
            self.ora.step(a[i * a_stride:(i + 1) * a_stride].reshape(self.E, self.P).copy())
            self.current_step += 1
        self._ms = max((time.perf_counter() - t0) * 1e3, 1e-6)

    def last_step_n_kernel_ms(self):
        return self._ms

    def stats(self, out=None):
        st = np.nan_to_num(self.ora.stats())
        if out is None:
            return st
        out.copy_(__import__("torch").from_numpy(st))
        return out

    def check_faults(self):
        pass

    def synchronize(self):
        pass

    def close(self):
        self.ora.close()
