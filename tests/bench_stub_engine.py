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

    # ---- the C-ABI's own statistics exchange (ev2g_comm_* / ev2g_gather_stats, csrc/ev2g_host.hip), restated over the process group the test
    #      runs on: the same protocol -- one int per rank first, unequal shards refused on EVERY rank before any statistics row moves ----
    @staticmethod
    def comm_unique_id() -> bytes:
        return bytes(range(128))

    def comm_init(self, unique_id, rank, world_size):
        assert len(unique_id) == _abi.COMM_ID_BYTES
        self._comm = (int(rank), int(world_size))
        self._comm_checked = -1
        self.comm_gathers = 0

    @property
    def comm_world_size(self):
        return getattr(self, "_comm", (0, 0))[1]

    def gather_stats(self, out=None):
        import torch
        import torch.distributed as dist
        if not self.comm_world_size:
            raise RuntimeError("ev2g_gather_stats: no communicator (call ev2g_comm_init on every rank first)")
        world = self.comm_world_size
        if self._comm_checked != self.E:
            mine = torch.tensor([self.E], dtype=torch.int32)
            counts = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(counts, mine)
            for r, c in enumerate(counts):
                if int(c.item()) != self.E:
                    raise RuntimeError(f"ev2g_gather_stats: rank {r} steps {int(c.item())} envs, this rank {self.E} "
                                       "(the gather needs equal shards; pad the batch or use torch's uneven gather)")
            self._comm_checked = self.E
        st = torch.from_numpy(np.nan_to_num(self.ora.stats()))
        res = out if out is not None else torch.empty((world * self.E, _abi.N_STATS), dtype=torch.float64)
        dist.all_gather_into_tensor(res, st.contiguous())
        self.comm_gathers += 1
        return res

    def synchronize(self):
        pass

    def close(self):
        self.ora.close()
