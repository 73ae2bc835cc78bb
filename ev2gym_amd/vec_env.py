"""EV2GymVec: thousands of EV2Gym envs advanced per call by the HIP engine.

Keeps the reference's surface (ev2gym/models/ev2gym_env.py: constructor kwargs :38-56, `reset()` :243-331,
`step(actions)` :333-447, `set_reward_function/set_cost_function`), batched over a leading env axis:

    env = EV2GymVec(config_file="V2GProfitPlusLoads.yaml", num_envs=4096, device=0,
                    state_function=V2G_profit_max_loads, reward_function=ProfitMax_TrPenalty_UserIncentives)
    obs, _ = env.reset()                       # [E, D]
    obs, reward, done, truncated, info = env.step(actions)   # actions [E, P] in [-1, 1]

Scenario draw: the reference draws a new scenario in every `reset()` (ev2gym_env.py:243-296).  Here a POOL of
`pool_factor x num_envs` scenarios is generated once and stays resident in HBM; every `reset()` picks, at no cost, which
window of the pool the envs run next (env e <- scenario (e + offset) mod M, offset drawn from the env's seeded generator;
`reset(seed=s)` maps s to an offset reproducibly), so consecutive episodes differ and auto-resetting rollouts never
replay a fixed set of `num_envs` episodes.  `resample_every=N` additionally re-draws the whole pool on the host every N
episodes.

Arrays are torch CUDA tensors when torch sees the GPU (zero-copy: the engine writes through `data_ptr()` on
torch's current stream -- the SB3 path), otherwise engine-owned device buffers mirrored to numpy.
Only the fused built-in state / reward functions run here; arbitrary Python callables need the single-env
facade (`ev2gym_amd.env.EV2Gym`), which says so instead of silently falling back.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import _abi
from .config import gen_config_from_yaml, load_yaml
from .engine import Engine, EngineError
from .scenario import ScenarioBatch
from .scenario_gen import GenConfig, generate


from .gym_compat import Box  # noqa: E402,F401  (gymnasium.spaces.Box when gymnasium is present)


def _kind(fn, table, what):
    """Resolve a state/reward plugin to a fused kernel id (by marker attribute, then by name)."""
    if fn is None:
        return None
    if isinstance(fn, str):
        if fn in table:
            return table[fn]
        raise ValueError(f"unknown {what} '{fn}'")
    k = getattr(fn, "_ev2g_kind", None)
    if k is not None:
        return int(k)
    name = getattr(fn, "__name__", "")
    if name in table and getattr(fn, "__module__", "").startswith("ev2gym"):
        return table[name]   # the reference's own built-in of that name
    return None


class EV2GymVec:
    def __init__(self, config_file=None, num_envs: int = 1, device: int = 0, state_function="PublicPST",
                 reward_function="SquaredTrackingErrorReward", cost_function=None, seed: Optional[int] = None,
                 scenarios: Optional[ScenarioBatch] = None, auto_reset: bool = False, log_cs_history: bool = False, log_soc: bool = True,
                 use_torch: Optional[bool] = None, rank: int = 0, world_size: int = 1, verbose: bool = False,
                 load_from_replay_path=None, pool_factor: int = 8, resample_every: Optional[int] = None, generator: str = "native", data_dir=None, device_refill: bool = False, sorted_pool: bool = True, **unused):
        self.state_kind = _kind(state_function, _abi.STATE_KINDS, "state_function")
        self.reward_kind = _kind(reward_function, _abi.REWARD_KINDS, "reward_function")
        if self.state_kind is None or self.reward_kind is None:
            raise NotImplementedError(
                "EV2GymVec fuses only the built-in state/reward functions "
                f"({sorted(_abi.STATE_KINDS)} / {sorted(_abi.REWARD_KINDS)}); "
                "user-defined callables run through the single-env facade ev2gym_amd.env.EV2Gym")
        self.cost_kind = 0
        if cost_function is not None:
            self.cost_kind = _kind(cost_function, _abi.COST_KINDS, "cost_function")
            if self.cost_kind is None:
                raise NotImplementedError(f"EV2GymVec fuses only the built-in cost functions ({sorted(_abi.COST_KINDS)}); "
                                          "user-defined callables run through the single-env facade ev2gym_amd.env.EV2Gym")
        # sorted_pool (default on): every E-sized block of the generated pool is ordered by busy window, so that the envs a workgroup advances in
        # lockstep are busy and idle together (-1.5 % / -4 % kernel time at cfg2 / cfg3).  Statistically a window is still an i.i.d. sample, but
        # env INDEX then correlates with arrival time inside a window; pass sorted_pool=False for per-env monitors that must not see that order.
        self.sorted_pool = bool(sorted_pool)
        self.seed = 0 if seed is None else int(seed)
        self._rng = np.random.default_rng(self.seed)     # the stream reset() draws scenario offsets from
        self.pool_factor = max(1, int(pool_factor))
        self.resample_every = resample_every
        if generator not in ("numpy", "native"):
            raise ValueError("generator: 'numpy' (scenario_gen.generate) or 'native' (the library's ev2g_generate)")
        self.generator = generator   # which implementation of the scenario model draws the pool (same model, different random streams)
        self._episodes = 0
        self._pool_generation = 0
        # device_refill: every window of the pool is re-drawn ON THE DEVICE (ev2g_pool_refill) right after the episode that used it, so
        # no scenario is ever stepped twice and no reset does host work -- the reference's fresh draw per reset (ev2gym_env.py:243-296)
        # at the GPU's pace.  The pool must come from the library's own generator (the device continues ITS stream of scenarios).
        self.device_refill = bool(device_refill)
        if self.device_refill:
            if config_file is None or scenarios is not None or load_from_replay_path is not None:
                raise ValueError("device_refill draws scenarios from a config: pass config_file (not scenarios / a replay file)")
            self.generator = generator = "native"
            if self.pool_factor < 2:
                raise ValueError("device_refill re-draws the window an episode used while the next episode runs on another one: it needs pool_factor >= 2")
        if scenarios is None and load_from_replay_path is not None:
            # one replay file, or a list of them recorded with the same config: one env per file (ev2gym_env.py:102-116)
            from .replay import load_replay
            paths = [load_from_replay_path] if isinstance(load_from_replay_path, (str, bytes, bytearray)) else list(load_from_replay_path)
            scenarios = ScenarioBatch.concat([load_replay(p) for p in paths]) if len(paths) > 1 else load_replay(paths[0])
        if scenarios is None:
            if config_file is None:
                raise AssertionError("Please provide a config file!!!")   # ev2gym_env.py:64
            self.config = load_yaml(config_file)
            if data_dir is not None:   # an EV2Gym install's ev2gym/data: its spawn tables / PV year / EV-spec files instead of the fitted stand-ins
                self.config = {**self.config, "data_dir": str(data_dir)}
            self._n_req, self._world = int(num_envs), int(world_size)
            scenarios = self._draw_pool(rank, world_size)
            n_active = int(num_envs)
        else:
            self.config = None
            n_active = min(int(num_envs), scenarios.n_envs) if num_envs and int(num_envs) > 1 else scenarios.n_envs
        self.scenarios = scenarios
        self.rank, self.world_size = rank, world_size
        if use_torch is None:
            try:
                import torch
                use_torch = torch.cuda.is_available()
            except Exception:
                use_torch = False
        self._torch = None
        stream = None
        if use_torch:
            import torch
            self._torch = torch
            torch.cuda.set_device(device)
            stream = torch.cuda.current_stream(device).cuda_stream or None
        flags = _abi.FLAG_LOG_CS_HISTORY if log_cs_history else 0
        if log_soc:
            flags |= _abi.FLAG_LOG_SOC   # battery-degradation statistics need the SoC log (ev.py:442-521)
        if use_torch and stream is None:
            flags |= _abi.FLAG_NULL_STREAM   # torch's current stream is the default stream: share it
        if self.device_refill:
            flags |= _abi.FLAG_REFILLABLE
        self.engine = Engine(scenarios, self.reward_kind, self.state_kind, device=device, flags=flags, stream=stream,
                             cost_kind=self.cost_kind, n_active_envs=n_active)
        e = self.engine
        self.num_envs, self.number_of_ports, self.obs_dim = e.E, e.P, e.D
        self.simulation_length = e.T
        self.v2g_enabled = scenarios.v2g_enabled
        self.auto_reset = auto_reset
        self.device = device
        low = -1.0 if self.v2g_enabled else 0.0
        self.action_space = Box(low, 1.0, (self.number_of_ports,))           # ev2gym_env.py:226-231
        self.observation_space = Box(-np.inf, np.inf, (self.obs_dim,))       # ev2gym_env.py:234-238
        self._obs = self._alloc((e.E, e.D))
        self._rew = self._alloc((e.E,))
        self._done = self._alloc((e.E,), np.uint8)
        self._mask = self._alloc((e.E, e.P), np.uint8)
        self._act = self._alloc((e.E, e.P))
        self._cost = None
        if self.cost_kind:
            self._cost = self._alloc((e.E,))
            e.set_extras(cost=self._cost)
        self.stats = None
        self._refill_next = e.M * max(1, world_size)   # next unused scenario index of this env's stream (every rank continues after the whole first pool)
        self._refill_stride = max(1, world_size)
        self._last_offset = None
        self.reset(seed=self.seed)

    def _refill_used_window(self):
        """device_refill: the window the finished episode ran on gets new scenarios (indices never used before: rank r of w takes
        the blocks r, r + w, ... of the stream), drawn on the device while the host goes on."""
        e = self.engine
        E, M = e.E, e.M
        off = self._last_offset
        if off is None or M < 2 * E:
            return
        cfg = gen_config_from_yaml(self.config, E, self._gen_seed)
        idx0 = self._refill_next + self.rank * E
        n1 = min(E, M - off)                       # the window may wrap around the end of the pool
        e.pool_refill(cfg, self._gen_seed, idx0, off, n1)
        if n1 < E:
            e.pool_refill(cfg, self._gen_seed, idx0 + n1, 0, E - n1)
        self._refill_next += E * self._refill_stride

    def _draw_pool(self, rank, world_size):
        """pool_factor x num_envs scenarios per rank from the vectorised generator (statistically matched to the
        reference's per-reset draw, scenario_gen.py); generation `g` of the pool uses the seed (seed, g)."""
        total = self._n_req * self.pool_factor * world_size
        gen_seed = self.seed if self._pool_generation == 0 else int(np.random.SeedSequence([self.seed, self._pool_generation]).generate_state(1)[0])
        draw = generate
        if self.generator == "native":
            from .scenario_gen import generate_native as draw
        self._gen_seed = gen_seed
        full = draw(gen_config_from_yaml(self.config, total, gen_seed))
        full = full.shard(rank, world_size) if world_size > 1 else full
        # (an i.i.d. pool has no meaningful order: scenarios with similar busy windows next to each other, so that the envs a workgroup
        # advances in lockstep are busy and idle together -- ScenarioBatch.sorted_by_busy_window; not with device_refill, whose re-drawn
        # windows must stay where the generator's stream puts them)
        # (the blocks are sorted INSIDE: episode windows are then aligned to them -- `_aligned` -- so that a window is one block, i.e. an i.i.d.
        # sample of scenarios; inside a window env index and busy window are correlated: per-env monitors see that order, the batch does not care)
        self._pool_sorted_blocks = self.sorted_pool and not self.device_refill
        return full.sorted_by_busy_window(self._n_req) if self._pool_sorted_blocks else full

    # ---- buffers ---------------------------------------------------------------------------------
    def _alloc(self, shape, dtype=np.float64):
        if self._torch is not None:
            t = self._torch
            td = {np.dtype(np.float64): t.float64, np.dtype(np.uint8): t.uint8}[np.dtype(dtype)]
            return t.empty(shape, dtype=td, device=f"cuda:{self.device}")
        return self.engine.empty(shape, dtype)

    def _out(self, buf):
        return buf if self._torch is not None else buf.to_host()

    def full_like_actions(self, value: float):
        if self._torch is not None:
            return self._torch.full((self.num_envs, self.number_of_ports), float(value), dtype=self._torch.float64,
                                    device=f"cuda:{self.device}")
        return np.full((self.num_envs, self.number_of_ports), float(value))

    def uniform_actions(self, seed: int, low: float, high: float):
        """Device-side uniform actions (counter-based, reproducible on the host with engine.host_uniform)."""
        self.engine.fill_uniform(self._act, self.num_envs * self.number_of_ports, seed, low, high)
        return self._act

    # ---- gym surface -------------------------------------------------------------------------------
    @property
    def current_step(self) -> int:
        return self.engine.current_step

    def reset(self, seed=None, options=None, **kwargs):
        """EV2Gym.reset() (ev2gym_env.py:243-331) for every env: draws the scenarios of the coming episode -- a window of the
        resident pool: with `seed` reproducibly (same seed, same episode), without it the next draw of the env's generator,
        so consecutive episodes differ -- and re-arms the state.  O(1) on the host, one small kernel on the device; with
        `resample_every=N` the whole pool is regenerated on the host every N episodes (0.16 s per 4096 x 50 scenarios)."""
        M = self.engine.M
        if (self.resample_every and self.config is not None and self._episodes
                and self._episodes % int(self.resample_every) == 0 and seed is None):
            self._pool_generation += 1
            self.scenarios = self._draw_pool(self.rank, self.world_size)
            self.engine.load(self.scenarios)
            self._window_queue = None
            self._last_offset = None   # names a window of the OLD pool: nothing of the new one has been used yet
            self._refill_next = self.engine.M * max(1, self.world_size)   # the new pool is a new stream of scenarios (its own seed)
        if seed is not None:
            offset = self._aligned(int(np.random.default_rng(int(seed)).integers(0, M)), M) if M > self.num_envs else 0
        else:
            offset = self._next_window(M)
        if self.device_refill and self._episodes > 0:
            self._refill_used_window()
        self._last_offset = offset
        self.engine.reset(self._obs, offset=offset)
        self._episodes += 1
        if not kwargs.get("_keep_stats"):
            self.stats = None
        return self._out(self._obs), {}

    def _aligned(self, offset, M):
        """A pool sorted by busy window inside blocks of num_envs scenarios (generated pools, `_draw_pool`): window offsets are multiples of the
        block size -- an unaligned window would be the late-arrival tail of one block plus the early-arrival head of the next, no longer an
        i.i.d. sample of scenarios."""
        E = self.num_envs
        if getattr(self, "_pool_sorted_blocks", False) and M >= 2 * E:
            return (offset // E) * E % (M // E * E)
        return offset

    def _next_window(self, M):
        """Scenario-pool windows WITHOUT replacement: a pass over the pool visits its M // E disjoint windows in a random order
        (from a random base offset), so no scenario is stepped twice before every other one has been; the next pass draws a new base
        and order.  (A uniformly random offset per episode made consecutive windows overlap.)"""
        E = self.num_envs
        if M <= E:
            return 0
        if not getattr(self, "_window_queue", None):
            base = self._aligned(int(self._rng.integers(0, M)), M)
            if getattr(self, "_pool_sorted_blocks", False) and M >= 2 * E:
                # block-aligned windows wrap modulo the aligned part of the pool: (base + k E) % M leaves the block grid whenever M is not
                # a multiple of E and a window would again straddle two sorted blocks
                Ma = M // E * E
                self._window_queue = [(base + int(k) * E) % Ma for k in self._rng.permutation(M // E)]
            else:
                self._window_queue = [(base + int(k) * E) % M for k in self._rng.permutation(M // E)]
        q = self._window_queue
        if len(q) > 1 and q[-1] == self._last_offset:   # (a new pass, or a seeded reset before: never the window the last episode ran on)
            q[-1], q[0] = q[0], q[-1]
        return q.pop()

    def _as_device_actions(self, actions):
        if self._torch is not None and self._torch.is_tensor(actions):
            a = actions
            if a.dtype != self._torch.float64 or not a.is_cuda or not a.is_contiguous():
                a = a.to(device=f"cuda:{self.device}", dtype=self._torch.float64).contiguous()  # widened on entry
            assert tuple(a.shape) == (self.num_envs, self.number_of_ports)
            return a
        if actions is self._act:
            return actions
        a = np.ascontiguousarray(actions, np.float64)   # contract: actions are widened to float64 on entry
        assert a.shape == (self.num_envs, self.number_of_ports), a.shape
        if self._torch is not None:
            self._act.copy_(self._torch.from_numpy(a))
        else:
            self._act.upload(a)
        return self._act

    def step(self, actions):
        if self.engine.current_step >= self.simulation_length:
            raise AssertionError("Episode is done, please reset the environment")   # ev2gym_env.py:343
        a = self._as_device_actions(actions)
        self.engine.step(a, self._obs, self._rew, self._done, self._mask)
        info = {"action_mask": self._out(self._mask), "cost": self._out(self._cost) if self._cost is not None else None}
        finished = self.engine.current_step >= self.simulation_length
        if finished:
            self.engine.check_faults()
            self.stats = self.get_statistics()
            info.update(self.stats)
            if self.auto_reset:
                info["terminal_observation"] = self._out(self._obs).clone() if self._torch is not None else self._out(self._obs)
                rew, done = self._out(self._rew), self._out(self._done)
                if self._torch is not None:
                    rew, done = rew.clone(), done.clone()
                if self._cost is not None and self._torch is not None:
                    info["cost"] = info["cost"].clone()
                self.reset(_keep_stats=True)   # the next episode's scenarios: a fresh window of the pool; env.stats stays the finished episode's, like the reference's
                return self._out(self._obs), rew, done, self._false(), info
        return self._out(self._obs), self._out(self._rew), self._out(self._done), self._false(), info

    def _false(self):
        """`truncated`: always False (the reference never truncates, ev2gym_env.py:476-500).  With torch ONE device tensor is created and
        returned every step -- a fresh `torch.zeros` per step was a kernel launch and 7 us of host time, half of the loop (treat it as
        read-only, like the observation / reward / done tensors, which are the engine's own output buffers)."""
        if self._torch is not None:
            t = getattr(self, "_trunc", None)
            if t is None:
                t = self._trunc = self._torch.zeros(self.num_envs, dtype=self._torch.bool, device=f"cuda:{self.device}")
            return t
        return np.zeros(self.num_envs, bool)

    def get_statistics(self) -> dict:
        """Per-env episode statistics, keys of get_statistics() (utilities/utils.py:84-101), each an [E] array."""
        st = self.engine.stats()
        out = {k: st[:, i] for i, k in enumerate(_abi.STAT_NAMES)}
        zero = st[:, 0] * 0
        out.update({k: zero for k in _abi.GRID_STAT_ZEROS})
        return out

    def get_statistics_all_ranks(self):
        """[world*E, 17] statistics of every rank (one RCCL all-gather per episode)."""
        from .dist import gather_stats
        return gather_stats(self)

    def set_reward_function(self, reward_function):
        raise NotImplementedError("the fused reward is chosen at construction (it selects the kernel specialisation)")

    def close(self):
        self.engine.close()
