"""Stable-Baselines3 `VecEnv` protocol over the batched engine (the SB3 rollout path of BASELINE configs[4]).

The reference trains SB3 agents on one `gym.make('EV2Gym-v1')` env at a time
(`/root/reference/train_stable_baselines.py:62-130`, README "Train RL Agents").  SB3's own scaling unit is the
`VecEnv`: `reset() -> obs[n,D]`, `step_async(actions[n,P])`, `step_wait() -> (obs, rewards, dones, infos)`,
envs that finish are reset inside `step_wait` and report `infos[i]["terminal_observation"]`.  This adapter gives
that protocol to `EV2GymVec`, so `DDPG("MlpPolicy", EV2GymSB3VecEnv(...))` collects rollouts from thousands of
envs per call.  All envs of a batch share the simulation length, so they all finish on the same step.

stable_baselines3 / gymnasium are optional: when importable the class derives from SB3's `VecEnv` and exposes
gymnasium `Box` spaces; otherwise it is a duck-typed equivalent (same methods) with the local `Box`.
"""
from __future__ import annotations

from typing import Any, List, Optional, Sequence

import numpy as np

from .vec_env import Box, EV2GymVec

try:  # optional
    from stable_baselines3.common.vec_env import VecEnv as _SB3VecEnv  # type: ignore
except Exception:  # pragma: no cover - not installed in this image
    _SB3VecEnv = object

try:  # optional
    from gymnasium import spaces as _gspaces  # type: ignore
except Exception:  # pragma: no cover
    _gspaces = None


def _space(box: Box, dtype):
    if _gspaces is not None:
        return _gspaces.Box(low=box.low.astype(dtype), high=box.high.astype(dtype), shape=box.shape, dtype=dtype)
    return Box(box.low, box.high, box.shape, dtype)


class EV2GymSB3VecEnv(_SB3VecEnv):
    """`venv = EV2GymSB3VecEnv(config_file=..., num_envs=4096, state_function=..., reward_function=...)`.

    Arrays cross to the host as numpy (SB3's buffers are numpy); `obs_dtype` float32 matches SB3's policies,
    the engine computes in float64.  Terminal infos carry the `get_statistics()` keys of that env
    (utils.py:84-101), `episode = {"r", "l"}` (what SB3's Monitor would add) and `terminal_observation`.
    """

    def __init__(self, vec: Optional[EV2GymVec] = None, obs_dtype=np.float32, copy_obs: bool = True, **vec_kwargs):
        if vec is None:
            vec_kwargs.setdefault("use_torch", False)
            vec = EV2GymVec(**vec_kwargs)
        if vec.auto_reset:
            raise ValueError("pass an EV2GymVec with auto_reset=False: this adapter resets at episode ends itself")
        self.vec = vec
        self.obs_dtype = np.dtype(obs_dtype)
        self.num_envs = vec.num_envs
        self.observation_space = _space(vec.observation_space, self.obs_dtype)
        self.action_space = _space(vec.action_space, np.float32)
        if _SB3VecEnv is not object:
            super().__init__(self.num_envs, self.observation_space, self.action_space)
        self.render_mode = None
        self.reset_infos: List[dict] = [{} for _ in range(self.num_envs)]
        self._actions = None
        # Fast path (engine-owned buffers, float32 observations): the policy-side types cross the boundary directly -- float32 actions
        # up, float32 observations down (the kernel widens / rounds once; no float64 copies or host conversions), and the per-env
        # info dicts of non-terminal steps are PERSISTENT: their "action_mask" entries are views of one host array refreshed in
        # place every step (consumers that keep infos across steps must copy them; terminal infos are fresh objects).
        self._fast = hasattr(vec, "engine") and getattr(vec, "_torch", 0) is None and self.obs_dtype == np.float32
        if self._fast:
            eng, E, P, D = vec.engine, vec.num_envs, vec.number_of_ports, vec.engine.D
            # Round 6: what a step hands down -- float32 observations, float64 rewards, done flags, action masks -- lives in ONE device block and comes
            # down in ONE copy into page-locked host memory (four copies into pageable arrays before: 0.135 of a step's 0.37 ms at 4096 x 50); the
            # actions go up from a page-locked staging array.
            a16 = lambda n: (n + 15) & ~15   # noqa: E731
            self._o_rew = a16(E * D * 4)
            self._o_done = self._o_rew + a16(E * 8)
            self._o_mask = self._o_done + a16(E)
            self._blk_bytes = self._o_mask + a16(E * P)
            self._d_blk = eng.empty((self._blk_bytes,), np.uint8)
            self._d_obs32, self._d_act32 = self._d_blk.ptr, eng.empty((E, P), np.float32)
            self._d_rew, self._d_done, self._d_mask = self._d_blk.ptr + self._o_rew, self._d_blk.ptr + self._o_done, self._d_blk.ptr + self._o_mask
            eng.set_extras(cost=vec._cost, obs_f32=self._d_obs32, obs_f32_stride=0, actions_f32=self._d_act32)
            # copy_obs=False (opt-in): the observations returned by reset() / step() are VIEWS of two page-locked blocks used in turn -- an array stays valid
            # until the second next step() (SB3's own loops keep the last observation only; anything that collects the returned arrays must copy them).
            # The fresh array per step that the default returns is a third of the adapter's step time at 4096 x 50.
            self._copy_obs = bool(copy_obs)
            self._blocks = []
            for _ in range(1 if self._copy_obs else 2):
                blk = eng.pinned((self._blk_bytes,), np.uint8)
                blk[:] = 0
                mask = np.zeros((E, P), np.uint8)   # (an ordinary array, refreshed from the block every step: the info dicts users may keep must not point into memory that goes with the handle)
                self._blocks.append(dict(blk=blk, obs=blk[:E * D * 4].view(np.float32).reshape(E, D), rew=blk[self._o_rew:self._o_rew + E * 8].view(np.float64),
                                         done=blk[self._o_done:self._o_done + E], mask_src=blk[self._o_mask:self._o_mask + E * P].reshape(E, P), mask=mask,
                                         infos=[{"action_mask": mask[i]} for i in range(E)]))
            self._cur = 0
            self._use_block(0)
            self._h_act = eng.pinned((E, P), np.float32)
        self._ep_return = np.zeros(self.num_envs)
        self._seeds = [None] * self.num_envs
        self._options = [{} for _ in range(self.num_envs)]

    def _use_block(self, k):
        b = self._blocks[k]
        self._cur = k
        self._h_blk, self._h_obs, self._h_rew, self._h_done, self._h_mask, self._infos = b["blk"], b["obs"], b["rew"], b["done"], b["mask"], b["infos"]
        self._h_mask_src = b["mask_src"]

    # ---- VecEnv protocol ---------------------------------------------------------------------------
    def _host(self, x):
        return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)

    def reset(self):
        obs, _ = self.vec.reset()
        self._ep_return[:] = 0.0
        self.reset_infos = [{} for _ in range(self.num_envs)]
        if self._fast:
            self._use_block((self._cur + 1) % len(self._blocks))
            self.vec.engine.memcpy_d2h(self._h_blk, self._d_obs32, self._o_rew)   # (the reset observation: the block's first segment)
            return self._h_obs.copy() if self._copy_obs else self._h_obs
        return self._host(obs).astype(self.obs_dtype, copy=False)

    def step_async(self, actions) -> None:
        self._actions = np.ascontiguousarray(actions, np.float32 if self._fast else np.float64)

    def _step_wait_fast(self):
        vec, eng = self.vec, self.vec.engine
        if eng.current_step >= vec.simulation_length:
            raise AssertionError("Episode is done, please reset the environment")   # ev2gym_env.py:343
        assert self._actions.shape == (self.num_envs, vec.number_of_ports), self._actions.shape
        np.copyto(self._h_act, self._actions)
        self._d_act32.upload(self._h_act)
        eng.step(None, None, self._d_rew, self._d_done, self._d_mask)     # float32 actions in, float32 observations out (the extras)
        self._use_block((self._cur + 1) % len(self._blocks))
        eng.memcpy_d2h(self._h_blk, self._d_blk, self._blk_bytes)        # observations | rewards | dones | masks: one copy
        np.copyto(self._h_mask, self._h_mask_src)
        obs, rew = self._h_obs, self._h_rew
        done = self._h_done.astype(bool)
        self._ep_return += rew
        infos = self._infos
        if eng.current_step >= vec.simulation_length:
            eng.check_faults()
            vec.stats = stats = vec.get_statistics()
            T = vec.simulation_length
            term, mask = obs.copy(), self._h_mask.copy()
            infos = []
            for i in range(self.num_envs):
                d = {k: float(v[i]) for k, v in stats.items()}
                d.update({"action_mask": mask[i], "terminal_observation": term[i], "TimeLimit.truncated": False,
                          "episode": {"r": float(self._ep_return[i]), "l": T}})
                infos.append(d)
            vec.reset(_keep_stats=True)
            eng.memcpy_d2h(self._h_blk, self._d_obs32, self._o_rew)   # the next episode's reset observation (rew / done / mask of the terminal step stay)
            self._ep_return[:] = 0.0
        return (obs.copy() if self._copy_obs else obs), rew.astype(np.float32), done, infos

    def step_wait(self):
        if self._fast:
            return self._step_wait_fast()
        obs, rew, done, _trunc, info = self.vec.step(self._actions)
        rew = self._host(rew).astype(np.float64)
        done = self._host(done).astype(bool)
        obs = self._host(obs)
        self._ep_return += rew
        mask = self._host(info["action_mask"])
        infos = [{"action_mask": mask[i]} for i in range(self.num_envs)]
        if done.all():
            stats = {k: self._host(v) for k, v in info.items() if k not in ("action_mask", "cost")}
            term = obs.astype(self.obs_dtype)
            T = self.vec.simulation_length
            for i, d in enumerate(infos):
                d.update({k: float(v[i]) for k, v in stats.items()})
                d["terminal_observation"] = term[i]
                d["TimeLimit.truncated"] = False
                d["episode"] = {"r": float(self._ep_return[i]), "l": T}
            obs = self._host(self.vec.reset()[0])
            self._ep_return[:] = 0.0
        return obs.astype(self.obs_dtype, copy=False), rew.astype(np.float32), done, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self) -> None:
        self.vec.close()

    def seed(self, seed: Optional[int] = None) -> Sequence[Optional[int]]:
        """SB3 seeds its envs once before training: re-seed the generator the per-episode scenario draws come from and
        start from the window of the pool that `seed` selects (EV2GymVec.reset(seed=...))."""
        if seed is not None:
            self.vec._rng = np.random.default_rng(int(seed))
            self.vec.reset(seed=int(seed))
        return [None if seed is None else seed + i for i in range(self.num_envs)]

    def _indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        return [indices] if isinstance(indices, int) else indices

    def get_attr(self, attr_name: str, indices=None) -> List[Any]:
        return [getattr(self.vec, attr_name) for _ in self._indices(indices)]

    def set_attr(self, attr_name: str, value: Any, indices=None) -> None:
        raise NotImplementedError("engine state is device-resident; per-env attributes are read-only")

    def env_method(self, method_name: str, *args, indices=None, **kwargs) -> List[Any]:
        raise NotImplementedError("no per-env Python objects exist behind the batched engine")

    def env_is_wrapped(self, wrapper_class, indices=None) -> List[bool]:
        return [False for _ in self._indices(indices)]

    def get_images(self):
        return [None] * self.num_envs

    def render(self, mode: Optional[str] = None):
        return None


class DeviceReplayCollector:
    """Off-policy rollout collection that never leaves the device: the `collect_rollouts()` half of an SB3 DDPG / TD3 / SAC loop
    (`/root/reference/train_stable_baselines.py:62-130`) for thousands of envs.

    SB3's `VecEnv` protocol (the adapter above) moves numpy arrays every step -- 3.6 MB per step at 4096 x 50 -- and its replay buffer lives on
    the host.  Here the replay buffer is a ring of EPISODE BLOCKS in device memory and `ev2g_collect` fills a block in place: the fused actor
    reads observation row t and writes action row t, the step kernel reads that action row and writes observation row t + 1, reward, done and
    action-mask row t.  `next_obs[t] = obs[t + 1]`, so a block IS the (obs, action, reward, next_obs, done) sequence of its episode (SB3's
    `ReplayBuffer(optimize_memory_usage=True)` layout); its row T is every env's `terminal_observation`, and the reset observation of the
    next episode lands in row 0 of the next block (`ev2g_get_stats_reset_f32`, which also yields the terminal `info` statistics).

        col = DeviceReplayCollector(engine, actor_weights, lo=-1.0, capacity_episodes=4)
        col.collect_episode()                       # one episode of all E envs: T x (actor forward -> env step), no host copies
        obs, act, rew, nxt, done = col.sample(256)  # host-side draw of indices, device gathers (torch) -- what a learner consumes
        col.set_weights(new_weights)                # the learner's updated actor

    `engine` must not have float32 hand-over buffers registered (`ev2g_set_step_extras`): the collector's rows are the hand-over.
    Arrays are torch tensors when torch is importable (PyTorch-ROCm is the SB3 side's tensor type), DeviceBuffers otherwise.

    Stream order: the engine works on ITS OWN stream (non-blocking unless it was created on torch's), the torch side on torch's current
    stream.  The blocks are allocated uninitialised and torch's stream is drained once before the engine first writes them (nothing torch
    queued can land behind the reset observation); `sample()` / `terminal_observation()` wait for the engine's stream before torch reads."""

    def __init__(self, engine, weights, lo: float, capacity_episodes: int = 2, precision: str = "bf16", offset_stride: Optional[int] = None, use_torch: Optional[bool] = None):
        from . import _abi
        self.eng = eng = engine
        self.E, self.P, self.D, self.T, self.M = eng.E, eng.P, eng.D, eng.T, eng.M
        self.lo, self.precision = float(lo), precision
        self.mlp = eng.mlp_create(*weights, out_lo=self.lo, precision=precision)
        self.cap = max(2, int(capacity_episodes))   # (the reset observation of the next episode is written into the NEXT block)
        if use_torch is None:
            try:
                import torch
                use_torch = torch.cuda.is_available()
            except Exception:
                use_torch = False
        self._torch = None
        if use_torch:
            import torch
            self._torch = torch

        def alloc(shape, dtype):
            if self._torch is not None:
                td = {np.float32: self._torch.float32, np.float64: self._torch.float64, np.uint8: self._torch.uint8}[dtype]
                return self._torch.empty(shape, dtype=td, device=f"cuda:{eng.device if hasattr(eng, 'device') else 0}")
            return eng.empty(shape, dtype)
        E, P, D, T = self.E, self.P, self.D, self.T
        self.obs = [alloc((T + 1, E, D), np.float32) for _ in range(self.cap)]
        self.actions = [alloc((T, E, P), np.float32) for _ in range(self.cap)]
        self.reward = [alloc((T, E), np.float64) for _ in range(self.cap)]
        self.done = [alloc((T, E), np.uint8) for _ in range(self.cap)]
        self.mask = [alloc((T, E, P), np.uint8) for _ in range(self.cap)]
        self.stats = alloc((E, _abi.N_STATS), np.float64)   # get_statistics() of the episode collected last (terminal info)
        self.head = 0            # block the next episode goes into
        self.filled = 0          # complete episodes in the ring
        self.offset = 0
        self.offset_stride = self.E if offset_stride is None else int(offset_stride)
        self.episodes = 0
        if self._torch is not None:
            self._torch.cuda.synchronize()   # the caching allocator may hand out memory with work of torch's stream still queued on it
        eng.reset_f32(self.obs[0], self.offset)

    def set_weights(self, weights):
        """The learner's new actor (host float32 arrays, torch.nn.Linear layout)."""
        old = self.mlp
        self.mlp = self.eng.mlp_create(*weights, out_lo=self.lo, precision=self.precision)
        self.eng.mlp_destroy(old)

    def collect_episode(self):
        """One whole episode of every env into the head block; statistics of the finished episode; reset onto the next pool window with the
        reset observation in row 0 of the next block.  Returns the index of the block that was filled."""
        b, eng = self.head, self.eng
        eng.collect(self.mlp, self.T - eng.current_step, self.obs[b], self.actions[b], self.reward[b], self.done[b], self.mask[b])
        nb = (b + 1) % self.cap
        self.offset = (self.offset + self.offset_stride) % self.M
        eng.stats_reset_f32(self.stats, self.obs[nb], self.offset)
        self.head = nb
        self.filled = min(self.filled + 1, self.cap - 1)
        self.episodes += 1
        return b

    def terminal_observation(self, block):
        """[E, D] float32: the observation after the block's last step (SB3's infos[i]["terminal_observation"]) -- a device view of the block's
        last row with torch, a host copy of it otherwise."""
        if self._torch is not None:
            self.eng.synchronize()   # (the engine's stream wrote the row; torch's stream reads it)
            return self.obs[block][self.T]
        return self.obs[block].to_host()[self.T]

    def sample(self, batch_size: int, rng: Optional[np.random.Generator] = None):
        """Uniform transitions from the complete blocks as device tensors (torch only): (obs, action, reward, next_obs, done)."""
        if self._torch is None:
            raise RuntimeError("sample() gathers with torch; without it read the blocks directly (col.obs[b], col.actions[b], ...)")
        torch = self._torch
        self.eng.synchronize()   # collect_episode() is asynchronous on the engine's stream; the gathers below run on torch's
        rng = rng or np.random.default_rng()
        blocks = [(self.head - 1 - i) % self.cap for i in range(self.filled)]
        bi = rng.integers(0, len(blocks), batch_size)
        t = torch.as_tensor(rng.integers(0, self.T, batch_size), device=self.stats.device)
        e = torch.as_tensor(rng.integers(0, self.E, batch_size), device=self.stats.device)
        out = [[], [], [], [], []]
        for j, b in enumerate(blocks):
            sel = torch.as_tensor(np.nonzero(bi == j)[0], device=self.stats.device)
            if sel.numel() == 0:
                continue
            tt, ee = t[sel], e[sel]
            out[0].append(self.obs[b][tt, ee]); out[1].append(self.actions[b][tt, ee]); out[2].append(self.reward[b][tt, ee])
            out[3].append(self.obs[b][tt + 1, ee]); out[4].append(self.done[b][tt, ee])
        return tuple(torch.cat(x) for x in out)

    def close(self):
        self.eng.mlp_destroy(self.mlp)
