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

    def __init__(self, vec: Optional[EV2GymVec] = None, obs_dtype=np.float32, **vec_kwargs):
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
            self._d_obs32, self._d_act32 = eng.empty((E, D), np.float32), eng.empty((E, P), np.float32)
            eng.set_extras(cost=vec._cost, obs_f32=self._d_obs32, obs_f32_stride=0, actions_f32=self._d_act32)
            self._h_obs, self._h_mask = np.empty((E, D), np.float32), np.zeros((E, P), np.uint8)
            self._h_rew, self._h_done = np.empty(E, np.float64), np.empty(E, np.uint8)
            self._infos = [{"action_mask": self._h_mask[i]} for i in range(E)]
        self._ep_return = np.zeros(self.num_envs)
        self._seeds = [None] * self.num_envs
        self._options = [{} for _ in range(self.num_envs)]

    # ---- VecEnv protocol ---------------------------------------------------------------------------
    def _host(self, x):
        return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)

    def reset(self):
        obs, _ = self.vec.reset()
        self._ep_return[:] = 0.0
        self.reset_infos = [{} for _ in range(self.num_envs)]
        if self._fast:
            return self._d_obs32.to_host().astype(self.obs_dtype, copy=False)
        return self._host(obs).astype(self.obs_dtype, copy=False)

    def step_async(self, actions) -> None:
        self._actions = np.ascontiguousarray(actions, np.float32 if self._fast else np.float64)

    def _step_wait_fast(self):
        vec, eng = self.vec, self.vec.engine
        if eng.current_step >= vec.simulation_length:
            raise AssertionError("Episode is done, please reset the environment")   # ev2gym_env.py:343
        assert self._actions.shape == (self.num_envs, vec.number_of_ports), self._actions.shape
        self._d_act32.upload(self._actions)
        eng.step(None, None, vec._rew, vec._done, vec._mask)     # float32 actions in, float32 observations out (the extras)
        obs = self._d_obs32.to_host(self._h_obs)
        rew = vec._rew.to_host(self._h_rew)
        done = vec._done.to_host(self._h_done).astype(bool)
        vec._mask.to_host(self._h_mask)
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
            obs = self._d_obs32.to_host(self._h_obs)
            self._ep_return[:] = 0.0
        return obs.copy(), rew.astype(np.float32), done, infos

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
