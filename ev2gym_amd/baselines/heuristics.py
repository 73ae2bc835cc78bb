"""Rule-based agents with the reference's `get_action(env)` surface (ev2gym/baselines/heuristics.py).

Both work on the single-env facade (returning `np.ndarray[number_of_ports]` like the reference) and on
`EV2GymVec` (returning a `[num_envs, number_of_ports]` array / device tensor).
"""
import numpy as np


class ChargeAsFastAsPossible:
    """heuristics.py:152-166: every port at full charging power."""
    algo_name = "Charge As Fast As Possible"

    def __init__(self, verbose=False, **kwargs):
        self.verbose = verbose

    def get_action(self, env):
        if hasattr(env, "num_envs"):
            return env.full_like_actions(1.0)
        return np.ones(env.number_of_ports)


class RandomAgent:
    """heuristics.py:546-558: uniform samples of the action box."""
    algo_name = "Random Actions"

    def __init__(self, env=None, seed=0, **kwargs):
        self.rng = np.random.default_rng(seed)
        self._calls = 0
        self.seed = seed

    def get_action(self, env):
        low = -1.0 if env.v2g_enabled else 0.0
        if hasattr(env, "num_envs"):
            self._calls += 1
            return env.uniform_actions(self.seed * 1000003 + self._calls, low, 1.0)
        return self.rng.uniform(low, 1.0, env.number_of_ports)


class DoNothing:
    """heuristics.py:533-544: no port charges or discharges."""
    algo_name = "DO NOTHING"

    def __init__(self, verbose=False, **kwargs):
        self.verbose = verbose

    def get_action(self, env):
        if hasattr(env, "num_envs"):
            return env.full_like_actions(0.0)
        return np.zeros(env.number_of_ports)
