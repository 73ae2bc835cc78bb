"""gymnasium integration (optional dependency).

The reference's env IS a `gymnasium.Env` (ev2gym_env.py:36) and registers the id `EV2Gym-v1` on import (ev2gym/__init__.py:3-7).
When gymnasium is importable the facade `ev2gym_amd.env.EV2Gym` subclasses `gymnasium.Env`, its spaces are `gymnasium.spaces.Box`
and `import ev2gym_amd` registers the same id (entry point: the facade, default config: the packaged V2GProfitMax.yaml like the
reference's), so `gymnasium.make("EV2Gym-v1", config_file=...)`, SB3's env checker and gym wrappers see what they expect.  Without
gymnasium everything still works on the minimal stand-ins below.
"""
import os

import numpy as np

try:
    import gymnasium as _gym
    if not (hasattr(_gym, "Env") and hasattr(_gym, "spaces") and hasattr(_gym.spaces, "Box")):
        _gym = None   # not a real gymnasium (e.g. a stub somebody parked in sys.modules)
except ImportError:   # optional
    _gym = None

GYM_ID = "EV2Gym-v1"
EnvBase = _gym.Env if _gym is not None else object


class _Box:
    """Minimal stand-in for gymnasium.spaces.Box."""

    def __init__(self, low, high, shape, dtype=np.float64):
        self.low = np.full(shape, low, dtype) if np.isscalar(low) else np.asarray(low, dtype)
        self.high = np.full(shape, high, dtype) if np.isscalar(high) else np.asarray(high, dtype)
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)


def Box(low, high, shape, dtype=np.float64):
    """gymnasium.spaces.Box when gymnasium is present (ev2gym_env.py:226-238 builds its spaces with it), the stand-in otherwise."""
    if _gym is not None:
        return _gym.spaces.Box(low=low, high=high, shape=tuple(shape), dtype=dtype)
    return _Box(low, high, shape, dtype)


def register_gym_id():
    """Register `EV2Gym-v1` (once).  Returns True when gymnasium is present."""
    if _gym is None:
        return False
    try:   # a partial `gymnasium` in sys.modules (a test stand-in, the oracle's import stub) must never break `import ev2gym_amd`
        from gymnasium.envs.registration import register, registry
    except (ImportError, AttributeError):
        return False
    if GYM_ID not in registry:
        default_cfg = os.path.join(os.path.dirname(os.path.abspath(__file__)), "example_config_files", "V2GProfitMax.yaml")
        register(id=GYM_ID, entry_point="ev2gym_amd.env:EV2Gym", kwargs={"config_file": default_cfg})
    return True
