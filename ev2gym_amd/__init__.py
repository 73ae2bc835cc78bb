"""ev2gym_amd -- MI355X-native vectorised EV2Gym step engine (hand-written HIP for gfx950 behind a C-ABI)."""
from . import _abi  # noqa: F401
from .scenario import ScenarioBatch  # noqa: F401
from .gym_compat import register_gym_id

register_gym_id()   # `EV2Gym-v1`, like `import ev2gym` does (ev2gym/__init__.py:3-7); a no-op without gymnasium

__all__ = ["ScenarioBatch"]
