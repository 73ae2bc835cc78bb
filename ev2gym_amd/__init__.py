"""ev2gym_amd -- MI355X-native vectorised EV2Gym step engine (hand-written HIP for gfx950 behind a C-ABI)."""
from . import _abi  # noqa: F401
from .scenario import ScenarioBatch  # noqa: F401

__all__ = ["ScenarioBatch"]
