import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
# plugin_*.npz: reference episodes with the reward built-ins beyond the three of the shipped configs (fused since round 2:
# they run through the oracle and the engine like every other fixture; the facade tests also evaluate them on the host)
# b2b_*.npz (back-to-back sessions) were captured after the last device run of round 2: the CPU tests use them like every other fixture
# (ALL_GOLDEN_*), the GPU suite meets them in its LAST file (tests/test_zz_late_additions_gpu.py) rather than in the middle of
# test_engine_gpu.py, so that a surprise there cannot cut a `pytest -x` run short of the tests that were already seen green on the device
ALL_GOLDEN_FILES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
B2B_FILES = [f for f in ALL_GOLDEN_FILES if os.path.basename(f).startswith("b2b_")]
GOLDEN_FILES = [f for f in ALL_GOLDEN_FILES if f not in B2B_FILES]
GOLDEN_IDS = [os.path.basename(f)[:-4] for f in GOLDEN_FILES]
ALL_GOLDEN_IDS = [os.path.basename(f)[:-4] for f in ALL_GOLDEN_FILES]
B2B_IDS = [os.path.basename(f)[:-4] for f in B2B_FILES]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(path):
    from ev2gym_amd import _abi
    from ev2gym_amd.scenario import ScenarioBatch
    z = np.load(path)
    name, cfg, sf, rf, seed, pol = [str(x) for x in z["case"]]
    batch = ScenarioBatch.from_single(z)
    return z, batch, _abi.REWARD_KINDS[rf], _abi.STATE_KINDS[sf]


@pytest.fixture(params=ALL_GOLDEN_FILES, ids=ALL_GOLDEN_IDS)
def golden(request):
    return load_golden(request.param)


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
