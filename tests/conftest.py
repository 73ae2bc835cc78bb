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
# alphabetical, the back-to-back fixtures (added last, at the end of round 2) at the end
GOLDEN_FILES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz")), key=lambda f: (os.path.basename(f).startswith("b2b_"), f))
GOLDEN_IDS = [os.path.basename(f)[:-4] for f in GOLDEN_FILES]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(path):
    from ev2gym_amd import _abi
    from ev2gym_amd.scenario import ScenarioBatch
    z = np.load(path)
    name, cfg, sf, rf, seed, pol = [str(x) for x in z["case"]]
    batch = ScenarioBatch.from_single(z)
    return z, batch, _abi.REWARD_KINDS[rf], _abi.STATE_KINDS[sf]


@pytest.fixture(params=GOLDEN_FILES, ids=GOLDEN_IDS)
def golden(request):
    return load_golden(request.param)


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
