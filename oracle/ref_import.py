"""Import shim for the *reference* EV2Gym (test infrastructure, THIS container only).

Only `oracle/capture_golden.py` uses this module.  It makes `/root/reference`
importable without network access by
  (1) installing `sys.modules` stubs for the packages the reference imports at
      module scope but that are absent from the image (gymnasium, pandapower,
      numba, multicopula)  -- none of them is touched by `EV2Gym.step()` when
      `simulate_grid: False` (ev2gym_env.py:387-397 is the only grid branch);
  (2) serving synthetic stand-ins for the two data blobs the checkout lacks
      (/root/reference/.MISSING_LARGE_BLOBS): the day-ahead price CSV
      (loaders.py:405-421) and the residential load CSV (loaders.py:113-115);
  (3) chdir-ing to /root/reference because the YAML configs use relative paths
      (V2GProfitPlusLoads.yaml:110).
The stand-in *values* are synthetic.  That does not weaken step() parity:
step() is a deterministic function of the scenario tensors captured after
reset() (SURVEY.md header), which is exactly what the goldens record.

Nothing from the reference is copied here; the reference source never travels.
"""
import os
import sys
import types

REF_ROOT = "/root/reference"
STANDIN_DIR = os.environ.get("EV2G_STANDIN_DIR", "/tmp/ev2g_standins")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_stubs():
    if "gymnasium" not in sys.modules:
        class Env:  # gymnasium.Env surface the reference touches
            metadata = {}

            @property
            def unwrapped(self):
                return self

        class Wrapper:
            def __init__(self, env=None):
                self.env = env

        class Box:
            def __init__(self, low=None, high=None, shape=None, dtype=None):
                import numpy as np
                self.low = np.asarray(low)
                self.high = np.asarray(high)
                self.shape = tuple(shape) if shape is not None else self.low.shape
                self.dtype = dtype

        class _Space:
            def __init__(self, *a, **k):
                pass

        class RecordConstructorArgs:
            def __init__(self, *a, **k):
                pass

        spaces = _mod("gymnasium.spaces", Box=Box, MultiDiscrete=_Space, Discrete=_Space)
        reg = _mod("gymnasium.envs.registration", register=lambda *a, **k: None)
        envs = _mod("gymnasium.envs", registration=reg, register=lambda *a, **k: None)
        core = _mod("gymnasium.core", WrapperObsType=object, ActType=object, ObsType=object,
                    WrapperActType=object)
        utils = _mod("gymnasium.utils", RecordConstructorArgs=RecordConstructorArgs)
        _mod("gymnasium", Env=Env, Wrapper=Wrapper, ActionWrapper=Wrapper,
             ObservationWrapper=Wrapper, spaces=spaces, envs=envs, core=core, utils=utils)
    if "pandapower" not in sys.modules:
        top = _mod("pandapower.topology")
        _mod("pandapower", topology=top)
    if "numba" not in sys.modules:
        def njit(*a, **k):
            if len(a) == 1 and callable(a[0]) and not k:
                return a[0]
            return lambda f: f
        _mod("numba", njit=njit, jit=njit, prange=range, set_num_threads=lambda n: None)
    if "multicopula" not in sys.modules:
        class EllipticalCopula:
            def __init__(self, *a, **k):
                pass
        _mod("multicopula", EllipticalCopula=EllipticalCopula)


def _make_standins():
    """Synthetic replacements for the two missing CSV blobs (schemas: SURVEY.md App. B)."""
    import numpy as np
    os.makedirs(STANDIN_DIR, exist_ok=True)
    price = os.path.join(STANDIN_DIR, "Netherlands_day-ahead-2015-2024.csv")
    loads = os.path.join(STANDIN_DIR, "residential_loads.csv")
    if not os.path.exists(price):
        import pandas as pd
        idx = pd.date_range("2015-01-01", "2024-12-31 23:00:00", freq="h")
        rng = np.random.default_rng(2024)
        hod = idx.hour.values
        base = 60 + 35 * np.sin((hod - 7) / 24 * 2 * np.pi) + 25 * np.sin((hod - 17) / 12 * 2 * np.pi)
        val = np.round(base + rng.normal(0, 12, len(idx)) + 30, 2)
        val = np.maximum(val, 5.0)
        df = pd.DataFrame({"Country": "Netherlands",
                           "Datetime (UTC)": idx.strftime("%Y-%m-%d %H:%M:%S"),
                           "Datetime (Local)": idx.strftime("%Y-%m-%d %H:%M:%S"),
                           "Price (EUR/MWhe)": val})
        df.to_csv(price, index=False)
    if not os.path.exists(loads):
        rng = np.random.default_rng(7)
        n = 35040
        tod = (np.arange(n) % 96) / 96.0
        cols = []
        for c in range(16):
            prof = 0.35 + 0.25 * np.sin((tod - 0.3) * 2 * np.pi) ** 2 + 0.5 * np.exp(-((tod - 0.8) / 0.08) ** 2)
            cols.append(np.round(np.abs(prof * rng.uniform(0.6, 1.4) + rng.normal(0, 0.05, n)), 4))
        np.savetxt(loads, np.stack(cols, 1), delimiter=",", fmt="%.4f")
    return {"data/Netherlands_day-ahead-2015-2024.csv": price,
            "data/residential_loads.csv": loads}


def import_reference():
    """Returns the reference `ev2gym` package (imported from /root/reference)."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present: this only works in the build container")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(1, repo)
    import ev2gym_amd  # noqa: F401  -- BEFORE the gymnasium stub exists: the package must see the image's gymnasium (none), not ours
    _install_stubs()
    standins = _make_standins()
    import pkg_resources
    if not getattr(pkg_resources.resource_filename, "_ev2g_wrapped", False):
        orig = pkg_resources.resource_filename

        def resource_filename(pkg, path):
            if pkg == "ev2gym" and path in standins:
                return standins[path]
            return orig(pkg, path)
        resource_filename._ev2g_wrapped = True
        pkg_resources.resource_filename = resource_filename
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    os.chdir(REF_ROOT)
    import matplotlib
    matplotlib.use("Agg")
    import ev2gym  # noqa: F401
    return ev2gym
