"""gymnasium is optional and absent from this image: a stand-in module shows what `import ev2gym_amd` does when it is present --
the id `EV2Gym-v1` is registered with the facade as its entry point (ev2gym/__init__.py:3-7) and the facade subclasses
gymnasium.Env with gymnasium.spaces.Box spaces (ev2gym_env.py:36,226-238).  Run in a subprocess so the stand-in never leaks."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STANDIN = textwrap.dedent('''
    import sys, types
    gym = types.ModuleType("gymnasium")
    class Env:
        metadata = {}
        @property
        def unwrapped(self): return self
    class Box:
        def __init__(self, low, high, shape, dtype): self.low, self.high, self.shape, self.dtype = low, high, shape, dtype
    gym.Env = Env
    gym.spaces = types.ModuleType("gymnasium.spaces"); gym.spaces.Box = Box
    envs = types.ModuleType("gymnasium.envs"); reg = types.ModuleType("gymnasium.envs.registration")
    reg.registry = {}
    def register(id, entry_point, kwargs=None): reg.registry[id] = (entry_point, kwargs)
    reg.register = register
    envs.registration = reg; gym.envs = envs
    sys.modules.update({"gymnasium": gym, "gymnasium.spaces": gym.spaces, "gymnasium.envs": envs, "gymnasium.envs.registration": reg})
''')


def _run(body):
    r = subprocess.run([sys.executable, "-c", STANDIN + textwrap.dedent(body)], capture_output=True, text=True, cwd=ROOT,
                       env={**os.environ, "PYTHONPATH": ROOT}, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_import_registers_the_reference_id_and_the_facade_is_a_gym_env():
    out = _run('''
        import importlib, os
        import ev2gym_amd
        from ev2gym_amd.env import EV2Gym
        from ev2gym_amd.gym_compat import Box
        ep, kw = reg.registry["EV2Gym-v1"]
        mod, cls = ep.split(":")
        assert getattr(importlib.import_module(mod), cls) is EV2Gym
        assert os.path.exists(kw["config_file"]) and kw["config_file"].endswith("V2GProfitMax.yaml")
        assert issubclass(EV2Gym, gym.Env)
        assert isinstance(Box(-1.0, 1.0, (5,)), gym.spaces.Box)
        ev2gym_amd.gym_compat.register_gym_id()      # idempotent
        assert len(reg.registry) == 1
        print("ok")
    ''')
    assert out.strip() == "ok"


def test_without_gymnasium_the_stand_ins_are_used():
    import ev2gym_amd  # noqa: F401
    from ev2gym_amd import gym_compat
    if gym_compat._gym is None:
        assert gym_compat.EnvBase is object and gym_compat.register_gym_id() is False
        b = gym_compat.Box(0.0, 1.0, (4,))
        assert b.shape == (4,) and (b.sample() >= 0).all()
