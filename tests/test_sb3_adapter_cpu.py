"""The SB3 VecEnv adapter's protocol logic on the CPU, against a stub of EV2GymVec (the GPU test drives the real one):
numpy in / out, dtype handling, reset inside step_wait at episode end, terminal_observation / episode infos."""
import numpy as np

from ev2gym_amd.sb3_vec_env import EV2GymSB3VecEnv
from ev2gym_amd.vec_env import Box


class StubVec:
    """Three envs, two ports, episodes of four steps; observation = [step, env id], reward = sum of actions."""
    auto_reset = False
    num_envs, number_of_ports, obs_dim, simulation_length = 3, 2, 2, 4
    action_space = Box(-1.0, 1.0, (2,))
    observation_space = Box(-np.inf, np.inf, (2,))

    def __init__(self):
        self.t = 0
        self.resets = 0

    def _obs(self):
        return np.stack([np.full(3, float(self.t)), np.arange(3.0)], 1)

    def reset(self, **kw):
        self.t = 0
        self.resets += 1
        return self._obs(), {}

    def step(self, a):
        assert a.dtype == np.float64 and a.shape == (3, 2)
        self.t += 1
        done = np.full(3, self.t >= 4, np.uint8)
        info = {"action_mask": np.ones((3, 2), np.uint8), "cost": None}
        if self.t >= 4:
            info["total_profits"] = np.array([1.0, 2.0, 3.0])
        return self._obs(), a.sum(1), done, np.zeros(3, bool), info

    def close(self):
        pass


def test_adapter_protocol_on_a_stub():
    venv = EV2GymSB3VecEnv(vec=StubVec())
    assert venv.num_envs == 3 and venv.observation_space.shape == (2,) and venv.action_space.shape == (2,)
    obs = venv.reset()
    assert obs.dtype == np.float32 and obs.shape == (3, 2) and (obs[:, 0] == 0).all()
    total = np.zeros(3)
    for t in range(1, 7):
        a = np.full((3, 2), 0.25 * t, np.float32)
        venv.step_async(a)
        obs, rew, done, infos = venv.step_wait()
        assert rew.dtype == np.float32 and done.dtype == bool and len(infos) == 3
        assert np.allclose(rew, 0.5 * t)
        total = total + 0.5 * t if t <= 4 else total
        if t == 4:   # episode end: reset happened inside step_wait, the terminal observation rides in the infos
            assert done.all() and (obs[:, 0] == 0).all() and venv.vec.resets == 2
            for i, info in enumerate(infos):
                assert info["terminal_observation"][0] == 4 and info["terminal_observation"][1] == i
                assert info["episode"] == {"r": float(total[i]), "l": 4} and info["TimeLimit.truncated"] is False
                assert info["total_profits"] == float(i + 1)
        else:
            assert not done.any() and "terminal_observation" not in infos[0]
            assert (obs[:, 0] == (t if t < 4 else t - 4)).all()
    assert venv.env_is_wrapped(object) == [False] * 3 and venv.get_attr("simulation_length", [0, 2]) == [4, 4]
    assert venv.seed(7) == [7, 8, 9]
    venv.close()
