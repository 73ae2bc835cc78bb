"""Round-3 additions, through the C-ABI on the GPU (each block says which review item it closes)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu
CFG = os.path.join(os.path.dirname(GOLDEN_DIR), "..", "ev2gym_amd", "example_config_files")


def test_vec_env_windows_tile_the_pool_and_stats_survive_the_auto_reset():
    """EV2GymVec.reset() visits the M // E disjoint windows of the resident pool without replacement (no scenario is stepped twice
    before all have been); after an auto-reset `env.stats` still holds the finished episode's statistics, like the reference's
    env.stats (ev2gym_env.py:476-480) -- it used to be None."""
    from ev2gym_amd.vec_env import EV2GymVec
    env = EV2GymVec(config_file=os.path.join(CFG, "V2GProfitPlusLoads.yaml"), num_envs=16, seed=5, pool_factor=4, auto_reset=True,
                    state_function="V2G_profit_max_loads", reward_function="ProfitMax_TrPenalty_UserIncentives", use_torch=False)
    M, E = env.engine.M, env.engine.E
    assert (M, E) == (64, 16)
    a = env.full_like_actions(1.0)
    seen = []
    for ep in range(8):   # two passes over the pool
        seen.append(env.engine.scenario_offset)
        for _ in range(env.simulation_length):
            obs, rew, done, trunc, info = env.step(a)
        assert np.asarray(done).all() and env.stats is not None
        assert np.array_equal(env.stats["total_ev_served"], info["total_ev_served"])
        assert env.engine.current_step == 0   # auto-reset happened
    for p in (seen[:4], seen[4:]):
        covered = np.concatenate([(np.arange(E) + o) % M for o in p])
        assert len(set(covered.tolist())) == M, "a pass over the pool must step every scenario exactly once"
    env.reset()
    assert env.stats is None   # a user-initiated reset starts a new episode record
    env.close()
