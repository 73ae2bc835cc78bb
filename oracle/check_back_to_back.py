"""Back-to-back sessions (the next EV plugs in at the end of the step its predecessor leaves in): does the C oracle follow the
REFERENCE there?  (test infrastructure, THIS container only)

The reference's spawner keeps a gap between two sessions of a port, so no reference-generated fixture holds this case; a replayed
scenario may.  For three generated scenarios the stays are extended up to the step before the port's next arrival, the scenario is
written with write_replay, the live reference loads and steps it, and the oracle must produce the same observations / rewards /
masks on the same scenario.  The engine is held to the oracle on such scenarios by tests/test_fuzz_gpu.py.
Usage: python oracle/check_back_to_back.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from ref_import import import_reference  # noqa: E402
from capture_golden import _yaml_variant  # noqa: E402


def main():
    from ev2gym_amd import _abi
    from ev2gym_amd.replay import load_replay, write_replay
    from ev2gym_amd.scenario import resolve_ports
    from oracle import Oracle
    import_reference()
    from ev2gym.models.ev2gym_env import EV2Gym
    import ev2gym.rl_agent.state as S
    import ev2gym.rl_agent.reward as RW
    from ev2gym_amd.config import gen_config_from_yaml, load_yaml
    from ev2gym_amd.engine import host_uniform
    from ev2gym_amd.scenario_gen import generate
    base = "ev2gym/example_config_files/"
    mine = os.path.join(os.path.dirname(HERE), "ev2gym_amd", "example_config_files")
    cases = [("V2GProfitPlusLoads.yaml", {"number_of_charging_stations": 6, "spawn_multiplier": 10, "scenario": "public"}, "V2G_profit_max_loads", "ProfitMax_TrPenalty_UserIncentives", -1.0),
             ("V2GProfitPlusLoads.yaml", {"number_of_charging_stations": 4, "number_of_ports_per_cs": 2, "spawn_multiplier": 10, "scenario": "public"}, "V2G_profit_max", "profit_maximization", -1.0),
             ("PublicPST.yaml", {"number_of_charging_stations": 5, "spawn_multiplier": 10}, "PublicPST", "SquaredTrackingErrorReward", 0.0)]
    worst = 0.0
    for ci, (yml, over, sname, rname, lo) in enumerate(cases):
        batch = generate(gen_config_from_yaml({**load_yaml(os.path.join(mine, yml)), **over}, 1, 40 + ci))
        a, port, last, n = batch.arrays, resolve_ports(batch), {}, 0
        for s in range(batch.n_sessions):
            if port[s] in last and a["ev_t_dep"][last[port[s]]] != a["ev_t_arr"][s] - 1:
                a["ev_t_dep"][last[port[s]]] = a["ev_t_arr"][s] - 1
                n += 1
            last[port[s]] = s
        assert n > 0 and np.array_equal(resolve_ports(batch), port), n
        path = write_replay(os.path.join(tempfile.mkdtemp(), "replay_sim_back_to_back.pkl"), batch)
        cfg = _yaml_variant(base + yml, over, f"b2b{ci}")
        env = EV2Gym(config_file=cfg, load_from_replay_path=path, state_function=getattr(S, sname), reward_function=getattr(RW, rname))
        ora = Oracle(batch, _abi.REWARD_KINDS[rname], _abi.STATE_KINDS[sname])
        T, P = batch.n_steps, batch.n_ports
        acts = host_uniform(T * P, 7 + ci, lo, 1.0).reshape(T, 1, P)
        obs, _ = env.reset()
        err = float(np.abs(obs - ora.reset()[0]).max())
        for t in range(T):
            obs, rew, done, _, info = env.step(acts[t, 0].copy())
            o, r, d, m, rc = ora.step(acts[t].copy())
            err = max(err, float(np.abs(obs - o[0]).max()), abs(float(rew) - float(r[0])))
            assert rc == 0 and np.array_equal(np.asarray(info["action_mask"], np.uint8), m[0]), (yml, t)
        assert done
        print(f"{yml} {over}: {n} of {batch.n_sessions} stays extended to back-to-back, {T} steps, worst |reference - oracle| {err:.3g}")
        worst = max(worst, err)
        ora.close()
    assert worst < 1e-9, worst
    print("OK")


if __name__ == "__main__":
    main()
