#!/usr/bin/env python3
"""Summary statistics of the reference's scenario construction (EV_spawner / spawn_single_EV, utilities/utils.py:177-345,
477-557) over many resets, written to tests/golden/spawn_stats.json.  Build container only (imports the reference);
the vectorised generator (ev2gym_amd/scenario_gen.py) is pinned against these numbers by tests/test_host_logic.py --
statistically, not bit for bit (the generator does not reproduce the reference's RNG streams or CSVs).
"""
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
warnings.filterwarnings("ignore")
from ref_import import import_reference  # noqa: E402


def stats_for(config, n_resets, state_fn, reward_fn):
    from ev2gym.models.ev2gym_env import EV2Gym
    import ev2gym.rl_agent.state as S
    import ev2gym.rl_agent.reward as RW
    env = EV2Gym(config_file=config, seed=0, state_function=getattr(S, state_fn), reward_function=getattr(RW, reward_fn),
                 generate_rnd_game=True)
    T, P = env.simulation_length, env.number_of_ports
    occ, nsess, stay, soc0, bcap, tarr = [], [], [], [], [], []
    sp_ratio, sp_max, sp_nz, sp_mean = [], [], [], []
    for seed in range(n_resets):
        env.reset(seed=seed)
        evs = env.EVs_profiles
        sp = np.array(env.power_setpoints, float)
        need = sum(ev.battery_capacity - ev.battery_capacity_at_arrival for ev in evs)
        if sp.any() and need > 0:   # generate_power_setpoints (utils.py:664-757): energy of the setpoint curve vs the energy the EVs need
            sp_ratio.append(sp.sum() * env.timescale / 60 / need); sp_max.append(sp.max()); sp_nz.append((sp > 0).mean()); sp_mean.append(sp.mean())
        nsess.append(len(evs) / P)
        m = np.zeros((T + 1, P), bool)
        for ev in evs:
            ta, td = ev.time_of_arrival, min(ev.time_of_departure, T)
            m[ta:td + 1, ev.location * env.number_of_ports_per_cs + ev.id] = True
            stay.append(ev.time_of_departure - ev.time_of_arrival)
            soc0.append(ev.battery_capacity_at_arrival / ev.battery_capacity)
            bcap.append(ev.battery_capacity)
            tarr.append(ev.time_of_arrival)
        occ.append(m[1:T + 1].mean())
    q = lambda x, p: float(np.quantile(x, p))  # noqa: E731
    hist = np.histogram(tarr, bins=np.linspace(0, T, 8))[0]
    tarr_a, stay_a = np.array(tarr), np.array(stay)
    hourly = np.histogram(tarr_a, bins=np.arange(0, T + 4, 4))[0] / len(tarr_a)                  # per hour of the episode
    stay_by_2h = [float(stay_a[(tarr_a >= a) & (tarr_a < a + 8)].mean()) if ((tarr_a >= a) & (tarr_a < a + 8)).any() else None
                  for a in range(0, T, 8)]
    req = np.array(bcap) * (1 - np.array(soc0))
    return dict(n_resets=int(n_resets), ports=int(P), steps=int(T),
                occupancy_mean=float(np.mean(occ)), occupancy_std=float(np.std(occ)),
                sessions_per_port_mean=float(np.mean(nsess)), sessions_per_port_std=float(np.std(nsess)),
                stay_mean=float(np.mean(stay)), stay_q10=q(stay, .1), stay_q50=q(stay, .5), stay_q90=q(stay, .9),
                soc_at_arrival_mean=float(np.mean(soc0)), soc_at_arrival_q10=q(soc0, .1), soc_at_arrival_q90=q(soc0, .9),
                battery_capacity_mean=float(np.mean(bcap)), battery_capacity_min=float(np.min(bcap)),
                battery_capacity_max=float(np.max(bcap)),
                arrival_hist_7bins=[float(h) / len(tarr) for h in hist],
                arrival_share_per_hour=[float(x) for x in hourly], stay_mean_by_2h_arrival_bin=stay_by_2h,
                required_energy_mean=float(req.mean()), required_energy_q10=q(req, .1), required_energy_q50=q(req, .5),
                required_energy_q90=q(req, .9), stay_q05=q(stay, .05), stay_q25=q(stay, .25), stay_q75=q(stay, .75),
                stay_q95=q(stay, .95), small_battery_share=float((np.array(bcap) < 20).mean()),
                setpoint_energy_ratio_mean=float(np.mean(sp_ratio)) if sp_ratio else None,
                setpoint_max_mean=float(np.mean(sp_max)) if sp_max else None,
                setpoint_nonzero_fraction=float(np.mean(sp_nz)) if sp_nz else None,
                setpoint_mean_kw=float(np.mean(sp_mean)) if sp_mean else None)


def main():
    import_reference()
    from capture_golden import _yaml_variant
    base = "ev2gym/example_config_files/"
    jobs = {"V2GProfitPlusLoads": lambda: stats_for(base + "V2GProfitPlusLoads.yaml", 300, "V2G_profit_max_loads", "ProfitMax_TrPenalty_UserIncentives"),
            "PublicPST": lambda: stats_for(base + "PublicPST.yaml", 300, "PublicPST", "SquaredTrackingErrorReward"),
            # the third scenario kind of the reference's arrival / stay / energy distributions ('private': home charging), on the V2GPPL config
            "PrivateV2GPPL": lambda: stats_for(_yaml_variant(base + "V2GProfitPlusLoads.yaml", {"scenario": "private"}, "v2gppl_private"), 300,
                                               "V2G_profit_max_loads", "ProfitMax_TrPenalty_UserIncentives"),
            # weekend days (simulation_days: weekends; ev2gym_env.py:151-154, utils.py:366-382,519-528) of the two scenarios that have them
            "PublicPSTWeekend": lambda: stats_for(_yaml_variant(base + "PublicPST.yaml", {"simulation_days": "weekends"}, "pst_weekend"), 300,
                                                  "PublicPST", "SquaredTrackingErrorReward"),
            "PrivateV2GPPLWeekend": lambda: stats_for(_yaml_variant(base + "V2GProfitPlusLoads.yaml", {"scenario": "private", "simulation_days": "weekends"},
                                                                    "v2gppl_private_weekend"), 300, "V2G_profit_max_loads", "ProfitMax_TrPenalty_UserIncentives")}
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "spawn_stats.json")
    only = set(sys.argv[1:]) or set(jobs)
    out = json.load(open(path)) if os.path.exists(path) else {}   # entries that were not asked for are kept as committed
    for k in jobs:
        if k in only:
            out[k] = jobs[k]()
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: out[k] for k in only}, indent=1))


if __name__ == "__main__":
    main()
