#!/usr/bin/env python3
"""Fits the generator's hourly arrival-rate, stay and energy curves (ev2gym_amd/scenario_gen.py: _HOURLY) to the
reference's summary statistics (tests/golden/spawn_stats.json, written by oracle/capture_spawn_stats.py).
Development tool: prints the tables to paste into scenario_gen.py.  Multiplicative fixed-point updates:
rate_h *= ref_share_h / our_share_h (then a global scale to the sessions-per-port target), stay_h += ref - ours, ...
"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ev2gym_amd import scenario_gen as G
from ev2gym_amd.config import gen_config_from_yaml, load_yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref = json.load(open(os.path.join(ROOT, "tests", "golden", "spawn_stats.json")))


def measure(batch):
    a, T, P = batch.arrays, batch.n_steps, batch.n_ports
    tarr, stay = a["ev_t_arr"], a["ev_t_dep"] - a["ev_t_arr"]
    hourly = np.histogram(tarr, bins=np.arange(0, T + 4, 4))[0] / max(len(tarr), 1)
    s2 = np.array([stay[(tarr >= x) & (tarr < x + 8)].mean() if ((tarr >= x) & (tarr < x + 8)).any() else np.nan for x in range(0, T, 8)])
    req = a["ev_B"] - a["ev_cap0"]
    st = a["env_session_start"]
    return dict(hourly=hourly, stay2h=s2, req=req.mean(), spp=(st[1:] - st[:-1]).mean() / P, occ=G.occupancy_fraction(batch))


want = set(sys.argv[1:])
for name, yaml, over in [("V2GProfitPlusLoads", "V2GProfitPlusLoads.yaml", {}), ("PublicPST", "PublicPST.yaml", {}),
                         ("PrivateV2GPPL", "V2GProfitPlusLoads.yaml", {"scenario": "private"}),
                         ("PublicPSTWeekend", "PublicPST.yaml", {"simulation_days": "weekends"}),
                         ("PrivateV2GPPLWeekend", "V2GProfitPlusLoads.yaml", {"scenario": "private", "simulation_days": "weekends"})]:
    if want and name not in want:
        continue
    r = ref[name]
    cfg = gen_config_from_yaml({**load_yaml(os.path.join(ROOT, "ev2gym_amd", "example_config_files", yaml)), **over}, 400, 1)
    sc = cfg.scenario + ("_weekend" if cfg.simulation_days == "weekends" else "")
    rh = np.array(r["arrival_share_per_hour"]); rs = np.array([x if x is not None else np.nan for x in r["stay_mean_by_2h_arrival_bin"]])
    for it in range(12):
        m = measure(G.generate(cfg))
        H = G._HOURLY[sc]
        n = len(H["rate"])
        for h in range(n):   # hour index = hour of day; episode hour k <-> hour of day cfg.hour + k
            k = h - cfg.hour
            if 0 <= k < len(rh):
                if m["hourly"][k] > 1e-4 and rh[k] > 0: H["rate"][h] *= (rh[k] / m["hourly"][k]) ** 0.7
                elif rh[k] == 0: H["rate"][h] = 0.0
                elif m["hourly"][k] <= 1e-4 and rh[k] > 0: H["rate"][h] = max(H["rate"][h], 0.05) * 1.5
        H["rate"] *= (r["sessions_per_port_mean"] / max(m["spp"], 1e-6)) ** 0.7
        for b in range(len(rs)):
            if not np.isnan(rs[b]) and not np.isnan(m["stay2h"][b]):
                for h in (cfg.hour + 2 * b, cfg.hour + 2 * b + 1):
                    if h < n: H["stay"][h] = max(0.5, H["stay"][h] + 0.6 * (rs[b] - m["stay2h"][b]) * cfg.timescale / 60.0)
        H["energy"] *= (r["required_energy_mean"] / m["req"]) ** 0.7
        print(name, it, "spp %.3f/%.3f occ %.3f/%.3f req %.2f/%.2f" % (m["spp"], r["sessions_per_port_mean"], m["occ"], r["occupancy_mean"], m["req"], r["required_energy_mean"]))
    print(sc, "rate  =", np.round(G._HOURLY[sc]["rate"], 3).tolist())
    print(sc, "stay  =", np.round(G._HOURLY[sc]["stay"], 2).tolist())
    print(sc, "energy=", np.round(G._HOURLY[sc]["energy"], 2).tolist())
