#!/usr/bin/env python3
"""Differential fuzz of the C oracle against the live reference (build container only; nothing here ships to the GPU box).

For many (config variant, seed, action policy) draws it builds the reference env, flattens its scenario with
capture_golden.extract_scenario, steps both the reference and oracle/ev2g_oracle.c with the same actions and
compares observation / reward / mask every step and the final statistics.  The committed fixtures pin a fixed set of
cases; this widens the net whenever the oracle is touched:   python oracle/fuzz_vs_reference.py [n_cases]
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
warnings.filterwarnings("ignore")
from ref_import import import_reference  # noqa: E402
import capture_golden as cg  # noqa: E402


def main(n_cases):
    import_reference()
    from ev2gym.models.ev2gym_env import EV2Gym
    import ev2gym.rl_agent.state as S
    import ev2gym.rl_agent.reward as RW
    from ev2gym_amd import _abi
    from ev2gym_amd.scenario import ScenarioBatch
    from oracle import Oracle
    base = "ev2gym/example_config_files/"
    rng = np.random.default_rng(int(os.environ.get("EV2G_FUZZ_SEED", "2026")))
    worst = 0.0
    for case in range(n_cases):
        kind = case % 3
        over = {"number_of_charging_stations": int(rng.integers(3, 40)), "number_of_ports_per_cs": int(rng.choice([1, 1, 2, 3])),
                "number_of_transformers": int(rng.integers(1, 4)), "timescale": int(rng.choice([5, 15, 15, 30])),
                "heterogeneous_ev_specs": bool(rng.random() < 0.7)}
        topo = None
        if rng.random() < 0.25:   # a topology file: chargers with their own port counts (falling order: the reference's mask index
            # i*n_ports+j stays inside the array), current limits, voltage and phases (loaders.py:259-276, 312-340)
            nps = sorted(rng.integers(1, 5, int(rng.integers(3, 9))).tolist(), reverse=True)
            cut = sorted(rng.choice(np.arange(1, len(nps)), size=min(int(over["number_of_transformers"]) - 1, len(nps) - 1), replace=False).tolist())
            groups = np.split(np.array(nps), cut)
            topo = cg._topology_file(f"fuzz{case}", [(float(rng.choice([40, 60, 100])),
                                                      [(int(n), float(rng.choice([16, 32])), float(rng.choice([0, -16, -32])) if kind != 1 else 0.0,
                                                        float(rng.choice([230, 400])), int(rng.choice([1, 3]))) for n in g]) for g in groups])
            over["charging_network_topology"] = topo
        yaml_file, sf = [("V2GProfitPlusLoads.yaml", "V2G_profit_max_loads"), ("PublicPST.yaml", "PublicPST"), ("V2GProfitMax.yaml", "V2G_profit_max")][kind]
        rf = str(rng.choice(sorted(_abi.REWARD_KINDS)))      # every fused reward built-in, with every state
        cfg = cg._yaml_variant(base + yaml_file, over, f"fuzz{case}")
        seed = int(rng.integers(0, 10 ** 6))
        env = EV2Gym(config_file=cfg, seed=seed, state_function=getattr(S, sf), reward_function=getattr(RW, rf), generate_rnd_game=True)
        obs0, _ = env.reset(seed=seed)
        scn = cg.extract_scenario(env)
        batch = ScenarioBatch.from_single(scn)
        ora = Oracle(batch, _abi.REWARD_KINDS[rf], _abi.STATE_KINDS[sf])
        o = ora.reset()
        err = np.abs(o[0] - obs0).max()
        T, P = env.simulation_length, env.number_of_ports
        lo = -1.0 if env.config["v2g_enabled"] else 0.0
        pol = rng.choice(["rand", "wild", "mixed"])
        info = None
        faulted = False
        for t in range(T):
            a = rng.uniform(lo, 1.0, P) if pol == "rand" else (rng.uniform(-1.6 if lo < 0 else 0, 1.6, P) if pol == "wild"
                                                                 else rng.uniform(lo, 1.0, P) * (rng.random(P) < 0.7))
            try:
                ro, rr, rd, _, info = env.step(a.copy())
            except Exception as ex:   # the reference's over-current exception (ev_charger.py:203-205): the oracle must flag the same step
                assert "sum of amps" in str(ex), ex
                oo, orr, od, om, rc = ora.step(a[None].copy())
                assert rc != 0, (case, t, "reference raised over-current, oracle did not")
                faulted = True
                break
            oo, orr, od, om, rc = ora.step(a[None].copy())
            assert rc == 0 and bool(od[0]) == bool(rd), (case, t)
            assert np.array_equal(om[0].astype(float), np.asarray(info["action_mask"], float)), (case, t, "mask")
            err = max(err, np.abs(oo[0] - ro).max() / max(1.0, np.abs(ro).max()), abs(orr[0] - rr) / max(1.0, abs(rr)))
        st = ora.stats()[0]
        for i, k in enumerate(cg.STAT_KEYS if not faulted else []):
            rv = float(info[k])
            if np.isnan(rv) and np.isnan(st[i]):
                continue
            err = max(err, abs(st[i] - rv) / max(1.0, abs(rv)))
        ora.close()
        worst = max(worst, err)
        print(f"case {case:3d} {sf:22s} {rf:40s} {'topology ' + str([c.n_ports for c in env.charging_stations]) if topo else ''} C={len(env.charging_stations):2d} npc={over['number_of_ports_per_cs']} R={len(env.transformers)} "
              f"dt={over['timescale']:2d} het={int(over['heterogeneous_ev_specs'])} {pol:5s} max rel err {err:.2e}" + (" (over-current fault at the same step)" if faulted else ""), flush=True)
        assert err < 1e-9, "oracle and reference disagree"
    print(f"{n_cases} cases, worst relative error {worst:.2e}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 30)
