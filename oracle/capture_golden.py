#!/usr/bin/env python3
"""Golden-vector capture for the EV2Gym.step() hot path (test infrastructure).

Runs ONLY in the build container, where /root/reference exists.  It imports the
reference (through oracle/ref_import.py), constructs `EV2Gym` for a matrix of
(config, seed, action policy), and writes one `.npz` per case to tests/golden/:

  scn_*   the *scenario*: every tensor `step()` reads, in this repo's schema
          (ev2gym_amd/scenario.py documents the fields) -- captured right after
          construction/reset (ev2gym_env.py:243-331).
  act     the float64 action tensor [T, P] fed to the reference (a fresh copy
          per step, because EV_Charger.step zeroes empty ports in place,
          ev_charger.py:137-140).
  trj_*   the *trajectory*: everything `step()` produces, every step.

A fixture is data only (inputs + expected outputs); no reference source text
is stored.  Re-generate with:  python oracle/capture_golden.py
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
OUT = os.environ.get("EV2G_GOLDEN_OUT") or os.path.join(REPO, "tests", "golden")   # the recipe test regenerates into a temp dir
sys.path.insert(0, HERE)
sys.path.insert(1, os.path.dirname(HERE))   # ev2gym_amd (scenario generator + replay writer of the back-to-back cases); after HERE: `oracle` must stay oracle/oracle.py
warnings.filterwarnings("ignore")

from ref_import import import_reference  # noqa: E402

STAT_KEYS = ['total_ev_served', 'total_profits', 'total_energy_charged', 'total_energy_discharged',
             'average_user_satisfaction', 'power_tracker_violation', 'tracking_error',
             'energy_tracking_error', 'energy_user_satisfaction', 'std_energy_user_satisfaction',
             'min_energy_user_satisfaction', 'total_steps_min_emergency_battery_capacity_violation',
             'total_transformer_overload', 'battery_degradation', 'battery_degradation_calendar',
             'battery_degradation_cycling', 'total_reward']


def _yaml_variant(base, overrides, name):
    import yaml
    cfg = yaml.load(open(base), Loader=yaml.FullLoader)

    def deep(d, o):
        for k, v in o.items():
            if isinstance(v, dict):
                deep(d[k], v)
            else:
                d[k] = v
    deep(cfg, overrides)
    os.makedirs("/tmp/ev2g_cfg", exist_ok=True)
    p = f"/tmp/ev2g_cfg/{name}.yaml"
    yaml.dump(cfg, open(p, "w"))
    return p


def _topology_file(name, transformers):
    """Write a charging_network_topology JSON in the layout of example_config_files/charging_topology_10.json:
    transformers = [(max_power, [(n_ports, max_charge_current, max_discharge_current, voltage, phases), ...]), ...]."""
    import json
    topo, k = {}, 0
    for i, (max_power, chargers) in enumerate(transformers):
        chs = {}
        for n_ports, imax, dmax, volt, ph in chargers:
            chs[f"charger_{k + 1}"] = dict(id=k, min_charge_current=6, max_charge_current=imax, min_discharge_current=0,
                                           max_discharge_current=dmax, voltage=volt, n_ports=n_ports, charger_type="AC", phases=ph)
            k += 1
        topo[f"transformer_{i + 1}"] = dict(id=i + 1, max_power=max_power, charging_stations=chs)
    os.makedirs("/tmp/ev2g_cfg", exist_ok=True)
    p = f"/tmp/ev2g_cfg/topology_{name}.json"
    json.dump(topo, open(p, "w"), indent=1)
    return p


def extract_scenario(env):
    """Flatten the reference object graph into this repo's scenario schema."""
    T = env.simulation_length
    cs = env.charging_stations
    C = len(cs)
    R = len(env.transformers)
    s = {}
    if len({c.n_ports for c in cs}) > 1:     # topology file with different port counts (loaders.py:312-340)
        s["scn_cs_n_ports"] = np.array([c.n_ports for c in cs], np.int32)
    s["scn_meta"] = np.array([T, env.timescale, C, max(c.n_ports for c in cs), R,
                              int(bool(env.config['v2g_enabled'])), 20], dtype=np.int64)
    s["scn_cs_min_charge_current"] = np.array([c.min_charge_current for c in cs], float)
    s["scn_cs_max_charge_current"] = np.array([c.max_charge_current for c in cs], float)
    s["scn_cs_min_discharge_current"] = np.array([c.min_discharge_current for c in cs], float)
    s["scn_cs_max_discharge_current"] = np.array([c.max_discharge_current for c in cs], float)
    s["scn_cs_voltage"] = np.array([c.voltage for c in cs], float)
    s["scn_cs_phases"] = np.array([c.phases for c in cs], np.int32)
    s["scn_cs_transformer"] = np.array([c.connected_transformer for c in cs], np.int32)
    assert (env.charge_prices == env.charge_prices[0]).all()
    assert (env.discharge_prices == env.discharge_prices[0]).all()
    s["scn_charge_price"] = np.array(env.charge_prices[0], float)
    s["scn_discharge_price"] = np.array(env.discharge_prices[0], float)
    s["scn_power_setpoints"] = np.array(env.power_setpoints, float)
    trs = env.transformers
    s["scn_tr_max_power"] = np.array([t.max_power for t in trs], float)
    s["scn_tr_min_power"] = np.array([t.min_power for t in trs], float)
    s["scn_tr_inflexible_load"] = np.array([t.inflexible_load for t in trs], float)
    s["scn_tr_solar_power"] = np.array([t.solar_power for t in trs], float)
    s["scn_tr_load_forecast"] = np.array([t.inflexible_load_forecast for t in trs], float)
    s["scn_tr_pv_forecast"] = np.array([t.pv_generation_forecast for t in trs], float)
    s["scn_tr_voltage"] = np.array([t.voltage for t in trs], float)
    nd = max([len(t.dr_events) for t in trs] + [1])
    dr = np.zeros((R, nd, 3))
    ndr = np.zeros(R, np.int32)
    for i, t in enumerate(trs):
        ndr[i] = len(t.dr_events)
        for j, ev in enumerate(t.dr_events):
            dr[i, j] = (ev['event_start_step'], ev['event_end_step'], ev['capacity_percentage'])
    s["scn_tr_dr"] = dr
    s["scn_tr_n_dr"] = ndr
    s["scn_tr_steps_ahead"] = np.array([t.steps_ahead for t in trs], np.int32)
    evs = env.EVs_profiles
    luts, lut_keys = [], {}

    def lut_id(d):
        # 101-entry table of percent values; keys outside 0..100 fall back to `.get(..., 1)` (ev.py:288)
        tab = tuple(float(d.get(i, 1)) for i in range(101))
        assert all(0 <= k <= 100 for k in d.keys())
        if tab not in lut_keys:
            lut_keys[tab] = len(luts)
            luts.append(tab)
        return lut_keys[tab]
    f = lambda name: np.array([getattr(e, name) for e in evs], float)  # noqa: E731
    s["scn_ev_cs"] = np.array([e.location for e in evs], np.int32)
    s["scn_ev_t_arr"] = np.array([e.time_of_arrival for e in evs], np.int32)
    s["scn_ev_t_dep"] = np.array([e.time_of_departure for e in evs], np.int32)
    s["scn_ev_cap0"] = f("battery_capacity_at_arrival")
    s["scn_ev_B"] = f("battery_capacity")
    s["scn_ev_desired"] = f("desired_capacity")
    s["scn_ev_minB"] = f("min_battery_capacity")
    s["scn_ev_min_emerg"] = f("min_emergency_battery_capacity")
    s["scn_ev_pac_max"] = f("max_ac_charge_power")
    s["scn_ev_pac_min"] = f("min_ac_charge_power")
    s["scn_ev_pdis_max"] = f("max_discharge_power")
    s["scn_ev_pdis_min"] = f("min_discharge_power")
    s["scn_ev_ts"] = f("transition_soc")
    s["scn_ev_tsm"] = f("transition_soc_multiplier")
    s["scn_ev_phases"] = np.array([e.ev_phases for e in evs], np.int32)
    eta_ch, eta_dis, lid = [], [], []
    for e in evs:
        if isinstance(e.charge_efficiency, dict):
            assert e.discharge_efficiency == e.charge_efficiency
            lid.append(lut_id(e.charge_efficiency))
            eta_ch.append(np.nan)
            eta_dis.append(np.nan)
        else:
            lid.append(-1)
            eta_ch.append(float(e.charge_efficiency))
            eta_dis.append(float(e.discharge_efficiency))
    s["scn_ev_eta_ch"] = np.array(eta_ch, float)
    s["scn_ev_eta_dis"] = np.array(eta_dis, float)
    s["scn_ev_lut"] = np.array(lid, np.int32)
    s["scn_lut"] = np.array(luts, float).reshape(-1, 101) if luts else np.zeros((0, 101))
    return s


def run_case(name, config, state_fn, reward_fn, seed, policy, steps=None, env=None, extra=None):
    from ev2gym.models.ev2gym_env import EV2Gym
    import ev2gym.rl_agent.state as S
    import ev2gym.rl_agent.reward as RW
    if env is None:
        env = EV2Gym(config_file=config, seed=seed, state_function=getattr(S, state_fn),
                     reward_function=getattr(RW, reward_fn), generate_rnd_game=True)
    obs0, _ = env.reset(seed=seed)
    scn = extract_scenario(env)
    T = env.simulation_length
    P = env.number_of_ports
    C = len(env.charging_stations)
    R = len(env.transformers)
    base = np.concatenate([[0], np.cumsum([c.n_ports for c in env.charging_stations])])   # cumulative port numbering (ev2gym_env.py:364-385)
    rng = np.random.default_rng(1000 + seed)
    lo = -1.0 if env.config['v2g_enabled'] else 0.0
    if policy == "ones":
        act = np.ones((T, P))
    elif policy == "neg":
        act = -np.ones((T, P))
    elif policy == "rand":
        act = rng.uniform(lo, 1.0, (T, P))
    elif policy == "wild":      # outside the action box -> normalisation / clamp path (ev_charger.py:143-149)
        act = rng.uniform(-1.6 if lo < 0 else 0.0, 1.6, (T, P))
    elif policy == "mixed":     # many exact zeros and sign flips (cycle counter, a==0 branch)
        act = rng.uniform(lo, 1.0, (T, P)) * (rng.random((T, P)) < 0.7)
        act[rng.random((T, P)) < 0.1] = 1.0
    elif policy.startswith("agent:"):
        # actions chosen, step by step, by one of the reference's env-reading heuristics (baselines/heuristics.py:7-267); the
        # fixture records what the agent chose.  The restated agent of ev2gym_amd (same name) is driven on the SAME reference env
        # in lockstep and must choose the same actions bit for bit: that pins the restatement here, the GPU test pins the facade.
        import ev2gym.baselines.heuristics as H
        import ev2gym_amd.baselines.heuristics as MINE
        agent, mine = getattr(H, policy[6:])(env=env), getattr(MINE, policy[6:])(env=env)
        act = np.zeros((T, P))
    else:
        raise ValueError(policy)
    nT = T if steps is None else steps
    D = len(obs0)
    trj = dict(
        trj_obs=np.zeros((nT + 1, D)), trj_reward=np.zeros(nT), trj_done=np.zeros(nT, np.uint8),
        trj_mask=np.zeros((nT, P), np.uint8), trj_act_after=np.zeros((nT, P)),
        trj_cap=np.full((nT, P), np.nan), trj_energy=np.full((nT, P), np.nan),
        trj_current=np.full((nT, P), np.nan), trj_tot_e=np.full((nT, P), np.nan),
        trj_req_e=np.full((nT, P), np.nan), trj_prev_power=np.full((nT, P), np.nan),
        trj_cycles=np.full((nT, P), -1, np.int32),
        trj_cs_power=np.zeros((nT, C)), trj_cs_amps=np.zeros((nT, C)), trj_cs_profits=np.zeros((nT, C)),
        trj_cs_e_ch=np.zeros((nT, C)), trj_cs_e_dis=np.zeros((nT, C)),
        trj_tr_power=np.zeros((nT, R)), trj_tr_amps=np.zeros((nT, R)), trj_tr_overload=np.zeros((nT, R)),
        trj_n_departed=np.zeros(nT, np.int32), trj_sat_sum=np.zeros(nT),
        trj_dep_port=np.full((nT, P), -1, np.int32), trj_dep_score=np.full((nT, P), np.nan),
    )
    trj["trj_obs"][0] = obs0
    info = None
    for t in range(nT):
        if policy.startswith("agent:"):
            act[t] = agent.get_action(env)
            assert np.array_equal(act[t], mine.get_action(env)), (policy, t)
        a = act[t].copy()
        obs, rew, done, trunc, info = env.step(a)
        trj["trj_obs"][t + 1] = obs
        trj["trj_reward"][t] = rew
        trj["trj_done"][t] = done
        trj["trj_mask"][t] = info["action_mask"].astype(np.uint8)
        trj["trj_act_after"][t] = a
        for i, cs in enumerate(env.charging_stations):
            trj["trj_cs_power"][t, i] = cs.current_power_output
            trj["trj_cs_amps"][t, i] = cs.current_total_amps
            trj["trj_cs_profits"][t, i] = cs.total_profits
            trj["trj_cs_e_ch"][t, i] = cs.total_energy_charged
            trj["trj_cs_e_dis"][t, i] = cs.total_energy_discharged
            for j, ev in enumerate(cs.evs_connected):
                if ev is not None:
                    p = base[i] + j
                    trj["trj_cap"][t, p] = ev.current_capacity
                    trj["trj_energy"][t, p] = ev.current_energy
                    trj["trj_current"][t, p] = ev.actual_current
                    trj["trj_tot_e"][t, p] = ev.total_energy_exchanged
                    trj["trj_req_e"][t, p] = ev.required_energy
                    trj["trj_prev_power"][t, p] = ev.previous_power
                    trj["trj_cycles"][t, p] = ev.charging_cycles
        for i, tr in enumerate(env.transformers):
            trj["trj_tr_power"][t, i] = tr.current_power
            trj["trj_tr_amps"][t, i] = tr.current_amps
            trj["trj_tr_overload"][t, i] = env.tr_overload[i, t]
        trj["trj_n_departed"][t] = len(env.departing_evs)
        for k, ev in enumerate(env.departing_evs):
            trj["trj_dep_port"][t, k] = base[ev.location] + ev.id
            trj["trj_dep_score"][t, k] = ev.get_user_satisfaction()
            trj["trj_sat_sum"][t] += ev.get_user_satisfaction()
    trj["trj_usage"] = np.array(env.current_power_usage[:nT])
    trj["trj_potential"] = np.array(env.charge_power_potential[:nT])
    # per-session results at the end (env.EVs is in spawn order == profile order, ev2gym_env.py:401-414)
    S_ = len(env.EVs_profiles)
    trj["trj_ev_port"] = np.full(S_, -1, np.int32)
    trj["trj_ev_final_cap"] = np.full(S_, np.nan)
    trj["trj_ev_afap"] = np.full(S_, np.nan)
    for k, ev in enumerate(env.EVs):
        trj["trj_ev_port"][k] = base[ev.location] + ev.id
        trj["trj_ev_final_cap"][k] = ev.current_capacity
        trj["trj_ev_afap"][k] = ev.max_energy_AFAP
    if nT == T:
        trj["trj_stats"] = np.array([float(info[k]) for k in STAT_KEYS])
    out = dict(scn)
    out.update(trj)
    out.update(extra or {})
    out["act"] = act[:nT]
    out["case"] = np.array([name, os.path.basename(config), state_fn, reward_fn, str(seed), policy])
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    occ = trj["trj_mask"].mean()
    print(f"{name:28s} P={P:5d} R={R:3d} D={D:5d} S={S_:4d} occ={occ:.3f} "
          f"size={os.path.getsize(path)/1024:.0f} KB", flush=True)


REPLAY_TENSORS = ("u", "ev_arrival", "t_dep", "energy_at_arrival", "ev_max_energy", "ev_max_ch_power",
                  "ev_max_dis_power", "ev_des_energy", "port_max_charge_current", "port_min_charge_current",
                  "port_max_discharge_current", "port_min_discharge_current", "voltages", "cs_transformer",
                  "charge_prices", "discharge_prices", "power_setpoints")


def run_replay_case(name, config, state_fn, reward_fn, seed, policy):
    """Replay round trip (replay.py:10-174, ev2gym_env.py:102-116): episode A is recorded with save_replay=True,
    env B is constructed from the pickle and driven with `policy`; the fixture holds the pickle bytes (data the
    reference wrote), B's scenario / trajectory and the tensors EvCityReplay derived."""
    import glob
    import pickle
    import shutil
    from ev2gym.models.ev2gym_env import EV2Gym
    import ev2gym.rl_agent.state as S
    import ev2gym.rl_agent.reward as RW
    rdir = f"/tmp/ev2g_replay/{name}/"
    shutil.rmtree(rdir, ignore_errors=True)
    os.makedirs(rdir)
    kw = dict(config_file=config, state_function=getattr(S, state_fn), reward_function=getattr(RW, reward_fn))
    env_a = EV2Gym(seed=seed, generate_rnd_game=True, save_replay=True, replay_save_path=rdir, **kw)
    env_a.reset(seed=seed)
    for _ in range(env_a.simulation_length):
        env_a.step(np.ones(env_a.number_of_ports))
    pkl = glob.glob(rdir + "*.pkl")[0]
    blob = open(pkl, "rb").read()
    rep = pickle.loads(blob)
    extra = {"replay_pkl": np.frombuffer(blob, np.uint8)}
    for k in REPLAY_TENSORS:
        extra["rep_" + k] = np.asarray(getattr(rep, k), float)
    env_b = EV2Gym(load_from_replay_path=pkl, **kw)
    run_case(name, config, state_fn, reward_fn, seed, policy, env=env_b, extra=extra)


def run_back_to_back_case(name, yml, over, state_fn, reward_fn, seed, policy):
    """Back-to-back sessions (the next EV plugs in at the end of the very step its predecessor leaves in).  The reference's spawner keeps a
    gap between two sessions of a port, so no scenario it generates holds this case; a replayed scenario may.  A scenario drawn by
    ev2gym_amd's generator has its stays extended up to the step before the port's next arrival, is written as a replay file
    (ev2gym_amd.replay.write_replay) and handed to the reference, which loads and steps it: the fixture holds the scenario as the
    REFERENCE env carries it and the trajectory the reference produced."""
    import tempfile
    from ev2gym.models.ev2gym_env import EV2Gym
    import ev2gym.rl_agent.state as S
    import ev2gym.rl_agent.reward as RW
    from ev2gym_amd.config import gen_config_from_yaml, load_yaml
    from ev2gym_amd.replay import write_replay
    from ev2gym_amd.scenario import resolve_ports
    from ev2gym_amd.scenario_gen import generate
    mine = os.path.join(os.path.dirname(HERE), "ev2gym_amd", "example_config_files", os.path.basename(yml))
    batch = generate(gen_config_from_yaml({**load_yaml(mine), **over}, 1, seed))
    a, port, last, n = batch.arrays, resolve_ports(batch), {}, 0
    for s in range(batch.n_sessions):
        if port[s] in last:
            n += int(a["ev_t_dep"][last[port[s]]] != a["ev_t_arr"][s] - 1)
            a["ev_t_dep"][last[port[s]]] = a["ev_t_arr"][s] - 1
        last[port[s]] = s
    assert n > 0, "no port with two sessions: pick another seed"
    path = write_replay(os.path.join(tempfile.mkdtemp(), f"replay_sim_{name}.pkl"), batch)
    cfg = _yaml_variant(yml, over, name)
    env = EV2Gym(config_file=cfg, load_from_replay_path=path, state_function=getattr(S, state_fn), reward_function=getattr(RW, reward_fn))
    run_case(name, cfg, state_fn, reward_fn, seed, policy, env=env, extra={"b2b_sessions_extended": np.array(n)})


def main():
    import_reference()
    base = "ev2gym/example_config_files/"
    ppl = base + "V2GProfitPlusLoads.yaml"
    pst = base + "PublicPST.yaml"
    vmax = base + "V2GProfitMax.yaml"
    PPL = ("V2G_profit_max_loads", "ProfitMax_TrPenalty_UserIncentives")
    PST = ("PublicPST", "SquaredTrackingErrorReward")
    VMX = ("V2G_profit_max", "profit_maximization")
    only = set(sys.argv[1:])
    cases = []
    for seed, pol in [(1, "ones"), (2, "rand"), (3, "neg"), (4, "rand"), (5, "mixed"), (6, "wild")]:
        cases.append((f"v2gppl_{pol}_s{seed}", ppl, *PPL, seed, pol, None))
    for seed, pol in [(1, "ones"), (2, "rand"), (3, "rand"), (4, "mixed"), (5, "wild")]:
        cases.append((f"pst_{pol}_s{seed}", pst, *PST, seed, pol, None))
    # ceil-to-0.01 boundary trap: homogeneous specs, eta=1, transition_soc=1 (SURVEY.md §7 hard parts)
    homog = _yaml_variant(vmax, {"heterogeneous_ev_specs": False}, "v2gmax_homog")
    cases.append(("v2gmax_homog_ones_s1", homog, *VMX, 1, "ones", None))
    cases.append(("v2gmax_homog_rand_s2", homog, *VMX, 2, "rand", None))
    cases.append(("v2gmax_het_rand_s3", vmax, *VMX, 3, "rand", None))
    # homogeneous two-stage model with non-zero minimum currents / powers (gates of ev_charger.py:168-186, ev.py:151-154)
    gates = _yaml_variant(ppl, {"heterogeneous_ev_specs": False,
                                "charging_station": {"min_charge_current": 6, "min_discharge_current": -6},
                                "ev": {"transition_soc": 0.8, "min_ac_charge_power": 5, "min_discharge_power": -5,
                                       "charge_efficiency": 0.93, "discharge_efficiency": 0.91, "ev_phases": 1}},
                          "v2gppl_gates")
    cases.append(("v2gppl_gates_rand_s7", gates, *PPL, 7, "rand", None))
    cases.append(("v2gppl_gates_mixed_s8", gates, *PPL, 8, "mixed", None))
    # BASELINE cfg2 shape: 50 chargers
    c50 = _yaml_variant(ppl, {"number_of_charging_stations": 50}, "v2gppl_c50")
    cases.append(("v2gppl_c50_rand_s9", c50, *PPL, 9, "rand", None))
    # two ports per charger: normalisation + first-free port assignment
    p2 = _yaml_variant(ppl, {"number_of_charging_stations": 12, "number_of_ports_per_cs": 2}, "v2gppl_p2")
    cases.append(("v2gppl_p2_wild_s10", p2, *PPL, 10, "wild", None))
    cases.append(("v2gppl_p2_rand_s11", p2, *PPL, 11, "rand", None))
    p3 = _yaml_variant(pst, {"number_of_charging_stations": 7, "number_of_ports_per_cs": 3}, "pst_p3")
    cases.append(("pst_p3_rand_s12", p3, *PST, 12, "rand", None))
    # multi-transformer round-robin map (loaders.py:494-498): BASELINE cfg4 shape, scaled down
    r5 = _yaml_variant(ppl, {"number_of_charging_stations": 60, "number_of_transformers": 5}, "v2gppl_c60_r5")
    cases.append(("v2gppl_c60r5_rand_s13", r5, *PPL, 13, "rand", None))
    r3 = _yaml_variant(ppl, {"number_of_charging_stations": 10, "number_of_transformers": 3}, "v2gppl_c10_r3")
    cases.append(("v2gppl_c10r3_mixed_s14", r3, *PPL, 14, "mixed", None))
    big = _yaml_variant(ppl, {"number_of_charging_stations": 1000, "number_of_transformers": 50}, "v2gppl_c1000_r50")
    cases.append(("v2gppl_c1000r50_rand_s15", big, *PPL, 15, "rand", 16))
    # other timescales: 60/dt = 12 (not a power of two -> the engine's division path) and 2 (multiplication path)
    ts5 = _yaml_variant(ppl, {"timescale": 5}, "v2gppl_ts5")
    cases.append(("v2gppl_ts5_rand_s16", ts5, *PPL, 16, "rand", None))
    ts30 = _yaml_variant(pst, {"timescale": 30}, "pst_ts30")
    cases.append(("pst_ts30_rand_s17", ts30, *PST, 17, "rand", None))
    # combinations the single-feature cases above do not cross: multi-port chargers x several transformers x other timescales
    x1 = _yaml_variant(pst, {"number_of_charging_stations": 12, "number_of_ports_per_cs": 3, "number_of_transformers": 2,
                             "timescale": 5, "heterogeneous_ev_specs": False}, "pst_c12p3r2_ts5")
    cases.append(("pst_c12p3r2ts5_rand_s31", x1, *PST, 31, "rand", None))
    x2 = _yaml_variant(ppl, {"number_of_charging_stations": 9, "number_of_ports_per_cs": 3, "number_of_transformers": 3},
                       "v2gppl_c9p3r3")
    cases.append(("v2gppl_c9p3r3_wild_s32", x2, *PPL, 32, "wild", None))
    x3 = _yaml_variant(vmax, {"number_of_charging_stations": 28, "number_of_ports_per_cs": 2, "number_of_transformers": 3,
                              "timescale": 30}, "v2gmax_c28p2r3_ts30")
    cases.append(("v2gmax_c28p2r3ts30_rand_s33", x3, *VMX, 33, "rand", None))
    # topology files (loaders.py:259-276, 312-340): chargers with their own port counts, current limits, voltage and phases
    tp1 = _topology_file("het_a", [(60, [(3, 32, -32, 400, 3), (2, 16, -16, 230, 3)]),
                                   (40, [(2, 32, -32, 400, 1), (1, 16, -16, 230, 3), (1, 32, -32, 400, 3)])])
    t1 = _yaml_variant(ppl, {"charging_network_topology": tp1, "spawn_multiplier": 10}, "v2gppl_topo_het_a")
    cases.append(("topo_v2gppl_het_rand_s41", t1, *PPL, 41, "rand", None))
    cases.append(("topo_v2gppl_het_wild_s42", t1, *PPL, 42, "wild", None))
    tp2 = _topology_file("het_b", [(100, [(4, 32, 0, 400, 3), (2, 16, 0, 230, 1), (2, 32, 0, 400, 3), (1, 16, 0, 400, 3), (1, 32, 0, 400, 3), (1, 32, 0, 230, 3)])])
    t2 = _yaml_variant(pst, {"charging_network_topology": tp2, "spawn_multiplier": 10}, "pst_topo_het_b")
    cases.append(("topo_pst_het_mixed_s43", t2, *PST, 43, "mixed", None))
    for c in cases:
        if only and c[0] not in only:
            continue
        run_case(*c)
    # reward plugins that are not fused in the kernel (evaluated on the host through the facade): plugin_*.npz
    for c in [("plugin_pst_sqtr_rand_s23", pst, "PublicPST", "SqTrError_TrPenalty_UserIncentives", 23, "rand", None),
              ("plugin_pst_surplus_rand_s24", pst, "PublicPST", "MinimizeTrackerSurplusWithChargeRewards", 24, "rand", None),
              ("plugin_pst_idlepen_mixed_s25", pst, "PublicPST", "SquaredTrackingErrorRewardWithPenalty", 25, "mixed", None),
              ("plugin_v2gppl_sqtr_rand_s26", ppl, "V2G_profit_max_loads", "SqTrError_TrPenalty_UserIncentives", 26, "rand", None),
              ("plugin_pst_simple_rand_s27", pst, "PublicPST", "SimpleReward", 27, "rand", None),
              ("plugin_v2gppl_costs_rand_s28", ppl, "V2G_profit_max_loads", "V2G_costs_simple", 28, "rand", None),
              ("plugin_v2gppl_profitmax_rand_s29", ppl, "V2G_profit_max_loads", "V2G_profitmax", 29, "rand", None),
              ("plugin_v2gmax_profitmax_neg_s30", vmax, "V2G_profit_max", "V2G_profitmax", 30, "neg", None),
              ("plugin_v2gppl_c10r3_sqtr_mixed_s31", r3, "V2G_profit_max_loads", "SqTrError_TrPenalty_UserIncentives", 31, "mixed", None),
              ("plugin_pst_p3_idlepen_mixed_s32", p3, "PublicPST", "SquaredTrackingErrorRewardWithPenalty", 32, "mixed", None),
              ("plugin_v2gppl_pmaxv2_rand_s33", ppl, "V2G_profit_max_loads", "V2G_profitmaxV2", 33, "rand", None),
              ("plugin_v2gppl_p2_pmaxv2_mixed_s34", p2, "V2G_profit_max_loads", "V2G_profitmaxV2", 34, "mixed", None),
              ("plugin_pst_pstpmaxv2_rand_s35", pst, "PublicPST", "pst_V2G_profitmaxV2", 35, "rand", None),
              ("plugin_v2gmax_c28p2r3_pstpmaxv2_s36", x3, "V2G_profit_max", "pst_V2G_profitmaxV2", 36, "rand", None)]:
        if only and c[0] not in only:
            continue
        run_case(*c)
    # env-reading heuristic agents of the reference (heuristics.py:7-267) choosing the actions: agent_*.npz
    des80 = _yaml_variant(ppl, {"ev": {"desired_capacity": 0.8}}, "v2gppl_des80")
    for c in [("agent_roundrobin_pst_s61", pst, *PST, 61, "agent:RoundRobin", None),
              ("agent_roundrobin_pst_p3_s62", p3, *PST, 62, "agent:RoundRobin", None),
              ("agent_calap_v2gppl_s63", ppl, *PPL, 63, "agent:ChargeAsLateAsPossible", None),
              ("agent_calap_pst_s64", pst, *PST, 64, "agent:ChargeAsLateAsPossible", None),
              ("agent_afapdes_v2gppl_des80_s65", des80, *PPL, 65, "agent:ChargeAsFastAsPossibleToDesiredCapacity", None),
              ("agent_afapdes_v2gppl_p2_s66", p2, *PPL, 66, "agent:ChargeAsFastAsPossibleToDesiredCapacity", None)]:
        if only and c[0] not in only:
            continue
        run_case(*c)
    # replay files: the reference's on-disk scenario format
    for c in [("replay_v2gppl_p2_rand_s21", p2, *PPL, 21, "rand"), ("replay_pst_rand_s22", pst, *PST, 22, "rand")]:
        if only and c[0] not in only:
            continue
        run_replay_case(*c)
    # back-to-back sessions on a port (replayed scenarios only: the reference's spawner keeps a gap)
    for c in [("b2b_v2gppl_public_rand_s53", ppl, {"number_of_charging_stations": 6, "spawn_multiplier": 10, "scenario": "public"}, *PPL, 53, "rand"),
              ("b2b_v2gmax_p2_mixed_s53", vmax, {"number_of_charging_stations": 4, "number_of_ports_per_cs": 2, "spawn_multiplier": 10, "scenario": "public"},
               *VMX, 53, "mixed"),
              ("b2b_pst_rand_s53", pst, {"number_of_charging_stations": 5, "spawn_multiplier": 10}, *PST, 53, "rand")]:
        if only and c[0] not in only:
            continue
        run_back_to_back_case(*c)


if __name__ == "__main__":
    main()
