"""Single-env facade with the reference's object graph (`ev2gym.models.ev2gym_env.EV2Gym`).

`EV2Gym(config_file, ..., state_function, reward_function, seed)` keeps the reference constructor
(ev2gym_env.py:38-56), `reset()` (:243-331) and `step(actions)` (:333-447) and exposes the attribute graph that
heuristics, state functions and reward functions read: `env.charging_stations[i].evs_connected[j]
.get_soc()`, `env.transformers[k].get_power_limits(...)`, `env.current_power_usage`, ... as READ-ONLY views over a
host copy of one env's device state (`ev2g_peek`).  The maths still runs on the GPU (a 1-env batch through the same
HIP kernel).  Built-in state / reward functions are fused; any other callable is evaluated here on the host with
this facade as its `env` argument -- an explicit slow path, never a silent one.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import numpy as np

from . import _abi
from .config import gen_config_from_yaml, load_yaml
from .engine import Engine
from .scenario import ScenarioBatch
from .scenario_gen import generate
from .gym_compat import Box, EnvBase
from .vec_env import _kind


class EVView:
    """Read-only view of one EV session (models/ev.py:68-113)."""

    def __init__(self, env, k):
        a = env._arr
        self._env, self._k = env, k
        self.location = int(a["ev_cs"][k])
        self.time_of_arrival = int(a["ev_t_arr"][k])
        self.time_of_departure = int(a["ev_t_dep"][k])
        self.battery_capacity_at_arrival = float(a["ev_cap0"][k])
        self.battery_capacity = float(a["ev_B"][k])
        self.desired_capacity = float(a["ev_desired"][k])
        self.min_battery_capacity = float(a["ev_minB"][k])
        self.min_emergency_battery_capacity = float(a["ev_min_emerg"][k])
        self.max_ac_charge_power = float(a["ev_pac_max"][k])
        self.min_ac_charge_power = float(a["ev_pac_min"][k])
        self.max_discharge_power = float(a["ev_pdis_max"][k])
        self.min_discharge_power = float(a["ev_pdis_min"][k])
        self.transition_soc = float(a["ev_ts"][k])
        self.transition_soc_multiplier = float(a["ev_tsm"][k])
        self.ev_phases = int(a["ev_phases"][k])
        lut = int(a["ev_lut"][k])
        if lut >= 0:
            self.charge_efficiency = {i: float(v) for i, v in enumerate(a["lut"][lut])}
            self.discharge_efficiency = dict(self.charge_efficiency)
        else:
            self.charge_efficiency = float(a["ev_eta_ch"][k])
            self.discharge_efficiency = float(a["ev_eta_dis"][k])
        self.timescale = env.timescale

    # dynamic fields come from the snapshot of the port the EV sits on
    def _port(self):
        return int(self._env._snap()["session_port"][self._k])

    @property
    def id(self):
        return self._port() - int(self._env._port_base[self.location])

    def _dyn(self, key, default=0.0):
        s = self._env._snap()
        p = self._port()
        if s["port_session"][p] == self._k:
            return float(s[key][p])
        return default

    @property
    def current_capacity(self):
        s = self._env._snap()
        p = self._port()
        if s["port_session"][p] == self._k:
            return float(s["port_capacity"][p])
        if self._env.current_step > self.time_of_departure:  # departed: capacity at departure
            return float(s["session_final_cap"][self._k])
        return self.battery_capacity_at_arrival

    current_energy = property(lambda self: self._dyn("port_energy"))
    actual_current = property(lambda self: self._dyn("port_current"))
    total_energy_exchanged = property(lambda self: self._dyn("port_total_energy"))
    previous_power = property(lambda self: self._dyn("port_prev_power"))
    charging_cycles = property(lambda self: int(self._dyn("port_cycles", 0)))

    @property
    def required_energy(self):  # ev.py:100,353,399: B - cap0 - (energy exchanged), up to rounding
        return self.battery_capacity - self.battery_capacity_at_arrival - self.total_energy_exchanged

    @property
    def max_energy_AFAP(self):
        return float(self._env._snap()["session_afap"][self._k])

    def get_soc(self):  # ev.py:223-229
        return self.current_capacity / self.battery_capacity

    def get_user_satisfaction(self):  # ev.py:204-214
        c = self.current_capacity
        return c / self.desired_capacity if c < self.desired_capacity - 0.001 else 1

    def is_departing(self, timestep):  # ev.py:191-202
        return None if timestep < self.time_of_departure else self.get_user_satisfaction()


class ChargerView:
    """Read-only view of one charging station (models/ev_charger.py:41-94)."""

    def __init__(self, env, i):
        a = env._arr
        self._env, self.id = env, i
        self.n_ports = int(a["cs_n_ports"][i])
        self._base = int(env._port_base[i])
        self.connected_transformer = int(a["cs_transformer"][i])
        self.connected_bus = self.connected_transformer
        self.min_charge_current = float(a["cs_min_charge_current"][i])
        self.max_charge_current = float(a["cs_max_charge_current"][i])
        self.min_discharge_current = float(a["cs_min_discharge_current"][i])
        self.max_discharge_current = float(a["cs_max_discharge_current"][i])
        self.voltage = float(a["cs_voltage"][i])
        self.phases = int(a["cs_phases"][i])
        self.charger_type = "AC"
        self.timescale = env.timescale

    @property
    def evs_connected(self):
        s = self._env._snap()
        out = []
        for j in range(self.n_ports):
            k = int(s["port_session"][self._base + j])
            out.append(self._env._ev(k) if k >= 0 else None)
        return out

    @property
    def n_evs_connected(self):
        return sum(e is not None for e in self.evs_connected)

    @property
    def current_step(self):
        return self._env.current_step

    def _cs(self, key):
        v = self._env._snap()[key][self.id]
        if np.isnan(v):
            raise AttributeError(f"{key} needs EV2Gym(..., log_cs_history=True)")
        return float(v)

    current_power_output = property(lambda self: self._cs("cs_power"))
    current_total_amps = property(lambda self: self._cs("cs_amps"))
    total_profits = property(lambda self: self._cs("cs_profits"))
    total_energy_charged = property(lambda self: self._cs("cs_energy_charged"))
    total_energy_discharged = property(lambda self: self._cs("cs_energy_discharged"))

    def get_max_power(self):  # ev_charger.py:251-252
        return self.max_charge_current * self.voltage * math.sqrt(self.phases) / 1000

    def get_min_charge_power(self):
        return self.min_charge_current * self.voltage * math.sqrt(self.phases) / 1000

    def get_min_power(self):
        return self.max_discharge_current * self.voltage * math.sqrt(self.phases) / 1000


class TransformerView:
    """Read-only view of one transformer (models/transformer.py)."""

    def __init__(self, env, r):
        a = env._arr
        self._env, self.id = env, r
        self.max_power = a["tr_max_power"][0, r]
        self.min_power = a["tr_min_power"][0, r]
        self.inflexible_load = a["tr_inflexible_load"][0, r]
        self.solar_power = a["tr_solar_power"][0, r]
        self._lf0 = a["tr_load_forecast"][0, r]
        self._pvf0 = a["tr_pv_forecast"][0, r]
        self.dr_events = [dict(event_start_step=int(e[0]), event_end_step=int(e[1]), capacity_percentage=float(e[2]))
                          for e in a["tr_dr"][0, r][:int(a["tr_n_dr"][0, r])]]
        self.steps_ahead = int(a["tr_steps_ahead"][0, r])
        self.voltage = float(a["cs_voltage"][0]) * math.sqrt(int(a["cs_phases"][0]))
        self.max_current = self.max_power * 1000 / self.voltage
        self.min_current = -self.max_current
        self.cs_ids = np.where(np.asarray(a["cs_transformer"]) == r)[0]
        self.simulation_length = env.simulation_length

    @property
    def current_step(self):  # the step the last Transformer.reset(step) was called with (transformer.py:258-262)
        return max(self._env.current_step - 1, 0)

    @property
    def current_power(self):
        return float(self._env._snap()["tr_power"][self.id])

    @property
    def current_amps(self):
        return self.current_power * 1000 / self.voltage

    def is_overloaded(self):  # transformer.py:276-290
        s = self.current_step
        return self.current_power > self.max_power[s] + 0.0001 or self.current_power < self.min_power[s] - 0.0001

    def get_how_overloaded(self):  # transformer.py:292-302
        return abs(self.current_power - self.max_power[self.current_step]) if self.is_overloaded() else 0

    def get_power_limits(self, step, horizon):  # transformer.py:142-171
        limit = max(self.max_power)
        known = limit * np.ones(horizon)
        for ev in self.dr_events:
            if step + self.steps_ahead >= ev["event_start_step"] and ev["event_end_step"] >= step:
                red = limit - limit * ev["capacity_percentage"] / 100
                if step > ev["event_start_step"]:
                    known[:ev["event_end_step"] - step] = red
                else:
                    known[abs(ev["event_start_step"] - step):abs(ev["event_end_step"] - step)] = red
        return known

    def get_load_pv_forecast(self, step, horizon):  # transformer.py:173-188 (the in-place overwrite, replayed)
        T = self.simulation_length
        observed = min(self._env._max_obs_step, T - 1)   # forecast[s] := actual[s] for every observed s
        lf, pvf = self._lf0.copy(), self._pvf0.copy()
        lf[:observed + 1] = self.inflexible_load[:observed + 1]
        pvf[:observed + 1] = self.solar_power[:observed + 1]
        if step < T:
            lf[step], pvf[step] = self.inflexible_load[step], self.solar_power[step]
        l, p = lf[step:step + horizon], pvf[step:step + horizon]
        if len(l) < horizon:
            l = np.append(l, np.ones(horizon - len(l)) * lf[-1])
            p = np.append(p, np.ones(horizon - len(p)) * pvf[-1])
        return l, p


class EV2Gym(EnvBase):
    """Drop-in single-env `EV2Gym` running on the HIP engine (one env per handle); a `gymnasium.Env` when gymnasium is installed
    (ev2gym_env.py:36), registered as `EV2Gym-v1` (gym_compat.py)."""
    metadata = {"render_modes": []}   # plots / rendering are outside the accelerated path

    def __init__(self, config_file=None, load_from_replay_path=None, replay_save_path='./replay/', generate_rnd_game=True,
                 seed=None, save_replay=False, save_plots=False, state_function="PublicPST",
                 reward_function="SquaredTrackingErrorReward", cost_function=None, eval_mode="Normal",
                 lightweight_plots=False, empty_ports_at_end_of_simulation=True, extra_sim_name=None, verbose=False,
                 render_mode=None, scenario: Optional[ScenarioBatch] = None, device: int = 0,
                 log_cs_history: bool = True, data_dir=None):
        if save_plots or render_mode:
            raise NotImplementedError("plots and rendering are outside the accelerated path (SURVEY.md §2)")
        self.save_replay, self.replay_path, self.extra_sim_name = bool(save_replay), replay_save_path, extra_sim_name
        self.load_from_replay_path = load_from_replay_path
        if scenario is None and load_from_replay_path is not None:   # ev2gym_env.py:102-116
            from .replay import load_replay
            v2g = bool(load_yaml(config_file)["v2g_enabled"]) if config_file is not None else None
            scenario = load_replay(load_from_replay_path, v2g_enabled=v2g)
        if scenario is None:
            assert config_file is not None, "Please provide a config file!!!"   # ev2gym_env.py:64
            self.config = load_yaml(config_file)
            if data_dir is not None:   # an EV2Gym install's ev2gym/data: its spawn tables / PV year / EV-spec files instead of the fitted stand-ins
                self.config = {**self.config, "data_dir": str(data_dir)}
            self.seed = np.random.randint(0, 1000000) if seed is None else seed
            scenario = generate(gen_config_from_yaml(self.config, 1, self.seed))
        else:
            self.config, self.seed = None, seed
            assert scenario.n_envs == 1
        self.state_function, self.reward_function, self.cost_function = state_function, reward_function, cost_function
        sk = _kind(state_function, _abi.STATE_KINDS, "state_function")
        rk = _kind(reward_function, _abi.REWARD_KINDS, "reward_function")
        self._host_state = sk is None     # user-defined callables: evaluated here on the host
        self._host_reward = rk is None
        flags = _abi.FLAG_LOG_CS_HISTORY if (log_cs_history or self._host_reward or cost_function) else 0
        flags |= _abi.FLAG_LOG_SOC
        self.engine = Engine(scenario, rk if rk is not None else 2, sk if sk is not None else 2, device=device, flags=flags)
        self._d = None
        # the reference keeps a calendar date for plots / week-day logic (ev2gym_env.py:262-296); scenarios here carry none,
        # so the date is the reference's base day at the configured start hour
        import datetime
        c = self.config or {}
        self.sim_starting_date = datetime.datetime(2022, 1, 1, int(c.get("hour", 5)), int(c.get("minute", 0)))
        self.sim_date = self.sim_starting_date
        self.sim_name = (extra_sim_name or "") + "sim_" + datetime.datetime.now().strftime("%Y_%m_%d_%f")   # ev2gym_env.py:163-189
        if load_from_replay_path is not None:                                 # ev2gym_env.py:106-108
            self.sim_name = os.path.basename(str(load_from_replay_path)).split("replay_")[-1].split(".")[0] + "_replay"
        if self.save_replay:
            os.makedirs(self.replay_path, exist_ok=True)                      # ev2gym_env.py:213-214
        self._bind_scenario(scenario)
        low = -1.0 if self.v2g_enabled else 0.0
        self.action_space = Box(low, 1.0, (self.engine.P,))
        self.departing_evs = []
        self.total_reward = 0.0
        self._scenario_seed = self.seed if self.config is not None else None
        self._episodes = 0
        self.resample_on_reset = True     # False: reset() re-arms the current scenario (round-1 behaviour)
        self._seed_rng = np.random.default_rng(self.seed if isinstance(self.seed, (int, np.integer)) else None)
        self.reset()
        self.observation_space = Box(-np.inf, np.inf, (len(self._last_obs),))
        self.observation_mask = np.zeros(self.engine.P)

    def _bind_scenario(self, scenario):
        """Everything the facade derives from the loaded scenario (also after reset(seed=...) drew a new one)."""
        self._batch = scenario
        self._arr = scenario.arrays
        e = self.engine
        self.simulation_length, self.timescale = e.T, scenario.timescale
        self.cs, self.number_of_ports, self.number_of_ports_per_cs = e.C, e.P, scenario.ports_per_charger
        self._port_base = scenario.port_base          # cumulative port numbering (ev2gym_env.py:364-385)
        self.number_of_transformers = e.R
        self.v2g_enabled = scenario.v2g_enabled
        self.cs_transformers = [int(x) for x in self._arr["cs_transformer"]]
        self.charge_prices = np.tile(self._arr["charge_price"][0], (e.C, 1))      # [C,T] like the reference
        self.discharge_prices = np.tile(self._arr["discharge_price"][0], (e.C, 1))
        self.power_setpoints = self._arr["power_setpoints"][0]
        self.charging_stations = [ChargerView(self, i) for i in range(e.C)]
        self.transformers = [TransformerView(self, r) for r in range(e.R)]
        self._evs = {}
        self.EVs_profiles = [self._ev(k) for k in range(scenario.n_sessions)]
        if self._d is None:
            # what a step hands down -- observation, reward, done, mask -- in ONE device block, mirrored by one page-locked host block: one copy per step
            # instead of three into pageable arrays (round 6; with ev2g_peek's staging block: 2.7 k -> 6 k steps/s of the single-env loop)
            o_rew = e.D * 8
            o_done, o_mask = o_rew + 8, o_rew + 16
            nbytes = (o_mask + e.P + 15) & ~15
            blk = e.empty((nbytes,), np.uint8)
            hb = e.pinned((nbytes,), np.uint8)
            hb[:] = 0
            self._d = dict(act=e.empty((1, e.P)), blk=blk, obs=blk.ptr, rew=blk.ptr + o_rew, done=blk.ptr + o_done, mask=blk.ptr + o_mask, nbytes=nbytes,
                           h_blk=hb, h_obs=hb[:o_rew].view(np.float64), h_rew=hb[o_rew:o_rew + 8].view(np.float64), h_done=hb[o_done:o_done + 1],
                           h_mask=hb[o_mask:o_mask + e.P], h_act=e.pinned((1, e.P), np.float64), stale=True)

    # ---- snapshot plumbing ------------------------------------------------------------------------
    def _ev(self, k):
        if k not in self._evs:
            self._evs[k] = EVView(self, k)
        return self._evs[k]

    def _snap(self):
        if self._snapshot is None:
            self._snapshot = self.engine.peek(0)
        return self._snapshot

    @property
    def current_step(self):
        return self.engine.current_step

    @property
    def current_power_usage(self):
        return self._snap()["power_usage"]

    @property
    def charge_power_potential(self):
        return self._snap()["power_potential"]

    @property
    def tr_overload(self):
        return self._snap()["tr_overload"]

    @property
    def EVs(self):  # spawned so far, in spawn order (ev2gym_env.py:401-414)
        return [ev for ev in self.EVs_profiles if ev.time_of_arrival <= self.current_step]

    # ---- gym surface --------------------------------------------------------------------------------
    def reset(self, seed=None, options=None, **kwargs):
        """EV2Gym.reset() (ev2gym_env.py:243-331).  Built from a config file, every reset draws a NEW scenario like the
        reference does (new EV profiles, prices, loads, PV, demand-response events): `reset(seed=s)` the scenario of seed s,
        `reset()` the next seed of the env's own generator (itself seeded by the constructor's `seed`).  Built from explicit
        scenario tensors or a replay file, reset() re-arms that scenario (the reference's replay behaviour, :102-116)."""
        if self.config is not None:
            if seed is None and self._episodes > 0 and self.resample_on_reset:
                seed = int(self._seed_rng.integers(0, 1000000))    # ev2gym_env.py:250-253 draws its seed the same way
            if seed is not None and seed != self._scenario_seed:
                self.seed = seed
                new = generate(gen_config_from_yaml(self.config, 1, seed))
                self.engine.load(new)
                self._bind_scenario(new)
                self._scenario_seed = seed
        self._episodes += 1
        self.engine.reset(self._d["obs"])
        self._d["stale"] = True
        self._snapshot = None
        self._max_obs_step = 0
        self.done = False
        self.stats = None
        self.total_reward = 0.0
        self.departing_evs = []
        self._prev_profits = np.zeros(self.cs)
        self._last_obs = self._get_observation()
        return self._last_obs, {}

    def _get_observation(self):
        self._max_obs_step = max(self._max_obs_step, self.current_step)
        if self._host_state:
            return np.asarray(self.state_function(self), dtype=np.float64)
        self._pull()
        return self._d["h_obs"].copy()

    def _pull(self):
        """The step's (or the reset's) hand-over block, once."""
        d = self._d
        if d["stale"]:
            self.engine.memcpy_d2h(d["h_blk"], d["blk"], d["nbytes"])
            d["stale"] = False

    def step(self, actions, visualize=False):
        assert not self.done, "Episode is done, please reset the environment"   # ev2gym_env.py:343
        t = self.current_step
        occupied = self._snap()["port_session"] >= 0
        try:
            actions[~occupied] = 0      # the reference zeroes empty ports in the caller's array (ev_charger.py:139)
        except (TypeError, ValueError):
            pass
        a = np.ascontiguousarray(np.asarray(actions, dtype=np.float64).reshape(1, -1))
        assert a.shape[1] == self.number_of_ports
        np.copyto(self._d["h_act"], a)
        self._d["act"].upload(self._d["h_act"])
        self.engine.step(self._d["act"], self._d["obs"], self._d["rew"], self._d["done"], self._d["mask"])
        self._d["stale"] = True
        self.engine.check_faults()
        self._snapshot = None
        self.departing_evs = [ev for ev in self.EVs_profiles if ev.time_of_departure == t and ev.time_of_arrival <= t]
        user_satisfaction_list = [ev.get_user_satisfaction() for ev in self.departing_evs]
        invalid = int((~occupied).sum())
        total_costs = None
        if self._host_reward or self.cost_function is not None:
            prof = self._snap()["cs_profits"]
            total_costs = float((prof - self._prev_profits).sum())
            self._prev_profits = prof.copy()
        if self._host_reward:
            reward = self.reward_function(self, total_costs, user_satisfaction_list, invalid)
        else:
            self._pull()
            reward = float(self._d["h_rew"][0])
        self.total_reward += reward
        cost = self.cost_function(self, total_costs, user_satisfaction_list, invalid) if self.cost_function else None
        self._pull()
        mask = self._d["h_mask"].astype(np.float64)
        self._last_obs = self._get_observation()
        if self.current_step >= self.simulation_length:   # _check_termination ev2gym_env.py:449-496
            self.done = True
            st = self.engine.stats()[0]
            self.stats = {k: st[i] for i, k in enumerate(_abi.STAT_NAMES)}
            self.stats.update({k: 0 for k in _abi.GRID_STAT_ZEROS})
            if self._host_reward:
                self.stats["total_reward"] = self.total_reward
            self.stats["action_mask"] = mask
            self.cost = cost
            if self.save_replay:
                self._save_sim_replay()
            return self._last_obs, reward, True, False, self.stats
        return self._last_obs, reward, False, False, {"cost": cost, "action_mask": mask}

    def _save_sim_replay(self):
        """ev2gym_env.py:503-510: the finished episode as `<replay_save_path>/replay_<sim_name>.pkl`, an EvCityReplay pickle
        the reference's `EV2Gym(load_from_replay_path=...)` (and this class) loads."""
        from .replay import write_replay
        c = self.config or {}
        path = os.path.join(self.replay_path, f"replay_{self.sim_name}.pkl")
        stats = {k: v for k, v in (self.stats or {}).items() if k != "action_mask"}
        write_replay(path, self._batch, 0, run=self._snap(), stats=stats, sim_date=self.sim_starting_date,
                     scenario=str(c.get("scenario", "workplace")), heterogeneous_specs=bool(c.get("heterogeneous_ev_specs", True)),
                     sim_name=self.sim_name, replay_path=self.replay_path)
        print(f"Saving replay file at {path}")
        return path

    def set_cost_function(self, cost_function):
        self.cost_function = cost_function

    def set_reward_function(self, reward_function):
        self.reward_function = reward_function
        self._host_reward = _kind(reward_function, _abi.REWARD_KINDS, "reward_function") != self.engine.reward_kind

    def close(self):
        self.engine.close()
