"""ctypes mirrors of include/ev2g.h (the C-ABI structs).  Keep in sync with the header."""
import ctypes as C

ABI_VERSION = 4
COMM_ID_BYTES = 128   # EV2G_COMM_ID_BYTES = sizeof(ncclUniqueId)
LUT_LEN = 101
N_STATS = 17

REWARD_KINDS = {
    "ProfitMax_TrPenalty_UserIncentives": 0,   # rl_agent/reward.py:34-44
    "SquaredTrackingErrorReward": 1,           # rl_agent/reward.py:7-14
    "profit_maximization": 2,                  # rl_agent/reward.py:78-87
    "SqTrError_TrPenalty_UserIncentives": 3,   # rl_agent/reward.py:16-32
    "SquaredTrackingErrorRewardWithPenalty": 4,  # rl_agent/reward.py:46-58
    "SimpleReward": 5,                         # rl_agent/reward.py:60-65
    "MinimizeTrackerSurplusWithChargeRewards": 6,  # rl_agent/reward.py:67-76
    "V2G_costs_simple": 7,                     # rl_agent/reward.py:151-154
    "V2G_profitmax": 8,                        # rl_agent/reward.py:120-148
    "V2G_profitmaxV2": 9,                      # rl_agent/reward.py:156-211
    "pst_V2G_profitmaxV2": 10,                 # rl_agent/reward.py:278-339
}
STATE_KINDS = {
    "V2G_profit_max_loads": 0,                 # rl_agent/state.py:108-155
    "PublicPST": 1,                            # rl_agent/state.py:6-63
    "V2G_profit_max": 2,                       # rl_agent/state.py:65-106
}
STAT_NAMES = [  # get_statistics(), utilities/utils.py:84-101
    'total_ev_served', 'total_profits', 'total_energy_charged', 'total_energy_discharged',
    'average_user_satisfaction', 'power_tracker_violation', 'tracking_error',
    'energy_tracking_error', 'energy_user_satisfaction', 'std_energy_user_satisfaction',
    'min_energy_user_satisfaction', 'total_steps_min_emergency_battery_capacity_violation',
    'total_transformer_overload', 'battery_degradation', 'battery_degradation_calendar',
    'battery_degradation_cycling', 'total_reward']

# grid-simulation keys of the same dict (utils.py:103-112): constant 0 because simulate_grid is out of scope
GRID_STAT_ZEROS = ('saved_grid_energy', 'voltage_violation', 'voltage_violation_counter',
                   'voltage_violation_counter_per_step')

COST_KINDS = {
    "transformer_overload_usrpenalty_cost": 1,         # rl_agent/cost.py:8-18
    "ProfitMax_TrPenalty_UserIncentives_safety": 2,    # rl_agent/cost.py:22-27
}
AUTO_RESET_SAME = 1
AUTO_RESET_NEXT = 2

ERR_DONE = -4
ERR_OVERCURRENT = -5
FLAG_LOG_CS_HISTORY = 1
FLAG_NULL_STREAM = 2
FLAG_LOG_SOC = 4
FLAG_REFILLABLE = 8   # fixed-size session blocks per scenario: ev2g_pool_refill can re-draw scenarios on the device

_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)
_pl = C.POINTER(C.c_int64)

# (field name, element ctype) of every pointer member, in header order
BATCH_ARRAYS = [
    ("cs_min_charge_current", C.c_double), ("cs_max_charge_current", C.c_double),
    ("cs_min_discharge_current", C.c_double), ("cs_max_discharge_current", C.c_double),
    ("cs_voltage", C.c_double), ("cs_phases", C.c_int32), ("cs_transformer", C.c_int32), ("cs_n_ports", C.c_int32),
    ("charge_price", C.c_double), ("discharge_price", C.c_double), ("power_setpoints", C.c_double),
    ("tr_max_power", C.c_double), ("tr_min_power", C.c_double), ("tr_inflexible_load", C.c_double),
    ("tr_solar_power", C.c_double), ("tr_load_forecast", C.c_double), ("tr_pv_forecast", C.c_double),
    ("tr_dr", C.c_double), ("tr_n_dr", C.c_int32), ("tr_steps_ahead", C.c_int32),
    ("env_session_start", C.c_int64),
    ("ev_cs", C.c_int32), ("ev_t_arr", C.c_int32), ("ev_t_dep", C.c_int32), ("ev_phases", C.c_int32),
    ("ev_lut", C.c_int32),
    ("ev_cap0", C.c_double), ("ev_B", C.c_double), ("ev_desired", C.c_double), ("ev_minB", C.c_double),
    ("ev_min_emerg", C.c_double), ("ev_pac_max", C.c_double), ("ev_pac_min", C.c_double),
    ("ev_pdis_max", C.c_double), ("ev_pdis_min", C.c_double), ("ev_ts", C.c_double), ("ev_tsm", C.c_double),
    ("ev_eta_ch", C.c_double), ("ev_eta_dis", C.c_double), ("lut", C.c_double),
]


class ScenarioBatchC(C.Structure):
    _fields_ = [
        ("n_envs", C.c_int32), ("n_steps", C.c_int32), ("timescale", C.c_int32), ("n_chargers", C.c_int32),
        ("ports_per_charger", C.c_int32), ("n_transformers", C.c_int32), ("horizon", C.c_int32),
        ("n_dr_max", C.c_int32), ("n_lut", C.c_int32), ("reserved0", C.c_int32), ("n_sessions", C.c_int64),
    ] + [(name, C.POINTER(ct)) for name, ct in BATCH_ARRAYS]


class ConfigC(C.Structure):
    _fields_ = [("device", C.c_int32), ("reward_kind", C.c_int32), ("state_kind", C.c_int32),
                ("flags", C.c_int32), ("stream", C.c_void_p), ("cost_kind", C.c_int32), ("n_active_envs", C.c_int32)]


class StepExtrasC(C.Structure):
    _fields_ = [("cost", C.c_void_p), ("cost_step_stride", C.c_int64), ("obs_f32", C.c_void_p),
                ("obs_f32_step_stride", C.c_int64), ("actions_f32", C.c_void_p)]


class EnvViewC(C.Structure):
    _fields_ = [
        ("current_step", C.c_int32), ("n_ports", C.c_int32), ("n_chargers", C.c_int32),
        ("n_transformers", C.c_int32), ("n_steps", C.c_int32),
        ("port_capacity", _pd), ("port_energy", _pd), ("port_current", _pd), ("port_total_energy", _pd),
        ("port_required_energy", _pd), ("port_prev_power", _pd), ("port_cycles", _pi), ("port_session", _pi),
        ("cs_power", _pd), ("cs_amps", _pd), ("cs_profits", _pd), ("cs_energy_charged", _pd),
        ("cs_energy_discharged", _pd), ("tr_power", _pd), ("tr_overload", _pd), ("power_usage", _pd),
        ("power_potential", _pd), ("session_port", _pi), ("session_afap", _pd), ("session_final_cap", _pd),
    ]


# ---- scenario generator (ev2g_gen_config, include/ev2g.h) ----
GEN_INT_FIELDS = ["simulation_length", "timescale", "number_of_charging_stations", "number_of_ports_per_cs", "number_of_transformers",
                  "scenario", "simulation_days", "hour", "minute", "random_hour", "v2g_enabled", "power_setpoint_enabled",
                  "inflexible_loads", "solar_power", "demand_response", "dr_events_per_day", "dr_event_length_minutes_min",
                  "dr_event_length_minutes_max", "dr_notification_of_event_minutes", "heterogeneous_ev_specs", "fleet_with_efficiency_tables",
                  "fleet", "cs_phases", "ev_phases", "ev_min_time_of_stay", "n_ev_specs"]
GEN_DOUBLE_FIELDS = ["spawn_multiplier", "discharge_price_factor", "power_setpoint_flexiblity",
                     "inflexible_loads_capacity_multiplier_mean", "inflexible_loads_forecast_mean", "inflexible_loads_forecast_std",
                     "solar_power_capacity_multiplier_mean", "solar_power_forecast_mean", "solar_power_forecast_std",
                     "dr_event_capacity_percentage_mean", "dr_event_capacity_percentage_std", "dr_event_start_hour_mean", "dr_event_start_hour_std",
                     "transformer_max_power", "cs_min_charge_current", "cs_max_charge_current", "cs_min_discharge_current",
                     "cs_max_discharge_current", "cs_voltage", "ev_battery_capacity", "ev_max_ac_charge_power", "ev_min_ac_charge_power",
                     "ev_max_discharge_power", "ev_min_discharge_power", "ev_charge_efficiency", "ev_discharge_efficiency", "ev_transition_soc",
                     "ev_transition_soc_multiplier", "ev_min_battery_capacity", "ev_min_emergency_battery_capacity", "ev_desired_capacity"]
GEN_TOPO_INT = ["topo_n_ports", "topo_transformer", "topo_phases"]
GEN_TOPO_DOUBLE = ["topo_min_charge_current", "topo_max_charge_current", "topo_min_discharge_current", "topo_max_discharge_current",
                   "topo_voltage", "topo_tr_max_power"]
GEN_SCENARIOS = {"workplace": 0, "public": 1, "private": 2}
GEN_DAYS = {"weekdays": 0, "weekends": 1, "both": 2}
GEN_FLEETS = {"v2g2024": 0, "ev_plus_phev": 1}


GEN_SPEC_DOUBLE = ["spec_registrations", "spec_battery_capacity", "spec_max_ac_charge_power", "spec_max_ac_discharge_power", "spec_efficiency"]
GEN_TABLE_DOUBLE = ["tab_arrival_week", "tab_arrival_weekend", "tab_stay", "tab_energy", "tab_pv"]


class GenConfigC(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in GEN_INT_FIELDS] + [("tr_seed", C.c_int64)] + [(n, C.c_double) for n in GEN_DOUBLE_FIELDS]
                + [(n, _pi) for n in GEN_TOPO_INT] + [(n, _pd) for n in GEN_TOPO_DOUBLE]
                + [(n, _pd) for n in GEN_SPEC_DOUBLE] + [(n, _pd) for n in GEN_TABLE_DOUBLE] + [("n_pv", C.c_int64)])
