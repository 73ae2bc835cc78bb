/*
 * ev2g.h -- C-ABI of the MI355X-native vectorised EV2Gym step engine (libev2g_hip.so).
 *
 * The reference (StavrosOrf/EV2Gym) has no FFI/operator layer for this path; its boundary is the
 * Python class ev2gym.models.ev2gym_env.EV2Gym (constructor :38-56, reset :243-331, step :333-447)
 * plus the rl_agent.state / rl_agent.reward hooks.  Each entry point below names the reference
 * interface it replaces.  Everything is `extern "C"`, plain pointers and sizes, no torch types.
 *
 * Conventions
 *   E envs stepped out of M >= E resident scenarios, C chargers/env, P = sum of the chargers' n_ports
 *   (C * ports_per_charger when they are equal), R transformers/env, T steps/episode, H = 20 observation
 *   horizon (state.py:119,129-132).  All floating point is IEEE float64 (float32 hand-over optional).
 *   Port index p = first port of the charger + port: cumulative in charger order, the reference's action
 *   order (ev2gym_env.py:363-385); the action mask follows the reference's own index i*n_ports+j (:452-457).
 *   Return value: 0 = ok, negative = EV2G_ERR_*; ev2g_last_error() returns a message.
 *   A handle is bound to one GPU and one HIP stream and is not thread-safe; distinct handles may be
 *   driven from distinct threads / processes (one process per GPU).
 *   Unless stated otherwise every `double*` / `uint8_t*` argument of reset/step/get_* is a DEVICE
 *   pointer (hipMalloc'd, or a torch-ROCm tensor's data_ptr()); the scenario batch is HOST memory,
 *   borrowed only for the duration of ev2g_load_scenarios().
 */
#ifndef EV2G_H
#define EV2G_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EV2G_ABI_VERSION 4

/* reward_function built-ins (rl_agent/reward.py) */
#define EV2G_REWARD_PROFITMAX_TRPENALTY_USERINCENTIVES 0 /* reward.py:34-44  */
#define EV2G_REWARD_SQUARED_TRACKING_ERROR 1             /* reward.py:7-14   */
#define EV2G_REWARD_PROFIT_MAXIMIZATION 2                /* reward.py:78-87  */
#define EV2G_REWARD_SQTR_TRPENALTY_USERINCENTIVES 3      /* reward.py:16-32: SqTrError_TrPenalty_UserIncentives        */
#define EV2G_REWARD_SQUARED_TRACKING_ERROR_PENALTY 4     /* reward.py:46-58: SquaredTrackingErrorRewardWithPenalty     */
#define EV2G_REWARD_SIMPLE 5                             /* reward.py:60-65: SimpleReward                              */
#define EV2G_REWARD_MINIMIZE_TRACKER_SURPLUS 6           /* reward.py:67-76: MinimizeTrackerSurplusWithChargeRewards   */
#define EV2G_REWARD_V2G_COSTS_SIMPLE 7                   /* reward.py:151-154: V2G_costs_simple                        */
#define EV2G_REWARD_V2G_PROFITMAX 8                      /* reward.py:120-148: V2G_profitmax                           */
#define EV2G_REWARD_V2G_PROFITMAX_V2 9                   /* reward.py:156-211: V2G_profitmaxV2                         */
#define EV2G_REWARD_PST_V2G_PROFITMAX_V2 10              /* reward.py:278-339: pst_V2G_profitmaxV2                     */
#define EV2G_N_REWARDS 11
/* Kinds 3, 8, 9 and 10 carry their own per-departure user term; the fused transformer_overload_usrpenalty cost (EV2G_COST_TR_OVERLOAD_
 * USRPENALTY) shares that staging slot and cannot be combined with them (ev2g_create refuses the pair).  The reference's other
 * reward built-ins need the grid simulation (V2G_grid_*) or are host-evaluated plugins through the Python facade. */
/* state_function built-ins (rl_agent/state.py) */
#define EV2G_STATE_V2G_PROFIT_MAX_LOADS 0 /* state.py:108-155, D = 2+H + 2H*R + 2P */
#define EV2G_STATE_PUBLIC_PST 1           /* state.py:6-63,    D = 3 + 3P          */
#define EV2G_STATE_V2G_PROFIT_MAX 2       /* state.py:65-106,  D = 2+H + 2P        */

/* cost_function built-ins (rl_agent/cost.py), evaluated next to the reward (ev2gym_env.py:434-438) */
#define EV2G_COST_NONE 0
#define EV2G_COST_TR_OVERLOAD_USRPENALTY 1 /* cost.py:8-18: 100*sum(overload) + 100*sum(exp(-10*score))  */
#define EV2G_COST_PROFIT_ONLY 2            /* cost.py:22-27 (ProfitMax_TrPenalty_UserIncentives_safety): total_costs */

#define EV2G_OK 0
#define EV2G_ERR_ARG -1       /* bad argument / inconsistent scenario                         */
#define EV2G_ERR_HIP -2       /* HIP runtime error                                            */
#define EV2G_ERR_STATE -3     /* call out of order (e.g. step before load)                    */
#define EV2G_ERR_DONE -4      /* step() after done: `assert not self.done` ev2gym_env.py:343  */
#define EV2G_ERR_OVERCURRENT -5 /* charger over-current Exception, ev_charger.py:203-205       */

#define EV2G_LUT_LEN 101 /* efficiency table 0..100 A, utils.py:282-288 */
#define EV2G_N_STATS 17  /* get_statistics() scalar keys, utils.py:84-101 */

/* flags for ev2g_config.flags */
#define EV2G_FLAG_LOG_CS_HISTORY 1 /* keep cs_power / cs_current [E,C,T] (ev2gym_env.py:533-535) */
#define EV2G_FLAG_NULL_STREAM 2    /* launch on the legacy default stream (torch's default stream) */
#define EV2G_FLAG_LOG_SOC 4        /* keep the per-step SoC log + |energy| sums that EV.get_battery_degradation needs
                                      (ev.py:156,162,180,185,442-521); without it the three degradation stats are NaN */
#define EV2G_FLAG_REFILLABLE 8     /* the resident scenario pool keeps a fixed-size block of session slots per scenario, so that
                                      ev2g_pool_refill can draw new scenarios into it ON THE DEVICE (default: packed storage)          */

typedef struct ev2g_handle ev2g_handle;

typedef struct ev2g_config {
    int32_t device;      /* HIP device ordinal                                                   */
    int32_t reward_kind; /* EV2G_REWARD_*  -- replaces the `reward_function` ctor kwarg (:47)    */
    int32_t state_kind;  /* EV2G_STATE_*   -- replaces the `state_function` ctor kwarg (:46)     */
    int32_t flags;       /* EV2G_FLAG_*                                                          */
    void *stream;        /* hipStream_t to launch on; NULL = a stream owned by the handle (or the
                            default stream with EV2G_FLAG_NULL_STREAM)                          */
    int32_t cost_kind;   /* EV2G_COST_*    -- replaces the `cost_function` ctor kwarg (:48)      */
    int32_t n_active_envs; /* E: envs stepped concurrently.  0 = as many as the loaded scenario pool holds.  With
                            E < pool size M the pool is a device-resident reservoir of scenarios and every reset
                            picks which M-cyclic window of it the E envs run (ev2g_reset_ex) -- the per-episode
                            scenario draw of EV2Gym.reset() (ev2gym_env.py:243-296) without a host round trip */
} ev2g_config;

/*
 * Scenario batch: every tensor EV2Gym.step() reads, for E independent envs that share one YAML
 * config (same chargers / sizes).  It is what the reference builds in __init__/reset()
 * (load_ev_charger_profiles loaders.py:299-365, load_transformers :227-296, EV_spawner
 * utils.py:477-557, load_electricity_prices loaders.py:392-461, load_power_setpoints :92-103).
 * HOST pointers, row-major, borrowed during the call.
 */
typedef struct ev2g_scenario_batch {
    int32_t n_envs;            /* E */
    int32_t n_steps;           /* T  = simulation_length                                  */
    int32_t timescale;         /* minutes per step                                        */
    int32_t n_chargers;        /* C                                                       */
    int32_t ports_per_charger; /* n_ports of every charger; with cs_n_ports: their maximum */
    int32_t n_transformers;    /* R                                                       */
    int32_t horizon;           /* H, must be 20                                           */
    int32_t n_dr_max;          /* ND: slots per transformer in tr_dr                      */
    int32_t n_lut;             /* NL efficiency tables                                    */
    int32_t reserved0;
    int64_t n_sessions;        /* total EV sessions over all envs = env_session_start[E]  */

    /* chargers [C]  (EV_Charger.__init__ ev_charger.py:41-94) */
    const double *cs_min_charge_current;
    const double *cs_max_charge_current;
    const double *cs_min_discharge_current; /* <= 0 */
    const double *cs_max_discharge_current; /* <= 0 */
    const double *cs_voltage;
    const int32_t *cs_phases;
    const int32_t *cs_transformer; /* connected_transformer, loaders.py:494-498 */
    /* n_ports of each charger when a topology file gives them different counts (loaders.py:312-340), or NULL: every
       charger has ports_per_charger ports.  Ports are numbered cumulatively in charger order (ev2gym_env.py:364-385). */
    const int32_t *cs_n_ports;

    /* per env [E,T]: row 0 of charge_prices / discharge_prices (identical for all chargers,
       loaders.py:423-424,439-442; charge price is negative) and power_setpoints */
    const double *charge_price;
    const double *discharge_price;
    const double *power_setpoints;

    /* per (env, transformer) [E,R,T]  (Transformer.__init__ transformer.py:38-78) */
    const double *tr_max_power;
    const double *tr_min_power;
    const double *tr_inflexible_load;
    const double *tr_solar_power;
    const double *tr_load_forecast; /* inflexible_load_forecast as it stands after reset() */
    const double *tr_pv_forecast;   /* pv_generation_forecast  as it stands after reset()  */
    const double *tr_dr;            /* [E,R,ND,3] (event_start_step, event_end_step, capacity_percentage) */
    const int32_t *tr_n_dr;         /* [E,R] number of valid events                        */
    const int32_t *tr_steps_ahead;  /* [E,R] transformer.py:66                             */

    /* EV sessions, CSR by env, in EVs_profiles order (sorted by arrival; utils.py:531-553) */
    const int64_t *env_session_start; /* [E+1] */
    const int32_t *ev_cs;             /* EV.location                                       */
    const int32_t *ev_t_arr;          /* time_of_arrival                                   */
    const int32_t *ev_t_dep;          /* time_of_departure                                 */
    const int32_t *ev_phases;         /* ev_phases                                         */
    const int32_t *ev_lut;            /* efficiency table id, -1 = scalar efficiencies     */
    const double *ev_cap0;            /* battery_capacity_at_arrival                       */
    const double *ev_B;               /* battery_capacity                                  */
    const double *ev_desired;         /* desired_capacity (kWh)                            */
    const double *ev_minB;            /* min_battery_capacity                              */
    const double *ev_min_emerg;       /* min_emergency_battery_capacity                    */
    const double *ev_pac_max;         /* max_ac_charge_power                               */
    const double *ev_pac_min;         /* min_ac_charge_power                               */
    const double *ev_pdis_max;        /* max_discharge_power (<= 0)                        */
    const double *ev_pdis_min;        /* min_discharge_power (<= 0)                        */
    const double *ev_ts;              /* transition_soc                                    */
    const double *ev_tsm;             /* transition_soc_multiplier                         */
    const double *ev_eta_ch;          /* scalar charge efficiency (ignored if ev_lut >= 0) */
    const double *ev_eta_dis;         /* scalar discharge efficiency                       */
    const double *lut;                /* [NL,101] percent, spawner fill utils.py:273-290   */
} ev2g_scenario_batch;

/* ---- lifetime ------------------------------------------------------------------------------ */
int ev2g_abi_version(void);
/* EV2Gym.__init__ (ev2gym_env.py:38-241): bind a device/stream and choose the fused hooks. */
int ev2g_create(const ev2g_config *cfg, ev2g_handle **out);
void ev2g_destroy(ev2g_handle *h);
const char *ev2g_last_error(const ev2g_handle *h); /* h may be NULL: last create() error */

/* The scenario-construction half of __init__/reset(): copies the batch to HBM, resolves every
 * session's port by replaying EV_Charger.spawn_ev's first-free rule (ev_charger.py:266-286,
 * action-independent), precomputes max_energy_AFAP (ev.py:407-440).  May be called again to
 * swap the scenario pool (same shapes or not).  Leaves the handle in the reset state. */
int ev2g_load_scenarios(ev2g_handle *h, const ev2g_scenario_batch *b);

/* shapes */
int ev2g_n_envs(const ev2g_handle *h);      /* E: envs stepped per call (n_active_envs)            */
int ev2g_n_scenarios(const ev2g_handle *h); /* M: scenarios resident in the pool (batch->n_envs)   */
int ev2g_n_ports(const ev2g_handle *h); /* P */
int ev2g_obs_dim(const ev2g_handle *h); /* D */
int ev2g_n_steps(const ev2g_handle *h); /* T */

/* ---- the hot path -------------------------------------------------------------------------- */
/* EV2Gym.reset() state-init part (ev2gym_env.py:298-306,329-331; init_statistic_variables
 * utils.py:794-861; EV_Charger.reset ev_charger.py:96-112) for every env of the batch, on the same
 * scenarios.  obs [E,D] may be NULL. */
int ev2g_reset(ev2g_handle *h, double *obs);
/* The same, preceded by the scenario draw of EV2Gym.reset() (ev2gym_env.py:243-296: new EV sessions, prices, loads,
 * PV, demand-response events, setpoints): env e runs scenario (e + scenario_offset) mod M of the resident pool for
 * the coming episode.  Distinct envs always run distinct scenarios (E <= M).  ev2g_reset() keeps the current offset
 * (re-arms the same scenarios); the host picks offsets (seeded, or fresh per episode).  O(1): nothing is copied. */
int ev2g_reset_ex(ev2g_handle *h, double *obs, int64_t scenario_offset);
int64_t ev2g_scenario_offset(const ev2g_handle *h);

/* EV2Gym.step(actions) for all E envs (ev2gym_env.py:333-447).
 *   actions     [E,P] float64, read-only here (the reference zeroes empty ports in the caller's
 *               array, ev_charger.py:139; the single-env Python facade reproduces that)
 *   obs         [E,D] state_function(env) after the step
 *   reward      [E]
 *   done        [E]   current_step >= simulation_length (:460)
 *   action_mask [E,P] 1 where a port holds an EV after the spawn phase (:452-457); may be NULL
 * Asynchronous on the handle's stream.  Returns EV2G_ERR_DONE if the batch is already done. */
int ev2g_step(ev2g_handle *h, const double *actions, double *obs, double *reward, uint8_t *done,
              uint8_t *action_mask);

/* K consecutive steps from a device-resident action source, enqueued without host round trips.
 * actions: [K,E,P] (action_step_stride = E*P) or one [E,P] block reused (stride 0).
 * Output k is written at base + k*<stride> elements (stride 0 = overwrite one buffer).
 *   mode 0  one kernel launch per step, enqueued back to back from C;
 *   mode 1  ONE persistent launch: every workgroup loops over the K steps of its own envs (envs are
 *           independent, so no grid-wide synchronisation is needed).
 * If the episode ends inside the K steps: with auto_reset != 0 the envs are reset before the next step -- the
 * terminal step still reports its own obs -- otherwise stepping stops there and EV2G_ERR_DONE is returned.
 * auto_reset == 1 re-arms the same scenarios (ev2g_reset); auto_reset == 2 moves on to the next E scenarios of
 * the pool (ev2g_reset_ex with scenario_offset + E), inside the persistent launch as well -- as long as the windows
 * the run visits are disjoint ((resets + 1) * E <= M); a run whose windows overlap is issued as one persistent
 * launch per episode (results are identical: per-session results are indexed by pool scenario, and two envs must
 * not write one scenario's slots within a single launch). */
#define EV2G_AUTO_RESET_SAME 1
#define EV2G_AUTO_RESET_NEXT 2
#define EV2G_STEPN_PER_STEP_LAUNCH 0
#define EV2G_STEPN_PERSISTENT 1
int ev2g_step_n(ev2g_handle *h, int k_steps, int mode, const double *actions, int64_t action_step_stride,
                double *obs, int64_t obs_step_stride, double *reward, int64_t reward_step_stride,
                uint8_t *done, int64_t done_step_stride, uint8_t *action_mask,
                int64_t mask_step_stride, int auto_reset);

/* Optional extra step outputs / inputs, sticky until changed (all-NULL = off, the default).  DEVICE pointers.
 *   cost        [E] float64: cost_function value of the step (config.cost_kind must not be EV2G_COST_NONE)
 *   obs_f32     [E,D] float32 copy of the observation, written next to `obs` (which may then be NULL): policy
 *               networks take float32, this saves them a conversion pass over the largest stream of the path
 *   actions_f32 [E,P] float32 actions, read (and widened to float64 on entry, as every action is) when the
 *               `actions` argument of ev2g_step / ev2g_step_n is NULL; its step stride is action_step_stride */
typedef struct ev2g_step_extras {
    double *cost;
    int64_t cost_step_stride;
    float *obs_f32;
    int64_t obs_f32_step_stride;
    const float *actions_f32;
} ev2g_step_extras;
int ev2g_set_step_extras(ev2g_handle *h, const ev2g_step_extras *x); /* x == NULL clears */

int ev2g_current_step(const ev2g_handle *h);
/* Which step kernel ev2g_load_scenarios selected for the loaded shape ("ev2g_step_wave<0,0>", "ev2g_step_v2<1024>",
 * "ev2g_step_kernel"), and -- when the common-shape fast path was not taken -- why ("" otherwise). */
const char *ev2g_kernel_name(const ev2g_handle *h);
const char *ev2g_fallback_reason(const ev2g_handle *h);
/* Round 6: big envs (512 < ports <= 1024) loaded on "ev2g_step_v2<1024>" run their specialised launches (ev2g_last_launch_specialisation 5) on
 * "ev2g_step_big" when the batch qualifies; this returns why it does NOT ("" when it does, or when the shape is not a big env). */
const char *ev2g_big_kernel_reason(const ev2g_handle *h);
/* Which instantiation of the fast-path kernel the last ev2g_step / ev2g_step_n launch used: 0 = the general one (any subset of outputs,
 * strides, extras, in-launch resets); 1 = "full" (all four outputs with step stride 0 -- float64 actions in and float64 observations out, or, with the float32 action and
 * observation buffers of ev2g_set_step_extras registered and no float64 ones passed, float32 in and out: ev2g_rollout --, no cost output, no charger
 * histories, the launch ends within the episode, one of the three compiled-in rewards): their checks are compiled out; 2 = full, plus
 * EV2G_FLAG_LOG_SOC on and an env wide enough for one observation-head column pair per lane.  The general kernel ("ev2g_step_v2<..>") has
 * one such instantiation (1) for the reference's default plugin pair -- V2G_profit_max_loads + ProfitMax_TrPenalty_UserIncentives, single-port
 * chargers, everything of the float64 list above, EV2G_FLAG_LOG_SOC, 15 / 30 / 60-minute steps -- and 0 otherwise.  -1: no launch yet, or the
 * generic kernel.
 * Round 5: 3 = 2 for outputs with STEP STRIDES (float64 [K,E,*] observation / reward / done / mask blocks of a persistent launch, every step kept:
 * generate_trajectories.py:69-83 style use; needs what 2 needs); 4 = the fused actor + step launch of ev2g_rollout / ev2g_collect (the policy
 * evaluated inside the step kernel's launch, one launch per segment; below).
 * Round 6: 5 = big envs (512 < ports <= 1024, single-port chargers, <= 50 transformers, <= 16 distinct charger tuples, windows below 16384 steps):
 * what 1 covers is run by "ev2g_step_big" -- 512 threads, two ports per home lane, 70 bytes of LDS per port, TWO workgroups per CU.  Port state,
 * observations, masks and transformer powers are bit-identical to 0 / 1; rewards and the episode sums of profits / energies come from a different
 * fixed summation tree (last-bit differences, tests hold them to 1e-12).  EV2G_NO_BIG=1 at load time keeps 1.
 * Results are identical in all of them (tests/test_round3_gpu.py, test_round4_gpu.py, test_round5_gpu.py); EV2G_NO_FULL / EV2G_NO_WIDE /
 * EV2G_NO_STRIDED in the environment at load time force 0 / 1 / "strided outputs run 0"; EV2G_NO_DICT=1 at load time keeps the battery-maths operands
 * one record per session instead of in the per-model dictionary, EV2G_NO_FUSED=1 keeps ev2g_rollout / ev2g_collect at two launches per step
 * (EV2G_NO_FUSED_F32=1: only the float32 policy). */
int ev2g_last_launch_specialisation(const ev2g_handle *h);
/* When the last fast-path launch got the general instantiation (0): what the caller passed or configured that ruled the full one out (the
 * first such thing), "" otherwise.  The Python Engine warns once with it: the general instantiation is ~20 % slower, silently. */
const char *ev2g_last_launch_general_reason(const ev2g_handle *h);
/* data-dependent faults recorded since the last reset (per-env flag word, device side):
 * returns 0 or EV2G_ERR_OVERCURRENT; synchronises the stream. */
int ev2g_check_faults(ev2g_handle *h, int32_t *first_bad_env);

/* ---- episode statistics (get_statistics utils.py:12-123) ------------------------------------ */
/* stats [E,EV2G_N_STATS] float64 DEVICE pointer, key order = ev2g_stat_name(i). */
int ev2g_get_stats(ev2g_handle *h, double *stats);
/* ev2g_get_stats followed by ev2g_reset_ex(h, obs, scenario_offset) in ONE kernel launch: what an auto-resetting vectorised env does at an
 * episode end (terminal info = get_statistics(), then reset(); ev2gym_env.py:243-331 + utils.py:12-123).  The wavefront that computed an env's
 * statistics re-arms that env on the new pool window; `obs` (DEVICE, may be NULL) receives the reset observation like ev2g_reset_ex. */
int ev2g_get_stats_reset(ev2g_handle *h, double *stats, double *obs, int64_t scenario_offset);
/* The same with the reset observation as float32 (the policy-network side of ev2g_collect / ev2g_rollout), and the plain reset with a float32
 * observation: `obs32` DEVICE [E, D] or NULL. */
int ev2g_get_stats_reset_f32(ev2g_handle *h, double *stats, float *obs32, int64_t scenario_offset);
int ev2g_reset_f32(ev2g_handle *h, float *obs32, int64_t scenario_offset);
const char *ev2g_stat_name(int i);

/* ---- multi-GPU: one process per GPU, envs sharded, statistics gathered over RCCL ---------------
 * The reference has no multi-process path (one env, one process).  Sharding envs over GPUs needs no data-path collective;
 * the only exchange is this per-episode statistics block.  For hosts without torch.distributed: rank 0 obtains an id,
 * ships its EV2G_COMM_ID_BYTES bytes to the other ranks by any host-side means, every rank calls ev2g_comm_init, and
 * ev2g_gather_stats then computes this rank's statistics and all-gathers them with ncclAllGather on the handle's stream
 * (stream-ordered, asynchronous to the host).  Every rank must hold the same number of envs.  librccl is opened on first
 * use; without it these calls fail with EV2G_ERR_STATE and the step path is unaffected. */
#define EV2G_COMM_ID_BYTES 128
int ev2g_comm_get_unique_id(void *id_out);
int ev2g_comm_init(ev2g_handle *h, const void *id, int rank, int world_size);
void ev2g_comm_destroy(ev2g_handle *h);
int ev2g_comm_world_size(const ev2g_handle *h);   /* 0: no communicator */
long long ev2g_comm_gathers(const ev2g_handle *h); /* all-gathers issued so far */
/* stats_all: DEVICE [world_size * E, EV2G_N_STATS] float64, rank-major */
int ev2g_gather_stats(ev2g_handle *h, double *stats_all);

/* ---- inspection (feeds the read-only Python facade of the reference object graph) ----------- */
/* HOST output buffers; any may be NULL.  Synchronises.  Port arrays are in reference port order;
 * empty ports are NaN / -1.  */
typedef struct ev2g_env_view {
    int32_t current_step;
    int32_t n_ports, n_chargers, n_transformers, n_steps;
    double *port_capacity;        /* [P] EV.current_capacity                */
    double *port_energy;          /* [P] EV.current_energy                  */
    double *port_current;         /* [P] EV.actual_current                  */
    double *port_total_energy;    /* [P] EV.total_energy_exchanged          */
    double *port_required_energy; /* [P] EV.required_energy                 */
    double *port_prev_power;      /* [P] EV.previous_power                  */
    int32_t *port_cycles;         /* [P] EV.charging_cycles                 */
    int32_t *port_session;        /* [P] index into the env's session list  */
    double *cs_power;             /* [C] EV_Charger.current_power_output    */
    double *cs_amps;              /* [C] EV_Charger.current_total_amps      */
    double *cs_profits;           /* [C] total_profits                      */
    double *cs_energy_charged;    /* [C] total_energy_charged               */
    double *cs_energy_discharged; /* [C] total_energy_discharged            */
    double *tr_power;             /* [R] Transformer.current_power          */
    double *tr_overload;          /* [R,T] env.tr_overload                  (the three histories: entries of steps the running episode has   */
    double *power_usage;          /* [T] env.current_power_usage             not reached yet are zeros, like the reference's arrays, whatever  */
    double *power_potential;      /* [T] env.charge_power_potential          an earlier episode left in the device buffers)                   */
    int32_t *session_port;        /* [S_env] resolved port of every session */
    double *session_afap;         /* [S_env] EV.max_energy_AFAP             */
    double *session_final_cap;    /* [S_env] capacity at departure (NaN while not departed) */
} ev2g_env_view;
int ev2g_peek(ev2g_handle *h, int env, ev2g_env_view *view);

/* ---- policy in the loop (BASELINE configs[4]: an SB3-MlpPolicy-shaped actor produces the actions between steps) ------ */
/* A three-layer MLP actor  obs[E,d_in] -> ReLU(h1) -> ReLU(h2) -> tanh(d_out)  evaluated by ONE kernel (bf16 MFMA, fp32
 * accumulation).  Weights are HOST pointers in torch.nn.Linear layout (W[out,in] row-major, b[out]), copied and packed once.
 * out_lo = -1: actions in [-1,1] (tanh); out_lo = 0: (tanh + 1) / 2, the action box of configs without V2G (ev2gym_env.py:226-231).
 * The reference has no counterpart: its agents are SB3 objects stepping one CPU env (train_stable_baselines.py:62-130). */
typedef struct ev2g_mlp ev2g_mlp;
int ev2g_mlp_create(ev2g_handle *h, int d_in, int h1, int h2, int d_out, const float *W1, const float *b1, const float *W2,
                    const float *b2, const float *W3, const float *b3, float out_lo, ev2g_mlp **out);
/* The same with the operand precision chosen, for policies trained in float32 (SB3's are):
 *   EV2G_MLP_BF16   (the call above) weights and activations rounded to bf16: fastest, actions within ~1e-2 of a float32 forward;
 *   EV2G_MLP_F32    float32 weights held as TWO bf16 terms (16 significant bits), activations as three, five MFMA products per k-step: the
 *                   device forward agrees with a float64 forward of the same weights to 1e-5 (observed 4e-6), at twice the bf16 time;
 *   EV2G_MLP_F32X3  three terms per weight (all 24 bits), six products: agreement at the 1e-7 level -- what float32 operands give --
 *                   at 2.6x the bf16 time.
 * Networks that fit the shipped shapes (inputs <= 192, hidden layers <= 400 / 304 with at least one >= 128, outputs <= 64) run all three modes on
 * the same streaming kernel on the bf16 matrix cores, zero-padded where they are smaller (the weight stream is the cost: 1x, 2x, 3x the bytes);
 * other networks fall back to generic kernels (bf16, and float32 operands on v_mfma_f32_32x32x2_f32 for both float32 modes). */
#define EV2G_MLP_BF16 0
#define EV2G_MLP_F32 1
#define EV2G_MLP_F32X3 2
int ev2g_mlp_create_ex(ev2g_handle *h, int d_in, int h1, int h2, int d_out, const float *W1, const float *b1, const float *W2,
                    const float *b2, const float *W3, const float *b3, float out_lo, int precision, ev2g_mlp **out);
void ev2g_mlp_destroy(ev2g_handle *h, ev2g_mlp *m);
/* y[n_rows,d_out] = actor(x[n_rows,d_in]); float32 DEVICE pointers; asynchronous on the handle's stream. */
int ev2g_mlp_forward(ev2g_handle *h, const ev2g_mlp *m, const float *x, float *y, int n_rows);
/* K rollout steps enqueued by one call: actor(obs_f32) -> actions_f32 -> EV2Gym.step, K times (the float32 buffers are the ones
 * registered with ev2g_set_step_extras, obs_f32_step_stride 0; d_in == obs dim, d_out == ports).  reward / done /
 * action_mask as in ev2g_step_n (mode EV2G_STEPN_PER_STEP_LAUNCH); auto_reset as there.
 * Round 5: ONE launch per segment where the shape is eligible -- the fast path (3..64 ports per env: the shipped V2GProfitPlusLoads
 * file's 25 chargers, BASELINE's 50), a V2G_profit_max(_loads) state, a compiled-in reward, EV2G_FLAG_LOG_SOC, no cost buffer, all three outputs
 * present, the segment inside the episode, and the bf16 policy in the 162 -> 400 -> 300 -> 64 packing: the workgroup that steps 16 envs
 * evaluates the policy on their 16 observation rows between the steps (same MFMA chains as ev2g_mlp_forward: bit-identical actions), so
 * neither kernel pays a cold start per step and the port state stays in LDS across the segment.  Anything else runs actor and step as two
 * launches per step, as before (train_stable_baselines.py:62-130 is the loop this replaces).
 * Round 6: PublicPST too (two envs per wavefront), and the FLOAT32 policy (EV2G_MLP_F32; one env per wavefront for every state): the same five-product chain per
 * k-step as ev2g_mlp_forward's (bit-identical actions), input rows kept float32 in LDS and split into their three bf16 terms where they are read;
 * EV2G_NO_FUSED_F32=1 keeps that policy at two launches per step.  EV2G_MLP_F32X3 policies always run two launches per step. */
int ev2g_rollout(ev2g_handle *h, const ev2g_mlp *m, int k_steps, double *reward, int64_t reward_step_stride, uint8_t *done,
                 int64_t done_step_stride, uint8_t *action_mask, int64_t mask_step_stride, int auto_reset);
/* Segments that contain no episode end are captured once as a HIP graph (keyed by their full launch signature) and replayed;
 * EV2G_ROLLOUT_GRAPHS=0 in the environment falls back to plain launches.  Number of graph replays so far: */
long long ev2g_rollout_graph_launches(const ev2g_handle *h);

/* Off-policy ROLLOUT COLLECTION into device memory (the collect_rollouts() half of an SB3 DDPG / TD3 / SAC loop,
 * train_stable_baselines.py:62-130): k_steps x (actor forward -> env step) whose transitions land directly in the caller's device arrays --
 * no host copy, no staging copy: the actor reads observation row i and writes action row i, the step kernel reads that action row and writes
 * observation row i + 1, reward / done / mask row i.  With next_obs[i] = obs[i + 1] these arrays ARE a replay-buffer segment (SB3's
 * ReplayBuffer(optimize_memory_usage=True) layout).  obs[0] is input: the observation the first action is computed from (the reset
 * observation, or the last row of the previous segment).  The segment must end at or before the episode end; statistics, the reset and
 * the terminal observation (= the segment's last observation row) are the caller's (ev2g_get_stats_reset).  All pointers DEVICE.
 * Eligible shapes run the whole segment as ONE fused launch (see ev2g_rollout); rows and values are the same either way. */
typedef struct {
    float *obs;         /* [k_steps + 1, E, D]; row 0 read, rows 1.. written */
    float *actions;     /* [k_steps, E, P] written */
    double *reward;     /* [k_steps, E] written */
    uint8_t *done;      /* [k_steps, E] written */
    uint8_t *mask;      /* [k_steps, E, P] written */
} ev2g_transitions;
int ev2g_collect(ev2g_handle *h, const ev2g_mlp *m, int k_steps, const ev2g_transitions *tr);

/* ---- plain device-memory helpers so a ctypes host needs no other HIP binding --------------- */
void *ev2g_malloc(ev2g_handle *h, size_t bytes);
void ev2g_free(ev2g_handle *h, void *p);
int ev2g_memcpy_h2d(ev2g_handle *h, void *dst, const void *src, size_t bytes);
int ev2g_memcpy_d2h(ev2g_handle *h, void *dst, const void *src, size_t bytes);
/* page-locked host memory (hipHostMalloc) for the buffers a host loop copies every step: the per-step numpy hand-over of the Stable-Baselines3 VecEnv
 * protocol (train_stable_baselines.py:62-130: observations / rewards / dones down, actions up) runs at the PCIe rate from such a buffer, at less than half
 * of it from pageable memory.  Freed by ev2g_host_free or with the handle. */
void *ev2g_host_malloc(ev2g_handle *h, size_t bytes);
void ev2g_host_free(ev2g_handle *h, void *p);
int ev2g_synchronize(ev2g_handle *h);
/* fills [n] doubles with uniform(lo,hi) from a counter-based generator (Philox-free splitmix64
 * keyed on (seed, index)); the same function exists on the host as ev2g_host_uniform so CPU and
 * GPU legs of a benchmark see identical action tensors (RandomAgent, heuristics.py:546-558). */
int ev2g_fill_uniform(ev2g_handle *h, double *dst, int64_t n, uint64_t seed, double lo, double hi);
void ev2g_host_uniform(double *dst, int64_t n, uint64_t seed, double lo, double hi);
/* elapsed GPU time of the kernels enqueued by the last ev2g_step_n(), measured with HIP events on
 * the handle's stream; total over the K launches, in milliseconds. */
double ev2g_last_step_n_kernel_ms(ev2g_handle *h);
/* the same for the timed call `back` calls before the last one (0 = the last; the handle keeps the event pairs of its last 32 ev2g_step_n /
 * ev2g_rollout / ev2g_collect calls): launches can be queued back to back and their durations read afterwards.  -1.0: out of range. */
double ev2g_step_n_kernel_ms_back(ev2g_handle *h, int back);

/* ---- scenario generator (host side, no GPU involved) ------------------------------------------------------------------
 * What EV2Gym.reset() draws for one episode -- EV sessions (EV_spawner utils.py:477-557, spawn_single_EV :177-345), prices
 * (load_electricity_prices loaders.py:392-461), transformer loads / PV / forecasts / demand-response events
 * (load_transformers loaders.py:227-296, transformer.py:80-256), power setpoints (utils.py:664-757) -- for n_scenarios
 * independent scenarios at once, as an ev2g_scenario_batch ready for ev2g_load_scenarios.  STATISTICALLY matched to the
 * reference (fitted hour-of-day tables and fleet classes; not its CSV data or RNG streams): tests/test_host_logic.py holds it
 * to summary statistics of the reference's own resets.  Every draw is a pure function of (seed, scenario index, counters):
 * the batch does not depend on n_threads, and scenario i of a run with a larger n_scenarios is the same scenario.
 * The fields carry the names and meaning of the YAML keys (ev2gym_env.py:65-166) / of GenConfig in ev2gym_amd/scenario_gen.py. */
typedef struct ev2g_gen_config {
    int32_t simulation_length, timescale;
    int32_t number_of_charging_stations, number_of_ports_per_cs, number_of_transformers;
    int32_t scenario;        /* 0 workplace, 1 public, 2 private                                  */
    int32_t simulation_days; /* 0 weekdays, 1 weekends, 2 both (a uniformly random day of the week) */
    int32_t hour, minute, random_hour;
    int32_t v2g_enabled, power_setpoint_enabled;
    int32_t inflexible_loads, solar_power, demand_response;
    int32_t dr_events_per_day, dr_event_length_minutes_min, dr_event_length_minutes_max, dr_notification_of_event_minutes;
    int32_t heterogeneous_ev_specs, fleet_with_efficiency_tables;
    int32_t fleet;           /* 0 "v2g2024", 1 "ev_plus_phev"                                      */
    int32_t cs_phases, ev_phases, ev_min_time_of_stay;
    int32_t n_ev_specs;      /* > 0: the car models of an EV-specification file replace the built-in fleets (spec_* below) */
    int64_t tr_seed;         /* != -1: loads / PV / events from their own seed (ev2gym_env.py:97-100) */
    double spawn_multiplier, discharge_price_factor, power_setpoint_flexiblity;
    double inflexible_loads_capacity_multiplier_mean, inflexible_loads_forecast_mean, inflexible_loads_forecast_std;
    double solar_power_capacity_multiplier_mean, solar_power_forecast_mean, solar_power_forecast_std;
    double dr_event_capacity_percentage_mean, dr_event_capacity_percentage_std, dr_event_start_hour_mean, dr_event_start_hour_std;
    double transformer_max_power;
    double cs_min_charge_current, cs_max_charge_current, cs_min_discharge_current, cs_max_discharge_current, cs_voltage;
    double ev_battery_capacity, ev_max_ac_charge_power, ev_min_ac_charge_power, ev_max_discharge_power, ev_min_discharge_power;
    double ev_charge_efficiency, ev_discharge_efficiency, ev_transition_soc, ev_transition_soc_multiplier;
    double ev_min_battery_capacity, ev_min_emergency_battery_capacity, ev_desired_capacity;
    /* charging_network_topology file (loaders.py:259-276,312-340), or all NULL: per-charger arrays
       [number_of_charging_stations] and the transformers' max_power [number_of_transformers] */
    const int32_t *topo_n_ports, *topo_transformer, *topo_phases;
    const double *topo_min_charge_current, *topo_max_charge_current, *topo_min_discharge_current, *topo_max_discharge_current;
    const double *topo_voltage, *topo_tr_max_power;
    /* the file `ev_specs_file` names (loaders.py:25-41), or n_ev_specs = 0: per model [n_ev_specs] the registrations (sampling
       weight), battery kWh, max AC charge / discharge kW; spec_efficiency [n_ev_specs][101] percent by charging current in A (the
       nearest-level fill of utils.py:268-288), a row starting with NaN = the model has no table (random scalar efficiency per EV,
       utils.py:290-296); NULL = no model has one */
    const double *spec_registrations, *spec_battery_capacity, *spec_max_ac_charge_power, *spec_max_ac_discharge_power, *spec_efficiency;
    /* tables of an EV2Gym data directory, or all NULL (the fitted hour-of-day tables / synthetic sun curve are used): arrivals per
       port per hour in percent by quarter hour [96] on weekdays / weekend days, mean stay in hours and mean required energy in kWh
       by half hour of arrival [48] (utils.py:199-233,505-528); tab_pv [n_pv] the hourly PV output of a year (loaders.py:165-224) */
    const double *tab_arrival_week, *tab_arrival_weekend, *tab_stay, *tab_energy, *tab_pv;
    int64_t n_pv;
} ev2g_gen_config;
/* fills *cfg with the values of V2GProfitPlusLoads.yaml (kind 0) or PublicPST.yaml (kind 1) */
int ev2g_gen_default_config(int kind, ev2g_gen_config *cfg);
typedef struct ev2g_gen_result ev2g_gen_result;
/* n_threads <= 0: one per hardware thread.  On failure returns a negative code, *out = NULL (ev2g_last_error(NULL) says why). */
int ev2g_generate(const ev2g_gen_config *cfg, int32_t n_scenarios, uint64_t seed, int32_t n_threads, ev2g_gen_result **out);
const ev2g_scenario_batch *ev2g_gen_batch(const ev2g_gen_result *r); /* arrays owned by r */
void ev2g_gen_free(ev2g_gen_result *r);
/* ---- scenario generation ON THE DEVICE: new scenarios drawn straight into the resident pool ------------------------------
 * EV2Gym.reset() draws a new scenario every episode (ev2gym_env.py:243-296).  The resident pool gives every episode fresh scenarios
 * for pool / envs episodes; this call re-draws pool slots [first_slot, first_slot + n) WITHOUT host work or PCIe traffic: slot
 * first_slot + j becomes scenario first_index + j of the stream (cfg, seed) -- bit for bit what ev2g_generate(cfg, ., seed) yields at
 * that index followed by ev2g_load_scenarios (one wavefront per scenario runs the generator's own code, csrc/ev2g_refill.h).
 * Requirements: the pool was loaded with EV2G_FLAG_REFILLABLE from a batch drawn with the same config (shape, fleet, topology).  Chargers
 * with several ports and charging_network_topology files are supported up to 256 steps and 256 ports (the kernel replays the reference's
 * first-free port assignment, ev_charger.py:266-286, per charger; a generator port that draws more than 8 sessions is cut and counted as an
 * overflow).  Asynchronous on the handle's stream; refill slots that no env is currently stepping (e.g. the window
 * the previous episode used).  Afterwards ev2g_peek is refused (the host holds no copy of the new scenarios).
 * ev2g_pool_refill_overflows: scenarios (since load) that drew more sessions than ev2g_pool_session_capacity slots and were
 * truncated (synchronises; 0 in practice: the blocks are 25 % + 8 larger than the largest scenario of the loaded batch). */
int ev2g_pool_refill(ev2g_handle *h, const ev2g_gen_config *cfg, uint64_t seed, int64_t first_index, int32_t first_slot, int32_t n);
long long ev2g_pool_refill_overflows(ev2g_handle *h);
int ev2g_pool_session_capacity(const ev2g_handle *h);
/* the generator's fitted tables: which = 0 arrivals per port per hour in percent, 1 mean stay in hours (24 values each),
 * 2 mean required energy (1 value), for table kind 0 workplace, 1 public, 2 private, 3 public weekend, 4 private weekend;
 * which = 3 / 4: the V2G / EV+PHEV fleet as rows of (share, battery kWh, max AC kW); returns the number of values written */
int ev2g_gen_table(int which, int kind, double *out, int n_max);

#ifdef __cplusplus
}
#endif
#endif /* EV2G_H */
