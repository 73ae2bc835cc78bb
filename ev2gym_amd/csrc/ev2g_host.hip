// ev2g_host.hip -- host side of libev2g_hip.so: the C-ABI of include/ev2g.h over HIP.
//
// Scenario packing (the host half of ev2g_load_scenarios) turns the reference-shaped batch
// (chargers, transformers, EVs_profiles-ordered sessions) into the device layout documented in
// ev2g_device.h / DESIGN.md:  transformer-major port slots, sessions sorted by (env, slot, arrival)
// with the next session's window chained in, per-port first-session tables, max_energy_AFAP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <array>
#include <map>
#include <string>
#include <vector>

#include "../../include/ev2g.h"
#include "ev2g_device.h"
#include "ev2g_step_v2.h"
#include "ev2g_step_wave.h"
#include "ev2g_step_big.h"
#include "ev2g_mlp.h"
#include "ev2g_comm.h"
#include "ev2g_refill.h"
#include <cstdlib>

static thread_local std::string g_create_error;
#include "ev2g_gen_host.h"

#define EV2G_EV_RING 32
struct ev2g_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    ev2g_config cfg{};
    bool loaded = false;
    DevScn scn{};
    DevState st{};
    std::vector<void *> scn_allocs, st_allocs, user_allocs;
    std::vector<void *> user_host_allocs;       // ev2g_host_malloc: page-locked host buffers of the caller's per-step copies
    void *peek_stage = nullptr; size_t peek_stage_bytes = 0;   // ev2g_peek's page-locked staging block
    // host mirrors for peek / stats
    int E = 0, M = 0, T = 0, C = 0, npc = 0, P = 0, R = 0, D = 0;   // E envs stepped concurrently, M scenarios in the pool
    long long scn_off = 0;                      // env e runs scenario (e + scn_off) mod M
    ev2g_step_extras extras{};
    long long S = 0;
    std::vector<int> slot_port, port_slot;
    std::vector<long long> env_sess_start;      // [E+1] host order
    std::vector<int> host_to_dev;               // [S] device session index of host session
    std::vector<int> sess_port;                 // [S] resolved reference port, host order
    std::vector<double> sess_afap;              // [S] host order
    long long *d_env_sess = nullptr;            // unused placeholder for the stats kernel signature
    double *d_ss_afap = nullptr;                // [S] device order
    double *d_step_tab = nullptr;               // [M,T,8] (fast path)
    V2P *d_v2p = nullptr;                       // device copy of the v2 kernel's parameter block
    int block = 0;                              // 256/512/1024: v2 kernel; 0: generic kernel (P > 1024)
    double *d_head_tab = nullptr; int head_nh = 0;   // observation head table of the fast path (rebuilt for refilled slots)
    double *d_lut_rowmax = nullptr;             // [n_lut] largest entry of every efficiency table
    bool refilled = false;                      // ev2g_pool_refill ran: the host copies of the scenarios (peek) no longer describe the pool
    int *d_refill_overflow = nullptr;
    struct RefillCache {                        // device copies of the generator config's arrays, kept while the config does not change
        std::vector<unsigned char> key;
        std::vector<void *> allocs;
        RefillArgs args{};
    } refill_cache;
    int sess_cap = 0;                           // EV2G_FLAG_REFILLABLE: session slots per scenario of the resident pool (0: packed storage)
    bool wave_path = false;                     // ev2g_step_wave: P <= 64, one transformer, single-port chargers
    int wave_epw = 1, wave_es = 64;             // ... its envs per wavefront and the lane stride between them (WaveArgs::epw / es)
    bool big_path = false;                      // ev2g_step_big (512 < P <= 1024, two workgroups per CU) takes the launches ev2g_step_v2<1024, 1> would
    std::string big_reason;                     // why not ("" when it does / when the shape is not a big env)
    size_t lds_big = 0;
    BigArgs big_args{};
    bool no_full = false, no_wide = false;      // EV2G_NO_FULL / EV2G_NO_WIDE at load time: A/B and routing tests only
    bool no_strided = false;                    // EV2G_NO_STRIDED at load time: strided outputs run the general instantiation (round 4's routing; parity tests)
    // battery-maths dictionary (ClsRec, ev2g_device.h): host mirror of the entries in use, so that ev2g_pool_refill can append the
    // classes its fleet may draw; d_cls_rec has room for EV2G_CLS_CAP entries when DevScn::dict is set
    typedef std::array<uint64_t, 11> ClsKey;
    std::map<ClsKey, int> cls_map;
    ClsRec *d_cls_rec = nullptr;
    std::vector<double> cs_vk_host;             // [C,4] voltage*sqrt(k) and [C] phases of the loaded chargers (the refill's dictionary entries)
    std::vector<int> cs_ph_host;
    int load_gen = 0;                           // counts ev2g_load_scenarios calls (part of the refill cache's key)
    int last_spec = -1;                         // ev2g_last_launch_specialisation
    const char *general_reason = "";            // ev2g_last_launch_general_reason
    bool pow2_dt = false;                       // 60 / timescale is a power of two (15, 30, 60 minutes): compiled into ev2g_step_v2<.., 1>
    std::string kernel_name;                    // the step kernel ev2g_load_scenarios selected (ev2g_kernel_name)
    std::string fallback_reason;                // why the common-shape fast path was NOT taken ("" when it was / does not apply)
    int current_step = 0;
    size_t lds_bytes = 0;
    // HIP-event pairs of the last EV2G_EV_RING timed calls (ev2g_step_n / ev2g_rollout / ev2g_collect): a caller that queues several launches and
    // reads their durations afterwards (bench.py's roofline pass) does not have to drain the stream after each one
    hipEvent_t ev0s[EV2G_EV_RING] = {}, ev1s[EV2G_EV_RING] = {};
    int ev_slot = 0;
    bool ev_valid[EV2G_EV_RING] = {};          // the slot's closing event was recorded (a call that failed half-way leaves it false: its duration reads -1)
    unsigned fused_attr_mask = 0;               // fused instantiations whose dynamic-LDS attribute was set for THIS handle's device (bit = state kind * 4 + reward kind)
    long long ev_calls = 0;
    bool timed = false;
    std::string err;
    CommState comm;                             // RCCL communicator of the statistics exchange (ev2g_comm_init), if any
    // ev2g_rollout segments captured as HIP graphs: a policy-in-the-loop step is two short dependent kernels, so the enqueue cost
    // of 2k launches (and the gaps between them) is comparable to the kernels; a segment with the same signature is replayed
    struct RolloutGraph {
        const void *mlp; int k, t0; long long scn_off; const void *rew, *done, *mask; long long rs, ds, ms;
        ev2g_step_extras x; hipGraphExec_t exec;
    };
    std::vector<RolloutGraph> rollout_graphs;
    long long graph_launches = 0;
};

#define HIPCHK(h, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
            return EV2G_ERR_HIP;                                                                     \
        }                                                                                            \
    } while (0)

static int fail(ev2g_handle *h, int code, const std::string &msg) {
    if (h) h->err = msg;
    else g_create_error = msg;
    return code;
}

template <typename T>
static int upload(ev2g_handle *h, std::vector<void *> &pool, const T *src, size_t n, T **dst) {
    void *p = nullptr;
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    HIPCHK(h, hipMalloc(&p, bytes));
    pool.push_back(p);
    if (n) HIPCHK(h, hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyHostToDevice, h->stream));
    *dst = (T *)p;
    return 0;
}
template <typename T>
static int dalloc(ev2g_handle *h, std::vector<void *> &pool, size_t n, T **dst) {
    void *p = nullptr;
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    HIPCHK(h, hipMalloc(&p, bytes));
    HIPCHK(h, hipMemsetAsync(p, 0, bytes, h->stream));
    pool.push_back(p);
    *dst = (T *)p;
    return 0;
}
static void free_pool(std::vector<void *> &pool) {
    for (void *p : pool) (void)hipFree(p);
    pool.clear();
}

// dictionary entry of a ClsRec (by value, bit for bit); -1 when the dictionary is full
static int cls_find_or_add(std::map<ev2g_handle::ClsKey, int> &map, std::vector<ClsRec> &tab, const ClsRec &c) {
    ev2g_handle::ClsKey k;
    const double f[11] = {c.pacmax, c.tsm, c.gate_ch, c.B, c.rB, c.v, c.rv, c.gate_dis, c.minB, c.emerg, c.pdismax};
    std::memcpy(k.data(), f, sizeof f);
    auto it = map.find(k);
    if (it != map.end()) return it->second;
    if (map.size() >= EV2G_CLS_CAP) return -1;
    const int id = (int)map.size();
    map.emplace(k, id);
    if ((size_t)id >= tab.size()) tab.resize((size_t)id + 1);
    tab[(size_t)id] = c;
    return id;
}

#include "ev2g_refill_host.h"

extern "C" {

int ev2g_abi_version(void) { return EV2G_ABI_VERSION; }

const char *ev2g_last_error(const ev2g_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int ev2g_create(const ev2g_config *cfg, ev2g_handle **out) {
    if (!cfg || !out) return fail(nullptr, EV2G_ERR_ARG, "ev2g_create: null argument");
    if (cfg->reward_kind < 0 || cfg->reward_kind >= EV2G_N_REWARDS || cfg->state_kind < 0 || cfg->state_kind > 2)
        return fail(nullptr, EV2G_ERR_ARG, "ev2g_create: unknown reward_kind/state_kind");
    if (cfg->cost_kind == EV2G_COST_TR_OVERLOAD_USRPENALTY &&
        (cfg->reward_kind == EV2G_REWARD_SQTR_TRPENALTY_USERINCENTIVES || cfg->reward_kind >= EV2G_REWARD_V2G_PROFITMAX))
        return fail(nullptr, EV2G_ERR_ARG, "ev2g_create: the fused transformer_overload_usrpenalty cost shares its per-departure staging slot with "
                                           "the user term of this reward (SqTrError_TrPenalty_UserIncentives / V2G_profitmax / *V2G_profitmaxV2): evaluate one of the two on the host");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(nullptr, EV2G_ERR_HIP, "ev2g_create: no HIP device visible (the engine has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= n) return fail(nullptr, EV2G_ERR_ARG, "ev2g_create: device ordinal out of range");
    ev2g_handle *h = new ev2g_handle();
    h->cfg = *cfg;
    h->device = cfg->device;
    if (hipSetDevice(h->device) != hipSuccess) {
        delete h;
        return fail(nullptr, EV2G_ERR_HIP, "ev2g_create: hipSetDevice failed");
    }
    if (cfg->stream) {
        h->stream = (hipStream_t)cfg->stream;
    } else if (cfg->flags & EV2G_FLAG_NULL_STREAM) {
        h->stream = nullptr;  // legacy default stream
    } else {
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
            delete h;
            return fail(nullptr, EV2G_ERR_HIP, "ev2g_create: hipStreamCreate failed");
        }
        h->own_stream = true;
    }
    for (int i = 0; i < EV2G_EV_RING; i++) { (void)hipEventCreate(&h->ev0s[i]); (void)hipEventCreate(&h->ev1s[i]); }
    *out = h;
    return EV2G_OK;
}

// captured rollout segments hold kernel arguments (device pointers, shapes, actor weights) of the moment they were recorded
static void drop_rollout_graphs(ev2g_handle *h) {
    for (auto &g : h->rollout_graphs) (void)hipGraphExecDestroy(g.exec);
    h->rollout_graphs.clear();
}

void ev2g_destroy(ev2g_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    free_pool(h->scn_allocs);
    free_pool(h->st_allocs);
    free_pool(h->user_allocs);
    for (void *p : h->user_host_allocs) (void)hipHostFree(p);
    h->user_host_allocs.clear();
    if (h->peek_stage) (void)hipHostFree(h->peek_stage);
    free_pool(h->refill_cache.allocs);
    if (h->d_refill_overflow) (void)hipFree(h->d_refill_overflow);
    ev2g_comm_destroy(h);
    drop_rollout_graphs(h);
    for (int i = 0; i < EV2G_EV_RING; i++) { if (h->ev0s[i]) (void)hipEventDestroy(h->ev0s[i]); if (h->ev1s[i]) (void)hipEventDestroy(h->ev1s[i]); }
    if (h->own_stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int ev2g_n_envs(const ev2g_handle *h) { return h ? h->E : 0; }
int ev2g_n_ports(const ev2g_handle *h) { return h ? h->P : 0; }
int ev2g_obs_dim(const ev2g_handle *h) { return h ? h->D : 0; }
int ev2g_n_steps(const ev2g_handle *h) { return h ? h->T : 0; }
int ev2g_current_step(const ev2g_handle *h) { return h ? h->current_step : 0; }
const char *ev2g_kernel_name(const ev2g_handle *h) { return (h && h->loaded) ? h->kernel_name.c_str() : ""; }
const char *ev2g_fallback_reason(const ev2g_handle *h) { return (h && h->loaded) ? h->fallback_reason.c_str() : ""; }
const char *ev2g_big_kernel_reason(const ev2g_handle *h) { return (h && h->loaded) ? h->big_reason.c_str() : ""; }
int ev2g_last_launch_specialisation(const ev2g_handle *h) { return (h && h->loaded) ? h->last_spec : -1; }
const char *ev2g_last_launch_general_reason(const ev2g_handle *h) { return (h && h->loaded && h->last_spec == 0) ? h->general_reason : ""; }

static const char *kStatNames[EV2G_N_STATS] = {
    "total_ev_served", "total_profits", "total_energy_charged", "total_energy_discharged",
    "average_user_satisfaction", "power_tracker_violation", "tracking_error", "energy_tracking_error",
    "energy_user_satisfaction", "std_energy_user_satisfaction", "min_energy_user_satisfaction",
    "total_steps_min_emergency_battery_capacity_violation", "total_transformer_overload",
    "battery_degradation", "battery_degradation_calendar", "battery_degradation_cycling", "total_reward"};
const char *ev2g_stat_name(int i) { return (i >= 0 && i < EV2G_N_STATS) ? kStatNames[i] : ""; }

// EV.calculate_max_energy_with_AFAP (ev.py:407-440)
static double afap_energy(const ev2g_scenario_batch *b, long long s, double max_cs_power) {
    const double pac = b->ev_pac_max[s];
    const double max_power = (std::fabs(max_cs_power) > std::fabs(pac)) ? pac : max_cs_power;
    double eff;
    if (b->ev_lut[s] >= 0) {
        double m = 0;
        for (int k = 0; k < EV2G_LUT_LEN; k++) m = std::max(m, b->lut[(size_t)b->ev_lut[s] * EV2G_LUT_LEN + k]);
        eff = m / 100.0;
    } else
        eff = b->ev_eta_ch[s];
    double x = b->ev_cap0[s];
    for (int k = b->ev_t_arr[s]; k < b->ev_t_dep[s] + 1; k++) {
        x += max_power * eff * b->timescale / 60.0;
        x = std::ceil(x * 100.0) / 100.0;
        if (x > b->ev_B[s]) {
            x = b->ev_B[s];
            break;
        }
    }
    return x;
}

int ev2g_reset_ex(ev2g_handle *h, double *obs, int64_t scenario_offset);

int ev2g_load_scenarios(ev2g_handle *h, const ev2g_scenario_batch *b) {
    if (!h || !b) return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: null argument");
    (void)hipSetDevice(h->device);
    const int M = b->n_envs, T = b->n_steps, C = b->n_chargers, npc = b->ports_per_charger, R = b->n_transformers;
    const int ND = std::max(b->n_dr_max, 0);
    const int E = h->cfg.n_active_envs > 0 ? h->cfg.n_active_envs : M;   // envs stepped concurrently
    if (E > M) return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: n_active_envs exceeds the number of scenarios in the batch");
    if (M <= 0 || T <= 0 || C <= 0 || npc <= 0 || R <= 0 || b->timescale <= 0)
        return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: non-positive size");
    if (b->horizon != 20) return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: horizon must be 20 (state.py:119,129-132)");
    if (npc > 32) return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: more than 32 ports per charger unsupported");
    // ports of each charger: uniform, or per charger from a topology file (loaders.py:312-340); numbered cumulatively in charger
    // order like the reference's port_counter (ev2gym_env.py:364-385)
    std::vector<int> np_of(C, npc), pbase(C + 1, 0);
    bool het = false;
    if (b->cs_n_ports) {
        int mx = 0;
        for (int c = 0; c < C; c++) {
            np_of[c] = b->cs_n_ports[c];
            if (np_of[c] < 1) return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: cs_n_ports must be >= 1");
            mx = std::max(mx, np_of[c]);
            het = het || np_of[c] != npc;
        }
        if (mx != npc) return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: ports_per_charger must be the maximum of cs_n_ports");
    }
    for (int c = 0; c < C; c++) pbase[c + 1] = pbase[c] + np_of[c];
    const int P = pbase[C];
    if (het)   // the reference's action mask is indexed i*cs.n_ports + j (ev2gym_env.py:452-457): past the array it raises IndexError
        for (int c = 0; c < C; c++)
            if (c * np_of[c] + np_of[c] > P)
                return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: this charger order makes the reference's action mask index "
                                             "i*n_ports+j leave the mask array (ev2gym_env.py:457 raises IndexError); order the chargers by falling port count");
    const long long S = b->env_session_start[M];
    if (S != b->n_sessions || b->env_session_start[0] != 0)
        return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: env_session_start inconsistent with n_sessions");
    if (S > 0x7ffffff0LL) return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: too many sessions for 32-bit indices");
    {   // the kernels index with 32-bit ints: every element offset they form must stay below 2^31
        const long long lim = 0x7fffffffLL, Pq = P;
        const long long Dq = 3 + 3 * Pq > 22 + 40LL * R + 2 * Pq ? 3 + 3 * Pq : 22 + 40LL * R + 2 * Pq;
        const bool log_cs = (h->cfg.flags & EV2G_FLAG_LOG_CS_HISTORY) != 0;
        if ((long long)M * Pq > lim || (long long)M * Dq > lim || (long long)M * R * (T + 1) * 40 > lim ||
            (log_cs && (long long)T * M * std::max<long long>(C, Pq) > lim) || (long long)M * T * 8 > lim)
            return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: batch too large for 32-bit element offsets "
                                         "(need M*P, M*D, M*R*(T+1)*40, T*M*C < 2^31): split it over more handles / GPUs");
    }
    if (T > 65535 || b->n_lut > 65534)
        return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: simulation_length and the number of efficiency tables must stay below 65536 (a port's state line "
                                     "packs charging_cycles and the table id into 16 bits each)");
    for (int c = 0; c < C; c++) {
        if (b->cs_transformer[c] < 0 || b->cs_transformer[c] >= R)
            return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: cs_transformer out of range");
        if (b->cs_phases[c] < 1 || b->cs_phases[c] > 3)
            return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: cs_phases must be 1..3");
    }
    (void)hipStreamSynchronize(h->stream);
    drop_rollout_graphs(h);
    free_pool(h->scn_allocs);
    free_pool(h->st_allocs);
    h->loaded = false;

    // ---- slot order: transformer-major, chargers in id order inside a transformer, ports adjacent ----
    std::vector<int> slot_port(P), slot_cs(P), slot_tr(P), slot_obs(P), slot_mask(P), port_slot(P), tr_seg(R + 1, 0), tr_obs(R);
    {
        int q = 0;
        for (int r = 0; r < R; r++) {
            tr_seg[r] = q;
            for (int c = 0; c < C; c++)
                if (b->cs_transformer[c] == r)
                    for (int j = 0; j < np_of[c]; j++) {
                        slot_port[q] = pbase[c] + j;
                        slot_mask[q] = c * np_of[c] + j;   // where the reference sets this port's action-mask entry (ev2gym_env.py:457)
                        slot_cs[q] = c;
                        slot_tr[q] = r;
                        port_slot[pbase[c] + j] = q;
                        q++;
                    }
        }
        tr_seg[R] = q;
    }
    const int sk = h->cfg.state_kind;
    int D;
    if (sk == EV2G_STATE_PUBLIC_PST) {
        D = 3 + 3 * P;
        for (int q = 0; q < P; q++) slot_obs[q] = 3 + 3 * q;
        for (int r = 0; r < R; r++) tr_obs[r] = 0;
    } else if (sk == EV2G_STATE_V2G_PROFIT_MAX) {
        D = 22 + 2 * P;
        for (int q = 0; q < P; q++) slot_obs[q] = 22 + 2 * q;
        for (int r = 0; r < R; r++) tr_obs[r] = 0;
    } else {
        D = 22 + 40 * R + 2 * P;
        for (int r = 0; r < R; r++) tr_obs[r] = 22 + 40 * r + 2 * tr_seg[r];
        for (int q = 0; q < P; q++) slot_obs[q] = 22 + 40 * (slot_tr[q] + 1) + 2 * q;
    }
    int max_seg = 1;
    for (int r = 0; r < R; r++) max_seg = std::max(max_seg, tr_seg[r + 1] - tr_seg[r]);

    // ---- resolve ports (first-free replay, ev_charger.py:266-286) and order sessions by (env, slot, arrival) ----
    // Device session storage.  Packed (default): scenario m owns the device sessions [env_session_start[m], env_session_start[m+1]).
    // EV2G_FLAG_REFILLABLE: every scenario owns a fixed-size block of `cap` session slots (the largest count of the batch + 25 % + 8,
    // or EV2G_POOL_SESSION_CAP), so that ev2g_pool_refill can regenerate a scenario in place on the device; SD counts slots, holes included.
    const bool refillable = (h->cfg.flags & EV2G_FLAG_REFILLABLE) != 0;
    long long cap = 0;
    if (refillable) {
        for (int m = 0; m < M; m++) cap = std::max<long long>(cap, b->env_session_start[m + 1] - b->env_session_start[m]);
        cap = ((cap + cap / 4 + 8) + 7) / 8 * 8;
        if (const char *e = std::getenv("EV2G_POOL_SESSION_CAP")) cap = std::max<long long>(cap, std::atoll(e));
        if (cap * M > 0x7ffffff0LL) return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: too many session slots for 32-bit indices (refillable pool)");
    }
    const long long SD = refillable ? cap * M : S;
    h->sess_cap = (int)cap;
    std::vector<int> sess_port((size_t)S), host_to_dev((size_t)S), ss_slot((size_t)std::max<long long>(SD, 1));   // ss_slot: port slot of a session, device order
    std::vector<int> scn_sess((size_t)M + 1), scn_sess_end((size_t)M);   // device sessions of scenario m: [scn_sess[m], scn_sess_end[m]) (device order is scenario-major)
    for (int m = 0; m <= M; m++) scn_sess[(size_t)m] = refillable ? (int)(cap * m) : (int)b->env_session_start[m];
    for (int m = 0; m < M; m++) scn_sess_end[(size_t)m] = scn_sess[(size_t)m] + (int)(b->env_session_start[m + 1] - b->env_session_start[m]);
    std::vector<long long> dev_to_host((size_t)SD, -1);
    std::vector<int> port_first((size_t)M * P, -1);
    std::vector<int> port_end((size_t)M * P, -1);   // one past the port's last session (device order: a port's sessions are consecutive)
    std::vector<int2> port_first_win((size_t)M * P, make_int2(EV2G_INT_MAX, EV2G_INT_MAX));
    {
        std::vector<int> free_at((size_t)C * npc);
        std::vector<std::pair<long long, long long>> keyed;  // (slot, host idx)
        long long d = 0;
        for (int e = 0; e < M; e++) {
            d = scn_sess[(size_t)e];
            std::fill(free_at.begin(), free_at.end(), 0);
            const long long s0 = b->env_session_start[e], s1 = b->env_session_start[e + 1];
            if (s1 < s0) return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: env_session_start not monotone");
            keyed.clear();
            int prev_arr = 0;
            for (long long s = s0; s < s1; s++) {
                const int cs = b->ev_cs[s], ta = b->ev_t_arr[s], td = b->ev_t_dep[s];
                if (cs < 0 || cs >= C) return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: ev_cs out of range");
                if (ta < 1 || td < ta) return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: need 1 <= t_arr <= t_dep");
                if (ta < prev_arr) return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: sessions must be sorted by arrival");
                if (b->ev_phases[s] < 1 || b->ev_phases[s] > 3)
                    return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: ev_phases must be 1..3");
                if (b->ev_lut[s] >= b->n_lut) return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: ev_lut out of range");
                prev_arr = ta;
                int slot = -1;
                for (int j = 0; j < np_of[cs]; j++)
                    if (free_at[(size_t)cs * npc + j] <= ta - 1) {  // attached at the end of step ta-1
                        slot = j;
                        break;
                    }
                if (slot < 0)
                    return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: no free port for a session (assert n_evs_connected < n_ports, ev_charger.py:271)");
                free_at[(size_t)cs * npc + slot] = td;  // freed inside step td, before that step's spawns
                sess_port[s] = pbase[cs] + slot;
                keyed.emplace_back((long long)port_slot[pbase[cs] + slot], s);
            }
            std::stable_sort(keyed.begin(), keyed.end(),
                             [](const auto &x, const auto &y) { return x.first < y.first; });
            for (auto &kv : keyed) {
                host_to_dev[kv.second] = (int)d;
                dev_to_host[d] = kv.second;
                ss_slot[(size_t)d] = (int)kv.first;
                const size_t g = (size_t)e * P + kv.first;
                if (port_first[g] < 0) {
                    port_first[g] = (int)d;
                    port_first_win[g] = make_int2(b->ev_t_arr[kv.second], b->ev_t_dep[kv.second]);
                }
                port_end[g] = (int)d + 1;
                d++;
            }
        }
    }
    // ---- gather session fields into device order, chain the next window ----
#define GATHER(type, name, src)                                 \
    std::vector<type> name((size_t)SD);                         \
    for (long long d = 0; d < SD; d++) if (dev_to_host[d] >= 0) name[d] = b->src[dev_to_host[d]];
    GATHER(int, ss_tarr, ev_t_arr)
    GATHER(int, ss_tdep, ev_t_dep)
    GATHER(int, ss_phases, ev_phases)
    GATHER(int, ss_lut, ev_lut)
    GATHER(double, ss_cap0, ev_cap0)
    GATHER(double, ss_B, ev_B)
    GATHER(double, ss_des, ev_desired)
    GATHER(double, ss_minB, ev_minB)
    GATHER(double, ss_emerg, ev_min_emerg)
    GATHER(double, ss_pacmax, ev_pac_max)
    GATHER(double, ss_pacmin, ev_pac_min)
    GATHER(double, ss_pdismax, ev_pdis_max)
    GATHER(double, ss_pdismin, ev_pdis_min)
    GATHER(double, ss_ts, ev_ts)
    GATHER(double, ss_tsm, ev_tsm)
    GATHER(double, ss_etach, ev_eta_ch)
    GATHER(double, ss_etadis, ev_eta_dis)
#undef GATHER
    std::vector<int> ss_ntarr((size_t)SD, EV2G_INT_MAX), ss_ntdep((size_t)SD, EV2G_INT_MAX);
    std::vector<double> ss_afap((size_t)SD), sess_afap_host((size_t)S);
    for (long long d = 0; d + 1 < SD; d++) {
        const long long a = dev_to_host[d], c = dev_to_host[d + 1];
        if (a < 0 || c < 0) continue;   // (an unused slot of a refillable pool)
        // same env and same port => the next device session is this port's next session
        bool same_env = false;
        {
            // env of a host session: binary search in env_session_start
            const int64_t *st = b->env_session_start;
            const int64_t *ua = std::upper_bound(st, st + M + 1, (int64_t)a);
            const int64_t *uc = std::upper_bound(st, st + M + 1, (int64_t)c);
            same_env = (ua == uc);
        }
        if (same_env && sess_port[a] == sess_port[c]) {
            ss_ntarr[d] = ss_tarr[d + 1];
            ss_ntdep[d] = ss_tdep[d + 1];
        }
    }
    std::vector<double> cs_maxp(C), cs_minp(C), cs_vk((size_t)C * 4), cs_dmax_abs(C);
    for (int c = 0; c < C; c++) {
        const double V = b->cs_voltage[c];
        for (int k = 0; k < 4; k++) cs_vk[(size_t)c * 4 + k] = V * std::sqrt((double)k);
        const double sq = std::sqrt((double)b->cs_phases[c]);
        cs_maxp[c] = sq * V * b->cs_max_charge_current[c] / 1000;  // utils.py:779-782
        cs_minp[c] = sq * V * b->cs_min_charge_current[c] / 1000;
        cs_dmax_abs[c] = std::fabs(b->cs_max_discharge_current[c]);
    }
    for (long long s = 0; s < S; s++) {
        const int cs = b->ev_cs[s];
        // EV_Charger.get_max_power (ev_charger.py:251-252)
        const double mp = b->cs_max_charge_current[cs] * b->cs_voltage[cs] * std::sqrt((double)b->cs_phases[cs]) / 1000;
        sess_afap_host[s] = afap_energy(b, s, mp);
        ss_afap[host_to_dev[s]] = sess_afap_host[s];
    }
    std::vector<double> tr_peak((size_t)M * R);
    for (size_t er = 0; er < (size_t)M * R; er++) {
        double m = b->tr_max_power[er * T];
        for (int t = 1; t < T; t++) m = std::max(m, b->tr_max_power[er * T + t]);
        tr_peak[er] = m;
    }

    // ---- launch geometry ----
    DevScn &s = h->scn;
    s = DevScn{};
    s.E = E; s.M = M; s.T = T; s.C = C; s.npc = npc; s.P = P; s.R = R; s.D = D; s.ND = std::max(ND, 1); s.dt = b->timescale;
    s.reward_kind = h->cfg.reward_kind; s.state_kind = sk; s.flags = h->cfg.flags; s.cost_kind = h->cfg.cost_kind; s.n_lut = b->n_lut;
    // v2 kernel: one home lane per port, BLOCK >= P; the generic kernel handles larger envs
    h->block = het ? 0 : (P <= 256) ? 256 : (P <= 512) ? 512 : (P <= 1024) ? 1024 : 0;   // different port counts per charger: generic kernel
    const int blk = h->block ? h->block : EV2G_BLOCK;
    s.G = std::max(1, blk / P);
    s.G = std::min(s.G, E);
    // fast path (ev2g_step_wave): the common shape.  Anything else runs the general kernels; which one was chosen, and why the
    // fast path was not, is reported by ev2g_kernel_name() / ev2g_fallback_reason() -- routing is never silent.
    h->fallback_reason.clear();
    if (P < 2 || P > 64) h->fallback_reason = "ports per env outside 2..64";
    else if (R != 1) h->fallback_reason = "more than one transformer";
    else if (npc != 1) h->fallback_reason = "multi-port chargers";
    if (het) h->fallback_reason = "chargers with different port counts (topology file)";
    h->wave_path = h->fallback_reason.empty();
    h->no_full = std::getenv("EV2G_NO_FULL") != nullptr; h->no_wide = std::getenv("EV2G_NO_WIDE") != nullptr; h->last_spec = -1;
    h->no_strided = std::getenv("EV2G_NO_STRIDED") != nullptr;
    if (h->wave_path) {   // ev2g_step_wave addresses every array as base + 32-bit byte offset: all of them must stay below 4 GiB
        const unsigned long long lim = 1ull << 32;
        const unsigned long long biggest = std::max({(unsigned long long)E * P * 8, (unsigned long long)E * D * 8,
                                                     (unsigned long long)M * (T + 1) * 60 * 8, (unsigned long long)SD * sizeof(SessRec),
                                                     (unsigned long long)M * T * 64, (unsigned long long)E * T * 8 * 3, (unsigned long long)M * P * 8, (unsigned long long)E * P * sizeof(PortLine),
                                                     (h->cfg.flags & EV2G_FLAG_LOG_SOC) ? (unsigned long long)E * T * P * 8 : 0ull,
                                                     (h->cfg.flags & EV2G_FLAG_LOG_CS_HISTORY) ? (unsigned long long)T * E * C * 8 : 0ull});
        if (biggest >= lim) { h->wave_path = false; h->fallback_reason = "an array of the batch reaches 4 GiB (32-bit byte offsets)"; }
    }
    if (h->wave_path) {   // wave-aligned: 64/P envs per wavefront, packed (EV2G_EPW_CAP / EV2G_EPW_ALIGN: A/B switches, read at load)
        h->wave_epw = 64 / P; h->wave_es = P;
        if (const char *c = std::getenv("EV2G_EPW_CAP")) h->wave_epw = std::max(1, std::min(h->wave_epw, std::atoi(c)));
        if (std::getenv("EV2G_EPW_ALIGN") && 64 / h->wave_epw >= P) h->wave_es = 64 / h->wave_epw;
        s.G = (EV2G_WAVE_BLOCK / 64) * h->wave_epw;
    }
    {   // EV2G_KERNEL=v2 forces the general kernel on the common shape (parity tests compare the two)
        const char *kn = std::getenv("EV2G_KERNEL");
        if (h->wave_path && kn && std::string(kn) == "v2") {
            h->wave_path = false; h->fallback_reason = "EV2G_KERNEL=v2"; s.G = std::min(std::max(1, blk / P), E);
        }
    }
    {
        char nm[64];
        if (h->wave_path) std::snprintf(nm, sizeof nm, "ev2g_step_wave<%d,%d>", sk, std::min(h->cfg.reward_kind, 3));
        else if (h->block) std::snprintf(nm, sizeof nm, "ev2g_step_v2<%d>", h->block);
        else std::snprintf(nm, sizeof nm, "ev2g_step_kernel");
        h->kernel_name = nm;
    }
    {
        int gs = 4;
        while (gs < 64 && gs < max_seg) gs <<= 1;
        s.gs = gs;
    }
    s.n_groups = (E + s.G - 1) / s.G;
    s.sixty_over_dt = 60.0 / (double)b->timescale;
    s.dt_over_60 = (double)b->timescale / 60.0;
    if (h->wave_path)
        h->lds_bytes = ev2g_wave_lds_bytes(s.G);
    else if (h->block)
        h->lds_bytes = ev2g_v2_lds_bytes(s.G * P, s.G * R, s.G, R);
    else
        h->lds_bytes = ev2g_generic_lds_bytes(s.G, P, R, npc);
    if (h->lds_bytes > 160 * 1024)
        return fail(h, EV2G_ERR_ARG, "ev2g_load_scenarios: ports per env exceed the LDS staging capacity (P <= ~2400)");
    {
        const void *fn = h->block == 256 ? (const void *)ev2g_step_v2<256>
                         : h->block == 512 ? (const void *)ev2g_step_v2<512>
                         : h->block == 1024 ? (const void *)ev2g_step_v2<1024> : (const void *)ev2g_step_kernel;
        if (h->lds_bytes > 48 * 1024)
            HIPCHK(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
        const void *fs = h->block == 256 ? (const void *)ev2g_step_v2<256, 1>
                         : h->block == 512 ? (const void *)ev2g_step_v2<512, 1>
                         : h->block == 1024 ? (const void *)ev2g_step_v2<1024, 1> : nullptr;
        if (fs && h->lds_bytes > 48 * 1024)
            HIPCHK(h, hipFuncSetAttribute(fs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
    }
    // AoS session records (one cache line each) for the v2 kernel
    std::vector<SessRec> recs((size_t)std::max<long long>(SD, 1));
    std::vector<SessTail> tails((size_t)std::max<long long>(SD, 1));
    std::memset(recs.data(), 0, recs.size() * sizeof(SessRec));
    std::memset(tails.data(), 0, tails.size() * sizeof(SessTail));
    for (long long d = 0; d < SD; d++) {
        const long long hs = dev_to_host[d];
        if (hs < 0) continue;
        const int cs = b->ev_cs[hs];
        SessRec &r = recs[d];
        r.B = ss_B[d]; r.cap0 = ss_cap0[d]; r.minB = ss_minB[d]; r.emerg = ss_emerg[d];
        r.pacmax = ss_pacmax[d]; r.pdismax = ss_pdismax[d]; r.ts = ss_ts[d]; r.tsm = ss_tsm[d];
        r.eta_ch = ss_etach[d]; r.eta_dis = ss_etadis[d];
        const double v_gate = cs_vk[(size_t)cs * 4 + b->cs_phases[cs]];
        r.gate_ch = ss_pacmin[d] * 1000.0 / v_gate;
        r.gate_dis = ss_pdismin[d] * 1000.0 / v_gate;
        r.v = cs_vk[(size_t)cs * 4 + std::min(b->cs_phases[cs], ss_phases[d])];
        r.rB = 1.0 / r.B; r.rv = 1.0 / r.v;   // correctly rounded reciprocals (IEEE division): what ev2g_fdiv2 divides through
        {   // this EV's charge-power-potential term before the charger clamp (utils.py:773-777), the reference's operations in its order
            const double evc = r.pacmax * 1000.0 / r.v, imax = b->cs_max_charge_current[cs];
            r.potc = r.v * ((evc < imax) ? evc : imax) / 1000.0;
        }
        tails[d].des = ss_des[d]; tails[d].nt_arr = ss_ntarr[d]; tails[d].nt_dep = ss_ntdep[d];
    }
    // Battery-maths dictionary (fast path): the distinct (car model x charger kind) operand tuples of the batch, and per session what is left
    // (SessDyn).  More than EV2G_CLS_CAP tuples (arbitrary ev_* arrays through the ABI), or EV2G_NO_DICT: one ClsRec per session, entry = session.
    std::vector<SessDyn> dyns((size_t)std::max<long long>(SD, 1));
    std::memset(dyns.data(), 0, dyns.size() * sizeof(SessDyn));
    std::vector<ClsRec> cls_tab;
    h->cls_map.clear();
    bool dict = h->wave_path && std::getenv("EV2G_NO_DICT") == nullptr;
    if (h->wave_path) {
        for (long long d = 0; d < SD && dict; d++) {
            if (dev_to_host[d] < 0) continue;
            const int k = cls_find_or_add(h->cls_map, cls_tab, ev2g_cls_of(recs[d]));
            if (k < 0) { dict = false; break; }
            dyns[d].cls = k;
        }
        if (!dict) {
            h->cls_map.clear();
            cls_tab.assign((size_t)std::max<long long>(SD, 1), ClsRec{});
            for (long long d = 0; d < SD; d++) if (dev_to_host[d] >= 0) { cls_tab[d] = ev2g_cls_of(recs[d]); dyns[d].cls = (int)d; }
        } else
            cls_tab.resize(EV2G_CLS_CAP, ClsRec{});   // room for the classes a device refill may add
        for (long long d = 0; d < SD; d++) {
            if (dev_to_host[d] < 0) continue;
            dyns[d].ts = ss_ts[d]; dyns[d].eta_ch = ss_etach[d]; dyns[d].eta_dis = ss_etadis[d]; dyns[d].lut = ss_lut[d];
        }
        if (b->n_lut > 4094) h->no_full = true;   // the full kernels keep table id + 1 in 12 bits of a port's LDS word
    }

    // ---- upload ----
    auto &pool = h->scn_allocs;
    int rc = 0;
#define UP(dst, vec) if ((rc = upload(h, pool, (vec).data(), (vec).size(), &(dst)))) return rc;
#define UPP(dst, ptr, n) if ((rc = upload(h, pool, (ptr), (size_t)(n), &(dst)))) return rc;
    int *ip; double *dp; int2 *i2p;
    UP(ip, slot_port) s.slot_port = ip;
    UP(ip, slot_cs) s.slot_cs = ip;
    UP(ip, slot_obs) s.slot_obs = ip;
    UP(ip, slot_tr) s.slot_tr = ip;
    UP(ip, slot_mask) s.slot_mask = ip;
    UP(ip, np_of) s.cs_np = ip;
    UP(ip, pbase) s.cs_pbase = ip;
    {
        std::vector<int> cs_slot0(C);
        for (int c = 0; c < C; c++) cs_slot0[c] = port_slot[pbase[c]];
        UP(ip, cs_slot0) s.cs_slot0 = ip;
    }
    s.het = het ? 1 : 0;
    UPP(dp, b->cs_min_charge_current, C) s.cs_imin = dp;
    UPP(dp, b->cs_max_charge_current, C) s.cs_imax = dp;
    UPP(dp, b->cs_min_discharge_current, C) s.cs_dmin = dp;
    UP(dp, cs_dmax_abs) s.cs_dmax_abs = dp;
    UPP(dp, b->cs_voltage, C) s.cs_volt = dp;
    UP(dp, cs_maxp) s.cs_maxp = dp;
    UP(dp, cs_minp) s.cs_minp = dp;
    UP(dp, cs_vk) s.cs_vk = dp;
    {   // the six per-charger operands of the fast path side by side: (imax, |dmax|), (imin, dmin), (max power, min power)
        std::vector<double> cs_pack((size_t)C * 6);
        for (int c = 0; c < C; c++) {
            double *r = &cs_pack[(size_t)c * 6];
            r[0] = b->cs_max_charge_current[c]; r[1] = cs_dmax_abs[c]; r[2] = b->cs_min_charge_current[c];
            r[3] = b->cs_min_discharge_current[c]; r[4] = cs_maxp[c]; r[5] = cs_minp[c];
        }
        UP(dp, cs_pack) s.cs_pack = dp;
    }
    {   // ev2g_step_big (ev2g_step_big.h): big single-env workgroups with two ports per home lane.  What it needs beyond the launch-time conditions
        // of the specialised instantiation: single-port chargers, at most 64 transformers, at most EV2G_BIG_NCC distinct charger tuples, windows that fit
        // 16-bit step numbers.  EV2G_NO_BIG keeps ev2g_step_v2<1024> (A/B runs, parity tests).
        h->big_path = false; h->big_reason.clear(); h->big_args = BigArgs{};
        if (h->block == 1024 && !h->wave_path) {
            std::vector<unsigned char> ccls(P);
            std::vector<double> ctab;
            bool many = false;
            for (int q = 0; q < P && !many; q++) {
                const int c = slot_cs[q];
                const double r[6] = {b->cs_max_charge_current[c], b->cs_min_charge_current[c], b->cs_min_discharge_current[c], cs_dmax_abs[c], cs_maxp[c], cs_minp[c]};
                int k = 0;
                const int n = (int)(ctab.size() / 6);
                while (k < n && std::memcmp(&ctab[(size_t)k * 6], r, sizeof r) != 0) k++;
                if (k == n) { if (n == EV2G_BIG_NCC) { many = true; break; } ctab.insert(ctab.end(), r, r + 6); }
                ccls[q] = (unsigned char)k;
            }
            int tmax = T, tmin = 0;
            for (long long d = 0; d < SD; d++) if (dev_to_host[d] >= 0) { tmax = std::max({tmax, ss_tarr[d], ss_tdep[d]}); tmin = std::min({tmin, ss_tarr[d], ss_tdep[d]}); }
            bool even = D % 2 == 0;
            for (int q = 0; q < P; q++) even = even && slot_obs[q] % 2 == 0;
            const size_t lb = ev2g_big_lds_bytes(P, R);
            if (std::getenv("EV2G_NO_BIG")) h->big_reason = "EV2G_NO_BIG is set";
            else if (npc != 1) h->big_reason = "multi-port chargers";
            else if (sk != EV2G_STATE_V2G_PROFIT_MAX_LOADS) h->big_reason = "the state function is not V2G_profit_max_loads";
            else if (20 * R + 20 > 2 * EV2G_BIG_BLOCK) h->big_reason = "more than 50 transformers (their 20 R window-column pairs + 20 price columns ride in 1024 pair slots)";
            else if (many) h->big_reason = "more than 16 distinct charger constant tuples";
            else if (tmax > EV2G_BIG_TMAX / 2 || tmin < -1) h->big_reason = "a session window or the episode length exceeds 16383 steps";
            else if (!even || D >= 65536) h->big_reason = "observation columns are not 16-byte aligned pairs";
            else if (lb > 80 * 1024) h->big_reason = "the port state exceeds half of a CU's LDS";
            else if (E < 1) h->big_reason = "no envs";
            else {
                unsigned char *cp; UP(cp, ccls) h->big_args.slot_ccls = cp;
                UP(dp, ctab) h->big_args.ccls_tab = dp;
                std::vector<double> ptab(15, std::nan(""));   // the distinct potential terms of the loaded sessions (the first 15: any other value is fetched from the state line)
                int npot = 0;
                for (long long d = 0; d < SD && npot < 15; d++) {
                    if (dev_to_host[d] < 0) continue;
                    int k = 0;
                    while (k < npot && ptab[(size_t)k] != recs[d].potc) k++;
                    if (k == npot) ptab[(size_t)npot++] = recs[d].potc;
                }
                UP(dp, ptab) h->big_args.potc_tab = dp;
                h->big_args.ncc = (int)(ctab.size() / 6);
                h->lds_big = lb; h->big_path = true;
                HIPCHK(h, hipFuncSetAttribute((const void *)ev2g_step_big<EV2G_BIG_BLOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));
            }
        }
    }
    UPP(ip, b->cs_phases, C) s.cs_ph = ip;
    UP(ip, tr_seg) s.tr_seg = ip;
    UP(ip, tr_obs) s.tr_obs = ip;
    UPP(dp, b->charge_price, (size_t)M * T) s.price_ch = dp;
    UPP(dp, b->discharge_price, (size_t)M * T) s.price_dis = dp;
    UPP(dp, b->power_setpoints, (size_t)M * T) s.setpoint = dp;
    UPP(dp, b->tr_max_power, (size_t)M * R * T) s.tr_maxp = dp;
    UPP(dp, b->tr_min_power, (size_t)M * R * T) s.tr_minp = dp;
    UPP(dp, b->tr_inflexible_load, (size_t)M * R * T) s.tr_infl = dp;
    UPP(dp, b->tr_solar_power, (size_t)M * R * T) s.tr_solar = dp;
    UPP(dp, b->tr_load_forecast, (size_t)M * R * T) s.tr_lf = dp;
    UPP(dp, b->tr_pv_forecast, (size_t)M * R * T) s.tr_pvf = dp;
    UP(dp, tr_peak) s.tr_peak = dp;
    {
        std::vector<double> tr_base((size_t)M * R * T);
        for (size_t i = 0; i < tr_base.size(); i++) tr_base[i] = b->tr_inflexible_load[i] + b->tr_solar_power[i];
        UP(dp, tr_base) s.tr_base = dp;
    }
    UPP(dp, b->tr_dr, (size_t)M * R * ND * 3) s.tr_dr = dp;
    UPP(ip, b->tr_n_dr, (size_t)M * R) s.tr_ndr = ip;
    UPP(ip, b->tr_steps_ahead, (size_t)M * R) s.tr_ahead = ip;
    UP(ip, ss_tarr) s.ss_tarr = ip;
    UP(ip, ss_tdep) s.ss_tdep = ip;
    UP(ip, ss_ntarr) s.ss_ntarr = ip;
    UP(ip, ss_ntdep) s.ss_ntdep = ip;
    UP(ip, ss_phases) s.ss_phases = ip;
    UP(ip, ss_lut) s.ss_lut = ip;
    UP(dp, ss_cap0) s.ss_cap0 = dp;
    UP(dp, ss_B) s.ss_B = dp;
    UP(dp, ss_des) s.ss_des = dp;
    UP(dp, ss_minB) s.ss_minB = dp;
    UP(dp, ss_emerg) s.ss_emerg = dp;
    UP(dp, ss_pacmax) s.ss_pacmax = dp;
    UP(dp, ss_pacmin) s.ss_pacmin = dp;
    UP(dp, ss_pdismax) s.ss_pdismax = dp;
    UP(dp, ss_pdismin) s.ss_pdismin = dp;
    UP(dp, ss_ts) s.ss_ts = dp;
    UP(dp, ss_tsm) s.ss_tsm = dp;
    UP(dp, ss_etach) s.ss_etach = dp;
    UP(dp, ss_etadis) s.ss_etadis = dp;
    UPP(dp, b->lut, (size_t)b->n_lut * EV2G_LUT_LEN) s.lut = dp;
    // the same tables as efficiencies (percent / 100, the division of ev.py:290,379 done once): what V2P::lut points at
    std::vector<double> lut_eta((size_t)b->n_lut * EV2G_LUT_LEN);
    for (size_t i = 0; i < lut_eta.size(); i++) lut_eta[i] = b->lut[i] / 100.0;
    double *d_lut_eta = nullptr;
    UP(d_lut_eta, lut_eta)
    {   // the largest entry of every efficiency table (percent): what EV.calculate_max_energy_with_AFAP uses (ev.py:418-421); device-side refills need it
        std::vector<double> rowmax((size_t)std::max(b->n_lut, 1), 0.0);
        for (int l = 0; l < b->n_lut; l++)
            for (int k = 0; k < EV2G_LUT_LEN; k++) rowmax[l] = std::max(rowmax[l], b->lut[(size_t)l * EV2G_LUT_LEN + k]);
        UP(dp, rowmax) h->d_lut_rowmax = dp;
    }
    h->refilled = false;
    UP(ip, port_first) s.port_first = ip;
    UP(ip, port_end) s.port_end = ip;
    UP(ip, ss_slot) s.ss_slot = ip;
    UP(ip, scn_sess) s.scn_sess = ip;
    UP(ip, scn_sess_end) s.scn_sess_end = ip;
    UP(i2p, port_first_win) s.port_first_win = i2p;
    UP(dp, ss_afap) h->d_ss_afap = dp;
    { SessRec *rp; UP(rp, recs) s.rec = rp; }
    { SessTail *tp; UP(tp, tails) s.tail = tp; }
    s.sess_dyn = nullptr; s.cls_rec = nullptr; s.dict = 0; s.n_cls = 0; h->d_cls_rec = nullptr;
    if (h->wave_path) {
        SessDyn *dp2; UP(dp2, dyns) s.sess_dyn = dp2;
        ClsRec *cp; UP(cp, cls_tab) s.cls_rec = cp; h->d_cls_rec = cp;
        s.dict = dict ? 1 : 0; s.n_cls = dict ? (int)h->cls_map.size() : 0;
    }
#undef UP
#undef UPP
    s.win_tab = nullptr;
    if (sk == EV2G_STATE_V2G_PROFIT_MAX_LOADS && h->block) {
        double *tab = nullptr;
        const size_t n = (size_t)M * R * (T + 1) * 40;
        HIPCHK(h, hipMalloc((void **)&tab, n * sizeof(double)));
        pool.push_back(tab);
        const int nb = (int)std::min<size_t>((n + 255) / 256, 4096);
        hipLaunchKernelGGL(ev2g_build_window_table_kernel, dim3(nb), dim3(256), 0, h->stream, s, tab, 0, M);
        HIPCHK(h, hipGetLastError());
        s.win_tab = tab;
    }
    double *d_step_tab = nullptr;
    h->d_step_tab = nullptr;
    if (h->wave_path) {   // R == 1: [M,T] series interleaved per (env, step)
        const size_t n = (size_t)M * T * 8;
        HIPCHK(h, hipMalloc((void **)&d_step_tab, n * sizeof(double)));
        pool.push_back(d_step_tab);
        const int nb = (int)std::min<size_t>(((size_t)M * T + 255) / 256, 4096);
        hipLaunchKernelGGL(ev2g_build_step_table_kernel, dim3(nb), dim3(256), 0, h->stream, s, d_step_tab, 0, M);
        HIPCHK(h, hipGetLastError());
        hipLaunchKernelGGL(ev2g_build_occ_mask_kernel, dim3(std::min(M, 8192)), dim3(64), 0, h->stream, s, d_step_tab, 0, M);   // slots 6, 7
        HIPCHK(h, hipGetLastError());
        h->d_step_tab = d_step_tab;
    }
    double *d_head_tab = nullptr;
    h->d_head_tab = nullptr; h->head_nh = 0;
    s.head_tab = nullptr; s.head_nh = 0;
    if (h->wave_path && sk != EV2G_STATE_PUBLIC_PST) {
        const int NH = (sk == EV2G_STATE_V2G_PROFIT_MAX_LOADS) ? 60 : 20;
        const size_t n = (size_t)M * (T + 1) * NH;
        HIPCHK(h, hipMalloc((void **)&d_head_tab, n * sizeof(double)));
        pool.push_back(d_head_tab);
        const int nb = (int)std::min<size_t>((n + 255) / 256, 4096);
        hipLaunchKernelGGL(ev2g_build_head_table_kernel, dim3(nb), dim3(256), 0, h->stream, s.price_ch, s.win_tab, 0, M, T, NH, d_head_tab);
        HIPCHK(h, hipGetLastError());
        h->d_head_tab = d_head_tab; h->head_nh = NH;
        s.head_tab = d_head_tab; s.head_nh = NH;   // (the reset observation copies its head from row 0, write_obs_env)
    }
    // ---- state ----
    DevState &st = h->st;
    st = DevState{};
    auto &sp = h->st_allocs;
    const size_t EP = (size_t)E * P, EC = (size_t)E * C;
#define AL(field, n) if ((rc = dalloc(h, sp, (size_t)(n), &st.field))) return rc;
    {   // per-port state: one 64-byte line per port + one slab of EV2G_PS_* slices for what is not on the step's path
        AL(line, EP)
        HIPCHK(h, hipMemsetAsync(st.line, 0, EP * sizeof(PortLine), h->stream));
        if (h->wave_path) { AL(port_dyn, EP) }
        const size_t slice = std::max(EP, EC) * 8;
        if ((rc = dalloc(h, sp, slice * EV2G_PS_N, &st.slab_port))) return rc;
        st.slab_port_slice = slice;
#define SLICE(T, k) ((T *)(st.slab_port + slice * (size_t)(k)))
        st.port_energy = SLICE(double, EV2G_PS_PENERGY); st.port_current = SLICE(double, EV2G_PS_PCURRENT);
        st.cs_sat_sum = SLICE(double, EV2G_PS_SATSUM); st.cs_served = SLICE(int, EV2G_PS_SERVED);
#undef SLICE
    }
    if (h->cfg.flags & EV2G_FLAG_LOG_CS_HISTORY) {
        AL(cs_profits, EC) AL(cs_e_ch, EC) AL(cs_e_dis, EC) AL(cs_power_now, EC) AL(cs_cur_now, EC)
        AL(cs_power_hist, (size_t)T * EC) AL(cs_cur_hist, (size_t)T * EC)
    }
    AL(env_acc, (size_t)E * 8) AL(env_fault, E)
    AL(slab_hist, (size_t)T * E * (2 + R))
    st.hist = st.slab_hist;
    AL(slab_sess, (size_t)std::max<long long>(SD, 1) * 2)
    st.sess_final_cap = st.slab_sess;
    AL(tr_power_now, (size_t)E * R)
    if (h->cfg.flags & EV2G_FLAG_LOG_SOC) { AL(soc_log, (size_t)T * EP) st.sess_abs_e = st.slab_sess + (size_t)std::max<long long>(SD, 1); }
#ifdef EV2G_PHASE_TIMING
    AL(dbg, (size_t)s.n_groups * 18)
#endif
#undef AL
    {
        V2P v2p;
        ev2g_v2_fill_params(v2p, h->scn, h->st);
        EV2G_SETP(v2p.lut, d_lut_eta);
        EV2G_SETP(v2p.head_tab, d_head_tab);
        EV2G_SETP(v2p.step_tab, d_step_tab);
        int ex = 0;   // 60/dt a power of two and dt/60 its exact reciprocal -> divisions by them are multiplications
        v2p.pow2_dt = (std::frexp(h->scn.sixty_over_dt, &ex) == 0.5 && h->scn.sixty_over_dt * h->scn.dt_over_60 == 1.0) ? 1 : 0;
        h->pow2_dt = v2p.pow2_dt != 0;
        if ((rc = upload(h, sp, &v2p, 1, &h->d_v2p))) return rc;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));  // host staging vectors die here

    h->E = E; h->M = M; h->scn_off = 0; h->T = T; h->C = C; h->npc = npc; h->P = P; h->R = R; h->D = D; h->S = S;
    h->slot_port = slot_port;
    h->port_slot = port_slot;
    h->env_sess_start.assign(b->env_session_start, b->env_session_start + M + 1);
    h->host_to_dev = host_to_dev;
    h->cs_vk_host = cs_vk; h->cs_ph_host.assign(b->cs_phases, b->cs_phases + C); h->load_gen += 1;
    h->sess_port = sess_port;
    h->sess_afap = sess_afap_host;
    h->loaded = true;
    if (h->extras.cost || h->extras.obs_f32 || h->extras.actions_f32) {
        const ev2g_step_extras keep = h->extras;
        if ((rc = ev2g_set_step_extras(h, &keep))) return rc;
    }
    return ev2g_reset_ex(h, nullptr, 0);
}

int ev2g_reset_ex(ev2g_handle *h, double *obs, int64_t scenario_offset) {
    if (!h || !h->loaded) return fail(h, EV2G_ERR_STATE, "ev2g_reset: no scenarios loaded");
    (void)hipSetDevice(h->device);
    const DevScn &s = h->scn;
    long long off = scenario_offset % (long long)s.M;
    if (off < 0) off += s.M;
    h->scn_off = off;
    // (the usage | potential | overload history slab is cleared by the reset kernel itself)
    if (h->st.cs_power_hist) {
        HIPCHK(h, hipMemsetAsync(h->st.cs_power_hist, 0, sizeof(double) * (size_t)s.T * s.E * s.C, h->stream));
        HIPCHK(h, hipMemsetAsync(h->st.cs_cur_hist, 0, sizeof(double) * (size_t)s.T * s.E * s.C, h->stream));
    }
    hipLaunchKernelGGL(ev2g_reset_kernel, dim3(s.n_groups), dim3(EV2G_BLOCK), 0, h->stream, s, h->st, obs, h->extras.obs_f32, (int)off);
    HIPCHK(h, hipGetLastError());
    h->current_step = 0;
    return EV2G_OK;
}

int ev2g_reset(ev2g_handle *h, double *obs) { return ev2g_reset_ex(h, obs, h ? h->scn_off : 0); }
int ev2g_reset_f32(ev2g_handle *h, float *obs32, int64_t scenario_offset) {
    if (!h || !h->loaded) return fail(h, EV2G_ERR_STATE, "ev2g_reset: no scenarios loaded");
    const ev2g_step_extras keep = h->extras;
    if (obs32) h->extras.obs_f32 = obs32;   // (the reset kernel writes the float32 observation where the extras point; only the host-side copy changes here)
    const int rc = ev2g_reset_ex(h, nullptr, scenario_offset);
    h->extras = keep;
    return rc;
}
int64_t ev2g_scenario_offset(const ev2g_handle *h) { return h ? h->scn_off : 0; }
int ev2g_n_scenarios(const ev2g_handle *h) { return h ? h->M : 0; }

int ev2g_set_step_extras(ev2g_handle *h, const ev2g_step_extras *x) {
    if (!h) return EV2G_ERR_ARG;
    if (x && x->cost && h->cfg.cost_kind == EV2G_COST_NONE)
        return fail(h, EV2G_ERR_ARG, "ev2g_set_step_extras: a cost buffer needs ev2g_config.cost_kind != EV2G_COST_NONE");
    h->extras = x ? *x : ev2g_step_extras{};
    if (h->loaded && h->d_v2p) {   // refresh the extras inside the device-resident parameter block (stream-ordered)
        (void)hipSetDevice(h->device);
        struct { void *cost; long long cs; void *o32; long long os; const void *a32; } blk{
            h->extras.cost, h->extras.cost_step_stride, h->extras.obs_f32, h->extras.obs_f32_step_stride, h->extras.actions_f32};
        static_assert(sizeof(blk) == sizeof(V2P) - offsetof(V2P, x_cost), "StepExtras block is the tail of V2P");
        HIPCHK(h, hipMemcpyAsync((char *)h->d_v2p + offsetof(V2P, x_cost), &blk, sizeof blk, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));   // `blk` is a stack temporary
    }
    return EV2G_OK;
}

// StepIO of one call: the caller's buffers plus the scenario-pool window; step0 offsets the sticky extras' step strides
static StepIO make_io(const ev2g_handle *h, const double *actions, long long a_stride, double *obs, long long o_stride,
                      double *reward, long long r_stride, uint8_t *done, long long d_stride, uint8_t *mask, long long m_stride,
                      long long step0, int auto_reset) {
    StepIO io{};
    io.actions = actions; io.a_stride = a_stride;
    io.obs = obs; io.o_stride = o_stride;
    io.reward = reward; io.r_stride = r_stride;
    io.done = done; io.d_stride = d_stride;
    io.mask = mask; io.m_stride = m_stride;
    io.scn_off = (int)h->scn_off;
    io.scn_stride = (auto_reset == EV2G_AUTO_RESET_NEXT) ? h->E % h->M : 0;
    io.step0 = (int)step0;
    io.log_soc = (h->cfg.flags & EV2G_FLAG_LOG_SOC) ? 1 : 0;
    io.act32 = actions ? nullptr : h->extras.actions_f32;
    io.obs32 = (h->extras.obs_f32 && h->extras.obs_f32_step_stride == 0) ? h->extras.obs_f32 : nullptr;
    return io;
}

static int launch_steps(ev2g_handle *h, const StepIO &io, int t0, int k, int auto_reset) {
    const DevScn &s = h->scn;
    if (h->wave_path) {
        // the fast path advances its output pointers by 32-bit byte strides
        const long long lim = 1ll << 32;
        const ev2g_step_extras &x = h->extras;
        if (io.a_stride * 8 >= lim || io.o_stride * 8 >= lim || io.r_stride * 8 >= lim || io.d_stride >= lim || io.m_stride >= lim ||
            x.cost_step_stride * 8 >= lim || x.obs_f32_step_stride * 4 >= lim || io.a_stride < 0 || io.o_stride < 0 || io.r_stride < 0 ||
            io.d_stride < 0 || io.m_stride < 0 || x.cost_step_stride < 0 || x.obs_f32_step_stride < 0)
            return fail(h, EV2G_ERR_ARG, "ev2g_step_n: a step stride is negative or reaches 4 GiB (unsupported by the fast-path kernel)");
        const V2P *pp = (const V2P *)h->d_v2p;
        const DevState &st = h->st;
        const FusedArgs fa0{};
        const WaveArgs wa{s.P, s.T, s.E, s.D, s.M, st.slab_port, st.slab_port_slice, st.hist,
                          st.env_acc, s.cs_pack, (char *)st.line, h->d_step_tab, (char *)st.port_dyn, s.dict, h->wave_epw, h->wave_es};
        // every float64 output present, no extras, no charger histories: the specialisation without their checks (not for the run-time rewards)
        // ... in two flavours: float64 actions in / float64 observations out (a loop that consumes them, the benchmark), or the policy
        // network's hand-over, float32 actions in / float32 observations out and no float64 observation (ev2g_rollout)
        const bool f64io = io.actions && io.obs && !x.obs_f32, f32io = !io.actions && io.act32 && !io.obs && io.obs32;
        const bool full0 = (f64io || f32io) && io.reward && io.done && io.mask && !x.cost && !(h->cfg.flags & EV2G_FLAG_LOG_CS_HISTORY) &&
                           !auto_reset && t0 + k <= s.T && std::min(s.reward_kind, 3) != 3 && !h->no_full;
        // ... and: SoC log on, one observation-head column pair per lane at most (PublicPST has no head table), three lanes for the history store
        const bool wide0 = full0 && (h->cfg.flags & EV2G_FLAG_LOG_SOC) && s.P >= 3 && !h->no_wide &&
                           s.P >= (s.state_kind == EV2G_STATE_PUBLIC_PST ? 3 : (s.state_kind == EV2G_STATE_V2G_PROFIT_MAX_LOADS ? 30 : 10));
        // outputs with step strides ([K, E, *] blocks): the wide float64 instantiation with running output pointers (3); elsewhere stride 0 only
        const bool strided = io.o_stride != 0 || io.r_stride != 0 || io.d_stride != 0 || io.m_stride != 0;
        const bool str3 = strided && wide0 && f64io && !h->no_strided;
        const bool full = full0 && (!strided || str3), wide = wide0 && full;
        h->last_spec = full ? (str3 ? 3 : (wide ? 2 : 1)) : 0;
        {   // why not the full instantiation: the FIRST thing the caller passed (or configured) that rules it out -- ev2g_last_launch_general_reason
            const char *why = "";
            if (!full) {
                if (h->no_full) why = "EV2G_NO_FULL is set (or the batch has more than 4094 efficiency tables)";
                else if (std::min(s.reward_kind, 3) == 3) why = "the reward function is one of the eight selected at run time (only the shipped configs' three are compiled in)";
                else if (h->cfg.flags & EV2G_FLAG_LOG_CS_HISTORY) why = "EV2G_FLAG_LOG_CS_HISTORY (charger histories)";
                else if (x.cost) why = "a cost buffer is registered (ev2g_set_step_extras)";
                else if (auto_reset) why = "auto_reset";
                else if (!(io.reward && io.done && io.mask)) why = "a reward / done / mask output is NULL";
                else if (!(f64io || f32io)) why = "the observation / action buffers are neither the float64 pair nor the float32 hand-over pair (e.g. obs NULL, or a float32 observation copy next to the float64 one)";
                else if (strided) why = "an output step stride is not 0 (strided outputs keep the specialisation only with float64 observations, EV2G_FLAG_LOG_SOC and an env wide enough for the wide instantiation)";
                else why = "the launch would run past the episode end";
            }
            h->general_reason = why;
        }
#define EV2G_WAVE_CASE(SK, RK)                                                                                              \
    case SK * 4 + RK:                                                                                                       \
        if (full && RK != 3 && str3)                                                                                        \
            hipLaunchKernelGGL((ev2g_step_wave<SK, (RK == 3 ? 0 : RK), false, 3>), dim3(s.n_groups), dim3(EV2G_WAVE_BLOCK), h->lds_bytes, \
                               h->stream, pp, io, t0, k, auto_reset, wa, fa0);                                                   \
        else if (full && RK != 3 && wide && f32io)                                                                          \
            hipLaunchKernelGGL((ev2g_step_wave<SK, (RK == 3 ? 0 : RK), true, 2>), dim3(s.n_groups), dim3(EV2G_WAVE_BLOCK), h->lds_bytes, \
                               h->stream, pp, io, t0, k, auto_reset, wa, fa0);                                                   \
        else if (full && RK != 3 && f32io)                                                                                  \
            hipLaunchKernelGGL((ev2g_step_wave<SK, (RK == 3 ? 0 : RK), true, 1>), dim3(s.n_groups), dim3(EV2G_WAVE_BLOCK), h->lds_bytes, \
                               h->stream, pp, io, t0, k, auto_reset, wa, fa0);                                                   \
        else if (full && RK != 3 && wide)                                                                                   \
            hipLaunchKernelGGL((ev2g_step_wave<SK, (RK == 3 ? 0 : RK), false, 2>), dim3(s.n_groups), dim3(EV2G_WAVE_BLOCK), h->lds_bytes, \
                               h->stream, pp, io, t0, k, auto_reset, wa, fa0);                                                   \
        else if (full && RK != 3)                                                                                           \
            hipLaunchKernelGGL((ev2g_step_wave<SK, (RK == 3 ? 0 : RK), false, 1>), dim3(s.n_groups), dim3(EV2G_WAVE_BLOCK), h->lds_bytes, \
                               h->stream, pp, io, t0, k, auto_reset, wa, fa0);                                                   \
        else if (!io.actions)                                                                                               \
            hipLaunchKernelGGL((ev2g_step_wave<SK, RK, true>), dim3(s.n_groups), dim3(EV2G_WAVE_BLOCK), h->lds_bytes,       \
                               h->stream, pp, io, t0, k, auto_reset, wa, fa0);                                                   \
        else                                                                                                                \
            hipLaunchKernelGGL((ev2g_step_wave<SK, RK, false>), dim3(s.n_groups), dim3(EV2G_WAVE_BLOCK), h->lds_bytes,      \
                               h->stream, pp, io, t0, k, auto_reset, wa, fa0);                                                   \
        break;
        switch (s.state_kind * 4 + std::min(s.reward_kind, 3)) {   // rewards beyond the three compiled-in ones share instantiation 3
#ifdef EV2G_ONLY_00   /* tuning builds (tools/): one specialisation, seconds to compile */
            case 0:
                if (str3) hipLaunchKernelGGL((ev2g_step_wave<0, 0, false, 3>), dim3(s.n_groups), dim3(EV2G_WAVE_BLOCK), h->lds_bytes, h->stream, pp, io, t0, k, auto_reset, wa, fa0);
                else if (wide && f32io) hipLaunchKernelGGL((ev2g_step_wave<0, 0, true, 2>), dim3(s.n_groups), dim3(EV2G_WAVE_BLOCK), h->lds_bytes, h->stream, pp, io, t0, k, auto_reset, wa, fa0);
                else if (full && f32io) hipLaunchKernelGGL((ev2g_step_wave<0, 0, true, 1>), dim3(s.n_groups), dim3(EV2G_WAVE_BLOCK), h->lds_bytes, h->stream, pp, io, t0, k, auto_reset, wa, fa0);
                else if (wide) hipLaunchKernelGGL((ev2g_step_wave<0, 0, false, 2>), dim3(s.n_groups), dim3(EV2G_WAVE_BLOCK), h->lds_bytes, h->stream, pp, io, t0, k, auto_reset, wa, fa0);
                else if (full) hipLaunchKernelGGL((ev2g_step_wave<0, 0, false, 1>), dim3(s.n_groups), dim3(EV2G_WAVE_BLOCK), h->lds_bytes, h->stream, pp, io, t0, k, auto_reset, wa, fa0);
                else if (!io.actions) hipLaunchKernelGGL((ev2g_step_wave<0, 0, true>), dim3(s.n_groups), dim3(EV2G_WAVE_BLOCK), h->lds_bytes, h->stream, pp, io, t0, k, auto_reset, wa, fa0);
                else hipLaunchKernelGGL((ev2g_step_wave<0, 0, false>), dim3(s.n_groups), dim3(EV2G_WAVE_BLOCK), h->lds_bytes, h->stream, pp, io, t0, k, auto_reset, wa, fa0);
                break;
            default: return fail(h, EV2G_ERR_ARG, "EV2G_ONLY_00 build: only the cfg2 specialisation exists");
#else
            EV2G_WAVE_CASE(0, 0) EV2G_WAVE_CASE(0, 1) EV2G_WAVE_CASE(0, 2) EV2G_WAVE_CASE(0, 3)
            EV2G_WAVE_CASE(1, 0) EV2G_WAVE_CASE(1, 1) EV2G_WAVE_CASE(1, 2) EV2G_WAVE_CASE(1, 3)
            EV2G_WAVE_CASE(2, 0) EV2G_WAVE_CASE(2, 1) EV2G_WAVE_CASE(2, 2) EV2G_WAVE_CASE(2, 3)
#endif
        }
#undef EV2G_WAVE_CASE
        HIPCHK(h, hipGetLastError());
        return EV2G_OK;
    }
    // the general kernel's instantiation for the default plugin pair launched with everything present (ev2g_step_v2.h, SPEC)
    const bool spec = h->block && s.state_kind == EV2G_STATE_V2G_PROFIT_MAX_LOADS && s.reward_kind == 0 && s.npc == 1 && io.actions && io.obs &&
                      io.reward && io.done && io.mask && !h->extras.cost && !h->extras.obs_f32 && !(h->cfg.flags & EV2G_FLAG_LOG_CS_HISTORY) &&
                      (h->cfg.flags & EV2G_FLAG_LOG_SOC) && io.o_stride == 0 && io.r_stride == 0 && io.d_stride == 0 && io.m_stride == 0 &&
                      !auto_reset && t0 + k <= s.T && h->pow2_dt && !h->no_full;
    h->last_spec = h->block ? (spec ? (h->big_path ? 5 : 1) : 0) : -1;
    if (spec && h->big_path) {   // big envs: 512 threads, two ports per home lane, two workgroups per CU (ev2g_step_big.h)
        hipLaunchKernelGGL(ev2g_step_big<EV2G_BIG_BLOCK>, dim3(s.E), dim3(EV2G_BIG_BLOCK), h->lds_big, h->stream, (const V2P *)h->d_v2p, io, t0, k, h->big_args);
        HIPCHK(h, hipGetLastError());
        return EV2G_OK;
    }
    if (spec) {
        switch (h->block) {
        case 256: hipLaunchKernelGGL((ev2g_step_v2<256, 1>), dim3(s.n_groups), dim3(256), h->lds_bytes, h->stream, (const V2P *)h->d_v2p, io, t0, k, auto_reset); break;
        case 512: hipLaunchKernelGGL((ev2g_step_v2<512, 1>), dim3(s.n_groups), dim3(512), h->lds_bytes, h->stream, (const V2P *)h->d_v2p, io, t0, k, auto_reset); break;
        default: hipLaunchKernelGGL((ev2g_step_v2<1024, 1>), dim3(s.n_groups), dim3(1024), h->lds_bytes, h->stream, (const V2P *)h->d_v2p, io, t0, k, auto_reset); break;
        }
        HIPCHK(h, hipGetLastError());
        return EV2G_OK;
    }
    switch (h->block) {
    case 256:
        hipLaunchKernelGGL(ev2g_step_v2<256>, dim3(s.n_groups), dim3(256), h->lds_bytes, h->stream, (const V2P *)h->d_v2p, io, t0, k, auto_reset);
        break;
    case 512:
        hipLaunchKernelGGL(ev2g_step_v2<512>, dim3(s.n_groups), dim3(512), h->lds_bytes, h->stream, (const V2P *)h->d_v2p, io, t0, k, auto_reset);
        break;
    case 1024:
        hipLaunchKernelGGL(ev2g_step_v2<1024>, dim3(s.n_groups), dim3(1024), h->lds_bytes, h->stream, (const V2P *)h->d_v2p, io, t0, k, auto_reset);
        break;
    default:
        hipLaunchKernelGGL(ev2g_step_kernel, dim3(s.n_groups), dim3(EV2G_BLOCK), h->lds_bytes, h->stream, s, h->st, io,
                           StepExtras{h->extras.cost, h->extras.cost_step_stride, h->extras.obs_f32, h->extras.obs_f32_step_stride, h->extras.actions_f32},
                           t0, k, auto_reset);
    }
    HIPCHK(h, hipGetLastError());
    return EV2G_OK;
}

int ev2g_step(ev2g_handle *h, const double *actions, double *obs, double *reward, uint8_t *done, uint8_t *action_mask) {
    if (!h || !h->loaded) return fail(h, EV2G_ERR_STATE, "ev2g_step: no scenarios loaded");
    if (!actions && !h->extras.actions_f32) return fail(h, EV2G_ERR_ARG, "ev2g_step: actions is null (and no float32 actions are set)");
    if (h->current_step >= h->T)
        return fail(h, EV2G_ERR_DONE, "ev2g_step: episode is done, reset the environment (ev2gym_env.py:343)");
    (void)hipSetDevice(h->device);
    const StepIO io = make_io(h, actions, 0, obs, 0, reward, 0, done, 0, action_mask, 0, 0, 0);
    int rc = launch_steps(h, io, h->current_step, 1, 0);
    if (rc) return rc;
    h->current_step += 1;
    h->timed = false;
    return EV2G_OK;
}

int ev2g_step_n(ev2g_handle *h, int k_steps, int mode, const double *actions, int64_t a_stride, double *obs,
                int64_t o_stride, double *reward, int64_t r_stride, uint8_t *done, int64_t d_stride, uint8_t *mask,
                int64_t m_stride, int auto_reset) {
    if (!h || !h->loaded) return fail(h, EV2G_ERR_STATE, "ev2g_step_n: no scenarios loaded");
    if ((!actions && !h->extras.actions_f32) || k_steps < 0) return fail(h, EV2G_ERR_ARG, "ev2g_step_n: bad arguments");
    (void)hipSetDevice(h->device);
    int rc = EV2G_OK;
    const long long adv = (auto_reset == EV2G_AUTO_RESET_NEXT) ? h->E % h->M : 0;   // pool offset advance per in-run reset
    h->ev_slot = (h->ev_slot + 1) % EV2G_EV_RING; h->ev_calls += 1; h->ev_valid[h->ev_slot] = false;
    HIPCHK(h, hipEventRecord(h->ev0s[h->ev_slot], h->stream));
    if (mode == EV2G_STEPN_PERSISTENT) {
        int k = k_steps;
        if (!auto_reset) k = std::min(k, h->T - h->current_step);
        // Per-session results (sess_final_cap, sess_abs_e) are indexed by pool session.  Two envs that run the same scenario inside
        // ONE launch store to the same slots from different XCDs, whose L2s are written back at the kernel boundary in no defined
        // order, so a launch covers several AUTO_RESET_NEXT episodes only while their windows are disjoint: (resets + 1) * E <= M.
        // Otherwise the run is cut at the episode ends (one launch per episode, host-side reset in between).
        int resets = 0;
        { int t = h->current_step; for (int i = 0; i < k; i++) { if (t >= h->T) { t = 0; resets++; } t++; } }
        if (adv != 0 && resets > 0 && (long long)(resets + 1) * h->E > h->M) {
            for (int i0 = 0; i0 < k;) {
                if (h->current_step >= h->T) {
                    int r2 = ev2g_reset_ex(h, nullptr, h->scn_off + adv);
                    if (r2) return r2;
                }
                const int kc = std::min(k - i0, h->T - h->current_step);
                const StepIO io = make_io(h, actions ? actions + (long long)i0 * a_stride : nullptr, a_stride,
                                          obs ? obs + (long long)i0 * o_stride : nullptr, o_stride,
                                          reward ? reward + (long long)i0 * r_stride : nullptr, r_stride,
                                          done ? done + (long long)i0 * d_stride : nullptr, d_stride,
                                          mask ? mask + (long long)i0 * m_stride : nullptr, m_stride, i0, 0);
                int r2 = launch_steps(h, io, h->current_step, kc, 0);
                if (r2) return r2;
                h->current_step += kc;
                i0 += kc;
            }
            HIPCHK(h, hipEventRecord(h->ev1s[h->ev_slot], h->stream)); h->ev_valid[h->ev_slot] = true;
            h->timed = true;
            return rc;
        }
        const StepIO io = make_io(h, actions, a_stride, obs, o_stride, reward, r_stride, done, d_stride, mask, m_stride, 0, auto_reset);
        if (k > 0) rc = launch_steps(h, io, h->current_step, k, auto_reset);
        if (rc) return rc;
        if (auto_reset) {
            // replay the step counter on the host: reset happens lazily before the step that follows a terminal one
            int t = h->current_step;
            for (int i = 0; i < k; i++) { if (t >= h->T) { t = 0; h->scn_off = (h->scn_off + adv) % h->M; } t++; }
            h->current_step = t;
        } else {
            h->current_step += k;
        }
        if (k < k_steps) rc = fail(h, EV2G_ERR_DONE, "ev2g_step_n: episode finished before k_steps (auto_reset off)");
    } else {
        for (int i = 0; i < k_steps; i++) {
            if (h->current_step >= h->T) {
                if (!auto_reset) { rc = fail(h, EV2G_ERR_DONE, "ev2g_step_n: episode finished before k_steps (auto_reset off)"); break; }
                int r2 = ev2g_reset_ex(h, nullptr, h->scn_off + adv);
                if (r2) return r2;
            }
            const StepIO io = make_io(h, actions ? actions + (long long)i * a_stride : nullptr, a_stride,   // (float32 actions: the kernel applies step0 * a_stride)
                                      obs ? obs + (long long)i * o_stride : nullptr, 0,
                                      reward ? reward + (long long)i * r_stride : nullptr, 0,
                                      done ? done + (long long)i * d_stride : nullptr, 0,
                                      mask ? mask + (long long)i * m_stride : nullptr, 0, i, 0);
            int r2 = launch_steps(h, io, h->current_step, 1, 0);
            if (r2) return r2;
            h->current_step += 1;
        }
    }
    HIPCHK(h, hipEventRecord(h->ev1s[h->ev_slot], h->stream)); h->ev_valid[h->ev_slot] = true;
    h->timed = true;
    return rc;
}

// ---- policy in the loop -------------------------------------------------------------------------------------------
struct ev2g_mlp {
    MlpDev dev{};
    std::vector<void *> allocs;
    size_t lds = 0;
    const void *fn = nullptr;   // the kernel for this shape
    int rows = EV2G_MLP_ROWS;   // env rows per workgroup of that kernel
    int threads = EV2G_MLP_BLOCK;
    // batches of more rows than 16 x CUs: the same streaming kernel with 32 rows per workgroup -- a weight fragment then feeds two MFMAs, and the
    // weights are streamed once per CU instead of once per 16-row workgroup (two or more of which would share a CU)
    const void *fn_big = nullptr;
    size_t lds_big = 0;
    int rows_big = 0, threads_big = 0, big_from = 0;
    int s16_ks1 = 0, s16_nt1 = 0, s16_nt2 = 0, s16_nt3 = 0, s16_nw = 0;   // the streaming kernel's fragment packing (0: another kernel's)
};

// the fixed-shape kernels exist for the layer widths of the shipped configs (obs 162 / 63 -> 400 -> 300 -> ports); anything else
// runs the generic one
static const void *mlp_kernel_for(const MlpDev &d) {
    const int k1 = d.k1 / 16, k2 = d.n1 / 16, k3 = d.n2 / 16;
    // (the fixed kernels unroll over at most 4 / 3 / 1 column tiles per wavefront: 400 -> 13 tiles, 300 -> 10, ports <= 128)
    if (d.n1 / 32 > 16 || d.n2 / 32 > 12 || d.n3 / 32 > 4) return (const void *)ev2g_mlp3_any;
    if (k1 == 11 && k2 == 26 && k3 == 20) return (const void *)ev2g_mlp3_fixed<11, 26, 20>;
    if (k1 == 4 && k2 == 26 && k3 == 20) return (const void *)ev2g_mlp3_fixed<4, 26, 20>;
    return (const void *)ev2g_mlp3_any;
}

// the 16-row streaming kernel (ev2g_mlp3_s16) exists for the shipped shapes; EV2G_MLP_OLD=1 keeps round 3's 32-row kernel (A/B runs)
struct MlpS16Pick { const void *fn; size_t lds; int ks1, nt1, nt2, nt3, nw, threads; };
// nw: bf16 terms per weight -- 1: the bf16 network; 2 / 3: the float32 network as split bf16 operands (EV2G_MLP_F32 / EV2G_MLP_F32X3, ev2g_mlp.h)
static MlpS16Pick mlp_s16_for(int d_in, int h1, int h2, int d_out, int nw) {
    // The instantiations are for the shipped shapes (162 / 63 observations -> 400 -> 300 -> 50 / 20 ports); a network that FITS one of them runs on
    // it zero-padded (weights and biases of the missing rows / columns are zeros, ReLU(0) = 0): any input up to 192 (64), hidden layers up to
    // 400 / 304, outputs up to 64 (32).  Small networks (both hidden layers under 128) keep the generic kernel: they would pay the full-size stream.
    int ks1 = (d_in + 31) / 32, nt1 = (h1 + 15) / 16, nt2 = (h2 + 15) / 16, nt3 = (d_out + 15) / 16;
    if (nt1 <= 25 && nt2 <= 19 && (h1 >= 128 || h2 >= 128)) {
        if (ks1 <= 2 && nt3 <= 2) { ks1 = 2; nt1 = 25; nt2 = 19; nt3 = 2; }
        else if (ks1 <= 6 && nt3 <= 4) { ks1 = 6; nt1 = 25; nt2 = 19; nt3 = 4; }
    }
    const char *old = std::getenv("EV2G_MLP_OLD");
    if (old && old[0] == '1') return {nullptr, 0, 0, 0, 0, 0, 0, 0};
    // the bf16 network runs eight wavefronts per workgroup (two per SIMD: one's epilogue and LDS waits under the other's MFMAs -- 7.48 -> 7.39 us at
    // 162 inputs, 6.35 -> 5.88 at 63); the float32 modes need the registers of four.  EV2G_MLP_WAVES=4 selects four for the bf16 network (A/B runs).
    const char *w8 = std::getenv("EV2G_MLP_WAVES");
    const int wv = (nw == 1 && !(w8 && w8[0] == '4')) ? 8 : 4;
#define EV2G_S16_CASE(K, A, B, Cc, N, W) \
    if (ks1 == K && nt1 == A && nt2 == B && nt3 == Cc && nw == N && wv == W) return {(const void *)ev2g_mlp3_s16<K, A, B, Cc, N, W>, MlpS16<K, A, B, Cc, N, W>::lds_bytes, ks1, nt1, nt2, nt3, nw, W * 64};
    EV2G_S16_CASE(6, 25, 19, 4, 1, 4) EV2G_S16_CASE(2, 25, 19, 2, 1, 4)
    EV2G_S16_CASE(6, 25, 19, 4, 2, 4) EV2G_S16_CASE(2, 25, 19, 2, 2, 4)
    EV2G_S16_CASE(6, 25, 19, 4, 3, 4) EV2G_S16_CASE(2, 25, 19, 2, 3, 4)
    EV2G_S16_CASE(6, 25, 19, 4, 1, 8) EV2G_S16_CASE(2, 25, 19, 2, 1, 8)
#undef EV2G_S16_CASE
    return {nullptr, 0, 0, 0, 0, 0, 0, 0};
}

static uint16_t host_bf16(float f) {   // round to nearest even (same as the kernel's)
    uint32_t u;
    std::memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// torch.nn.Linear weight W[n_out, n_in] -> MFMA B-fragment order [n_tile][k_step][lane][8] (bf16, zero padded):
// lane l of tile (nt, ks) holds B[k][j] = W[j][k] for j = nt*32 + (l & 31), k = ks*16 + (l >> 5)*8 + 0..7
static std::vector<uint16_t> pack_linear(const float *W, int n_out, int n_in, int N, int K) {
    const int NT = N / 32, KS = K / 16;
    std::vector<uint16_t> p((size_t)NT * KS * 64 * 8, 0);
    for (int nt = 0; nt < NT; nt++)
        for (int ks = 0; ks < KS; ks++)
            for (int l = 0; l < 64; l++) {
                const int j = nt * 32 + (l & 31);
                for (int i = 0; i < 8; i++) {
                    const int k = ks * 16 + (l >> 5) * 8 + i;
                    if (j < n_out && k < n_in) p[(((size_t)nt * KS + ks) * 64 + l) * 8 + i] = host_bf16(W[(size_t)j * n_in + k]);
                }
            }
    return p;
}

// ... and for ev2g_mlp3_s16 (weights are the MFMA's A operand there): [tile of 16 outputs][k-step of 32][term][lane][8],
// lane l holds W[tile*16 + (l & 15)][ks*32 + 8*(l >> 4) + 0..7]; term t of NW is the bf16 rounding of what terms 0..t-1 left of the float32 weight
static std::vector<uint16_t> pack_linear_s16(const float *W, int n_out, int n_in, int NT, int KS, int NW) {
    std::vector<uint16_t> p((size_t)NT * KS * NW * 64 * 8, 0);
    for (int t = 0; t < NT; t++)
        for (int ks = 0; ks < KS; ks++)
            for (int l = 0; l < 64; l++) {
                const int j = t * 16 + (l & 15);
                for (int i = 0; i < 8; i++) {
                    const int k = ks * 32 + (l >> 4) * 8 + i;
                    if (j >= n_out || k >= n_in) continue;
                    float r = W[(size_t)j * n_in + k];
                    for (int q = 0; q < NW; q++) {
                        const uint16_t hb = host_bf16(r);
                        p[((((size_t)t * KS + ks) * NW + q) * 64 + l) * 8 + i] = hb;
                        uint32_t u = (uint32_t)hb << 16; float hf; std::memcpy(&hf, &u, 4);
                        r -= hf;   // (exact)
                    }
                }
            }
    return p;
}

// float32 weights in the operand order of ev2g_mlp32_layer: [n_tile][k_group of 8][lane][4], lane l <-> (n = tile*32 + (l & 31), k = 8 g + 4 (l >> 5) + 0..3)
static std::vector<float> pack_linear_f32(const float *W, int n_out, int n_in, int N, int K) {
    std::vector<float> v((size_t)N * K, 0.f);
    const int KG = K / 8;
    for (int nt = 0; nt < N / 32; nt++)
        for (int g = 0; g < KG; g++)
            for (int l = 0; l < 64; l++)
                for (int j = 0; j < 4; j++) {
                    const int n = nt * 32 + (l & 31), k = g * 8 + 4 * (l >> 5) + j;
                    v[(((size_t)nt * KG + g) * 64 + l) * 4 + j] = (n < n_out && k < n_in) ? W[(size_t)n * n_in + k] : 0.f;
                }
    return v;
}

int ev2g_mlp_create_ex(ev2g_handle *h, int d_in, int h1, int h2, int d_out, const float *W1, const float *b1, const float *W2,
                    const float *b2, const float *W3, const float *b3, float out_lo, int precision, ev2g_mlp **out) {
    if (!h || !out || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || d_in <= 0 || h1 <= 0 || h2 <= 0 || d_out <= 0)
        return fail(h, EV2G_ERR_ARG, "ev2g_mlp_create: bad arguments");
    if (out_lo != -1.0f && out_lo != 0.0f) return fail(h, EV2G_ERR_ARG, "ev2g_mlp_create: out_lo must be -1 or 0");
    (void)hipSetDevice(h->device);
    auto up = [](int x, int m) { return (x + m - 1) / m * m; };
    ev2g_mlp *m = new ev2g_mlp();
    MlpDev &d = m->dev;
    d.d_in = d_in; d.h1 = h1; d.h2 = h2; d.d_out = d_out;
    d.k1 = up(d_in, 16); d.n1 = up(h1, 32); d.n2 = up(h2, 32); d.n3 = up(d_out, 32);
    d.out_lo = out_lo;
    d.dbg = nullptr;
#ifdef EV2G_MLP_TIMING
    { unsigned long long *p; if (dalloc(h, m->allocs, 16, &p)) { delete m; return EV2G_ERR_HIP; } d.dbg = p; }
#endif
#ifdef EV2G_F32_STAMPS
    { unsigned long long *p; if (dalloc(h, m->allocs, 8 * 16 * 8 + 8 * 16 * 16, &p)) { delete m; return EV2G_ERR_HIP; } d.dbg = p; }
#endif
    if (precision != EV2G_MLP_BF16 && precision != EV2G_MLP_F32 && precision != EV2G_MLP_F32X3) { delete m; return fail(h, EV2G_ERR_ARG, "ev2g_mlp_create_ex: precision must be EV2G_MLP_BF16, EV2G_MLP_F32 or EV2G_MLP_F32X3"); }
    const bool f32 = precision != EV2G_MLP_BF16;
    const MlpS16Pick s16 = mlp_s16_for(d_in, h1, h2, d_out, precision == EV2G_MLP_BF16 ? 1 : (precision == EV2G_MLP_F32 ? 2 : 3));
    m->lds = s16.fn ? s16.lds : (f32 ? ev2g_mlp32_lds_bytes(d) : ev2g_mlp_lds_bytes(d));
    if (s16.fn) { m->rows = EV2G_MLPS_ROWS; m->threads = s16.threads; m->s16_ks1 = s16.ks1; m->s16_nt1 = s16.nt1; m->s16_nt2 = s16.nt2; m->s16_nt3 = s16.nt3; m->s16_nw = s16.nw; }
    if (s16.fn && s16.nw == 1 && !std::getenv("EV2G_MLP_NO_BIG")) {
        if (s16.ks1 == 6 && s16.nt3 == 4) { m->fn_big = (const void *)ev2g_mlp3_s16<6, 25, 19, 4, 1, 4, 2>; m->lds_big = MlpS16<6, 25, 19, 4, 1, 4, 2>::lds_bytes; }
        else if (s16.ks1 == 2 && s16.nt3 == 2) { m->fn_big = (const void *)ev2g_mlp3_s16<2, 25, 19, 2, 1, 4, 2>; m->lds_big = MlpS16<2, 25, 19, 2, 1, 4, 2>::lds_bytes; }
        if (m->fn_big) {
            int cus = 256;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
            m->rows_big = 2 * EV2G_MLPS_ROWS; m->threads_big = 256; m->big_from = EV2G_MLPS_ROWS * cus + 1;
            if (m->lds_big > 48 * 1024) HIPCHK(h, hipFuncSetAttribute(m->fn_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m->lds_big));
        }
    }
    if (m->lds > 160 * 1024) { delete m; return fail(h, EV2G_ERR_ARG, "ev2g_mlp_create: layers too wide for the LDS-resident activations"); }
    int rc = 0;
    auto upw = [&](const std::vector<uint16_t> &v, const uint16_t **dst) { uint16_t *p; rc = upload(h, m->allocs, v.data(), v.size(), &p); *dst = p; return rc; };
    auto upb = [&](const float *b, int n, int N, const float **dst) { std::vector<float> v((size_t)N, 0.f); std::copy(b, b + n, v.begin()); float *p; rc = upload(h, m->allocs, v.data(), v.size(), &p); *dst = p; return rc; };
    auto upw32 = [&](const std::vector<float> &v, const uint16_t **dst) { float *p; rc = upload(h, m->allocs, v.data(), v.size(), &p); *dst = (const uint16_t *)p; return rc; };
    if (f32 && !s16.fn) {
        if (upw32(pack_linear_f32(W1, h1, d_in, d.n1, d.k1), &d.w1) || upw32(pack_linear_f32(W2, h2, h1, d.n2, d.n1), &d.w2) ||
            upw32(pack_linear_f32(W3, d_out, h2, d.n3, d.n2), &d.w3) || upb(b1, h1, d.n1, &d.b1) || upb(b2, h2, d.n2, &d.b2) || upb(b3, d_out, d.n3, &d.b3)) {
            free_pool(m->allocs); delete m; return rc;
        }
    } else if (s16.fn) {
        if (upw(pack_linear_s16(W1, h1, d_in, s16.nt1, s16.ks1, s16.nw), &d.w1) || upw(pack_linear_s16(W2, h2, h1, s16.nt2, (s16.nt1 * 16 + 31) / 32, s16.nw), &d.w2) ||
            upw(pack_linear_s16(W3, d_out, h2, s16.nt3, (s16.nt2 * 16 + 31) / 32, s16.nw), &d.w3)) {
            free_pool(m->allocs); delete m; return rc;
        }
        {   // the three bias vectors as ONE array (b1 | b2 | b3, each padded with zeros to its 16-column tiles): one coalesced load in the kernel
            std::vector<float> ball((size_t)(s16.nt1 + s16.nt2 + s16.nt3) * 16, 0.f);
            std::copy(b1, b1 + h1, ball.begin()); std::copy(b2, b2 + h2, ball.begin() + s16.nt1 * 16); std::copy(b3, b3 + d_out, ball.begin() + (s16.nt1 + s16.nt2) * 16);
            float *p;
            if ((rc = upload(h, m->allocs, ball.data(), ball.size(), &p))) { free_pool(m->allocs); delete m; return rc; }
            d.b1 = p; d.b2 = p + s16.nt1 * 16; d.b3 = p + (s16.nt1 + s16.nt2) * 16;
        }
    } else
    if (upw(pack_linear(W1, h1, d_in, d.n1, d.k1), &d.w1) || upw(pack_linear(W2, h2, h1, d.n2, d.n1), &d.w2) ||
        upw(pack_linear(W3, d_out, h2, d.n3, d.n2), &d.w3) || upb(b1, h1, d.n1, &d.b1) || upb(b2, h2, d.n2, &d.b2) || upb(b3, d_out, d.n3, &d.b3)) {
        free_pool(m->allocs); delete m; return rc;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));   // the staging vectors are temporaries
    m->fn = s16.fn ? s16.fn : (f32 ? (const void *)ev2g_mlp3_f32 : mlp_kernel_for(d));
    if (m->lds > 48 * 1024) HIPCHK(h, hipFuncSetAttribute(m->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m->lds));
    *out = m;
    return EV2G_OK;
}

int ev2g_mlp_create(ev2g_handle *h, int d_in, int h1, int h2, int d_out, const float *W1, const float *b1, const float *W2,
                    const float *b2, const float *W3, const float *b3, float out_lo, ev2g_mlp **out) {
    return ev2g_mlp_create_ex(h, d_in, h1, h2, d_out, W1, b1, W2, b2, W3, b3, out_lo, EV2G_MLP_BF16, out);
}

void ev2g_mlp_destroy(ev2g_handle *h, ev2g_mlp *m) {
    if (!m) return;
    if (h) { (void)hipSetDevice(h->device); (void)hipStreamSynchronize(h->stream); drop_rollout_graphs(h); }
    free_pool(m->allocs);
    delete m;
}

int ev2g_mlp_forward(ev2g_handle *h, const ev2g_mlp *m, const float *x, float *y, int n_rows) {
    if (!h || !m || !x || !y || n_rows <= 0) return fail(h, EV2G_ERR_ARG, "ev2g_mlp_forward: bad arguments");
    (void)hipSetDevice(h->device);
    MlpDev dev = m->dev;
    void *args[] = {&dev, &x, &y, &n_rows};
    if (m->fn_big && n_rows >= m->big_from)
        HIPCHK(h, hipLaunchKernel(m->fn_big, dim3((n_rows + m->rows_big - 1) / m->rows_big), dim3(m->threads_big), args, m->lds_big, h->stream));
    else
        HIPCHK(h, hipLaunchKernel(m->fn, dim3((n_rows + m->rows - 1) / m->rows), dim3(m->threads), args, m->lds, h->stream));
    return EV2G_OK;
}

#ifdef EV2G_F32_STAMPS
extern "C" int ev2g_mlp_debug_f32_stamps(ev2g_handle *h, const ev2g_mlp *m, unsigned long long *out1024) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out1024, m->dev.dbg, (8 * 16 * 8 + 8 * 16 * 16) * 8, hipMemcpyDeviceToHost));   // [workgroup 0..7][wavefront][stamp 0..7], then [workgroup][wavefront][16]: layer 3's k-steps
    return 0;
}
#endif
#ifdef EV2G_MLP_TIMING
int ev2g_mlp_debug_stamps(ev2g_handle *h, const ev2g_mlp *m, unsigned long long *out8) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out8, m->dev.dbg, 128, hipMemcpyDeviceToHost));   // (16 stamps: 0..7 the phases, 8.. finer ones of the prologue)
    return 0;
}
#endif

// ---- one launch per rollout segment (round 5): ev2g_step_wave<.., 1024, true> evaluates the policy between the steps, inside the launch ----
// Eligible: the fast path (3..64 ports: the shipped YAMLs' 25 chargers, BASELINE configs[1] / configs[4]'s 50; every env gets a wavefront of its own in
// this instantiation, whatever its width), a head-table state (V2G_profit_max_loads /
// V2G_profit_max), one of the three compiled-in rewards, EV2G_FLAG_LOG_SOC, no extras beyond the float32 hand-over, and the bf16 policy in the
// streaming kernel's 162 -> 400 -> 300 -> 64 packing.  Anything else (and EV2G_NO_FUSED=1) keeps the two launches per step.
static bool fused_eligible(const ev2g_handle *h, const ev2g_mlp *m) {
    const DevScn &s = h->scn;
    // (round 6: PublicPST too -- its 3 + 3 P <= 63 inputs and P <= 20 outputs in the 64 -> 400 -> 300 -> 32 packing; the other states in 192 -> 400 -> 300 -> 64)
    const bool pst = s.state_kind == EV2G_STATE_PUBLIC_PST;
    return h->wave_path && s.P >= 3 && s.P <= 64 && std::min(s.reward_kind, 3) != 3 && (h->cfg.flags & EV2G_FLAG_LOG_SOC) &&
           !(h->cfg.flags & EV2G_FLAG_LOG_CS_HISTORY) && !h->extras.cost && !h->no_full && !h->no_wide && (pst || (s.D & 1) == 0) &&
           m->s16_ks1 == (pst ? 2 : 6) && m->s16_nt1 == 25 && m->s16_nt2 == 19 && m->s16_nt3 == (pst ? 2 : 4) &&
           (m->s16_nw == 1 || (m->s16_nw == 2 && !std::getenv("EV2G_NO_FUSED_F32"))) &&   // (last session of round 6: the float32 policy, two bf16 terms per weight; one env per wavefront)
           !std::getenv("EV2G_NO_FUSED");
}
// k steps from the current one; obs0: the [E, D] float32 rows the first forward reads; obs / act / reward / done / mask: the rows of the segment's first
// step with their step strides (elements; 0 = one row, overwritten)
static int launch_fused(ev2g_handle *h, const ev2g_mlp *m, int k, const float *obs0, float *obs, long long o_stride, float *act, long long a_stride,
                        double *reward, long long r_stride, uint8_t *done, long long d_stride, uint8_t *mask, long long m_stride) {
    const DevScn &s = h->scn;
    const DevState &st = h->st;
    const long long lim = 1ll << 32;
    if (o_stride * 4 >= lim || a_stride * 4 >= lim || r_stride * 8 >= lim || d_stride >= lim || m_stride >= lim || o_stride < 0 || a_stride < 0 || r_stride < 0 || d_stride < 0 || m_stride < 0)
        return fail(h, EV2G_ERR_ARG, "ev2g_collect / ev2g_rollout: a step stride is negative or reaches 4 GiB");
    StepIO io = make_io(h, nullptr, a_stride, nullptr, o_stride, reward, r_stride, done, d_stride, mask, m_stride, 0, 0);
    io.act32 = act; io.obs32 = obs;
    // round 6: PublicPST envs of at most 32 ports go TWO to a wavefront (32 policy rows per workgroup; EV2G_FUSED_ONE_ENV=1: the one-env form, for A/B)
    const int ae = (s.state_kind == EV2G_STATE_PUBLIC_PST && s.P <= 32 && m->s16_nw == 1 && !std::getenv("EV2G_FUSED_ONE_ENV")) ? 2 : 1;
    const WaveArgs wa{s.P, s.T, s.E, s.D, s.M, st.slab_port, st.slab_port_slice, st.hist, st.env_acc, s.cs_pack, (char *)st.line, h->d_step_tab, (char *)st.port_dyn, s.dict, ae, ae == 1 ? s.P : 32};
    FusedArgs fa{};
    fa.m = m->dev; fa.obs0 = obs0;
    const V2P *pp = (const V2P *)h->d_v2p;
    const int nwf = m->s16_nw;
    const size_t lds = ev2g_fused_lds_bytes(ae, nwf);
    const dim3 grid((s.E + 16 * ae - 1) / (16 * ae)), block(EV2G_FUSED_BLOCK);
    const int t0 = h->current_step;
#define EV2G_FUSED_CASE(SK, RK)                                                                                                            \
    case SK * 4 + RK: {                                                                                                                    \
        auto kfn = ev2g_step_wave<SK, RK, true, 2, EV2G_FUSED_BLOCK, true>;                                                                 \
        if (!(h->fused_attr_mask & (1u << (SK * 4 + RK)))) {   /* function attributes are per device: once per handle, not per process */       \
            HIPCHK(h, hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                      \
            h->fused_attr_mask |= 1u << (SK * 4 + RK);                                                                                     \
        }                                                                                                                                  \
        hipLaunchKernelGGL(kfn, grid, block, lds, h->stream, pp, io, t0, k, 0, wa, fa);                                                     \
    } break;
#define EV2G_FUSED_CASE2(RK)                                                                                                               \
    case 16 + RK: {                                                                                                                        \
        auto kfn = ev2g_step_wave<1, RK, true, 2, EV2G_FUSED_BLOCK, true, 2>;                                                               \
        if (!(h->fused_attr_mask & (1u << (16 + RK)))) {                                                                                   \
            HIPCHK(h, hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                      \
            h->fused_attr_mask |= 1u << (16 + RK);                                                                                         \
        }                                                                                                                                  \
        hipLaunchKernelGGL(kfn, grid, block, lds, h->stream, pp, io, t0, k, 0, wa, fa);                                                     \
    } break;
#define EV2G_FUSED_CASEF(SK, RK)   /* the float32 policy (two bf16 terms per weight) inside the launch */                                  \
    case 20 + SK * 4 + RK: {                                                                                                               \
        auto kfn = ev2g_step_wave<SK, RK, true, 2, EV2G_FUSED_BLOCK, true, 1, 2>;                                                           \
        if (!(h->fused_attr_mask & (1u << (20 + SK * 4 + RK)))) {                                                                          \
            HIPCHK(h, hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                      \
            h->fused_attr_mask |= 1u << (20 + SK * 4 + RK);                                                                                \
        }                                                                                                                                  \
        hipLaunchKernelGGL(kfn, grid, block, lds, h->stream, pp, io, t0, k, 0, wa, fa);                                                     \
    } break;
    switch (nwf == 2 ? 20 + s.state_kind * 4 + std::min(s.reward_kind, 3) : (ae == 2 ? 16 + std::min(s.reward_kind, 3) : s.state_kind * 4 + std::min(s.reward_kind, 3))) {
        EV2G_FUSED_CASE(0, 0) EV2G_FUSED_CASE(0, 1) EV2G_FUSED_CASE(0, 2)
        EV2G_FUSED_CASEF(0, 0) EV2G_FUSED_CASEF(0, 1) EV2G_FUSED_CASEF(0, 2)
#ifndef EV2G_ONLY_00
        EV2G_FUSED_CASEF(2, 0) EV2G_FUSED_CASEF(2, 1) EV2G_FUSED_CASEF(2, 2)
        EV2G_FUSED_CASEF(1, 0) EV2G_FUSED_CASEF(1, 1) EV2G_FUSED_CASEF(1, 2)
        EV2G_FUSED_CASE(1, 0) EV2G_FUSED_CASE(1, 1) EV2G_FUSED_CASE(1, 2)
        EV2G_FUSED_CASE(2, 0) EV2G_FUSED_CASE(2, 1) EV2G_FUSED_CASE(2, 2)
        EV2G_FUSED_CASE2(0) EV2G_FUSED_CASE2(1) EV2G_FUSED_CASE2(2)
#endif
        default: return fail(h, EV2G_ERR_STATE, "internal: no fused instantiation for this plugin pair");
    }
#undef EV2G_FUSED_CASE
#undef EV2G_FUSED_CASE2
#undef EV2G_FUSED_CASEF
    HIPCHK(h, hipGetLastError());
    h->last_spec = 4;
    h->general_reason = "";
    return EV2G_OK;
}

int ev2g_rollout(ev2g_handle *h, const ev2g_mlp *m, int k_steps, double *reward, int64_t r_stride, uint8_t *done, int64_t d_stride,
                 uint8_t *mask, int64_t m_stride, int auto_reset) {
    if (!h || !h->loaded) return fail(h, EV2G_ERR_STATE, "ev2g_rollout: no scenarios loaded");
    if (!m || k_steps < 0) return fail(h, EV2G_ERR_ARG, "ev2g_rollout: bad arguments");
    const ev2g_step_extras &x = h->extras;
    if (!x.obs_f32 || !x.actions_f32 || x.obs_f32_step_stride != 0)
        return fail(h, EV2G_ERR_ARG, "ev2g_rollout: register float32 observation (step stride 0) and action buffers with ev2g_set_step_extras first");
    if (m->dev.d_in != h->D || m->dev.d_out != h->P) return fail(h, EV2G_ERR_ARG, "ev2g_rollout: actor shape != (obs dim, ports)");
    (void)hipSetDevice(h->device);
    const long long adv = (auto_reset == EV2G_AUTO_RESET_NEXT) ? h->E % h->M : 0;
    // the k x (actor forward, env step) launches; `capturing`: no host-side reset inside (the caller made sure none is needed)
    auto enqueue = [&](int k) -> int {
        for (int i = 0; i < k; i++) {
            if (h->current_step >= h->T) {
                if (!auto_reset) return fail(h, EV2G_ERR_DONE, "ev2g_rollout: episode finished before k_steps (auto_reset off)");
                int r2 = ev2g_reset_ex(h, nullptr, h->scn_off + adv);
                if (r2) return r2;
            }
            int r2 = ev2g_mlp_forward(h, m, x.obs_f32, (float *)x.actions_f32, h->E);
            if (r2) return r2;
            StepIO io = make_io(h, nullptr, 0, nullptr, 0, reward ? reward + (long long)i * r_stride : nullptr, 0,
                                done ? done + (long long)i * d_stride : nullptr, 0, mask ? mask + (long long)i * m_stride : nullptr, 0, 0, 0);
            if (x.cost) io.step0 = i;   // (a cost buffer may record every step; the float32 buffers do not advance)
            r2 = launch_steps(h, io, h->current_step, 1, 0);
            if (r2) return r2;
            h->current_step += 1;
        }
        return EV2G_OK;
    };
    h->ev_slot = (h->ev_slot + 1) % EV2G_EV_RING; h->ev_calls += 1; h->ev_valid[h->ev_slot] = false;
    HIPCHK(h, hipEventRecord(h->ev0s[h->ev_slot], h->stream));
    int rc = EV2G_OK;
    static const bool use_graphs = [] { const char *e = std::getenv("EV2G_ROLLOUT_GRAPHS"); return !(e && e[0] == '0'); }();
    const bool whole = h->current_step + k_steps <= h->T;   // no episode end inside the segment: nothing but kernel launches
    if (whole && k_steps >= 1 && reward && done && mask && fused_eligible(h, m)) {   // ONE launch: the policy between the steps, inside it
        rc = launch_fused(h, m, k_steps, x.obs_f32, x.obs_f32, 0, (float *)x.actions_f32, 0, reward, r_stride, done, d_stride, mask, m_stride);
        if (rc) return rc;
        h->current_step += k_steps;
    } else
    if (use_graphs && whole && k_steps >= 4) {
        ev2g_handle::RolloutGraph *hit = nullptr;
        for (auto &g : h->rollout_graphs)
            if (g.mlp == m && g.k == k_steps && g.t0 == h->current_step && g.scn_off == h->scn_off && g.rew == reward && g.done == done &&
                g.mask == mask && g.rs == r_stride && g.ds == d_stride && g.ms == m_stride && std::memcmp(&g.x, &x, sizeof x) == 0) { hit = &g; break; }
        if (!hit) {
            const int t_before = h->current_step;
            hipGraph_t graph = nullptr;
            HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
            rc = enqueue(k_steps);
            const hipError_t ce = hipStreamEndCapture(h->stream, &graph);
            h->current_step = t_before;
            if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
            if (ce != hipSuccess) return fail(h, EV2G_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
            hipGraphExec_t exec = nullptr;
            const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ie != hipSuccess) return fail(h, EV2G_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie));
            if (h->rollout_graphs.size() >= 64) {   // bounded cache: drop the oldest
                (void)hipGraphExecDestroy(h->rollout_graphs.front().exec);
                h->rollout_graphs.erase(h->rollout_graphs.begin());
            }
            h->rollout_graphs.push_back({m, k_steps, t_before, h->scn_off, reward, done, mask, r_stride, d_stride, m_stride, x, exec});
            hit = &h->rollout_graphs.back();
        }
        HIPCHK(h, hipGraphLaunch(hit->exec, h->stream));
        h->current_step += k_steps;
        h->graph_launches++;
    } else {
        rc = enqueue(k_steps);
    }
    HIPCHK(h, hipEventRecord(h->ev1s[h->ev_slot], h->stream)); h->ev_valid[h->ev_slot] = true;
    h->timed = true;
    return rc;
}

long long ev2g_rollout_graph_launches(const ev2g_handle *h) { return h ? h->graph_launches : 0; }

int ev2g_collect(ev2g_handle *h, const ev2g_mlp *m, int k_steps, const ev2g_transitions *tr) {
    if (!h || !h->loaded) return fail(h, EV2G_ERR_STATE, "ev2g_collect: no scenarios loaded");
    if (!m || !tr || k_steps < 0 || !tr->obs || !tr->actions || !tr->reward || !tr->done || !tr->mask)
        return fail(h, EV2G_ERR_ARG, "ev2g_collect: null argument (every transition array is required)");
    if (m->dev.d_in != h->D || m->dev.d_out != h->P) return fail(h, EV2G_ERR_ARG, "ev2g_collect: actor shape != (obs dim, ports)");
    if (h->current_step + k_steps > h->T) return fail(h, EV2G_ERR_DONE, "ev2g_collect: the segment would run past the episode end");
    (void)hipSetDevice(h->device);
    const size_t ED = (size_t)h->E * h->D, EP = (size_t)h->E * h->P;
    // On the fast path the policy hand-over instantiations take their float32 buffers per launch (StepIO::act32 / obs32): every step
    // reads and writes the caller's rows directly.  Elsewhere (general kernels; run-time rewards; a registered cost buffer) the step
    // works on the registered hand-over buffers of ev2g_set_step_extras and the rows are copied device-to-device around it.
    const ev2g_step_extras &x = h->extras;
    const bool direct = h->wave_path && !x.cost && !x.obs_f32 && !x.actions_f32 && !(h->cfg.flags & EV2G_FLAG_LOG_CS_HISTORY) &&
                        std::min(h->scn.reward_kind, 3) != 3 && !h->no_full;
    if (!direct && !(x.obs_f32 && x.actions_f32 && x.obs_f32_step_stride == 0))
        return fail(h, EV2G_ERR_ARG, "ev2g_collect: this configuration steps through the registered float32 hand-over buffers: register them with "
                                     "ev2g_set_step_extras (observation step stride 0) first");
    h->ev_slot = (h->ev_slot + 1) % EV2G_EV_RING; h->ev_calls += 1; h->ev_valid[h->ev_slot] = false;
    HIPCHK(h, hipEventRecord(h->ev0s[h->ev_slot], h->stream));
    if (direct && k_steps >= 1 && fused_eligible(h, m)) {   // ONE launch for the segment: rows read and written in place, the policy inside the launch
        const int rc = launch_fused(h, m, k_steps, tr->obs, tr->obs + ED, (long long)ED, tr->actions, (long long)EP, tr->reward, h->E, tr->done, h->E, tr->mask, (long long)EP);
        if (rc) return rc;
        h->current_step += k_steps;
        k_steps = 0;
    }
    for (int i = 0; i < k_steps; i++) {
        float *obs_i = tr->obs + (size_t)i * ED, *obs_n = obs_i + ED, *act_i = tr->actions + (size_t)i * EP;
        int rc;
        if (direct) {
            if ((rc = ev2g_mlp_forward(h, m, obs_i, act_i, h->E))) return rc;
            StepIO io = make_io(h, nullptr, 0, nullptr, 0, tr->reward + (size_t)i * h->E, 0, tr->done + (size_t)i * h->E, 0, tr->mask + (size_t)i * EP, 0, 0, 0);
            io.act32 = act_i; io.obs32 = obs_n;
            if ((rc = launch_steps(h, io, h->current_step, 1, 0))) return rc;
            if (h->last_spec <= 0) return fail(h, EV2G_ERR_STATE, "ev2g_collect: internal: the direct path needs the full instantiation");
        } else {
            if (i == 0) HIPCHK(h, hipMemcpyAsync(x.obs_f32, obs_i, ED * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
            if ((rc = ev2g_mlp_forward(h, m, x.obs_f32, (float *)x.actions_f32, h->E))) return rc;
            StepIO io = make_io(h, nullptr, 0, nullptr, 0, tr->reward + (size_t)i * h->E, 0, tr->done + (size_t)i * h->E, 0, tr->mask + (size_t)i * EP, 0, 0, 0);
            if ((rc = launch_steps(h, io, h->current_step, 1, 0))) return rc;
            HIPCHK(h, hipMemcpyAsync(act_i, x.actions_f32, EP * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(h, hipMemcpyAsync(obs_n, x.obs_f32, ED * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
        }
        h->current_step += 1;
    }
    HIPCHK(h, hipEventRecord(h->ev1s[h->ev_slot], h->stream)); h->ev_valid[h->ev_slot] = true;
    h->timed = true;
    return EV2G_OK;
}

double ev2g_last_step_n_kernel_ms(ev2g_handle *h) {
    return ev2g_step_n_kernel_ms_back(h, 0);
}

double ev2g_step_n_kernel_ms_back(ev2g_handle *h, int back) {
    if (!h || !h->timed || back < 0 || back >= EV2G_EV_RING || back >= h->ev_calls) return -1.0;
    const int slot = ((h->ev_slot - back) % EV2G_EV_RING + EV2G_EV_RING) % EV2G_EV_RING;
    if (!h->ev_valid[slot]) return -1.0;
    if (hipEventSynchronize(h->ev1s[slot]) != hipSuccess) return -1.0;
    float ms = 0;
    if (hipEventElapsedTime(&ms, h->ev0s[slot], h->ev1s[slot]) != hipSuccess) return -1.0;
    return (double)ms;
}

int ev2g_check_faults(ev2g_handle *h, int32_t *first_bad_env) {
    if (!h || !h->loaded) return fail(h, EV2G_ERR_STATE, "ev2g_check_faults: no scenarios loaded");
    (void)hipSetDevice(h->device);
    std::vector<int> f(h->E);
    HIPCHK(h, hipMemcpyAsync(f.data(), h->st.env_fault, sizeof(int) * h->E, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int e = 0; e < h->E; e++)
        if (f[e]) {
            if (first_bad_env) *first_bad_env = e;
            return fail(h, EV2G_ERR_OVERCURRENT, "charger over-current: sum of amps is higher than max charge current (ev_charger.py:203-205)");
        }
    return EV2G_OK;
}

static int launch_stats(ev2g_handle *h, double *stats, bool reset, double *obs, long long off, float *obs32 = nullptr) {
    (void)hipSetDevice(h->device);
    // two envs per wavefront where an env's sessions fit 32 lanes with room to spare (PublicPST: ~14 per env -- 74.6 -> 58.0 us at cfg3; at
    // cfg2's ~35 per env half of the lanes would need a second pass: 47 -> 52 us, so it keeps a wavefront per env).  Refillable pools: by the
    // session slots per scenario, not by what the loaded batch happened to hold -- the choice (and the summation order) stays put across refills.
    const long long per_scn = h->sess_cap > 0 ? (long long)h->sess_cap : (h->S + h->M - 1) / std::max(h->M, 1);
    const bool pair = (h->sess_cap > 0 ? per_scn <= 48 : h->S <= (long long)h->M * 24) && h->C <= 32 && !std::getenv("EV2G_STATS_ONE_ENV");
    const dim3 grid(pair ? (h->E + 1) / 2 : h->E);
#define EV2G_STATS_LAUNCH(EPWS, RESET)                                                                                                    \
    hipLaunchKernelGGL((ev2g_stats_kernel<EPWS, RESET>), grid, dim3(64), (size_t)EV2G_STATS_LK * 64 * sizeof(double), h->stream, h->scn, h->st, (int)h->scn_off, (const double *)h->d_ss_afap, \
                       h->current_step, stats, (int)off, obs, RESET ? (obs32 ? obs32 : (float *)h->extras.obs_f32) : (float *)nullptr)
    if (pair) { if (reset) EV2G_STATS_LAUNCH(2, true); else EV2G_STATS_LAUNCH(2, false); }
    else { if (reset) EV2G_STATS_LAUNCH(1, true); else EV2G_STATS_LAUNCH(1, false); }
#undef EV2G_STATS_LAUNCH
    HIPCHK(h, hipGetLastError());
    return EV2G_OK;
}

int ev2g_get_stats(ev2g_handle *h, double *stats) {
    if (!h || !h->loaded) return fail(h, EV2G_ERR_STATE, "ev2g_get_stats: no scenarios loaded");
    if (!stats) return fail(h, EV2G_ERR_ARG, "ev2g_get_stats: null output");
    return launch_stats(h, stats, false, nullptr, 0);
}

static int stats_reset_impl(ev2g_handle *h, double *stats, double *obs, float *obs32, int64_t scenario_offset) {
    if (!h || !h->loaded) return fail(h, EV2G_ERR_STATE, "ev2g_get_stats_reset: no scenarios loaded");
    if (!stats) return fail(h, EV2G_ERR_ARG, "ev2g_get_stats_reset: null output");
    long long off = scenario_offset % (long long)h->scn.M;
    if (off < 0) off += h->scn.M;
    const int rc = launch_stats(h, stats, true, obs, off, obs32);
    if (rc) return rc;
    if (h->st.cs_power_hist) {
        HIPCHK(h, hipMemsetAsync(h->st.cs_power_hist, 0, sizeof(double) * (size_t)h->scn.T * h->scn.E * h->scn.C, h->stream));
        HIPCHK(h, hipMemsetAsync(h->st.cs_cur_hist, 0, sizeof(double) * (size_t)h->scn.T * h->scn.E * h->scn.C, h->stream));
    }
    h->scn_off = off;
    h->current_step = 0;
    return EV2G_OK;
}
int ev2g_get_stats_reset(ev2g_handle *h, double *stats, double *obs, int64_t scenario_offset) { return stats_reset_impl(h, stats, obs, nullptr, scenario_offset); }
int ev2g_get_stats_reset_f32(ev2g_handle *h, double *stats, float *obs32, int64_t scenario_offset) { return stats_reset_impl(h, stats, nullptr, obs32, scenario_offset); }

// ---- multi-GPU statistics exchange (RCCL) --------------------------------------------------------------------------------
int ev2g_comm_get_unique_id(void *id) {
    RcclApi *api = rccl_api();
    if (!api->lib) return fail(nullptr, EV2G_ERR_STATE, "ev2g_comm_get_unique_id: " + api->err);
    if (!id) return fail(nullptr, EV2G_ERR_ARG, "ev2g_comm_get_unique_id: null output");
    static_assert(sizeof(ncclUniqueId) == EV2G_COMM_ID_BYTES, "EV2G_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
    ncclUniqueId u;
    const ncclResult_t r = api->GetUniqueId(&u);
    if (r != ncclSuccess) return fail(nullptr, EV2G_ERR_HIP, std::string("ncclGetUniqueId: ") + api->GetErrorString(r));
    std::memcpy(id, &u, sizeof u);
    return EV2G_OK;
}

int ev2g_comm_init(ev2g_handle *h, const void *id, int rank, int world_size) {
    if (!h) return fail(h, EV2G_ERR_ARG, "ev2g_comm_init: null handle");
    if (!id || world_size < 1 || rank < 0 || rank >= world_size) return fail(h, EV2G_ERR_ARG, "ev2g_comm_init: bad arguments");
    RcclApi *api = rccl_api();
    if (!api->lib) return fail(h, EV2G_ERR_STATE, "ev2g_comm_init: " + api->err);
    ev2g_comm_destroy(h);
    (void)hipSetDevice(h->device);
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    const ncclResult_t r = api->CommInitRank(&h->comm.comm, world_size, u, rank);
    if (r != ncclSuccess) { h->comm.comm = nullptr; return fail(h, EV2G_ERR_HIP, std::string("ncclCommInitRank: ") + api->GetErrorString(r)); }
    h->comm.rank = rank; h->comm.world = world_size;
    return EV2G_OK;
}

void ev2g_comm_destroy(ev2g_handle *h) {
    if (!h || !h->comm.comm) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    rccl_api()->CommDestroy(h->comm.comm);
    if (h->comm.d_send) (void)hipFree(h->comm.d_send);
    h->comm = CommState{};
}

int ev2g_comm_world_size(const ev2g_handle *h) { return (h && h->comm.comm) ? h->comm.world : 0; }
long long ev2g_comm_gathers(const ev2g_handle *h) { return h ? h->comm.gathers : 0; }

int ev2g_gather_stats(ev2g_handle *h, double *stats_all) {
    if (!h || !h->loaded) return fail(h, EV2G_ERR_STATE, "ev2g_gather_stats: no scenarios loaded");
    if (!h->comm.comm) return fail(h, EV2G_ERR_STATE, "ev2g_gather_stats: no communicator (call ev2g_comm_init on every rank first)");
    if (!stats_all) return fail(h, EV2G_ERR_ARG, "ev2g_gather_stats: null output");
    (void)hipSetDevice(h->device);
    if (h->comm.send_envs != h->E) {
        if (h->comm.d_send) { (void)hipStreamSynchronize(h->stream); (void)hipFree(h->comm.d_send); h->comm.d_send = nullptr; }
        HIPCHK(h, hipMalloc((void **)&h->comm.d_send, sizeof(double) * (size_t)h->E * EV2G_N_STATS));
        h->comm.send_envs = h->E;
    }
    RcclApi *api = rccl_api();
    if (h->comm.checked_envs != h->E) {
        // ncclAllGather needs the same count on every rank: verify it once per communicator / loaded batch, with a collective whose
        // own count cannot differ (one int per rank), instead of gathering mismatched blocks into the wrong rows.  Every rank gets here
        // on its first gather after ev2g_comm_init (ranks that reload a batch of another size must do so together).
        int *d_n = nullptr;
        HIPCHK(h, hipMalloc((void **)&d_n, sizeof(int) * (size_t)(h->comm.world + 1)));
        HIPCHK(h, hipMemcpyAsync(d_n, &h->E, sizeof(int), hipMemcpyHostToDevice, h->stream));
        const ncclResult_t rn = api->AllGather(d_n, d_n + 1, 1, ncclInt32, h->comm.comm, h->stream);
        std::vector<int> n(h->comm.world + 1, 0);
        hipError_t he = hipMemcpyAsync(n.data(), d_n, sizeof(int) * n.size(), hipMemcpyDeviceToHost, h->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(h->stream);
        (void)hipFree(d_n);
        if (rn != ncclSuccess) return fail(h, EV2G_ERR_HIP, std::string("ncclAllGather (env counts): ") + api->GetErrorString(rn));
        HIPCHK(h, he);
        for (int r = 0; r < h->comm.world; r++)
            if (n[r + 1] != h->E)
                return fail(h, EV2G_ERR_STATE, "ev2g_gather_stats: rank " + std::to_string(r) + " steps " + std::to_string(n[r + 1]) + " envs, this rank " +
                                                   std::to_string(h->E) + " (the gather needs equal shards; pad the batch or use torch's uneven gather)");
        h->comm.checked_envs = h->E;
    }
    const int rc = ev2g_get_stats(h, h->comm.d_send);
    if (rc) return rc;
    const ncclResult_t r = api->AllGather(h->comm.d_send, stats_all, (size_t)h->E * EV2G_N_STATS, ncclDouble, h->comm.comm, h->stream);
    if (r != ncclSuccess) return fail(h, EV2G_ERR_HIP, std::string("ncclAllGather: ") + api->GetErrorString(r));
    h->comm.gathers++;
    return EV2G_OK;
}

int ev2g_peek(ev2g_handle *h, int env, ev2g_env_view *v) {
    if (!h || !h->loaded) return fail(h, EV2G_ERR_STATE, "ev2g_peek: no scenarios loaded");
    if (!v || env < 0 || env >= h->E) return fail(h, EV2G_ERR_ARG, "ev2g_peek: bad arguments");
    if (h->refilled) return fail(h, EV2G_ERR_STATE, "ev2g_peek: the scenario pool was refilled on the device (ev2g_pool_refill): the host holds no copy of its scenarios");
    (void)hipSetDevice(h->device);
    const int P = h->P, C = h->C, R = h->R, T = h->T, E = h->E;
    const DevState &st = h->st;
    std::vector<double> cap(P), tot(P), prev(P);
    std::vector<int2> win(P), sc(P);
    const size_t off = (size_t)env * P;
#define D2H(dst, src, n, type) HIPCHK(h, hipMemcpyAsync((dst), (src), sizeof(type) * (size_t)(n), hipMemcpyDeviceToHost, h->stream))
    // One env's pieces come down into ONE page-locked staging block (kept with the handle): copies into pageable vectors are staged by the runtime one
    // after the other (~15 us each; the facade peeks after every step: round 6, 0.19 -> 0.05 ms per call), these are queued together and waited for once.
    const bool log_cs = st.cs_profits != nullptr;
    const size_t n_lines = (size_t)P * sizeof(PortLine) / 8, n_hist = (size_t)T * (2 + R);
    const size_t need = 8 * (n_lines + 2 * (size_t)P + (size_t)R + (log_cs ? 5 * (size_t)C : 0) + n_hist);
    if (h->peek_stage_bytes < need) {
        if (h->peek_stage) { (void)hipHostFree(h->peek_stage); h->peek_stage = nullptr; h->peek_stage_bytes = 0; }
        HIPCHK(h, hipHostMalloc(&h->peek_stage, need, hipHostMallocDefault));
        h->peek_stage_bytes = need;
    }
    double *stg = (double *)h->peek_stage;
    const PortLine *lines = (const PortLine *)stg;
    double *pe = stg + n_lines, *pc = pe + P, *trp = pc + P, *csv = trp + R, *hist_rows = csv + (log_cs ? 5 * (size_t)C : 0);
    D2H((void *)lines, st.line + off, P, PortLine);
    D2H(pe, st.port_energy + off, P, double);
    D2H(pc, st.port_current + off, P, double);
    D2H(trp, st.tr_power_now + (size_t)env * R, R, double);
    if (log_cs) {
        D2H(csv + 0 * C, st.cs_power_now + (size_t)env * C, C, double);
        D2H(csv + 1 * C, st.cs_cur_now + (size_t)env * C, C, double);
        D2H(csv + 2 * C, st.cs_profits + (size_t)env * C, C, double);
        D2H(csv + 3 * C, st.cs_e_ch + (size_t)env * C, C, double);
        D2H(csv + 4 * C, st.cs_e_dis + (size_t)env * C, C, double);
    }
    std::vector<double> usage(T), pot(T), over((size_t)T * R);
    D2H(hist_rows, st.hist + (size_t)env * T * (2 + R), n_hist, double);   // this env's rows of the history array [E, T, 2 + R]: contiguous
#undef D2H
    HIPCHK(h, hipStreamSynchronize(h->stream));
    // rows the running episode has not written yet read as zeros (the reference's arrays are zero-initialised at reset); the slab itself may still
    // hold the previous episode's values there -- after an in-kernel reset of a fused run, and after ev2g_get_stats_reset, which re-arms an env
    // without clearing its history rows (the statistics kernel ignores them the same way).  usage / overload of step t exist once step t has run,
    // the charge-power potential of step t once step t - 1 has.
    const int cur = h->current_step;
    for (int t = 0; t < T; t++) {
        usage[t] = (t < cur) ? hist_rows[(size_t)t * (2 + R)] : 0.0;
        pot[t] = (t <= cur) ? hist_rows[(size_t)t * (2 + R) + 1] : 0.0;
        for (int r = 0; r < R; r++) over[(size_t)t * R + r] = (t < cur) ? hist_rows[(size_t)t * (2 + R) + 2 + r] : 0.0;
    }
    for (int q = 0; q < P; q++) {
        const PortLine &l = lines[q];
        cap[q] = l.cap; tot[q] = l.tot; prev[q] = l.prev; win[q] = make_int2(l.ta, l.td); sc[q] = make_int2(l.ss, ev2g_line_cycles(l.cyc_lut));
    }
    const int t = h->current_step;
    v->current_step = t;
    v->n_ports = P; v->n_chargers = C; v->n_transformers = R; v->n_steps = T;
    const long long scn = ((long long)env + h->scn_off) % h->M;   // the scenario this env is running
    const long long s0 = h->env_sess_start[scn], s1 = h->env_sess_start[scn + 1];
    std::vector<int> dev_to_local;  // device idx -> env-local host idx
    if (v->port_session) {
        // inverse map restricted to this env
        int dmin = 0x7fffffff;
        for (long long s = s0; s < s1; s++) dmin = std::min(dmin, h->host_to_dev[s]);
        dev_to_local.assign((size_t)(s1 - s0), -1);
        for (long long s = s0; s < s1; s++) dev_to_local[h->host_to_dev[s] - dmin] = (int)(s - s0);
        for (int q = 0; q < P; q++) {
            const bool occ = win[q].x <= t && t <= win[q].y && sc[q].x >= 0;
            v->port_session[h->slot_port[q]] = occ ? dev_to_local[sc[q].x - dmin] : -1;
        }
    }
    const double nan = std::nan("");
    for (int q = 0; q < P; q++) {
        const int p = h->slot_port[q];
        // after step t-1 the port holds an EV iff its window covers the current step counter
        const bool occ = win[q].x <= t && t <= win[q].y;
        if (v->port_capacity) v->port_capacity[p] = occ ? cap[q] : nan;
        if (v->port_energy) v->port_energy[p] = occ ? pe[q] : nan;
        if (v->port_current) v->port_current[p] = occ ? pc[q] : nan;
        if (v->port_total_energy) v->port_total_energy[p] = occ ? tot[q] : nan;
        if (v->port_prev_power) v->port_prev_power[p] = occ ? prev[q] : nan;
        if (v->port_required_energy) v->port_required_energy[p] = nan;  // filled by the Python facade (B - cap0 - tot_e)
        if (v->port_cycles) v->port_cycles[p] = occ ? sc[q].y : -1;
    }
    for (int c = 0; c < C; c++) {
        if (v->cs_power) v->cs_power[c] = log_cs ? csv[0 * C + c] : nan;
        if (v->cs_amps) v->cs_amps[c] = log_cs ? csv[1 * C + c] : nan;
        if (v->cs_profits) v->cs_profits[c] = log_cs ? csv[2 * C + c] : nan;
        if (v->cs_energy_charged) v->cs_energy_charged[c] = log_cs ? csv[3 * C + c] : nan;
        if (v->cs_energy_discharged) v->cs_energy_discharged[c] = log_cs ? csv[4 * C + c] : nan;
    }
    // entries the running episode has not written yet read as zeros (after an in-kernel reset of a fused run the slab still holds the
    // previous episode's values there); charge_power_potential is written one step ahead (utils.py:760-791)
    for (int k = t; k < T; k++) { usage[k] = 0.0; for (int r = 0; r < R; r++) over[(size_t)k * R + r] = 0.0; }
    for (int k = t + 1; k < T; k++) pot[k] = 0.0;
    if (v->tr_power) std::copy(trp, trp + R, v->tr_power);
    if (v->tr_overload)
        for (int r = 0; r < R; r++)
            for (int k = 0; k < T; k++) v->tr_overload[(size_t)r * T + k] = over[(size_t)k * R + r];
    if (v->power_usage) std::copy(usage.begin(), usage.end(), v->power_usage);
    if (v->power_potential) std::copy(pot.begin(), pot.end(), v->power_potential);
    std::vector<double> fcap;
    if (v->session_final_cap && s1 > s0) {
        int dmin = 0x7fffffff;
        for (long long s = s0; s < s1; s++) dmin = std::min(dmin, h->host_to_dev[s]);
        fcap.resize((size_t)(s1 - s0));
        HIPCHK(h, hipMemcpy(fcap.data(), st.sess_final_cap + dmin, sizeof(double) * (size_t)(s1 - s0), hipMemcpyDeviceToHost));
        for (long long s = s0; s < s1; s++) v->session_final_cap[s - s0] = fcap[h->host_to_dev[s] - dmin];
    }
    for (long long s = s0; s < s1; s++) {
        if (v->session_port) v->session_port[s - s0] = h->sess_port[s];
        if (v->session_afap) v->session_afap[s - s0] = h->sess_afap[s];
    }
    return EV2G_OK;
}

#ifdef EV2G_PHASE_TIMING
int ev2g_debug_phase_ticks(ev2g_handle *h, unsigned long long *out18) {
    std::vector<unsigned long long> v((size_t)h->scn.n_groups * 18);
    HIPCHK(h, hipMemcpy(v.data(), h->st.dbg, v.size() * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < 18; i++) { out18[i] = 0; for (int b = 0; b < h->scn.n_groups; b++) out18[i] += v[(size_t)b * 18 + i]; }
    HIPCHK(h, hipMemset(h->st.dbg, 0, v.size() * 8));
    return 0;
}
#endif

void *ev2g_malloc(ev2g_handle *h, size_t bytes) {
    if (!h) return nullptr;
    (void)hipSetDevice(h->device);
    void *p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(bytes, 1)) != hipSuccess) {
        h->err = "ev2g_malloc: hipMalloc failed";
        return nullptr;
    }
    h->user_allocs.push_back(p);
    return p;
}
void ev2g_free(ev2g_handle *h, void *p) {
    if (!h || !p) return;
    auto it = std::find(h->user_allocs.begin(), h->user_allocs.end(), p);
    if (it != h->user_allocs.end()) h->user_allocs.erase(it);
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(p);
}
void *ev2g_host_malloc(ev2g_handle *h, size_t bytes) {
    if (!h) return nullptr;
    (void)hipSetDevice(h->device);
    void *p = nullptr;
    if (hipHostMalloc(&p, std::max<size_t>(bytes, 1), hipHostMallocDefault) != hipSuccess) {
        h->err = "ev2g_host_malloc: hipHostMalloc failed";
        return nullptr;
    }
    h->user_host_allocs.push_back(p);
    return p;
}
void ev2g_host_free(ev2g_handle *h, void *p) {
    if (!h || !p) return;
    auto it = std::find(h->user_host_allocs.begin(), h->user_host_allocs.end(), p);
    if (it == h->user_host_allocs.end()) return;   // (not ours, or freed already)
    h->user_host_allocs.erase(it);
    (void)hipStreamSynchronize(h->stream);
    (void)hipHostFree(p);
}
int ev2g_memcpy_h2d(ev2g_handle *h, void *dst, const void *src, size_t bytes) {
    if (!h) return EV2G_ERR_ARG;
    (void)hipSetDevice(h->device);
    HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return EV2G_OK;
}
int ev2g_memcpy_d2h(ev2g_handle *h, void *dst, const void *src, size_t bytes) {
    if (!h) return EV2G_ERR_ARG;
    (void)hipSetDevice(h->device);
    HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return EV2G_OK;
}
int ev2g_synchronize(ev2g_handle *h) {
    if (!h) return EV2G_ERR_ARG;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return EV2G_OK;
}
int ev2g_fill_uniform(ev2g_handle *h, double *dst, int64_t n, uint64_t seed, double lo, double hi) {
    if (!h || !dst || n < 0) return fail(h, EV2G_ERR_ARG, "ev2g_fill_uniform: bad arguments");
    (void)hipSetDevice(h->device);
    const int nb = (int)std::min<int64_t>((n + 255) / 256, 2048);
    if (n) hipLaunchKernelGGL(ev2g_fill_uniform_kernel, dim3(std::max(nb, 1)), dim3(256), 0, h->stream, dst, (long long)n, seed, lo, hi);
    HIPCHK(h, hipGetLastError());
    return EV2G_OK;
}
void ev2g_host_uniform(double *dst, int64_t n, uint64_t seed, double lo, double hi) {
    for (int64_t i = 0; i < n; i++) dst[i] = lo + (hi - lo) * ev2g_u01(seed, (uint64_t)i);
}

// ---- scenario generator (host only) ----
int ev2g_gen_default_config(int kind, ev2g_gen_config *cfg) { return ev2g_gen_default_config_impl(kind, cfg); }
int ev2g_pool_refill(ev2g_handle *h, const ev2g_gen_config *cfg, uint64_t seed, int64_t first_index, int32_t first_slot, int32_t n) {
    try { return ev2g_pool_refill_impl(h, cfg, seed, first_index, first_slot, n); }
    catch (const std::exception &e) { return fail(h, EV2G_ERR_ARG, std::string("ev2g_pool_refill: ") + e.what()); }
}
long long ev2g_pool_refill_overflows(ev2g_handle *h) { return ev2g_pool_refill_overflows_impl(h); }
int ev2g_pool_session_capacity(const ev2g_handle *h) { return h ? h->sess_cap : 0; }
int ev2g_generate(const ev2g_gen_config *cfg, int32_t n_scenarios, uint64_t seed, int32_t n_threads, ev2g_gen_result **out) {
    return ev2g_generate_impl(cfg, n_scenarios, seed, n_threads, out);
}
const ev2g_scenario_batch *ev2g_gen_batch(const ev2g_gen_result *r) { return r ? &r->b : nullptr; }
void ev2g_gen_free(ev2g_gen_result *r) { delete r; }
int ev2g_gen_table(int which, int kind, double *out, int n_max) { return ev2g_gen_table_impl(which, kind, out, n_max); }

}  // extern "C"
