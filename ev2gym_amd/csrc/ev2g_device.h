// ev2g_device.h -- CDNA4 (gfx950) device code of the vectorised EV2Gym step engine.
//
// One launch advances E independent envs by one (or K) timesteps of EV2Gym.step()
// (reference ev2gym/models/ev2gym_env.py:333-447).  Work decomposition:
//   * a 256-thread workgroup owns G = max(1, 256 / P) whole envs; lane <-> (env, port slot);
//   * port slots are stored TRANSFORMER-MAJOR (the order both state functions emit, state.py:37-57,
//     :128-151), so every transformer is a contiguous lane run and the observation is written in order;
//   * phase 1  per-port: action -> amps -> two-stage charge / discharge -> ceil2, departures, arrivals,
//              per-port observation columns, action mask (EV_Charger.step ev_charger.py:114-233,
//              EV.step ev.py:138-186, spawn ev2gym_env.py:399-417);
//   * phase 2  LDS-staged segmented reduction: per-port results are staged in LDS, each (env, transformer)
//              segment is summed by a sub-wave lane group with __shfl_xor butterflies (fixed tree =>
//              bit-reproducible), then R partials per env are folded (Transformer.step transformer.py:269-274);
//   * phase 3  per-env: overload, reward, histories, observation head + forecast / limit windows
//              (rl_agent/reward.py, rl_agent/state.py, transformer.py:142-188).
// No cross-workgroup communication exists (envs are independent), so the K-step variant just loops in
// the block.  Everything is float64; compiled with -ffp-contract=off and the reference's operation order,
// because EV.my_ceil (ev.py:188-189) amplifies 1-ulp differences to 0.01 kWh.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EV2G_BLOCK 256
#define EV2G_NQ 8  // staged quantities per port

// One EV session, 128 bytes = one cache line: what the per-step battery maths and an arrival need (ev.py:68-113), laid out by CONSUMER so that
// each of them fetches a contiguous run of 16-byte chunks: a charging step reads chunks 0..4 (80 bytes), a discharging step chunks 3..6
// (64 bytes), an arrival chunks 2 and 7.  `rB` / `rv` are the correctly rounded reciprocals of `B` / `v` (computed once per session, by the
// loader or the device generator, with an IEEE division): the battery maths divides by B and v through them (ev2g_fdiv2, below) instead of
// through ~11-instruction hardware division sequences.
struct __attribute__((aligned(128))) SessRec {
    double pacmax, ts;        // chunk 0  (charge)
    double tsm, eta_ch;       // chunk 1  (charge)
    double gate_ch;           // chunk 2  (charge)  min_ac_charge_power*1000/(voltage*sqrt(charger phases))   (ev.py:151)
    double B;                 //          (charge, arrival)
    double rB;                // chunk 3  (charge)  RN(1 / B)
    double v;                 //          (charge, discharge)  voltage*sqrt(min(charger phases, ev_phases))   (ev.py:169,279,365)
    double rv;                // chunk 4  (charge, discharge)  RN(1 / v)
    double gate_dis;          //          (discharge)  min_discharge_power*1000/(voltage*sqrt(charger phases))   (ev.py:153)
    double minB, emerg;       // chunk 5  (discharge)
    double pdismax, eta_dis;  // chunk 6  (discharge)
    double cap0;              // chunk 7  (arrival)  battery_capacity_at_arrival
    double potc;              //          (arrival)  this EV's term of calculate_charge_power_potential before the charger clamp:
                              //          v * min(pacmax*1000/v, charger max current) / 1000   (utils.py:773-777), evaluated once per session
};
// What a departure reads (and the rewards that look at every connected EV's desired capacity), 16 bytes per session next to the records
struct SessTail {
    double des;          // desired_capacity
    int nt_arr, nt_dep;  // window of the next session on the same port (EV2G_INT_MAX = none)
};

// Round 5: the battery maths' operands that are the SAME for every session of one car model on one kind of charger (ev.py:68-113: the model's
// powers, battery size, gates; the charger's voltage and phases) live in a small dictionary instead of in every session's record: a few
// dozen 128-byte entries that stay in the vector L1, so the fetch in the middle of the battery-maths phase is an L1 hit instead of an L2
// round trip to the session's own line (the one lever that reached 0.60 of the roofline in round 4's ablation).  What really differs per
// session -- transition_soc and, without an efficiency table, the two efficiencies (utils.py:293-296,309-310) -- is SessDyn: it travels with the
// arrival's other operands into the port's LDS state (ev2g_step_wave.h) and, across launches, into the port's PortDyn entry.
// Laid out by consumer: a charging step reads chunks 0..3 (one 64-byte sector), a discharging step chunks 4..6.
struct __attribute__((aligned(128))) ClsRec {
    double pacmax, tsm;       // chunk 0  (charge)
    double gate_ch, B;        // chunk 1  (charge)
    double rB, v;             // chunk 2  (charge)
    double rv, pad0;          // chunk 3  (charge)
    double v_d, rv_d;         // chunk 4  (discharge: copies of v, rv)
    double gate_dis, minB;    // chunk 5  (discharge)
    double emerg, pdismax;    // chunk 6  (discharge)
    double pad1[2];
};
#define EV2G_CLS_CAP 4096     // dictionary entries (12 bits of the port's LDS word); a batch with more distinct tuples keeps one ClsRec per SESSION instead
struct __attribute__((aligned(32))) SessDyn {
    double ts, eta_ch;        // EV.transition_soc, EV.charge_efficiency (a number: sessions without an efficiency table)
    double eta_dis;           // EV.discharge_efficiency
    int lut, cls;             // efficiency-table id (-1: none), dictionary entry (the session's own index when the batch has no dictionary)
};
typedef SessDyn PortDyn;      // the same four words for the EV attached to a port, written at its arrival: what a later launch's prologue reads
__host__ __device__ inline ClsRec ev2g_cls_of(const SessRec &r) {
    ClsRec c;
    c.pacmax = r.pacmax; c.tsm = r.tsm; c.gate_ch = r.gate_ch; c.B = r.B; c.rB = r.rB; c.v = r.v; c.rv = r.rv; c.pad0 = 0.0;
    c.v_d = r.v; c.rv_d = r.rv; c.gate_dis = r.gate_dis; c.minB = r.minB; c.emerg = r.emerg; c.pdismax = r.pdismax; c.pad1[0] = 0.0; c.pad1[1] = 0.0;
    return c;
}

struct DevScn {  // read-only scenario + layout, device pointers
    int E;        // envs stepped concurrently (state arrays are [E, ...])
    int M;        // scenarios resident in the pool (scenario arrays are [M, ...]); env e runs scenario (e + off) mod M
    int T, C, npc, P, R, D, ND, dt;
    int reward_kind, state_kind, flags, cost_kind;
    int n_lut;    // number of efficiency tables
    int G;        // envs per workgroup
    int gs;       // lanes per reduction group (power of two, 4..64)
    int n_groups; // workgroups = ceil(E / G)
    double sixty_over_dt;  // 60 / timescale   (ev.py:296)
    double dt_over_60;     // timescale / 60   (ev.py:355)
    // slot tables [P] (slot = transformer-major port order)
    const int *slot_port, *slot_cs, *slot_obs, *slot_tr;
    // chargers whose port counts differ (topology file; generic kernel only): ports [C] and first port [C+1] of every charger, and per
    // slot the action-mask entry the reference sets for it, i*cs.n_ports + j (ev2gym_env.py:452-457) -- the port itself when counts are equal
    const int *cs_np, *cs_pbase, *slot_mask;
    const int *cs_slot0;   // [C] slot of each charger's port 0 (a charger's ports are adjacent slots, in port order)
    int het;
    // chargers [C]
    const double *cs_imin, *cs_imax, *cs_dmin, *cs_dmax_abs, *cs_volt, *cs_maxp, *cs_minp;
    const double *cs_pack;   // [C][6]: imax, |dmax|, imin, dmin, max power, min power (ev2g_step_wave prologue: 3 loads instead of 6)
    const double *cs_vk;  // [C,4] voltage*sqrt(k), k = 0..3
    const int *cs_ph;
    // transformers: slot segments [R+1], obs column of the 40-wide window block [R]
    const int *tr_seg, *tr_obs;
    // env series [E,T]
    const double *price_ch, *price_dis, *setpoint;
    // transformer series [E,R,T], [E,R]
    const double *tr_maxp, *tr_minp, *tr_infl, *tr_solar, *tr_lf, *tr_pvf, *tr_peak;
    const double *tr_base;  // [E,R,T] inflexible_load + solar_power (Transformer.reset, transformer.py:262-263)
    const double *tr_dr;  // [E,R,ND,3]
    const int *tr_ndr, *tr_ahead;
    // sessions in device order (env, slot, arrival)
    const int *ss_tarr, *ss_tdep, *ss_ntarr, *ss_ntdep, *ss_phases, *ss_lut;
    const double *ss_cap0, *ss_B, *ss_des, *ss_minB, *ss_emerg, *ss_pacmax, *ss_pacmin, *ss_pdismax, *ss_pdismin,
        *ss_ts, *ss_tsm, *ss_etach, *ss_etadis;
    const double *lut;  // [NL,101]
    // per port [E*P]: first session (or -1) and its window
    const int *port_first;
    const int *port_end;   // one past the port's last session
    const int *ss_slot;    // [S] port slot of every session; const int *scn_sess: [M+1] session range of every scenario (statistics kernel)
    const int *scn_sess;
    const int *scn_sess_end;   // [M] one past the scenario's last session (== scn_sess[m+1] unless the pool is refillable: fixed-size blocks)
    const int2 *port_first_win;
    const SessRec *rec;  // [S] AoS twin of the ss_* arrays (v2 kernels)
    const SessTail *tail;  // [S] departure-side fields of the same sessions
    const SessDyn *sess_dyn;   // [S] per-session battery-maths operands + dictionary entry (fast path, round 5)
    const ClsRec *cls_rec;     // dictionary [EV2G_CLS_CAP] (n_cls used), or one entry per session [S] when `dict` is 0
    int dict, n_cls;
    const double *win_tab;  // [E,R,T+1,40] precomputed (loads-pv)[20] | power_limits[20] per observation step, or nullptr
    const double *head_tab; // fast path: [M,T+1,head_nh] columns 2.. of the observation of every step counter (prices | window), or nullptr
    int head_nh;
};

// Per-port dynamic state: ONE 64-byte line per (env, port slot) -- one memory sector.  A launch that runs a single step (the RL loop with a
// policy between steps) starts with cold caches and fetches state only for the ports that hold an EV or receive one in this step (a fifth of
// them; which ones is known before the step from the scenario's occupancy masks, ev2g_build_occ_mask_kernel): one line each, instead of
// eight-byte pieces of nine arrays whose sectors were nearly all touched (round 3: 30.7 MB of traffic per launch against 16 MB algorithmic).
struct __attribute__((aligned(64))) PortLine {
    int ta, td;          // {t_arr, t_dep} of the attached-or-next session (EV2G_INT_MAX = none)
    int ss;              // its index in the session arrays (-1: the port's list is exhausted)
    int cyc_lut;         // EV.charging_cycles (low 16 bits) | (efficiency-table id + 1) << 16 of the attached EV; T and the table count are < 65536 (checked at load)
    double cap, tot;     // EV.current_capacity, EV.total_energy_exchanged
    double prev, abse;   // EV.previous_power, EV.abs_total_energy_exchanged
    double bcap, potc;   // battery_capacity and charge-power-potential term of the attached EV
};
__host__ __device__ __forceinline__ int ev2g_line_cycles(int cyc_lut) { return cyc_lut & 0xffff; }
__host__ __device__ __forceinline__ int ev2g_line_lut(int cyc_lut) { return (int)((unsigned)cyc_lut >> 16) - 1; }
__host__ __device__ __forceinline__ int ev2g_line_pack(int cycles, int lut) { return (cycles & 0xffff) | (int)((unsigned)(lut + 1) << 16); }

// The other [E*P]- / [E*C]-shaped state arrays live in ONE allocation of equal slices (slice = max(E*P, E*C) * 8 bytes), in this order;
// likewise {usage, potential, overload} histories and the two per-session result arrays.  Every kernel keeps using the individual pointers
// below; the fast-path kernel derives them from the slab base with scalar adds (ev2g_step_wave.h).
// word index of row (env e, step t) of the history array [E, T, 2 + R]: + 0 usage, + 1 charge-power potential, + 2 + r overload of transformer r
#define EV2G_HIST(e, t, T, R) (((long long)(e) * (T) + (t)) * (2 + (R)))
enum { EV2G_PS_PENERGY = 0, EV2G_PS_PCURRENT, EV2G_PS_SATSUM, EV2G_PS_SERVED, EV2G_PS_N };

struct DevState {  // mutable engine state, device pointers
    char *slab_port; unsigned long long slab_port_slice;   // EV2G_PS_* slices, bytes per slice
    double *slab_hist;   // == hist
    double *slab_sess;   // sess_final_cap | sess_abs_e         ([S] each)
    PortLine *line;                    // [E*P] per-port dynamic state (above)
    PortDyn *port_dyn;                 // [E*P] SessDyn of the attached EV (fast path; written at its arrival, read by a launch's prologue)
    double *cs_sat_sum;                // [E*C] EV_Charger.total_user_satisfaction
    int *cs_served;                    // [E*C] EV_Charger.total_evs_served
    double *cs_profits, *cs_e_ch, *cs_e_dis;  // [E*C] (EV2G_FLAG_LOG_CS_HISTORY) else nullptr
    double *cs_power_hist, *cs_cur_hist;      // [T,E,C] (flag) else nullptr
    double *cs_power_now, *cs_cur_now;        // [E*C] last step (flag)
    double *env_acc;                   // [E,8] total_reward, profits, e_charged, e_discharged, emerg_violations
    int *env_fault;                    // [E]
    double *hist;                      // [E, T, 2 + R] per (env, step): env.current_power_usage[t], charge_power_potential[t], tr_overload[0..R)[t] -- env-major and
                                       // interleaved since round 4: the three values a step writes share a sector, and the statistics kernel reads an env's
                                       // rows contiguously (time-major [T,E] arrays cost it one sector per value: 88 MB of traffic for 11 MB of data at cfg2)
    double *tr_power_now;              // [E,R]  Transformer.current_power of the last step
    double *sess_final_cap;            // [S] capacity at departure
    double *soc_log;                   // [E,T,P] (EV2G_FLAG_LOG_SOC; env-major blocks, time-major inside: the step kernel's writes of one env-step fall into a few
                                       // neighbouring sectors.  A PORT-major log [E*P,T] -- a session's entries contiguous for the statistics kernel -- was measured
                                       // twice: round 1 (statistics -20 %, step kernel +13 %) and round 4 (statistics 48 -> 44 us, cfg2 step kernel 3.49 -> 4.15
                                       // us/step: every occupied lane then writes its own sector)) capacity before each EV.step, negated when the step was inactive
    double *sess_abs_e;                // [S]    (flag) the same, frozen at departure
    double *port_energy, *port_current;  // [E*P] EV.current_energy / actual_current of the last step
    unsigned long long *dbg;             // [n_groups*8] phase timing (EV2G_PHASE_TIMING builds only), else nullptr
};

struct StepIO {
    const double *actions; long long a_stride;
    double *obs;           long long o_stride;
    double *reward;        long long r_stride;
    uint8_t *done;         long long d_stride;
    uint8_t *mask;         long long m_stride;
    // scenario pool window of this launch: env e runs scenario (e + scn_off) mod M; an in-launch auto-reset advances the
    // offset by scn_stride (0: re-arm the same scenarios)
    int scn_off, scn_stride;
    int step0;   // index of this launch's first step inside the caller's ev2g_step_n run (offsets the extras' step strides)
    int log_soc; // EV2G_FLAG_LOG_SOC (the fast path's prologue reads it from here: no parameter-block fetch on its first round trip)
    const float *act32;   // StepExtras::act32 when `actions` is null (same reason)
    float *obs32;         // StepExtras::obs32 when its step stride is 0 (the full float32 kernels of the fast path write it instead of `obs`)
};

// optional extra step outputs / inputs (ev2g_set_step_extras), device-resident next to the kernel parameter block: they are
// sticky and rarely used, so they do not occupy kernel-argument SGPRs
struct StepExtras {
    double *cost;          long long c_stride;     // cost_function value [E]
    float *obs32;          long long o32_stride;   // float32 observation copy [E,D]
    const float *act32;                            // float32 actions [E,P], used when StepIO::actions is null; step stride a_stride
};

// scenario of env e under pool offset `off` (0 <= off < M, e < E <= M)
__host__ __device__ __forceinline__ int ev2g_scn(int e, int off, int M) { const int s = e + off; return s >= M ? s - M : s; }
// action of (env-port element i) from whichever action array the caller supplied; widened to float64 on entry
__device__ __forceinline__ double ev2g_action(const StepIO &io, const float *act32, long long step_off, long long i) {
    return io.actions ? io.actions[step_off + i] : (double)act32[step_off + i];
}

#define EV2G_INT_MAX 0x7fffffff

__device__ __forceinline__ double ceil2(double a) { return ceil(a * 100.0) / 100.0; }       // ev.py:188-189
__device__ __forceinline__ double rnd5(double x) { return rint(x * 100000.0) / 100000.0; }  // ev_charger.py:157

// n / b for an INTEGRAL n with |n| <= bound and a constant b: q0 = n*RN(1/b), exact residual by fma, one fma
// correction.  Bit-identical to the IEEE division for every such n -- checked exhaustively on the host for the two
// uses below (tests/test_fma_division.py: all |n| <= 2e5 for b = 1e5, all |n| <= 2e7 for b = 100); 3 full-rate
// instructions instead of the ~14 (two of them quarter-rate) of a float64 division.  Outside the bound: the division.
__device__ __forceinline__ double div_int_by_const(double n, double b, double rb, double bound) {
    if (fabs(n) <= bound) {
        const double q0 = n * rb;
        const double r = fma(-q0, b, n);
        return fma(r, rb, q0);
    }
    return n / b;
}
__device__ __forceinline__ double ceil2_x(double a) { return div_int_by_const(ceil(a * 100.0), 100.0, 1.0 / 100.0, 2.0e7); }
__device__ __forceinline__ double rnd5_x(double x) { return div_int_by_const(rint(x * 100000.0), 100000.0, 1.0 / 100000.0, 2.0e5); }

// a / b through the correctly rounded reciprocal rb = RN(1/b) (Markstein): q0 = RN(a*rb); r = a - b*q0 (exact, one fma); q = RN(q0 + r*rb).
// THEOREM (Markstein 1990; Muller et al., Handbook of Floating-Point Arithmetic, "division with an FMA"): if rb is 1/b rounded to nearest and
// q0 is a FAITHFUL rounding of a/b (one of its two floating-point neighbours), then q is a/b correctly rounded -- bit for bit what the IEEE
// division returns (no overflow / underflow / subnormals: every operand here is a physical quantity of magnitude 1e-6 .. 1e6, or zero).
//   * ev2g_fdiv1 -- ONE correction step -- is used where b is a compile-time (or launch-time) CONSTANT whose reciprocal error
//     delta_b = |b*RN(1/b) - 1| is <= 2^-54: then |a*rb - a/b| <= ulp/2 for EVERY a, so RN(a*rb) is faithful and the theorem applies.
//     1000, 100, 60, 15, 30 qualify (0.375, 0.375, 0.25, 0.25, 0.25 x 2^-54; tests/test_fma_division.py recomputes them with exact rational
//     arithmetic); the step length dt is checked on the host at load (V2P::dt_fdiv) and falls back to the division otherwise.
//   * ev2g_fdiv2 -- TWO correction steps -- is used for the per-session divisors B and v, whose delta can reach 2^-53: the first step's result
//     is within ulp/2 + 2^-52 ulp of a/b, hence faithful, and the second step makes it exact by the theorem.  5 full-rate instructions instead of the
//     ~11 of the hardware sequence (v_div_scale x2, quarter-rate v_rcp_f64, 5 fma, v_div_fmas, v_div_fixup).
// tests/test_fma_division.py runs both forms against the IEEE division on 4e8 random and near-midpoint operand pairs (0 differences).
__host__ __device__ __forceinline__ double ev2g_fdiv1(double a, double b, double rb) {
    const double q0 = a * rb;
    return fma(fma(-q0, b, a), rb, q0);
}
__host__ __device__ __forceinline__ double ev2g_fdiv2(double a, double b, double rb) {
    const double q1 = ev2g_fdiv1(a, b, rb);
    return fma(fma(-q1, b, a), rb, q1);
}

// xor-butterfly partners inside 8-lane groups through DPP (VALU cross-lane moves, a few cycles) instead of
// ds_bpermute (an LDS crossbar round trip per step): quad_perm [1,0,3,2], quad_perm [2,3,0,1], and
// quad_perm [3,2,1,0] followed by row_half_mirror (i -> 3-i in the quad, then 7-i in the half row = i ^ 4).
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double xor1_f64(double v) { return dpp_mov_f64<0xB1>(v); }
__device__ __forceinline__ double xor2_f64(double v) { return dpp_mov_f64<0x4E>(v); }
__device__ __forceinline__ double xor4_f64(double v) { return dpp_mov_f64<0x141>(dpp_mov_f64<0x1B>(v)); }
__device__ __forceinline__ double readlane_f64(double v, int lane) {   // wave-uniform copy of lane `lane` (a compile-time constant)
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// Sum over the 64 lanes of a wavefront without touching the LDS crossbar: three DPP butterflies inside 8-lane groups, row_ror:8
// for the 16-lane rows, then the four row sums through v_readlane.  The result is wave-uniform; fixed order (bit-reproducible).
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += xor1_f64(v);
    v += xor2_f64(v);
    v += xor4_f64(v);
    v += dpp_mov_f64<0x128>(v);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ double wave_min_dpp(double v) {   // the same walk with fmin
    v = fmin(v, xor1_f64(v));
    v = fmin(v, xor2_f64(v));
    v = fmin(v, xor4_f64(v));
    v = fmin(v, dpp_mov_f64<0x128>(v));
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return fmin(fmin(r0, r1), fmin(r2, r3));
}

// dict.get(np.round(amps), 1): integer keys 0..100 exist, anything else -> 1   (ev.py:287-290, :375-379)
__device__ __forceinline__ double lut_get(const double *__restrict__ lut, int id, double key) {
    if (key >= 0.0 && key <= 100.0) return lut[id * 101 + (int)key];
    return 1.0;
}

struct EvOut {
    double energy, current;  // EV.current_energy (kWh this step), EV.actual_current (A)
    int emerg;               // crossed min_emergency_battery_capacity (ev.py:401-402)
    int active;              // EV.step went past the `amps == 0` early return (state was updated)
};

// EV.step + _charge/_discharge (ev.py:138-186, :240-355, :357-405) on one session's registers.
// `cap`, `tot_e`, `prev_power`, `cycles` are updated in place.
__device__ __forceinline__ EvOut ev_step(const DevScn &s, int ss, int cs, double amps, int ph_cs, double &cap,
                                         double &tot_e, double &prev_power, int &cycles) {
    EvOut o;
    o.energy = 0.0;
    o.current = 0.0;
    o.emerg = 0;
    o.active = 0;
    const double *vk = s.cs_vk + cs * 4;
    const double v_gate = vk[ph_cs];  // voltage*sqrt(charger phases): the min-power gates use the charger's phases
    if (amps > 0.0 && amps < s.ss_pacmin[ss] * 1000.0 / v_gate)
        amps = 0.0;
    else if (amps < 0.0 && amps > s.ss_pdismin[ss] * 1000.0 / v_gate)
        amps = 0.0;
    if (amps == 0.0) return o;  // no ceil, previous_power untouched (ev.py:158-163)
    o.active = 1;
    if (prev_power == 0.0 || (prev_power / amps) < 0.0) cycles += 1;
    const int evph = s.ss_phases[ss];
    const double v = vk[evph < ph_cs ? evph : ph_cs];
    const double B = s.ss_B[ss];
    const int lut = s.ss_lut[ss];
    if (amps > 0.0) {
        const double eta = (lut >= 0) ? lut_get(s.lut, lut, rint(amps)) / 100.0 : s.ss_etach[ss];
        const double pacmax = s.ss_pacmax[ss];
        double pilot_dsoc = eta * amps * v / 1000.0 / B / s.sixty_over_dt;
        const double max_dsoc = eta * pacmax / B / s.sixty_over_dt;
        if (pilot_dsoc > max_dsoc) pilot_dsoc = max_dsoc;
        const double soc = cap / B;
        const double ts = s.ss_ts[ss];
        double curr_soc;
        if (ts == 1.0) {
            curr_soc = pilot_dsoc + soc;
            if (curr_soc > 1.0) curr_soc = 1.0;
        } else {
            const double tsm = s.ss_tsm[ss];
            const double pts = ts + (pilot_dsoc - max_dsoc) / max_dsoc * (ts - 1.0);
            double new_soc;
            if (soc < pts) {
                if (1.0 <= (pts - soc) / pilot_dsoc)
                    new_soc = pilot_dsoc + soc;
                else
                    new_soc = 1.0 + exp(tsm * (pilot_dsoc + soc - pts) / (pts - 1.0)) * (pts - 1.0);
            } else {
                new_soc = 1.0 + exp(tsm * pilot_dsoc / (pts - 1.0)) * (soc - 1.0);
            }
            const double lim = (max_dsoc > pilot_dsoc) ? pilot_dsoc : max_dsoc;
            curr_soc = (new_soc - soc > lim) ? (lim + soc) : new_soc;
        }
        const double dsoc = curr_soc - soc;
        cap = curr_soc * B;
        o.energy = dsoc * B;
        o.current = o.energy / s.dt_over_60 * 1000.0 / v;
    } else {
        double given_power = amps * v / 1000.0;
        const double pdismax = s.ss_pdismax[ss];
        if (fabs(given_power) > fabs(pdismax)) given_power = pdismax;
        const double eta = (lut >= 0) ? lut_get(s.lut, lut, fabs(rint(amps))) / 100.0 : s.ss_etadis[ss];
        double given_energy = given_power * eta * (double)s.dt / 60.0;
        const double minB = s.ss_minB[ss];
        const double cap_before = cap;
        if (cap + given_energy < minB) {
            if (cap > minB) {
                o.energy = -(cap - minB);
                given_energy = o.energy;
            } else {
                o.energy = 0.0;
                given_energy = 0.0;
            }
            cap = minB;
        } else {
            o.energy = given_energy;
            cap += given_energy;
        }
        const double emerg = s.ss_emerg[ss];
        if (cap_before > emerg && cap < emerg) o.emerg = 1;
        o.current = given_energy * 60.0 / (double)s.dt * 1000.0 / v;
    }
    prev_power = o.energy;
    tot_e += o.energy;
    cap = ceil2(cap);
    return o;
}

// Transformer.get_power_limits(step, 20)[j]  (transformer.py:142-171)
__device__ __forceinline__ double power_limit_at(const DevScn &s, int er, int step, int j) {
    const double peak = s.tr_peak[er];
    double v = peak * 1.0;
    const int nd = s.tr_ndr[er];
    const int ahead = s.tr_ahead[er];
    for (int k = 0; k < nd; k++) {
        const double *ev = s.tr_dr + ((long long)er * s.ND + k) * 3;
        const int es = (int)ev[0], ee = (int)ev[1];
        if (step + ahead >= es && ee >= step) {
            int a, b;
            if (step > es) { a = 0; b = ee - step; }
            else { a = es - step; b = ee - step; }
            if (a < 0) a = -a;
            if (b < 0) b = -b;
            if (j >= a && j < b) v = peak - peak * ev[2] / 100.0;
        }
    }
    return v;
}

// (loads - pv)[j] of Transformer.get_load_pv_forecast(step, 20)  (transformer.py:173-188).  The reference
// overwrites forecast[step] with the actual value on every call, so after observing step s the window is
// [actual[s], forecast[s+1..]] tail-padded with forecast[T-1], which itself became actual[T-1] once s >= T-1.
__device__ __forceinline__ double load_minus_pv_at(const DevScn &s, long long erT, int step, int j) {
    const int T = s.T;
    const int k = step + j;
    double l, p;
    if (k < T) {
        if (j == 0) { l = s.tr_infl[erT + k]; p = s.tr_solar[erT + k]; }
        else        { l = s.tr_lf[erT + k];   p = s.tr_pvf[erT + k]; }
    } else if (step >= T - 1) { l = 1.0 * s.tr_infl[erT + T - 1]; p = 1.0 * s.tr_solar[erT + T - 1]; }
    else                      { l = 1.0 * s.tr_lf[erT + T - 1];   p = 1.0 * s.tr_pvf[erT + T - 1]; }
    return l - p;
}

// ---- reward plugins (rl_agent/reward.py) -----------------------------------------------------------------------------
// What a departing EV contributes to the step's "user" sum (staged per port, summed per env): the term of the reward in use,
// or -- for rewards without one -- of the transformer_overload_usrpenalty cost (cost.py:8-20) when that is enabled.
__device__ __forceinline__ double ev2g_departure_term(int reward_kind, int cost_kind, double score, double cap, double des) {
    if (reward_kind == 3) return 1000.0 * (1.0 - score);                           // SqTrError_TrPenalty_UserIncentives :30-31
    if (reward_kind == 8) return (des > cap) ? 100.0 * (des - cap) : 0.0;          // V2G_profitmax :130-136
    if (reward_kind >= 9) return (des > cap) ? 0.05 * ((des - cap) * (des - cap)) : 0.0;   // (pst_)V2G_profitmaxV2 :197-207
    if (reward_kind == 0 || reward_kind == 2 || cost_kind == 1) return 100.0 * exp(-10.0 * score);   // :41-42, :84-86, cost.py:17-18
    return 0.0;
}
// (pst_)V2G_profitmaxV2 also charge every EV that is connected AFTER the step's departures and arrivals and can no longer reach its
// desired capacity at full power (reward.py:173-195); sstep = env.current_step after its increment.  Staged with the departure terms.
__device__ __forceinline__ double ev2g_connected_term(double des, double cap, double pacmax, double sixty_over_dt, int t_dep, int sstep) {
    const double min_steps_to_full = (des - cap) / (pacmax / sixty_over_dt);
    const double departing_step = (double)(t_dep - sstep);
    if (min_steps_to_full > departing_step) {
        const double gap = (des - ((departing_step + 1.0) * pacmax / sixty_over_dt)) - cap;
        return 0.05 * (gap * gap);
    }
    return 0.0;
}
// The reward of a step from its env-level quantities.  costs: total profit of the step; usage: current_power_usage[t]; sp:
// power_setpoints[t]; pot_t / pot_tm1: charge_power_potential[t] / [t-1] (0 before the episode's first entry); over100: sum over
// the transformers of 100 * get_how_overloaded(); user: sum of ev2g_departure_term; tr0_maxp: transformers[0].max_power[t].
// usage_seq: current_power_usage as the reference accumulates it -- charger by charger (ev2gym_env.py:375), each charger's output EV by EV
// (ev_charger.py:180,196).  Every consumer of the usage tolerates the last-bit difference to the kernels' fixed summation tree except ONE:
// SquaredTrackingErrorRewardWithPenalty tests it against an exact zero (reward.py:50), and with V2G cancelling charge and discharge powers leave
// a rounding residue or an exact zero depending on the order.  The kernels therefore add the powers up a second time, sequentially, when (and
// only when) that reward is selected (ev2g_usage_seq); for every other reward usage_seq is simply `usage`.
struct RewardIn { double costs, usage, usage_seq, sp, pot_t, pot_tm1, over100, user, tr0_maxp; };
// the sequential sum over one env's `stage` row of port powers (LDS, slot order); npc > 0: equal port counts, else per charger (cs_np)
template <class IP>
__device__ __forceinline__ double ev2g_usage_seq(const double *pw_row, int C, int npc, IP cs_slot0, IP cs_np) {
    double u = 0.0;
    if (npc == 1) {   // single-port chargers: a charger's output is its EV's power; loads batched eight at a time, additions in order
        for (int c0 = 0; c0 < C; c0 += 8) {
            double x[8];
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = pw_row[cs_slot0[min(c0 + i, C - 1)]];
#pragma unroll
            for (int i = 0; i < 8; i++) if (c0 + i < C) u += x[i];
        }
    } else {
        for (int c = 0; c < C; c++) {
            const int q0 = cs_slot0[c], np = (npc > 0) ? npc : cs_np[c];
            double pw = 0.0;
            for (int j = 0; j < np; j++) pw += pw_row[q0 + j];
            u += pw;
        }
    }
    return u;
}
__device__ __forceinline__ double ev2g_reward(int kind, const RewardIn &x) {
    switch (kind) {
    case 1: { const double m = (x.pot_t < x.sp) ? x.pot_t : x.sp; const double d = m - x.usage; return -(d * d); }   // reward.py:7-14
    case 2: return x.costs - x.user;                                                                                 // :78-87
    case 3: {                                                                                                        // :16-32
        double m = x.sp;
        if (x.pot_t < m) m = x.pot_t;
        if (x.tr0_maxp < m) m = x.tr0_maxp;
        const double d = m - x.usage;
        return -(d * d) - x.over100 - x.user;
    }
    case 4: {                                                                                                        // :46-58
        const double m = (x.pot_t < x.sp) ? x.pot_t : x.sp; const double d = m - x.usage;
        return (x.usage_seq == 0.0 && x.pot_tm1 != 0.0) ? -(d * d) - 100.0 : -(d * d);
    }
    case 5: { const double d = x.sp - x.usage; return -(d * d); }                                                    // :60-65
    case 6: { double r = 0.0; if (x.sp < x.usage) r -= (x.usage - x.sp) * (x.usage - x.sp); return r + x.usage; }   // :67-76
    case 7: return x.costs;                                                                                          // :151-154
    case 8: return x.costs + (-x.user);                                                                              // :120-148
    case 9: return x.costs + (-x.user);                                                                              // :156-211
    case 10: return (x.costs + (-x.user)) + 1000.0 * ((x.sp < x.usage) ? (x.sp - x.usage) : 0.0);                   // :278-339
    default: return x.costs - x.over100 - x.user;                                                                    // :34-44
    }
}

// butterfly sum over aligned lane groups of width gs (power of two <= 64): fixed tree, deterministic
__device__ __forceinline__ double group_sum(double v, int gs) {
    for (int d = gs >> 1; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Observation columns that do not belong to a port: head + per-transformer windows, for env `e` at step
// counter `sstep` (= current_step after the increment), written cooperatively by `nl` lanes (lane id `l`).
template <class OT>
__device__ __forceinline__ void write_obs_env(const DevScn &s, OT *__restrict__ obs_e, int e /* scenario index */, int sstep,
                                              double usage_prev, int l, int nl) {
    const int T = s.T;
    if (s.state_kind == 1) {  // PublicPST state.py:6-35
        if (l == 0) {
            obs_e[0] = (OT)((double)sstep / (double)T);
            obs_e[1] = (OT)((sstep < T) ? s.setpoint[(long long)e * T + sstep] : 0.0);
            obs_e[2] = (OT)usage_prev;
        }
        return;
    }
    // V2G_profit_max(_loads) state.py:65-83, :108-135
    if (s.head_tab) {   // fast path: the row is in the observation head table (built from the same functions at load): a copy instead of the window logic
        const double *row = s.head_tab + ((long long)e * (T + 1) + sstep) * s.head_nh;
        for (int c = l; c < 2 + s.head_nh; c += nl) obs_e[c] = (OT)((c == 0) ? (double)sstep : ((c == 1) ? usage_prev : row[c - 2]));
        return;
    }
    for (int c = l; c < 22; c += nl) {
        double v;
        if (c == 0) v = (double)sstep;
        else if (c == 1) v = usage_prev;
        else {
            const int k = sstep + (c - 2);
            v = (k < T) ? fabs(s.price_ch[(long long)e * T + k]) : 0.0;
        }
        obs_e[c] = (OT)v;
    }
    if (s.state_kind == 0) {
        const int n = s.R * 40;
        for (int i = l; i < n; i += nl) {
            const int r = i / 40, j = i - r * 40;
            const int er = e * s.R + r;
            double v = (j < 20) ? load_minus_pv_at(s, (long long)er * T, sstep, j) : power_limit_at(s, er, sstep, j - 20);
            obs_e[s.tr_obs[r] + j] = (OT)v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// reset: EV2Gym.reset() state-init part (ev2gym_env.py:298-306,329-331; utils.py:794-861; ev_charger.py:96-112)
__global__ void __launch_bounds__(EV2G_BLOCK) ev2g_reset_kernel(DevScn s, DevState st, double *__restrict__ obs,
                                                                float *__restrict__ obs32, int scn_off) {
    const int grp = blockIdx.x;
    const int e0 = grp * s.G;
    const int ne = min(s.G, s.E - e0);
    const int P = s.P;
    for (int idx = threadIdx.x; idx < ne * P; idx += EV2G_BLOCK) {
        const int el = idx / P, q = idx - el * P;
        const int e = e0 + el;
        const long long g = (long long)e * P + q;
        const long long gs = (long long)ev2g_scn(e, scn_off, s.M) * P + q;   // this env's scenario for the coming episode
        const int2 w = s.port_first_win[gs];
        const int first = s.port_first[gs];
        st.line[g].ta = w.x; st.line[g].td = w.y;
        st.line[g].ss = first; st.line[g].cyc_lut = 0;
        st.line[g].cap = 0.0;
        st.line[g].tot = 0.0;
        st.line[g].prev = 0.0;
        st.port_energy[g] = 0.0;
        st.port_current[g] = 0.0;
        if (obs) {
            double *o = obs + (long long)e * s.D + s.slot_obs[q];
            o[0] = 0.0;
            o[1] = 0.0;
            if (s.state_kind == 1) o[2] = 0.0;
        }
        if (obs32) {
            float *o = obs32 + (long long)e * s.D + s.slot_obs[q];
            o[0] = 0.f;
            o[1] = 0.f;
            if (s.state_kind == 1) o[2] = 0.f;
        }
    }
    for (int idx = threadIdx.x; idx < ne * s.C; idx += EV2G_BLOCK) {
        const long long g = (long long)e0 * s.C + idx;
        st.cs_sat_sum[g] = 0.0;
        st.cs_served[g] = 0;
        if (st.cs_profits) { st.cs_profits[g] = 0.0; st.cs_e_ch[g] = 0.0; st.cs_e_dis[g] = 0.0; st.cs_power_now[g] = 0.0; st.cs_cur_now[g] = 0.0; }
    }
    for (int idx = threadIdx.x; idx < ne * 8; idx += EV2G_BLOCK) st.env_acc[(long long)e0 * 8 + idx] = 0.0;
    {   // usage | potential | overload histories (one slab): cleared here, grid-strided, instead of by a separate fill launch
        const long long n = (long long)s.T * s.E * (2 + s.R);
        for (long long i = (long long)blockIdx.x * EV2G_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * EV2G_BLOCK)
            st.slab_hist[i] = 0.0;
    }
    for (int idx = threadIdx.x; idx < ne; idx += EV2G_BLOCK) st.env_fault[e0 + idx] = 0;
    for (int idx = threadIdx.x; idx < ne * s.R; idx += EV2G_BLOCK) st.tr_power_now[(long long)e0 * s.R + idx] = 0.0;
    if (obs || obs32) {
        const int lpe = EV2G_BLOCK / ne;  // lanes per env
        const int el = threadIdx.x / lpe;
        if (el < ne) {
            const int scn = ev2g_scn(e0 + el, scn_off, s.M);
            if (obs) write_obs_env(s, obs + (long long)(e0 + el) * s.D, scn, 0, 0.0, threadIdx.x - el * lpe, lpe);
            if (obs32) write_obs_env(s, obs32 + (long long)(e0 + el) * s.D, scn, 0, 0.0, threadIdx.x - el * lpe, lpe);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// step: EV2Gym.step() for the envs of this workgroup, k_steps consecutive timesteps starting at t0.  The generic
// kernel: ports looped per thread, any P that fits the LDS staging (used above 1024 ports per env).
// Dynamic LDS: double stage[EV2G_NQ][N] ; tsum[EV2G_NQ][G*R] ; esum[EV2G_NQ][G] ; amask[N] (multi-port chargers)   (N = G*P)
__host__ __device__ inline size_t ev2g_generic_lds_bytes(int G, int P, int R, int npc) {
    return sizeof(double) * ((size_t)EV2G_NQ * G * P + (size_t)EV2G_NQ * G * R + (size_t)EV2G_NQ * G + (npc > 1 ? (size_t)G * P : 0));
}
__global__ void __launch_bounds__(EV2G_BLOCK) ev2g_step_kernel(DevScn s, DevState st, StepIO io, StepExtras xt, int t0, int k_steps,
                                                               int auto_reset) {
    extern __shared__ double lds[];
    // XCD-aware env-group mapping: workgroup b runs on XCD b % 8 (observed dispatch order); give every XCD a
    // contiguous range of env groups so that each XCD's L2 holds a contiguous slice of the state arrays.
    int grp;
    {
        const int nb = gridDim.x, b = blockIdx.x;
        const int per = nb >> 3;
        grp = (nb & 7) == 0 ? (b & 7) * per + (b >> 3) : b;
    }
    const int P = s.P, R = s.R, T = s.T, C = s.C, npc = s.npc;
    const int e0 = grp * s.G;
    const int ne = min(s.G, s.E - e0);
    const int N = ne * P;
    const int NS = s.G * P;  // stride of the staging arrays
    double *stage = lds;
    double *tsum = lds + (size_t)EV2G_NQ * NS;
    double *esum = tsum + (size_t)EV2G_NQ * s.G * R;
    double *amask = esum + (size_t)EV2G_NQ * s.G;   // [NS] this step's action of every port, 0 where the port is empty (npc > 1)
    const int tid = threadIdx.x;
    const bool log_cs = st.cs_profits != nullptr;
    int off = io.scn_off;

    int t = t0;
    for (int kk = 0; kk < k_steps; kk++) {
        if (t >= T) {  // episode finished inside a fused run
            if (!auto_reset) break;
            off = ev2g_scn(off, io.scn_stride, s.M);
            // in-kernel ev2g_reset for this workgroup's envs
            for (int idx = tid; idx < N; idx += EV2G_BLOCK) {
                const int el = idx / P, q = idx - el * P;
                const long long g = (long long)e0 * P + idx;
                const long long gs = (long long)ev2g_scn(e0 + el, off, s.M) * P + q;
                { const int2 w0 = s.port_first_win[gs]; st.line[g].ta = w0.x; st.line[g].td = w0.y; }
                st.line[g].ss = s.port_first[gs]; st.line[g].cyc_lut = 0;
                st.port_energy[g] = 0.0;
                st.port_current[g] = 0.0;
            }
            for (int idx = tid; idx < ne * C; idx += EV2G_BLOCK) {
                const long long g = (long long)e0 * C + idx;
                st.cs_sat_sum[g] = 0.0;
                st.cs_served[g] = 0;
                if (log_cs) { st.cs_profits[g] = 0.0; st.cs_e_ch[g] = 0.0; st.cs_e_dis[g] = 0.0; }
            }
            for (int idx = tid; idx < ne * 8; idx += EV2G_BLOCK) st.env_acc[(long long)e0 * 8 + idx] = 0.0;
            for (int idx = tid; idx < ne; idx += EV2G_BLOCK) st.hist[EV2G_HIST(e0 + idx, 0, T, s.R) + 1] = 0.0;
            t = 0;
            __syncthreads();
        }
        const long long a_off = (long long)kk * io.a_stride + (io.actions ? 0 : (long long)io.step0 * io.a_stride);
        double *__restrict__ obs = io.obs ? io.obs + (long long)kk * io.o_stride : nullptr;
        float *__restrict__ obs32 = xt.obs32 ? xt.obs32 + (long long)(io.step0 + kk) * xt.o32_stride : nullptr;
        uint8_t *__restrict__ mask = io.mask ? io.mask + (long long)kk * io.m_stride : nullptr;
        const int sstep = t + 1;

        // ---------------- phase 0 (multi-port chargers): the actions the normalisation sums, ev_charger.py:137-149 ----------------
        // Snapshot BEFORE any window changes: phase 1 frees departing ports, and a charger's ports may sit in different
        // wavefronts / loop iterations -- the sum must see this step's occupancy, not the post-departure one.
        if (npc > 1) {
            for (int idx = tid; idx < N; idx += EV2G_BLOCK) {
                const int el = idx / P, q = idx - el * P;
                const int e = e0 + el;
                const int2 w = make_int2(st.line[(long long)e * P + q].ta, st.line[(long long)e * P + q].td);
                const bool occ = (w.x <= t) && (t <= w.y);
                amask[idx] = occ ? ev2g_action(io, xt.act32, a_off, (long long)e * P + s.slot_port[q]) : 0.0;
                if (s.het && mask) mask[(long long)e * P + q] = 0;   // entries are OR-ed below: two ports can share one (ev2gym_env.py:452-457)
            }
            __syncthreads();
        }

        // ---------------- phase 1: per port ----------------
        for (int idx = tid; idx < N; idx += EV2G_BLOCK) {
            const int el = idx / P, q = idx - el * P;
            const int e = e0 + el;
            const int scn = ev2g_scn(e, off, s.M);
            const long long g = (long long)e * P + q;
            const int cs = s.slot_cs[q];
            const int pref = s.slot_port[q];
            int2 w = make_int2(st.line[g].ta, st.line[g].td);
            const bool occ = (w.x <= t) && (t <= w.y);
            double a;
            if (npc == 1) {     // ev_charger.py:143-149 with one port: a/a
                a = occ ? ev2g_action(io, xt.act32, a_off, (long long)e * P + pref) : 0.0;  // ev_charger.py:137-140
                if (a > 1.0) a = a / a;
                else if (a < -1.0) a = -a / a;
            } else {
                a = amask[idx];
                const int j0 = idx - (pref - s.cs_pbase[cs]);  // slot of the charger's port 0 (ports of a charger are adjacent, in port order)
                const int np = s.cs_np[cs];
                double S = 0.0;
                for (int j = 0; j < np; j++) S = S + amask[j0 + j];   // sequential python sum()
                if (S > 1.0) a = a / S;
                else if (S < -1.0) a = -a / S;
            }
            double energy = 0.0, current = 0.0, profit = 0.0, e_ch = 0.0, e_dis = 0.0, satpen = 0.0, pot = 0.0, emerg = 0.0;
            double cap = 0.0, tot_e = 0.0;
            int ss = -1;
            if (occ) {
                const double x = rnd5(a);
                int2 sc = make_int2(st.line[g].ss, ev2g_line_cycles(st.line[g].cyc_lut));
                ss = sc.x;
                cap = st.line[g].cap;
                tot_e = st.line[g].tot;
                const double cap_before = cap;
                if (x != 0.0) {
                    double amps;
                    const int ph = s.cs_ph[cs];
                    if (x > 0.0) {
                        amps = x * s.cs_imax[cs];
                        if (amps < s.cs_imin[cs] - 0.01) amps = 0.0;
                    } else {
                        amps = x * s.cs_dmax_abs[cs];
                        if (amps > s.cs_dmin[cs] - 0.01) amps = s.cs_dmin[cs];
                    }
                    double prev_power = st.line[g].prev;
                    int cycles = sc.y;
                    EvOut o = ev_step(s, ss, cs, amps, ph, cap, tot_e, prev_power, cycles);
                    energy = o.energy;
                    current = o.current;
                    emerg = (double)o.emerg;
                    const double ae = fabs(energy);
                    if (x > 0.0) { profit = ae * s.price_ch[(long long)scn * T + t]; e_ch = ae; }
                    else         { profit = ae * s.price_dis[(long long)scn * T + t]; e_dis = ae; }
                    if (o.active) {
                        st.line[g].cap = cap;
                        st.line[g].tot = tot_e;
                        st.line[g].prev = prev_power;
                        if (cycles != sc.y) st.line[g].cyc_lut = ev2g_line_pack(cycles, s.ss_lut[ss]);
                    }
                }
                st.port_energy[g] = energy;
                st.port_current[g] = current;
                if (st.soc_log) {  // historic_soc / active_steps (ev.py:156,162,185) and abs_total_energy_exchanged (:180)
                    st.soc_log[((long long)e * T + t) * P + q] = (current != 0.0) ? cap_before : -cap_before;
                    if (energy != 0.0) st.line[g].abse += fabs(energy);
                }
                // departure (ev_charger.py:209-229, ev.py:191-214)
                if (t >= w.y) {
                    const double des = s.ss_des[ss];
                    const double score = (cap < des - 0.001) ? cap / des : 1.0;
                    satpen = ev2g_departure_term(s.reward_kind, s.cost_kind, score, cap, des);
                    const long long gc = (long long)e * C + cs;
                    if (npc == 1) {
                        st.cs_served[gc] += 1;
                        st.cs_sat_sum[gc] += score;
                    } else {
                        atomicAdd(&st.cs_served[gc], 1);
                        atomicAdd(&st.cs_sat_sum[gc], score);
                    }
                    st.sess_final_cap[ss] = cap;
                    if (st.soc_log) st.sess_abs_e[ss] = st.line[g].abse;
                    w = make_int2(s.ss_ntarr[ss], s.ss_ntdep[ss]);
                    ss = (w.x != EV2G_INT_MAX) ? ss + 1 : -1;
                    st.line[g].ta = w.x; st.line[g].td = w.y;
                    st.line[g].ss = ss; st.line[g].cyc_lut = 0;
                }
            }
            // arrival at the end of this step (ev2gym_env.py:399-417, ev.py:115-136)
            bool occ_after = (w.x <= sstep) && (sstep <= w.y);
            if (w.x == sstep) {
                if (ss < 0) ss = st.line[g].ss;
                cap = s.ss_cap0[ss];
                tot_e = 0.0;
                st.line[g].cap = cap;
                st.line[g].tot = 0.0;
                st.line[g].prev = 0.0;
                if (st.soc_log) st.line[g].abse = 0.0;
                st.line[g].cyc_lut = ev2g_line_pack(0, s.ss_lut[ss]);   // (the line stays complete whichever kernel wrote it)
                st.line[g].bcap = s.ss_B[ss]; st.line[g].potc = s.rec[ss].potc;
                st.port_energy[g] = 0.0;
                st.port_current[g] = 0.0;
            }
            if (occ_after && s.reward_kind >= 9) satpen += ev2g_connected_term(s.ss_des[ss], cap, s.ss_pacmax[ss], s.sixty_over_dt, w.y, sstep);
            if (mask) {
                if (!s.het) mask[(long long)e * P + pref] = occ_after ? 1 : 0;
                else if (occ_after) mask[(long long)e * P + s.slot_mask[q]] = 1;
            }
            // per-port observation columns + charge power potential (state.py, utils.py:760-791)
            double o0 = 0.0, o1 = 0.0, o2 = 0.0;
            if (occ_after) {
                const double B = s.ss_B[ss];
                const double soc = cap / B;
                if (s.state_kind == 1) { o0 = (soc == 1.0) ? 1.0 : 0.5; o1 = tot_e; o2 = (double)(sstep - w.x); }
                else { o0 = soc; o1 = (double)(w.y - sstep); }
                if (soc < 1.0 && w.y > sstep) {
                    const int ph = s.cs_ph[cs];
                    const int evph = s.ss_phases[ss];
                    const int k = evph < ph ? evph : ph;
                    const double sq_v = s.cs_vk[cs * 4 + k];  // sqrt(min(phases, ev_phases)) * voltage
                    const double ev_current = s.ss_pacmax[ss] * 1000.0 / sq_v;
                    const double cur = (ev_current < s.cs_imax[cs]) ? ev_current : s.cs_imax[cs];
                    pot = sq_v * cur / 1000.0;
                }
            }
            if (npc == 1) {  // per-charger clamp (utils.py:779-789)
                const double mx = s.cs_maxp[cs], mn = s.cs_minp[cs];
                pot = (pot > mx) ? mx : ((pot < mn) ? 0.0 : pot);
            }
            if (obs) {
                double *o = obs + (long long)e * s.D + s.slot_obs[q];
                o[0] = o0;
                o[1] = o1;
                if (s.state_kind == 1) o[2] = o2;
            }
            if (obs32) {
                float *o = obs32 + (long long)e * s.D + s.slot_obs[q];
                o[0] = (float)o0;
                o[1] = (float)o1;
                if (s.state_kind == 1) o[2] = (float)o2;
            }
            stage[0 * NS + idx] = energy * 60.0 / (double)s.dt;  // contribution to current_power_output
            stage[1 * NS + idx] = current;
            stage[2 * NS + idx] = profit;
            stage[3 * NS + idx] = satpen;
            stage[4 * NS + idx] = pot;
            stage[5 * NS + idx] = e_ch;
            stage[6 * NS + idx] = e_dis;
            stage[7 * NS + idx] = emerg;
        }
        __syncthreads();

        // ---------------- phase 1b: per charger (multi-port chargers, or charger history) ----------------
        if (npc > 1 || log_cs) {
            for (int idx = tid; idx < N; idx += EV2G_BLOCK) {
                const int el = idx / P, q = idx - el * P;
                const int cs = s.slot_cs[q];
                const int pref = s.slot_port[q];
                if (pref != s.cs_pbase[cs]) continue;  // leader = port 0 of the charger
                const int np = s.cs_np[cs];
                const int e = e0 + el;
                double pw = 0.0, cur = 0.0, pr = 0.0, ec = 0.0, ed = 0.0, pp = 0.0;
                bool fault = false;
                for (int j = 0; j < np; j++) {  // sequential, port order (ev_charger.py:155-205)
                    pw += stage[0 * NS + idx + j];
                    cur += stage[1 * NS + idx + j];
                    pr += stage[2 * NS + idx + j];
                    ec += stage[5 * NS + idx + j];
                    ed += stage[6 * NS + idx + j];
                    pp += stage[4 * NS + idx + j];
                    if (cur - 0.0001 > s.cs_imax[cs]) fault = true;
                }
                if (fault) st.env_fault[e] = 1;
                if (npc > 1) {
                    const double mx = s.cs_maxp[cs], mn = s.cs_minp[cs];
                    pp = (pp > mx) ? mx : ((pp < mn) ? 0.0 : pp);
                    stage[4 * NS + idx] = pp;
                    for (int j = 1; j < np; j++) stage[4 * NS + idx + j] = 0.0;
                }
                if (log_cs) {
                    const long long gc = (long long)e * C + cs;
                    st.cs_profits[gc] += pr;
                    st.cs_e_ch[gc] += ec;
                    st.cs_e_dis[gc] += ed;
                    st.cs_power_now[gc] = pw;
                    st.cs_cur_now[gc] = cur;
                    st.cs_power_hist[((long long)t * s.E + e) * C + cs] = pw;
                    st.cs_cur_hist[((long long)t * s.E + e) * C + cs] = cur;
                }
            }
            __syncthreads();
        } else {
            // single-port chargers: over-current check per port (ev_charger.py:203-205)
            for (int idx = tid; idx < N; idx += EV2G_BLOCK) {
                const int el = idx / P, q = idx - el * P;
                if (stage[1 * NS + idx] - 0.0001 > s.cs_imax[s.slot_cs[q]]) st.env_fault[e0 + el] = 1;
            }
        }

        // ---------------- phase 2: LDS-staged segmented reduction ----------------
        {
            const int gs = s.gs;
            const int gid = tid / gs, gl = tid - gid * gs;
            const int ngr = EV2G_BLOCK / gs;
            const int ntask = ne * R;
            const int npass = (ntask + ngr - 1) / ngr;
            for (int pass = 0; pass < npass; pass++) {
                const int task = pass * ngr + gid;
                const bool live = task < ntask;
                const int el = live ? task / R : 0;
                const int r = live ? task - el * R : 0;
                const int a = el * P + s.tr_seg[r], b = el * P + s.tr_seg[r + 1];
                double acc[EV2G_NQ];
#pragma unroll
                for (int k = 0; k < EV2G_NQ; k++) acc[k] = 0.0;
                if (live)
                    for (int i = a + gl; i < b; i += gs) {
#pragma unroll
                        for (int k = 0; k < EV2G_NQ; k++) acc[k] += stage[k * NS + i];
                    }
#pragma unroll
                for (int k = 0; k < EV2G_NQ; k++) acc[k] = group_sum(acc[k], gs);
                if (live && gl == 0) {
#pragma unroll
                    for (int k = 0; k < EV2G_NQ; k++) tsum[k * (s.G * R) + task] = acc[k];
                }
            }
        }
        __syncthreads();
        // fold the R transformer partials of each env: thread (el, k)
        for (int idx = tid; idx < ne * EV2G_NQ; idx += EV2G_BLOCK) {
            const int el = idx / EV2G_NQ, k = idx - el * EV2G_NQ;
            double v = 0.0;
            for (int r = 0; r < R; r++) v += tsum[k * (s.G * R) + el * R + r];
            esum[k * s.G + el] = v;
        }
        __syncthreads();

        // ---------------- phase 3: per env ----------------
        {
            const int lpe = EV2G_BLOCK / ne;  // lanes per env
            const int el = tid / lpe, l = tid - el * lpe;
            if (el < ne) {
                const int e = e0 + el;
                const int scn = ev2g_scn(e, off, s.M);
                const double usage = esum[0 * s.G + el];
                // transformers: Transformer.reset + step + get_how_overloaded (transformer.py:258-302)
                double over_sum = 0.0;  // only lane 0 uses it
                if (l == 0) {
                    for (int r = 0; r < R; r++) {
                        const long long erT = ((long long)scn * R + r) * T + t;
                        double ptr = s.tr_infl[erT] + s.tr_solar[erT];
                        ptr += tsum[0 * (s.G * R) + el * R + r];
                        const double mx = s.tr_maxp[erT], mn = s.tr_minp[erT];
                        const double over = (ptr > mx + 0.0001 || ptr < mn - 0.0001) ? fabs(ptr - mx) : 0.0;
                        st.hist[EV2G_HIST(e, t, T, R) + 2 + r] = over;
                        st.tr_power_now[(long long)e * R + r] = ptr;
                        over_sum += 100.0 * over;
                    }
                    st.hist[EV2G_HIST(e, t, T, R)] = usage;
                    const double pot = esum[4 * s.G + el];
                    if (sstep < T) st.hist[EV2G_HIST(e, sstep, T, R) + 1] = pot;
                    const double costs = esum[2 * s.G + el];
                    RewardIn ri;
                    ri.costs = costs; ri.usage = usage; ri.over100 = over_sum; ri.user = esum[3 * s.G + el];
                    ri.usage_seq = (s.reward_kind == 4) ? ev2g_usage_seq(stage + (size_t)el * P, C, s.het ? 0 : s.npc, s.cs_slot0, s.cs_np) : usage;
                    ri.sp = s.setpoint[(long long)scn * T + t];
                    ri.pot_t = st.hist[EV2G_HIST(e, t, T, R) + 1];
                    ri.pot_tm1 = (t > 0) ? st.hist[EV2G_HIST(e, t - 1, T, R) + 1] : 0.0;
                    ri.tr0_maxp = s.tr_maxp[(long long)scn * R * T + t];
                    const double reward = ev2g_reward(s.reward_kind, ri);
                    double *acc = st.env_acc + (long long)e * 8;
                    acc[0] += reward;
                    acc[1] += costs;
                    acc[2] += esum[5 * s.G + el];
                    acc[3] += esum[6 * s.G + el];
                    acc[4] += esum[7 * s.G + el];
                    if (io.reward) io.reward[(long long)kk * io.r_stride + e] = reward;
                    if (io.done) io.done[(long long)kk * io.d_stride + e] = (sstep >= T) ? 1 : 0;
                    if (xt.cost)   // cost_function (rl_agent/cost.py:8-27)
                        xt.cost[(long long)(io.step0 + kk) * xt.c_stride + e] = (s.cost_kind == 2) ? costs : over_sum + esum[3 * s.G + el];
                }
                if (obs) write_obs_env(s, obs + (long long)e * s.D, scn, sstep, usage, l, lpe);
                if (obs32) write_obs_env(s, obs32 + (long long)e * s.D, scn, sstep, usage, l, lpe);
            }
        }
        t += 1;
        __syncthreads();
    }
}

// One-off at load time: materialise the per-(env, transformer, observation step) 40-wide window
// [ (loads - pv)[20] | power_limits[20] ] (transformer.py:142-188) so that the step kernel streams it with
// one coalesced load per lane instead of re-deriving it through dependent gathers every step.
// (m0, m1: the scenario slots to (re)build -- all of them at load, the refilled ones after ev2g_pool_refill)
__global__ void ev2g_build_window_table_kernel(DevScn s, double *__restrict__ tab, int m0, int m1) {
    const long long per = (long long)s.R * (s.T + 1) * 40, n = per * m1;
    for (long long i = per * m0 + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i % 40);
        const long long k = i / 40;
        const int step = (int)(k % (s.T + 1));
        const int er = (int)(k / (s.T + 1));
        tab[i] = (j < 20) ? load_minus_pv_at(s, (long long)er * s.T, step, j) : power_limit_at(s, er, step, j - 20);
    }
}

// Observation head table of the one-transformer fast path: row (env, step s) holds columns 2..2+NH of the observation
// that describes step s -- |charge price| for steps s..s+19, zero past the horizon (state.py:75-83 / :121-129), then,
// when the state has them (NH == 60), the 40 window columns of ev2g_build_window_table_kernel.  The step kernel copies
// a row per env-step with one base pointer and no per-column logic.
__global__ void ev2g_build_head_table_kernel(const double *__restrict__ price_ch, const double *__restrict__ win_tab,
                                             int m0, int m1, int T, int NH, double *__restrict__ tab) {
    const long long per = (long long)(T + 1) * NH, n = per * m1;
    for (long long i = per * m0 + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % NH);
        const long long row = i / NH;
        const int step = (int)(row % (T + 1));
        const long long e = row / (T + 1);
        double v;
        if (c < 20) { const int k = step + c; v = (k < T) ? fabs(price_ch[e * T + k]) : 0.0; }
        else v = win_tab[row * 40 + (c - 20)];
        tab[i] = v;
    }
}

// Per (env, step) scalars of the one-transformer fast path, interleaved so that one base pointer and three 16-byte
// loads fetch them: {charge price, discharge price, inflexible+solar, max_power, min_power, setpoint, 0, 0}.
__global__ void ev2g_build_step_table_kernel(DevScn s, double *__restrict__ tab, int m0, int m1) {
    const long long n = (long long)m1 * s.T;
    for (long long i = (long long)m0 * s.T + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        double *o = tab + i * 8;
        o[0] = s.price_ch[i]; o[1] = s.price_dis[i]; o[2] = s.tr_base[i]; o[3] = s.tr_maxp[i]; o[4] = s.tr_minp[i];
        o[5] = s.setpoint[i]; o[6] = 0.0; o[7] = 0.0;
    }
}

// Occupancy does not depend on the actions (ev.py:191-202; arrivals are fixed when the scenario is drawn): which ports hold an EV during step t,
// and which receive one at its end, is a property of (scenario, step).  One wavefront per scenario (P <= 64: the fast path), a lane per port
// slot, walks the port's session chain exactly like the step kernels do and leaves the two 64-bit masks in slots 6 and 7 of the step table --
// what a single-step launch reads first, so that it fetches state lines only for those ports (ev2g_step_wave.h, prologue).
__global__ void __launch_bounds__(64) ev2g_build_occ_mask_kernel(DevScn s, double *__restrict__ step_tab, int m0, int m1) {
    const int lane = threadIdx.x, P = s.P, T = s.T;
    for (int m = m0 + blockIdx.x; m < m1; m += gridDim.x) {
        int ta = EV2G_INT_MAX, td = -1, ss = -1;
        if (lane < P) {
            const long long gs = (long long)m * P + lane;
            const int2 w = s.port_first_win[gs];
            ta = w.x; td = w.y; ss = s.port_first[gs];
        }
        for (int t = 0; t < T; t++) {
            const bool occ = (ta <= t) && (t <= td);
            if (occ && t >= td) {   // departure during step t: the window of the port's next session (ev_charger.py:209-229)
                const SessTail tl = s.tail[ss];
                ta = tl.nt_arr; td = tl.nt_dep;
                ss = (ta != EV2G_INT_MAX) ? ss + 1 : -1;
            }
            const unsigned long long m_occ = __ballot(occ), m_arr = __ballot(ta == t + 1);
            if (lane == 0) {
                double *o = step_tab + ((long long)m * T + t) * 8;
                o[6] = __longlong_as_double((long long)m_occ); o[7] = __longlong_as_double((long long)m_arr);
            }
        }
    }
}

// counter-based uniform generator (splitmix64 of (seed, index)); identical on host (ev2g_host_uniform)
__host__ __device__ inline double ev2g_u01(uint64_t seed, uint64_t i) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

__global__ void ev2g_fill_uniform_kernel(double *dst, long long n, uint64_t seed, double lo, double hi) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = lo + (hi - lo) * ev2g_u01(seed, (uint64_t)i);
}

// episode statistics (get_statistics utils.py:12-123) -- one wavefront per env, lanes strided over chargers / steps /
// ports; partial sums are combined with fixed xor-butterflies (bit-reproducible).
__device__ __forceinline__ double wave_sum(double v) {
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
    for (int d = 32; d > 0; d >>= 1) v = fmin(v, __shfl_xor(v, d, 64));
    return v;
}

// sessions [first, last) of port g have been spawned; `attached`: the last of them is still on the port
__device__ __forceinline__ void port_sessions(const DevScn &s, const DevState &st, long long g, long long gs, int cur_step, int &first,
                                              int &last, bool &attached) {
    first = s.port_first[gs];   // gs: the port in the scenario pool, g: the port in the env state
    last = first;
    attached = false;
    if (first < 0) return;
    const int2 w = make_int2(st.line[g].ta, st.line[g].td);
    const int cur = st.line[g].ss;  // attached-or-next session, -1 when the port's list is exhausted
    if (cur < 0) {
        last = s.port_end[gs];
    } else {
        attached = (w.x <= cur_step);  // spawned at the end of step t_arr-1
        last = attached ? cur + 1 : cur;
    }
}

// reductions over the W = 64 / EPWS lanes of one env's segment of a wavefront (EPWS == 1: the whole wavefront, wave_sum_dpp's order)
template <int EPWS> __device__ __forceinline__ double seg_sum(double v) {
    v += xor1_f64(v);
    v += xor2_f64(v);
    v += xor4_f64(v);
    v += dpp_mov_f64<0x128>(v);   // row_ror:8 -- every lane now holds the sum of its 16-lane row
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    if (EPWS == 1) return (r0 + r1) + (r2 + r3);
    return (threadIdx.x < 32) ? r0 + r1 : r2 + r3;
}
template <int EPWS> __device__ __forceinline__ double seg_min(double v) {
    v = fmin(v, xor1_f64(v));
    v = fmin(v, xor2_f64(v));
    v = fmin(v, xor4_f64(v));
    v = fmin(v, dpp_mov_f64<0x128>(v));
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    if (EPWS == 1) return fmin(fmin(r0, r1), fmin(r2, r3));
    return (threadIdx.x < 32) ? fmin(r0, r1) : fmin(r2, r3);
}
template <int EPWS> __device__ __forceinline__ bool seg_all(bool p) {
    const unsigned long long m = __ballot(p);
    if (EPWS == 1) return m == ~0ull;
    return (threadIdx.x < 32) ? ((unsigned)m == 0xffffffffu) : ((unsigned)(m >> 32) == 0xffffffffu);
}

// RESET: the episode-end sequence "statistics, then reset onto the next scenario window" (what an auto-resetting vectorised env does) in ONE
// launch: the wavefront that computed an env's statistics re-arms that env's state right behind them (ev2g_reset_kernel's work for one env:
// state lines from the new window's first-session tables, charger and env accumulators, history rows, reset observation) -- one kernel
// launch and one cold start per episode less (ev2g_get_stats_reset, include/ev2g.h).
template <int EPWS, bool RESET = false>
__global__ void __launch_bounds__(64) ev2g_stats_kernel(DevScn s, DevState st, int scn_off,
                                                        const double *__restrict__ ss_afap, int cur_step,
                                                        double *__restrict__ out, int reset_off = 0, double *__restrict__ r_obs = nullptr,
                                                        float *__restrict__ r_obs32 = nullptr) {
    // EPWS envs share a wavefront (W = 64 / EPWS lanes each): a wavefront's time is its chain of dependent memory round trips, not its lane
    // count, so small envs (their sessions fit 32 lanes) are paired -- half the wavefronts for the same chain (ev2g_get_stats decides).
    constexpr int W = 64 / EPWS;
    const int sub = threadIdx.x / W, lane = threadIdx.x - sub * W;   // `lane`: inside the env's segment
    const int e_raw = blockIdx.x * EPWS + sub;
    const bool e_valid = e_raw < s.E;
    const int e = e_valid ? e_raw : s.E - 1;   // (an odd env count: the idle segment recomputes the last env and stores nothing)
    const int scn = ev2g_scn(e, scn_off, s.M);
    const int T = s.T, C = s.C, R = s.R, P = s.P;
    double served = 0.0, sat = 0.0, nsat = 0.0;
    for (int c = lane; c < C; c += W) {
        const int n = st.cs_served[(long long)e * C + c];
        served += n;
        if (n > 0) { sat += st.cs_sat_sum[(long long)e * C + c] / n; nsat += 1.0; }
    }
    served = seg_sum<EPWS>(served); sat = seg_sum<EPWS>(sat); nsat = seg_sum<EPWS>(nsat);
    double over = 0.0, te = 0.0, ete = 0.0, ptv = 0.0;
    for (int t = lane; t < T; t += W) {
        // steps the running episode has not reached count as zeros (the reference's arrays are zero-initialised at reset); the
        // history slab may still hold the previous episode's values there after an in-kernel reset of a fused run
        const bool past = t < cur_step;
        if (past) for (int r = 0; r < R; r++) over += st.hist[EV2G_HIST(e, t, T, R) + 2 + r];
        const double sp = s.setpoint[(long long)scn * T + t], u = past ? st.hist[EV2G_HIST(e, t, T, R)] : 0.0;
        const double d = sp - u;
        te += d * d;
        ete += fabs(d);
        if (u > sp) ptv += u - sp;
    }
    over = seg_sum<EPWS>(over); te = seg_sum<EPWS>(te); ete = seg_sum<EPWS>(ete); ptv = seg_sum<EPWS>(ptv);
    ete *= (double)s.dt / 60.0;
    // energy user satisfaction (utils.py:57-63) and battery degradation (ev.py:442-521) over every spawned session
    const double e0 = 7.543e6, e1 = 23.75e6, e2 = 6976, z0 = 7.348e-3, z1 = 3.667, z2 = 7.6e-4, z3 = 4.081e-3;
    const double b_cap_ah = 2.05, b_cap_kwh = 78, d_dist = 15000, b_age = 2 * 365, G_ = 0.186;
    const double theta = 298.15, kk = 0.8263, v_min = 3.3324;
    // per-session constants of get_battery_degradation, evaluated once (same operations, same values)
    const double k_arrh = exp(-e2 / theta), k_age = pow(b_age, 0.25);
    const double Q_acc = 2 * (b_age * (d_dist / 365) * G_ * b_cap_ah) / b_cap_kwh, k_qacc = pow(Q_acc, 0.5);
    const bool log_soc = st.soc_log != nullptr;
    double sum = 0.0, mn = INFINITY, cnt = 0.0, deg_cal = 0.0, deg_cyc = 0.0;
    // One SESSION per lane (an env has ~0.7 sessions per port: 35 at cfg2): every lane's chain is the two or three memory round
    // trips of ONE session's SoC log.  With a lane per port the wavefront waited for its busiest port (up to six sessions in a row).
    // The satisfaction values of a lane's first two sessions are kept for the variance pass below instead of being fetched again.
    double vkeep[2];
    int nkeep = 0;
    // Which sessions have been spawned so far, and which of them is still attached, follows from the session's own window (arrivals and
    // departures do not depend on the actions: a session is attached from step t_arr until the step t_dep has run; the loader refuses
    // sessions that would depart before they arrive): everything a lane needs of its session depends on the session index alone and is
    // requested in ONE round trip -- the port's state line (the attached EV's capacity) and the SoC log follow in a second one.  (Round 3
    // walked the port's chain first: first session, the line's current session, last session -- three dependent round trips more.)
    const int d0 = s.scn_sess[scn], d1 = s.scn_sess_end[scn];
    for (int k0 = d0; k0 < d1; k0 += W) {
        const int k = min(k0 + lane, d1 - 1);
        const bool has = k0 + lane < d1;
        const int q = s.ss_slot[k], ta = s.ss_tarr[k], td = s.ss_tdep[k];
        const double B = s.ss_B[k], afap = ss_afap[k], fin_cap = st.sess_final_cap[k];
        const double fin_abs = log_soc ? st.sess_abs_e[k] : 0.0;
        const long long g = (long long)e * P + q;
        const bool spawned = has && ta <= cur_step;   // spawned so far
        const bool live = td >= cur_step;             // still attached
        const double l_cap = st.line[g].cap, l_abs = st.line[g].abse;
        const double capk = live ? l_cap : fin_cap;
        if (spawned) {
            const double v = capk / afap * 100.0;
            sum += v;
            mn = fmin(mn, v);
            cnt += 1.0;
            if (nkeep < 2) vkeep[nkeep] = v;
            nkeep++;
        }
        if (spawned) {
            if (log_soc) {
                const double *__restrict__ slog = st.soc_log + (long long)e * T * P + q;   // this port's column of the env's [T, P] block
                const int tend = min(td, cur_step - 1);
                const double soc_f = capk / B;
                // historic_soc entries are capacity / battery_capacity (ev.py:156): one reciprocal per session and a multiplication per
                // entry instead of a float64 division per entry and pass.  Each entry differs from the quotient by at
                // most one ulp; the statistics are sums of ~30 of them, compared at 1e-9 (relative) like every float64 output.
                const double invB = 1.0 / B;
                double hs = 0.0, fs = 0.0;
                int n = 0, nf = 0;
                // historic_soc / active_steps (sign bit set = inactive step).  The first NK entries of a session are fetched by unconditional
                // (clamped) loads issued together and KEPT in registers for the second pass; only the tail of longer sessions is read twice,
                // in batches of eight.  Accumulation order is the sequential one.  NK = 40 (80 registers, two wavefronts per SIMD) since the end of
                // round 4: the kernel's time follows the number of dependent batches, not the occupancy -- NK 12 / 24 / 32 / 40 / 56: 43.5 / 41.6 /
                // 39.9 / 38.6 / 55 us at cfg2, 52.0 -> 45.2 at cfg3, 482 -> 380 at cfg4 (profiles/r04_stats_log_pass_variants.txt).
#ifndef EV2G_STATS_NK
#define EV2G_STATS_NK 40
#endif
#ifndef EV2G_STATS_TB
#define EV2G_STATS_TB 8   // entries per batch of a session's tail (beyond the NK kept ones)
#endif
#ifndef EV2G_STATS_LK
#define EV2G_STATS_LK 16  // entries behind the kept ones that the first pass parks in LDS for the second (8 KB per wavefront; 16 / 32 / 48: 37.2 / 37.3 / 41.8 us at cfg2 against 38.5 without)
#endif
                constexpr int NK = EV2G_STATS_NK;
                constexpr int LK = EV2G_STATS_LK;   // (a multiple of the tail batch)
                extern __shared__ double l_keep[];  // [LK][64]
                double xk[NK];
#pragma unroll
                for (int u = 0; u < NK; u++) xk[u] = slog[(long long)min(ta + u, tend) * P];
#pragma unroll
                for (int u = 0; u < NK; u++) {
                    if (ta + u <= tend) {
                        const double soc = fabs(xk[u]) * invB;
                        hs += soc; n++;
                        if (__double_as_longlong(xk[u]) >= 0) { fs += soc; nf++; }
                    }
                }
                for (int t = ta + NK; t <= tend; t += EV2G_STATS_TB) {
                    double x[EV2G_STATS_TB];
#pragma unroll
                    for (int u = 0; u < EV2G_STATS_TB; u++) x[u] = slog[(long long)min(t + u, tend) * P];
                    if (LK > 0 && t - ta - NK < LK) {   // the next LK entries behind the kept ones are parked in LDS for the second pass ([entry][lane]: no bank conflicts)
#pragma unroll
                        for (int u = 0; u < EV2G_STATS_TB; u++) l_keep[(t - ta - NK + u) * 64 + (int)threadIdx.x] = x[u];
                    }
#pragma unroll
                    for (int u = 0; u < EV2G_STATS_TB; u++) {
                        if (t + u <= tend) {
                            const double soc = fabs(x[u]) * invB;
                            hs += soc; n++;
                            if (__double_as_longlong(x[u]) >= 0) { fs += soc; nf++; }
                        }
                    }
                }
                hs += soc_f; n++;
                fs += soc_f; nf++;
                const double avg_soc = hs / n, avg_f = fs / nf;
                double mad = 0.0;
#pragma unroll
                for (int u = 0; u < NK; u++)
                    if (ta + u <= tend && __double_as_longlong(xk[u]) >= 0) mad += fabs(avg_f - fabs(xk[u]) * invB);
                for (int t = ta + NK; t <= tend; t += EV2G_STATS_TB) {
                    double x[EV2G_STATS_TB];
                    if (LK > 0 && t - ta - NK < LK) {
#pragma unroll
                        for (int u = 0; u < EV2G_STATS_TB; u++) x[u] = l_keep[(t - ta - NK + u) * 64 + (int)threadIdx.x];
                    } else {
#pragma unroll
                        for (int u = 0; u < EV2G_STATS_TB; u++) x[u] = slog[(long long)min(t + u, tend) * P];
                    }
#pragma unroll
                    for (int u = 0; u < EV2G_STATS_TB; u++)
                        if (t + u <= tend && __double_as_longlong(x[u]) >= 0) mad += fabs(avg_f - fabs(x[u]) * invB);
                }
                mad += fabs(avg_f - soc_f);
                const double delta_DoD = 2 * (mad / nf);
                const double T_sim = (td - ta + 1) * (double)s.dt / (60 * 24);
                const double v_avg = v_min + kk * avg_soc;
                const double alpha = (e0 * v_avg - e1) * k_arrh;
                deg_cal += alpha * 0.75 * T_sim / k_age;
                const double v_half = v_min + kk * 0.5;
                const double beta = z0 * (v_half - z1) * (v_half - z1) + z2 + z3 * delta_DoD;
                const double abs_e = live ? l_abs : fin_abs;
                const double Q_sim = (abs_e / b_cap_kwh) * b_cap_ah;
                deg_cyc += beta * 0.5 * Q_sim / k_qacc;
            }
        }
    }
    sum = seg_sum<EPWS>(sum); cnt = seg_sum<EPWS>(cnt); mn = seg_min<EPWS>(mn);
    deg_cal = seg_sum<EPWS>(deg_cal); deg_cyc = seg_sum<EPWS>(deg_cyc);
    double mean = NAN, sd = NAN, mnv = NAN;
    if (cnt > 0.0) {
        mean = sum / cnt;
        double var = 0.0;
        if (seg_all<EPWS>(nkeep <= 2)) {   // every lane kept all of its values (an env with at most 128 spawned sessions)
            if (nkeep > 0) { const double v = vkeep[0] - mean; var += v * v; }
            if (nkeep > 1) { const double v = vkeep[1] - mean; var += v * v; }
        } else {
            for (int k = d0 + lane; k < d1; k += W) {
                const int ta = s.ss_tarr[k];
                if (ta <= cur_step) {
                    const double capk = (s.ss_tdep[k] >= cur_step) ? st.line[(long long)e * P + s.ss_slot[k]].cap : st.sess_final_cap[k];
                    const double v = capk / ss_afap[k] * 100.0 - mean;
                    var += v * v;
                }
            }
        }
        var = seg_sum<EPWS>(var);
        sd = sqrt(var / cnt);
        mnv = mn;
    }
    if (lane == 0 && e_valid) {
        const double *acc = st.env_acc + (long long)e * 8;
        double *o = out + (long long)e * 17;
        o[0] = served;
        o[1] = acc[1];
        o[2] = acc[2];
        o[3] = acc[3];
        o[4] = (nsat > 0.0) ? sat / nsat : NAN;
        o[5] = ptv;
        o[6] = te;
        o[7] = ete;
        o[8] = mean;
        o[9] = sd;
        o[10] = mnv;
        o[11] = acc[4];
        o[12] = over;
        o[13] = log_soc ? deg_cal + deg_cyc : NAN;
        o[14] = log_soc ? deg_cal : NAN;
        o[15] = log_soc ? deg_cyc : NAN;
        o[16] = acc[0];
    }
    if (RESET && e_valid) {   // EV2Gym.reset()'s state-init part for this env, on scenario (e + reset_off) mod M (ev2g_reset_kernel, one env per segment)
        const int scn_n = ev2g_scn(e, reset_off, s.M);
        // the reset observation's head row (fast path: a copy of row 0 of the new scenario's head table) is requested before the state is
        // re-armed, together with the ports' first-session tables: one round trip for the block's loads, then only stores
        const bool hfast = s.head_tab != nullptr && s.state_kind != 1 && 2 + s.head_nh <= 2 * W && (r_obs || r_obs32);
        double hrow0 = 0.0, hrow1 = 0.0;
        if (hfast) {
            const double *row = s.head_tab + (long long)scn_n * (T + 1) * s.head_nh;
            hrow0 = row[min(max(lane - 2, 0), s.head_nh - 1)];
            hrow1 = row[min(lane + W - 2, s.head_nh - 1)];
        }
        for (int q = lane; q < P; q += W) {
            const long long g = (long long)e * P + q, gs = (long long)scn_n * P + q;
            const int2 w = s.port_first_win[gs];
            st.line[g].ta = w.x; st.line[g].td = w.y; st.line[g].ss = s.port_first[gs]; st.line[g].cyc_lut = 0;
            st.line[g].cap = 0.0; st.line[g].tot = 0.0; st.line[g].prev = 0.0;
            st.port_energy[g] = 0.0; st.port_current[g] = 0.0;
            if (r_obs) { double *o = r_obs + (long long)e * s.D + s.slot_obs[q]; o[0] = 0.0; o[1] = 0.0; if (s.state_kind == 1) o[2] = 0.0; }
            if (r_obs32) { float *o = r_obs32 + (long long)e * s.D + s.slot_obs[q]; o[0] = 0.f; o[1] = 0.f; if (s.state_kind == 1) o[2] = 0.f; }
        }
        for (int c = lane; c < C; c += W) {
            const long long gc = (long long)e * C + c;
            st.cs_sat_sum[gc] = 0.0; st.cs_served[gc] = 0;
            if (st.cs_profits) { st.cs_profits[gc] = 0.0; st.cs_e_ch[gc] = 0.0; st.cs_e_dis[gc] = 0.0; st.cs_power_now[gc] = 0.0; st.cs_cur_now[gc] = 0.0; }
        }
        for (int i = lane; i < 8; i += W) st.env_acc[(long long)e * 8 + i] = 0.0;
        // (history rows: only charge_power_potential[0], which the first step reads, is cleared -- rows the new episode has not reached are ignored by
        // the statistics kernel and by ev2g_peek; clearing all T x (2 + R) values of every env was a quarter of this block's bytes)
        if (lane == 0) st.hist[EV2G_HIST(e, 0, T, R) + 1] = 0.0;
        for (int r = lane; r < R; r += W) st.tr_power_now[(long long)e * R + r] = 0.0;
        if (lane == 0) st.env_fault[e] = 0;
        if (hfast) {   // columns 0, 1: step counter 0 and no usage yet; 2 ..: the head row
            const int c0 = lane, c1 = lane + W, nc = 2 + s.head_nh;
            const double v0 = (c0 < 2) ? 0.0 : hrow0;
            if (r_obs) { double *o = r_obs + (long long)e * s.D; if (c0 < nc) o[c0] = v0; if (c1 < nc) o[c1] = hrow1; }
            if (r_obs32) { float *o = r_obs32 + (long long)e * s.D; if (c0 < nc) o[c0] = (float)v0; if (c1 < nc) o[c1] = (float)hrow1; }
        } else {
            if (r_obs) write_obs_env(s, r_obs + (long long)e * s.D, scn_n, 0, 0.0, lane, W);
            if (r_obs32) write_obs_env(s, r_obs32 + (long long)e * s.D, scn_n, 0, 0.0, lane, W);
        }
    }
}
