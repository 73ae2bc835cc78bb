// ev2g_step_v2.h -- the production step kernel (P <= BLOCK ports per env; the generic kernel in
// ev2g_device.h covers larger envs).
//
// Per workgroup: G = BLOCK / P whole envs, one HOME lane per port; the port's dynamic state (window, session,
// capacity, energy counters) stays resident in LDS across the fused steps of one launch, so neither the home
// lanes nor the compacted worker lanes carry long-lived registers and the kernel fits 4 workgroups per CU.
// Per step:
//   A  home lanes: action -> charger-level amps (ev_charger.py:137-186); lanes whose EV really charges or
//      discharges append themselves to a compact work list in LDS (charge items grow from the front,
//      discharge items from the back: ~20 % of ports hold an EV, so the expensive float64 battery maths runs
//      on densely packed wavefronts instead of on every lane of every wavefront, and each wavefront takes
//      one branch);
//   B  worker lanes: one 128-byte session record (one cache line) + the staged state -> EV.step /
//      _charge / _discharge (ev.py:138-405) -> results back to LDS;
//   C  home lanes: departures, arrivals, per-port observation columns, action mask (ev_charger.py:209-229,
//      ev2gym_env.py:399-417,452-457, rl_agent/state.py);
//   D  LDS-staged segmented reduction: each (env, transformer) segment is summed by one wavefront laid out
//      as 8 quantities x 8 lanes, finished with 3 __shfl_xor steps (fixed tree => bit-reproducible);
//   E  per env: transformer overload, reward, histories, observation head and forecast / limit windows.
// Envs never communicate, so the K-step variant simply loops inside the workgroup (no grid sync).
#pragma once
#include "ev2g_device.h"

// Optional per-phase cycle accounting (tools/phase_timing.py builds a private .so with -DEV2G_PHASE_TIMING;
// the product library never has it).  dbg[block][phase] accumulates s_memtime ticks seen by wave 0 lane 0.
#ifdef EV2G_PHASE_TIMING
// slots 0..7: steps in which the workgroup had battery-maths items; 8..15: steps without any (empty nights, idle EVs);
// slot 16 / 17: number of such workgroup-steps
#define PT_DECL unsigned long long pt_last = __builtin_readcyclecounter(); unsigned long long pt_acc[18] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}; unsigned long long pt_cur[8] = {0,0,0,0,0,0,0,0};
#define PT_MARK(i) { unsigned long long n_ = __builtin_readcyclecounter(); pt_cur[i] += n_ - pt_last; pt_last = n_; }
#define PT_STEP_END(empty) { const int o_ = (empty) ? 8 : 0; for (int i_ = 0; i_ < 8; i_++) { pt_acc[o_ + i_] += pt_cur[i_]; pt_cur[i_] = 0; } pt_acc[16 + ((empty) ? 1 : 0)] += 1; }
#ifndef EV2G_PT_TID
#define EV2G_PT_TID 0   /* the lane whose clock is recorded (tools: -DEV2G_PT_TID=448 looks at wavefront 7 of a 512-thread workgroup) */
#endif
#define PT_FLUSH if (threadIdx.x == EV2G_PT_TID && S->dbg) { for (int i_ = 0; i_ < 8; i_++) pt_acc[i_] += pt_cur[i_]; for (int i_ = 0; i_ < 18; i_++) S->dbg[(size_t)blockIdx.x * 18 + i_] += pt_acc[i_]; }
#elif defined(EV2G_PHASE_MARKERS)   /* ISA analysis only: phase boundaries as comments in the -S output */
#define PT_DECL
#define PT_MARK(i) asm volatile("; PHASE_MARK " #i);
#define PT_STEP_END(empty)
#define PT_FLUSH
#else
#define PT_DECL
#define PT_MARK(i)
#define PT_STEP_END(empty)
#define PT_FLUSH
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, which would serialise
// every global prefetch issued at the top of a step against the first barrier; global memory is never used
// to communicate inside a workgroup here (only LDS is), so "s_waitcnt lgkmcnt(0); s_barrier" is sufficient.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct EvRes {
    double cap, prev_power, tot_e, energy, current;
    int cycles, emerg;
};

// EV.step + _charge/_discharge (ev.py:138-186, :240-355, :357-405) from one session record.
// Same operation order as the reference (and as oracle/ev2g_oracle.c); -ffp-contract=off.
// `lutv` is the efficiency-table entry for this step's current ALREADY divided by 100 (V2P::lut), looked up by the caller:
// dict.get(np.round(amps), 1) for charging, dict.get(abs(np.round(amps)), 1) for discharging (ev.py:287-290, :375-379).
__device__ __forceinline__ int ev_lut_index(int lut, double amps) {
    const double key = fabs(rint(amps));  // np.round = half-even; charging amps are positive
    return (key <= 100.0) ? lut * 101 + (int)key : -1;  // -1: key outside 0..100 -> the dict default 1
}
//
// Divisions are the expensive part (~14 instructions each).  Only rewrites that are exact for EVERY input are used:
//   * `previous_power / amps < 0` (ev.py:166) is a sign test: both operands are finite and non-zero here and the
//     quotient of values of these magnitudes cannot underflow to -0;
//   * `lutv` arrives already divided by 100 (the host divides the table once, same IEEE operation);
//   * `1 <= (pts - soc) / pilot_dsoc` (ev.py:318) is `pts - soc >= pilot_dsoc`: for y > 0 the correctly rounded
//     quotient is >= 1 exactly when x >= y (x < y gives x/y <= 1 - 2^-53, which rounds below 1);
//   * when 60/dt is a power of two (dt = 15, 30, 60 -- every shipped config) `x / (60/dt)` is `x * (dt/60)` and
//     `x / (dt/60)` is `x * (60/dt)`, bit for bit (`pow2_dt`, uniform);
//   * the final `ceil(cap*100)/100` goes through div_int_by_const (exhaustively verified range).
// The compacted lists separate charging from discharging items (each on its own wavefronts), so each kind has its own function and reads only
// its own chunks of the record (SessRec, ev2g_device.h).  `amps` is > 0 for ev_math_charge and < 0 for ev_math_discharge on entry.
// Divisions by the constants 1000 and 60 and by the per-session B and v go through their correctly rounded reciprocals (ev2g_fdiv1 / ev2g_fdiv2,
// ev2g_device.h: bit-identical to the IEEE division, 3 / 5 instructions instead of ~11); the two divisions by values of this very step
// (max_dsoc, pts - 1) are the hardware sequence.
__device__ __forceinline__ void ev_math_finish(EvRes &o, double tot_e) {
    o.prev_power = o.energy;
    o.tot_e = tot_e + o.energy;
    o.cap = ceil2_x(o.cap);
}
__device__ __forceinline__ EvRes ev_math_charge(const SessRec &r, double lutv, double amps, double cap, double prev_power, double tot_e, int cycles,
                                                double sixty_over_dt, double dt_over_60, bool pow2_dt, bool has_lut) {
    EvRes o;
    o.cap = cap; o.prev_power = prev_power; o.tot_e = tot_e; o.energy = 0.0; o.current = 0.0; o.cycles = cycles; o.emerg = 0;
    if (amps < r.gate_ch) return o;  // ev.py:151-163: below the EV's minimum power the step is a no-op (no ceil, previous_power untouched)
    if (prev_power == 0.0 || prev_power < 0.0) o.cycles = cycles + 1;   // previous_power / amps < 0 as a sign test (amps > 0)
    const double B = r.B, rB = r.rB, v = r.v, rv = r.rv;
    const double eta = has_lut ? lutv : r.eta_ch;
    const double pd0 = ev2g_fdiv2(ev2g_fdiv1(eta * amps * v, 1000.0, 1.0 / 1000.0), B, rB), md0 = ev2g_fdiv2(eta * r.pacmax, B, rB);
    double pilot_dsoc = pow2_dt ? pd0 * dt_over_60 : pd0 / sixty_over_dt;
    const double max_dsoc = pow2_dt ? md0 * dt_over_60 : md0 / sixty_over_dt;
    if (pilot_dsoc > max_dsoc) pilot_dsoc = max_dsoc;
    const double soc = ev2g_fdiv2(cap, B, rB);
    double curr_soc;
    if (r.ts == 1.0) {
        curr_soc = pilot_dsoc + soc;
        if (curr_soc > 1.0) curr_soc = 1.0;
    } else {
        const double pts = r.ts + (pilot_dsoc - max_dsoc) / max_dsoc * (r.ts - 1.0);
        // the two exponential branches of ev.py:318-334 share one exp(): per lane exactly one of them applies, the
        // operands of the selected one are evaluated in the reference's order, the other one costs nothing
        const bool below = soc < pts;
        const double num = below ? r.tsm * (pilot_dsoc + soc - pts) : r.tsm * pilot_dsoc;
        const double fac = below ? (pts - 1.0) : (soc - 1.0);
        double new_soc = 1.0 + exp(num / (pts - 1.0)) * fac;
        if (below && (pilot_dsoc > 0.0 ? (pts - soc >= pilot_dsoc) : (1.0 <= (pts - soc) / pilot_dsoc))) new_soc = pilot_dsoc + soc;
        const double lim = (max_dsoc > pilot_dsoc) ? pilot_dsoc : max_dsoc;
        curr_soc = (new_soc - soc > lim) ? (lim + soc) : new_soc;
    }
    const double dsoc = curr_soc - soc;
    o.cap = curr_soc * B;
    o.energy = dsoc * B;
    o.current = ev2g_fdiv2((pow2_dt ? o.energy * sixty_over_dt : o.energy / dt_over_60) * 1000.0, v, rv);
    ev_math_finish(o, tot_e);
    return o;
}
__device__ __forceinline__ EvRes ev_math_discharge(const SessRec &r, double lutv, double amps, double cap, double prev_power, double tot_e, int cycles,
                                                   double dt, bool has_lut, double rdt, bool dt_fdiv) {
    EvRes o;
    o.cap = cap; o.prev_power = prev_power; o.tot_e = tot_e; o.energy = 0.0; o.current = 0.0; o.cycles = cycles; o.emerg = 0;
    if (amps > r.gate_dis) return o;  // ev.py:153-163
    if (prev_power == 0.0 || prev_power > 0.0) o.cycles = cycles + 1;   // previous_power / amps < 0 as a sign test (amps < 0)
    const double v = r.v, rv = r.rv;
    double given_power = ev2g_fdiv1(amps * v, 1000.0, 1.0 / 1000.0);
    if (fabs(given_power) > fabs(r.pdismax)) given_power = r.pdismax;
    const double eta = has_lut ? lutv : r.eta_dis;
    double given_energy = ev2g_fdiv1(given_power * eta * dt, 60.0, 1.0 / 60.0);
    if (cap + given_energy < r.minB) {
        if (cap > r.minB) { o.energy = -(cap - r.minB); given_energy = o.energy; }
        else { o.energy = 0.0; given_energy = 0.0; }
        o.cap = r.minB;
    } else {
        o.energy = given_energy;
        o.cap = cap + given_energy;
    }
    if (cap > r.emerg && o.cap < r.emerg) o.emerg = 1;
    const double e60 = given_energy * 60.0;
    o.current = ev2g_fdiv2((dt_fdiv ? ev2g_fdiv1(e60, dt, rdt) : e60 / dt) * 1000.0, v, rv);
    ev_math_finish(o, tot_e);
    return o;
}

// Pointer members carry the global address space explicitly, so that loads through a struct that itself lives in
// device memory still compile to global_load / s_load (a plain `T*` read from memory would be a flat pointer, and
// flat accesses tick lgkmcnt, which the LDS-only barriers wait on).
#if defined(__HIP_DEVICE_COMPILE__)
#define EV2G_GP(T) T __attribute__((address_space(1))) *
#else
#define EV2G_GP(T) T *
#endif
#define EV2G_SETP(dst, src) dst = (decltype(dst))(unsigned long long)(src)

// Everything ev2g_step_v2 reads, resident in device memory (uploaded once per ev2g_load_scenarios).  The kernel
// reads it through the scalar cache INSIDE the step loop (opaque pointer per iteration): passing these ~70 values
// as by-value kernel arguments made LLVM hoist all of them above the loop and spill >200 SGPRs to VGPR lanes,
// which was 40 % of the VALU instruction stream.
struct V2P {
    int E, M, T, C, npc, P, R, D, G, dt, reward_kind, state_kind, cost_kind, n_lut, pow2_dt;
    int dt_fdiv;   // |dt*RN(1/dt) - 1| <= 2^-54: x / dt may go through ev2g_fdiv1 (ev2g_device.h)
    double sixty_over_dt, dt_over_60, rdt;
    EV2G_GP(const int) slot_cs; EV2G_GP(const int) slot_port; EV2G_GP(const int) slot_obs; EV2G_GP(const int) cs_slot0;
    EV2G_GP(const int) tr_seg; EV2G_GP(const int) tr_obs; EV2G_GP(const int) port_first;
    EV2G_GP(const int2) port_first_win;
    EV2G_GP(const double) cs_imax; EV2G_GP(const double) cs_imin; EV2G_GP(const double) cs_dmin;
    EV2G_GP(const double) cs_dmax_abs; EV2G_GP(const double) cs_maxp; EV2G_GP(const double) cs_minp;
    EV2G_GP(const double) price_ch; EV2G_GP(const double) price_dis; EV2G_GP(const double) setpoint;
    EV2G_GP(const double) tr_infl; EV2G_GP(const double) tr_solar; EV2G_GP(const double) tr_base; EV2G_GP(const double) tr_maxp;
    EV2G_GP(const double) tr_minp; EV2G_GP(const double) win_tab; EV2G_GP(const double) lut;
    EV2G_GP(const double) step_tab;   // [E, T, 8] per (env, step) scalars (fast path only, ev2g_build_step_table_kernel)
    EV2G_GP(char) slab_port; unsigned long long slab_port_slice;   // DevState slabs (ev2g_device.h)
    EV2G_GP(double) slab_hist; EV2G_GP(double) slab_sess; unsigned long long sess_slice;   // bytes
    EV2G_GP(const double) head_tab;   // [E, T+1, NH] observation head rows (fast path only, ev2g_build_head_table_kernel)
    EV2G_GP(const SessRec) rec; EV2G_GP(const SessTail) tail; EV2G_GP(const int) ss_lut;
    EV2G_GP(const SessDyn) sess_dyn; EV2G_GP(const ClsRec) cls_rec;   // fast path, round 5 (ev2g_device.h)
    EV2G_GP(PortLine) line;
    EV2G_GP(double) cs_sat_sum; EV2G_GP(int) cs_served;
    EV2G_GP(double) cs_profits; EV2G_GP(double) cs_e_ch; EV2G_GP(double) cs_e_dis;
    EV2G_GP(double) cs_power_hist; EV2G_GP(double) cs_cur_hist; EV2G_GP(double) cs_power_now; EV2G_GP(double) cs_cur_now;
    EV2G_GP(double) env_acc; EV2G_GP(int) env_fault;
    EV2G_GP(double) hist; EV2G_GP(double) tr_power_now;
    EV2G_GP(double) sess_final_cap; EV2G_GP(double) port_energy; EV2G_GP(double) port_current;
    EV2G_GP(double) soc_log; EV2G_GP(double) sess_abs_e;
    EV2G_GP(unsigned long long) dbg;
    // StepExtras (ev2g_set_step_extras), refreshed in place when they change
    EV2G_GP(double) x_cost; long long x_c_stride;
    EV2G_GP(float) x_obs32; long long x_o32_stride;
    EV2G_GP(const float) x_act32;
};

inline void ev2g_v2_fill_params(V2P &p, const DevScn &s, const DevState &st) {
    p.E = s.E; p.M = s.M; p.cost_kind = s.cost_kind; p.T = s.T; p.C = s.C; p.npc = s.npc; p.P = s.P; p.R = s.R; p.D = s.D; p.G = s.G; p.dt = s.dt;
    p.reward_kind = s.reward_kind; p.state_kind = s.state_kind; p.n_lut = s.n_lut;
    p.sixty_over_dt = s.sixty_over_dt; p.dt_over_60 = s.dt_over_60;
    p.rdt = 1.0 / (double)s.dt; p.dt_fdiv = fabs(fma((double)s.dt, p.rdt, -1.0)) <= 0x1p-54 ? 1 : 0;   // (the fma's result is exact: |dt*rdt - 1| < 2^-52 has at most 53 significant bits)
    EV2G_SETP(p.x_cost, (double *)nullptr); EV2G_SETP(p.x_obs32, (float *)nullptr); EV2G_SETP(p.x_act32, (const float *)nullptr);
    p.x_c_stride = 0; p.x_o32_stride = 0;
    p.pow2_dt = 0; EV2G_SETP(p.head_tab, (const double *)nullptr);
    EV2G_SETP(p.step_tab, (const double *)nullptr);
    EV2G_SETP(p.slab_port, st.slab_port); p.slab_port_slice = st.slab_port_slice;
    EV2G_SETP(p.slab_hist, st.slab_hist); EV2G_SETP(p.slab_sess, st.slab_sess);
    p.sess_slice = (unsigned long long)(st.sess_abs_e ? (st.sess_abs_e - st.slab_sess) : 0) * 8ull;
#define CPS(f) EV2G_SETP(p.f, s.f);
#define CPT(f) EV2G_SETP(p.f, st.f);
    CPS(slot_cs) CPS(slot_port) CPS(slot_obs) CPS(cs_slot0) CPS(tr_seg) CPS(tr_obs) CPS(port_first) CPS(port_first_win)
    CPS(cs_imax) CPS(cs_imin) CPS(cs_dmin) CPS(cs_dmax_abs) CPS(cs_maxp) CPS(cs_minp)
    CPS(price_ch) CPS(price_dis) CPS(setpoint) CPS(tr_infl) CPS(tr_solar) CPS(tr_base) CPS(tr_maxp) CPS(tr_minp)
    CPS(win_tab) CPS(lut) CPS(rec) CPS(tail) CPS(ss_lut) CPS(sess_dyn) CPS(cls_rec)
    CPT(line) CPT(cs_sat_sum) CPT(cs_served)
    CPT(cs_profits) CPT(cs_e_ch) CPT(cs_e_dis) CPT(cs_power_hist) CPT(cs_cur_hist) CPT(cs_power_now) CPT(cs_cur_now)
    CPT(env_acc) CPT(env_fault) CPT(hist) CPT(tr_power_now)
    CPT(sess_final_cap) CPT(port_energy) CPT(port_current) CPT(soc_log) CPT(sess_abs_e) CPT(dbg)
#undef CPS
#undef CPT
}

// LDS carve-up for ev2g_step_v2 (doubles first, then ints); NS = G*P, NT = G*R
__host__ __device__ inline size_t ev2g_v2_lds_bytes(int NS, int NT, int G, int R) {
    return sizeof(double) * ((size_t)(EV2G_NQ + 7) * NS + (size_t)EV2G_NQ * NT + (size_t)EV2G_NQ * G + (size_t)NT +
                             (size_t)G * 9) +
           sizeof(int) * (6 * (size_t)NS + 2 * (size_t)R + 1 + 4);
}

// SPEC = 1: the instantiation for the reference's default plugin pair on big envs (BASELINE configs[3]: V2G_profit_max_loads state,
// ProfitMax_TrPenalty_UserIncentives reward, single-port chargers) launched the way a loop that consumes the outputs -- or the benchmark --
// launches it: float64 actions, all four float64 outputs with step stride 0, no extras, no charger histories, SoC log on, the launch ends
// within the episode, 15 / 30 / 60-minute steps.  What the general instantiation decides per step with uniform branches and parameter-block
// fetches (plugin kinds, null checks, ports per charger, the in-launch reset) is a compile-time constant here: measured 65.1 -> 58.6 us/step
// at cfg4 (0.359 -> 0.399), same results bit for bit (tests/test_round3_gpu.py).  V2C(run-time expression, its value under SPEC).
#define V2C(x, c) (SPEC ? (c) : (x))
template <int BLOCK, int SPEC = 0>
__global__ void __launch_bounds__(BLOCK, 4) ev2g_step_v2(const V2P *__restrict__ params, StepIO io, int t0,
                                                         int k_steps, int auto_reset) {
    extern __shared__ double lds[];
    typedef const V2P __attribute__((address_space(4))) *ParamPtr;  // constant address space: scalar loads
    ParamPtr S = (ParamPtr)(unsigned long long)params;
    const int P = S->P, R = S->R, T = S->T, C = S->C, npc = V2C(S->npc, 1), E = S->E, D = S->D, G = S->G, M = S->M;
    int off = io.scn_off;   // scenario-pool window: env e runs scenario (e + off) mod M
    int grp;
    {   // XCD-aware mapping: workgroup b runs on XCD b % 8; give each XCD a contiguous range of env groups
        const int nb = gridDim.x, b = blockIdx.x, per = nb >> 3;
        grp = (nb & 7) == 0 ? (b & 7) * per + (b >> 3) : b;
    }
    const int e0 = grp * G;
    const int ne = min(G, E - e0);
    const int N = ne * P, NS = G * P, NT = G * R;
    // ---- LDS: per-port results, resident port state, reduction scratch ----
    double *stage = lds;                                   // [NQ][NS] per-port step results, by home index
    double *s_cap = stage + (size_t)EV2G_NQ * NS;          // EV.current_capacity
    double *s_tot = s_cap + NS, *s_prev = s_tot + NS;      // total_energy_exchanged, previous_power
    double *s_bcap = s_prev + NS, *s_potc = s_bcap + NS;   // battery_capacity, charge-power-potential term
    double *s_amps = s_potc + NS;                          // phase A: amps; phase B: replaced by EV.current_energy
    double *s_abse = s_amps + NS;                          // EV.abs_total_energy_exchanged (EV2G_FLAG_LOG_SOC)
    double *tsum = s_abse + NS;                            // [NQ][NT]
    double *esum = tsum + (size_t)EV2G_NQ * NT;            // [NQ][G]
    double *over_l = esum + (size_t)EV2G_NQ * G;           // [NT] 100 * overload of each (env, transformer)
    double *eacc = over_l + NT;                            // [G][5] episode accumulators
    double *pot_prev = eacc + (size_t)G * 5;               // [G] charge_power_potential[t]
    double *osum = pot_prev + G;                           // [G] sum of 100 * overload over the env's transformers
    double *pot_prev2 = osum + G;                          // [G] charge_power_potential[t-1] (SquaredTrackingErrorRewardWithPenalty)
    double *tr0max = pot_prev2 + G;                        // [G] transformers[0].max_power[t] (SqTrError_TrPenalty_UserIncentives)
    int *s_ta = (int *)(tr0max + G);                   // window {t_arr, t_dep} of the attached-or-next session
    int *s_td = s_ta + NS, *s_ss = s_td + NS, *s_cyc = s_ss + NS;  // session index, charging_cycles
    int *s_dirty = s_cyc + NS;                             // bit0: cap/tot/prev/cycles changed, bit1: window changed
    int *items = s_dirty + NS, *seg = items + NS, *trobs = seg + R + 1, *cnt = trobs + R;  // cnt[0] charge, cnt[1] discharge
    const int tid = threadIdx.x;
    const bool log_cs = V2C(S->cs_profits != nullptr, false);
    const bool log_soc = V2C(S->soc_log != nullptr, true);
    const double dtd = (double)S->dt, sixty_over_dt = S->sixty_over_dt, dt_over_60 = S->dt_over_60;
    const bool pow2_dt = V2C(S->pow2_dt != 0, true);

    // ---- home lane set-up (once per launch): global state -> LDS ----
    const bool valid = tid < N;
    const int el = valid ? tid / P : 0;
    const int q = valid ? tid - el * P : 0;
    const int e = e0 + el;
    const int g = e * P + q;  // 32-bit element offsets: the engine requires E*P, E*D, E*R*T < 2^31
    const int cs = S->slot_cs[q], pref = S->slot_port[q], ocol = S->slot_obs[q];
    // charger constants of this lane's port (ev_charger.py:41-94), live in registers for the whole launch
    const double c_imax = S->cs_imax[cs], c_thr_ch = S->cs_imin[cs] - 0.01, c_dmin = S->cs_dmin[cs], c_dmaxabs = S->cs_dmax_abs[cs];
    const double c_maxp = S->cs_maxp[cs], c_minp = S->cs_minp[cs];
    int t = t0;
    if (valid) {
        const EV2G_GP(PortLine) ln = S->line + g;   // this port's 64-byte state line (ev2g_device.h)
        const int2 w = make_int2(ln->ta, ln->td);
        s_ta[tid] = w.x; s_td[tid] = w.y; s_ss[tid] = ln->ss; s_cyc[tid] = ev2g_line_cycles(ln->cyc_lut); s_dirty[tid] = 0;
        if (w.x <= t && t <= w.y) {
            s_cap[tid] = ln->cap; s_tot[tid] = ln->tot; s_prev[tid] = ln->prev;
            s_bcap[tid] = ln->bcap; s_potc[tid] = ln->potc;
            s_abse[tid] = log_soc ? ln->abse : 0.0;
        } else {
            s_abse[tid] = 0.0;
            s_cap[tid] = 0.0; s_tot[tid] = 0.0; s_prev[tid] = 0.0; s_bcap[tid] = 1.0; s_potc[tid] = 0.0;
        }
    }
    // env-level lanes: thread i < ne*R owns (env, transformer) pair i; thread pel*lpe owns env pel
    const int lpe = BLOCK / ne;
    const int pel = tid / lpe, pl = tid - pel * lpe;
    const bool env_lane = pel < ne;
    const int pe = e0 + (env_lane ? pel : 0);
    if (tid < 2) cnt[tid] = 0;
    for (int i = tid; i <= R; i += BLOCK) seg[i] = S->tr_seg[i];
    for (int i = tid; i < R; i += BLOCK) trobs[i] = S->tr_obs[i];
    for (int i = tid; i < ne * 5; i += BLOCK) eacc[i] = 0.0;
    for (int i = tid; i < ne; i += BLOCK) {
        pot_prev[i] = (t < T) ? S->hist[EV2G_HIST(e0 + i, t, T, R) + 1] : 0.0;
        pot_prev2[i] = (t > 0 && t <= T) ? S->hist[EV2G_HIST(e0 + i, t - 1, T, R) + 1] : 0.0;
    }
    // observation-head role of this lane (columns pl and pl + lpe of the env it serves at env level), fixed for the launch:
    // destination column, and where the value comes from -- charge price `hsrc` steps ahead (hsrc < 20), or entry hsrc - 20 of the
    // env's window table row block (then + sstep*40 per step); hdst < 0: no column
    int hdst0 = -1, hsrc0 = 0, hdst1 = -1, hsrc1 = 0;
    {
        const int nhead0 = (V2C(S->state_kind, 0) == 1) ? 0 : 20 + ((V2C(S->state_kind, 0) == 0) ? 40 * R : 0);
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int c = pl + u * lpe;
            int dst = -1, src = 0;
            if (c < 20 && c < nhead0) { dst = 2 + c; src = c; }
            else if (c < nhead0) { const int i = c - 20, r = i / 40, j = i - r * 40; dst = S->tr_obs[r] + j; src = 20 + r * (T + 1) * 40 + j; }
            if (u == 0) { hdst0 = dst; hsrc0 = src; } else { hdst1 = dst; hsrc1 = src; }
        }
    }
    const float *act32 = (const float *)S->x_act32;   // float32 actions (StepExtras), used when io.actions is null
    const long long a_base = io.actions ? 0 : (long long)io.step0 * io.a_stride;
    double a_next = SPEC ? io.actions[valid ? e * P + pref : e0 * P] : ev2g_action(io, act32, a_base, valid ? e * P + pref : e0 * P);
    __syncthreads();
    // SPEC: the rows this lane reads every step (its transformer's series, its envs' prices / setpoints) do not move during the launch
    int h_erT0 = 0, h_pec = 0, h_pecT = 0, h_evcT = 0;
    if (SPEC) {
        const int trl = min(tid, ne * R - 1), trl_e = trl / R;
        h_erT0 = (ev2g_scn(e0 + trl_e, off, M) * R + (trl - trl_e * R)) * T;
        h_pec = ev2g_scn(e0 + min(pel, ne - 1), off, M);
        h_pecT = h_pec * T;
        h_evcT = ev2g_scn(valid ? e : e0, off, M) * T;
    }

    PT_DECL
    for (int kk = 0; kk < k_steps; kk++) {
        PT_MARK(7)
        // Defeat loop-invariant hoisting of per-lane addresses: LLVM would otherwise precompute ~60 LDS / global
        // addresses before the step loop and keep them alive across it (they end up in scratch).  Re-deriving an
        // address costs one or two VALU ops per use; the indices below are opaque to the optimiser per iteration.
        asm volatile("" : "+s"(S));  // parameters are (re)loaded through the scalar cache where they are used
        int tid_l = tid, g_l = g, e_l = e, cs_l = cs, pref_l = pref, ocol_l = ocol, pe_l = pe, pl_l = pl, pel_l = pel;
        asm volatile("" : "+v"(tid_l), "+v"(g_l), "+v"(e_l), "+v"(cs_l), "+v"(pref_l), "+v"(ocol_l), "+v"(pe_l), "+v"(pl_l), "+v"(pel_l));
        int hdst0_l = hdst0, hsrc0_l = hsrc0, hdst1_l = hdst1, hsrc1_l = hsrc1;
        asm volatile("" : "+v"(hdst0_l), "+v"(hsrc0_l), "+v"(hdst1_l), "+v"(hsrc1_l));
        if (V2C(t >= T, false)) {  // episode finished inside a fused run: in-kernel ev2g_reset for this workgroup
            if (!auto_reset) break;
            off = ev2g_scn(off, io.scn_stride, M);
            if (valid) {
                const int gs = ev2g_scn(e_l, off, M) * P + (g_l - e_l * P);
                const int2 w = S->port_first_win[gs];
                s_ta[tid_l] = w.x; s_td[tid_l] = w.y; s_ss[tid_l] = S->port_first[gs]; s_cyc[tid_l] = 0;
                s_cap[tid_l] = 0.0; s_tot[tid_l] = 0.0; s_prev[tid_l] = 0.0; s_abse[tid_l] = 0.0; s_dirty[tid_l] = 3;
                S->port_energy[g_l] = 0.0;
                S->port_current[g_l] = 0.0;
            }
            for (int i = tid_l; i < ne * C; i += BLOCK) {
                const int gc = e0 * C + i;
                S->cs_sat_sum[gc] = 0.0;
                S->cs_served[gc] = 0;
                if (log_cs) { S->cs_profits[gc] = 0.0; S->cs_e_ch[gc] = 0.0; S->cs_e_dis[gc] = 0.0; }
            }
            for (int i = tid_l; i < ne * 8; i += BLOCK) S->env_acc[e0 * 8 + i] = 0.0;
            for (int i = tid_l; i < ne * 5; i += BLOCK) eacc[i] = 0.0;
            for (int i = tid_l; i < ne; i += BLOCK) { pot_prev[i] = 0.0; pot_prev2[i] = 0.0; }
            t = 0;
            lds_barrier();
        }
        double *__restrict__ obs = SPEC ? io.obs : (io.obs ? io.obs + (long long)kk * io.o_stride : nullptr);
        float *__restrict__ obs32 = V2C(S->x_obs32 != nullptr, false) ? (float *)S->x_obs32 + (long long)(io.step0 + kk) * S->x_o32_stride : nullptr;
        uint8_t *__restrict__ mask = SPEC ? io.mask : (io.mask ? io.mask + (long long)kk * io.m_stride : nullptr);
        const int sstep = t + 1;
        const bool last_step = (kk == k_steps - 1) || (!SPEC && sstep >= T && !auto_reset);

        // ---------------- A: home lanes, charger level (ev_charger.py:137-186) ----------------
        bool occ = false;
        double cap_before = 0.0;
        double amps = 0.0;
        if (valid) {
            const int ta = s_ta[tid_l], td = s_td[tid_l];
            occ = (ta <= t) && (t <= td);
            if (log_soc && occ) cap_before = s_cap[tid_l];
            double a = occ ? a_next : 0.0;
            if (npc == 1) {   // one port per charger: a / sum(a) = a / a and -a / a, exactly +-1 for every finite action (ev_charger.py:143-149)
                if (a > 1.0) a = 1.0;
                else if (a < -1.0) a = -1.0;
            } else {
                const long long a_off = a_base + (long long)kk * io.a_stride;
                const int j0 = tid_l - (pref_l - cs_l * npc);
                double Ssum = 0.0;
                for (int j = 0; j < npc; j++) {  // sequential python sum() over the charger's ports
                    const bool oj = (s_ta[j0 + j] <= t) && (t <= s_td[j0 + j]);
                    Ssum = Ssum + (oj ? ev2g_action(io, act32, a_off, e_l * P + cs_l * npc + j) : 0.0);
                }
                if (Ssum > 1.0) a = a / Ssum;
                else if (Ssum < -1.0) a = -a / Ssum;
            }
            if (occ) {
                const double x = rnd5_x(a);
                if (x > 0.0) { amps = x * c_imax; if (amps < c_thr_ch) amps = 0.0; }
                else if (x < 0.0) { amps = x * c_dmaxabs; if (amps > c_dmin - 0.01) amps = c_dmin; }
            }
            s_amps[tid_l] = amps;
            stage[0 * NS + tid_l] = 0.0;
            stage[4 * NS + tid_l] = 0.0;
            stage[5 * NS + tid_l] = 0.0;
            stage[6 * NS + tid_l] = 0.0;
            stage[7 * NS + tid_l] = 0.0;
        }
        {   // compact the ports that have battery maths to do: charging items from the front of `items`, discharging ones from its
            // back.  One LDS atomic per wavefront and list (ballot + lane prefix count); an atomic per item serialises on the
            // two counters -- ~200 same-address atomics per step at 1000 ports.
            const unsigned long long mch = __ballot(amps > 0.0), mdis = __ballot(amps < 0.0);
            const int lane = tid & 63;
            int bch = 0, bdis = 0;
            if (lane == 0) {
                if (mch) bch = atomicAdd(&cnt[0], __popcll(mch));
                if (mdis) bdis = atomicAdd(&cnt[1], __popcll(mdis));
            }
            bch = __shfl(bch, 0, 64);
            bdis = __shfl(bdis, 0, 64);
            if (amps > 0.0) items[bch + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mch >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mch, 0u))] = tid_l;
            else if (amps < 0.0) items[NS - 1 - (bdis + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mdis >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mdis, 0u)))] = tid_l;
        }
        // ---- prefetch what phases C-E of this step need (issued AFTER phase A consumed its own operands, so that
        //      phase A never waits on them; the loads stay in flight across the LDS-only barriers) ----
        {
            const bool more = (kk + 1 < k_steps) && (sstep < T || auto_reset);
            a_next = SPEC ? io.actions[(long long)(more ? kk + 1 : kk) * io.a_stride + (valid ? e_l * P + pref_l : e0 * P)]
                          : ev2g_action(io, act32, a_base + (long long)(more ? kk + 1 : kk) * io.a_stride, valid ? e_l * P + pref_l : e0 * P);
        }
        // Every prefetch is ONE unconditional load from a clamped (always valid) address; the conditions are applied where
        // the value is consumed.  A load in a divergent branch whose result merges with a default makes the compiler
        // serialise on the destination register (write-after-write) with a full vmcnt(0) drain.
        int erT;
        if (SPEC) erT = h_erT0 + t;   // (no in-launch reset: this lane's rows do not move)
        else {
            const int trl = min(tid_l, ne * R - 1), trl_e = trl / R;   // this lane's (env, transformer) role
            erT = (ev2g_scn(e0 + trl_e, off, M) * R + (trl - trl_e * R)) * T + t;
        }
        const double pf_infl = S->tr_infl[erT], pf_solar = S->tr_solar[erT], pf_maxp = S->tr_maxp[erT], pf_minp = S->tr_minp[erT];
        const int pec = SPEC ? h_pec : ev2g_scn(e0 + min(pel_l, ne - 1), off, M);   // scenario of the (clamped) env of this lane's env-level role
        const int pecT = SPEC ? h_pecT : pec * T;
        const double pf_sp = S->setpoint[pecT + t];
        const int evcT = SPEC ? h_evcT : ev2g_scn(valid ? e_l : e0, off, M) * T;   // scenario of the (clamped) env of this lane's home role
        const double pf_pch = S->price_ch[evcT + t], pf_pdis = S->price_dis[evcT + t];
        // head / window columns of the observation this step emits (step counter sstep): one coalesced load per lane
        double pf_ob0 = 0.0, pf_ob1 = 0.0;
        const int nhead = (V2C(S->state_kind, 0) == 1) ? 0 : 20 + ((V2C(S->state_kind, 0) == 0) ? 40 * R : 0);
        if (V2C(S->state_kind, 0) == 1) {
            pf_ob0 = S->setpoint[pecT + min(sstep, T - 1)];   // consumed by pl == 0, masked by sstep < T
        } else {
            const double *pprice = (const double *)S->price_ch + pecT;
            const double *pwin = (const double *)S->win_tab + ((long long)pec * R * (T + 1) + sstep) * 40 - 20;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int src = u ? hsrc1_l : hsrc0_l;
                const double *pa = (src < 20) ? pprice + min(sstep + src, T - 1) : pwin + src;
                const double v = *pa;
                if (u == 0) pf_ob0 = v; else pf_ob1 = v;
            }
        }

        PT_MARK(0)
        lds_barrier();
        PT_MARK(1)

        // ---------------- B: worker lanes, battery maths on the compact list ----------------
        __builtin_amdgcn_s_setprio(3);   // the wavefronts holding list items are the workgroup's critical path
        {
            const int nch = cnt[0], ndis = cnt[1];
            const int nchp = (nch + 63) & ~63;  // discharge items start on a wavefront boundary
            for (int i = tid_l; i < nchp + ndis; i += BLOCK) {
                int h = -1;
                if (i < nch) h = items[i];
                else if (i >= nchp) h = items[NS - 1 - (i - nchp)];
                if (h >= 0) {
                    // the chunks of the record this wavefront's kind of step reads (charging: 0..4, discharging: 3..6 -- `i < nchp` is uniform: the
                    // discharge items start on a wavefront boundary) + the efficiency-table id, issued together
                    typedef double d2_t __attribute__((ext_vector_type(2)));
                    const int ssh = s_ss[h];
                    const char __attribute__((address_space(1))) *rp = (const char __attribute__((address_space(1))) *)(S->rec + ssh);
                    union { SessRec r; d2_t v[8]; } u;
                    static_assert(sizeof(SessRec) == 128 && offsetof(SessRec, rB) == 48 && offsetof(SessRec, minB) == 80 && offsetof(SessRec, cap0) == 112, "record layout");
                    const int r_lut = S->ss_lut[ssh];
                    const double cap0 = s_cap[h], prev0 = s_prev[h];
                    const int cyc0 = s_cyc[h];
                    const double amps_h = s_amps[h];
                    double lutv = 1.0 / 100.0;
                    EvRes o;
                    if (i < nchp) {   // (uniform)
#pragma unroll
                        for (int c = 0; c < 5; c++) u.v[c] = *(const d2_t __attribute__((address_space(1))) *)(rp + 16 * c);
                        if (r_lut >= 0) { const int li = ev_lut_index(r_lut, amps_h); if (li >= 0) lutv = S->lut[li]; }
                        o = ev_math_charge(u.r, lutv, amps_h, cap0, prev0, s_tot[h], cyc0, sixty_over_dt, dt_over_60, pow2_dt, r_lut >= 0);
                    } else {
#pragma unroll
                        for (int c = 3; c < 7; c++) u.v[c] = *(const d2_t __attribute__((address_space(1))) *)(rp + 16 * c);
                        if (r_lut >= 0) { const int li = ev_lut_index(r_lut, amps_h); if (li >= 0) lutv = S->lut[li]; }
                        o = ev_math_discharge(u.r, lutv, amps_h, cap0, prev0, s_tot[h], cyc0, dtd, r_lut >= 0, S->rdt, S->dt_fdiv != 0);
                    }
                    if (o.cycles != cyc0 || o.energy != 0.0 || o.cap != cap0 || o.prev_power != prev0) s_dirty[h] |= 1;
                    s_cap[h] = o.cap;
                    s_prev[h] = o.prev_power;
                    s_tot[h] = o.tot_e;
                    s_cyc[h] = o.cycles;
                    s_amps[h] = o.energy;
                    if (log_soc) s_abse[h] += fabs(o.energy);
                    stage[0 * NS + h] = o.energy * 60.0 / dtd;
                    stage[(i < nch ? 4 : 5) * NS + h] = fabs(o.energy);
                    stage[6 * NS + h] = (double)o.emerg;
                    stage[7 * NS + h] = o.current;
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        // Departures and arrivals are known before the step (occupancy does not depend on the actions): the session-record fields
        // phase C needs are requested here, behind this wavefront's battery maths (the register file has no room to hold them across it) and before the
        // barrier -- wavefronts without items overlap them with the others' maths -- instead of dependently inside that phase's branches.
        // A lane has at most one of the two events: an arrival takes {B, cap0, potc} from the record, a departure {des, next window} from the
        // session's tail entry -- three 8-byte loads, issued only by wavefront-steps that have such an event; the other lanes of such a
        // wavefront read session 0.
        double pf_ra = 0.0, pf_rb = 0.0, pf_rc = 0.0;
        const bool ev_dep = occ && t >= s_td[valid ? tid_l : 0], ev_arr = valid && (s_ta[tid_l] == sstep);   // (the battery maths does not touch the windows)
        if (__ballot(ev_dep || ev_arr) != 0ull) {   // (uniform)
            const int sse = (ev_dep || ev_arr) ? s_ss[tid_l] : 0;
            const char *rp = (const char *)((const SessRec *)S->rec + sse), *tp = (const char *)((const SessTail *)S->tail + sse);
            pf_ra = *(const double *)(ev_arr ? rp + offsetof(SessRec, B) : tp + offsetof(SessTail, des));
            pf_rb = *(const double *)(ev_arr ? rp + offsetof(SessRec, cap0) : tp + offsetof(SessTail, nt_arr));   // (the window: two ints)
            pf_rc = *(const double *)(rp + offsetof(SessRec, potc));
        }
        PT_MARK(2)
        lds_barrier();
        PT_MARK(1)
        if (tid_l < 2) cnt[tid_l] = 0;

        // ---------------- C: home lanes: departures, arrivals, observation columns ----------------
        // All prefetches (issued at the end of phase A, one battery-maths phase ago) are collected HERE, before this
        // phase issues its global stores: on gfx9-family ISAs vmcnt counts loads and stores together and they
        // retire out of order with respect to each other, so a load consumed while younger stores are pending costs
        // a full vmcnt(0) drain of those stores.  s_waitcnt vmcnt(0) expcnt(7) lgkmcnt(15):
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (valid) {
            double profit = 0.0, satpen = 0.0, pot = 0.0;
            int ta = s_ta[tid_l], td = s_td[tid_l];
            double cap = s_cap[tid_l];
            bool departed = false;
            if (occ) {
                const double energy = s_amps[tid_l];  // 0 for idle EVs (phase A stored amps == 0)
                const double current = stage[7 * NS + tid_l];
                if (energy != 0.0) {  // profit += |E| * price, by the sign of the ACTION (ev_charger.py:178,194): the
                    // worker staged |E| under "charged" (4) or "discharged" (5); a charge step can return a tiny
                    // negative energy when ceil2 left the capacity above the battery size
                    const double ech = stage[4 * NS + tid_l];
                    profit = (ech != 0.0) ? ech * pf_pch : stage[5 * NS + tid_l] * pf_pdis;
                }
                if (npc == 1 && current - 0.0001 > c_imax) S->env_fault[e_l] = 1;  // ev_charger.py:203-205
                if (last_step) { S->port_energy[g_l] = energy; S->port_current[g_l] = current; }
                if (log_soc)  // historic_soc / active_steps (ev.py:156,162,185): capacity before the step, negated if inactive
                    S->soc_log[((long long)e_l * T + t) * P + (g_l - e_l * P)] = (current != 0.0) ? cap_before : -cap_before;
                if (t >= td) {  // departure (ev_charger.py:209-229, ev.py:191-214)
                    const int ss = s_ss[tid_l];
                    const double des = pf_ra;
                    const double score = (cap < des - 0.001) ? cap / des : 1.0;
                    satpen = ev2g_departure_term(V2C(S->reward_kind, 0), V2C(S->cost_kind, 0), score, cap, des);
                    const int gc = e_l * C + cs_l;
                    // fire-and-forget device atomics (no returned value => no memory round trip on this path)
                    __hip_atomic_fetch_add(&S->cs_served[gc], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(&S->cs_sat_sum[gc], score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    S->sess_final_cap[ss] = cap;
                    if (log_soc) S->sess_abs_e[ss] = s_abse[tid_l];
                    ta = __double2loint(pf_rb); td = __double2hiint(pf_rb);   // window of the port's next session
                    departed = true;
                    s_ta[tid_l] = ta; s_td[tid_l] = td;
                    s_ss[tid_l] = (ta != EV2G_INT_MAX) ? ss + 1 : -1;
                    s_cyc[tid_l] = 0;
                    s_dirty[tid_l] |= 2;
                }
            }
            if (ta == sstep) {  // arrival at the end of this step (ev2gym_env.py:399-417, ev.py:115-136)
                if (departed) {   // the next session arrives right behind a departure of this very step (the reference's spawner leaves a
                                  // gap, replayed scenarios need not): its record was not the one prefetched
                    const SessRec &r = *(const SessRec *)(S->rec + s_ss[tid_l]);
                    pf_ra = r.B; pf_rb = r.cap0; pf_rc = r.potc;
                }
                cap = pf_rb;
                const double B = pf_ra;
                const double potc = pf_rc;   // v * min(pacmax*1000/v, charger max current) / 1000 (utils.py:773-777), evaluated when the session was loaded
                s_cap[tid_l] = cap; s_tot[tid_l] = 0.0; s_prev[tid_l] = 0.0; s_cyc[tid_l] = 0; s_bcap[tid_l] = B; s_potc[tid_l] = potc;
                s_abse[tid_l] = 0.0;
                S->line[g_l].bcap = B;
                S->line[g_l].potc = potc;
                S->port_energy[g_l] = 0.0;
                S->port_current[g_l] = 0.0;
                s_dirty[tid_l] |= 1;
            }
            const bool occ_after = (ta <= sstep) && (sstep <= td);
            if (occ_after && V2C(S->reward_kind, 0) >= 9) {   // (pst_)V2G_profitmaxV2: every connected EV (reward.py:173-195)
                const int ssc = s_ss[tid_l];
                satpen += ev2g_connected_term(S->tail[ssc].des, cap, S->rec[ssc].pacmax, sixty_over_dt, td, sstep);
            }
            if (SPEC || mask) mask[e_l * P + pref_l] = occ_after ? 1 : 0;
            double o0 = 0.0, o1 = 0.0, o2 = 0.0;
            if (occ_after) {
                const double soc = cap / s_bcap[tid_l];
                if (V2C(S->state_kind, 0) == 1) { o0 = (soc == 1.0) ? 1.0 : 0.5; o1 = s_tot[tid_l]; o2 = (double)(sstep - ta); }
                else { o0 = soc; o1 = (double)(td - sstep); }
                if (soc < 1.0 && td > sstep) pot = s_potc[tid_l];  // utils.py:771
            }
            if (npc == 1) pot = (pot > c_maxp) ? c_maxp : ((pot < c_minp) ? 0.0 : pot);  // per-charger clamp (utils.py:779-789)
            if (SPEC || obs) {
                double *o = obs + (e_l * D + ocol_l);
                o[0] = o0;
                o[1] = o1;
                if (V2C(S->state_kind, 0) == 1) o[2] = o2;
            }
            if (!SPEC && obs32) {
                float *o = obs32 + (e_l * D + ocol_l);
                o[0] = (float)o0;
                o[1] = (float)o1;
                if (V2C(S->state_kind, 0) == 1) o[2] = (float)o2;
            }
            stage[1 * NS + tid_l] = profit;
            stage[2 * NS + tid_l] = satpen;
            stage[3 * NS + tid_l] = pot;
        }
        PT_MARK(3)
        lds_barrier();
        PT_MARK(1)

        // ---------------- C2: per charger (multi-port chargers, or charger history) ----------------
        if (npc > 1 || log_cs) {
            if (valid && pref_l == cs_l * npc) {  // leader = port 0 of the charger
                double pw = 0.0, cur = 0.0, pr = 0.0, ec = 0.0, ed = 0.0, pp = 0.0;
                bool fault = false;
                const double imax = c_imax;
                for (int j = 0; j < npc; j++) {  // sequential, port order (ev_charger.py:155-205)
                    pw += stage[0 * NS + tid_l + j];
                    cur += stage[7 * NS + tid_l + j];
                    pr += stage[1 * NS + tid_l + j];
                    ec += stage[4 * NS + tid_l + j];
                    ed += stage[5 * NS + tid_l + j];
                    pp += stage[3 * NS + tid_l + j];
                    if (cur - 0.0001 > imax) fault = true;
                }
                if (fault) S->env_fault[e_l] = 1;
                if (npc > 1) {
                    const double mx = c_maxp, mn = c_minp;
                    pp = (pp > mx) ? mx : ((pp < mn) ? 0.0 : pp);
                    stage[3 * NS + tid_l] = pp;
                    for (int j = 1; j < npc; j++) stage[3 * NS + tid_l + j] = 0.0;
                }
                if (log_cs) {
                    const int gc = e_l * C + cs_l;
                    S->cs_profits[gc] += pr;
                    S->cs_e_ch[gc] += ec;
                    S->cs_e_dis[gc] += ed;
                    S->cs_power_now[gc] = pw;
                    S->cs_cur_now[gc] = cur;
                    S->cs_power_hist[(t * E + e_l) * C + cs_l] = pw;
                    S->cs_cur_hist[(t * E + e_l) * C + cs_l] = cur;
                }
            }
            lds_barrier();
        }

        // SquaredTrackingErrorRewardWithPenalty only: the env's owner lane also adds the port powers up charger by charger, the order in
        // which the reference tests the sum against an exact zero (RewardIn::usage_seq).  Here, in front of phase D's closing barrier: in
        // the one-env scheme the other wavefronts enter the next step (and clear their slots of this row) without waiting for the owner.
        double q_useq = 0.0;
        const bool seq_usage = V2C(S->reward_kind == 4, false);   // (uniform)
        if (seq_usage && env_lane && pl_l == 0) q_useq = ev2g_usage_seq(stage + (size_t)pel_l * P, C, npc, S->cs_slot0, S->cs_slot0);

        // ---------------- D: LDS-staged segmented reduction, one wavefront per (env, transformer) ----------------
        const bool one_env = (BLOCK >= 512) && (G == 1) && (R > 1) && (R <= 64);   // a big env owns the whole workgroup (P > 256), several transformers
        if (one_env) {
            // Only quantity 0 (power) is needed per transformer; the other seven are needed per ENV.  Power: 8 lanes per
            // transformer segment, DPP butterfly.  Totals: BLOCK/8 lanes per quantity read the whole row with unit stride
            // (bank-conflict free), one butterfly per wavefront, per-wavefront partials into the unused rows 1..7 of tsum.
            if (tid_l < R * 8) {
                const int r = tid_l >> 3, j = tid_l & 7;
                const int b = seg[r + 1];
                double acc = 0.0;
                for (int i = seg[r] + j; i < b; i += 8) acc += stage[i];
                acc += xor1_f64(acc);
                acc += xor2_f64(acc);
                acc += xor4_f64(acc);
                if (j == 0) tsum[r] = acc;
            }
            constexpr int LPK = BLOCK / 8;          // lanes per quantity
            const int k = 1 + tid_l / LPK, c = tid_l - (k - 1) * LPK;
            if (k < EV2G_NQ) {
                const double *sp = stage + k * NS;
                double c0 = 0.0, c1 = 0.0;
                int i = c;
                for (; i + LPK < P; i += 2 * LPK) { c0 += sp[i]; c1 += sp[i + LPK]; }
                if (i < P) c0 += sp[i];
                const double v = wave_sum_dpp(c0 + c1);
                if ((tid_l & 63) == 0) tsum[k * NT + (c >> 6)] = v;
            }
        } else {
            const int wv = tid_l >> 6, lane = tid_l & 63, nw = BLOCK >> 6;
            const int k = lane >> 3, j = lane & 7;
            const int ntask = ne * R;
            for (int task = wv; task < ntask; task += nw) {
                const int tel = task / R, r = task - tel * R;
                const int a = tel * P + seg[r], b = tel * P + seg[r + 1];
                double acc = 0.0, acc2 = 0.0;  // two chains: the LDS reads of a segment overlap
                int i = a + j;
                for (; i + 8 < b; i += 16) { acc += stage[k * NS + i]; acc2 += stage[k * NS + i + 8]; }
                if (i < b) acc += stage[k * NS + i];
                acc += acc2;
                acc += xor1_f64(acc);   // DPP cross-lane moves (ev2g_device.h), not LDS-crossbar shuffles
                acc += xor2_f64(acc);
                acc += xor4_f64(acc);
                if (j == 0) tsum[k * NT + task] = acc;
            }
        }
        PT_MARK(4)
        lds_barrier();
        PT_MARK(1)

        // ---------------- E1: per (env, transformer): Transformer.reset + step + get_how_overloaded ----------------
        double over100 = 0.0;
        if (tid_l < ne * R) {  // transformer.py:258-302
            double ptr = pf_infl + pf_solar;
            ptr += tsum[0 * NT + tid_l];
            const double over = (ptr > pf_maxp + 0.0001 || ptr < pf_minp - 0.0001) ? fabs(ptr - pf_maxp) : 0.0;
            const int tel = tid_l / R, r = tid_l - tel * R;
            S->hist[EV2G_HIST(e0 + tel, t, T, R) + 2 + r] = over;
            if (last_step) S->tr_power_now[(e0 + tel) * R + r] = ptr;
            over100 = 100.0 * over;
            over_l[tid_l] = over100;
            if (r == 0) tr0max[tel] = pf_maxp;   // (its only reader is the env's owner lane: after the barrier, or -- one-env scheme -- this very lane)
        }
        // env-level quantities of this step, as the lane that owns the env (pl == 0) needs them
        double q_usage = 0.0, q_costs = 0.0, q_sat = 0.0, q_pot = 0.0, q_ech = 0.0, q_edis = 0.0, q_emerg = 0.0, q_over = 0.0;
        if (one_env) {
            // All transformer lanes sit in wavefront 0, which also holds the env's owner lane: overload sum and total power by
            // in-register butterflies, the other totals from phase D's per-wavefront partials (lane k sums quantity k) handed to
            // the owner through v_readlane.  No pass over tsum, no LDS hand-off, NO barrier: the other wavefronts go straight from
            // their observation-head stores into the next step.
            if (tid_l < 64) {
                q_over = wave_sum_dpp(over100);
                q_usage = wave_sum_dpp((tid_l < R) ? tsum[tid_l] : 0.0);
                double tot = 0.0;
                if (tid_l >= 1 && tid_l < EV2G_NQ)
                    for (int w = 0; w < BLOCK / 8 / 64; w++) tot += tsum[tid_l * NT + w];
                q_costs = readlane_f64(tot, 1); q_sat = readlane_f64(tot, 2); q_pot = readlane_f64(tot, 3);
                q_ech = readlane_f64(tot, 4); q_edis = readlane_f64(tot, 5); q_emerg = readlane_f64(tot, 6);
            }
        } else {
            if (R > 1) {
                // (env, quantity) sums over the env's R transformers: one wavefront per sum, lanes strided over the
                // transformers, fixed xor tree (a single lane adding R values is an R-long chain of LDS round trips)
                const int wave = tid_l >> 6, ln = tid_l & 63;
                for (int task = wave; task < ne * EV2G_NQ; task += (BLOCK >> 6)) {
                    const int tel = task / EV2G_NQ, k = task - tel * EV2G_NQ;
                    double v = 0.0;
                    for (int r = ln; r < R; r += 64) v += tsum[k * NT + tel * R + r];
                    v = wave_sum(v);
                    if (ln == 0) esum[k * G + tel] = v;
                }
            }
            lds_barrier();
            if (R > 1) {   // the overload penalties of E1 are visible now: their per-env sum, same scheme
                const int wave = tid_l >> 6, ln = tid_l & 63;
                for (int tel = wave; tel < ne; tel += (BLOCK >> 6)) {
                    double v = 0.0;
                    for (int r = ln; r < R; r += 64) v += over_l[tel * R + r];
                    v = wave_sum(v);
                    if (ln == 0) osum[tel] = v;
                }
                lds_barrier();
            }
            const double *es = (R > 1) ? esum : tsum;
            const int esn = (R > 1) ? G : NT;
            if (env_lane) {
                q_usage = es[0 * esn + pel_l];
                if (pl_l == 0) {
                    q_over = (R > 1) ? osum[pel_l] : over_l[pel_l];
                    q_costs = es[1 * esn + pel_l]; q_sat = es[2 * esn + pel_l]; q_pot = es[3 * esn + pel_l];
                    q_ech = es[4 * esn + pel_l]; q_edis = es[5 * esn + pel_l]; q_emerg = es[6 * esn + pel_l];
                }
            }
        }

        // ---------------- E2: per env: reward, histories, observation head ----------------
        // the parameters the owner lane's chain needs, fetched in ONE scalar-load batch (one wait) instead of a round trip per use
        double *const p_hist = (double *)S->hist, *const p_cost = (double *)S->x_cost;
        double *const p_acc = (double *)S->env_acc;
        const long long c_stride = S->x_c_stride;
        const int rkind = V2C(S->reward_kind, 0), ckind = V2C(S->cost_kind, 0);
        if (env_lane) {
            const double usage = q_usage;
            if (pl_l == 0) {
                const double over_sum = q_over;
                p_hist[EV2G_HIST(pe_l, t, T, R)] = usage;
                const double potn = q_pot;
                if (sstep < T) p_hist[EV2G_HIST(pe_l, sstep, T, R) + 1] = potn;
                const double costs = q_costs;
                RewardIn ri;
                ri.costs = costs; ri.usage = usage; ri.usage_seq = seq_usage ? q_useq : usage; ri.sp = pf_sp; ri.over100 = over_sum; ri.user = q_sat;
                ri.pot_t = pot_prev[pel_l]; ri.pot_tm1 = pot_prev2[pel_l]; ri.tr0_maxp = tr0max[pel_l];
                const double reward = ev2g_reward(rkind, ri);
                pot_prev2[pel_l] = ri.pot_t;
                pot_prev[pel_l] = potn;
                double *acc = eacc + pel_l * 5;
                acc[0] += reward;
                acc[1] += costs;
                acc[2] += q_ech;
                acc[3] += q_edis;
                acc[4] += q_emerg;
                if (SPEC) { io.reward[pe_l] = reward; io.done[pe_l] = (sstep >= T) ? 1 : 0; }
                if (!SPEC && io.reward) io.reward[(long long)kk * io.r_stride + pe_l] = reward;
                if (!SPEC && io.done) io.done[(long long)kk * io.d_stride + pe_l] = (sstep >= T) ? 1 : 0;
                if (!SPEC && p_cost)   // cost_function (rl_agent/cost.py:8-27)
                    p_cost[(long long)(io.step0 + kk) * c_stride + pe_l] = (ckind == 2) ? costs : over_sum + q_sat;
                if (sstep >= T || last_step) {  // flush the episode accumulators (get_statistics reads them)
                    double *ga = p_acc + pe_l * 8;
                    for (int i = 0; i < 5; i++) { ga[i] += acc[i]; acc[i] = 0.0; }
                }
            }
            if (SPEC || obs || obs32) {
                double *o = (SPEC || obs) ? obs + pe_l * D : nullptr;
                float *o32 = (!SPEC && obs32) ? obs32 + pe_l * D : nullptr;
#define EV2G_OBS_PUT(col, val) { const double v_ = (val); if (SPEC || o) o[col] = v_; if (!SPEC && o32) o32[col] = (float)v_; }
                if (V2C(S->state_kind, 0) == 1) {  // PublicPST state.py:6-35
                    if (pl_l == 0) { EV2G_OBS_PUT(0, (double)sstep / (double)T) EV2G_OBS_PUT(1, (sstep < T) ? pf_ob0 : 0.0) EV2G_OBS_PUT(2, usage) }
                } else {  // V2G_profit_max(_loads) state.py:65-83, :108-135
                    if (pl_l == 0) { EV2G_OBS_PUT(0, (double)sstep) EV2G_OBS_PUT(1, usage) }
                    if (hdst0_l >= 0) EV2G_OBS_PUT(hdst0_l, (hsrc0_l < 20) ? ((sstep + hsrc0_l < T) ? fabs(pf_ob0) : 0.0) : pf_ob0)
                    if (hdst1_l >= 0) EV2G_OBS_PUT(hdst1_l, (hsrc1_l < 20) ? ((sstep + hsrc1_l < T) ? fabs(pf_ob1) : 0.0) : pf_ob1)
                    // envs with more head columns than two passes of their lanes (many transformers): the rest, unprefetched
                    for (int c = pl_l + 2 * lpe; c < nhead; c += lpe) {
                        if (c < 20) {   // (more than 25 envs per workgroup: fewer than ten lanes per env, price columns are left too)
                            EV2G_OBS_PUT(2 + c, (sstep + c < T) ? fabs(S->price_ch[pec * T + min(sstep + c, T - 1)]) : 0.0)
                        } else {
                            const int i = c - 20, r = i / 40, j = i - r * 40;
                            EV2G_OBS_PUT(trobs[r] + j, S->win_tab[(((long long)pec * R + r) * (T + 1) + sstep) * 40 + j])
                        }
                    }
                }
#undef EV2G_OBS_PUT
            }
        }
        PT_MARK(5)
        PT_STEP_END(false)
        t += 1;
        // no barrier needed here: the next step's phase A only touches stage[0,4..7], s_amps, items and cnt, none
        // of which phase E reads; tsum/esum/over_l are rewritten only after three more barriers (in the one-env scheme wavefront 0
        // may still be in phase E while the others start the next step: same argument).
    }
    PT_FLUSH
    // ---- write the LDS-resident port state back ----
    __syncthreads();
    if (valid) {
        const int d = s_dirty[tid];
        EV2G_GP(PortLine) ln = S->line + g;
        if (d & 2) { ln->ta = s_ta[tid]; ln->td = s_td[tid]; }
        if (d) {   // the line carries the efficiency-table id of the attached EV next to its cycle count (the fast path reads it from there)
            const int ssd = s_ss[tid];
            ln->ss = ssd; ln->cyc_lut = ev2g_line_pack(s_cyc[tid], ssd >= 0 ? S->ss_lut[ssd] : -1);
        }
        if (d & 1) { ln->cap = s_cap[tid]; ln->tot = s_tot[tid]; ln->prev = s_prev[tid]; if (log_soc) ln->abse = s_abse[tid]; }
    }
}
#undef V2C
