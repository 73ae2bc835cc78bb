// ev2g_refill.h -- scenario generation ON THE DEVICE: what EV2Gym.reset() draws for an episode (ev2gym_env.py:243-296: EV_spawner
// utils.py:477-557, spawn_single_EV :177-345, load_transformers loaders.py:227-296 + transformer.py:80-256, load_electricity_prices
// loaders.py:392-461, generate_power_setpoints utils.py:664-757), written straight into the resident scenario pool in the layout the
// step kernels read -- no host work, no PCIe.
//
// One wavefront per scenario runs the SAME element functions as the host generator (ev2g_gen.h; ev2g_generate walks them with loops,
// here a lane takes a step or a port), so slot s refilled as scenario i of the stream (config, seed) holds bit for bit what
// ev2g_generate(config, ., seed) yields at index i followed by ev2g_load_scenarios: counter-based random numbers, elementary functions
// with one result everywhere (ev2g_dlog ...), reductions that are max / min or the fixed 64-leaf tree.
//   * prices, transformer loads / PV / forecasts: a lane per step;
//   * EV sessions: a lane per port slot -- a port's sessions depend on that port's history only -- in two passes (count, prefix over
//     the slots, write): device order (scenario, slot, arrival) is the order they are produced in, so there is no sort; the loader's
//     per-session work (session record with its gates, efficiency-table id, AFAP energy, next-window chain, first-session tables) is
//     done where the session is drawn;
//   * demand-response events: their slices are applied lane-parallel, `any` / `max` over the slice by ballot / wave max;
//   * power setpoints: sessions port by port, a lane per step, weight sums on the wavefront's xor tree (ev2g_tree64 on the host).
// The observation tables of the refilled slots are rebuilt afterwards by the loader's own table kernels, restricted to those slots.
// Scope: a pool loaded with EV2G_FLAG_REFILLABLE (fixed-size session blocks per scenario).  Single-port chargers (every shipped config; the
// fast path's shape): a port's sessions are the slot's.  Chargers with several ports and topology files (round 4; up to 256 steps / 256
// ports): an arriving EV takes its charger's FIRST FREE port (ev_charger.py:266-286) -- what ev2g_load_scenarios replays on the host for a
// loaded batch is replayed here per charger (a lane each) between the two passes: every session then knows the slot it lands on and its rank
// there, and the slots' first-session tables and next-window chains are written behind the second pass.  A scenario that draws more sessions than its block holds keeps the
// first `cap` of them in device order and is counted in RefillArgs::overflow (the block is 25 % + 8 larger than the largest scenario of
// the loaded batch; ev2g_pool_refill reports the count).
#pragma once
#include "ev2g_device.h"
#include "ev2g_gen.h"

struct RefillArgs {
    ev2g_gen_config cfg;        // spec_* / tab_* pointers: device copies
    Ev2gGenRun g0;              // c / pv_series fixed up on the device
    const double *pv_series;    // device copy (or null)
    const int *spec_row;        // device copy (or null)
    const double *lut_rowmax;   // [n_lut] max of every efficiency table (percent)
    unsigned long long seed;
    long long first_index;
    int first_slot, n, cap;
    int *overflow;
    double *head_tab; int head_nh;   // fast path: observation head table [M, T+1, NH] (or null)
    double *step_tab;                // fast path: [M, T, 8] per-step scalars (or null)
    unsigned long long *dbg;    // [16] cycle stamps of workgroup 0 (tools/refill_time.py --stamps), or null
    int multi;                  // chargers with several ports (or a topology file): an arriving EV takes the charger's first free port (ev_charger.py:266-286) --
                                // the kernel replays that per charger before it places the sessions (needs T <= 256 and at most EV2G_RF_K sessions per port)
    const double *tr_cap;       // [R] transformer capacities of a topology file (device copy), or null: cfg.transformer_max_power
    const int *cls_of;          // [models][C] battery-maths dictionary entry of (car model, charger) when DevScn::dict is set (ev2g_pool_refill builds it), else null
};
#define RF_STAMP(i) if (a.dbg && threadIdx.x == 0) { if (blockIdx.x == 0) a.dbg[i] = __builtin_readcyclecounter(); \
        if (blockIdx.x == gridDim.x - 1 && ((i) == 0 || (i) == 6)) a.dbg[8 + ((i) != 0)] = __builtin_readcyclecounter(); \
        if (blockIdx.x == gridDim.x / 2 && ((i) == 0 || (i) == 6)) a.dbg[10 + ((i) != 0)] = __builtin_readcyclecounter(); }

__device__ __forceinline__ double rf_wave_max(double v) {
    for (int d = 32; d > 0; d >>= 1) v = fmax(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ double rf_wave_min(double v) {
    for (int d = 32; d > 0; d >>= 1) v = fmin(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ int rf_wave_incl_scan(int v, int lane) {
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(v, d, 64); if (lane >= d) v += o; }
    return v;
}

// EV.calculate_max_energy_with_AFAP (ev.py:407-440): the loader's afap_energy, same operations
__device__ inline double rf_afap(double cap0, double B, double pac, double max_cs_power, double eff, int ta, int td, int dt) {
    const double max_power = (fabs(max_cs_power) > fabs(pac)) ? pac : max_cs_power;
    double x = cap0;
    for (int k = ta; k < td + 1; k++) {
        x += max_power * eff * dt / 60.0;
        x = ceil(x * 100.0) / 100.0;
        if (x > B) { x = B; break; }
    }
    return x;
}

// LDS (dynamic), kept under 10 KB at the shipped shapes so that 16 one-wavefront workgroups share a CU (4096 scenarios = one round):
//   doubles l_cp[T] | l_a[T] | l_b[T] | X[3T + 96] | demand-response events [16][3] | per staged session need, lo, hi [cap]
//   X is, phase after phase: the spawner's per-step {stay, energy} + uint2 {key, threshold} [T]  ->  solar, load forecast, PV forecast of
//   the transformer in work [T each]  ->  the median filter's padded row [T + 96]
//   u64 id [cap]; ints t_arr, t_dep [cap], base / count per PORT [P]; uint8 spawn steps [P][EV2G_RF_K]
#define EV2G_RF_K 8   // spawn steps remembered per port between the two passes (a port with more re-runs its trials in pass 2)
//   multi-port chargers / topology files (RefillArgs::multi) add: uint8 departure steps, resolved slot and rank in it [P][EV2G_RF_K] each;
//   ints base / count per SLOT after the first-free replay, slot of a port, free-from step of a port [P each]
__host__ __device__ inline size_t ev2g_refill_lds_bytes(int T, int P, int cap, int multi = 0) {
    return sizeof(double) * ((size_t)6 * T + 96 + 3 * 16 + 3 * (size_t)cap) + sizeof(unsigned long long) * (size_t)cap + sizeof(int) * (2 * (size_t)cap + 2 * (size_t)P) +
           (((size_t)P * EV2G_RF_K + 7) & ~(size_t)7) + (multi ? 3 * (((size_t)P * EV2G_RF_K + 7) & ~(size_t)7) + sizeof(int) * 4 * (size_t)P : 0);
}

__global__ void __launch_bounds__(64) ev2g_refill_kernel(DevScn s, DevState st, RefillArgs a, double *ss_afap) {
    extern __shared__ double rlds[];
    const int lane = threadIdx.x;
    const int T = s.T, P = s.P, R = s.R, cap = a.cap;
    double *l_cp = rlds, *l_a = l_cp + T, *l_b = l_a + T, *l_x = l_b + T;
    double *l_stay = l_x, *l_emean = l_x + T; uint2 *l_kt = (uint2 *)(l_x + 2 * T);          // phase 1 (sessions)
    double *l_sol = l_x, *l_lf = l_x + T, *l_pvf = l_x + 2 * T;                                 // phase 2 (transformers)
    double *l_pad = l_x;                                                                        // phase 3 (setpoints)
    double *l_dr = l_x + 3 * T + 96;
    double *l_need = l_dr + 3 * 16, *l_lo = l_need + cap, *l_hi = l_lo + cap;
    unsigned long long *l_id = (unsigned long long *)(l_hi + cap);
    int *l_ta = (int *)(l_id + cap), *l_td = l_ta + cap, *l_pbase = l_td + cap, *l_pcnt = l_pbase + P;
    unsigned char *l_spawn = (unsigned char *)(l_pcnt + P);
    constexpr int K = EV2G_RF_K;
    const size_t spb = ((size_t)P * K + 7) & ~(size_t)7;
    unsigned char *l_dep = l_spawn + spb, *l_res = l_dep + spb, *l_rank = l_res + spb;   // (multi only)
    int *l_rbase = (int *)(l_rank + spb), *l_rcnt = l_rbase + P, *l_pslot = l_rcnt + P, *l_free = l_pslot + P;
    const bool multi = a.multi != 0;
    const int ms = a.first_slot + blockIdx.x;                  // pool slot
    const unsigned long long m = (unsigned long long)(a.first_index + blockIdx.x);   // scenario index of the stream
    const ev2g_gen_config &c = a.cfg;
    Ev2gGenRun g = a.g0;
    g.c = &a.cfg;
    g.pv_series = a.pv_series;
    const Ev2gRng rng = ev2g_rng(a.seed, m);
    const Ev2gRng rng_tr = (c.tr_seed != -1) ? ev2g_rng((unsigned long long)c.tr_seed, m) : rng;
    const Ev2gScenarioDraw dr = ev2g_gen_scenario_draw(g, rng, rng_tr, a.pv_series != nullptr);   // (g.hour is still the config's start hour here)
    g.hour = dr.hour;
    const int dt = g.dt;
#define RW(type, field) (const_cast<type *>(s.field))
    RF_STAMP(0)

    // ---- prices ----
    for (int t = lane; t < T; t += 64) {
        const double pr = ev2g_gen_price_at(g, rng, dr.price_scale, t);
        l_cp[t] = pr;
        RW(double, price_ch)[(size_t)ms * T + t] = -pr;
        RW(double, price_dis)[(size_t)ms * T + t] = pr * c.discharge_price_factor;
    }

    RF_STAMP(1)
    // ---- the spawner's tables of every step, once per scenario (a lane per step) ----
    for (int t = lane; t < T; t += 64) {
        const Ev2gStepTables stt = ev2g_gen_step_tables(g, rng, dr.weekend, t);
        l_stay[t] = stt.stay; l_emean[t] = stt.emean; l_kt[t] = make_uint2(stt.key, stt.threshold);
    }
    const Ev2gFleet fleet = ev2g_fleet(g);
    const double share_sum = ev2g_gen_share_sum(fleet);
    __syncthreads();
    auto tab = [&](int t) { const uint2 kt = l_kt[t]; return Ev2gStepTables{0.0, l_stay[t], l_emean[t], kt.y, kt.x}; };

    RF_STAMP(2)
    // ---- sessions, pass 1: how many each port slot draws (and at which steps); prefix over the slots (device order = scenario, slot, arrival) ----
    int carry = 0;
    for (int q0 = 0; q0 < P; q0 += 64) {
        const int q = q0 + lane;
        int n = 0, p = 0;
        if (q < P) {
            p = s.slot_port[q];
            n = ev2g_gen_port_sessions(g, rng, fleet, share_sum, p, tab, [&](int i, const Ev2gGenSession &e) { if (i < EV2G_RF_K) { l_spawn[p * EV2G_RF_K + i] = (unsigned char)(e.t_arr - 1); if (multi) l_dep[p * EV2G_RF_K + i] = (unsigned char)e.t_dep; } });
            if (multi) { l_pslot[p] = q; l_rcnt[q] = 0; l_free[p] = 0; }
        }
        if (multi && n > K) { n = K; if (a.overflow) atomicAdd(a.overflow, 1); }   // (a port with more sessions than the replay remembers: cut, and counted)
        const int incl = rf_wave_incl_scan(n, lane);
        const int base = carry + incl - n;
        if (q < P) { l_pbase[p] = base; l_pcnt[p] = n; }
        carry += __shfl(incl, 63, 64);
    }
    const int total = carry;
    if (lane == 0) {
        if (total > cap) atomicAdd(a.overflow, 1);
        RW(int, scn_sess_end)[ms] = ms * cap + min(total, cap);
    }
    __syncthreads();
    if (multi) {
        // ---- first-free replay (ev_charger.py:266-286; the loader's, ev2g_load_scenarios): a charger's arrivals in profile order (arrival step, then
        //      generator port) each take the charger's lowest port that is free at the end of the step before -- a lane per charger; afterwards every
        //      remembered session knows the slot it lands on and its rank among that slot's sessions ----
        for (int c0 = 0; c0 < s.C; c0 += 64) {
            const int cc = c0 + lane;
            if (cc < s.C) {
                const int pb = s.cs_pbase[cc], np = s.cs_np[cc];
                for (int j = 0; j < np; j++) l_rbase[pb + j] = 0;   // (cursor of generator port pb + j for the moment)
                for (;;) {
                    int bj = -1, bt = EV2G_INT_MAX;
                    for (int j = 0; j < np; j++) {
                        const int pp = pb + j, i = l_rbase[pp];
                        if (i < l_pcnt[pp]) { const int ta = (int)l_spawn[pp * K + i] + 1; if (ta < bt) { bt = ta; bj = j; } }
                    }
                    if (bj < 0) break;
                    const int pp = pb + bj, i = l_rbase[pp];
                    l_rbase[pp] = i + 1;
                    int jr = 0;
                    while (jr < np - 1 && l_free[pb + jr] > bt - 1) jr++;   // (the generator's ports of a charger never hold more EVs than it has ports)
                    l_free[pb + jr] = (int)l_dep[pp * K + i];              // freed inside step t_dep, before that step's arrivals
                    const int qs = l_pslot[pb + jr];
                    l_res[pp * K + i] = (unsigned char)qs;
                    l_rank[pp * K + i] = (unsigned char)l_rcnt[qs];
                    l_rcnt[qs] += 1;
                }
            }
        }
        __syncthreads();
        int carry2 = 0;
        for (int q0 = 0; q0 < P; q0 += 64) {   // device order: (scenario, slot, arrival)
            const int q = q0 + lane;
            const int n = (q < P) ? l_rcnt[q] : 0;
            const int incl = rf_wave_incl_scan(n, lane);
            if (q < P) l_rbase[q] = carry2 + incl - n;
            carry2 += __shfl(incl, 63, 64);
        }
        __syncthreads();
    }

    RF_STAMP(3)
    // ---- sessions, pass 2: draw again and write where they belong ----
    const size_t d0 = (size_t)ms * cap;
    for (int q0 = 0; q0 < P; q0 += 64) {
        const int q = q0 + lane;
        if (q >= P) continue;
        const int p = s.slot_port[q], cs = s.slot_cs[q];
        const int base = l_pbase[p];
        const int n_eff = max(0, min(l_pcnt[p], cap - base));
        const size_t gs = (size_t)ms * P + q;
        if (!multi) {
            RW(int, port_first)[gs] = n_eff > 0 ? (int)(d0 + base) : -1;
            RW(int, port_end)[gs] = n_eff > 0 ? (int)(d0 + base + n_eff) : -1;
            if (n_eff == 0) RW(int2, port_first_win)[gs] = make_int2(EV2G_INT_MAX, EV2G_INT_MAX);
        }
        const double V = s.cs_volt[cs];
        const int ph = s.cs_ph[cs];
        const double sq = sqrt((double)ph);
        const double mp = s.cs_imax[cs] * V * sq / 1000;                       // EV_Charger.get_max_power (ev_charger.py:251-252)
        const double min_cs = s.cs_imin[cs] * V * sq / 1000, max_cs = s.cs_imax[cs] * V * sq / 1000;   // (generate_power_setpoints)
        const double v_gate = s.cs_vk[(size_t)cs * 4 + ph];
        const double pac_min_sp = c.heterogeneous_ev_specs ? 0.0 : c.ev_min_ac_charge_power;
        auto write_session = [&](int i, const Ev2gGenSession &e) {
            // where the session goes: behind the port's earlier ones -- or, after the first-free replay, at its rank on the slot it was given (same charger)
            int dofs = base + i, qs = q;
            bool to_pool = i < n_eff, to_lds = i < n_eff;
            if (multi) { qs = l_res[p * K + i]; dofs = l_rbase[qs] + (int)l_rank[p * K + i]; to_pool = dofs < cap; to_lds = base + i < cap; }
            if (!to_pool && !to_lds) return;
            const size_t d = d0 + dofs;
            const Ev2gSessFields f = ev2g_gen_session_fields(g, rng, e, a.spec_row);
            if (to_lds) {   // what the power setpoints need of this session (LDS, in the generator's order: port by port)
                const int k = base + i;
                l_id[k] = (unsigned long long)(e.t_arr - 1) * (unsigned long long)P + (unsigned long long)e.port;
                l_ta[k] = e.t_arr; l_td[k] = e.t_dep;
                l_need[k] = (e.B - e.cap0) * (100 + c.power_setpoint_flexiblity) / 100;
                l_lo[k] = fmax(pac_min_sp, min_cs); l_hi[k] = fmin(e.pac, max_cs);
            }
            if (!to_pool) return;
            RW(int, ss_tarr)[d] = e.t_arr; RW(int, ss_tdep)[d] = e.t_dep; RW(int, ss_ntarr)[d] = EV2G_INT_MAX; RW(int, ss_ntdep)[d] = EV2G_INT_MAX;
            RW(int, ss_phases)[d] = f.phases; RW(int, ss_lut)[d] = f.lut; RW(int, ss_slot)[d] = qs;
            RW(double, ss_cap0)[d] = e.cap0; RW(double, ss_B)[d] = e.B; RW(double, ss_des)[d] = f.desired; RW(double, ss_minB)[d] = f.minB;
            RW(double, ss_emerg)[d] = f.min_emerg; RW(double, ss_pacmax)[d] = e.pac; RW(double, ss_pacmin)[d] = f.pac_min;
            RW(double, ss_pdismax)[d] = f.pdis_max; RW(double, ss_pdismin)[d] = f.pdis_min; RW(double, ss_ts)[d] = f.ts; RW(double, ss_tsm)[d] = f.tsm;
            RW(double, ss_etach)[d] = f.eta_ch; RW(double, ss_etadis)[d] = f.eta_dis;
            SessRec r;
            r.B = e.B; r.cap0 = e.cap0; r.minB = f.minB; r.emerg = f.min_emerg;
            r.pacmax = e.pac; r.pdismax = f.pdis_max; r.ts = f.ts; r.tsm = f.tsm; r.eta_ch = f.eta_ch; r.eta_dis = f.eta_dis;
            r.gate_ch = f.pac_min * 1000.0 / v_gate;
            r.gate_dis = f.pdis_min * 1000.0 / v_gate;
            r.v = s.cs_vk[(size_t)cs * 4 + min(ph, f.phases)];
            r.rB = 1.0 / r.B; r.rv = 1.0 / r.v;
            {
                const double evc = r.pacmax * 1000.0 / r.v, imax = s.cs_imax[cs];
                r.potc = r.v * ((evc < imax) ? evc : imax) / 1000.0;
            }
            RW(SessRec, rec)[d] = r;
            if (s.sess_dyn) {   // fast path: the per-session operands and the dictionary entry (ev2g_device.h, ClsRec / SessDyn)
                SessDyn dy;
                dy.ts = f.ts; dy.eta_ch = f.eta_ch; dy.eta_dis = f.eta_dis; dy.lut = f.lut;
                if (s.dict) dy.cls = a.cls_of[(size_t)e.model * s.C + cs];
                else { dy.cls = (int)d; RW(ClsRec, cls_rec)[d] = ev2g_cls_of(r); }
                RW(SessDyn, sess_dyn)[d] = dy;
            }
            SessTail tl;
            tl.des = f.desired; tl.nt_arr = EV2G_INT_MAX; tl.nt_dep = EV2G_INT_MAX;
            RW(SessTail, tail)[d] = tl;
            if (multi) {   // (a slot's sessions come from several lanes: chained behind the pass, below)
            } else if (i > 0) {   // this port's previous session learns its successor's window
                RW(int, ss_ntarr)[d - 1] = e.t_arr; RW(int, ss_ntdep)[d - 1] = e.t_dep;
                RW(SessTail, tail)[d - 1].nt_arr = e.t_arr; RW(SessTail, tail)[d - 1].nt_dep = e.t_dep;
            } else {
                RW(int2, port_first_win)[gs] = make_int2(e.t_arr, e.t_dep);
            }
            const double eff = f.lut >= 0 ? a.lut_rowmax[f.lut] / 100.0 : f.eta_ch;
            ss_afap[d] = rf_afap(e.cap0, e.B, e.pac, mp, eff, e.t_arr, e.t_dep, dt);
        };
        if (l_pcnt[p] <= EV2G_RF_K && T <= 256) {   // the steps at which this port spawned are known from pass 1: only the sessions are drawn again
            const int n_draw = multi ? l_pcnt[p] : n_eff;
            for (int i = 0; i < n_draw; i++) {
                const int t = l_spawn[p * EV2G_RF_K + i];
                Ev2gGenSession e;
                ev2g_gen_make_session(g, rng, fleet, share_sum, t, p, l_stay[t], l_emean[t], &e);
                write_session(i, e);
            }
        } else {
            ev2g_gen_port_sessions(g, rng, fleet, share_sum, p, tab, write_session);
        }
    }

    if (multi) {   // ---- the slots' first-session tables and next-window chains, from what the pass wrote (a lane per slot) ----
        __threadfence();
        __syncthreads();
        for (int q0 = 0; q0 < P; q0 += 64) {
            const int q = q0 + lane;
            if (q >= P) continue;
            const int base = l_rbase[q], cnt = max(0, min(l_rcnt[q], cap - base));
            const size_t gs = (size_t)ms * P + q, db = d0 + base;
            RW(int, port_first)[gs] = cnt > 0 ? (int)db : -1;
            RW(int, port_end)[gs] = cnt > 0 ? (int)(db + cnt) : -1;
            RW(int2, port_first_win)[gs] = cnt > 0 ? make_int2(s.ss_tarr[db], s.ss_tdep[db]) : make_int2(EV2G_INT_MAX, EV2G_INT_MAX);
            for (int j = 1; j < cnt; j++) {
                const int ta = s.ss_tarr[db + j], td = s.ss_tdep[db + j];
                RW(int, ss_ntarr)[db + j - 1] = ta; RW(int, ss_ntdep)[db + j - 1] = td;
                RW(SessTail, tail)[db + j - 1].nt_arr = ta; RW(SessTail, tail)[db + j - 1].nt_dep = td;
            }
        }
    }

    RF_STAMP(4)
    // ---- transformers ----
    for (int k = 0; k < R; k++) {
        const size_t o = ((size_t)ms * R + k) * T;
        const double capk = a.tr_cap ? a.tr_cap[k] : c.transformer_max_power;
        double *infl = l_a, *maxp = l_b;   // this transformer's inflexible load and max_power, in LDS while the events work on them
        __syncthreads();
        double lvl = 0.0, mult = 0.0, mx = 0.0;
        if (c.inflexible_loads) {
            lvl = rng_tr.uni(EV2G_RS_TR, (unsigned long long)k, 0, 0.6, 1.4);
            for (int t = lane; t < T; t += 64) { const double raw = ev2g_gen_infl_raw(g, rng_tr, k, lvl, t); infl[t] = raw; mx = fmax(mx, raw); }
            mx = rf_wave_max(mx);
            mult = rng_tr.normal(EV2G_RS_TR, (unsigned long long)k, 1, c.inflexible_loads_capacity_multiplier_mean, 0.1);
        }
        double sa = 0.0, sm = 0.0;
        if (c.solar_power) { sa = rng_tr.uni(EV2G_RS_TR, (unsigned long long)k, 2, 0.9, 1.1); sm = rng_tr.normal(EV2G_RS_TR, (unsigned long long)k, 3, c.solar_power_capacity_multiplier_mean, 0.1); }
        for (int t = lane; t < T; t += 64) {
            infl[t] = c.inflexible_loads ? ev2g_gen_infl_scaled(infl[t], mult, capk, mx) : 0.0;
            maxp[t] = capk;
        }
        __syncthreads();
        double *drs = RW(double, tr_dr) + ((size_t)ms * R + k) * s.ND * 3;
        for (int i = lane; i < s.ND * 3; i += 64) drs[i] = 0.0;
        if (lane == 0) { RW(int, tr_ndr)[(size_t)ms * R + k] = c.demand_response ? c.dr_events_per_day : 0; RW(int, tr_ahead)[(size_t)ms * R + k] = g.steps_ahead; }
        if (c.demand_response) {   // one event after the other (transformer.py:96-138): the slice is applied lane-parallel
            for (int e = 0; e < c.dr_events_per_day; e++) {
                Ev2gDrEvent ev = ev2g_gen_dr_event(g, rng_tr, k, e);
                bool over = false;
                double load_max = -INFINITY;
                for (int t = ev.s0 + lane; t < ev.s1; t += 64) {
                    const double v = maxp[t] - maxp[t] * ev.capp / 100;
                    maxp[t] = v;
                    if (infl[t] > v) over = true;
                    load_max = fmax(load_max, infl[t]);
                }
                over = __ballot(over) != 0ull;
                load_max = rf_wave_max(load_max);
                __syncthreads();
                if (over) {   // the load exceeds the reduced limit inside the event: the limit is lifted to the load's maximum
                    for (int t = ev.s0 + lane; t < ev.s1; t += 64) maxp[t] = load_max;
                    __syncthreads();
                    double mxp = -INFINITY;
                    for (int t = lane; t < T; t += 64) mxp = fmax(mxp, maxp[t]);
                    mxp = rf_wave_max(mxp);
                    ev.capp = 100 * (1 - load_max / mxp);
                }
                if (lane == 0) {
                    drs[e * 3 + 0] = ev.es; drs[e * 3 + 1] = ev.ee; drs[e * 3 + 2] = ev.capp;
                    if (e < 16) { l_dr[e * 3 + 0] = ev.es; l_dr[e * 3 + 1] = ev.ee; l_dr[e * 3 + 2] = ev.capp; }
                }
            }
        }
        __syncthreads();
        double peak = -INFINITY;
        for (int t = lane; t < T; t += 64) {
            const double mxv = maxp[t], mnv = -capk, il = infl[t];
            const double sol = c.solar_power ? ev2g_gen_solar_at(g, dr.sun, sa, sm, capk, t) : 0.0;
            RW(double, tr_maxp)[o + t] = mxv; RW(double, tr_minp)[o + t] = mnv; RW(double, tr_infl)[o + t] = il; RW(double, tr_solar)[o + t] = sol;
            RW(double, tr_base)[o + t] = il + sol;
            const double lfv = c.inflexible_loads ? ev2g_gen_load_forecast_at(g, rng_tr, k, t, il, mnv, mxv) : 0.0;
            const double pvv = c.solar_power ? ev2g_gen_pv_forecast_at(g, rng_tr, k, t, sol) : 0.0;
            RW(double, tr_lf)[o + t] = lfv; RW(double, tr_pvf)[o + t] = pvv;
            l_sol[t] = sol; l_lf[t] = lfv; l_pvf[t] = pvv;
            peak = fmax(peak, mxv);
        }
        peak = rf_wave_max(peak);
        if (lane == 0) RW(double, tr_peak)[(size_t)ms * R + k] = peak;
        // ---- this transformer's observation windows (ev2g_build_window_table_kernel's values: load_minus_pv_at / power_limit_at), from LDS;
        //      on the fast path (one transformer) they are also columns 20..59 of the observation head table ----
        RF_STAMP(7)
        if (s.win_tab || a.head_tab) {
            __syncthreads();
            const int nd = (c.demand_response && c.dr_events_per_day <= 16) ? c.dr_events_per_day : 0, ahead = g.steps_ahead;
            double *ht = (a.head_tab && a.head_nh == 60) ? a.head_tab + (size_t)ms * (size_t)(T + 1) * 60 : nullptr;
            // (the window table itself is read by ev2g_step_v2 only: where the fast path's head table exists nothing reads it after the load)
            double *wt = (s.win_tab && !ht) ? RW(double, win_tab) + ((size_t)ms * R + k) * (size_t)(T + 1) * 40 : nullptr;
            // events in registers (their limit is one value per event); beyond four, the rest from LDS
            int d_es[4], d_ee[4];
            double d_lim[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const bool on = e < nd;
                d_es[e] = on ? (int)l_dr[e * 3] : 0x7fffffff; d_ee[e] = on ? (int)l_dr[e * 3 + 1] : -1;
                d_lim[e] = on ? peak - peak * l_dr[e * 3 + 2] / 100.0 : 0.0;
            }
            // Two loops of (T + 1) * 20 elements each instead of one over both halves of a row: the halves share nothing, and in one loop every
            // wavefront executed both bodies for every element (round 3: 80 k of the kernel's 220 k cycles).  Element i = step * 20 + j, i = lane,
            // lane + 64, ...: (step, j) advance by (3, +4) with a carry -- no division per element.
            {   // (loads - pv) window, Transformer.get_load_pv_forecast transformer.py:173-188: columns 0..19
                int step = lane / 20, j = lane - step * 20;
                for (int i = lane; i < (T + 1) * 20; i += 64) {
                    const int kk = step + j;
                    // the actual series for the current step (j == 0) and behind the horizon of the last one, the forecast otherwise; the same
                    // element [min(kk, T - 1)] of either pair (1.0 * x == x)
                    const bool actual = (kk < T) ? (j == 0) : (step >= T - 1);
                    const double *lsrc = actual ? infl : l_lf, *psrc = actual ? l_sol : l_pvf;
                    const int ke = min(kk, T - 1);
                    const double v = lsrc[ke] - psrc[ke];
                    if (wt) wt[step * 40 + j] = v;
                    if (ht) ht[(size_t)step * 60 + 20 + j] = v;
                    step += 3; j += 4;
                    if (j >= 20) { j -= 20; step += 1; }
                }
            }
            {   // power limits, Transformer.get_power_limits transformer.py:142-171: columns 20..39
                int step = lane / 20, j = lane - step * 20;
                for (int i = lane; i < (T + 1) * 20; i += 64) {
                    double v = peak * 1.0;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (e < nd) {   // (uniform)
                            const int es = d_es[e], ee = d_ee[e];
                            if (step + ahead >= es && ee >= step) {
                                int aa, bb;
                                if (step > es) { aa = 0; bb = ee - step; } else { aa = es - step; bb = ee - step; }
                                if (aa < 0) aa = -aa;
                                if (bb < 0) bb = -bb;
                                if (j >= aa && j < bb) v = d_lim[e];
                            }
                        }
                    }
                    for (int e = 4; e < nd; e++) {
                        const int es = (int)l_dr[e * 3], ee = (int)l_dr[e * 3 + 1];
                        if (step + ahead >= es && ee >= step) {
                            int aa, bb;
                            if (step > es) { aa = 0; bb = ee - step; } else { aa = es - step; bb = ee - step; }
                            if (aa < 0) aa = -aa;
                            if (bb < 0) bb = -bb;
                            if (j >= aa && j < bb) v = peak - peak * l_dr[e * 3 + 2] / 100.0;
                        }
                    }
                    if (wt) wt[step * 40 + 20 + j] = v;
                    if (ht) ht[(size_t)step * 60 + 40 + j] = v;
                    step += 3; j += 4;
                    if (j >= 20) { j -= 20; step += 1; }
                }
            }
        }
    }
    // the price columns of the head table (|charge price| for the next 20 steps, zero past the horizon; state.py:75-83, :121-129)
    if (a.head_tab) {
        double *ht = a.head_tab + (size_t)ms * (size_t)(T + 1) * a.head_nh;
        int step = lane / 20, cc = lane - step * 20;   // element i = step * 20 + cc advances by 64 = 3 * 20 + 4
        for (int i = lane; i < (T + 1) * 20; i += 64) {
            const int kk = step + cc;
            ht[(size_t)step * a.head_nh + cc] = (kk < T) ? l_cp[kk] : 0.0;
            step += 3; cc += 4;
            if (cc >= 20) { cc -= 20; step += 1; }
        }
    }

    // ---- power setpoints ----
    __syncthreads();
    RF_STAMP(5)
    double *sp_out = RW(double, setpoint) + (size_t)ms * T;
    const int n_sess = min(total, cap);
    if (!c.power_setpoint_enabled || n_sess == 0) {
        for (int t = lane; t < T; t += 64) sp_out[t] = 0.0;
    } else {
        double pmax = 0.0, prmin = INFINITY;
        for (int t = lane; t < T; t += 64) pmax = fmax(pmax, l_cp[t]);
        pmax = rf_wave_max(pmax);
        for (int t = lane; t < T; t += 64) prmin = fmin(prmin, l_cp[t] / pmax);
        prmin = rf_wave_min(prmin);
        const double sd = fmax(prmin, 1e-3);
        double *sp = l_a;   // [T] accumulators (the transformer loop is done with l_a)
        double *prel = l_x; // [T] price relative to the day's maximum: once per step instead of once per session and step (the same quotient)
        for (int t = lane; t < T; t += 64) { sp[t] = 0.0; prel[t] = l_cp[t] / pmax; }
        for (int p = 0; p < P; p++) {   // port by port, a port's sessions in time order: the host's accumulation order
            const int base = l_pbase[p], cnt = max(0, min(l_pcnt[p], cap - base));
            for (int i = 0; i < cnt; i++) {
                const int k = base + i;
                const unsigned long long id = l_id[k];
                const int ta = l_ta[k], td = l_td[k];
                double leaf = 0.0;
                for (int t = lane; t < T; t += 64) {
                    const double w = ev2g_gen_setpoint_weight(rng, id, t, ta, td, prel[t], sd);
                    l_b[t] = w;
                    leaf += w;
                }
                for (int d = 32; d > 0; d >>= 1) leaf += __shfl_xor(leaf, d, 64);   // ev2g_tree64
                const double wsum = fmax(leaf, 1e-12);
                const double need = l_need[k], lo = l_lo[k], hi = l_hi[k];
                for (int t = lane; t < T; t += 64) sp[t] += ev2g_gen_setpoint_load(l_b[t], wsum, need, dt, lo, hi);
            }
        }
        __syncthreads();
        const int kw = ev2g_gen_median_window(dt), left = kw / 2;
        for (int i = lane; i < T + kw - 1; i += 64) { const int t = i - left; l_pad[i] = sp[t < 0 ? 0 : (t >= T ? T - 1 : t)]; }
        __syncthreads();
        for (int t = lane; t < T; t += 64) sp_out[t] = ev2g_gen_median(l_pad, t, kw);
    }
    // the fast path's per-step scalars (ev2g_build_step_table_kernel's rows: one transformer)
    if (a.step_tab) {
        __syncthreads();
        for (int t = lane; t < T; t += 64) {
            double *o8 = a.step_tab + ((size_t)ms * T + t) * 8;
            const size_t i = (size_t)ms * T + t;
            o8[0] = s.price_ch[i]; o8[1] = s.price_dis[i]; o8[2] = s.tr_base[i]; o8[3] = s.tr_maxp[i]; o8[4] = s.tr_minp[i];
            o8[5] = s.setpoint[i];
        }
        // slots 6, 7: the scenario's occupancy / arrival masks of every step -- ev2g_build_occ_mask_kernel's walk (ev2g_device.h), a lane per port slot,
        // over the session windows this wavefront holds in LDS
        int cur = 0, end = 0;
        if (lane < P) { const int p = s.slot_port[lane]; cur = l_pbase[p]; end = cur + max(0, min(l_pcnt[p], cap - cur)); }
        int ta = (cur < end) ? l_ta[cur] : EV2G_INT_MAX, td = (cur < end) ? l_td[cur] : EV2G_INT_MAX;
        for (int t = 0; t < T; t++) {
            const bool occ = (ta <= t) && (t <= td);
            if (occ && t >= td) { cur++; ta = (cur < end) ? l_ta[cur] : EV2G_INT_MAX; td = (cur < end) ? l_td[cur] : EV2G_INT_MAX; }
            const unsigned long long m_occ = __ballot(occ), m_arr = __ballot(ta == t + 1);
            if (lane == 0) {
                double *o8 = a.step_tab + ((size_t)ms * T + t) * 8;
                o8[6] = __longlong_as_double((long long)m_occ); o8[7] = __longlong_as_double((long long)m_arr);
            }
        }
    }
    RF_STAMP(6)
#undef RW
}
