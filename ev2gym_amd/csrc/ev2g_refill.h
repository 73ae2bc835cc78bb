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
//     done where the session is drawn.  Round 6: the spawn trials of all steps first (a bit per step), then the sessions of all lanes
//     that hit drawn together, round by round;
//   * demand-response events: their slices are applied lane-parallel, `any` / `max` over the slice by ballot / wave max;
//   * power setpoints: round 6 -- only the (session, step of its stay) pairs, packed side by side on the lanes, two batches at a time;
//     a session's weight sum still on the host's 64-leaf tree (ev2g_tree64), the accumulation in the host's session order;
//   * observation tables (head / window table, step table incl. the occupancy masks) from LDS, a row per store (round 6).
// What bounds it (round 6, profiles/r06_refill_*.txt): ~85 KB written per scenario (54 KB of it the head table) by 16 one-wavefront
// workgroups per CU whose chains are a mix of float64 elementary functions and LDS / lane exchanges: 226 -> 180 us per 4096-scenario window
// at cfg2, 361 -> 250 us per 8192 at cfg3.
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

#ifdef EV2G_RF_SUBSTAMPS   // development: cycles of workgroup 0 per part of the setpoint phase, accumulated in dbg[16 + i]
#define RF_SUB(i) { const unsigned long long now_ = __builtin_readcyclecounter(); if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0) a.dbg[16 + (i)] += now_ - sub_t; sub_t = now_; }
#else
#define RF_SUB(i)
#endif
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
//   X is, phase after phase: the spawner's per-step {stay, energy} + uint2 {key, threshold} [T]  ->  actual and forecast (loads - pv) of
//   the transformer in work [T each] and its peak limit [1, at 3T]  ->  the median filter's padded row [T + 96]
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
    double *l_dact = l_x, *l_dfc = l_x + T, *l_pk = l_x + 3 * T;                                // phase 2 (transformers): actual / forecast (loads - pv), the peak limit
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
        // (the fast path's per-step scalars, ev2g_build_step_table_kernel's rows, are written where their values are at hand: slots 0, 1 here,
        //  2..4 with the transformer's series, 5 with the setpoints, 6 / 7 -- the occupancy masks -- at the end)
        if (a.step_tab) { double *o8 = a.step_tab + ((size_t)ms * T + t) * 8; o8[0] = -pr; o8[1] = pr * c.discharge_price_factor; }
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
        if (q < P) p = s.slot_port[q];
#ifndef EV2G_RF_SERIAL_PASS1
        const int t_end = g.T - g.min_stay_steps - 1;
        if (t_end <= 128) {
            // ev2g_gen_port_sessions' walk (ev2g_gen.h) in two parts.  The spawn trials do not depend on the port's history -- one hash round on the
            // step's key each -- so all of them are taken first, the step index uniform over the wavefront (the keys come from LDS as broadcasts,
            // several in flight), and kept as a bit per step.  Then rounds: every lane jumps to its next hit at or behind the step its port is free
            // from (a count-trailing-zeros, no loop) and the sessions of all lanes that have one are drawn TOGETHER -- the two normal draws of a
            // session were executed once per session of the scenario with one lane active (a quarter of this kernel), now once per round (a port
            // draws ~0.7 sessions per episode: ~4 rounds).  Same trials, same draws, same order per port.
            unsigned long long m0 = 0ull, m1 = 0ull;
            const int e0 = min(t_end, 64);
#pragma unroll 4
            for (int t = 2; t < e0; t++) { const uint2 kt = l_kt[t]; if (kt.y != 0u && ev2g_gen_spawn_trial(kt.x, kt.y, p)) m0 |= 1ull << t; }
#pragma unroll 4
            for (int t = 64; t < t_end; t++) { const uint2 kt = l_kt[t]; if (kt.y != 0u && ev2g_gen_spawn_trial(kt.x, kt.y, p)) m1 |= 1ull << (t - 64); }
            if (q >= P) { m0 = 0ull; m1 = 0ull; }
            int free_from = 2;
            for (;;) {
                const unsigned long long a0 = (free_from < 64) ? ((m0 >> free_from) << free_from) : 0ull;
                const unsigned long long a1 = (free_from <= 64) ? m1 : ((free_from < 128) ? ((m1 >> (free_from - 64)) << (free_from - 64)) : 0ull);
                const bool hit = (a0 | a1) != 0ull;
                if (__ballot(hit) == 0ull) break;
                if (hit) {
                    const int t = a0 ? (__ffsll((long long)a0) - 1) : (63 + __ffsll((long long)a1));
                    Ev2gGenSession e;
                    if (ev2g_gen_make_session(g, rng, fleet, share_sum, t, p, l_stay[t], l_emean[t], &e)) {
                        free_from = e.t_dep + 2;
                        if (n < EV2G_RF_K) { l_spawn[p * EV2G_RF_K + n] = (unsigned char)(e.t_arr - 1); if (multi) l_dep[p * EV2G_RF_K + n] = (unsigned char)e.t_dep; }
                        n++;
                    } else {
                        free_from = t + 1;
                    }
                }
            }
        } else if (q < P) {
            n = ev2g_gen_port_sessions(g, rng, fleet, share_sum, p, tab, [&](int i, const Ev2gGenSession &e) { if (i < EV2G_RF_K) { l_spawn[p * EV2G_RF_K + i] = (unsigned char)(e.t_arr - 1); if (multi) l_dep[p * EV2G_RF_K + i] = (unsigned char)e.t_dep; } });
        }
        if (q < P && multi) { l_pslot[p] = q; l_rcnt[q] = 0; l_free[p] = 0; }
#else
        if (q < P) {
            n = ev2g_gen_port_sessions(g, rng, fleet, share_sum, p, tab, [&](int i, const Ev2gGenSession &e) { if (i < EV2G_RF_K) { l_spawn[p * EV2G_RF_K + i] = (unsigned char)(e.t_arr - 1); if (multi) l_dep[p * EV2G_RF_K + i] = (unsigned char)e.t_dep; } });
            if (multi) { l_pslot[p] = q; l_rcnt[q] = 0; l_free[p] = 0; }
        }
#endif
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
            if (a.step_tab && k == 0) { double *o8 = a.step_tab + ((size_t)ms * T + t) * 8; o8[2] = il + sol; o8[3] = mxv; o8[4] = mnv; }   // (one transformer on the fast path)
            const double lfv = c.inflexible_loads ? ev2g_gen_load_forecast_at(g, rng_tr, k, t, il, mnv, mxv) : 0.0;
            const double pvv = c.solar_power ? ev2g_gen_pv_forecast_at(g, rng_tr, k, t, sol) : 0.0;
            RW(double, tr_lf)[o + t] = lfv; RW(double, tr_pvf)[o + t] = pvv;
            l_dact[t] = il - sol; l_dfc[t] = lfv - pvv;   // the two (loads - pv) series the observation windows below are cut from
            peak = fmax(peak, mxv);
        }
        peak = rf_wave_max(peak);
        if (lane == 0) { RW(double, tr_peak)[(size_t)ms * R + k] = peak; l_pk[0] = peak * 1.0; }
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
            // One ROW per iteration, a lane per column (the head table's 60: |price|, loads - pv, power limits -- 20 each; the window table's 40
            // without the prices): a row is one contiguous 480-byte store, every line written once.  (Rounds 3-5 wrote the 20-column strips in
            // separate loops of 64 consecutive strip elements: 160-byte pieces of four rows per store, every line of a row completed by three
            // stores far apart in time -- with 16 wavefronts per CU writing 54 KB each, the partially written lines left the L2 in between.)
            {
                double *rows = ht ? ht : wt;
                const int ncol = ht ? 60 : 40, c_lpv = ht ? 20 : 0, c_lim = c_lpv + 20;
                const int typ = (lane < c_lpv) ? 0 : ((lane < c_lim) ? 1 : 2);
                const int j = lane - ((typ == 0) ? 0 : ((typ == 1) ? c_lpv : c_lim));   // column inside the strip (lanes behind the row: computed, not stored)
                if (rows) {
                    // Round 6 (third session): the rows 0 .. T-2 of a transformer with at most one event -- every shipped config -- by ONE LDS read per lane
                    // and row.  A lane's column is a slide over one series (|price|, the forecast (loads - pv), the actual one for the window's first
                    // column, the constant peak limit: base + stride * min(kk, T-1), stride 0 for the limit) with one override inside a lane-constant
                    // range of kk = step + j: zero past the horizon for a price column; the event's limit for kk in [event start, event end) once the
                    // event is known (step + ahead >= start  <=>  kk >= start - ahead + j).  That is transformer.py:142-188 / state.py:75-83 evaluated
                    // like the general loop below does (which keeps rows T-1 and T, where the padding switches to the actual series, and any
                    // transformer with more events): ~12 instead of ~45 instructions per row -- the phase was 71 k of the scenario's ~200 k cycles.
                    int s_begin = 0;
                    if (nd <= 1 && T >= 2) {   // (uniform)
                        const double *src = (typ == 0) ? l_cp : ((typ == 1) ? ((j == 0) ? l_dact : l_dfc) : l_pk);
                        const int stride = (typ == 2) ? 0 : 1;
                        int lo = 0x7fffffff, hi = 0x7fffffff;
                        double ov = 0.0;
                        if (typ == 0) lo = T;
                        if (typ == 2 && nd == 1) { lo = max(d_es[0], d_es[0] - ahead + j); hi = d_ee[0]; ov = d_lim[0]; }
                        if (lane < ncol) {   // (one exec mask around the whole loop, four rows' reads in flight: per row, the branch and the wait were the chain)
                            double *out = rows + lane;
                            int kk = j, step = 0;
                            auto put = [&](double *o, double v) {
#ifndef EV2G_RF_NO_NT_TABLES
                                __builtin_nontemporal_store(v, o);
#else
                                *o = v;
#endif
                            };
                            for (; step + 4 <= T - 1; step += 4, kk += 4, out += 4 * ncol) {
                                double x[4];
#pragma unroll
                                for (int u = 0; u < 4; u++) x[u] = src[min(kk + u, T - 1) * stride];
#pragma unroll
                                for (int u = 0; u < 4; u++) put(out + u * ncol, (kk + u >= lo && kk + u < hi) ? ov : x[u]);
                            }
                            for (; step <= T - 2; step++, kk++, out += ncol) put(out, (kk >= lo && kk < hi) ? ov : src[min(kk, T - 1) * stride]);
                        }
                        s_begin = T - 1;
                    }
#pragma unroll 2
                    for (int step = s_begin; step <= T; step++) {
                        const int kk = step + j, ke = min(kk, T - 1);
                        // |charge price| of the next 20 steps, zero past the horizon (state.py:75-83, :121-129)
                        const double vp = (kk < T) ? l_cp[ke] : 0.0;
                        // (loads - pv) window, Transformer.get_load_pv_forecast transformer.py:173-188: the actual series for the current step (j == 0) and
                        // behind the horizon of the last one, the forecast otherwise; the same element [min(kk, T - 1)] of either pair (1.0 * x == x)
                        const bool actual = (kk < T) ? (j == 0) : (step >= T - 1);
                        const double vl = (actual ? l_dact : l_dfc)[ke];
                        // power limits, Transformer.get_power_limits transformer.py:142-171
                        double vm = peak * 1.0;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            if (e < nd) {   // (uniform)
                                const int es = d_es[e], ee = d_ee[e];
                                if (step + ahead >= es && ee >= step) {
                                    int aa, bb;
                                    if (step > es) { aa = 0; bb = ee - step; } else { aa = es - step; bb = ee - step; }
                                    if (aa < 0) aa = -aa;
                                    if (bb < 0) bb = -bb;
                                    if (j >= aa && j < bb) vm = d_lim[e];
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
                                if (j >= aa && j < bb) vm = peak - peak * l_dr[e * 3 + 2] / 100.0;
                            }
                        }
                        const double v = (typ == 0) ? vp : ((typ == 1) ? vl : vm);
#ifndef EV2G_RF_NO_NT_TABLES   // streaming stores: 54 KB per scenario written once and read by a later episode's step kernel (-3.5 % of the window, tools/r6/gpu_rf6.sh)
                        if (lane < ncol) __builtin_nontemporal_store(v, &rows[(size_t)step * ncol + lane]);
#else
                        if (lane < ncol) rows[(size_t)step * ncol + lane] = v;
#endif
                    }
                }
            }
        }
    }
    // the price columns of the head table (|charge price| for the next 20 steps, zero past the horizon; state.py:75-83, :121-129)
    if (a.head_tab && a.head_nh != 60) {   // (the 60-column table got them with its rows above)
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
#ifdef EV2G_RF_SUBSTAMPS
    unsigned long long sub_t = __builtin_readcyclecounter();
#endif
    const int n_sess = min(total, cap);
    if (!c.power_setpoint_enabled || n_sess == 0) {
        for (int t = lane; t < T; t += 64) { sp_out[t] = 0.0; if (a.step_tab) a.step_tab[((size_t)ms * T + t) * 8 + 5] = 0.0; }
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
#ifndef EV2G_RF_SERIAL_SETPOINTS
        // A session's weights are zero outside its stay (ev2g_gen_setpoint_weight) and a zero weight adds 0.0 to the accumulators, so only the
        // (session, step-of-its-stay) pairs need the normal draw and the division: ~30 of a session's 112 steps.  Consecutive sessions (in the
        // host's accumulation order) are PACKED into the wavefront, a lane per pair, as long as their stays fit 64 lanes together, and TWO such
        // batches are in work at a time (their draws and divisions are independent chains the wavefront interleaves): this phase is a chain of
        // dependent operations per session, not a number of instructions.  Sessions are described by a lane each (read by v_readlane, no LDS
        // round trip per member); the accumulators sp[lane], sp[lane + 64] stay in registers.  Bit for bit the host's values: a session's weight
        // sum is still taken on the 64-leaf tree with leaf (t mod 64) -- each member's weights are fetched from LDS to the lanes of their leaves,
        // zeros elsewhere, the butterflies in the host's order 32 .. 1 -- and the accumulators receive the sessions one after the other.
        int *l_ord = (int *)l_dr;                  // [<= 64] sessions in accumulation order (the events' LDS is free by now)
        double *l_wb = l_x + T;                    // [2][64] weights (later: loads) of the pairs in work (the transformer's rows are free by now)
        const int ns_u = __builtin_amdgcn_readfirstlane(n_sess);   // (the same on every lane; said so for the compiler)
        bool packed_ok = ns_u <= 64 && T <= 128 && T >= 16;   // (a lane per session descriptor; two accumulators per lane; l_wb's 128 doubles inside X's 2 T + 96 free ones)
        int d_k = 0, d_w0 = 0, d_L = 0;
        if (packed_ok) {
            int carry_o = 0;
            for (int p0 = 0; p0 < P; p0 += 64) {   // rank of every port's first session in the accumulation order (generator ports ascending)
                const int p = p0 + lane;
                int base = 0, cnt = 0;
                if (p < P) { base = l_pbase[p]; cnt = max(0, min(l_pcnt[p], cap - base)); }
                const int incl = rf_wave_incl_scan(cnt, lane);
                const int r0 = carry_o + incl - cnt;
                for (int i = 0; i < cnt; i++) l_ord[r0 + i] = base + i;
                carry_o += __shfl(incl, 63, 64);
            }
            __syncthreads();
            bool plain = true;
            if (lane < ns_u) {   // lane j describes session j of the order: its record, the first step of its stay, the stay's length
                d_k = l_ord[lane];
                d_w0 = l_ta[d_k] + 1;
                d_L = max(min(l_td[d_k], T) - d_w0, 0);
                plain = d_L <= 64 && l_hi[d_k] >= 0.0;   // (a negative upper bound would make fmin(0, hi) nonzero outside the stay: the serial walk below)
            }
            packed_ok = __ballot(!plain) == 0ull;
        }
        RF_SUB(0)
        if (packed_ok) {
            double acc0 = 0.0, acc1 = 0.0;   // sp[lane], sp[lane + 64]
            int j = 0;
            while (j < ns_u) {
                // ---- two batches: sessions bj[u] .. bj[u] + bn[u] - 1 with their stays side by side on the lanes ----
                int bj[2], bn[2], myk[2], myt[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    int nb = 0, tot = 0;
                    bj[u] = j; myk[u] = -1; myt[u] = 0;
                    while (j < ns_u) {
                        const int L = __builtin_amdgcn_readlane(d_L, j);
                        if (tot + L > 64) break;
                        const int k = __builtin_amdgcn_readlane(d_k, j), w0 = __builtin_amdgcn_readlane(d_w0, j);
                        if (lane >= tot && lane < tot + L) { myk[u] = k; myt[u] = w0 + (lane - tot); }
                        tot += L; nb++; j++;
                    }
                    bn[u] = nb;
                }
                RF_SUB(1)
                // ---- one weight per lane and batch (inside the stay: ev2g_gen_setpoint_weight's `win`) ----
                double w[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const double x = fabs(rng.normal(EV2G_RS_SETPOINT, l_id[max(myk[u], 0)], (uint64_t)myt[u], 1 - prel[myt[u]], sd));
                    w[u] = (myk[u] >= 0) ? x : 0.0;
                    l_wb[u * 64 + lane] = w[u];
                }
                __syncthreads();
                RF_SUB(2)
                // ---- every member's weight sum on the host's tree: leaf v holds the weight of the member's step t with t mod 64 == v ----
                double my_wsum[2] = {1.0, 1.0};
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    int off = 0;
                    for (int b = 0; b < bn[u]; b++) {
                        const int jj = bj[u] + b;
                        const int L = __builtin_amdgcn_readlane(d_L, jj), k = __builtin_amdgcn_readlane(d_k, jj), w0 = __builtin_amdgcn_readlane(d_w0, jj);
                        const int dl = (lane - w0) & 63;       // this leaf's step of the stay is w0 + dl
                        double leaf = (dl < L) ? l_wb[u * 64 + off + dl] : 0.0;
                        leaf += __shfl_xor(leaf, 32, 64); leaf += __shfl_xor(leaf, 16, 64);          // ev2g_tree64: 32, 16, ...
                        leaf += dpp_mov_f64<0x128>(leaf); leaf += xor4_f64(leaf); leaf += xor2_f64(leaf); leaf += xor1_f64(leaf);   // ... 8, 4, 2, 1
                        if (myk[u] == k) my_wsum[u] = fmax(leaf, 1e-12);
                        off += L;
                    }
                }
                RF_SUB(3)
                // ---- one load per lane and batch ----
                double val[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int kk = max(myk[u], 0);
                    const double x = ev2g_gen_setpoint_load(w[u], my_wsum[u], l_need[kk], dt, l_lo[kk], l_hi[kk]);
                    val[u] = (myk[u] >= 0) ? x : 0.0;
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < 2; u++) l_wb[u * 64 + lane] = val[u];
                __syncthreads();
                RF_SUB(4)
                // ---- the sessions add theirs to the accumulators one after the other ----
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    int off = 0;
                    for (int b = 0; b < bn[u]; b++) {
                        const int jj = bj[u] + b;
                        const int L = __builtin_amdgcn_readlane(d_L, jj), w0 = __builtin_amdgcn_readlane(d_w0, jj);
                        const int e0 = lane - w0, e1 = lane + 64 - w0;
                        if (e0 >= 0 && e0 < L) acc0 += l_wb[u * 64 + off + e0];
                        if (e1 >= 0 && e1 < L) acc1 += l_wb[u * 64 + off + e1];
                        off += L;
                    }
                }
                __syncthreads();
                RF_SUB(5)
            }
            if (lane < T) sp[lane] = acc0;
            if (lane + 64 < T) sp[lane + 64] = acc1;
        } else
#endif
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
        RF_SUB(6)
        const int kw = ev2g_gen_median_window(dt), left = kw / 2;
        for (int i = lane; i < T + kw - 1; i += 64) { const int t = i - left; l_pad[i] = sp[t < 0 ? 0 : (t >= T ? T - 1 : t)]; }
        __syncthreads();
        for (int t = lane; t < T; t += 64) { const double v = ev2g_gen_median(l_pad, t, kw); sp_out[t] = v; if (a.step_tab) a.step_tab[((size_t)ms * T + t) * 8 + 5] = v; }
    }
    RF_SUB(7)
    // the fast path's occupancy / arrival masks of every step (step-table slots 6, 7; ev2g_build_occ_mask_kernel's values, ev2g_device.h)
    if (a.step_tab) {
        __syncthreads();
        RF_SUB(8)
        if (T <= 128) {
            // A lane per port slot gathers its sessions' stays as a bit per step (occupied: t_arr .. t_dep; arriving at the end of step t: t_arr == t + 1),
            // then one ballot per step turns the rows into the per-step masks over the ports; lane (t mod 64) keeps step t's pair and stores it.
            // (The walk over the steps with a session cursor per lane -- an LDS round trip in the chain of every departure -- was a sixth of this kernel.)
            unsigned long long o0 = 0ull, o1 = 0ull, r0 = 0ull, r1 = 0ull;
            if (lane < P) {
                const int p = s.slot_port[lane];
                const int cur = l_pbase[p], end = cur + max(0, min(l_pcnt[p], cap - cur));
                for (int i = cur; i < end; i++) {
                    const int ta = l_ta[i], td = l_td[i];          // 1 <= ta <= td < T
                    const int lo0 = min(ta, 64), hi0 = min(td + 1, 64);          // [ta, td] cut to steps 0..63
                    if (hi0 > lo0) o0 |= ((hi0 - lo0 >= 64) ? ~0ull : ((1ull << (hi0 - lo0)) - 1ull)) << lo0;
                    const int lo1 = max(ta, 64) - 64, hi1 = max(td + 1, 64) - 64;   // ... and to steps 64..127
                    if (hi1 > lo1) o1 |= ((hi1 - lo1 >= 64) ? ~0ull : ((1ull << (hi1 - lo1)) - 1ull)) << lo1;
                    if (ta - 1 < 64) r0 |= 1ull << (ta - 1); else r1 |= 1ull << (ta - 1 - 64);
                }
            }
            unsigned long long k_occ = 0ull, k_arr = 0ull;
            const int n0 = min(T, 64);
            for (int t = 0; t < n0; t++) {
                const unsigned long long m_occ = __ballot((o0 >> t) & 1ull), m_arr = __ballot((r0 >> t) & 1ull);
                if (lane == t) { k_occ = m_occ; k_arr = m_arr; }
            }
            if (lane < n0) {
                double *o8 = a.step_tab + ((size_t)ms * T + lane) * 8;
                o8[6] = __longlong_as_double((long long)k_occ); o8[7] = __longlong_as_double((long long)k_arr);
            }
            for (int t = 64; t < T; t++) {
                const unsigned long long m_occ = __ballot((o1 >> (t - 64)) & 1ull), m_arr = __ballot((r1 >> (t - 64)) & 1ull);
                if (lane == t - 64) { k_occ = m_occ; k_arr = m_arr; }
            }
            if (lane + 64 < T) {
                double *o8 = a.step_tab + ((size_t)ms * T + lane + 64) * 8;
                o8[6] = __longlong_as_double((long long)k_occ); o8[7] = __longlong_as_double((long long)k_arr);
            }
        } else {
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
    }
    RF_SUB(9)
    RF_STAMP(6)
#undef RW
}
