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
// Scope: single-port chargers without a topology file (every shipped config; the fast path's shape), a pool loaded with
// EV2G_FLAG_REFILLABLE (fixed-size session blocks per scenario).  A scenario that draws more sessions than its block holds keeps the
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
};

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

// LDS (dynamic): doubles l_cp[T] | l_a[T] | l_b[T] | l_pad[T + 96] | per staged session need, lo, hi [cap] ; ints t_arr, t_dep [cap],
// base / count per PORT [P]; u64 id [cap]
__host__ __device__ inline size_t ev2g_refill_lds_bytes(int T, int P, int cap) {
    return sizeof(double) * ((size_t)4 * T + 96 + 3 * (size_t)cap) + sizeof(unsigned long long) * (size_t)cap + sizeof(int) * (2 * (size_t)cap + 2 * (size_t)P);
}

__global__ void __launch_bounds__(64) ev2g_refill_kernel(DevScn s, DevState st, RefillArgs a, double *ss_afap) {
    extern __shared__ double rlds[];
    const int lane = threadIdx.x;
    const int T = s.T, P = s.P, R = s.R, cap = a.cap;
    double *l_cp = rlds, *l_a = l_cp + T, *l_b = l_a + T, *l_pad = l_b + T, *l_need = l_pad + T + 96, *l_lo = l_need + cap, *l_hi = l_lo + cap;
    unsigned long long *l_id = (unsigned long long *)(l_hi + cap);
    int *l_ta = (int *)(l_id + cap), *l_td = l_ta + cap, *l_pbase = l_td + cap, *l_pcnt = l_pbase + P;
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

    // ---- prices ----
    for (int t = lane; t < T; t += 64) {
        const double pr = ev2g_gen_price_at(g, rng, dr.price_scale, t);
        l_cp[t] = pr;
        RW(double, price_ch)[(size_t)ms * T + t] = -pr;
        RW(double, price_dis)[(size_t)ms * T + t] = pr * c.discharge_price_factor;
    }

    // ---- sessions, pass 1: how many each port slot draws; prefix over the slots (device order = scenario, slot, arrival) ----
    int carry = 0;
    for (int q0 = 0; q0 < P; q0 += 64) {
        const int q = q0 + lane;
        int n = 0, p = 0;
        if (q < P) {
            p = s.slot_port[q];
            n = ev2g_gen_port_sessions(g, rng, dr.weekend, p, [](int, const Ev2gGenSession &) {});
        }
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

    // ---- sessions, pass 2: draw again and write where they belong ----
    const size_t d0 = (size_t)ms * cap;
    for (int q0 = 0; q0 < P; q0 += 64) {
        const int q = q0 + lane;
        if (q >= P) continue;
        const int p = s.slot_port[q], cs = s.slot_cs[q];
        const int base = l_pbase[p];
        const int n_eff = max(0, min(l_pcnt[p], cap - base));
        const size_t gs = (size_t)ms * P + q;
        RW(int, port_first)[gs] = n_eff > 0 ? (int)(d0 + base) : -1;
        RW(int, port_end)[gs] = n_eff > 0 ? (int)(d0 + base + n_eff) : -1;
        if (n_eff == 0) RW(int2, port_first_win)[gs] = make_int2(EV2G_INT_MAX, EV2G_INT_MAX);
        const double V = s.cs_volt[cs];
        const int ph = s.cs_ph[cs];
        const double sq = sqrt((double)ph);
        const double mp = s.cs_imax[cs] * V * sq / 1000;                       // EV_Charger.get_max_power (ev_charger.py:251-252)
        const double min_cs = s.cs_imin[cs] * V * sq / 1000, max_cs = s.cs_imax[cs] * V * sq / 1000;   // (generate_power_setpoints)
        const double v_gate = s.cs_vk[(size_t)cs * 4 + ph];
        const double pac_min_sp = c.heterogeneous_ev_specs ? 0.0 : c.ev_min_ac_charge_power;
        ev2g_gen_port_sessions(g, rng, dr.weekend, p, [&](int i, const Ev2gGenSession &e) {
            if (i >= n_eff) return;
            const size_t d = d0 + base + i;
            const Ev2gSessFields f = ev2g_gen_session_fields(g, rng, e, a.spec_row);
            RW(int, ss_tarr)[d] = e.t_arr; RW(int, ss_tdep)[d] = e.t_dep; RW(int, ss_ntarr)[d] = EV2G_INT_MAX; RW(int, ss_ntdep)[d] = EV2G_INT_MAX;
            RW(int, ss_phases)[d] = f.phases; RW(int, ss_lut)[d] = f.lut; RW(int, ss_slot)[d] = q;
            RW(double, ss_cap0)[d] = e.cap0; RW(double, ss_B)[d] = e.B; RW(double, ss_des)[d] = f.desired; RW(double, ss_minB)[d] = f.minB;
            RW(double, ss_emerg)[d] = f.min_emerg; RW(double, ss_pacmax)[d] = e.pac; RW(double, ss_pacmin)[d] = f.pac_min;
            RW(double, ss_pdismax)[d] = f.pdis_max; RW(double, ss_pdismin)[d] = f.pdis_min; RW(double, ss_ts)[d] = f.ts; RW(double, ss_tsm)[d] = f.tsm;
            RW(double, ss_etach)[d] = f.eta_ch; RW(double, ss_etadis)[d] = f.eta_dis;
            SessRec r;
            r.B = e.B; r.cap0 = e.cap0; r.des = f.desired; r.minB = f.minB; r.emerg = f.min_emerg;
            r.pacmax = e.pac; r.pdismax = f.pdis_max; r.ts = f.ts; r.tsm = f.tsm; r.eta_ch = f.eta_ch; r.eta_dis = f.eta_dis;
            r.gate_ch = f.pac_min * 1000.0 / v_gate;
            r.gate_dis = f.pdis_min * 1000.0 / v_gate;
            r.v = s.cs_vk[(size_t)cs * 4 + min(ph, f.phases)];
            r.nt_arr = EV2G_INT_MAX; r.nt_dep = EV2G_INT_MAX; r.lut = f.lut; r.pad = 0;
            RW(SessRec, rec)[d] = r;
            if (i > 0) {   // this port's previous session learns its successor's window
                RW(int, ss_ntarr)[d - 1] = e.t_arr; RW(int, ss_ntdep)[d - 1] = e.t_dep;
                RW(SessRec, rec)[d - 1].nt_arr = e.t_arr; RW(SessRec, rec)[d - 1].nt_dep = e.t_dep;
            } else {
                RW(int2, port_first_win)[gs] = make_int2(e.t_arr, e.t_dep);
            }
            const double eff = f.lut >= 0 ? a.lut_rowmax[f.lut] / 100.0 : f.eta_ch;
            ss_afap[d] = rf_afap(e.cap0, e.B, e.pac, mp, eff, e.t_arr, e.t_dep, dt);
            // what the power setpoints need of this session (LDS, by its place in the scenario block)
            const int k = base + i;
            l_id[k] = (unsigned long long)(e.t_arr - 1) * (unsigned long long)P + (unsigned long long)e.port;
            l_ta[k] = e.t_arr; l_td[k] = e.t_dep;
            l_need[k] = (e.B - e.cap0) * (100 + c.power_setpoint_flexiblity) / 100;
            l_lo[k] = fmax(pac_min_sp, min_cs); l_hi[k] = fmin(e.pac, max_cs);
        });
    }

    // ---- transformers ----
    for (int k = 0; k < R; k++) {
        const size_t o = ((size_t)ms * R + k) * T;
        const double capk = c.transformer_max_power;
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
                if (lane == 0) { drs[e * 3 + 0] = ev.es; drs[e * 3 + 1] = ev.ee; drs[e * 3 + 2] = ev.capp; }
            }
        }
        __syncthreads();
        double peak = -INFINITY;
        for (int t = lane; t < T; t += 64) {
            const double mxv = maxp[t], mnv = -capk, il = infl[t];
            const double sol = c.solar_power ? ev2g_gen_solar_at(g, dr.sun, sa, sm, capk, t) : 0.0;
            RW(double, tr_maxp)[o + t] = mxv; RW(double, tr_minp)[o + t] = mnv; RW(double, tr_infl)[o + t] = il; RW(double, tr_solar)[o + t] = sol;
            RW(double, tr_base)[o + t] = il + sol;
            RW(double, tr_lf)[o + t] = c.inflexible_loads ? ev2g_gen_load_forecast_at(g, rng_tr, k, t, il, mnv, mxv) : 0.0;
            RW(double, tr_pvf)[o + t] = c.solar_power ? ev2g_gen_pv_forecast_at(g, rng_tr, k, t, sol) : 0.0;
            peak = fmax(peak, mxv);
        }
        peak = rf_wave_max(peak);
        if (lane == 0) RW(double, tr_peak)[(size_t)ms * R + k] = peak;
    }

    // ---- power setpoints ----
    __syncthreads();
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
        for (int t = lane; t < T; t += 64) sp[t] = 0.0;
        for (int p = 0; p < P; p++) {   // port by port, a port's sessions in time order: the host's accumulation order
            const int base = l_pbase[p], cnt = max(0, min(l_pcnt[p], cap - base));
            for (int i = 0; i < cnt; i++) {
                const int k = base + i;
                const unsigned long long id = l_id[k];
                const int ta = l_ta[k], td = l_td[k];
                double leaf = 0.0;
                for (int t = lane; t < T; t += 64) {
                    const double w = ev2g_gen_setpoint_weight(rng, id, t, ta, td, l_cp[t] / pmax, sd);
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
#undef RW
}
