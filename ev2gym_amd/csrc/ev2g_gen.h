// ev2g_gen.h -- the scenario generator as plain C++: one scenario = everything EV2Gym.reset() draws for one episode
// (EV_spawner utils.py:477-557, spawn_single_EV :177-345, load_transformers loaders.py:227-296 + transformer.py:80-256,
// load_electricity_prices loaders.py:392-461, generate_power_setpoints utils.py:664-757).
//
// Same model as ev2gym_amd/scenario_gen.py (its docstring says what is fitted to what; the hour-of-day tables and fleet classes
// below are the same numbers, tests/test_host_logic.py compares the two copies and holds BOTH generators to the reference's spawn
// statistics): statistically matched to the reference, not its RNG streams.  What differs from the numpy version is the random
// number source: every draw is a pure function of (seed, scenario index, stream, counters) -- a counter-based generator -- so a
// scenario does not depend on which thread (or, later, which wavefront) produces it, nor on how many scenarios are drawn with it.
// Everything per-scenario is written against caller-provided buffers and qualified EV2G_HD, so that the same code can be compiled
// for the device; today the host runs it (ev2g_generate, one thread per slice of the scenarios).
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/ev2g.h"

#if defined(__HIPCC__)
#define EV2G_HD __host__ __device__ inline
#else
#define EV2G_HD inline
#endif

// ---- tables (the role of the reference's distribution-of-arrival / time-of-connection / energy-demand data) ---------------
// [kind][24]: kind = 0 workplace, 1 public, 2 private, 3 public weekend, 4 private weekend; arrivals per port per hour in percent,
// mean stay in hours, mean required energy in kWh for an EV arriving in that hour.
#define EV2G_GEN_N_KINDS 5
static const double EV2G_GEN_RATE[EV2G_GEN_N_KINDS][24] = {
    {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.023, 1.241, 4.008, 8.163, 3.732, 1.516, 1.371, 1.463, 1.412, 0.945, 0.716, 0.568, 0.215, 0.091, 0.0, 0.0, 0.0, 0.0},
    {0.153, 0.153, 0.153, 0.153, 0.153, 0.041, 0.05, 0.156, 0.86, 2.444, 1.525, 1.113, 1.251, 1.322, 1.261, 1.221, 1.221, 1.272, 1.731, 2.13, 1.883, 0.649, 0.741, 0.934},
    {0.166, 0.166, 0.166, 0.166, 0.166, 0.03, 0.018, 0.079, 0.113, 0.426, 0.277, 0.307, 0.387, 0.587, 0.627, 0.639, 0.495, 0.69, 2.806, 3.265, 2.131, 1.17, 1.69, 1.514},
    {0.163, 0.163, 0.163, 0.163, 0.163, 0.035, 0.053, 0.083, 0.11, 0.541, 0.957, 1.38, 1.952, 1.951, 2.029, 2.049, 2.025, 1.987, 1.515, 1.608, 1.253, 0.895, 0.638, 1.424},
    {0.214, 0.214, 0.214, 0.214, 0.214, 0.021, 0.029, 0.08, 0.081, 0.231, 0.486, 0.731, 1.396, 1.606, 1.679, 1.604, 0.916, 1.576, 2.605, 1.286, 1.485, 1.02, 1.231, 0.526},
};
static const double EV2G_GEN_STAY[EV2G_GEN_N_KINDS][24] = {
    {8.0, 8.0, 8.0, 8.0, 8.0, 7.93, 7.93, 8.63, 8.63, 7.35, 6.75, 4.34, 4.04, 3.56, 3.36, 2.41, 2.41, 2.51, 2.51, 2.62, 2.62, 3.0, 3.0, 3.0},
    {8.0, 8.0, 8.0, 8.0, 8.0, 5.51, 5.51, 5.31, 5.31, 4.44, 4.44, 2.86, 2.86, 2.98, 2.98, 2.92, 2.92, 8.4, 8.4, 11.77, 11.77, 10.91, 9.91, 10.96},
    {8.0, 8.0, 8.0, 8.0, 8.0, 5.32, 5.32, 4.32, 4.32, 3.16, 3.66, 2.48, 4.48, 3.69, 4.69, 10.58, 10.58, 13.9, 13.4, 13.07, 12.07, 11.09, 10.59, 8.91},
    {8.0, 8.0, 8.0, 8.0, 8.0, 5.33, 5.33, 5.63, 5.63, 4.2, 4.2, 2.86, 2.86, 3.01, 3.01, 2.82, 3.32, 7.28, 8.28, 12.0, 12.0, 10.83, 9.83, 12.11},
    {8.0, 8.0, 8.0, 8.0, 8.0, 4.35, 4.35, 4.31, 4.31, 3.41, 3.41, 3.16, 4.16, 3.72, 4.72, 9.71, 10.71, 14.4, 14.4, 13.11, 12.11, 11.47, 10.47, 8.3},
};
static const double EV2G_GEN_ENERGY[EV2G_GEN_N_KINDS] = {14.35, 14.11, 22.0, 13.9, 19.37};

// fleets: share, battery kWh, max AC kW (+ efficiency % at 6, 8, .. 16 A for the V2G fleet)
#define EV2G_GEN_FLEET_MAX 8
static const double EV2G_FLEET_V2G[EV2G_GEN_FLEET_MAX][3] = {
    {0.22, 57.5, 11.0}, {0.18, 57.5, 11.0}, {0.13, 64.8, 11.0}, {0.11, 58.0, 11.0}, {0.10, 58.0, 11.0}, {0.09, 64.0, 11.0}, {0.09, 46.3, 7.4}, {0.08, 77.0, 11.0}};
static const double EV2G_FLEET_V2G_ETA[EV2G_GEN_FLEET_MAX][6] = {
    {87, 87, 90, 90, 90, 90}, {87, 87, 90, 90, 90, 90}, {90, 90, 90, 90, 90, 90}, {87, 87, 90, 90, 90, 90},
    {87, 90, 90, 90, 90, 90}, {90, 90, 93, 93, 93, 93}, {84, 87, 90, 90, 90, 90}, {90, 90, 90, 90, 90, 90}};
static const double EV2G_FLEET_EV_PHEV[EV2G_GEN_FLEET_MAX][3] = {
    {0.26, 8.0, 3.7}, {0.10, 14.5, 3.7}, {0.03, 39.0, 3.6}, {0.045, 46.3, 7.4}, {0.035, 52.0, 22.0}, {0.32, 57.7, 11.0}, {0.12, 64.5, 11.0}, {0.07, 76.0, 11.0}};

// ---- counter-based random numbers ------------------------------------------------------------------------------------------
enum { EV2G_RS_HOUR = 1, EV2G_RS_PRICE, EV2G_RS_WEEKEND, EV2G_RS_SPAWN, EV2G_RS_SESSION, EV2G_RS_TR, EV2G_RS_SOLAR_ENV, EV2G_RS_DR, EV2G_RS_FC, EV2G_RS_SETPOINT };

EV2G_HD uint64_t ev2g_mix64(uint64_t z) {   // splitmix64 finaliser
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct Ev2gRng {
    uint64_t key;   // mix of (seed, scenario)
    EV2G_HD uint64_t bits(uint64_t stream, uint64_t a, uint64_t b) const { return ev2g_mix64(ev2g_mix64(ev2g_mix64(key ^ (stream * 0xD1342543DE82EF95ull)) + a) + b * 0xA24BAED4963EE407ull); }
    EV2G_HD double uni(uint64_t stream, uint64_t a, uint64_t b) const { return (double)(bits(stream, a, b) >> 11) * (1.0 / 9007199254740992.0); }   // [0, 1)
    EV2G_HD double uni(uint64_t stream, uint64_t a, uint64_t b, double lo, double hi) const { return lo + (hi - lo) * uni(stream, a, b); }
    EV2G_HD long long integers(uint64_t stream, uint64_t a, uint64_t b, long long lo, long long hi) const {   // lo <= x < hi
        const long long n = hi - lo;
        return lo + (long long)floor(uni(stream, a, b) * (double)n);
    }
    // Box-Muller on two counters of their own: bit 62 / 63 of `b` are never set by a uni() / integers() call site, so a normal draw
    // shares no counter with any other draw of the same (stream, a) -- draws documented as independent are independent
    EV2G_HD double normal(uint64_t stream, uint64_t a, uint64_t b, double mean, double sd) const {
        const double u1 = 1.0 - uni(stream, a, b | (1ull << 62)), u2 = uni(stream, a, b | (1ull << 63));
        return mean + sd * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    }
};
EV2G_HD Ev2gRng ev2g_rng(uint64_t seed, uint64_t scenario) { return Ev2gRng{ev2g_mix64(ev2g_mix64(seed) ^ (scenario * 0x9E3779B97F4A7C15ull + 0x1234567ull))}; }

EV2G_HD double ev2g_gen_interp24(const double *tab, double hour_of_day) {   // np.interp over the 24 hourly values, cyclic
    double h = fmod(hour_of_day, 24.0);
    if (h < 0) h += 24.0;
    const int i = (int)floor(h);
    const double f = h - (double)i;
    const double a = tab[i % 24], b = tab[(i + 1) % 24];
    return a + (b - a) * f;
}
// a[start:stop] of a length-n Python sequence: [*s0, *s1) (empty when *s0 >= *s1)
EV2G_HD void ev2g_py_slice(int start, int stop, int n, int *s0, int *s1) {
    if (start < 0) { start += n; if (start < 0) start = 0; } else if (start > n) start = n;
    if (stop < 0) { stop += n; if (stop < 0) stop = 0; } else if (stop > n) stop = n;
    *s0 = start; *s1 = stop;
}
EV2G_HD double ev2g_round_dec(double x, double scale) { return rint(x * scale) / scale; }   // np.round(x, k): round-half-even at 10^-k

// The per-run constants every scenario shares (derived once from ev2g_gen_config)
struct Ev2gGenRun {
    const ev2g_gen_config *c;
    int T, dt, C, P, R, npc_max;
    int hour;              // start hour (random_hour: drawn once per run, ev2gym_env.py:131-133)
    int min_stay_steps;
    int steps_ahead;
    int n_dr;              // event slots per transformer
    int n_fleet;
    bool lut_fleet;
    uint64_t seed;
    const double *pv_series;   // tab_pv brought to the simulation timescale and smoothed (ev2g_gen_pv_series), two years long; or null
    long long pv_per_day;      // its entries per day
};

// ---- the fleet a session's car model is drawn from: the spec file's models, or a built-in representative fleet -------------------
struct Ev2gFleet {
    int n;
    const double (*builtin)[3];   // (share, battery kWh, max AC kW), when the config carries no spec file
    const ev2g_gen_config *c;
    EV2G_HD double share(int i) const { return c->n_ev_specs > 0 ? c->spec_registrations[i] : builtin[i][0]; }
    EV2G_HD double battery(int i) const { return c->n_ev_specs > 0 ? c->spec_battery_capacity[i] : builtin[i][1]; }
    EV2G_HD double pac(int i) const { return c->n_ev_specs > 0 ? c->spec_max_ac_charge_power[i] : builtin[i][2]; }
};
EV2G_HD Ev2gFleet ev2g_fleet(const Ev2gGenRun &g) {
    const ev2g_gen_config &c = *g.c;
    if (c.n_ev_specs > 0) return Ev2gFleet{c.n_ev_specs, nullptr, g.c};
    return Ev2gFleet{g.n_fleet, (c.fleet_with_efficiency_tables || c.fleet != 1) ? EV2G_FLEET_V2G : EV2G_FLEET_EV_PHEV, g.c};
}
// minute of the day of step t (unwrapped hours -> 0..1439)
EV2G_HD int ev2g_minute_of_day(const Ev2gGenRun &g, int t) {
    const long long m = (long long)g.hour * 60 + g.c->minute + (long long)t * g.dt;
    return (int)(((m % 1440) + 1440) % 1440);
}

// ---- one scenario: prices -----------------------------------------------------------------------------------------------------
EV2G_HD void ev2g_gen_prices(const Ev2gGenRun &g, const Ev2gRng &r, double *charge_price, double *discharge_price) {
    const ev2g_gen_config &c = *g.c;
    const double scale = r.uni(EV2G_RS_PRICE, 0, 0, 0.6, 1.6);
    int last_h = -1;
    double hp = 0.0;
    for (int t = 0; t < g.T; t++) {
        const double sh = g.hour + c.minute / 60.0 + t * (double)g.dt / 60.0;
        const int h = (int)floor(sh);
        if (h != last_h) {   // hourly day-ahead-like curve, EUR/MWh
            const double hh = (double)(h % 24);
            const double base = 75 + 40 * sin((hh - 7) / 24 * 6.283185307179586) + 30 * sin((hh - 17) / 12 * 6.283185307179586);
            hp = ev2g_round_dec(fmax(base * scale + r.normal(EV2G_RS_PRICE, 1, (uint64_t)h, 0.0, 12.0), 3.0), 100.0);
            last_h = h;
        }
        const double p = hp / 1000.0;
        charge_price[t] = -p;                                  // loaders.py:439-442
        discharge_price[t] = p * c.discharge_price_factor;
    }
}

// ---- one scenario: EV sessions, in EVs_profiles order (arrival step, then port) ---------------------------------------------------
struct Ev2gGenSession { int port, t_arr, t_dep, model; double B, pac, cap0; };

EV2G_HD int ev2g_gen_sessions(const Ev2gGenRun &g, const Ev2gRng &r, bool weekend, int *free_from /*[P] scratch*/, Ev2gGenSession *out, int cap) {
    const ev2g_gen_config &c = *g.c;
    const int kind = c.scenario + ((weekend && c.scenario != 0) ? 2 : 0);   // 0 workplace, 1 public, 2 private, 3 public weekend, 4 private weekend
    const Ev2gFleet fleet = ev2g_fleet(g);
    double share_sum = 0.0;
    for (int i = 0; i < fleet.n; i++) share_sum += fleet.share(i);
    for (int p = 0; p < g.P; p++) free_from[p] = 0;
    int n = 0;
    for (int t = 2; t < g.T - g.min_stay_steps - 1; t++) {
        const double hod = g.hour + c.minute / 60.0 + t * (double)g.dt / 60.0;
        double rate, stay_mean, e_mean;
        if (c.tab_arrival_week) {   // the reference's own tables, looked up its way (utils.py:199-233, 505-528)
            const int mod = ev2g_minute_of_day(g, t), hh = mod / 60;
            rate = (weekend ? c.tab_arrival_weekend : c.tab_arrival_week)[mod / 15];
            if (c.scenario == 0 && (hh < 6 || hh > 18)) rate = 0.0;
            rate *= (g.dt / 60.0) * c.spawn_multiplier;
            stay_mean = c.tab_stay[mod / 30]; e_mean = c.tab_energy[mod / 30];
        } else {
            rate = ev2g_gen_interp24(EV2G_GEN_RATE[kind], hod) * (g.dt / 60.0) * c.spawn_multiplier;   // percent per step
            stay_mean = ev2g_gen_interp24(EV2G_GEN_STAY[kind], hod); e_mean = EV2G_GEN_ENERGY[kind];
        }
        if (!(rate > 0.0)) continue;
        for (int p = 0; p < g.P; p++) {
            if (free_from[p] > t) continue;   // occupied, or inside the 3-step-empty rule (utils.py:534-552)
            const uint64_t id = (uint64_t)t * (uint64_t)g.P + (uint64_t)p;
            if (!(r.uni(EV2G_RS_SPAWN, id, 0) * 100.0 < rate)) continue;
            double req = r.normal(EV2G_RS_SESSION, id, 0, e_mean, 0.5 * e_mean);
            if (req < 5) req = (double)r.integers(EV2G_RS_SESSION, id, 10, 5, 10);
            int model = 0;
            double B = c.ev_battery_capacity, pac = c.ev_max_ac_charge_power;
            if (c.heterogeneous_ev_specs) {
                const double u = r.uni(EV2G_RS_SESSION, id, 11) * share_sum;
                double acc = 0.0;
                model = fleet.n - 1;
                for (int i = 0; i < fleet.n; i++) { acc += fleet.share(i); if (u < acc) { model = i; break; } }
                B = fleet.battery(model); pac = fleet.pac(model);
            }
            const long long Bi = (long long)B > 2 ? (long long)B : 2;
            double cap0 = (B < req) ? (double)r.integers(EV2G_RS_SESSION, id, 12, 1, Bi) : B - req;
            if (cap0 > c.ev_desired_capacity * B) cap0 = (double)r.integers(EV2G_RS_SESSION, id, 13, 1, Bi);
            if (cap0 < c.ev_min_battery_capacity && B > 2 * c.ev_min_battery_capacity) cap0 = c.ev_min_battery_capacity;
            double stay = r.normal(EV2G_RS_SESSION, id, 2, stay_mean, 0.2 * stay_mean) * 60.0 / g.dt + 1;
            if (stay < g.min_stay_steps) stay = g.min_stay_steps;
            if (stay + t + 4 >= g.T) continue;   // empty_ports_at_end_of_simulation (utils.py:254-256)
            const int tdep = (int)(stay + t + 3);
            free_from[p] = tdep + 2;             // occupancy_list[t+1 : t_dep] = 1 and the 3-step look-back
            if (n < cap) out[n] = Ev2gGenSession{p, t + 1, tdep, model, B, pac, cap0};
            n++;
        }
    }
    return n;   // > cap: the caller's buffer was too small (P * (T / 5 + 1) always suffices: a session keeps its port for at least 5 steps)
}

// ---- one scenario: one transformer -----------------------------------------------------------------------------------------------
// out arrays are this transformer's [T] rows; dr its [n_dr][3] slots; sun_scale is the env-wide cloudiness factor
EV2G_HD void ev2g_gen_transformer(const Ev2gGenRun &g, const Ev2gRng &r, int tr, double cap, double sun_scale, double *maxp, double *minp, double *infl,
                                  double *solar, double *lf, double *pvf, double *dr, int32_t *n_dr_out) {
    const ev2g_gen_config &c = *g.c;
    const int T = g.T;
    const uint64_t k = (uint64_t)tr;
    for (int t = 0; t < T; t++) { maxp[t] = cap; minp[t] = -cap; }
    if (c.inflexible_loads) {
        const double lvl = r.uni(EV2G_RS_TR, k, 0, 0.6, 1.4);
        double mx = 0.0;
        for (int t = 0; t < T; t++) {
            const double tod = fmod(g.hour + c.minute / 60.0 + t * (double)g.dt / 60.0, 24.0) / 24.0;
            const double s1 = sin((tod - 0.3) * 6.283185307179586), e1 = (tod - 0.8) / 0.08;
            const double shape = 0.35 + 0.25 * s1 * s1 + 0.5 * exp(-(e1 * e1));
            infl[t] = fabs(shape * lvl + r.normal(EV2G_RS_TR, k, 16 + (uint64_t)t, 0.0, 0.03));
            if (infl[t] > mx) mx = infl[t];
        }
        const double mult = r.normal(EV2G_RS_TR, k, 1, c.inflexible_loads_capacity_multiplier_mean, 0.1);
        for (int t = 0; t < T; t++) {
            double v = infl[t] * mult * (cap / mx + 0.0000001);
            infl[t] = v < minp[t] ? minp[t] : (v > maxp[t] ? maxp[t] : v);
        }
    } else {
        for (int t = 0; t < T; t++) infl[t] = 0.0;
    }
    if (c.solar_power) {
        const double a = r.uni(EV2G_RS_TR, k, 2, 0.9, 1.1), m = r.normal(EV2G_RS_TR, k, 3, c.solar_power_capacity_multiplier_mean, 0.1);
        for (int t = 0; t < T; t++) {
            if (g.pv_series) {   // the reference's PV year (loaders.py:165-224): sun_scale carries the scenario's day of the year
                const long long i0 = (long long)sun_scale * g.pv_per_day + (g.hour * 60 + c.minute) / g.dt;
                solar[t] = -(g.pv_series[i0 + t] * a) * m * cap;
                continue;
            }
            const double tod = fmod(g.hour + c.minute / 60.0 + t * (double)g.dt / 60.0, 24.0);
            double s = sin((tod - 6.5) / 13.0 * 3.141592653589793);
            s = s > 0 ? s * sqrt(s) : 0.0;   // clip(., 0) ** 1.5
            solar[t] = -(s * sun_scale * a) * m * cap;
        }
    } else {
        for (int t = 0; t < T; t++) solar[t] = 0.0;
    }
    for (int i = 0; i < g.n_dr * 3; i++) dr[i] = 0.0;
    *n_dr_out = 0;
    if (c.demand_response) {   // generate_demand_response_events transformer.py:80-140, one event after the other
        for (int e = 0; e < c.dr_events_per_day; e++) {
            const long long length = r.integers(EV2G_RS_DR, k, 4 * (uint64_t)e, c.dr_event_length_minutes_min, (long long)c.dr_event_length_minutes_max + 1);
            double start_min = r.normal(EV2G_RS_DR, k, 4 * (uint64_t)e + 1, c.dr_event_start_hour_mean * 60, c.dr_event_start_hour_std * 60);
            start_min = start_min < 0 ? 0 : (start_min > 23 * 60 ? 23 * 60 : start_min);
            const int es = (int)(floor(start_min / g.dt) - (double)((g.hour * 60 + c.minute) / g.dt));
            const int ee = es + (int)(length / g.dt);
            double capp = r.normal(EV2G_RS_DR, k, 4 * (uint64_t)e + 2, c.dr_event_capacity_percentage_mean, c.dr_event_capacity_percentage_std);
            capp = capp < 0 ? 0 : (capp > 100 ? 100 : capp);
            bool over = false;
            double load_max = -INFINITY;
            // max_power[es:ee] is a Python slice (transformer.py:118-131): an event that starts before the simulation does has
            // negative bounds, which count from the END of the array (es = -2, ee = 2 selects nothing; es = -8, ee = -4 hits the
            // last steps of the episode); the recorded event keeps the raw bounds
            int s0, s1;
            ev2g_py_slice(es, ee, T, &s0, &s1);
            for (int t = s0; t < s1; t++) {
                maxp[t] = maxp[t] - maxp[t] * capp / 100;
                if (infl[t] > maxp[t]) over = true;
                if (infl[t] > load_max) load_max = infl[t];
            }
            if (over) {   // the load exceeds the reduced limit inside the event: the limit is lifted to the load's maximum
                for (int t = s0; t < s1; t++) maxp[t] = load_max;
                double mxp = -INFINITY;
                for (int t = 0; t < T; t++) if (maxp[t] > mxp) mxp = maxp[t];
                capp = 100 * (1 - load_max / mxp);
            }
            dr[e * 3 + 0] = es; dr[e * 3 + 1] = ee; dr[e * 3 + 2] = capp;
        }
        *n_dr_out = c.dr_events_per_day;
    }
    const double fm = c.inflexible_loads_forecast_mean / 100, fs = c.inflexible_loads_forecast_std / 100;
    const double pm = c.solar_power_forecast_mean / 100, ps = c.solar_power_forecast_std / 100;
    for (int t = 0; t < T; t++) {
        if (c.inflexible_loads) {
            const double v = r.normal(EV2G_RS_FC, k, 2 * (uint64_t)t, fm * infl[t], fabs(fs * infl[t]));
            lf[t] = v < minp[t] ? minp[t] : (v > maxp[t] ? maxp[t] : v);
        } else lf[t] = 0.0;
        pvf[t] = c.solar_power ? r.normal(EV2G_RS_FC, k, 2 * (uint64_t)t + 1, pm * solar[t], fabs(ps * solar[t])) : 0.0;
    }
    lf[0] = infl[0];   // reset() already observed step 0 (transformer.py:178-180)
    pvf[0] = solar[0];
}

// ---- one scenario: power setpoints (generate_power_setpoints utils.py:664-757, simplified) ----------------------------------------
// price-weighted spread of every session's energy over its stay, median-smoothed; `load` and `tmp` are [T + 16] scratch
EV2G_HD void ev2g_gen_setpoints(const Ev2gGenRun &g, const Ev2gRng &r, const double *charge_price, const Ev2gGenSession *ss, int n, const double *min_cs /*[P]*/,
                                const double *max_cs /*[P]*/, double pac_min, double *sp /*[T]*/, double *w /*[T] scratch*/, double *pad /*[T+16] scratch*/) {
    const ev2g_gen_config &c = *g.c;
    const int T = g.T;
    for (int t = 0; t < T; t++) sp[t] = 0.0;
    if (!c.power_setpoint_enabled || n == 0) return;
    double pmax = 0.0;
    for (int t = 0; t < T; t++) pmax = fmax(pmax, fabs(charge_price[t]));
    double prmin = INFINITY;
    for (int t = 0; t < T; t++) prmin = fmin(prmin, fabs(charge_price[t]) / pmax);
    const double sd = fmax(prmin, 1e-3);
    for (int s = 0; s < n; s++) {
        const Ev2gGenSession &e = ss[s];
        double wsum = 0.0;
        for (int t = 0; t < T; t++) {
            const bool win = t >= e.t_arr + 1 && t < e.t_dep;   // steps t+2 .. t_dep-1
            w[t] = win ? fabs(r.normal(EV2G_RS_SETPOINT, (uint64_t)s, (uint64_t)t, 1 - fabs(charge_price[t]) / pmax, sd)) : 0.0;
            wsum += w[t];
        }
        wsum = fmax(wsum, 1e-12);
        const double need = (e.B - e.cap0) * (100 + c.power_setpoint_flexiblity) / 100;
        const double lo = fmax(pac_min, min_cs[e.port]), hi = fmin(e.pac, max_cs[e.port]);
        for (int t = 0; t < T; t++) {
            double l = w[t] / wsum * need * 60 / g.dt;
            l = (l > 0 && l < lo) ? 0.0 : fmin(l, hi);
            sp[t] += l;
        }
    }
    const int k = 5 * ((15 / g.dt) > 1 ? (15 / g.dt) : 1);   // median window (edge-padded)
    const int left = k / 2;
    for (int i = 0; i < T + k - 1; i++) { const int t = i - left; pad[i] = sp[t < 0 ? 0 : (t >= T ? T - 1 : t)]; }
    for (int t = 0; t < T; t++) {
        double win[80];
        for (int i = 0; i < k; i++) win[i] = pad[t + i];
        for (int i = 1; i < k; i++) { const double v = win[i]; int j = i - 1; while (j >= 0 && win[j] > v) { win[j + 1] = win[j]; j--; } win[j + 1] = v; }
        sp[t] = (k & 1) ? win[k / 2] : 0.5 * (win[k / 2 - 1] + win[k / 2]);
    }
}
