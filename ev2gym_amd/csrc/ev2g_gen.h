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

// ---- elementary functions with ONE result on every processor -----------------------------------------------------------------
// The device generates scenarios too (ev2g_pool_refill), and a scenario must not depend on where it was drawn: the same (seed,
// index) gives the same tensors bit for bit on the host and on the GPU.  IEEE +, -, *, / and sqrt are correctly rounded everywhere
// (the library is built with -ffp-contract=off); log / exp / sin / cos are not -- glibc and the device maths library round differently in
// the last bit -- so the generator uses these: fixed sequences of IEEE operations (argument reduction + Taylor / atanh series, accurate to a
// few 1e-16), identical wherever they run.  They are for the generator's argument ranges, not general-purpose replacements.
EV2G_HD double ev2g_two_pow(int e) {   // 2^e for -1022 <= e <= 1023, exact
    union { uint64_t u; double d; } v;
    v.u = (uint64_t)(e + 1023) << 52;
    return v.d;
}
EV2G_HD double ev2g_dlog(double x) {   // x > 0, normal
    union { uint64_t u; double d; } v;
    v.d = x;
    int e = (int)((v.u >> 52) & 0x7ff) - 1023;
    v.u = (v.u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;   // mantissa in [1, 2)
    double m = v.d;
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }           // [sqrt(1/2), sqrt(2))
    const double s = (m - 1.0) / (m + 1.0), s2 = s * s;           // log m = 2 atanh s, |s| <= 0.1716
    double p = 1.0 / 23.0;
    p = p * s2 + 1.0 / 21.0; p = p * s2 + 1.0 / 19.0; p = p * s2 + 1.0 / 17.0; p = p * s2 + 1.0 / 15.0; p = p * s2 + 1.0 / 13.0;
    p = p * s2 + 1.0 / 11.0; p = p * s2 + 1.0 / 9.0; p = p * s2 + 1.0 / 7.0; p = p * s2 + 1.0 / 5.0; p = p * s2 + 1.0 / 3.0;
    p = p * s2 + 1.0;
    return (double)e * 0.6931471805599453 + 2.0 * s * p;
}
EV2G_HD double ev2g_dexp(double x) {   // |x| < 700
    const double n = rint(x * 1.4426950408889634);
    const double r = (x - n * 0.693147180369123816490) - n * 1.90821492927058770002e-10;   // ln 2 = hi + lo
    double p = 1.0 / 6227020800.0;   // 1/13!
    p = p * r + 1.0 / 479001600.0; p = p * r + 1.0 / 39916800.0; p = p * r + 1.0 / 3628800.0; p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0; p = p * r + 1.0 / 5040.0; p = p * r + 1.0 / 720.0; p = p * r + 1.0 / 120.0; p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0; p = p * r + 0.5; p = p * r + 1.0; p = p * r + 1.0;
    return p * ev2g_two_pow((int)n);
}
// sin and cos of moderate arguments (|x| < 1e4): x = k pi/2 + r, |r| <= pi/4, Taylor series of sin r and cos r
EV2G_HD void ev2g_dsincos(double x, double *sn, double *cs) {
    const double k = rint(x * 0.6366197723675814);
    const double r = (x - k * 1.57079632673412561417) - k * 6.07710050650619224932e-11;   // pi/2 = hi + lo
    const double r2 = r * r;
    double ps = -1.0 / 1307674368000.0;   // -1/15!
    ps = ps * r2 + 1.0 / 6227020800.0; ps = ps * r2 - 1.0 / 39916800.0; ps = ps * r2 + 1.0 / 362880.0; ps = ps * r2 - 1.0 / 5040.0;
    ps = ps * r2 + 1.0 / 120.0; ps = ps * r2 - 1.0 / 6.0; ps = ps * r2 + 1.0;
    const double s = r * ps;
    double pc = 1.0 / 20922789888000.0;   // 1/16!
    pc = pc * r2 - 1.0 / 87178291200.0; pc = pc * r2 + 1.0 / 479001600.0; pc = pc * r2 - 1.0 / 3628800.0; pc = pc * r2 + 1.0 / 40320.0;
    pc = pc * r2 - 1.0 / 720.0; pc = pc * r2 + 1.0 / 24.0; pc = pc * r2 - 0.5; pc = pc * r2 + 1.0;
    const long long q = ((long long)k % 4 + 4) % 4;
    *sn = (q == 0) ? s : (q == 1) ? pc : (q == 2) ? -s : -pc;
    *cs = (q == 0) ? pc : (q == 1) ? -s : (q == 2) ? -pc : s;
}
EV2G_HD double ev2g_dsin(double x) { double s, c; ev2g_dsincos(x, &s, &c); return s; }
EV2G_HD double ev2g_dcos(double x) { double s, c; ev2g_dsincos(x, &s, &c); return c; }

// ---- counter-based random numbers ------------------------------------------------------------------------------------------
enum { EV2G_RS_HOUR = 1, EV2G_RS_PRICE, EV2G_RS_WEEKEND, EV2G_RS_SPAWN, EV2G_RS_SESSION, EV2G_RS_TR, EV2G_RS_SOLAR_ENV, EV2G_RS_DR, EV2G_RS_FC, EV2G_RS_SETPOINT };

EV2G_HD uint64_t ev2g_mix64(uint64_t z) {   // splitmix64 finaliser
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// A draw is a 32-bit hash of (scenario key, stream, a, b): two rounds of a multiply-xorshift finaliser ("lowbias32": avalanche bias
// < 0.2 %) keyed by two words of the scenario's splitmix64 key.  32-bit multiplies because the GPU draws scenarios too: a spawn trial
// -- one draw per port and step -- is what generating a scenario mostly costs, and a 64 x 64-bit product is four quarter-rate
// instructions there where this is one.  Integer arithmetic only: the same bits on every processor.
EV2G_HD uint32_t ev2g_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
struct Ev2gRng {
    uint64_t key;   // mix of (seed, scenario)
    EV2G_HD uint32_t bits(uint64_t stream, uint64_t a, uint64_t b) const {
        const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
        uint32_t x = ev2g_hash32((k0 + (uint32_t)stream * 0x9E3779B1u) ^ ((uint32_t)a * 0x85EBCA6Bu) ^ (uint32_t)(a >> 32));
        x = ev2g_hash32((x + k1) ^ ((uint32_t)b * 0xC2B2AE35u) ^ (uint32_t)(b >> 32));
        return x;
    }
    EV2G_HD double uni(uint64_t stream, uint64_t a, uint64_t b) const { return ((double)bits(stream, a, b) + 0.5) * (1.0 / 4294967296.0); }   // (0, 1)
    EV2G_HD double uni(uint64_t stream, uint64_t a, uint64_t b, double lo, double hi) const { return lo + (hi - lo) * uni(stream, a, b); }
    EV2G_HD long long integers(uint64_t stream, uint64_t a, uint64_t b, long long lo, long long hi) const {   // lo <= x < hi
        const long long n = hi - lo;
        return lo + (long long)floor(uni(stream, a, b) * (double)n);
    }
    // Box-Muller on two draws of their own: bit 62 / 63 of `b` are never set by a uni() / integers() call site, so a normal draw
    // shares no counter with any other draw of the same (stream, a) -- draws documented as independent are independent
    EV2G_HD double normal(uint64_t stream, uint64_t a, uint64_t b, double mean, double sd) const {
        const double u1 = uni(stream, a, b | (1ull << 62)), u2 = uni(stream, a, b | (1ull << 63));
        return mean + sd * sqrt(-2.0 * ev2g_dlog(u1)) * ev2g_dcos(6.283185307179586 * u2);
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

// ===================================================================================================================================
// One scenario, written so that ONE definition serves the host (ev2g_gen_host.h: a thread runs the loops below) and the device
// (ev2g_refill.h: a wavefront runs them with a lane per step / per port): every quantity is a function of (scenario, element) through
// the counter-based generator, reductions are max / min (order-free) or the fixed 64-leaf tree ev2g_tree64, and the sessions of a
// port depend on that port's own history only.
// ===================================================================================================================================

// what a scenario draws once
struct Ev2gScenarioDraw { int hour; bool weekend; double price_scale, sun; };
EV2G_HD Ev2gScenarioDraw ev2g_gen_scenario_draw(const Ev2gGenRun &g0, const Ev2gRng &rng, const Ev2gRng &rng_tr, bool pv_data) {
    const ev2g_gen_config &c = *g0.c;
    Ev2gScenarioDraw d;
    d.hour = c.random_hour ? (int)rng.integers(EV2G_RS_HOUR, 0, 0, 5, 16) : g0.hour;   // per scenario, like the reference's per-reset draw (ev2gym_env.py:131-133)
    // weekday or weekend tables: the reference's date decides; workplaces are always simulated on weekdays (ev2gym_env.py:141-154)
    d.weekend = (c.scenario == 0 || c.simulation_days == 0) ? false : (c.simulation_days == 1 ? true : rng.uni(EV2G_RS_WEEKEND, 0, 0) < 2.0 / 7.0);
    d.price_scale = rng.uni(EV2G_RS_PRICE, 0, 0, 0.6, 1.6);
    // the env-wide sun factor: a cloudiness scale for the synthetic curve, the day of the year for the PV data
    d.sun = !c.solar_power ? 0.0 : (pv_data ? (double)rng_tr.integers(EV2G_RS_SOLAR_ENV, 0, 0, 0, 365) : rng_tr.uni(EV2G_RS_SOLAR_ENV, 0, 0, 0.3, 1.0));
    return d;
}

// ---- prices: EUR/kWh at step t (hourly day-ahead-like curve; charge price = -p, discharge price = p * factor, loaders.py:439-442) ----
EV2G_HD double ev2g_gen_price_at(const Ev2gGenRun &g, const Ev2gRng &r, double scale, int t) {
    const double sh = g.hour + g.c->minute / 60.0 + t * (double)g.dt / 60.0;
    const int h = (int)floor(sh);
    const double hh = (double)(h % 24);
    const double base = 75 + 40 * ev2g_dsin((hh - 7) / 24 * 6.283185307179586) + 30 * ev2g_dsin((hh - 17) / 12 * 6.283185307179586);
    const double hp = ev2g_round_dec(fmax(base * scale + r.normal(EV2G_RS_PRICE, 1, (uint64_t)h, 0.0, 12.0), 3.0), 100.0);
    return hp / 1000.0;
}

// ---- EV sessions of ONE port, in time order (EV_spawner utils.py:477-557, spawn_single_EV :177-345) -----------------------------------
struct Ev2gGenSession { int port, t_arr, t_dep, model; double B, pac, cap0; };

// `rate` in percent; the comparison is done on the integer draw: hit <=> bits < rate / 100 * 2^32 (a threshold per step, computed once)
EV2G_HD uint32_t ev2g_gen_spawn_threshold(double rate) {
    const double x = rate * (4294967296.0 / 100.0);
    return !(x > 0.0) ? 0u : (x >= 4294967295.0 ? 4294967295u : (uint32_t)x);
}
// the spawner's tables at spawn step t: arrivals per port in percent per step, mean stay in hours, mean required energy in kWh
struct Ev2gStepTables { double rate, stay, emean; uint32_t threshold, key; };   // threshold / key: the integer form of the spawn trial
EV2G_HD Ev2gStepTables ev2g_gen_step_tables(const Ev2gGenRun &g, const Ev2gRng &r, bool weekend, int t) {
    const ev2g_gen_config &c = *g.c;
    Ev2gStepTables s;
    if (c.tab_arrival_week) {   // the reference's own tables, looked up its way (utils.py:199-233, 505-528)
        const int mod = ev2g_minute_of_day(g, t), hh = mod / 60;
        s.rate = (weekend ? c.tab_arrival_weekend : c.tab_arrival_week)[mod / 15];
        if (c.scenario == 0 && (hh < 6 || hh > 18)) s.rate = 0.0;
        s.rate *= (g.dt / 60.0) * c.spawn_multiplier;
        s.stay = c.tab_stay[mod / 30]; s.emean = c.tab_energy[mod / 30];
    } else {
        const int kind = c.scenario + ((weekend && c.scenario != 0) ? 2 : 0);   // 0 workplace, 1 public, 2 private, 3 public weekend, 4 private weekend
        const double hod = g.hour + c.minute / 60.0 + t * (double)g.dt / 60.0;
        s.rate = ev2g_gen_interp24(EV2G_GEN_RATE[kind], hod) * (g.dt / 60.0) * c.spawn_multiplier;
        s.stay = ev2g_gen_interp24(EV2G_GEN_STAY[kind], hod); s.emean = EV2G_GEN_ENERGY[kind];
    }
    s.threshold = ev2g_gen_spawn_threshold(s.rate);
    s.key = r.bits(EV2G_RS_SPAWN, (uint64_t)t, 0);   // the step's own key: a port's trial is one more hash round on top of it
    return s;
}
EV2G_HD double ev2g_gen_share_sum(const Ev2gFleet &fleet) {
    double a = 0.0;
    for (int i = 0; i < fleet.n; i++) a += fleet.share(i);
    return a;
}
// does an EV arrive at port p at the end of spawn step t?  (one draw per port and step: what generating a scenario mostly costs)
// does an EV arrive at port p at the end of spawn step t?  One draw per port and step -- what generating a scenario mostly costs --, so
// it is one hash round on the step's key, compared as an integer: hit <=> hash32(key_t ^ p * C) < rate_t / 100 * 2^32
EV2G_HD bool ev2g_gen_spawn_trial(uint32_t step_key, uint32_t threshold, int p) {
    return ev2g_hash32(step_key ^ ((uint32_t)p * 0x85EBCA6Bu + 0x9E3779B1u)) < threshold;
}
// the session spawned at step t on port p (spawn_single_EV utils.py:177-345); false: dropped by empty_ports_at_end_of_simulation (:254-256)
EV2G_HD bool ev2g_gen_make_session(const Ev2gGenRun &g, const Ev2gRng &r, const Ev2gFleet &fleet, double share_sum, int t, int p, double stay_mean,
                                   double e_mean, Ev2gGenSession *out) {
    const ev2g_gen_config &c = *g.c;
    const uint64_t id = (uint64_t)t * (uint64_t)g.P + (uint64_t)p;
    double stay = r.normal(EV2G_RS_SESSION, id, 2, stay_mean, 0.2 * stay_mean) * 60.0 / g.dt + 1;
    if (stay < g.min_stay_steps) stay = g.min_stay_steps;
    if (stay + t + 4 >= g.T) return false;
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
    *out = Ev2gGenSession{p, t + 1, (int)(stay + t + 3), model, B, pac, cap0};
    return true;
}
// a port's sessions in time order; tab(t) -> Ev2gStepTables of spawn step t (computed once per scenario by the caller)
template <class Tab, class Emit>
EV2G_HD int ev2g_gen_port_sessions(const Ev2gGenRun &g, const Ev2gRng &r, const Ev2gFleet &fleet, double share_sum, int p, Tab &&tab, Emit &&emit) {
    int free_from = 0, n = 0;   // first spawn step at which the port passes the 3-step-empty rule (utils.py:534-552)
    for (int t = 2; t < g.T - g.min_stay_steps - 1; t++) {
        if (free_from > t) continue;
        const Ev2gStepTables st = tab(t);
        if (st.threshold == 0u || !ev2g_gen_spawn_trial(st.key, st.threshold, p)) continue;
        Ev2gGenSession e;
        if (!ev2g_gen_make_session(g, r, fleet, share_sum, t, p, st.stay, st.emean, &e)) continue;
        free_from = e.t_dep + 2;                // occupancy_list[t+1 : t_dep] = 1 and the 3-step look-back
        emit(n, e);
        n++;
    }
    return n;
}

// ---- the per-session fields spawn_single_EV sets besides the ones above (utils.py:298-345) ---------------------------------------------
struct Ev2gSessFields { double desired, minB, min_emerg, pac_min, pdis_max, pdis_min, ts, tsm, eta_ch, eta_dis; int phases, lut; };
// spec_row [n_ev_specs]: the efficiency-table row of every spec model (-1: none); may be null without a spec file
EV2G_HD Ev2gSessFields ev2g_gen_session_fields(const Ev2gGenRun &g, const Ev2gRng &rng, const Ev2gGenSession &e, const int *spec_row) {
    const ev2g_gen_config &c = *g.c;
    Ev2gSessFields f;
    const uint64_t id = (uint64_t)(e.t_arr - 1) * (uint64_t)g.P + (uint64_t)e.port;   // the spawn trial this session came from
    f.desired = c.ev_desired_capacity * e.B; f.minB = c.ev_min_battery_capacity;
    f.min_emerg = c.ev_min_emergency_battery_capacity > e.B ? 0.7 * e.B : c.ev_min_emergency_battery_capacity;
    f.tsm = c.ev_transition_soc_multiplier;
    if (c.heterogeneous_ev_specs) {
        f.pac_min = 0.0; f.pdis_max = c.v2g_enabled ? -e.pac : 0.0; f.pdis_min = 0.0; f.phases = 3;
        f.ts = ev2g_round_dec(0.9 - (rng.uni(EV2G_RS_SESSION, id, 20) + 0.00001) / 5, 1000.0);
        if (c.n_ev_specs > 0) f.pdis_max = -c.spec_max_ac_discharge_power[e.model];   // as written in the file (utils.py:303-304)
        const bool table = c.n_ev_specs > 0 ? (spec_row && spec_row[e.model] >= 0) : (c.fleet_with_efficiency_tables != 0);
        if (table) { f.lut = c.n_ev_specs > 0 ? spec_row[e.model] : e.model; f.eta_ch = NAN; f.eta_dis = NAN; }
        else {
            f.lut = -1;
            f.eta_ch = ev2g_round_dec(1 - (rng.uni(EV2G_RS_SESSION, id, 21) + 0.00001) / 20, 1000.0);
            f.eta_dis = ev2g_round_dec(1 - (rng.uni(EV2G_RS_SESSION, id, 22) + 0.00001) / 20, 1000.0);
        }
    } else {
        f.pac_min = c.ev_min_ac_charge_power; f.pdis_max = c.ev_max_discharge_power; f.pdis_min = c.ev_min_discharge_power;
        f.phases = c.ev_phases; f.ts = c.ev_transition_soc; f.lut = -1;
        f.eta_ch = c.ev_charge_efficiency; f.eta_dis = c.ev_discharge_efficiency;
    }
    return f;
}

// ---- one transformer, element by element (load_transformers loaders.py:227-296, transformer.py:80-256) -------------------------------------
EV2G_HD double ev2g_gen_infl_raw(const Ev2gGenRun &g, const Ev2gRng &r, int tr, double lvl, int t) {   // the load shape before it is scaled to the transformer
    const double tod = fmod(g.hour + g.c->minute / 60.0 + t * (double)g.dt / 60.0, 24.0) / 24.0;
    const double s1 = ev2g_dsin((tod - 0.3) * 6.283185307179586), e1 = (tod - 0.8) / 0.08;
    const double shape = 0.35 + 0.25 * s1 * s1 + 0.5 * ev2g_dexp(-(e1 * e1));
    return fabs(shape * lvl + r.normal(EV2G_RS_TR, (uint64_t)tr, 16 + (uint64_t)t, 0.0, 0.03));
}
EV2G_HD double ev2g_gen_infl_scaled(double raw, double mult, double cap, double mx) {   // normalize_inflexible_loads transformer.py:213-233
    const double v = raw * mult * (cap / mx + 0.0000001);
    return v < -cap ? -cap : (v > cap ? cap : v);
}
EV2G_HD double ev2g_gen_solar_at(const Ev2gGenRun &g, double sun, double a, double m, double cap, int t) {
    if (g.pv_series) {   // the reference's PV year (loaders.py:165-224): `sun` carries the scenario's day of the year
        const long long i0 = (long long)sun * g.pv_per_day + (g.hour * 60 + g.c->minute) / g.dt;
        return -(g.pv_series[i0 + t] * a) * m * cap;
    }
    const double tod = fmod(g.hour + g.c->minute / 60.0 + t * (double)g.dt / 60.0, 24.0);
    double s = ev2g_dsin((tod - 6.5) / 13.0 * 3.141592653589793);
    s = s > 0 ? s * sqrt(s) : 0.0;   // clip(., 0) ** 1.5
    return -(s * sun * a) * m * cap;
}
struct Ev2gDrEvent { int es, ee, s0, s1; double capp; };   // raw bounds, the Python slice [s0, s1) they select, capacity percentage
EV2G_HD Ev2gDrEvent ev2g_gen_dr_event(const Ev2gGenRun &g, const Ev2gRng &r, int tr, int e) {   // generate_demand_response_events transformer.py:80-140
    const ev2g_gen_config &c = *g.c;
    const uint64_t k = (uint64_t)tr;
    Ev2gDrEvent ev;
    const long long length = r.integers(EV2G_RS_DR, k, 4 * (uint64_t)e, c.dr_event_length_minutes_min, (long long)c.dr_event_length_minutes_max + 1);
    double start_min = r.normal(EV2G_RS_DR, k, 4 * (uint64_t)e + 1, c.dr_event_start_hour_mean * 60, c.dr_event_start_hour_std * 60);
    start_min = start_min < 0 ? 0 : (start_min > 23 * 60 ? 23 * 60 : start_min);
    ev.es = (int)(floor(start_min / g.dt) - (double)((g.hour * 60 + c.minute) / g.dt));
    ev.ee = ev.es + (int)(length / g.dt);
    double capp = r.normal(EV2G_RS_DR, k, 4 * (uint64_t)e + 2, c.dr_event_capacity_percentage_mean, c.dr_event_capacity_percentage_std);
    ev.capp = capp < 0 ? 0 : (capp > 100 ? 100 : capp);
    // max_power[es:ee] is a Python slice (transformer.py:118-131): negative bounds count from the END of the array
    ev2g_py_slice(ev.es, ev.ee, g.T, &ev.s0, &ev.s1);
    return ev;
}
EV2G_HD double ev2g_gen_load_forecast_at(const Ev2gGenRun &g, const Ev2gRng &r, int tr, int t, double infl, double minp, double maxp) {
    if (t == 0) return infl;   // reset() already observed step 0 (transformer.py:178-180)
    const double fm = g.c->inflexible_loads_forecast_mean / 100, fs = g.c->inflexible_loads_forecast_std / 100;
    const double v = r.normal(EV2G_RS_FC, (uint64_t)tr, 2 * (uint64_t)t, fm * infl, fabs(fs * infl));
    return v < minp ? minp : (v > maxp ? maxp : v);
}
EV2G_HD double ev2g_gen_pv_forecast_at(const Ev2gGenRun &g, const Ev2gRng &r, int tr, int t, double solar) {
    if (t == 0) return solar;
    const double pm = g.c->solar_power_forecast_mean / 100, ps = g.c->solar_power_forecast_std / 100;
    return r.normal(EV2G_RS_FC, (uint64_t)tr, 2 * (uint64_t)t + 1, pm * solar, fabs(ps * solar));
}

// ---- fixed 64-leaf summation tree: leaf j holds v[j] + v[j + 64] + ... (in that order), leaves are combined by xor butterflies
// (32, 16, .. 1) -- what a wavefront does with one lane per leaf; the host walks the same tree.  leaves[64] is overwritten. ----
EV2G_HD double ev2g_tree64(double *leaves) {
    for (int d = 32; d > 0; d >>= 1)
        for (int l = 0; l < 64; l++)
            if ((l & d) == 0) { const double a = leaves[l] + leaves[l ^ d]; leaves[l] = a; leaves[l ^ d] = a; }
    return leaves[0];
}

// ---- power setpoints (generate_power_setpoints utils.py:664-757, simplified): price-weighted spread of every session's energy over
// its stay, median-smoothed.  A session's weight at step t: ----
EV2G_HD double ev2g_gen_setpoint_weight(const Ev2gRng &r, uint64_t id, int t, int t_arr, int t_dep, double price_rel, double sd) {
    const bool win = t >= t_arr + 1 && t < t_dep;   // steps t+2 .. t_dep-1
    return win ? fabs(r.normal(EV2G_RS_SETPOINT, id, (uint64_t)t, 1 - price_rel, sd)) : 0.0;
}
EV2G_HD double ev2g_gen_setpoint_load(double w, double wsum, double need, int dt, double lo, double hi) {
    double l = w / wsum * need * 60 / dt;
    return (l > 0 && l < lo) ? 0.0 : fmin(l, hi);
}
EV2G_HD int ev2g_gen_median_window(int dt) { return 5 * ((15 / dt) > 1 ? (15 / dt) : 1); }
// median of the k values pad[t .. t + k) (k <= 80)
EV2G_HD double ev2g_gen_median(const double *pad, int t, int k) {
    if (k == 5) {    // (15-minute steps) a selection network on five values read once: max(min(a,b), min(c,d)), min(max(a,b), max(c,d)) and e hold the median
                     // among them -- the same element the ranks below pick (finite values, no negative zeros: sums of non-negative loads)
        const double a = pad[t], b = pad[t + 1], c = pad[t + 2], d = pad[t + 3], e = pad[t + 4];
        const double f = fmax(fmin(a, b), fmin(c, d)), g = fmin(fmax(a, b), fmax(c, d));
        return fmax(fmin(e, f), fmin(fmax(e, f), g));
    }
    if (k <= 16) {   // by rank (no array to sort: on the device an indexed local array is scratch memory): element i has rank #{x_j < x_i} + #{j < i: x_j == x_i};
                     // the value(s) of the middle rank(s) are what the sort below finds
        double lo = 0.0, hi = 0.0;
        for (int i = 0; i < k; i++) {
            const double xi = pad[t + i];
            int r = 0;
            for (int j = 0; j < k; j++) { const double xj = pad[t + j]; r += (xj < xi || (xj == xi && j < i)) ? 1 : 0; }
            if (r == (k - 1) / 2) lo = xi;
            if (r == k / 2) hi = xi;
        }
        return (k & 1) ? hi : 0.5 * (lo + hi);
    }
    double win[80];
    for (int i = 0; i < k; i++) win[i] = pad[t + i];
    for (int i = 1; i < k; i++) { const double v = win[i]; int j = i - 1; while (j >= 0 && win[j] > v) { win[j + 1] = win[j]; j--; } win[j + 1] = v; }
    return (k & 1) ? win[k / 2] : 0.5 * (win[k / 2 - 1] + win[k / 2]);
}
