// ev2g_gen_host.h -- host driver of the scenario generator (ev2g_generate & co. of include/ev2g.h): slices the scenarios over
// threads, runs ev2g_gen.h's per-scenario code, and assembles one ev2g_scenario_batch (sessions in CSR order).  Included by ev2g_host.hip.
#pragma once
#include <atomic>
#include <exception>
#include <memory>
#include <thread>

#include "ev2g_gen.h"

struct ev2g_gen_result {
    ev2g_scenario_batch b{};
    std::vector<double> cs_min_c, cs_max_c, cs_min_d, cs_max_d, cs_volt, charge_price, discharge_price, setpoints;
    std::vector<int32_t> cs_phases, cs_tr, cs_np;
    std::vector<double> maxp, minp, infl, solar, lf, pvf, dr;
    std::vector<int32_t> n_dr, steps_ahead;
    std::vector<int64_t> sess_start;
    std::vector<int32_t> ev_cs, ev_ta, ev_td, ev_ph, ev_lut;
    std::vector<double> cap0, B, desired, minB, min_emerg, pac_max, pac_min, pdis_max, pdis_min, ts, tsm, eta_ch, eta_dis, lut;
};

// the efficiency-table row of every spec model (-1: the model has none); returns the number of tables
static int ev2g_gen_spec_rows(const ev2g_gen_config &c, std::vector<int> &spec_row) {
    spec_row.assign(std::max(c.n_ev_specs, 0), -1);
    int n = 0;
    if (c.n_ev_specs > 0 && c.spec_efficiency)
        for (int i = 0; i < c.n_ev_specs; i++)
            if (!std::isnan(c.spec_efficiency[(size_t)i * EV2G_LUT_LEN])) spec_row[i] = n++;
    return n;
}

// tab_pv brought to the simulation timescale (repeat / max-pool), rolling mean, exponentially weighted mean (loaders.py:178-193), two
// years long; empty when the config carries no PV data.  Returns an error text or null.
static const char *ev2g_gen_pv_series(const ev2g_gen_config &c, int dt, std::vector<double> &pv_series) {
    pv_series.clear();
    if (!(c.solar_power && c.tab_pv && c.n_pv >= 8760)) return nullptr;
    if (1440 % dt != 0 || (dt < 60 && 60 % dt != 0) || (dt > 60 && dt % 60 != 0)) return "ev2g_generate: tab_pv needs a timescale that divides the hour or a multiple of it";
    std::vector<double> x;
    if (dt > 60) { const int k = dt / 60; for (long long i = 0; i + k <= c.n_pv; i += k) { double m = c.tab_pv[i]; for (int j = 1; j < k; j++) m = std::max(m, c.tab_pv[i + j]); x.push_back(m); } }
    else { const int k = 60 / dt; x.reserve((size_t)c.n_pv * k); for (long long i = 0; i < c.n_pv; i++) for (int j = 0; j < k; j++) x.push_back(c.tab_pv[i]); }
    const int w = std::max(60 / dt, 1);
    std::vector<double> y(x.size());
    double run = 0.0;
    for (size_t i = 0; i < x.size(); i++) { run += x[i]; if (i >= (size_t)w) run -= x[i - w]; y[i] = run / (double)std::min<size_t>(i + 1, w); }   // rolling mean
    const double alpha = 2.0 / (w + 1.0);
    double num = 0.0, den = 0.0;
    for (size_t i = 0; i < y.size(); i++) { num = y[i] + (1 - alpha) * num; den = 1 + (1 - alpha) * den; y[i] = num / den; }   // ewm(span=w, adjust=True)
    pv_series = y;
    pv_series.insert(pv_series.end(), y.begin(), y.end());
    return nullptr;
}

// the per-run constants of a config (what ev2g_generate and ev2g_pool_refill derive alike); P = ports per scenario
static void ev2g_gen_make_run(const ev2g_gen_config &c, int P, int npc_max, uint64_t seed, Ev2gGenRun &g) {
    g = Ev2gGenRun{};
    g.c = &c; g.T = c.simulation_length; g.dt = c.timescale; g.C = c.number_of_charging_stations; g.P = P; g.R = c.number_of_transformers;
    g.npc_max = npc_max; g.seed = seed;
    g.hour = c.hour;   // random_hour: drawn per scenario, like the reference draws it per reset (ev2gym_env.py:131-133)
    g.min_stay_steps = c.ev_min_time_of_stay / g.dt;
    g.steps_ahead = c.dr_notification_of_event_minutes / g.dt;
    g.n_dr = c.demand_response ? std::max(c.dr_events_per_day, 1) : 1;
    g.lut_fleet = c.heterogeneous_ev_specs && c.fleet_with_efficiency_tables;
    g.n_fleet = EV2G_GEN_FLEET_MAX;
}

static int gen_fail(const char *msg) { g_create_error = msg; return EV2G_ERR_ARG; }

static int ev2g_generate_body(const ev2g_gen_config *cfg, int32_t M, uint64_t seed, int32_t n_threads, ev2g_gen_result **out) {
    if (out) *out = nullptr;
    if (!cfg || !out || M < 1) return gen_fail("ev2g_generate: bad arguments");
    const ev2g_gen_config &c = *cfg;
    if (c.scenario < 0 || c.scenario > 2) return gen_fail("ev2g_generate: scenario must be 0 workplace, 1 public or 2 private");
    if (c.simulation_days < 0 || c.simulation_days > 2) return gen_fail("ev2g_generate: simulation_days must be 0 weekdays, 1 weekends or 2 both");
    if (c.simulation_length < 8 || c.timescale < 1 || c.number_of_charging_stations < 1 || c.number_of_transformers < 1)
        return gen_fail("ev2g_generate: simulation_length / timescale / number_of_* out of range");
    const bool topo = c.topo_n_ports != nullptr;
    if (topo && !(c.topo_transformer && c.topo_phases && c.topo_min_charge_current && c.topo_max_charge_current && c.topo_min_discharge_current &&
                  c.topo_max_discharge_current && c.topo_voltage && c.topo_tr_max_power))
        return gen_fail("ev2g_generate: a topology needs all nine topo_* arrays");
    if (!topo && c.number_of_ports_per_cs < 1) return gen_fail("ev2g_generate: number_of_ports_per_cs < 1");
    const int T = c.simulation_length, dt = c.timescale, C = c.number_of_charging_stations, R = c.number_of_transformers;
    std::unique_ptr<ev2g_gen_result> holder(new ev2g_gen_result());   // freed on every early return and on exceptions
    ev2g_gen_result *res = holder.get();
    ev2g_gen_result &r = *res;
    // ---- chargers (load_ev_charger_profiles loaders.py:342-365; load_grid :494-498; topology :259-276,312-340) ----
    r.cs_min_c.resize(C); r.cs_max_c.resize(C); r.cs_min_d.resize(C); r.cs_max_d.resize(C); r.cs_volt.resize(C);
    r.cs_phases.resize(C); r.cs_tr.resize(C); r.cs_np.resize(C);
    std::vector<double> tr_cap(R);
    int npc_max = 0, P = 0;
    for (int i = 0; i < C; i++) {
        if (topo) {
            r.cs_min_c[i] = c.topo_min_charge_current[i]; r.cs_max_c[i] = c.topo_max_charge_current[i];
            r.cs_min_d[i] = c.topo_min_discharge_current[i]; r.cs_max_d[i] = c.topo_max_discharge_current[i];   // as written: v2g_enabled not consulted
            r.cs_volt[i] = c.topo_voltage[i]; r.cs_phases[i] = c.topo_phases[i]; r.cs_tr[i] = c.topo_transformer[i]; r.cs_np[i] = c.topo_n_ports[i];
            if (r.cs_tr[i] < 0 || r.cs_tr[i] >= R || r.cs_np[i] < 1) return gen_fail("ev2g_generate: topology entry out of range");
        } else {
            r.cs_min_c[i] = c.cs_min_charge_current; r.cs_max_c[i] = c.cs_max_charge_current;
            r.cs_min_d[i] = c.v2g_enabled ? c.cs_min_discharge_current : 0.0; r.cs_max_d[i] = c.v2g_enabled ? c.cs_max_discharge_current : 0.0;
            r.cs_volt[i] = c.cs_voltage; r.cs_phases[i] = c.cs_phases; r.cs_tr[i] = i % R; r.cs_np[i] = c.number_of_ports_per_cs;
        }
        npc_max = std::max(npc_max, (int)r.cs_np[i]);
        P += r.cs_np[i];
    }
    for (int k = 0; k < R; k++) tr_cap[k] = topo ? c.topo_tr_max_power[k] : c.transformer_max_power;
    std::vector<int> port_cs(P);
    std::vector<double> min_cs(P), max_cs(P);   // charger power limits seen by each port (generate_power_setpoints)
    for (int i = 0, p = 0; i < C; i++)
        for (int j = 0; j < r.cs_np[i]; j++, p++) {
            port_cs[p] = i;
            const double sq = std::sqrt((double)r.cs_phases[i]);
            min_cs[p] = r.cs_min_c[i] * r.cs_volt[i] * sq / 1000; max_cs[p] = r.cs_max_c[i] * r.cs_volt[i] * sq / 1000;
        }

    Ev2gGenRun g{};
    ev2g_gen_make_run(c, P, npc_max, seed, g);
    if (c.n_ev_specs < 0 || (c.n_ev_specs > 0 && !(c.spec_registrations && c.spec_battery_capacity && c.spec_max_ac_charge_power && c.spec_max_ac_discharge_power)))
        return gen_fail("ev2g_generate: n_ev_specs > 0 needs the four spec_* model arrays");
    if ((c.tab_arrival_week || c.tab_arrival_weekend || c.tab_stay || c.tab_energy) && !(c.tab_arrival_week && c.tab_arrival_weekend && c.tab_stay && c.tab_energy))
        return gen_fail("ev2g_generate: the arrival / stay / energy tables of a data directory come together");
    // which spec models carry an efficiency table, and the row of r.lut each one gets
    std::vector<int> spec_row;
    const int n_spec_lut = ev2g_gen_spec_rows(c, spec_row);
    // pv_netherlands.csv's hourly year at the simulation timescale, smoothed like loaders.py:178-193, two years long
    std::vector<double> pv_series;
    if (const char *err = ev2g_gen_pv_series(c, dt, pv_series)) return gen_fail(err);
    if (!pv_series.empty()) { g.pv_series = pv_series.data(); g.pv_per_day = 1440 / dt; }
    const int ND = g.n_dr;

    r.charge_price.resize((size_t)M * T); r.discharge_price.resize((size_t)M * T); r.setpoints.resize((size_t)M * T);
    const size_t ert = (size_t)M * R * T;
    r.maxp.resize(ert); r.minp.resize(ert); r.infl.resize(ert); r.solar.resize(ert); r.lf.resize(ert); r.pvf.resize(ert);
    r.dr.resize((size_t)M * R * ND * 3); r.n_dr.resize((size_t)M * R); r.steps_ahead.assign((size_t)M * R, g.steps_ahead);
    r.sess_start.assign((size_t)M + 1, 0);

    // default: the hardware threads, but no more than 64 -- measured on the 256-thread box of the MI355X node: 8192 cfg2 scenarios in 0.24 s with
    // 1 thread, 0.019 s with 16, 0.018 s with 64, 0.029 s with 256 (thread start-up; the serial merge is ~0.015 s)
    int nt = n_threads > 0 ? n_threads : std::min(64, (int)std::thread::hardware_concurrency());
    nt = std::max(1, std::min(nt, (int)M));
    std::vector<std::vector<Ev2gGenSession>> part(nt);   // the sessions of each thread's slice, scenario after scenario
    std::vector<int> count(M, 0);
    const Ev2gGenRun &g0 = g;
    // A worker thread that throws (std::bad_alloc for a batch too large for the host) must not reach std::terminate: the first
    // exception is parked here and re-thrown on the calling thread after the joins; threads are joined on every path
    std::exception_ptr failure;
    std::atomic<bool> failed{false};
    auto guarded = [&](auto &fn, int ti) {
        try { fn(ti); }
        catch (...) { if (!failed.exchange(true)) failure = std::current_exception(); }
    };
    auto run_slices = [&](auto &fn) {
        struct Joiner { std::vector<std::thread> th; ~Joiner() { for (auto &t : th) if (t.joinable()) t.join(); } } j;
        int ti = 1;
        try {
            j.th.reserve(nt);
            for (; ti < nt; ti++) j.th.emplace_back([&, ti] { guarded(fn, ti); });
        } catch (...) {   // thread creation failed (std::system_error): the calling thread takes the slices that have no thread
        }
        const int first_unthreaded = ti;
        guarded(fn, 0);
        for (int k = first_unthreaded; k < nt; k++) guarded(fn, k);
        for (auto &t : j.th) t.join();
        if (failed) std::rethrow_exception(failure);
    };
    auto work = [&, g0](int ti) {
        const int m0 = (int)((long long)M * ti / nt), m1 = (int)((long long)M * (ti + 1) / nt);
        std::vector<std::vector<Ev2gGenSession>> by_port(P);   // a port's sessions, in time order
        std::vector<Ev2gGenSession> buf;                        // the scenario's sessions in EVs_profiles order (arrival step, then port)
        std::vector<double> raw(T), w(T), pad(T + 96), leaves(64);
        std::vector<Ev2gStepTables> steptab(T);
        for (int m = m0; m < m1; m++) {
            const Ev2gRng rng = ev2g_rng(seed, (uint64_t)m);
            const Ev2gRng rng_tr = (c.tr_seed != -1) ? ev2g_rng((uint64_t)c.tr_seed, (uint64_t)m) : rng;
            const Ev2gScenarioDraw dr = ev2g_gen_scenario_draw(g0, rng, rng_tr, g0.pv_series != nullptr);
            Ev2gGenRun gm = g0;
            gm.hour = dr.hour;
            const Ev2gGenRun &g = gm;
            double *cp = &r.charge_price[(size_t)m * T], *dp = &r.discharge_price[(size_t)m * T];
            for (int t = 0; t < T; t++) { const double pr = ev2g_gen_price_at(g, rng, dr.price_scale, t); cp[t] = -pr; dp[t] = pr * c.discharge_price_factor; }
            // sessions: port by port (a port's sessions depend on its own history only), then merged into profile order
            buf.clear();
            const Ev2gFleet fleet = ev2g_fleet(g);
            const double share_sum = ev2g_gen_share_sum(fleet);
            for (int t = 0; t < T; t++) steptab[t] = ev2g_gen_step_tables(g, rng, dr.weekend, t);
            for (int p = 0; p < P; p++) {
                by_port[p].clear();
                ev2g_gen_port_sessions(g, rng, fleet, share_sum, p, [&](int t) { return steptab[t]; }, [&](int, const Ev2gGenSession &e) { by_port[p].push_back(e); });
                buf.insert(buf.end(), by_port[p].begin(), by_port[p].end());
            }
            std::stable_sort(buf.begin(), buf.end(), [](const Ev2gGenSession &a, const Ev2gGenSession &b) { return a.t_arr != b.t_arr ? a.t_arr < b.t_arr : a.port < b.port; });
            count[m] = (int)buf.size();
            part[ti].insert(part[ti].end(), buf.begin(), buf.end());
            // transformers
            for (int k = 0; k < R; k++) {
                const size_t o = ((size_t)m * R + k) * T;
                double *maxp = &r.maxp[o], *minp = &r.minp[o], *infl = &r.infl[o], *solar = &r.solar[o], *lf = &r.lf[o], *pvf = &r.pvf[o];
                double *drs = &r.dr[((size_t)m * R + k) * ND * 3];
                const double cap = tr_cap[k];
                for (int t = 0; t < T; t++) { maxp[t] = cap; minp[t] = -cap; }
                if (c.inflexible_loads) {
                    const double lvl = rng_tr.uni(EV2G_RS_TR, (uint64_t)k, 0, 0.6, 1.4);
                    double mx = 0.0;
                    for (int t = 0; t < T; t++) { raw[t] = ev2g_gen_infl_raw(g, rng_tr, k, lvl, t); mx = std::max(mx, raw[t]); }
                    const double mult = rng_tr.normal(EV2G_RS_TR, (uint64_t)k, 1, c.inflexible_loads_capacity_multiplier_mean, 0.1);
                    for (int t = 0; t < T; t++) infl[t] = ev2g_gen_infl_scaled(raw[t], mult, cap, mx);
                } else std::fill(infl, infl + T, 0.0);
                if (c.solar_power) {
                    const double a = rng_tr.uni(EV2G_RS_TR, (uint64_t)k, 2, 0.9, 1.1), mm = rng_tr.normal(EV2G_RS_TR, (uint64_t)k, 3, c.solar_power_capacity_multiplier_mean, 0.1);
                    for (int t = 0; t < T; t++) solar[t] = ev2g_gen_solar_at(g, dr.sun, a, mm, cap, t);
                } else std::fill(solar, solar + T, 0.0);
                for (int i = 0; i < ND * 3; i++) drs[i] = 0.0;
                r.n_dr[(size_t)m * R + k] = 0;
                if (c.demand_response) {   // one event after the other (transformer.py:96-138)
                    for (int e = 0; e < c.dr_events_per_day; e++) {
                        Ev2gDrEvent ev = ev2g_gen_dr_event(g, rng_tr, k, e);
                        bool over = false;
                        double load_max = -INFINITY;
                        for (int t = ev.s0; t < ev.s1; t++) {
                            maxp[t] = maxp[t] - maxp[t] * ev.capp / 100;
                            if (infl[t] > maxp[t]) over = true;
                            load_max = std::max(load_max, infl[t]);
                        }
                        if (over) {   // the load exceeds the reduced limit inside the event: the limit is lifted to the load's maximum
                            for (int t = ev.s0; t < ev.s1; t++) maxp[t] = load_max;
                            double mxp = -INFINITY;
                            for (int t = 0; t < T; t++) mxp = std::max(mxp, maxp[t]);
                            ev.capp = 100 * (1 - load_max / mxp);
                        }
                        drs[e * 3 + 0] = ev.es; drs[e * 3 + 1] = ev.ee; drs[e * 3 + 2] = ev.capp;
                    }
                    r.n_dr[(size_t)m * R + k] = c.dr_events_per_day;
                }
                for (int t = 0; t < T; t++) {
                    lf[t] = c.inflexible_loads ? ev2g_gen_load_forecast_at(g, rng_tr, k, t, infl[t], minp[t], maxp[t]) : 0.0;
                    pvf[t] = c.solar_power ? ev2g_gen_pv_forecast_at(g, rng_tr, k, t, solar[t]) : 0.0;
                }
            }
            // power setpoints: sessions port by port, the weight sum of a session on the fixed 64-leaf tree
            double *sp = &r.setpoints[(size_t)m * T];
            std::fill(sp, sp + T, 0.0);
            if (c.power_setpoint_enabled && !buf.empty()) {
                double pmax = 0.0, prmin = INFINITY;
                for (int t = 0; t < T; t++) pmax = std::max(pmax, std::fabs(cp[t]));
                for (int t = 0; t < T; t++) prmin = std::min(prmin, std::fabs(cp[t]) / pmax);
                const double sd = std::max(prmin, 1e-3);
                const double pac_min = c.heterogeneous_ev_specs ? 0.0 : c.ev_min_ac_charge_power;
                for (int p = 0; p < P; p++)
                    for (const Ev2gGenSession &e : by_port[p]) {
                        const uint64_t id = (uint64_t)(e.t_arr - 1) * (uint64_t)P + (uint64_t)e.port;
                        std::fill(leaves.begin(), leaves.end(), 0.0);
                        for (int t = 0; t < T; t++) { w[t] = ev2g_gen_setpoint_weight(rng, id, t, e.t_arr, e.t_dep, std::fabs(cp[t]) / pmax, sd); leaves[t & 63] += w[t]; }
                        const double wsum = std::max(ev2g_tree64(leaves.data()), 1e-12);
                        const double need = (e.B - e.cap0) * (100 + c.power_setpoint_flexiblity) / 100;
                        const double lo = std::max(pac_min, min_cs[e.port]), hi = std::min(e.pac, max_cs[e.port]);
                        for (int t = 0; t < T; t++) sp[t] += ev2g_gen_setpoint_load(w[t], wsum, need, dt, lo, hi);
                    }
                const int kw = ev2g_gen_median_window(dt), left = kw / 2;
                for (int i = 0; i < T + kw - 1; i++) { const int t = i - left; pad[i] = sp[t < 0 ? 0 : (t >= T ? T - 1 : t)]; }
                for (int t = 0; t < T; t++) sp[t] = ev2g_gen_median(pad.data(), t, kw);
            }
        }
    };
    run_slices(work);
    for (int m = 0; m < M; m++) r.sess_start[m + 1] = r.sess_start[m] + count[m];
    const size_t S = (size_t)r.sess_start[M];
    r.ev_cs.resize(S); r.ev_ta.resize(S); r.ev_td.resize(S); r.ev_ph.resize(S); r.ev_lut.resize(S);
    for (auto *v : {&r.cap0, &r.B, &r.desired, &r.minB, &r.min_emerg, &r.pac_max, &r.pac_min, &r.pdis_max, &r.pdis_min, &r.ts, &r.tsm, &r.eta_ch, &r.eta_dis}) v->resize(S);
    // per-session fields (spawn_single_EV utils.py:298-345): slice ti's sessions start at sess_start[first scenario of the slice]
    auto fill = [&](int ti) {
        const int m0 = (int)((long long)M * ti / nt), m1 = (int)((long long)M * (ti + 1) / nt);
        size_t k = 0;
        for (int m = m0; m < m1; m++) {
            const Ev2gRng rng = ev2g_rng(seed, (uint64_t)m);
            for (int i = 0; i < count[m]; i++, k++) {
                const Ev2gGenSession &e = part[ti][k];
                const size_t s = (size_t)r.sess_start[m] + i;
                const Ev2gSessFields f = ev2g_gen_session_fields(g, rng, e, spec_row.empty() ? nullptr : spec_row.data());
                r.ev_cs[s] = port_cs[e.port]; r.ev_ta[s] = e.t_arr; r.ev_td[s] = e.t_dep;
                r.cap0[s] = e.cap0; r.B[s] = e.B; r.desired[s] = f.desired; r.minB[s] = f.minB; r.min_emerg[s] = f.min_emerg;
                r.pac_max[s] = e.pac; r.tsm[s] = f.tsm; r.pac_min[s] = f.pac_min; r.pdis_max[s] = f.pdis_max; r.pdis_min[s] = f.pdis_min;
                r.ev_ph[s] = f.phases; r.ts[s] = f.ts; r.ev_lut[s] = f.lut; r.eta_ch[s] = f.eta_ch; r.eta_dis[s] = f.eta_dis;
            }
        }
    };
    run_slices(fill);
    int NL = 0;
    if (c.n_ev_specs > 0) {
        NL = n_spec_lut;
        r.lut.resize((size_t)NL * EV2G_LUT_LEN);
        for (int i = 0; i < c.n_ev_specs; i++)
            if (spec_row[i] >= 0) std::copy(c.spec_efficiency + (size_t)i * EV2G_LUT_LEN, c.spec_efficiency + (size_t)(i + 1) * EV2G_LUT_LEN, r.lut.begin() + (size_t)spec_row[i] * EV2G_LUT_LEN);
    } else if (g.lut_fleet) {   // efficiency-vs-current tables: nearest given level over 0..100 A (utils.py:279-288)
        NL = EV2G_GEN_FLEET_MAX;
        r.lut.resize((size_t)NL * EV2G_LUT_LEN);
        const int levels[6] = {6, 8, 10, 12, 14, 16};
        for (int f = 0; f < NL; f++)
            for (int i = 0; i < EV2G_LUT_LEN; i++) {
                int best = 0;
                for (int l = 1; l < 6; l++) if (std::abs(levels[l] - i) < std::abs(levels[best] - i)) best = l;
                r.lut[(size_t)f * EV2G_LUT_LEN + i] = EV2G_FLEET_V2G_ETA[f][best];
            }
    }
    ev2g_scenario_batch &b = r.b;
    b.n_envs = M; b.n_steps = T; b.timescale = dt; b.n_chargers = C; b.ports_per_charger = npc_max; b.n_transformers = R; b.horizon = 20;
    b.n_dr_max = ND; b.n_lut = NL; b.n_sessions = (int64_t)S;
    b.cs_min_charge_current = r.cs_min_c.data(); b.cs_max_charge_current = r.cs_max_c.data(); b.cs_min_discharge_current = r.cs_min_d.data();
    b.cs_max_discharge_current = r.cs_max_d.data(); b.cs_voltage = r.cs_volt.data(); b.cs_phases = r.cs_phases.data(); b.cs_transformer = r.cs_tr.data();
    b.cs_n_ports = r.cs_np.data();
    b.charge_price = r.charge_price.data(); b.discharge_price = r.discharge_price.data(); b.power_setpoints = r.setpoints.data();
    b.tr_max_power = r.maxp.data(); b.tr_min_power = r.minp.data(); b.tr_inflexible_load = r.infl.data(); b.tr_solar_power = r.solar.data();
    b.tr_load_forecast = r.lf.data(); b.tr_pv_forecast = r.pvf.data(); b.tr_dr = r.dr.data(); b.tr_n_dr = r.n_dr.data(); b.tr_steps_ahead = r.steps_ahead.data();
    b.env_session_start = r.sess_start.data(); b.ev_cs = r.ev_cs.data(); b.ev_t_arr = r.ev_ta.data(); b.ev_t_dep = r.ev_td.data(); b.ev_phases = r.ev_ph.data();
    b.ev_lut = r.ev_lut.data(); b.ev_cap0 = r.cap0.data(); b.ev_B = r.B.data(); b.ev_desired = r.desired.data(); b.ev_minB = r.minB.data();
    b.ev_min_emerg = r.min_emerg.data(); b.ev_pac_max = r.pac_max.data(); b.ev_pac_min = r.pac_min.data(); b.ev_pdis_max = r.pdis_max.data();
    b.ev_pdis_min = r.pdis_min.data(); b.ev_ts = r.ts.data(); b.ev_tsm = r.tsm.data(); b.ev_eta_ch = r.eta_ch.data(); b.ev_eta_dis = r.eta_dis.data();
    b.lut = r.lut.data();
    *out = holder.release();
    return EV2G_OK;
}

// no exception leaves the C-ABI: allocation failures (a batch too large for the host) and thread-creation failures become error codes
static int ev2g_generate_impl(const ev2g_gen_config *cfg, int32_t M, uint64_t seed, int32_t n_threads, ev2g_gen_result **out) {
    try {
        return ev2g_generate_body(cfg, M, seed, n_threads, out);
    } catch (const std::exception &e) {
        if (out) *out = nullptr;
        g_create_error = std::string("ev2g_generate: ") + e.what();
        return EV2G_ERR_ARG;
    }
}

static int ev2g_gen_default_config_impl(int kind, ev2g_gen_config *c) {
    if (!c || kind < 0 || kind > 1) return EV2G_ERR_ARG;
    std::memset(c, 0, sizeof(*c));
    c->simulation_length = 112; c->timescale = 15; c->number_of_charging_stations = 25; c->number_of_ports_per_cs = 1; c->number_of_transformers = 1;
    c->scenario = 0; c->simulation_days = 0; c->hour = 5; c->minute = 0; c->random_hour = 0; c->v2g_enabled = 1; c->power_setpoint_enabled = 0;
    c->inflexible_loads = 1; c->solar_power = 1; c->demand_response = 1; c->dr_events_per_day = 1; c->dr_event_length_minutes_min = 60;
    c->dr_event_length_minutes_max = 60; c->dr_notification_of_event_minutes = 60; c->heterogeneous_ev_specs = 1; c->fleet_with_efficiency_tables = 1;
    c->fleet = 0; c->cs_phases = 3; c->ev_phases = 3; c->ev_min_time_of_stay = 180; c->tr_seed = -1;
    c->spawn_multiplier = 5; c->discharge_price_factor = 1; c->power_setpoint_flexiblity = 80;
    c->inflexible_loads_capacity_multiplier_mean = 1; c->inflexible_loads_forecast_mean = 30; c->inflexible_loads_forecast_std = 5;
    c->solar_power_capacity_multiplier_mean = 1; c->solar_power_forecast_mean = 20; c->solar_power_forecast_std = 5;
    c->dr_event_capacity_percentage_mean = 35; c->dr_event_capacity_percentage_std = 5; c->dr_event_start_hour_mean = 12; c->dr_event_start_hour_std = 2;
    c->transformer_max_power = 100; c->cs_min_charge_current = 0; c->cs_max_charge_current = 32; c->cs_min_discharge_current = 0; c->cs_max_discharge_current = -32;
    c->cs_voltage = 400; c->ev_battery_capacity = 50; c->ev_max_ac_charge_power = 11; c->ev_min_ac_charge_power = 0; c->ev_max_discharge_power = -11;
    c->ev_min_discharge_power = 0; c->ev_charge_efficiency = 1; c->ev_discharge_efficiency = 1; c->ev_transition_soc = 1; c->ev_transition_soc_multiplier = 5;
    c->ev_min_battery_capacity = 5; c->ev_min_emergency_battery_capacity = 25; c->ev_desired_capacity = 1;
    if (kind == 1) {   // PublicPST.yaml
        c->number_of_charging_stations = 20; c->scenario = 1; c->v2g_enabled = 0; c->power_setpoint_enabled = 1; c->inflexible_loads = 0; c->solar_power = 0;
        c->demand_response = 0; c->fleet_with_efficiency_tables = 0; c->fleet = 1; c->cs_max_charge_current = 16; c->cs_max_discharge_current = 0;
        c->ev_min_time_of_stay = 60;
    }
    return EV2G_OK;
}

static int ev2g_gen_table_impl(int which, int kind, double *out, int n_max) {
    if (!out) return EV2G_ERR_ARG;
    if (which >= 0 && which <= 2) {
        if (kind < 0 || kind >= EV2G_GEN_N_KINDS) return EV2G_ERR_ARG;
        const int n = which == 2 ? 1 : 24;
        if (n_max < n) return EV2G_ERR_ARG;
        for (int i = 0; i < n; i++) out[i] = which == 0 ? EV2G_GEN_RATE[kind][i] : (which == 1 ? EV2G_GEN_STAY[kind][i] : EV2G_GEN_ENERGY[kind]);
        return n;
    }
    if (which == 3 || which == 4) {
        if (n_max < EV2G_GEN_FLEET_MAX * 3) return EV2G_ERR_ARG;
        for (int i = 0; i < EV2G_GEN_FLEET_MAX; i++) for (int j = 0; j < 3; j++) out[i * 3 + j] = (which == 3 ? EV2G_FLEET_V2G : EV2G_FLEET_EV_PHEV)[i][j];
        return EV2G_GEN_FLEET_MAX * 3;
    }
    return EV2G_ERR_ARG;
}
