/*
 * ev2g_oracle.c -- CPU restatement of the reference EV2Gym.step() path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product path (ev2gym_amd + libev2g_hip.so) never does.  It is a scalar, object-per-EV,
 * same-operation-order restatement of the Python reference, pinned against the golden vectors in
 * tests/golden/ that oracle/capture_golden.py recorded from the reference itself
 * (tests/test_oracle_golden.py).  Each function cites the reference file:line it follows
 * (paths relative to /root/reference/ev2gym/).
 *
 * Build:  gcc -O2 -ffp-contract=off -fPIC -shared -o libev2g_oracle.so ev2g_oracle.c -lm
 * (-ffp-contract=off: an FMA in `soc*B` or `cap*100` flips EV.my_ceil)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ev2g.h"

/* ------------------------------------------------------------------------------------------- */
/* models/ev.py:11-113  EV */
typedef struct {
    int present;
    int id, location, session; /* id = port index inside the charger after spawn (ev_charger.py:274) */
    int t_arr, t_dep, ev_phases, lut;
    double cap0, B, desired, minB, min_emerg, pac_max, pac_min, pdis_max, pdis_min, ts, tsm, eta_ch, eta_dis;
    /* status (ev.py:93-113) */
    double current_capacity, prev_capacity, current_energy, actual_current, previous_power;
    double required_energy, total_energy_exchanged, abs_total_energy_exchanged, max_energy_AFAP;
    int charging_cycles, min_emergency_battery_capacity_metric;
    /* degradation logs (ev.py:108-109) */
    double *historic_soc;
    int *active_steps;
    int n_hist, n_active;
} EV;

/* models/ev_charger.py:9-94 */
typedef struct {
    int id, n_ports, phases, connected_transformer;
    double min_charge_current, max_charge_current, min_discharge_current, max_discharge_current, voltage;
    double current_power_output, current_total_amps;
    int current_step, n_evs_connected;
    EV **evs_connected; /* [n_ports] NULL = empty */
    double total_energy_charged, total_energy_discharged, total_profits, total_user_satisfaction;
    int total_evs_served;
} Charger;

/* models/transformer.py:9-78 */
typedef struct {
    int id, current_step, n_dr, steps_ahead;
    double voltage;
    double *max_power, *min_power, *inflexible_load, *solar_power; /* [T] */
    double *load_forecast, *pv_forecast;                           /* [T] mutable copies */
    const double *dr;                                              /* [ND,3] */
    double current_power, current_amps;
} Transformer;

typedef struct {
    int T, timescale, C, npc, P, R, H, D, ND;
    int reward_kind, state_kind;
    int current_step, done, total_evs_spawned, fault;
    double total_reward;
    Charger *cs;
    Transformer *tr;
    EV *profiles; /* EVs_profiles, arrival order */
    EV *evs;      /* env.EVs: spawned copies */
    int n_profiles, n_evs;
    const double *charge_price, *discharge_price, *setpoints;  /* [T] */
    double *current_power_usage, *charge_power_potential;     /* [T] */
    double *tr_overload;                                       /* [R,T] */
    double *cs_power, *cs_current;                             /* [C,T] */
    /* per-step scratch: departing EVs */
    double *sat_list;
    double *short_list; /* per departing EV of the step: 100 * (desired - capacity) when short, else 0 */
    double *short2_list; /* ... and (desired - capacity)^2 when short (the *V2 rewards, reward.py:197-207) */
    int n_sat;
    const double *lut; /* [NL,101] */
    int n_lut;
} Env;

typedef struct {
    int E;
    Env *env;
    /* owned copies of the batch arrays */
    void **owned;
    int n_owned;
    ev2g_scenario_batch b;
} Oracle;

static void *dup_arr(Oracle *o, const void *src, size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (bytes) memcpy(p, src, bytes);
    o->owned = (void **)realloc(o->owned, sizeof(void *) * (o->n_owned + 1));
    o->owned[o->n_owned++] = p;
    return p;
}

/* ev.py:188-189  my_ceil: np.true_divide(np.ceil(a * 10**2), 10**2) */
static double my_ceil2(double a) { return ceil(a * 100.0) / 100.0; }
/* ev.py:223-229 */
static double ev_get_soc(const EV *ev) { return ev->current_capacity / ev->B; }
/* ev.py:204-214 */
static double ev_get_user_satisfaction(const EV *ev) {
    if (ev->current_capacity < ev->desired - 0.001) return ev->current_capacity / ev->desired;
    return 1.0;
}
/* np.float64.__round__(5) == np.round(x, 5): rint(x*1e5)/1e5 (ev_charger.py:157) */
static double rnd5(double x) { return rint(x * 100000.0) / 100000.0; }

static double lut_get(const Env *env, int lut, double key) {
    /* dict.get(np.round(amps), 1): integer keys 0..100 exist, anything else -> 1 (ev.py:287-290) */
    if (key >= 0.0 && key <= 100.0) return env->lut[(size_t)lut * EV2G_LUT_LEN + (int)key];
    return 1.0;
}

/* ev.py:240-355  EV._charge */
static double ev_charge(const Env *env, EV *ev, double amps, double voltage, int phases) {
    double pilot = amps;
    voltage = voltage * sqrt((double)phases);
    double period = (double)env->timescale;
    double charge_efficiency;
    if (ev->lut >= 0)
        charge_efficiency = lut_get(env, ev->lut, rint(amps)) / 100.0; /* np.round = half-even */
    else
        charge_efficiency = ev->eta_ch;
    double pilot_dsoc = charge_efficiency * pilot * voltage / 1000.0 / ev->B / (60.0 / period);
    double max_dsoc = charge_efficiency * ev->pac_max / ev->B / (60.0 / period);
    if (pilot_dsoc > max_dsoc) pilot_dsoc = max_dsoc;
    double curr_soc;
    if (ev->ts == 1.0) {
        curr_soc = pilot_dsoc + ev_get_soc(ev);
        if (curr_soc > 1.0) curr_soc = 1.0;
    } else {
        double pts = ev->ts + (pilot_dsoc - max_dsoc) / max_dsoc * (ev->ts - 1.0);
        double new_soc;
        if (ev_get_soc(ev) < pts) {
            if (1.0 <= (pts - ev_get_soc(ev)) / pilot_dsoc)
                new_soc = pilot_dsoc + ev_get_soc(ev);
            else
                new_soc = 1.0 + exp(ev->tsm * (pilot_dsoc + ev_get_soc(ev) - pts) / (pts - 1.0)) * (pts - 1.0);
        } else {
            new_soc = 1.0 + exp(ev->tsm * pilot_dsoc / (pts - 1.0)) * (ev_get_soc(ev) - 1.0);
        }
        double dsoc_limit = (max_dsoc > pilot_dsoc) ? pilot_dsoc : max_dsoc;
        if (new_soc - ev_get_soc(ev) > dsoc_limit)
            curr_soc = dsoc_limit + ev_get_soc(ev);
        else
            curr_soc = new_soc;
    }
    double dsoc = curr_soc - ev_get_soc(ev);
    ev->prev_capacity = ev->current_capacity;
    ev->current_capacity = curr_soc * ev->B;
    ev->current_energy = dsoc * ev->B;
    ev->required_energy = ev->required_energy - ev->current_energy;
    return ev->current_energy / (period / 60.0) * 1000.0 / voltage;
}

/* ev.py:357-405  EV._discharge */
static double ev_discharge(const Env *env, EV *ev, double amps, double voltage, int phases) {
    voltage = voltage * sqrt((double)phases);
    double given_power = amps * voltage / 1000.0;
    double prev_capacity = ev->current_capacity;
    if (fabs(given_power) > fabs(ev->pdis_max)) given_power = ev->pdis_max;
    double discharge_efficiency;
    if (ev->lut >= 0) /* keyed on isinstance(self.charge_efficiency, dict), ev.py:375 */
        discharge_efficiency = lut_get(env, ev->lut, fabs(rint(amps))) / 100.0;
    else
        discharge_efficiency = ev->eta_dis;
    double ts = (double)env->timescale;
    double given_energy = given_power * discharge_efficiency * ts / 60.0;
    if (ev->current_capacity + given_energy < ev->minB) {
        if (ev->current_capacity > ev->minB) {
            ev->current_energy = -(ev->current_capacity - ev->minB);
            given_energy = ev->current_energy;
            ev->prev_capacity = ev->current_capacity;
            ev->current_capacity = ev->minB;
        } else {
            ev->current_energy = 0.0;
            given_energy = 0.0;
            ev->prev_capacity = ev->current_capacity;
            ev->current_capacity = ev->minB;
        }
    } else {
        ev->current_energy = given_energy;
        ev->prev_capacity = ev->current_capacity;
        ev->current_capacity += given_energy;
    }
    ev->required_energy = ev->required_energy + ev->current_energy;
    if (prev_capacity > ev->min_emerg && ev->current_capacity < ev->min_emerg)
        ev->min_emergency_battery_capacity_metric += 1;
    return given_energy * 60.0 / ts * 1000.0 / voltage;
}

/* ev.py:138-186  EV.step */
static void ev_step(const Env *env, EV *ev, double amps, double voltage, int phases, double *energy, double *current) {
    if (amps > 0 && amps < ev->pac_min * 1000.0 / (voltage * sqrt((double)phases)))
        amps = 0;
    else if (amps < 0 && amps > ev->pdis_min * 1000.0 / (voltage * sqrt((double)phases)))
        amps = 0;
    ev->historic_soc[ev->n_hist++] = ev_get_soc(ev);
    if (amps == 0) {
        ev->current_energy = 0;
        ev->actual_current = 0;
        ev->active_steps[ev->n_active++] = 0;
        *energy = 0;
        *current = 0;
        return;
    }
    if (ev->previous_power == 0 || (ev->previous_power / amps) < 0) ev->charging_cycles += 1;
    if (ev->ev_phases < phases) phases = ev->ev_phases;
    if (amps > 0)
        ev->actual_current = ev_charge(env, ev, amps, voltage, phases);
    else
        ev->actual_current = ev_discharge(env, ev, amps, voltage, phases);
    ev->previous_power = ev->current_energy;
    ev->total_energy_exchanged += ev->current_energy;
    ev->abs_total_energy_exchanged += fabs(ev->current_energy);
    ev->current_capacity = my_ceil2(ev->current_capacity);
    ev->active_steps[ev->n_active++] = (ev->actual_current != 0) ? 1 : 0;
    *energy = ev->current_energy;
    *current = ev->actual_current;
}

/* ev.py:407-440  calculate_max_energy_with_AFAP */
static void ev_calc_afap(const Env *env, EV *ev, double max_cs_power) {
    double max_power = (fabs(max_cs_power) > fabs(ev->pac_max)) ? ev->pac_max : max_cs_power;
    ev->max_energy_AFAP = ev->cap0;
    double eff;
    if (ev->lut >= 0) {
        double m = 0;
        for (int k = 0; k < EV2G_LUT_LEN; k++) {
            double v = env->lut[(size_t)ev->lut * EV2G_LUT_LEN + k];
            if (v > m) m = v;
        }
        eff = m / 100.0;
    } else
        eff = ev->eta_ch;
    for (int k = ev->t_arr; k < ev->t_dep + 1; k++) {
        ev->max_energy_AFAP += max_power * eff * env->timescale / 60.0;
        ev->max_energy_AFAP = my_ceil2(ev->max_energy_AFAP);
        if (ev->max_energy_AFAP > ev->B) {
            ev->max_energy_AFAP = ev->B;
            break;
        }
    }
}

/* ev_charger.py:251-252 */
static double cs_get_max_power(const Charger *cs) { return cs->max_charge_current * cs->voltage * sqrt((double)cs->phases) / 1000.0; }

/* ev_charger.py:114-233  EV_Charger.step.  `actions` is the caller's slice, mutated in place. */
static void cs_step(Env *env, Charger *cs, double *actions, double charge_price, double discharge_price,
                    double *profit_out, int *invalid_out) {
    double profit = 0;
    cs->current_power_output = 0;
    cs->current_total_amps = 0;
    int invalid = 0;
    for (int i = 0; i < cs->n_ports; i++)
        if (cs->evs_connected[i] == NULL) {
            actions[i] = 0;
            invalid += 1;
        }
    double norm[64];
    double s = 0; /* python sum(): 0 + a0 + a1 ... */
    for (int i = 0; i < cs->n_ports; i++) s = s + actions[i];
    for (int i = 0; i < cs->n_ports; i++) {
        if (s > 1)
            norm[i] = actions[i] / s;
        else if (s < -1)
            norm[i] = -actions[i] / s;
        else
            norm[i] = actions[i];
    }
    for (int i = 0; i < cs->n_ports; i++) {
        double actual_energy = 0, actual_amps = 0;
        double action = rnd5(norm[i]);
        double amps = 0;
        if (action == 0 && cs->evs_connected[i] != NULL) {
            ev_step(env, cs->evs_connected[i], amps, cs->voltage, 1, &actual_energy, &actual_amps);
        } else if (action > 0) {
            amps = action * cs->max_charge_current;
            if (amps < cs->min_charge_current - 0.01) amps = 0;
            ev_step(env, cs->evs_connected[i], amps, cs->voltage, cs->phases, &actual_energy, &actual_amps);
            profit += fabs(actual_energy) * charge_price;
            cs->total_energy_charged += fabs(actual_energy);
            cs->current_power_output += actual_energy * 60 / env->timescale;
            cs->current_total_amps += actual_amps;
        } else if (action < 0) {
            amps = action * fabs(cs->max_discharge_current);
            if (amps > cs->min_discharge_current - 0.01) amps = cs->min_discharge_current;
            ev_step(env, cs->evs_connected[i], amps, cs->voltage, cs->phases, &actual_energy, &actual_amps);
            profit += fabs(actual_energy) * discharge_price;
            cs->total_energy_discharged += fabs(actual_energy);
            cs->current_power_output += actual_energy * 60 / env->timescale;
            cs->current_total_amps += actual_amps;
        }
        if (cs->current_total_amps - 0.0001 > cs->max_charge_current) env->fault = EV2G_ERR_OVERCURRENT; /* :203-205 raise */
    }
    cs->total_profits += profit;
    /* departures :207-229; EV.is_departing ev.py:191-202 */
    for (int i = 0; i < cs->n_ports; i++) {
        EV *ev = cs->evs_connected[i];
        if (ev != NULL && !(cs->current_step < ev->t_dep)) {
            cs->evs_connected[i] = NULL;
            cs->n_evs_connected -= 1;
            cs->total_evs_served += 1;
            double sat = ev_get_user_satisfaction(ev);
            cs->total_user_satisfaction += sat;
            /* what V2G_profitmax charges a departing EV (reward.py:130-136), next to its satisfaction score */
            env->short_list[env->n_sat] = (ev->desired > ev->current_capacity) ? 100 * (ev->desired - ev->current_capacity) : 0.0;
            env->short2_list[env->n_sat] = (ev->desired > ev->current_capacity) ? (ev->desired - ev->current_capacity) * (ev->desired - ev->current_capacity) : 0.0;
            env->sat_list[env->n_sat++] = sat;
        }
    }
    cs->current_step += 1;
    *profit_out = profit;
    *invalid_out = invalid;
}

/* transformer.py:258-267 */
static void tr_reset(Transformer *tr, int step) {
    tr->current_step = step;
    tr->current_power = tr->inflexible_load[step] + tr->solar_power[step];
    tr->current_amps = (tr->current_power * 1000.0) / tr->voltage;
}
/* transformer.py:276-302 */
static double tr_get_how_overloaded(const Transformer *tr) {
    double e = 0.0001;
    if (tr->current_power > tr->max_power[tr->current_step] + e || tr->current_power < tr->min_power[tr->current_step] - e)
        return fabs(tr->current_power - tr->max_power[tr->current_step]);
    return 0;
}
/* transformer.py:142-171 */
static void tr_get_power_limits(const Env *env, const Transformer *tr, int step, int horizon, double *out) {
    double power_limit = tr->max_power[0];
    for (int k = 1; k < env->T; k++)
        if (tr->max_power[k] > power_limit) power_limit = tr->max_power[k];
    for (int j = 0; j < horizon; j++) out[j] = power_limit * 1.0;
    for (int k = 0; k < tr->n_dr; k++) {
        int es = (int)tr->dr[k * 3 + 0], ee = (int)tr->dr[k * 3 + 1];
        double cap = tr->dr[k * 3 + 2];
        if (step + tr->steps_ahead >= es && ee >= step) {
            double v = power_limit - power_limit * cap / 100.0;
            int a, b;
            if (step > es) {
                a = 0;
                b = ee - step;
            } else {
                a = abs(es - step);
                b = abs(ee - step);
            }
            if (b > horizon) b = horizon;
            for (int j = a; j < b; j++) out[j] = v;
        }
    }
}
/* transformer.py:173-188 (writes the actual values into the forecast arrays through numpy views) */
static void tr_get_load_pv_forecast(const Env *env, Transformer *tr, int step, int horizon, double *loads, double *pv) {
    int T = env->T, n = 0;
    if (step < T) {
        tr->load_forecast[step] = tr->inflexible_load[step];
        tr->pv_forecast[step] = tr->solar_power[step];
    }
    for (int k = step; k < step + horizon && k < T; k++, n++) {
        loads[n] = tr->load_forecast[k];
        pv[n] = tr->pv_forecast[k];
    }
    for (; n < horizon; n++) {
        loads[n] = 1.0 * tr->load_forecast[T - 1];
        pv[n] = 1.0 * tr->pv_forecast[T - 1];
    }
}

/* rl_agent/state.py */
static void get_observation(Env *env, double *obs) {
    int n = 0, s = env->current_step, T = env->T;
    double usage_prev = env->current_power_usage[(s - 1 + T) % T]; /* index -1 wraps at reset (state.py:117) */
    if (env->state_kind == EV2G_STATE_PUBLIC_PST) { /* state.py:6-63 */
        obs[n++] = (double)s / (double)T;
        obs[n++] = (s < T) ? env->setpoints[s] : 0.0;
        obs[n++] = usage_prev;
        for (int t = 0; t < env->R; t++)
            for (int c = 0; c < env->C; c++) {
                Charger *cs = &env->cs[c];
                if (cs->connected_transformer != env->tr[t].id) continue;
                for (int j = 0; j < cs->n_ports; j++) {
                    EV *ev = cs->evs_connected[j];
                    if (ev) {
                        obs[n++] = (ev_get_soc(ev) == 1.0) ? 1.0 : 0.5;
                        obs[n++] = ev->total_energy_exchanged;
                        obs[n++] = (double)(s - ev->t_arr);
                    } else {
                        obs[n++] = 0;
                        obs[n++] = 0;
                        obs[n++] = 0;
                    }
                }
            }
        return;
    }
    /* V2G_profit_max_loads state.py:108-155 / V2G_profit_max :65-106 */
    obs[n++] = (double)s;
    obs[n++] = usage_prev;
    for (int j = 0; j < 20; j++) obs[n++] = (s + j < T) ? fabs(env->charge_price[s + j]) : 0.0;
    for (int t = 0; t < env->R; t++) {
        Transformer *tr = &env->tr[t];
        if (env->state_kind == EV2G_STATE_V2G_PROFIT_MAX_LOADS) {
            double loads[64], pv[64], lim[64];
            tr_get_load_pv_forecast(env, tr, s, 20, loads, pv);
            tr_get_power_limits(env, tr, s, 20, lim);
            for (int j = 0; j < 20; j++) obs[n++] = loads[j] - pv[j];
            for (int j = 0; j < 20; j++) obs[n++] = lim[j];
        }
        for (int c = 0; c < env->C; c++) {
            Charger *cs = &env->cs[c];
            if (cs->connected_transformer != tr->id) continue;
            for (int j = 0; j < cs->n_ports; j++) {
                EV *ev = cs->evs_connected[j];
                if (ev) {
                    obs[n++] = ev_get_soc(ev);
                    obs[n++] = (double)(ev->t_dep - s);
                } else {
                    obs[n++] = 0;
                    obs[n++] = 0;
                }
            }
        }
    }
}

/* utilities/utils.py:760-791 */
static double calculate_charge_power_potential(const Env *env) {
    double power_potential = 0;
    for (int c = 0; c < env->C; c++) {
        const Charger *cs = &env->cs[c];
        double cs_pp = 0;
        for (int p = 0; p < cs->n_ports; p++) {
            const EV *ev = cs->evs_connected[p];
            if (ev != NULL && ev_get_soc(ev) < 1 && ev->t_dep > env->current_step) {
                int phases = cs->phases < ev->ev_phases ? cs->phases : ev->ev_phases;
                double ev_current = ev->pac_max * 1000 / (sqrt((double)phases) * cs->voltage);
                double current = (ev_current < cs->max_charge_current) ? ev_current : cs->max_charge_current; /* min(a,b) */
                cs_pp += sqrt((double)phases) * cs->voltage * current / 1000;
            }
        }
        double max_cs_power = sqrt((double)cs->phases) * cs->voltage * cs->max_charge_current / 1000;
        double min_cs_power = sqrt((double)cs->phases) * cs->voltage * cs->min_charge_current / 1000;
        if (cs_pp > max_cs_power)
            power_potential += max_cs_power;
        else if (cs_pp < min_cs_power)
            power_potential += 0;
        else
            power_potential += cs_pp;
    }
    return power_potential;
}

/* rl_agent/reward.py */
static double calculate_reward(Env *env, double total_costs) {
    int t1 = env->current_step - 1;
    double reward;
    switch (env->reward_kind) {
    case EV2G_REWARD_SQUARED_TRACKING_ERROR: { /* reward.py:7-14; python min(a,b): b if b<a else a */
        double a = env->setpoints[t1], b = env->charge_power_potential[t1];
        double m = (b < a) ? b : a;
        double d = m - env->current_power_usage[t1];
        reward = -(d * d);
        break;
    }
    case EV2G_REWARD_PROFIT_MAXIMIZATION: /* reward.py:78-87 */
        reward = total_costs;
        for (int k = 0; k < env->n_sat; k++) reward -= 100 * exp(-10 * env->sat_list[k]);
        break;
    case EV2G_REWARD_SQTR_TRPENALTY_USERINCENTIVES: { /* reward.py:16-32; python min(a,b,c) keeps the first of equal values */
        double m = env->setpoints[t1];
        if (env->charge_power_potential[t1] < m) m = env->charge_power_potential[t1];
        if (env->tr[0].max_power[t1] < m) m = env->tr[0].max_power[t1];
        double d = m - env->current_power_usage[t1];
        reward = -(d * d);
        for (int t = 0; t < env->R; t++) reward -= 100 * tr_get_how_overloaded(&env->tr[t]);
        for (int k = 0; k < env->n_sat; k++) reward -= 1000 * (1 - env->sat_list[k]);
        break;
    }
    case EV2G_REWARD_SQUARED_TRACKING_ERROR_PENALTY: { /* reward.py:46-58; index current_step-2 wraps like a python list */
        double a = env->setpoints[t1], b = env->charge_power_potential[t1];
        double m = (b < a) ? b : a;
        double d = m - env->current_power_usage[t1];
        int t2 = t1 - 1 < 0 ? env->T - 1 : t1 - 1;
        reward = -(d * d);
        if (env->current_power_usage[t1] == 0 && env->charge_power_potential[t2] != 0) reward -= 100;
        break;
    }
    case EV2G_REWARD_SIMPLE: { /* reward.py:60-65 */
        double d = env->setpoints[t1] - env->current_power_usage[t1];
        reward = -(d * d);
        break;
    }
    case EV2G_REWARD_MINIMIZE_TRACKER_SURPLUS: { /* reward.py:67-76 */
        double u = env->current_power_usage[t1], sp = env->setpoints[t1];
        reward = 0;
        if (sp < u) reward -= (u - sp) * (u - sp);
        reward += u;
        break;
    }
    case EV2G_REWARD_V2G_COSTS_SIMPLE: /* reward.py:151-154 */
        reward = total_costs;
        break;
    case EV2G_REWARD_V2G_PROFITMAX: { /* reward.py:120-148 */
        double user_costs = 0;
        for (int k = 0; k < env->n_sat; k++) user_costs += -env->short_list[k];
        reward = total_costs + user_costs;
        break;
    }
    case EV2G_REWARD_V2G_PROFITMAX_V2:       /* reward.py:156-211 */
    case EV2G_REWARD_PST_V2G_PROFITMAX_V2: { /* reward.py:278-339 */
        double user_costs = 0;
        const double cost_multiplier = 0.05;
        for (int c = 0; c < env->C; c++) {
            Charger *cs = &env->cs[c];
            for (int j = 0; j < cs->n_ports; j++) {
                EV *ev = cs->evs_connected[j];
                if (ev == NULL) continue;
                double spd = 60.0 / env->timescale;
                double min_steps_to_full = (ev->desired - ev->current_capacity) / (ev->pac_max / spd);
                double departing_step = (double)(ev->t_dep - env->current_step);
                if (min_steps_to_full > departing_step) {
                    /* python: desired - ((departing_step+1) * max_ac_charge_power/(60/timescale)): product first, then the division */
                    double min_capacity_at_time = ev->desired - ((departing_step + 1) * ev->pac_max / spd);
                    double gap = min_capacity_at_time - ev->current_capacity;
                    double cost = cost_multiplier * (gap * gap); /* python: cost_multiplier * (gap)**2 */
                    user_costs += -cost;
                }
            }
        }
        for (int k = 0; k < env->n_sat; k++) user_costs += -cost_multiplier * env->short2_list[k];
        reward = total_costs + user_costs;
        if (env->reward_kind == EV2G_REWARD_PST_V2G_PROFITMAX_V2) {
            double pst_violation = 0;
            if (env->setpoints[t1] < env->current_power_usage[t1]) pst_violation += env->setpoints[t1] - env->current_power_usage[t1];
            reward = reward + 1000 * pst_violation;
        }
        break;
    }
    default: /* ProfitMax_TrPenalty_UserIncentives reward.py:34-44 */
        reward = total_costs;
        for (int t = 0; t < env->R; t++) reward -= 100 * tr_get_how_overloaded(&env->tr[t]);
        for (int k = 0; k < env->n_sat; k++) reward -= 100 * exp(-10 * env->sat_list[k]);
        break;
    }
    env->total_reward += reward;
    return reward;
}

/* ev2gym_env.py:298-306,329-331 + utils.py:794-861 + ev_charger.py:96-112 */
static void env_reset(Oracle *o, Env *env, int e, double *obs) {
    const ev2g_scenario_batch *b = &o->b;
    int T = env->T;
    env->current_step = 0;
    env->done = 0;
    env->total_evs_spawned = 0;
    env->total_reward = 0;
    env->n_evs = 0;
    env->fault = 0;
    memset(env->current_power_usage, 0, sizeof(double) * T);
    memset(env->charge_power_potential, 0, sizeof(double) * T);
    memset(env->tr_overload, 0, sizeof(double) * T * env->R);
    memset(env->cs_power, 0, sizeof(double) * T * env->C);
    memset(env->cs_current, 0, sizeof(double) * T * env->C);
    for (int c = 0; c < env->C; c++) {
        Charger *cs = &env->cs[c];
        cs->current_power_output = 0;
        cs->current_total_amps = 0;
        for (int j = 0; j < cs->n_ports; j++) cs->evs_connected[j] = NULL;
        cs->n_evs_connected = 0;
        cs->current_step = 0;
        cs->total_energy_charged = cs->total_energy_discharged = cs->total_profits = cs->total_user_satisfaction = 0;
        cs->total_evs_served = 0;
    }
    for (int t = 0; t < env->R; t++) {
        Transformer *tr = &env->tr[t];
        size_t off = ((size_t)e * env->R + t) * T;
        /* a new episode starts from the forecasts as they stood after the reference's reset() */
        memcpy(tr->load_forecast, b->tr_load_forecast + off, sizeof(double) * T);
        memcpy(tr->pv_forecast, b->tr_pv_forecast + off, sizeof(double) * T);
        tr_reset(tr, 0);
    }
    if (obs) get_observation(env, obs);
}

/* ev2gym_env.py:333-447 */
static void env_step(Env *env, double *actions, double *obs, double *reward_out, uint8_t *done, uint8_t *mask) {
    double total_costs = 0;
    int total_invalid = 0;
    env->n_sat = 0;
    int port_counter = 0;
    int t = env->current_step;
    for (int r = 0; r < env->R; r++) tr_reset(&env->tr[r], t);
    for (int i = 0; i < env->C; i++) {
        Charger *cs = &env->cs[i];
        double costs;
        int invalid;
        cs_step(env, cs, actions + port_counter, env->charge_price[t], env->discharge_price[t], &costs, &invalid);
        env->current_power_usage[t] += cs->current_power_output;
        Transformer *tr = &env->tr[cs->connected_transformer]; /* transformer.py:269-274 */
        tr->current_amps += cs->current_total_amps;
        tr->current_power += cs->current_power_output;
        total_costs += costs;
        total_invalid += invalid;
        port_counter += cs->n_ports;
    }
    /* spawn :399-417 */
    int counter = env->total_evs_spawned;
    for (int k = counter; k < env->n_profiles; k++) {
        const EV *prof = &env->profiles[k];
        if (prof->t_arr == t + 1) {
            EV *ev = &env->evs[env->n_evs];
            double *hs = ev->historic_soc;
            int *as = ev->active_steps;
            *ev = *prof; /* deepcopy + reset (ev.py:115-136) */
            ev->historic_soc = hs;
            ev->active_steps = as;
            ev->n_hist = ev->n_active = 0;
            ev->current_capacity = ev->cap0;
            ev->prev_capacity = ev->current_capacity;
            ev->current_energy = ev->actual_current = ev->previous_power = 0;
            ev->charging_cycles = 0;
            ev->required_energy = ev->B - ev->cap0;
            ev->total_energy_exchanged = ev->abs_total_energy_exchanged = 0;
            ev->max_energy_AFAP = 0;
            ev->min_emergency_battery_capacity_metric = 0;
            Charger *cs = &env->cs[ev->location]; /* spawn_ev ev_charger.py:266-286 */
            int index = -1;
            for (int j = 0; j < cs->n_ports; j++)
                if (cs->evs_connected[j] == NULL) {
                    index = j;
                    break;
                }
            if (index < 0) {
                env->fault = EV2G_ERR_ARG;
                break;
            }
            ev->id = index;
            cs->evs_connected[index] = ev;
            cs->n_evs_connected += 1;
            ev_calc_afap(env, ev, cs_get_max_power(cs));
            env->total_evs_spawned += 1;
            env->n_evs += 1;
        } else if (prof->t_arr > t + 1)
            break;
    }
    /* _update_power_statistics :520-556 */
    for (int r = 0; r < env->R; r++) env->tr_overload[(size_t)r * env->T + t] = tr_get_how_overloaded(&env->tr[r]);
    for (int i = 0; i < env->C; i++) {
        env->cs_power[(size_t)i * env->T + t] = env->cs[i].current_power_output;
        env->cs_current[(size_t)i * env->T + t] = env->cs[i].current_total_amps;
    }
    env->current_step += 1;
    if (env->current_step < env->T) env->charge_power_potential[env->current_step] = calculate_charge_power_potential(env);
    double reward = calculate_reward(env, total_costs);
    (void)total_invalid;
    /* _check_termination :449-496 */
    if (mask) {
        for (int p = 0; p < env->P; p++) mask[p] = 0;
        for (int i = 0; i < env->C; i++)
            for (int j = 0; j < env->cs[i].n_ports; j++)
                if (env->cs[i].evs_connected[j] != NULL) {
                    /* the reference indexes i*cs.n_ports + j, not the cumulative port number (ev2gym_env.py:452-457): the same
                       thing for equal port counts; with a topology file the entries collide or fall outside the array, where
                       numpy raises IndexError */
                    const int m = i * env->cs[i].n_ports + j;
                    if (m < env->P) mask[m] = 1;
                    else env->fault = EV2G_ERR_ARG;
                }
    }
    if (env->current_step >= env->T) env->done = 1;
    if (obs) get_observation(env, obs);
    if (reward_out) *reward_out = reward;
    if (done) *done = (uint8_t)env->done;
}

/* ev.py:442-521 get_battery_degradation (np.mean restated as a plain left-to-right mean) */
static void ev_get_battery_degradation(const Env *env, EV *ev, double *d_cal_out, double *d_cyc_out) {
    const double e0 = 7.543e6, e1 = 23.75e6, e2 = 6976, z0 = 7.348e-3, z1 = 3.667, z2 = 7.6e-4, z3 = 4.081e-3;
    const double b_cap_ah = 2.05, b_cap_kwh = 78, d_dist = 15000, b_age = 2 * 365, G = 0.186;
    double T_acc = b_age;
    double T_sim = (ev->t_dep - ev->t_arr + 1) * (double)env->timescale / (60 * 24);
    double theta = 298.15, k = 0.8263, v_min = 3.3324;
    ev->historic_soc[ev->n_hist++] = ev_get_soc(ev);
    double sum = 0;
    for (int i = 0; i < ev->n_hist; i++) sum += ev->historic_soc[i];
    double avg_soc = sum / ev->n_hist;
    double v_avg = v_min + k * avg_soc;
    double alpha = (e0 * v_avg - e1) * exp(-e2 / theta);
    double d_cal = alpha * 0.75 * T_sim / pow(T_acc, 0.25);
    ev->active_steps[ev->n_active++] = 1;
    double fs = 0;
    int nf = 0;
    for (int i = 0; i < ev->n_hist; i++)
        if (ev->active_steps[i] == 1) {
            fs += ev->historic_soc[i];
            nf++;
        }
    double avg_f = fs / nf;
    double mad = 0;
    for (int i = 0; i < ev->n_hist; i++)
        if (ev->active_steps[i] == 1) mad += fabs(avg_f - ev->historic_soc[i]);
    double delta_DoD = 2 * (mad / nf);
    double v_half_soc = v_min + k * 0.5;
    double beta = z0 * (v_half_soc - z1) * (v_half_soc - z1) + z2 + z3 * delta_DoD;
    double Q_sim = (ev->abs_total_energy_exchanged / b_cap_kwh) * b_cap_ah;
    double Q_acc = 2 * (b_age * (d_dist / 365) * G * b_cap_ah) / b_cap_kwh;
    double d_cyc = beta * 0.5 * Q_sim / pow(Q_acc, 0.5);
    *d_cal_out = d_cal;
    *d_cyc_out = d_cyc;
}

/* utilities/utils.py:12-123 get_statistics; out[EV2G_N_STATS] in the order of the returned dict */
static void env_get_statistics(Env *env, double *out) {
    double served = 0, profits = 0, e_ch = 0, e_dis = 0, sat_sum = 0;
    int n_sat_cs = 0;
    for (int c = 0; c < env->C; c++) {
        Charger *cs = &env->cs[c];
        served += cs->total_evs_served;
        profits += cs->total_profits;
        e_ch += cs->total_energy_charged;
        e_dis += cs->total_energy_discharged;
        if (cs->total_evs_served > 0) {
            sat_sum += cs->total_user_satisfaction / cs->total_evs_served;
            n_sat_cs++;
        }
    }
    double avg_sat = n_sat_cs ? sat_sum / n_sat_cs : NAN;
    double tr_over = 0;
    for (size_t i = 0; i < (size_t)env->R * env->T; i++) tr_over += env->tr_overload[i];
    double tracking_error = 0, energy_tracking_error = 0, power_tracker_violation = 0;
    for (int t = 0; t < env->T; t++) {
        double d = env->setpoints[t] - env->current_power_usage[t];
        tracking_error += d * d;
        energy_tracking_error += fabs(d);
        if (env->current_power_usage[t] > env->setpoints[t]) power_tracker_violation += env->current_power_usage[t] - env->setpoints[t];
    }
    energy_tracking_error *= (double)env->timescale / 60;
    double deg_cal = 0, deg_cyc = 0;
    for (int i = 0; i < env->n_evs; i++) {
        double a, b;
        ev_get_battery_degradation(env, &env->evs[i], &a, &b);
        deg_cal += a;
        deg_cyc += b;
    }
    double eus_mean = NAN, eus_std = NAN, eus_min = NAN;
    int viol = 0;
    if (env->n_evs > 0) {
        double s = 0, mn = INFINITY;
        for (int i = 0; i < env->n_evs; i++) {
            double v = (env->evs[i].current_capacity / env->evs[i].max_energy_AFAP) * 100;
            s += v;
            if (v < mn) mn = v;
            viol += env->evs[i].min_emergency_battery_capacity_metric;
        }
        eus_mean = s / env->n_evs;
        double var = 0;
        for (int i = 0; i < env->n_evs; i++) {
            double v = (env->evs[i].current_capacity / env->evs[i].max_energy_AFAP) * 100 - eus_mean;
            var += v * v;
        }
        eus_std = sqrt(var / env->n_evs);
        eus_min = mn;
    }
    int k = 0;
    out[k++] = served;
    out[k++] = profits;
    out[k++] = e_ch;
    out[k++] = e_dis;
    out[k++] = avg_sat;
    out[k++] = power_tracker_violation;
    out[k++] = tracking_error;
    out[k++] = energy_tracking_error;
    out[k++] = eus_mean;
    out[k++] = eus_std;
    out[k++] = eus_min;
    out[k++] = viol;
    out[k++] = tr_over;
    out[k++] = deg_cal + deg_cyc;
    out[k++] = deg_cal;
    out[k++] = deg_cyc;
    out[k++] = env->total_reward;
}

/* ------------------------------------------------------------------------------------------- */
/* exported test API */

void *ev2g_oracle_create(const ev2g_scenario_batch *bin, int reward_kind, int state_kind) {
    Oracle *o = (Oracle *)calloc(1, sizeof(Oracle));
    ev2g_scenario_batch *b = &o->b;
    *b = *bin;
    int E = b->n_envs, T = b->n_steps, C = b->n_chargers, R = b->n_transformers, ND = b->n_dr_max;
    int64_t S = b->n_sessions;
#define DUP(f, n, type) b->f = (const type *)dup_arr(o, bin->f, sizeof(type) * (size_t)(n))
    DUP(cs_min_charge_current, C, double);
    DUP(cs_max_charge_current, C, double);
    DUP(cs_min_discharge_current, C, double);
    DUP(cs_max_discharge_current, C, double);
    DUP(cs_voltage, C, double);
    DUP(cs_phases, C, int32_t);
    DUP(cs_transformer, C, int32_t);
    if (bin->cs_n_ports) DUP(cs_n_ports, C, int32_t);
    DUP(charge_price, (size_t)E * T, double);
    DUP(discharge_price, (size_t)E * T, double);
    DUP(power_setpoints, (size_t)E * T, double);
    DUP(tr_max_power, (size_t)E * R * T, double);
    DUP(tr_min_power, (size_t)E * R * T, double);
    DUP(tr_inflexible_load, (size_t)E * R * T, double);
    DUP(tr_solar_power, (size_t)E * R * T, double);
    DUP(tr_load_forecast, (size_t)E * R * T, double);
    DUP(tr_pv_forecast, (size_t)E * R * T, double);
    DUP(tr_dr, (size_t)E * R * ND * 3, double);
    DUP(tr_n_dr, (size_t)E * R, int32_t);
    DUP(tr_steps_ahead, (size_t)E * R, int32_t);
    DUP(env_session_start, E + 1, int64_t);
    DUP(ev_cs, S, int32_t);
    DUP(ev_t_arr, S, int32_t);
    DUP(ev_t_dep, S, int32_t);
    DUP(ev_phases, S, int32_t);
    DUP(ev_lut, S, int32_t);
    DUP(ev_cap0, S, double);
    DUP(ev_B, S, double);
    DUP(ev_desired, S, double);
    DUP(ev_minB, S, double);
    DUP(ev_min_emerg, S, double);
    DUP(ev_pac_max, S, double);
    DUP(ev_pac_min, S, double);
    DUP(ev_pdis_max, S, double);
    DUP(ev_pdis_min, S, double);
    DUP(ev_ts, S, double);
    DUP(ev_tsm, S, double);
    DUP(ev_eta_ch, S, double);
    DUP(ev_eta_dis, S, double);
    DUP(lut, (size_t)b->n_lut * EV2G_LUT_LEN, double);
#undef DUP
    o->E = E;
    o->env = (Env *)calloc(E, sizeof(Env));
    int npc = b->ports_per_charger;
    for (int e = 0; e < E; e++) {
        Env *env = &o->env[e];
        env->T = T;
        env->timescale = b->timescale;
        env->C = C;
        env->npc = npc;
        env->P = 0;
        for (int c = 0; c < C; c++) env->P += b->cs_n_ports ? b->cs_n_ports[c] : npc;   /* ev2gym_env.py:201-202 */
        env->R = R;
        env->H = b->horizon;
        env->ND = ND;
        env->reward_kind = reward_kind;
        env->state_kind = state_kind;
        env->D = (state_kind == EV2G_STATE_PUBLIC_PST)
                     ? 3 + 3 * env->P
                     : (state_kind == EV2G_STATE_V2G_PROFIT_MAX ? 22 + 2 * env->P : 22 + 40 * R + 2 * env->P);
        env->lut = b->lut;
        env->n_lut = b->n_lut;
        env->charge_price = b->charge_price + (size_t)e * T;
        env->discharge_price = b->discharge_price + (size_t)e * T;
        env->setpoints = b->power_setpoints + (size_t)e * T;
        env->cs = (Charger *)calloc(C, sizeof(Charger));
        for (int c = 0; c < C; c++) {
            Charger *cs = &env->cs[c];
            cs->id = c;
            cs->n_ports = b->cs_n_ports ? b->cs_n_ports[c] : npc;   /* topology file: per charger (loaders.py:330-331) */
            cs->phases = b->cs_phases[c];
            cs->connected_transformer = b->cs_transformer[c];
            cs->min_charge_current = b->cs_min_charge_current[c];
            cs->max_charge_current = b->cs_max_charge_current[c];
            cs->min_discharge_current = b->cs_min_discharge_current[c];
            cs->max_discharge_current = b->cs_max_discharge_current[c];
            cs->voltage = b->cs_voltage[c];
            cs->evs_connected = (EV **)calloc(cs->n_ports, sizeof(EV *));
        }
        env->tr = (Transformer *)calloc(R, sizeof(Transformer));
        for (int t = 0; t < R; t++) {
            Transformer *tr = &env->tr[t];
            size_t off = ((size_t)e * R + t) * T;
            tr->id = t;
            tr->max_power = (double *)(b->tr_max_power + off);
            tr->min_power = (double *)(b->tr_min_power + off);
            tr->inflexible_load = (double *)(b->tr_inflexible_load + off);
            tr->solar_power = (double *)(b->tr_solar_power + off);
            tr->load_forecast = (double *)malloc(sizeof(double) * T);
            tr->pv_forecast = (double *)malloc(sizeof(double) * T);
            tr->dr = b->tr_dr + ((size_t)e * R + t) * ND * 3;
            tr->n_dr = b->tr_n_dr[(size_t)e * R + t];
            tr->steps_ahead = b->tr_steps_ahead[(size_t)e * R + t];
            /* transformer.py:39-40: config voltage * sqrt(config phases); chargers without a topology
               file all carry the config values (loaders.py:359-360) */
            tr->voltage = b->cs_voltage[0] * sqrt((double)b->cs_phases[0]);
        }
        int64_t s0 = b->env_session_start[e], s1 = b->env_session_start[e + 1];
        env->n_profiles = (int)(s1 - s0);
        env->profiles = (EV *)calloc(env->n_profiles ? env->n_profiles : 1, sizeof(EV));
        env->evs = (EV *)calloc(env->n_profiles ? env->n_profiles : 1, sizeof(EV));
        for (int k = 0; k < env->n_profiles; k++) {
            EV *ev = &env->profiles[k];
            int64_t s = s0 + k;
            ev->session = k;
            ev->location = b->ev_cs[s];
            ev->t_arr = b->ev_t_arr[s];
            ev->t_dep = b->ev_t_dep[s];
            ev->ev_phases = b->ev_phases[s];
            ev->lut = b->ev_lut[s];
            ev->cap0 = b->ev_cap0[s];
            ev->B = b->ev_B[s];
            ev->desired = b->ev_desired[s];
            ev->minB = b->ev_minB[s];
            ev->min_emerg = b->ev_min_emerg[s];
            ev->pac_max = b->ev_pac_max[s];
            ev->pac_min = b->ev_pac_min[s];
            ev->pdis_max = b->ev_pdis_max[s];
            ev->pdis_min = b->ev_pdis_min[s];
            ev->ts = b->ev_ts[s];
            ev->tsm = b->ev_tsm[s];
            ev->eta_ch = b->ev_eta_ch[s];
            ev->eta_dis = b->ev_eta_dis[s];
            env->evs[k].historic_soc = (double *)malloc(sizeof(double) * (T + 4));
            env->evs[k].active_steps = (int *)malloc(sizeof(int) * (T + 4));
        }
        env->current_power_usage = (double *)calloc(T, sizeof(double));
        env->charge_power_potential = (double *)calloc(T, sizeof(double));
        env->tr_overload = (double *)calloc((size_t)T * R, sizeof(double));
        env->cs_power = (double *)calloc((size_t)T * C, sizeof(double));
        env->cs_current = (double *)calloc((size_t)T * C, sizeof(double));
        env->sat_list = (double *)calloc(env->P + 1, sizeof(double));
        env->short_list = (double *)calloc(env->P + 1, sizeof(double));
        env->short2_list = (double *)calloc(env->P + 1, sizeof(double));
        env_reset(o, env, e, NULL);
    }
    return o;
}

int ev2g_oracle_obs_dim(void *h) { return ((Oracle *)h)->env[0].D; }

void ev2g_oracle_reset(void *h, double *obs) {
    Oracle *o = (Oracle *)h;
    for (int e = 0; e < o->E; e++) env_reset(o, &o->env[e], e, obs ? obs + (size_t)e * o->env[e].D : NULL);
}

/* steps envs [e0,e1); actions [E,P] is mutated in place like the reference does */
int ev2g_oracle_step_range(void *h, int e0, int e1, double *actions, double *obs, double *reward, uint8_t *done, uint8_t *mask) {
    Oracle *o = (Oracle *)h;
    int rc = 0;
    for (int e = e0; e < e1; e++) {
        Env *env = &o->env[e];
        if (env->done) {
            rc = EV2G_ERR_DONE;
            continue;
        }
        env_step(env, actions + (size_t)e * env->P, obs ? obs + (size_t)e * env->D : NULL, reward ? reward + e : NULL,
                 done ? done + e : NULL, mask ? mask + (size_t)e * env->P : NULL);
        if (env->fault && !rc) rc = env->fault;
    }
    return rc;
}

/* k consecutive steps of envs [e0,e1) with actions[step] = actions + step*a_stride ([E,P] each); outputs hold the
 * last step.  One call per worker thread and episode: the multi-core leg of bench.py's cpu_baseline. */
int ev2g_oracle_run_range(void *h, int e0, int e1, int k, double *actions, long long a_stride, double *obs,
                          double *reward, uint8_t *done, uint8_t *mask) {
    int rc = 0;
    for (int s = 0; s < k; s++) {
        int r = ev2g_oracle_step_range(h, e0, e1, actions + (size_t)s * (size_t)a_stride, obs, reward, done, mask);
        if (r && !rc) rc = r;
    }
    return rc;
}

int ev2g_oracle_step(void *h, double *actions, double *obs, double *reward, uint8_t *done, uint8_t *mask) {
    return ev2g_oracle_step_range(h, 0, ((Oracle *)h)->E, actions, obs, reward, done, mask);
}

/* per-port state of one env after the last step (NaN / -1 for empty ports); any pointer may be NULL */
void ev2g_oracle_peek(void *h, int e, double *cap, double *energy, double *current, double *tot_e, double *req_e,
                      double *prev_power, int32_t *cycles, int32_t *session, double *cs_power, double *cs_amps,
                      double *cs_profits, double *cs_e_ch, double *cs_e_dis, double *tr_power, double *tr_amps,
                      double *tr_overload, double *usage, double *potential, int32_t *session_port, double *session_afap,
                      double *session_cap) {
    Oracle *o = (Oracle *)h;
    Env *env = &o->env[e];
    int base = 0;   /* first port of charger c in the cumulative numbering */
    for (int c = 0; c < env->C; c++) {
        Charger *cs = &env->cs[c];
        if (cs_power) cs_power[c] = cs->current_power_output;
        if (cs_amps) cs_amps[c] = cs->current_total_amps;
        if (cs_profits) cs_profits[c] = cs->total_profits;
        if (cs_e_ch) cs_e_ch[c] = cs->total_energy_charged;
        if (cs_e_dis) cs_e_dis[c] = cs->total_energy_discharged;
        for (int j = 0; j < cs->n_ports; j++) {
            int p = base + j;
            EV *ev = cs->evs_connected[j];
            if (cap) cap[p] = ev ? ev->current_capacity : NAN;
            if (energy) energy[p] = ev ? ev->current_energy : NAN;
            if (current) current[p] = ev ? ev->actual_current : NAN;
            if (tot_e) tot_e[p] = ev ? ev->total_energy_exchanged : NAN;
            if (req_e) req_e[p] = ev ? ev->required_energy : NAN;
            if (prev_power) prev_power[p] = ev ? ev->previous_power : NAN;
            if (cycles) cycles[p] = ev ? ev->charging_cycles : -1;
            if (session) session[p] = ev ? ev->session : -1;
        }
        base += cs->n_ports;
    }
    for (int t = 0; t < env->R; t++) {
        if (tr_power) tr_power[t] = env->tr[t].current_power;
        if (tr_amps) tr_amps[t] = env->tr[t].current_amps;
    }
    if (tr_overload) memcpy(tr_overload, env->tr_overload, sizeof(double) * env->R * env->T);
    if (usage) memcpy(usage, env->current_power_usage, sizeof(double) * env->T);
    if (potential) memcpy(potential, env->charge_power_potential, sizeof(double) * env->T);
    for (int k = 0; k < env->n_profiles; k++) {
        int spawned = k < env->n_evs;
        if (session_port && spawned) {
            int pb = 0;
            for (int c = 0; c < env->evs[k].location; c++) pb += env->cs[c].n_ports;
            session_port[k] = pb + env->evs[k].id;
        } else if (session_port) session_port[k] = -1;
        if (session_afap) session_afap[k] = spawned ? env->evs[k].max_energy_AFAP : NAN;
        if (session_cap) session_cap[k] = spawned ? env->evs[k].current_capacity : NAN;
    }
}

/* stats [E,EV2G_N_STATS]; call once, after the episode is done (appends to the SoC logs like the reference) */
void ev2g_oracle_stats(void *h, double *stats) {
    Oracle *o = (Oracle *)h;
    for (int e = 0; e < o->E; e++) env_get_statistics(&o->env[e], stats + (size_t)e * EV2G_N_STATS);
}

void ev2g_oracle_destroy(void *h) {
    Oracle *o = (Oracle *)h;
    for (int e = 0; e < o->E; e++) {
        Env *env = &o->env[e];
        for (int c = 0; c < env->C; c++) free(env->cs[c].evs_connected);
        for (int t = 0; t < env->R; t++) {
            free(env->tr[t].load_forecast);
            free(env->tr[t].pv_forecast);
        }
        for (int k = 0; k < env->n_profiles; k++) {
            free(env->evs[k].historic_soc);
            free(env->evs[k].active_steps);
        }
        free(env->cs);
        free(env->tr);
        free(env->profiles);
        free(env->evs);
        free(env->current_power_usage);
        free(env->charge_power_potential);
        free(env->tr_overload);
        free(env->cs_power);
        free(env->cs_current);
        free(env->sat_list);
        free(env->short_list);
        free(env->short2_list);
    }
    for (int i = 0; i < o->n_owned; i++) free(o->owned[i]);
    free(o->owned);
    free(o->env);
    free(o);
}
