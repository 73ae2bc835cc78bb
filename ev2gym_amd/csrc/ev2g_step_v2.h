// ev2g_step_v2.h -- the production step kernel (P <= BLOCK ports per env; the generic kernel in
// ev2g_device.h covers larger envs).
//
// Per workgroup: G = BLOCK / P whole envs, one HOME lane per port that keeps the port's dynamic state
// (window, session, capacity, energy counters) in REGISTERS across the fused steps of one launch.
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

struct EvRes {
    double cap, prev_power, tot_e, energy, current;
    int cycles, emerg;
};

// EV.step + _charge/_discharge (ev.py:138-186, :240-355, :357-405) from one session record.
// Same operation order as the reference (and as oracle/ev2g_oracle.c); -ffp-contract=off.
__device__ __forceinline__ EvRes ev_math(const SessRec &r, const double *__restrict__ lut, double amps, double cap,
                                         double prev_power, double tot_e, int cycles, double sixty_over_dt,
                                         double dt_over_60, double dt) {
    EvRes o;
    o.cap = cap; o.prev_power = prev_power; o.tot_e = tot_e; o.energy = 0.0; o.current = 0.0; o.cycles = cycles; o.emerg = 0;
    if (amps > 0.0 && amps < r.gate_ch) amps = 0.0;
    else if (amps < 0.0 && amps > r.gate_dis) amps = 0.0;
    if (amps == 0.0) return o;  // ev.py:158-163: no ceil, previous_power untouched
    if (prev_power == 0.0 || (prev_power / amps) < 0.0) o.cycles = cycles + 1;
    const double B = r.B, v = r.v;
    if (amps > 0.0) {
        const double eta = (r.lut >= 0) ? lut_get(lut, r.lut, rint(amps)) / 100.0 : r.eta_ch;
        double pilot_dsoc = eta * amps * v / 1000.0 / B / sixty_over_dt;
        const double max_dsoc = eta * r.pacmax / B / sixty_over_dt;
        if (pilot_dsoc > max_dsoc) pilot_dsoc = max_dsoc;
        const double soc = cap / B;
        double curr_soc;
        if (r.ts == 1.0) {
            curr_soc = pilot_dsoc + soc;
            if (curr_soc > 1.0) curr_soc = 1.0;
        } else {
            const double pts = r.ts + (pilot_dsoc - max_dsoc) / max_dsoc * (r.ts - 1.0);
            double new_soc;
            if (soc < pts) {
                if (1.0 <= (pts - soc) / pilot_dsoc) new_soc = pilot_dsoc + soc;
                else new_soc = 1.0 + exp(r.tsm * (pilot_dsoc + soc - pts) / (pts - 1.0)) * (pts - 1.0);
            } else {
                new_soc = 1.0 + exp(r.tsm * pilot_dsoc / (pts - 1.0)) * (soc - 1.0);
            }
            const double lim = (max_dsoc > pilot_dsoc) ? pilot_dsoc : max_dsoc;
            curr_soc = (new_soc - soc > lim) ? (lim + soc) : new_soc;
        }
        const double dsoc = curr_soc - soc;
        o.cap = curr_soc * B;
        o.energy = dsoc * B;
        o.current = o.energy / dt_over_60 * 1000.0 / v;
    } else {
        double given_power = amps * v / 1000.0;
        if (fabs(given_power) > fabs(r.pdismax)) given_power = r.pdismax;
        const double eta = (r.lut >= 0) ? lut_get(lut, r.lut, fabs(rint(amps))) / 100.0 : r.eta_dis;
        double given_energy = given_power * eta * dt / 60.0;
        if (cap + given_energy < r.minB) {
            if (cap > r.minB) { o.energy = -(cap - r.minB); given_energy = o.energy; }
            else { o.energy = 0.0; given_energy = 0.0; }
            o.cap = r.minB;
        } else {
            o.energy = given_energy;
            o.cap = cap + given_energy;
        }
        if (cap > r.emerg && o.cap < r.emerg) o.emerg = 1;
        o.current = given_energy * 60.0 / dt * 1000.0 / v;
    }
    o.prev_power = o.energy;
    o.tot_e = tot_e + o.energy;
    o.cap = ceil2(o.cap);
    return o;
}

// LDS carve-up for ev2g_step_v2 (doubles first, then ints); NS = G*P, NT = G*R
__host__ __device__ inline size_t ev2g_v2_lds_bytes(int NS, int NT, int G) {
    return sizeof(double) * ((size_t)EV2G_NQ * NS + 4 * (size_t)NS + (size_t)EV2G_NQ * NT + (size_t)EV2G_NQ * G) +
           sizeof(int) * (4 * (size_t)NS + 4);
}

template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) ev2g_step_v2(const DevScn *__restrict__ Sp, const DevState *__restrict__ STp,
                                                      StepIO io, int t0, int k_steps, int auto_reset) {
    extern __shared__ double lds[];
    const DevScn &S = *Sp;
    const DevState &st = *STp;
    const int P = S.P, R = S.R, T = S.T, C = S.C, npc = S.npc, E = S.E, D = S.D, G = S.G;
    int grp;
    {   // XCD-aware mapping: workgroup b runs on XCD b % 8; give each XCD a contiguous range of env groups
        const int nb = gridDim.x, b = blockIdx.x, per = nb >> 3;
        grp = (nb & 7) == 0 ? (b & 7) * per + (b >> 3) : b;
    }
    const int e0 = grp * G;
    const int ne = min(G, E - e0);
    const int N = ne * P, NS = G * P, NT = G * R;
    double *stage = lds;                          // [NQ][NS] per-port results, by home index
    double *hs_amps = stage + (size_t)EV2G_NQ * NS;  // work item inputs / outputs, by home index
    double *hs_cap = hs_amps + NS, *hs_prev = hs_cap + NS, *hs_tot = hs_prev + NS;
    double *tsum = hs_tot + NS;                   // [NQ][NT]
    double *esum = tsum + (size_t)EV2G_NQ * NT;   // [NQ][G]
    int *hs_ss = (int *)(esum + (size_t)EV2G_NQ * G);
    int *hs_cyc = hs_ss + NS, *items = hs_cyc + NS, *occf = items + NS, *cnt = occf + NS;  // cnt[0] charge, cnt[1] discharge
    const int tid = threadIdx.x;
    const bool log_cs = st.cs_profits != nullptr;
    const double dtd = (double)S.dt, sixty_over_dt = S.sixty_over_dt, dt_over_60 = S.dt_over_60;

    // ---- home lane set-up (once per launch) ----
    const bool valid = tid < N;
    const int el = valid ? tid / P : 0;
    const int q = valid ? tid - el * P : 0;
    const int e = e0 + el;
    const long long g = (long long)e * P + q;
    const int cs = S.slot_cs[q], pref = S.slot_port[q], ocol = S.slot_obs[q];
    const double imax = S.cs_imax[cs], imin = S.cs_imin[cs], dmin = S.cs_dmin[cs], dmaxabs = S.cs_dmax_abs[cs];
    const double cs_maxp = S.cs_maxp[cs], cs_minp = S.cs_minp[cs];
    int2 w = make_int2(EV2G_INT_MAX, EV2G_INT_MAX);
    int ss = -1, cycles = 0;
    double cap = 0.0, tot_e = 0.0, prev_power = 0.0, Bcap = 1.0, pot_c = 0.0, last_e = 0.0, last_i = 0.0;
    bool dirty_state = false, dirty_win = false, dirty_last = false;
    int t = t0;
    if (valid) {
        w = st.win[g];
        const int2 sc = st.sc[g];
        ss = sc.x;
        cycles = sc.y;
        if (w.x <= t && t <= w.y) {
            cap = st.cap[g];
            tot_e = st.tot_e[g];
            prev_power = st.prev_power[g];
            const SessRec &r = S.rec[ss];
            Bcap = r.B;
            const double evc = r.pacmax * 1000.0 / r.v;            // utils.py:773-777
            pot_c = r.v * ((evc < imax) ? evc : imax) / 1000.0;
        }
    }
    if (tid < 2) cnt[tid] = 0;
    if (npc > 1 && valid) occf[tid] = (w.x <= t && t <= w.y) ? 1 : 0;  // sibling ports read occupancy from LDS
    __syncthreads();

    for (int kk = 0; kk < k_steps; kk++) {
        if (t >= T) {  // episode finished inside a fused run: in-kernel ev2g_reset for this workgroup
            if (!auto_reset) break;
            if (valid) {
                w = S.port_first_win[g];
                ss = S.port_first[g];
                cycles = 0; cap = 0.0; tot_e = 0.0; prev_power = 0.0; last_e = 0.0; last_i = 0.0;
                dirty_state = dirty_win = dirty_last = true;
                if (npc > 1) occf[tid] = 0;
            }
            for (int i = tid; i < ne * C; i += BLOCK) {
                const long long gc = (long long)e0 * C + i;
                st.cs_sat_sum[gc] = 0.0;
                st.cs_served[gc] = 0;
                if (log_cs) { st.cs_profits[gc] = 0.0; st.cs_e_ch[gc] = 0.0; st.cs_e_dis[gc] = 0.0; }
            }
            for (int i = tid; i < ne * 8; i += BLOCK) st.env_acc[(long long)e0 * 8 + i] = 0.0;
            for (int i = tid; i < ne; i += BLOCK) st.pot_hist[e0 + i] = 0.0;
            t = 0;
            __syncthreads();
        }
        const double *__restrict__ actions = io.actions + (long long)kk * io.a_stride;
        double *__restrict__ obs = io.obs ? io.obs + (long long)kk * io.o_stride : nullptr;
        uint8_t *__restrict__ mask = io.mask ? io.mask + (long long)kk * io.m_stride : nullptr;
        const int sstep = t + 1;

        // ---------------- A: home lanes, charger level (ev_charger.py:137-186) ----------------
        const bool occ = valid && (w.x <= t) && (t <= w.y);
        double amps = 0.0, x = 0.0;
        if (valid) {
            double a = occ ? actions[(long long)e * P + pref] : 0.0;
            if (npc == 1) {
                if (a > 1.0) a = a / a;
                else if (a < -1.0) a = -a / a;
            } else {
                const int j0 = q - (pref - cs * npc);
                double Ssum = 0.0;
                for (int j = 0; j < npc; j++) {  // sequential python sum() over the charger's ports
                    const bool oj = occf[tid - q + j0 + j] != 0;
                    Ssum = Ssum + (oj ? actions[(long long)e * P + cs * npc + j] : 0.0);
                }
                if (Ssum > 1.0) a = a / Ssum;
                else if (Ssum < -1.0) a = -a / Ssum;
            }
            if (occ) {
                x = rnd5(a);
                if (x > 0.0) { amps = x * imax; if (amps < imin - 0.01) amps = 0.0; }
                else if (x < 0.0) { amps = x * dmaxabs; if (amps > dmin - 0.01) amps = dmin; }
            }
            stage[0 * NS + tid] = 0.0;
            stage[4 * NS + tid] = 0.0;
            stage[5 * NS + tid] = 0.0;
            stage[6 * NS + tid] = 0.0;
            stage[7 * NS + tid] = 0.0;
        }
        const bool active = occ && amps != 0.0;
        if (active) {
            const int pos = (amps > 0.0) ? atomicAdd(&cnt[0], 1) : NS - 1 - atomicAdd(&cnt[1], 1);
            items[pos] = tid;
            hs_amps[tid] = amps;
            hs_cap[tid] = cap;
            hs_prev[tid] = prev_power;
            hs_tot[tid] = tot_e;
            hs_ss[tid] = ss;
            hs_cyc[tid] = cycles;
        }
        __syncthreads();

        // ---------------- B: worker lanes, battery maths on the compact list ----------------
        {
            const int nch = cnt[0], ndis = cnt[1];
            const int nchp = (nch + 63) & ~63;  // discharge items start on a wavefront boundary
            for (int i = tid; i < nchp + ndis; i += BLOCK) {
                int h = -1;
                if (i < nch) h = items[i];
                else if (i >= nchp) h = items[NS - 1 - (i - nchp)];
                if (h >= 0) {
                    const SessRec r = S.rec[hs_ss[h]];
                    const EvRes o = ev_math(r, S.lut, hs_amps[h], hs_cap[h], hs_prev[h], hs_tot[h], hs_cyc[h],
                                            sixty_over_dt, dt_over_60, dtd);
                    hs_cap[h] = o.cap;
                    hs_prev[h] = o.prev_power;
                    hs_tot[h] = o.tot_e;
                    hs_cyc[h] = o.cycles;
                    hs_amps[h] = o.energy;  // slot reused: EV.current_energy
                    const double ae = fabs(o.energy);
                    stage[0 * NS + h] = o.energy * 60.0 / dtd;
                    stage[(i < nch ? 4 : 5) * NS + h] = ae;
                    stage[6 * NS + h] = (double)o.emerg;
                    stage[7 * NS + h] = o.current;
                }
            }
        }
        __syncthreads();
        if (tid < 2) cnt[tid] = 0;

        // ---------------- C: home lanes: read back, departures, arrivals, observation columns ----------------
        if (valid) {
            double profit = 0.0, satpen = 0.0, pot = 0.0;
            if (occ) {
                double energy = 0.0, current = 0.0;
                if (active) {
                    const double ncap = hs_cap[tid];
                    energy = hs_amps[tid];
                    current = stage[7 * NS + tid];
                    const int ncyc = hs_cyc[tid];
                    const double nprev = hs_prev[tid];
                    if (ncyc != cycles || energy != 0.0 || ncap != cap || nprev != prev_power) dirty_state = true;
                    cap = ncap;
                    prev_power = nprev;
                    tot_e = hs_tot[tid];
                    cycles = ncyc;
                    const double ae = fabs(energy);
                    // charge price is negative: profit += |E| * price (ev_charger.py:178,194)
                    profit = ae * ((x > 0.0) ? S.price_ch[(long long)e * T + t] : S.price_dis[(long long)e * T + t]);
                    if (npc == 1 && current - 0.0001 > imax) st.env_fault[e] = 1;  // ev_charger.py:203-205
                }
                dirty_last = true;  // the stored last-step values are not loaded at launch: rewrite whenever occupied
                last_e = energy;
                last_i = current;
                if (t >= w.y) {  // departure (ev_charger.py:209-229, ev.py:191-214)
                    const SessRec &r = S.rec[ss];
                    const double des = r.des;
                    const double score = (cap < des - 0.001) ? cap / des : 1.0;
                    if (S.reward_kind != 1) satpen = 100.0 * exp(-10.0 * score);
                    const long long gc = (long long)e * C + cs;
                    if (npc == 1) { st.cs_served[gc] += 1; st.cs_sat_sum[gc] += score; }
                    else { atomicAdd(&st.cs_served[gc], 1); atomicAdd(&st.cs_sat_sum[gc], score); }
                    st.sess_final_cap[ss] = cap;
                    w = make_int2(r.nt_arr, r.nt_dep);
                    ss = (w.x != EV2G_INT_MAX) ? ss + 1 : -1;
                    cycles = 0;
                    dirty_win = true;
                }
            }
            if (w.x == sstep) {  // arrival at the end of this step (ev2gym_env.py:399-417, ev.py:115-136)
                const SessRec &r = S.rec[ss];
                cap = r.cap0;
                tot_e = 0.0;
                prev_power = 0.0;
                cycles = 0;
                Bcap = r.B;
                const double evc = r.pacmax * 1000.0 / r.v;
                pot_c = r.v * ((evc < imax) ? evc : imax) / 1000.0;
                last_e = 0.0;
                last_i = 0.0;
                dirty_state = true;
                dirty_last = true;
            }
            const bool occ_after = (w.x <= sstep) && (sstep <= w.y);
            if (npc > 1) occf[tid] = occ_after ? 1 : 0;
            if (mask) mask[(long long)e * P + pref] = occ_after ? 1 : 0;
            double o0 = 0.0, o1 = 0.0, o2 = 0.0;
            if (occ_after) {
                const double soc = cap / Bcap;
                if (S.state_kind == 1) { o0 = (soc == 1.0) ? 1.0 : 0.5; o1 = tot_e; o2 = (double)(sstep - w.x); }
                else { o0 = soc; o1 = (double)(w.y - sstep); }
                if (soc < 1.0 && w.y > sstep) pot = pot_c;  // utils.py:771
            }
            if (npc == 1) pot = (pot > cs_maxp) ? cs_maxp : ((pot < cs_minp) ? 0.0 : pot);  // utils.py:779-789
            if (obs) {
                double *o = obs + (long long)e * D + ocol;
                o[0] = o0;
                o[1] = o1;
                if (S.state_kind == 1) o[2] = o2;
            }
            stage[1 * NS + tid] = profit;
            stage[2 * NS + tid] = satpen;
            stage[3 * NS + tid] = pot;
        }
        __syncthreads();

        // ---------------- C2: per charger (multi-port chargers, or charger history) ----------------
        if (npc > 1 || log_cs) {
            if (valid && pref == cs * npc) {  // leader = port 0 of the charger
                double pw = 0.0, cur = 0.0, pr = 0.0, ec = 0.0, ed = 0.0, pp = 0.0;
                bool fault = false;
                for (int j = 0; j < npc; j++) {  // sequential, port order (ev_charger.py:155-205)
                    pw += stage[0 * NS + tid + j];
                    cur += stage[7 * NS + tid + j];
                    pr += stage[1 * NS + tid + j];
                    ec += stage[4 * NS + tid + j];
                    ed += stage[5 * NS + tid + j];
                    pp += stage[3 * NS + tid + j];
                    if (cur - 0.0001 > imax) fault = true;
                }
                if (fault) st.env_fault[e] = 1;
                if (npc > 1) {
                    pp = (pp > cs_maxp) ? cs_maxp : ((pp < cs_minp) ? 0.0 : pp);
                    stage[3 * NS + tid] = pp;
                    for (int j = 1; j < npc; j++) stage[3 * NS + tid + j] = 0.0;
                }
                if (log_cs) {
                    const long long gc = (long long)e * C + cs;
                    st.cs_profits[gc] += pr;
                    st.cs_e_ch[gc] += ec;
                    st.cs_e_dis[gc] += ed;
                    st.cs_power_now[gc] = pw;
                    st.cs_cur_now[gc] = cur;
                    st.cs_power_hist[((long long)t * E + e) * C + cs] = pw;
                    st.cs_cur_hist[((long long)t * E + e) * C + cs] = cur;
                }
            }
            __syncthreads();
        }

        // ---------------- D: LDS-staged segmented reduction, one wavefront per (env, transformer) ----------------
        {
            const int wv = tid >> 6, lane = tid & 63, nw = BLOCK >> 6;
            const int k = lane >> 3, j = lane & 7;
            const int ntask = ne * R;
            for (int task = wv; task < ntask; task += nw) {
                const int tel = task / R, r = task - tel * R;
                const int a = tel * P + S.tr_seg[r], b = tel * P + S.tr_seg[r + 1];
                double acc = 0.0;
                for (int i = a + j; i < b; i += 8) acc += stage[k * NS + i];
                acc += __shfl_xor(acc, 1, 64);
                acc += __shfl_xor(acc, 2, 64);
                acc += __shfl_xor(acc, 4, 64);
                if (j == 0) tsum[k * NT + task] = acc;
            }
        }
        __syncthreads();
        if (R > 1) {
            for (int i = tid; i < ne * EV2G_NQ; i += BLOCK) {
                const int tel = i / EV2G_NQ, k = i - tel * EV2G_NQ;
                double v = 0.0;
                for (int r = 0; r < R; r++) v += tsum[k * NT + tel * R + r];
                esum[k * G + tel] = v;
            }
            __syncthreads();
        }
        const double *es = (R > 1) ? esum : tsum;
        const int esn = (R > 1) ? G : NT;

        // ---------------- E: per env ----------------
        {
            const int lpe = BLOCK / ne;
            const int pel = tid / lpe, l = tid - pel * lpe;
            if (pel < ne) {
                const int pe = e0 + pel;
                const double usage = es[0 * esn + pel];
                if (l == 0) {
                    double over_sum = 0.0;
                    for (int r = 0; r < R; r++) {  // Transformer.reset + step + get_how_overloaded (transformer.py:258-302)
                        const long long erT = ((long long)pe * R + r) * T + t;
                        double ptr = S.tr_infl[erT] + S.tr_solar[erT];
                        ptr += tsum[0 * NT + pel * R + r];
                        const double mx = S.tr_maxp[erT], mn = S.tr_minp[erT];
                        const double over = (ptr > mx + 0.0001 || ptr < mn - 0.0001) ? fabs(ptr - mx) : 0.0;
                        st.over_hist[((long long)t * E + pe) * R + r] = over;
                        st.tr_power_now[(long long)pe * R + r] = ptr;
                        over_sum += 100.0 * over;
                    }
                    st.usage_hist[(long long)t * E + pe] = usage;
                    if (sstep < T) st.pot_hist[(long long)sstep * E + pe] = es[3 * esn + pel];
                    const double costs = es[1 * esn + pel];
                    double reward;
                    if (S.reward_kind == 1) {  // SquaredTrackingErrorReward reward.py:7-14
                        const double sp = S.setpoint[(long long)pe * T + t];
                        const double pp = st.pot_hist[(long long)t * E + pe];
                        const double m = (pp < sp) ? pp : sp;
                        const double d = m - usage;
                        reward = -(d * d);
                    } else if (S.reward_kind == 2) {  // profit_maximization reward.py:78-87
                        reward = costs - es[2 * esn + pel];
                    } else {  // ProfitMax_TrPenalty_UserIncentives reward.py:34-44
                        reward = costs - over_sum - es[2 * esn + pel];
                    }
                    double *acc = st.env_acc + (long long)pe * 8;
                    acc[0] += reward;
                    acc[1] += costs;
                    acc[2] += es[4 * esn + pel];
                    acc[3] += es[5 * esn + pel];
                    acc[4] += es[6 * esn + pel];
                    if (io.reward) io.reward[(long long)kk * io.r_stride + pe] = reward;
                    if (io.done) io.done[(long long)kk * io.d_stride + pe] = (sstep >= T) ? 1 : 0;
                }
                if (obs) write_obs_env(S, obs + (long long)pe * D, pe, sstep, usage, l, lpe);
            }
        }
        t += 1;
        // no barrier needed here: the next step's phase A only touches stage[0,4..7], hs_*, items and cnt,
        // none of which phase E reads; tsum/esum are rewritten only after three more barriers.
    }

    // ---- write the register-resident port state back ----
    if (valid) {
        if (dirty_win) { st.win[g] = w; }
        if (dirty_win || dirty_state) st.sc[g] = make_int2(ss, cycles);
        if (dirty_state) { st.cap[g] = cap; st.tot_e[g] = tot_e; st.prev_power[g] = prev_power; }
        if (dirty_last) { st.port_energy[g] = last_e; st.port_current[g] = last_i; }
    }
}
