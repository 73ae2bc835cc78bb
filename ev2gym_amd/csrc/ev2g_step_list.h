// ev2g_step_list.h -- "attached-list" step kernel for the common shape (4 <= P <= 64 ports per env, one transformer,
// single-port chargers): per-step work proportional to the OCCUPIED ports.
//
// Only ~20 % of the ports hold an EV (phi = 0.21 / 0.17), yet a port-per-lane kernel makes every lane of
// every wavefront walk through the charger logic, departures / arrivals, observation columns and reduction staging.
// Here a step is:
//   H  home lanes (one per port, all wavefronts, a dozen instructions): occupancy test on the LDS-resident window;
//      attached or arriving ports append themselves to ONE compact list, idle empty ports write their zero
//      observation columns and mask byte right away;
//   W  worker lanes (one per list entry, i.e. one wavefront per workgroup in the typical case): the whole per-EV
//      pipeline in one lane -- action -> amps (ev_charger.py:137-186), EV.step / _charge / _discharge (ev.py:138-405),
//      profit, departure, arrival, observation columns, mask, potential, SoC log -- results staged by home index;
//   DE one wavefront per workgroup: fixed-order LDS reduction per env (6 quantities x 8 lanes, two chains, 3 xor
//      steps: bit-reproducible), transformer overload, reward, histories, observation head for all envs of the group.
// Two LDS-only barriers per step; the W and DE wavefronts rotate with (workgroup, step) so the heavy phases spread over
// the four SIMDs of a CU.  Arithmetic, operation order and results are those of ev2g_step_v2 / the oracle.
#pragma once
#include "ev2g_step_v2.h"

#define EV2G_LIST_BLOCK 256
#define EV2G_LIST_NQ 6   // staged per-port quantities: power, profit, satisfaction penalty, potential, e_charged, e_discharged
#define EV2G_LIST_GMAX 64

__host__ __device__ inline size_t ev2g_list_lds_bytes(int wb) {
    const size_t NS = (size_t)wb;
    return sizeof(double) * ((EV2G_LIST_NQ + 7) * NS + EV2G_LIST_GMAX * 6) + sizeof(int) * (6 * NS + EV2G_LIST_GMAX + 8);
}

template <int SK, int RK, int WB>
__global__ void __launch_bounds__(WB, 4) ev2g_step_list(const V2P *__restrict__ params, StepIO io, int t0,
                                                                     int k_steps, int auto_reset) {
    extern __shared__ double lds[];
    typedef const V2P __attribute__((address_space(4))) *ParamPtr;
    ParamPtr S = (ParamPtr)(unsigned long long)params;
    constexpr int NS = WB;
    constexpr int NW = WB / 64;   // wavefronts per workgroup
    constexpr int NHEAD = (SK == 1) ? 0 : (SK == 0 ? 60 : 20);   // observation head: 20 prices (+ 40 window columns)
    const int P = S->P, T = S->T, E = S->E, D = S->D;
    const int EPW = 64 / P;   // envs per wavefront (home mapping)
    const int G = NW * EPW;   // envs per workgroup
    int grp;
    {   // XCD-aware mapping: workgroup b runs on XCD b % 8; give each XCD a contiguous range of env groups
        const int nb = gridDim.x, b = blockIdx.x, per = nb >> 3;
        grp = (nb & 7) == 0 ? (b & 7) * per + (b >> 3) : b;
    }
    const int e0 = grp * G;
    const int ne = min(G, E - e0);
    double *stage = lds;                                   // [NQ][NS] per-port step results, by home index
    double *s_cap = stage + (size_t)EV2G_LIST_NQ * NS;
    double *s_tot = s_cap + NS, *s_prev = s_tot + NS, *s_bcap = s_prev + NS, *s_potc = s_bcap + NS;
    double *s_abse = s_potc + NS, *s_act = s_abse + NS;
    double *eacc = s_act + NS;                             // [GMAX][5] episode accumulators
    double *pot_prev = eacc + EV2G_LIST_GMAX * 5;          // [GMAX] charge_power_potential[t]
    int *s_ta = (int *)(pot_prev + EV2G_LIST_GMAX);
    int *s_td = s_ta + NS, *s_ss = s_td + NS, *s_cyc = s_ss + NS, *s_dirty = s_cyc + NS, *items = s_dirty + NS;
    int *emerg_l = items + NS;                             // [GMAX] emergency-capacity crossings of this step
    int *cnt = emerg_l + EV2G_LIST_GMAX;                   // cnt[kk & 1]
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const bool log_soc = S->soc_log != nullptr;
    const double dtd = (double)S->dt, sixty_over_dt = S->sixty_over_dt, dt_over_60 = S->dt_over_60;
    const bool pow2_dt = S->pow2_dt != 0;

    // ---- home lane set-up: port state -> LDS ----
    const int elw = lane / P;
    const int q = lane - elw * P;
    const int e = e0 + wv * EPW + elw;
    const bool valid = (elw < EPW) && (e < E);
    const int g = valid ? e * P + q : e0 * P;   // clamped for idle lanes
    int t = t0;
    if (valid) {
        const int2 w = S->win[g];
        const int2 sc = S->sc[g];
        s_ta[tid] = w.x; s_td[tid] = w.y; s_ss[tid] = sc.x; s_cyc[tid] = sc.y; s_dirty[tid] = 0;
        if (w.x <= t && t <= w.y) {
            s_cap[tid] = S->cap[g]; s_tot[tid] = S->tot_e[g]; s_prev[tid] = S->prev_power[g];
            s_bcap[tid] = S->bcap[g]; s_potc[tid] = S->potc[g];
            s_abse[tid] = log_soc ? S->abs_e[g] : 0.0;
        } else {
            s_cap[tid] = 0.0; s_tot[tid] = 0.0; s_prev[tid] = 0.0; s_bcap[tid] = 1.0; s_potc[tid] = 0.0; s_abse[tid] = 0.0;
        }
    } else {
        s_ta[tid] = EV2G_INT_MAX; s_td[tid] = EV2G_INT_MAX;
    }
    for (int k = 0; k < EV2G_LIST_NQ; k++) stage[k * NS + tid] = 0.0;
    if (tid < EV2G_LIST_GMAX) {
        emerg_l[tid] = 0;
        for (int i = 0; i < 5; i++) eacc[tid * 5 + i] = 0.0;
        pot_prev[tid] = (tid < ne && t < T) ? S->pot_hist[t * E + e0 + tid] : 0.0;
    }
    if (tid < 2) cnt[tid] = 0;
    double a_next = io.actions[g];
    __syncthreads();

    for (int kk = 0; kk < k_steps; kk++) {
        asm volatile("" : "+s"(S));
        int tid_l = tid, g_l = g, lane_l = lane;
        asm volatile("" : "+v"(tid_l), "+v"(g_l), "+v"(lane_l));
        if (t >= T) {  // episode finished inside a fused run: in-kernel ev2g_reset for this workgroup
            if (!auto_reset) break;
            if (valid) {
                const int2 w = S->port_first_win[g_l];
                s_ta[tid_l] = w.x; s_td[tid_l] = w.y; s_ss[tid_l] = S->port_first[g_l]; s_cyc[tid_l] = 0;
                s_cap[tid_l] = 0.0; s_tot[tid_l] = 0.0; s_prev[tid_l] = 0.0; s_abse[tid_l] = 0.0; s_dirty[tid_l] = 3;
                S->port_energy[g_l] = 0.0;
                S->port_current[g_l] = 0.0;
                S->cs_sat_sum[g_l] = 0.0;   // single-port chargers: charger index == port index
                S->cs_served[g_l] = 0;
            }
            if (tid_l < ne) {
                for (int i = 0; i < 8; i++) S->env_acc[(e0 + tid_l) * 8 + i] = 0.0;
                for (int i = 0; i < 5; i++) eacc[tid_l * 5 + i] = 0.0;
                pot_prev[tid_l] = 0.0;
            }
            t = 0;
            lds_barrier();
        }
        double *__restrict__ obs = io.obs ? io.obs + (long long)kk * io.o_stride : nullptr;
        uint8_t *__restrict__ mask = io.mask ? io.mask + (long long)kk * io.m_stride : nullptr;
        const int sstep = t + 1;
        const bool last_step = (kk == k_steps - 1) || (sstep >= T && !auto_reset);
        int *cntk = cnt + (kk & 1);
        const int rot = (blockIdx.x + kk) % NW;    // first worker wavefront of this step; the next one does DE
        const bool de_wave = wv == ((rot + 1) % NW);

        // ---------------- H: home lanes ----------------
        if (valid) {
            const int ta = s_ta[tid_l], td = s_td[tid_l];
            const bool occ = (ta <= t) && (t <= td);
            if (occ) s_act[tid_l] = a_next;
            if (occ || ta == sstep) {
                items[atomicAdd(cntk, 1)] = tid_l;
            } else {   // idle empty port: zero observation columns, mask 0 (state.py:55-57, ev2gym_env.py:452-457)
                if (mask) mask[g_l] = 0;
                if (obs) {
                    const int lq = lane_l - (lane_l / P) * P;
                    double *o = obs + ((g_l / P) * D + ((SK == 1) ? 3 + 3 * lq : (SK == 0 ? 62 + 2 * lq : 22 + 2 * lq)));
                    o[0] = 0.0;
                    o[1] = 0.0;
                    if (SK == 1) o[2] = 0.0;
                }
            }
        }
        // ---- prefetches (unconditional, clamped addresses; collected with one vmcnt(0) before the consumers store) ----
        const bool more = (kk + 1 < k_steps) && (sstep < T || auto_reset);
        a_next = (io.actions + (long long)(more ? kk + 1 : kk) * io.a_stride)[g_l];
        // the DE wavefront prefetches the env-level operands: lane l < ne owns env e0 + l
        double pf_infl = 0.0, pf_solar = 0.0, pf_maxp = 0.0, pf_minp = 0.0, pf_sp = 0.0, pf_sp_next = 0.0;
        double pf_ob[4] = {0.0, 0.0, 0.0, 0.0};
        if (de_wave) {
            const int ec = e0 + min(lane_l, ne - 1);
            pf_infl = S->tr_infl[ec * T + t]; pf_solar = S->tr_solar[ec * T + t];
            pf_maxp = S->tr_maxp[ec * T + t]; pf_minp = S->tr_minp[ec * T + t];
            if (RK == 1) pf_sp = S->setpoint[ec * T + t];
            if (SK == 1) pf_sp_next = S->setpoint[ec * T + min(sstep, T - 1)];
            if (SK != 1 && obs) {   // head columns of all envs of the group: flat index i = env * NHEAD + c, 64 lanes x 4
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = lane_l + 64 * u;
                    const int w = min(i / (NHEAD > 0 ? NHEAD : 1), ne - 1), c = i - (i / (NHEAD > 0 ? NHEAD : 1)) * NHEAD;
                    const double *pprice = (const double *)S->price_ch + (e0 + w) * T;
                    const double *pa = (c < 20) ? pprice + min(sstep + c, T - 1)
                                                : (const double *)S->win_tab + ((long long)(e0 + w) * (T + 1) + sstep) * 40 + (c - 20);
                    pf_ob[u] = *pa;
                }
            }
        }
        lds_barrier();
        if (tid_l == 0) cnt[(kk + 1) & 1] = 0;   // next step's counter (last read two barriers ago)

        // ---------------- W: worker lanes, one per attached / arriving EV ----------------
        {
            const int n = *cntk;
            const int slot = (((wv - rot + NW) % NW) << 6) + lane_l;
            if (slot < n) {
                const int h = items[slot];
                const int hl = h & 63, hw = h >> 6;
                const int helw = hl / P, hq = hl - helw * P;
                const int hel = hw * EPW + helw;       // env inside the workgroup
                const int he = e0 + hel;
                const int hg = he * P + hq;
                int ta = s_ta[h], td = s_td[h];
                int ss = s_ss[h];
                const bool occ = (ta <= t) && (t <= td);
                const double c_imax = S->cs_imax[hq];
                double cap = s_cap[h];
                double power = 0.0, profit = 0.0, satpen = 0.0, pot = 0.0, e_ch = 0.0, e_dis = 0.0;
                bool dirty = false;
                if (occ) {
                    // charger level (ev_charger.py:137-186; one port per charger: a / sum(a))
                    double a = s_act[h];
                    if (a > 1.0) a = a / a;
                    else if (a < -1.0) a = -a / a;
                    const double x = rnd5_x(a);
                    double amps = 0.0;
                    if (x > 0.0) { amps = x * c_imax; if (amps < S->cs_imin[hq] - 0.01) amps = 0.0; }
                    else if (x < 0.0) { const double dmin = S->cs_dmin[hq]; amps = x * S->cs_dmax_abs[hq]; if (amps > dmin - 0.01) amps = dmin; }
                    const SessRec r = *(const SessRec *)(S->rec + ss);
                    const double cap_before = cap;
                    double energy = 0.0, current = 0.0;
                    if (amps != 0.0) {
                        double lutv = 1.0 / 100.0;
                        if (r.lut >= 0) { const int li = ev_lut_index(r.lut, amps); if (li >= 0) lutv = S->lut[li]; }
                        const double prev0 = s_prev[h];
                        const int cyc0 = s_cyc[h];
                        const EvRes o = ev_math(r, lutv, amps, cap, prev0, s_tot[h], cyc0, sixty_over_dt, dt_over_60, dtd, pow2_dt, r.lut >= 0);
                        dirty = (o.cycles != cyc0 || o.energy != 0.0 || o.cap != cap || o.prev_power != prev0);
                        cap = o.cap;
                        s_prev[h] = o.prev_power;
                        s_tot[h] = o.tot_e;
                        s_cyc[h] = o.cycles;
                        energy = o.energy;
                        current = o.current;
                        if (log_soc) s_abse[h] += fabs(o.energy);
                        if (o.emerg) atomicAdd(&emerg_l[hel], 1);
                        power = energy * 60.0 / dtd;
                        const double ae = fabs(energy);
                        // profit by the sign of the ACTION (ev_charger.py:178,194); charge price is negative
                        if (x > 0.0) { e_ch = ae; profit = ae * S->price_ch[he * T + t]; }
                        else { e_dis = ae; profit = ae * S->price_dis[he * T + t]; }
                        if (current - 0.0001 > c_imax) S->env_fault[he] = 1;  // ev_charger.py:203-205
                    }
                    if (last_step) { S->port_energy[hg] = energy; S->port_current[hg] = current; }
                    if (log_soc) S->soc_log[(long long)t * E * P + hg] = (current != 0.0) ? cap_before : -cap_before;
                    if (t >= td) {  // departure (ev_charger.py:209-229, ev.py:191-214)
                        const double score = (cap < r.des - 0.001) ? cap / r.des : 1.0;
                        if (RK != 1) satpen = 100.0 * exp(-10.0 * score);
                        // fire-and-forget device atomics: no returned value => no memory round trip on this path
                        __hip_atomic_fetch_add(&S->cs_served[hg], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_fetch_add(&S->cs_sat_sum[hg], score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        S->sess_final_cap[ss] = cap;
                        if (log_soc) S->sess_abs_e[ss] = s_abse[h];
                        ta = r.nt_arr; td = r.nt_dep;
                        ss = (ta != EV2G_INT_MAX) ? ss + 1 : -1;
                        s_ta[h] = ta; s_td[h] = td; s_ss[h] = ss; s_cyc[h] = 0;
                        s_dirty[h] |= 2;
                    }
                }
                if (ta == sstep) {  // arrival at the end of this step (ev2gym_env.py:399-417, ev.py:115-136)
                    const SessRec &r2 = *(const SessRec *)(S->rec + ss);
                    cap = r2.cap0;
                    const double B = r2.B, v = r2.v;
                    const double evc = r2.pacmax * 1000.0 / v;            // utils.py:773-777
                    const double potc = v * ((evc < c_imax) ? evc : c_imax) / 1000.0;
                    s_tot[h] = 0.0; s_prev[h] = 0.0; s_cyc[h] = 0; s_bcap[h] = B; s_potc[h] = potc; s_abse[h] = 0.0;
                    S->bcap[hg] = B;
                    S->potc[hg] = potc;
                    S->port_energy[hg] = 0.0;
                    S->port_current[hg] = 0.0;
                    dirty = true;
                }
                s_cap[h] = cap;
                if (dirty) s_dirty[h] |= 1;
                const bool occ_after = (ta <= sstep) && (sstep <= td);
                if (mask) mask[hg] = occ_after ? 1 : 0;
                double o0 = 0.0, o1 = 0.0, o2 = 0.0;
                if (occ_after) {
                    const double soc = cap / s_bcap[h];
                    if (SK == 1) { o0 = (soc == 1.0) ? 1.0 : 0.5; o1 = s_tot[h]; o2 = (double)(sstep - ta); }
                    else { o0 = soc; o1 = (double)(td - sstep); }
                    if (soc < 1.0 && td > sstep) pot = s_potc[h];  // utils.py:771
                }
                {   // per-charger clamp (utils.py:779-789)
                    const double mx = S->cs_maxp[hq], mn = S->cs_minp[hq];
                    pot = (pot > mx) ? mx : ((pot < mn) ? 0.0 : pot);
                }
                if (obs) {
                    double *o = obs + (he * D + ((SK == 1) ? 3 + 3 * hq : (SK == 0 ? 62 + 2 * hq : 22 + 2 * hq)));
                    o[0] = o0;
                    o[1] = o1;
                    if (SK == 1) o[2] = o2;
                }
                stage[0 * NS + h] = power;
                stage[1 * NS + h] = profit;
                stage[2 * NS + h] = satpen;
                stage[3 * NS + h] = pot;
                stage[4 * NS + h] = e_ch;
                stage[5 * NS + h] = e_dis;
            }
        }
        lds_barrier();

        // ---------------- DE: one wavefront: per-env reduction, reward, histories, observation head ----------------
        if (de_wave) {
            __builtin_amdgcn_s_waitcnt(0x0F70);   // collect the prefetches before this phase issues stores (vmcnt(0))
            // lane = k*8 + j (k < 6): quantity k, chain j; the sums of env w land in lanes k*8 and are handed to lane w
            double esum[EV2G_LIST_NQ];
#pragma unroll
            for (int kq = 0; kq < EV2G_LIST_NQ; kq++) esum[kq] = 0.0;
            const int k = min(lane_l >> 3, EV2G_LIST_NQ - 1), j = lane_l & 7;
            const bool kvalid = (lane_l >> 3) < EV2G_LIST_NQ;
#pragma unroll 1
            for (int w = 0; w < ne; w++) {
                const int a = (w / EPW) * 64 + (w - (w / EPW) * EPW) * P, b = a + P;
                double acc = 0.0, accb = 0.0;
                if (kvalid) {
                    int i = a + j;
                    for (; i + 8 < b; i += 16) {
                        acc += stage[k * NS + i]; accb += stage[k * NS + i + 8];
                        stage[k * NS + i] = 0.0; stage[k * NS + i + 8] = 0.0;   // leave zeros behind for the next step
                    }
                    if (i < b) { acc += stage[k * NS + i]; stage[k * NS + i] = 0.0; }
                }
                acc += accb;
                acc += __shfl_xor(acc, 1, 64);
                acc += __shfl_xor(acc, 2, 64);
                acc += __shfl_xor(acc, 4, 64);
#pragma unroll
                for (int kq = 0; kq < EV2G_LIST_NQ; kq++) {
                    const double v = __shfl(acc, kq * 8, 64);
                    if (lane_l == w) esum[kq] = v;
                }
            }
            if (lane_l < ne) {
                const int pe = e0 + lane_l;
                const double usage = esum[0];
                double ptr = pf_infl + pf_solar;   // Transformer.reset + step + get_how_overloaded (transformer.py:258-302)
                ptr += usage;
                const double over = (ptr > pf_maxp + 0.0001 || ptr < pf_minp - 0.0001) ? fabs(ptr - pf_maxp) : 0.0;
                S->over_hist[t * E + pe] = over;
                if (last_step) S->tr_power_now[pe] = ptr;
                S->usage_hist[t * E + pe] = usage;
                const double potn = esum[3];
                if (sstep < T) S->pot_hist[sstep * E + pe] = potn;
                const double costs = esum[1];
                double reward;
                if (RK == 1) {  // SquaredTrackingErrorReward reward.py:7-14
                    const double pp = pot_prev[lane_l];
                    const double m = (pp < pf_sp) ? pp : pf_sp;
                    const double d = m - usage;
                    reward = -(d * d);
                } else if (RK == 2) {  // profit_maximization reward.py:78-87
                    reward = costs - esum[2];
                } else {  // ProfitMax_TrPenalty_UserIncentives reward.py:34-44
                    reward = costs - 100.0 * over - esum[2];
                }
                pot_prev[lane_l] = potn;
                double *acc = eacc + lane_l * 5;
                acc[0] += reward; acc[1] += costs; acc[2] += esum[4]; acc[3] += esum[5]; acc[4] += (double)emerg_l[lane_l];
                emerg_l[lane_l] = 0;
                if (io.reward) io.reward[(long long)kk * io.r_stride + pe] = reward;
                if (io.done) io.done[(long long)kk * io.d_stride + pe] = (sstep >= T) ? 1 : 0;
                if (sstep >= T || last_step) {  // flush the episode accumulators (get_statistics reads them)
                    auto ga = S->env_acc + pe * 8;
                    for (int i = 0; i < 5; i++) { ga[i] += acc[i]; acc[i] = 0.0; }
                }
                if (obs) {
                    double *o = obs + pe * D;
                    if (SK == 1) { o[0] = (double)sstep / (double)T; o[1] = (sstep < T) ? pf_sp_next : 0.0; o[2] = usage; }
                    else { o[0] = (double)sstep; o[1] = usage; }
                }
            }
            if (SK != 1 && obs) {   // observation head columns (state.py:76-83, :119-134)
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = lane_l + 64 * u;
                    const int w = i / NHEAD, c = i - w * NHEAD;
                    if (w < ne) obs[(e0 + w) * D + 2 + c] = (c < 20) ? ((sstep + c < T) ? fabs(pf_ob[u]) : 0.0) : pf_ob[u];
                }
                for (int i = lane_l + 256; i < ne * NHEAD; i += 64) {   // groups with more than 256 head columns (small P)
                    const int w = i / NHEAD, c = i - w * NHEAD;
                    const int pe = e0 + w;
                    double v;
                    if (c < 20) { const int kx = sstep + c; v = (kx < T) ? fabs(S->price_ch[pe * T + kx]) : 0.0; }
                    else v = S->win_tab[((long long)pe * (T + 1) + sstep) * 40 + (c - 20)];
                    obs[pe * D + 2 + c] = v;
                }
            }
        }
        t += 1;
        // Next step: H only appends to items / cnt[(kk+1)&1] and reads the state that W finished writing before the
        // second barrier; stage is written again only by W after the next first barrier, which this DE wavefront has
        // to reach as well.  So no third barrier.
    }
    __syncthreads();
    if (valid) {
        const int d = s_dirty[tid];
        if (d & 2) S->win[g] = make_int2(s_ta[tid], s_td[tid]);
        if (d) S->sc[g] = make_int2(s_ss[tid], s_cyc[tid]);
        if (d & 1) { S->cap[g] = s_cap[tid]; S->tot_e[g] = s_tot[tid]; S->prev_power[g] = s_prev[tid]; if (log_soc) S->abs_e[g] = s_abse[tid]; }
    }
}
