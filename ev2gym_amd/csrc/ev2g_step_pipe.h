// ev2g_step_pipe.h -- software-pipelined variant of the fast-path step kernel (same shape as ev2g_step_wave:
// P <= 64 ports per env, one transformer, single-port chargers).
//
// ev2g_step_wave runs the phases of a step strictly one after the other and the battery maths (phase B) of a
// workgroup on its wavefronts 0 and 1, while the others wait at the barrier; a step is one long dependent chain
//     A | barrier | B | barrier | C  D  E                          (10.8 k cycles per workgroup-step at cfg2).
// Here the battery maths has its OWN wavefronts -- two WORKER wavefronts per workgroup, one for the charging items,
// one for the discharging ones -- next to EV2G_PIPE_ENVW env wavefronts that own whole envs as before.  Nothing in the
// env-level phase E(t) (transformer, reward, histories, observation head) depends on the next step's battery maths
// B(t+1), and B(t+1) only needs the charger-level phase A(t+1), which needs only the occupancy windows and the (pre-
// fetched) action.  So inside an episode a step becomes
//     env wavefronts:     C(t)  D(t)  A(t+1) | X |  E(t), prefetch(t+1)  | Y |
//     worker wavefronts:        (idle)       | X |        B(t+1)         | Y |
// i.e. the chain per step is C + D + A + max(E, B) instead of A + B + C + D + E.  The first step of a launch / of an
// episode has nothing to overlap with and runs A | X | B | Y unpipelined.
//
// Hazards (all LDS; global memory is never used to communicate inside a launch):
//   * stage rows 0,4..7 / s_amps of a port: D(t) reads, then the SAME lane zeroes / rewrites them in A(t+1) (program
//     order); the workers write them in B(t+1), after X.
//   * the env sums go from D(t) to E(t) through `esums` (not through the stage rows, which A(t+1) recycles).
//   * items / cnt: written by A(t+1) before X, read by B(t+1) after X; cnt is double-buffered by step parity and the
//     workers clear the other parity during B.
//   * per-port state (s_cap, s_tot, s_prev, s_cyc, s_abse, s_dirty): B(t+1) updates it between X and Y, E(t) never
//     touches it, C(t+1) reads it after Y.
// Arithmetic, reduction tree and every global store are those of ev2g_step_wave: results are bit-identical.
#pragma once
#include "ev2g_step_wave.h"

#ifndef EV2G_PIPE_ENVW
#define EV2G_PIPE_ENVW 8                                   // env wavefronts per workgroup
#endif
#define EV2G_PIPE_HOME (EV2G_PIPE_ENVW * 64)               // home slots (LDS array length)
#define EV2G_PIPE_BLOCK (EV2G_PIPE_HOME + 128)             // + charge worker + discharge worker

__host__ __device__ inline size_t ev2g_pipe_lds_bytes(int envs_per_group) {
    const size_t NS = EV2G_PIPE_HOME;
    return sizeof(double) * (EV2G_NQ * (NS + 8) + 7 * NS + 14 * (size_t)envs_per_group + 4 * 64) + sizeof(int) * (6 * NS + 8);
}

template <int SK, int RK>
__global__ void __launch_bounds__(EV2G_PIPE_BLOCK, 5) ev2g_step_pipe(const V2P *__restrict__ params, StepIO io, int t0,
                                                                     int k_steps, int auto_reset, WaveArgs wa) {
    extern __shared__ double lds[];
    typedef const V2P __attribute__((address_space(4))) *ParamPtr;
    ParamPtr S = (ParamPtr)(unsigned long long)params;
    constexpr int NS = EV2G_PIPE_HOME;
    constexpr int RS = NS + 8;
    const int P = wa.P, T = wa.T, E = wa.E, D = wa.D;
    const gptr slabP = (gptr)wa.slab_port, slabH = (gptr)wa.slab_hist, slabS = (gptr)S->slab_sess;
    const unsigned long long PS8 = wa.slab_port_slice, HS8 = wa.hist_slice, SS8 = S->sess_slice;
    const gptr env_acc = (gptr)wa.env_acc;
    const int EPW = 64 / P;                       // envs per env wavefront
    const int G = EV2G_PIPE_ENVW * EPW;           // envs per workgroup
    int grp;
    {   // XCD-aware mapping: workgroup b runs on XCD b % 8; give each XCD a contiguous range of env groups
        const int nb = gridDim.x, b = blockIdx.x, per = nb >> 3;
        grp = (nb & 7) == 0 ? (b & 7) * per + (b >> 3) : b;
    }
    const int e0 = grp * G;
    double *stage = lds;                                   // [NQ][RS] per-port step results, by home index
    double *s_cap = stage + (size_t)EV2G_NQ * RS;
    double *s_tot = s_cap + NS, *s_prev = s_tot + NS, *s_bcap = s_prev + NS, *s_potc = s_bcap + NS;
    double *s_amps = s_potc + NS, *s_abse = s_amps + NS;
    double *eacc = s_abse + NS;                            // [G][6] episode accumulators + charge_power_potential[t]
    double *esums = eacc + 6 * G;                          // [G][8] the env sums of the current step, D -> E
    double *s_cst = esums + 8 * G;                         // [4][64] per-charger gates and clamps
    int *s_ta = (int *)(s_cst + 4 * 64);
    int *s_td = s_ta + NS, *s_ss = s_td + NS, *s_cyc = s_ss + NS, *s_dirty = s_cyc + NS, *items = s_dirty + NS;
    int *cnt = items + NS;  // cnt[2*(t&1) + {0 charge, 1 discharge}]
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const bool worker = wv >= EV2G_PIPE_ENVW;              // wavefront role (uniform per wavefront)
    const int wk = wv - EV2G_PIPE_ENVW;                    // 0: charging items, 1: discharging items
    const bool log_soc = S->soc_log != nullptr;
    const double dtd = (double)S->dt, sixty_over_dt = S->sixty_over_dt, dt_over_60 = S->dt_over_60;
    const bool pow2_dt = S->pow2_dt != 0;

    // ---- home lane set-up (env wavefronts) ----
    const int elw = lane / P;            // env inside the wavefront
    const int q = lane - elw * P;        // port slot
    const int e = e0 + wv * EPW + elw;
    const bool valid = !worker && (elw < EPW) && (e < E);
    const int g = valid ? e * P + q : 0;
    const int ocol = (SK == 1) ? 3 + 3 * q : (SK == 0 ? 62 + 2 * q : 22 + 2 * q);
    const int cs = valid ? q : 0;
    const int hid = worker ? 0 : tid;    // home index (LDS slot); workers never use theirs
    int t = t0;
    const bool head = valid && q == 0;
    const int elg = worker ? 0 : wv * EPW + elw;      // env inside the workgroup
    double c_imax, c_dmaxabs, a_next;
    {   // launch prologue (see ev2g_step_wave): one round trip for windows / constants / first action / accumulators,
        // a second one for the per-EV state where an EV is attached
        const unsigned g8 = (unsigned)g * 8u, c8 = (unsigned)cs * 8u, cp8 = (unsigned)min(tid, P - 1) * 8u;
        const unsigned ec = (unsigned)(valid ? e : e0);
        i2v w = ldg32<i2v>(PA(EV2G_PS_WIN), g8), sc = ldg32<i2v>(PA(EV2G_PS_SC), g8);
        int lut0 = ldg32<int>(PA(EV2G_PS_LUT), g8 >> 1);
        c_imax = ldg32<double>(wa.cs_imax, c8); c_dmaxabs = ldg32<double>(wa.cs_dmax_abs, c8);
        double k_imin = ldg32<double>(wa.cs_imin, cp8), k_dmin = ldg32<double>(wa.cs_dmin, cp8);
        double k_maxp = ldg32<double>(wa.cs_maxp, cp8), k_minp = ldg32<double>(wa.cs_minp, cp8);
        a_next = ldg32<double>(io.actions, (unsigned)(valid ? g : e0 * P) * 8u);
        double l_pot = ldg32<double>(slabH + HS8, ((unsigned)min(t, T - 1) * (unsigned)E + ec) * 8u);
        d2v acc01 = ldg32<d2v>(env_acc, ec * 64u), acc23 = ldg32<d2v>(env_acc, ec * 64u + 16u);
        double acc4 = ldg32<double>(env_acc, ec * 64u + 32u);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(w), "+v"(sc), "+v"(lut0), "+v"(c_imax), "+v"(c_dmaxabs), "+v"(k_imin), "+v"(k_dmin),
                     "+v"(k_maxp), "+v"(k_minp), "+v"(a_next), "+v"(l_pot), "+v"(acc01), "+v"(acc23), "+v"(acc4));
        if (tid < P) {
            s_cst[0 * 64 + tid] = k_imin - 0.01; s_cst[1 * 64 + tid] = k_dmin;
            s_cst[2 * 64 + tid] = k_maxp; s_cst[3 * 64 + tid] = k_minp;
        }
        if (valid) {
            s_ta[hid] = w.x; s_td[hid] = w.y; s_ss[hid] = sc.x; s_cyc[hid] = sc.y;
            s_dirty[hid] = (lut0 + 1) << 8;   // bits 0,1: write-back flags; bits 8..: 1 + efficiency-table id
            if (w.x <= t && t <= w.y) {
                s_cap[hid] = ldg32<double>(PA(EV2G_PS_CAP), g8); s_tot[hid] = ldg32<double>(PA(EV2G_PS_TOT), g8);
                s_prev[hid] = ldg32<double>(PA(EV2G_PS_PREV), g8);
                s_bcap[hid] = ldg32<double>(PA(EV2G_PS_BCAP), g8); s_potc[hid] = ldg32<double>(PA(EV2G_PS_POTC), g8);
                s_abse[hid] = log_soc ? ldg32<double>(PA(EV2G_PS_ABSE), g8) : 0.0;
            } else {
                s_cap[hid] = 0.0; s_tot[hid] = 0.0; s_prev[hid] = 0.0; s_bcap[hid] = 1.0; s_potc[hid] = 0.0; s_abse[hid] = 0.0;
            }
        }
        if (head) {
            double *ea = eacc + elg * 6;
            ea[0] = acc01.x; ea[1] = acc01.y; ea[2] = acc23.x; ea[3] = acc23.y; ea[4] = acc4;
            ea[5] = (t < T) ? l_pot : 0.0;
        }
    }
    if (tid < 4) cnt[tid] = 0;
    if (!worker) for (int k = 0; k < EV2G_NQ; k++) stage[k * RS + hid] = 0.0;
    __syncthreads();

        // ---- phase A of step ta_step (charger level, ev_charger.py:137-186), env wavefronts ----
#define EV2G_PIPE_PHASE_A(ta_step)                                                                                       \
        {                                                                                                                \
            occ = false; cap_before = 0.0;                                                                               \
            if (valid) {                                                                                                 \
                const int ta = s_ta[tid_l], td = s_td[tid_l];                                                            \
                occ = (ta <= (ta_step)) && ((ta_step) <= td);                                                            \
                if (log_soc && occ) cap_before = s_cap[tid_l];                                                           \
                double a = occ ? a_next : 0.0;                                                                           \
                if (a > 1.0) a = 1.0;                                                                                    \
                else if (a < -1.0) a = -1.0;                                                                             \
                double amps = 0.0;                                                                                       \
                if (occ) {                                                                                               \
                    const double x = rnd5_x(a);                                                                          \
                    if (x > 0.0) { amps = x * c_imax; if (amps < s_cst[0 * 64 + q_l]) amps = 0.0; }                     \
                    else if (x < 0.0) { const double c_dmin = s_cst[1 * 64 + q_l]; amps = x * c_dmaxabs; if (amps > c_dmin - 0.01) amps = c_dmin; } \
                }                                                                                                        \
                s_amps[tid_l] = amps;                                                                                    \
                stage[0 * RS + tid_l] = 0.0; stage[4 * RS + tid_l] = 0.0; stage[5 * RS + tid_l] = 0.0;                   \
                stage[6 * RS + tid_l] = 0.0; stage[7 * RS + tid_l] = 0.0;                                                \
                int *cntk = cnt + 2 * ((ta_step) & 1);                                                                   \
                if (amps != 0.0) items[(amps > 0.0) ? atomicAdd(&cntk[0], 1) : NS - 1 - atomicAdd(&cntk[1], 1)] = tid_l; \
            }                                                                                                            \
        }

        // ---- what crosses a step boundary in registers: the prices C(tp) needs and the action of launch step ka ----
#define EV2G_PIPE_PREFETCH(tp, ka)                                                                                       \
        {                                                                                                                \
            a_next = ldg32<double>(io.actions + (long long)(ka) * io.a_stride, (unsigned)gc * 8u);                       \
            const d2v st0 = ldg32<d2v>(S->step_tab, (unsigned)(ec * T + (tp)) * 64u);                                    \
            pf_pch = st0.x; pf_pdis = st0.y;                                                                             \
        }

        // ---- phase B of step tb (battery maths on the compact list), one worker wavefront per item kind ----
#define EV2G_PIPE_PHASE_B(tb)                                                                                            \
        {                                                                                                                \
            int *cntk = cnt + 2 * ((tb) & 1);                                                                            \
            const int nitems = cntk[wk];                                                                                 \
            if (lane < 2) cnt[2 * (((tb) + 1) & 1) + lane] = 0;   /* the other parity: last read one step ago */         \
            for (int i = lane; i < nitems; i += 64) {                                                                    \
                const int h = (wk == 0) ? items[i] : items[NS - 1 - i];                                                  \
                const double amps_h = s_amps[h];                                                                         \
                const int lut_id = (s_dirty[h] >> 8) - 1;                                                                \
                const int li = (lut_id >= 0) ? ev_lut_index(lut_id, amps_h) : -1;                                        \
                double lut_raw = ldg32<double>(S->lut, (unsigned)max(li, 0) * 8u);                                       \
                const SessRec r = ldg32_rec(S->rec, (unsigned)s_ss[h] * (unsigned)sizeof(SessRec));                      \
                asm volatile("" : "+v"(lut_raw));                                                                        \
                const double cap0 = s_cap[h], prev0 = s_prev[h];                                                         \
                const int cyc0 = s_cyc[h];                                                                               \
                const double lutv = (li >= 0) ? lut_raw : 1.0 / 100.0;                                                   \
                const EvRes o = ev_math(r, lutv, amps_h, cap0, prev0, s_tot[h], cyc0, sixty_over_dt, dt_over_60, dtd, pow2_dt, lut_id >= 0); \
                if (o.cycles != cyc0 || o.energy != 0.0 || o.cap != cap0 || o.prev_power != prev0) s_dirty[h] |= 1;     \
                s_cap[h] = o.cap;                                                                                        \
                s_prev[h] = o.prev_power;                                                                                \
                s_tot[h] = o.tot_e;                                                                                      \
                s_cyc[h] = o.cycles;                                                                                     \
                s_amps[h] = o.energy;                                                                                    \
                if (log_soc) s_abse[h] += fabs(o.energy);                                                                \
                stage[0 * RS + h] = o.energy * 60.0 / dtd;                                                               \
                stage[(wk == 0 ? 4 : 5) * RS + h] = fabs(o.energy);                                                      \
                stage[6 * RS + h] = (double)o.emerg;                                                                     \
                stage[7 * RS + h] = o.current;                                                                           \
            }                                                                                                            \
        }

    // registers that live across phases of the env role
    bool occ = false;
    double cap_before = 0.0;
    double pf_pch = 0.0, pf_pdis = 0.0;
    constexpr int NHEAD = (SK == 1) ? 0 : (SK == 0 ? 60 : 20);
    bool have_a = false;   // phase A of step t already ran (pipelined behind the previous step)

    // The two roles run SEPARATE loops with the same (uniform) control flow and therefore the same barrier sequence:
    // kept in one loop, the env role's loop-carried registers (prefetches, action, charger constants) stay live
    // through the workers' battery maths and the kernel no longer fits the 96 VGPRs that 5 wavefronts per SIMD allow.
    if (worker) {
        for (int kk = 0; kk < k_steps; kk++) {
            if (t >= T) {
                if (!auto_reset) break;
                t = 0;
                have_a = false;
                lds_barrier();
            }
            const int sstep = t + 1;
            const bool more = (kk + 1 < k_steps) && (sstep < T);
            if (!have_a) {
                lds_barrier();
                EV2G_PIPE_PHASE_B(t)
                lds_barrier();
            }
            if (more) {
                lds_barrier();   // X
                EV2G_PIPE_PHASE_B(sstep)
                lds_barrier();   // Y
            }
            have_a = more;
            t += 1;
        }
    } else
    for (int kk = 0; kk < k_steps; kk++) {
        asm volatile("" : "+s"(S));
        int tid_l = hid, g_l = g, e_l = e, q_l = q, lane_l = lane;
        asm volatile("" : "+v"(tid_l), "+v"(g_l), "+v"(e_l), "+v"(q_l), "+v"(lane_l));
        const unsigned g8 = (unsigned)g_l * 8u;
        if (t >= T) {  // episode finished inside a fused run: in-kernel ev2g_reset for this workgroup
            if (!auto_reset) break;
            if (valid) {
                const i2v w = ldg32<i2v>(S->port_first_win, g8);
                s_ta[tid_l] = w.x; s_td[tid_l] = w.y; s_ss[tid_l] = ldg32<int>(S->port_first, g8 >> 1); s_cyc[tid_l] = 0;
                s_cap[tid_l] = 0.0; s_tot[tid_l] = 0.0; s_prev[tid_l] = 0.0; s_abse[tid_l] = 0.0; s_dirty[tid_l] = 3;
                stg32<double>(PA(EV2G_PS_PENERGY), g8, 0.0);
                stg32<double>(PA(EV2G_PS_PCURRENT), g8, 0.0);
                stg32<double>(PA(EV2G_PS_SATSUM), g8, 0.0);
                stg32<int>(PA(EV2G_PS_SERVED), g8 >> 1, 0);
            }
            if (head) {
                for (int i = 0; i < 8; i++) stg32<double>(env_acc, (unsigned)e_l * 64u + (unsigned)i * 8u, 0.0);
                for (int i = 0; i < 6; i++) eacc[elg * 6 + i] = 0.0;
            }
            if (tid < 4) cnt[tid] = 0;   // both parities: the new episode starts at step 0 whatever T was
            t = 0;
            have_a = false;
            lds_barrier();               // (uniform) the counters are clear before any lane of A(0) counts into them
        }
        double *obs = io.obs ? io.obs + (long long)kk * io.o_stride : nullptr;
        uint8_t *mask = io.mask ? io.mask + (long long)kk * io.m_stride : nullptr;
        const int sstep = t + 1;
        const bool last_step = (kk == k_steps - 1) || (sstep >= T && !auto_reset);
        const bool more = (kk + 1 < k_steps) && (sstep < T);   // the next step belongs to the same episode: pipeline it
        const int ec = valid ? e_l : e0;
        const int gc = valid ? g_l : e0 * P;

        if (!have_a) {   // first step of a launch / of an episode: nothing to overlap with
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a_next));
            EV2G_PIPE_PHASE_A(t)
            EV2G_PIPE_PREFETCH(t, more ? kk + 1 : kk)
            lds_barrier();
            lds_barrier();   // (the workers run B(t) between these two)
        }

        double usage = 0.0;
        {
            // ---------------- C: home lanes ----------------
            // pipelined steps collected their two prefetches inside E(t-1); only the unpipelined first step still has
            // them in flight (issued before its two barriers)
            if (!have_a) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a_next), "+v"(pf_pch), "+v"(pf_pdis));
            if (valid) {
                double profit = 0.0, satpen = 0.0, pot = 0.0;
                int ta = s_ta[tid_l], td = s_td[tid_l];
                double cap = s_cap[tid_l];
                if (occ) {
                    const double energy = s_amps[tid_l];
                    const double current = stage[7 * RS + tid_l];
                    if (energy != 0.0) {  // profit by the sign of the ACTION (ev_charger.py:178,194), staged under 4 / 5
                        const double ech = stage[4 * RS + tid_l];
                        profit = (ech != 0.0) ? ech * pf_pch : stage[5 * RS + tid_l] * pf_pdis;
                    }
                    if (current - 0.0001 > c_imax) stg32<int>(S->env_fault, (unsigned)e_l * 4u, 1);  // ev_charger.py:203-205
                    if (last_step) { stg32<double>(PA(EV2G_PS_PENERGY), g8, energy); stg32<double>(PA(EV2G_PS_PCURRENT), g8, current); }
                    if (log_soc) stg32<double>(S->soc_log + (long long)t * E * P, g8, (current != 0.0) ? cap_before : -cap_before);
                    if (t >= td) {  // departure (ev_charger.py:209-229, ev.py:191-214)
                        const int ss = s_ss[tid_l];
                        const unsigned r8 = (unsigned)ss * (unsigned)sizeof(SessRec);
                        const double des = ldg32<double>(S->rec, r8 + (unsigned)offsetof(SessRec, des));
                        const double score = (cap < des - 0.001) ? cap / des : 1.0;
                        if (RK != 1) satpen = 100.0 * exp(-10.0 * score);
                        __hip_atomic_fetch_add((int __attribute__((address_space(1))) *)(PA(EV2G_PS_SERVED) + (g8 >> 1)), 1,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_fetch_add((double __attribute__((address_space(1))) *)(PA(EV2G_PS_SATSUM) + g8), score,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        stg32<double>(slabS, (unsigned)ss * 8u, cap);
                        if (log_soc) stg32<double>((slabS + SS8), (unsigned)ss * 8u, s_abse[tid_l]);
                        const i2v nx = ldg32<i2v>(S->rec, r8 + (unsigned)offsetof(SessRec, nt_arr));
                        ta = nx.x; td = nx.y;
                        s_ta[tid_l] = ta; s_td[tid_l] = td;
                        s_ss[tid_l] = (ta != EV2G_INT_MAX) ? ss + 1 : -1;
                        s_cyc[tid_l] = 0;
                        s_dirty[tid_l] |= 2;
                    }
                }
                if (ta == sstep) {  // arrival at the end of this step (ev2gym_env.py:399-417, ev.py:115-136)
                    const unsigned r8 = (unsigned)s_ss[tid_l] * (unsigned)sizeof(SessRec);
                    cap = ldg32<double>(S->rec, r8 + (unsigned)offsetof(SessRec, cap0));
                    const double B = ldg32<double>(S->rec, r8 + (unsigned)offsetof(SessRec, B));
                    const double v = ldg32<double>(S->rec, r8 + (unsigned)offsetof(SessRec, v));
                    const double evc = ldg32<double>(S->rec, r8 + (unsigned)offsetof(SessRec, pacmax)) * 1000.0 / v;            // utils.py:773-777
                    const double potc = v * ((evc < c_imax) ? evc : c_imax) / 1000.0;
                    s_cap[tid_l] = cap; s_tot[tid_l] = 0.0; s_prev[tid_l] = 0.0; s_cyc[tid_l] = 0; s_bcap[tid_l] = B; s_potc[tid_l] = potc;
                    s_abse[tid_l] = 0.0;
                    const int lut_new = ldg32<int>(S->rec, r8 + (unsigned)offsetof(SessRec, lut));
                    stg32<int>(PA(EV2G_PS_LUT), g8 >> 1, lut_new);
                    stg32<double>(PA(EV2G_PS_BCAP), g8, B);
                    stg32<double>(PA(EV2G_PS_POTC), g8, potc);
                    stg32<double>(PA(EV2G_PS_PENERGY), g8, 0.0);
                    stg32<double>(PA(EV2G_PS_PCURRENT), g8, 0.0);
                    s_dirty[tid_l] = (s_dirty[tid_l] & 3) | 1 | ((lut_new + 1) << 8);
                }
                const bool occ_after = (ta <= sstep) && (sstep <= td);
                if (mask) stg32<uint8_t>(mask, (unsigned)g_l, occ_after ? 1 : 0);
                double o0 = 0.0, o1 = 0.0, o2 = 0.0;
                if (occ_after) {
                    const double soc = cap / s_bcap[tid_l];
                    if (SK == 1) { o0 = (soc == 1.0) ? 1.0 : 0.5; o1 = s_tot[tid_l]; o2 = (double)(sstep - ta); }
                    else { o0 = soc; o1 = (double)(td - sstep); }
                    if (soc < 1.0 && td > sstep) pot = s_potc[tid_l];  // utils.py:771
                }
                {   // per-charger clamp (utils.py:779-789)
                    const double c_maxp = s_cst[2 * 64 + q_l], c_minp = s_cst[3 * 64 + q_l];
                    pot = (pot > c_maxp) ? c_maxp : ((pot < c_minp) ? 0.0 : pot);
                }
                if (obs) {
                    const unsigned o8 = (unsigned)(e_l * D + ocol) * 8u;
                    stg32<double>(obs, o8, o0);
                    stg32<double>(obs, o8 + 8u, o1);
                    if (SK == 1) stg32<double>(obs, o8 + 16u, o2);
                }
                stage[1 * RS + tid_l] = profit;
                stage[2 * RS + tid_l] = satpen;
                stage[3 * RS + tid_l] = pot;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

            // ---------------- D: per-env reduction inside the wavefront (fixed tree, as in ev2g_step_wave) ----------------
            {
                const int k = lane_l >> 3, j = lane_l & 7;
                const int wbase = (tid_l & ~63);
                const double *row = stage + k * RS;
#pragma unroll 1
                for (int w = 0; w < EPW; w++) {
                    const int a = wbase + w * P, b = a + P;
                    double xa[4], xb[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int i = a + j + 16 * u;
                        const double ra = row[min(i, NS - 1)], rb = row[min(i + 8, NS - 1)];
                        xa[u] = (i < b) ? ra : 0.0;
                        xb[u] = (i + 8 < b) ? rb : 0.0;
                    }
                    double acc = 0.0, accb = 0.0;
#pragma unroll
                    for (int u = 0; u < 4; u++) { acc += xa[u]; accb += xb[u]; }
                    acc += accb;
                    acc += xor1_f64(acc);
                    acc += xor2_f64(acc);
                    acc += xor4_f64(acc);
                    if (j == 0) esums[((tid_l >> 6) * EPW + w) * 8 + k] = acc;
                }
            }
            // ---------------- A(t+1): the next step's charger level, pipelined behind this step ----------------
            if (more) EV2G_PIPE_PHASE_A(sstep)
        }

        if (more) lds_barrier();   // X: the work list of step t+1 is complete (the workers start B(t+1))
        {
            // ---------------- E: per env (head lane) + observation head (the env's lanes) ----------------
            // E fetches its own tables (they are not carried across the step in registers): issued first, together with
            // the next step's prices / action, they complete while the LDS reads below are in flight; everything here
            // runs in the shadow of the workers' B(t+1).  The one vmcnt(0) also retires phase C's stores, issued long ago.
            double pf_base = 0.0, pf_maxp = 0.0, pf_minp = 0.0, pf_sp = 0.0, pf_ob0 = 0.0, pf_ob1 = 0.0, pf_ob2 = 0.0;
            {
                const unsigned eT64 = (unsigned)(ec * T) * 64u, et64 = eT64 + (unsigned)t * 64u;
                if (RK == 0) {
                    const d2v st1 = ldg32<d2v>(S->step_tab, et64 + 16u);
                    pf_base = st1.x; pf_maxp = st1.y; pf_minp = ldg32<double>(S->step_tab, et64 + 32u);
                }
                if (RK == 1) pf_sp = ldg32<double>(S->step_tab, et64 + 40u);
                if (SK == 1) {
                    pf_ob0 = ldg32<double>(S->step_tab, eT64 + (unsigned)min(sstep, T - 1) * 64u + 40u);
                } else {
                    const unsigned h8 = (unsigned)((ec * (T + 1) + sstep) * NHEAD) * 8u;
                    pf_ob0 = ldg32<double>(S->head_tab, h8 + (unsigned)min(q_l, NHEAD - 1) * 8u);
                    pf_ob1 = ldg32<double>(S->head_tab, h8 + (unsigned)min(q_l + P, NHEAD - 1) * 8u);
                    pf_ob2 = ldg32<double>(S->head_tab, h8 + (unsigned)min(q_l + 2 * P, NHEAD - 1) * 8u);
                }
                if (more) EV2G_PIPE_PREFETCH(sstep, (kk + 2 < k_steps) ? kk + 2 : kk + 1)
                else if (kk + 1 < k_steps) a_next = ldg32<double>(io.actions + (long long)(kk + 1) * io.a_stride, (unsigned)gc * 8u);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            double esum[EV2G_NQ];
#pragma unroll
            for (int kq = 0; kq < EV2G_NQ; kq++) esum[kq] = esums[elg * 8 + kq];   // meaningful in valid lanes
            usage = esum[0];
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a_next), "+v"(pf_pch), "+v"(pf_pdis), "+v"(pf_base), "+v"(pf_maxp), "+v"(pf_minp),
                         "+v"(pf_sp), "+v"(pf_ob0), "+v"(pf_ob1), "+v"(pf_ob2));
            if (head) {
                double *ea = eacc + elg * 6;
                const double ea0 = ea[0], ea1 = ea[1], ea2 = ea[2], ea3 = ea[3], ea4 = ea[4], ea5 = ea[5];
                const unsigned e8 = (unsigned)e_l * 8u;
                double over100 = 0.0;
                if (RK == 0) {  // Transformer.reset + step + get_how_overloaded (transformer.py:258-302)
                    double ptr = pf_base;
                    ptr += usage;
                    const double over = (ptr > pf_maxp + 0.0001 || ptr < pf_minp - 0.0001) ? fabs(ptr - pf_maxp) : 0.0;
                    stg32<double>((slabH + 2 * HS8) + (long long)t * E * 8, e8, over);
                    if (last_step) stg32<double>(S->tr_power_now, e8, ptr);
                    over100 = 100.0 * over;
                } else {
                    const unsigned erT64 = (unsigned)(e_l * T + t) * 64u;
                    double ptr = ldg32<double>(S->step_tab, erT64 + 16u);
                    ptr += usage;
                    const double mx = ldg32<double>(S->step_tab, erT64 + 24u), mn = ldg32<double>(S->step_tab, erT64 + 32u);
                    const double over = (ptr > mx + 0.0001 || ptr < mn - 0.0001) ? fabs(ptr - mx) : 0.0;
                    stg32<double>((slabH + 2 * HS8) + (long long)t * E * 8, e8, over);
                    if (last_step) stg32<double>(S->tr_power_now, e8, ptr);
                }
                stg32<double>(slabH + (long long)t * E * 8, e8, usage);
                const double potn = esum[3];
                if (sstep < T) stg32<double>((slabH + HS8) + (long long)sstep * E * 8, e8, potn);
                const double costs = esum[1];
                double reward;
                if (RK == 1) {  // SquaredTrackingErrorReward reward.py:7-14
                    const double pp = ea5;
                    const double m = (pp < pf_sp) ? pp : pf_sp;
                    const double d = m - usage;
                    reward = -(d * d);
                } else if (RK == 2) {  // profit_maximization reward.py:78-87
                    reward = costs - esum[2];
                } else {  // ProfitMax_TrPenalty_UserIncentives reward.py:34-44
                    reward = costs - over100 - esum[2];
                }
                const double n0 = ea0 + reward, n1 = ea1 + costs, n2 = ea2 + esum[4], n3 = ea3 + esum[5], n4 = ea4 + esum[6];
                ea[0] = n0; ea[1] = n1; ea[2] = n2; ea[3] = n3; ea[4] = n4; ea[5] = potn;
                if (io.reward) stg32<double>(io.reward + (long long)kk * io.r_stride, e8, reward);
                if (io.done) stg32<uint8_t>(io.done + (long long)kk * io.d_stride, (unsigned)e_l, (sstep >= T) ? 1 : 0);
                if (sstep >= T || last_step) {  // publish the running episode totals (get_statistics reads them)
                    const unsigned a8 = (unsigned)e_l * 64u;
                    stg32<d2v>(env_acc, a8, (d2v){n0, n1});
                    stg32<d2v>(env_acc, a8 + 16u, (d2v){n2, n3});
                    stg32<double>(env_acc, a8 + 32u, n4);
                }
            }
            if (valid && obs) {
                const unsigned o8 = (unsigned)(e_l * D) * 8u;
                if (SK == 1) {  // PublicPST state.py:6-35
                    if (q_l == 0) {
                        stg32<double>(obs, o8, (double)sstep / (double)T);
                        stg32<double>(obs, o8 + 8u, (sstep < T) ? pf_ob0 : 0.0);
                        stg32<double>(obs, o8 + 16u, usage);
                    }
                } else {  // V2G_profit_max(_loads) state.py:65-83, :108-135: columns 2.. are a copy of the head table row
                    if (q_l == 0) { stg32<double>(obs, o8, (double)sstep); stg32<double>(obs, o8 + 8u, usage); }
                    int c = q_l;
                    if (c < NHEAD) stg32<double>(obs, o8 + (unsigned)(2 + c) * 8u, pf_ob0);
                    c = q_l + P;
                    if (c < NHEAD) stg32<double>(obs, o8 + (unsigned)(2 + c) * 8u, pf_ob1);
                    c = q_l + 2 * P;
                    if (c < NHEAD) stg32<double>(obs, o8 + (unsigned)(2 + c) * 8u, pf_ob2);
                    const unsigned h8 = (unsigned)((e_l * (T + 1) + sstep) * NHEAD) * 8u;
                    for (c = q_l + 3 * P; c < NHEAD; c += P)    // tiny envs (P < 20): the remaining columns, unprefetched
                        stg32<double>(obs, o8 + (unsigned)(2 + c) * 8u, ldg32<double>(S->head_tab, h8 + (unsigned)c * 8u));
                }
            }
        }
        if (more) lds_barrier();   // Y: the battery maths of step t+1 is done
        have_a = more;
        t += 1;
    }
#undef EV2G_PIPE_PHASE_A
#undef EV2G_PIPE_PHASE_B
#undef EV2G_PIPE_PREFETCH
    __syncthreads();
    if (valid) {
        const int d = s_dirty[hid];
        const unsigned g8 = (unsigned)g * 8u;
        if (d & 2) stg32<i2v>(PA(EV2G_PS_WIN), g8, (i2v){s_ta[hid], s_td[hid]});
        if (d & 3) stg32<i2v>(PA(EV2G_PS_SC), g8, (i2v){s_ss[hid], s_cyc[hid]});
        if (d & 1) {
            stg32<double>(PA(EV2G_PS_CAP), g8, s_cap[hid]); stg32<double>(PA(EV2G_PS_TOT), g8, s_tot[hid]); stg32<double>(PA(EV2G_PS_PREV), g8, s_prev[hid]);
            if (log_soc) stg32<double>(PA(EV2G_PS_ABSE), g8, s_abse[hid]);
        }
    }
}
