// ev2g_step_pipe.h -- software-pipelined form of the fast-path step kernel for PERSISTENT launches (k >= 2 fused steps):
// the same shape as ev2g_step_wave (P <= 64 ports per env, one transformer, single-port chargers), the same arithmetic, the same
// reduction tree, the same stores -- bit-identical results -- in a different order in time.
//
// ev2g_step_wave runs a step as one dependent chain   A | barrier | B | barrier | C  D  E   and the launch time of a persistent
// launch IS that chain (the chip holds exactly one round of workgroups at 4096 envs): 14.6 k cycles for a step with battery-maths
// items.  Of that chain only the STATE side feeds the next step: the battery maths B(t+1) needs the attached EV's capacity after
// B(t), the departures / arrivals of step t (action-independent, known from the occupancy windows) and the next action.  Everything
// else a step produces -- profit, satisfaction scores, observation columns, mask, the per-env reduction, transformer overload,
// reward, histories, the observation head: the OUTPUT side -- hangs off the state chain and nothing in step t+1 waits for it.
// So here the battery maths has its own wavefronts (EV2G_PIPE_NW workers next to EV2G_PIPE_ENVW env wavefronts per workgroup)
// and the env wavefronts emit the outputs of step t WHILE the workers integrate step t+1:
//
//     env wavefronts:    grab B(t), Cs(t), A(t+1) | X |  Co(t)  D(t)  E(t)      | Y | grab B(t+1), Cs(t+1), A(t+2) | X | ...
//     worker wavefronts:         (wait)           | X |        B(t+1)           | Y |           (wait)              | X | ...
//
//   grab   the env lane copies what B(t) left for its port (energy, current, capacity, emergency flag) into registers
//   Cs     state side of phase C: departures and arrivals update the port's window / session / capacity in LDS
//   A      charger level of the next step: action -> amps, work-list items
//   Co D E output side of phase C, the per-env reduction, the env level -- from registers and wave-private LDS rows only
//
// The chain per step is  Cs + A + max(B, Co + D + E)  instead of  A + B + C + D + E.  Round 1 had a first version of this idea
// (ev2g_step_pipe, removed in round 2): it overlapped only E, and at 118 VGPRs one 10-wavefront workgroup per CU was all that
// fitted.  This one needs 5 wavefronts per SIMD (two 10-wavefront workgroups per CU: 16 envs per CU, like ev2g_step_wave), i.e.
// at most 96 VGPRs: the env path and the worker path are separate loops (the allocator sees their maxima, not their sum), the
// occupancy window lives in registers, the session-record tail is fetched field by field.
//
// Hazards (all LDS; global memory is never used to communicate inside a launch):
//   * per-port state s_cap / s_tot / s_prev / s_cyc / s_abse / s_dirty / s_amps / s_cur: written by the workers between X and Y,
//     by the env lanes (grab, Cs, A) between Y and X -- never both in one interval.  Co / D / E (between X and Y) touch only
//     registers, the stage rows, eacc and s_cst.
//   * stage rows: written (Co) and reduced (D) by the same wavefront, ordered by s_waitcnt.
//   * items / cnt: written by A before X, read by B after X; cnt is double-buffered by step parity, the workers clear the other
//     parity during B.
// Global memory: every prefetch of an iteration is issued between X and Y and collected by ONE s_waitcnt vmcnt(0) at the top of the
// next iteration, after Y -- a battery-maths phase later; the stores of Co / D / E were issued in the same interval and have long
// drained by then (loads and stores share vmcnt on gfx9-family ISAs and retire out of order with respect to each other).
//
// Scope: persistent launches without in-launch auto-reset and without charger histories (EV2G_FLAG_LOG_CS_HISTORY), float64
// actions.  Everything else (single-step launches, float32 actions, in-launch resets) runs ev2g_step_wave; the host routes
// (launch_steps, ev2g_host.hip) and reports the kernel of a launch shape through ev2g_launch_kernel_name.
#pragma once
#include "ev2g_step_wave.h"

// Workgroup shape: wavefronts of a workgroup go to the CU's four SIMDs round-robin, and five wavefronts per SIMD is all the
// register file holds (96 VGPRs each).  Measured: two 10-wavefront workgroups (8 + 2) do NOT become co-resident on a CU although
// the occupancy calculator says they fit (3 + 3 + 2 + 2 wavefronts per SIMD twice over) -- the launch ran in two rounds; 4 + 1
// keeps every workgroup at one or two wavefronts per SIMD, four of them per CU (16 envs per CU, like ev2g_step_wave).
#ifndef EV2G_PIPE_ENVW
#define EV2G_PIPE_ENVW 4                                   // env wavefronts per workgroup
#endif
#ifndef EV2G_PIPE_NW
#define EV2G_PIPE_NW 1                                     // worker wavefronts per workgroup
#endif
#define EV2G_PIPE_HOME (EV2G_PIPE_ENVW * 64)               // home slots (LDS array length)
#define EV2G_PIPE_BLOCK (EV2G_PIPE_HOME + EV2G_PIPE_NW * 64)

#ifdef EV2G_PHASE_TIMING   /* tools/pipe_timing.py: cycles per segment as wavefront 0 (env) and the first worker see them */
#define PP_DECL unsigned long long pp_last = __builtin_readcyclecounter(); unsigned long long pp_acc[12] = {0,0,0,0,0,0,0,0,0,0,0,0};
#define PP_MARK(i) { unsigned long long n_ = __builtin_readcyclecounter(); pp_acc[i] += n_ - pp_last; pp_last = n_; }
#define PP_FLUSH(base, n) if ((threadIdx.x & 63) == 0 && S->dbg) { for (int i_ = 0; i_ < (n); i_++) S->dbg[(size_t)blockIdx.x * 18 + (base) + i_] += pp_acc[i_]; }
#else
#define PP_DECL
#define PP_MARK(i)
#define PP_FLUSH(base, n)
#endif

__host__ __device__ inline size_t ev2g_pipe_lds_bytes(int envs_per_group) {
    const size_t NS = EV2G_PIPE_HOME;
    return sizeof(double) * (EV2G_NQ * (NS + 8) + 6 * NS + 7 * (size_t)envs_per_group + 4 * 64) + sizeof(int) * (4 * NS + 8);
}

template <int SK, int RK>
__global__ void __launch_bounds__(EV2G_PIPE_BLOCK, 5) ev2g_step_pipe(const V2P *__restrict__ params, StepIO io, int t0, int k_steps, WaveArgs wa) {
    extern __shared__ double lds[];
    typedef const V2P __attribute__((address_space(4))) *ParamPtr;
    ParamPtr S = (ParamPtr)(unsigned long long)params;
    constexpr int NS = EV2G_PIPE_HOME;
    constexpr int RS = NS + 8;
    const int P = wa.P, T = wa.T, E = wa.E, D = wa.D, M = wa.M;
    const int off = io.scn_off;
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
    double *stage = lds;                                   // [NQ][RS] per-port step results by home index: env wavefronts only
    double *s_cap = stage + (size_t)EV2G_NQ * RS;
    double *s_tot = s_cap + NS, *s_prev = s_tot + NS;
    double *s_amps = s_prev + NS, *s_abse = s_amps + NS, *s_cur = s_abse + NS;
    double *eacc = s_cur + NS;                             // [G][7] episode accumulators + charge_power_potential[t], [t-1], per env
    double *s_cst = eacc + 7 * G;                          // [4][64] per-charger gates and clamps
    int *s_ss = (int *)(s_cst + 4 * 64);
    int *s_cyc = s_ss + NS, *s_dirty = s_cyc + NS, *items = s_dirty + NS;
    int *cnt = items + NS;  // cnt[2*(kk&1) + {0 charge, 1 discharge}]
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const bool worker = wv >= EV2G_PIPE_ENVW;              // wavefront role (uniform per wavefront)
    const bool log_soc = io.log_soc != 0;
    k_steps = min(k_steps, T - t0);                        // (the host never asks for more: no in-launch reset here)

    if (tid < 4) cnt[tid] = 0;   // (cnt[0], cnt[2]: the item counters of even / odd steps)
    if (!worker) for (int k = 0; k < EV2G_NQ; k++) stage[k * RS + tid] = 0.0;

    if (worker) {
        // ======================================= worker wavefronts: battery maths only =======================================
        const int wtid = tid - NS;
        const double dtd = (double)S->dt, sixty_over_dt = S->sixty_over_dt, dt_over_60 = S->dt_over_60;
        const bool pow2_dt = S->pow2_dt != 0;
        __syncthreads();
        __builtin_amdgcn_s_setprio(3);   // the workers are the critical path of the workgroup whenever they have items
        PP_DECL
        for (int kk = 0; kk < k_steps; kk++) {
            asm volatile("" : "+s"(S));
            int wtid_l = wtid;
            asm volatile("" : "+v"(wtid_l));
            lds_barrier();   // X: the items of step kk are posted
            PP_MARK(0)
            const int *cntk = cnt + 2 * (kk & 1);
            if (wtid_l == 0) cnt[2 * ((kk + 1) & 1)] = 0;   // next step's counter (last used before the previous Y)
            const int n_items = cntk[0];
            // ONE list, charging and discharging items mixed (a workgroup of four envs posts ~40 items a step: one wavefront pass;
            // the two kinds diverge inside ev_math, whose memory round trips -- the expensive part -- they share); worker w takes
            // items w, w + NW, ...
            for (int i = (wtid_l & 63) * EV2G_PIPE_NW + (wtid_l >> 6); i < n_items; i += EV2G_PIPE_NW * 64) {
                const int h = items[i];
                {
                    const double amps_h = s_amps[h];
                    const int dirty0 = s_dirty[h];
                    const int lut_id = (dirty0 >> 8) - 1;
                    const int li = (lut_id >= 0) ? ev_lut_index(lut_id, amps_h) : -1;
                    double lut_raw = ldg32<double>(S->lut, (unsigned)max(li, 0) * 8u);
                    const SessRec r = ldg32_rec(S->rec, (unsigned)s_ss[h] * (unsigned)sizeof(SessRec));
                    asm volatile("" : "+v"(lut_raw));
                    const double cap0 = s_cap[h], prev0 = s_prev[h];
                    const int cyc0 = s_cyc[h];
                    const double lutv = (li >= 0) ? lut_raw : 1.0 / 100.0;
                    const EvRes o = ev_math(r, lutv, amps_h, cap0, prev0, s_tot[h], cyc0, sixty_over_dt, dt_over_60, dtd, pow2_dt, lut_id >= 0);
                    const bool changed = (o.cycles != cyc0 || o.energy != 0.0 || o.cap != cap0 || o.prev_power != prev0);
                    // bit 0: state to write back; bit 2: this step's emergency-SoC crossing (ev.py:401-402), picked up by the env lane
                    s_dirty[h] = (dirty0 & ~4) | (changed ? 1 : 0) | (o.emerg ? 4 : 0);
                    s_cap[h] = o.cap;
                    s_prev[h] = o.prev_power;
                    s_tot[h] = o.tot_e;
                    s_cyc[h] = o.cycles;
                    s_amps[h] = o.energy;
                    s_cur[h] = o.current;
                    if (log_soc) s_abse[h] += fabs(o.energy);
                }
            }
            PP_MARK(1)
            lds_barrier();   // Y: the results of step kk are in LDS
            PP_MARK(2)
        }
        if (wtid == 0) { PP_FLUSH(9, 3) }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        return;
    }

    // ============================================ env wavefronts ================================================================
    const int elw = lane / P;            // env inside the wavefront
    const int q = lane - elw * P;        // port slot (== reference port: one transformer, single-port chargers)
    const int e = e0 + wv * EPW + elw;
    const bool valid = (elw < EPW) && (e < E);
    const int g = valid ? e * P + q : 0;
    const int ocol = (SK == 1) ? 3 + 3 * q : (SK == 0 ? 62 + 2 * q : 22 + 2 * q);
    const bool head = valid && q == 0;   // one lane per env: env-level scalars
    const int elg = wv * EPW + elw;      // env inside the workgroup
    const int ec = valid ? e : e0;       // clamped env / port for the unconditional loads of idle lanes
    const int gc = valid ? g : e0 * P;
    const int scn = ev2g_scn(ec, off, M);             // this env's scenario in the resident pool
    double c_imax, c_dmaxabs, a_next;
    int ta, td;                          // the port's occupancy window: env-lane-only state, kept in registers
    double bcap = 1.0, potc = 0.0;       // ... and the attached EV's battery size / charge-power-potential term (LDS: two workgroups per CU must fit)
    {   // launch prologue (ev2g_step_wave's): windows, charger constants, first action and accumulators in one round trip, the
        // per-EV state in a second one where an EV is attached
        const unsigned g8 = (unsigned)g * 8u, c8 = (unsigned)(valid ? q : 0) * 8u, cp8 = (unsigned)min(tid, P - 1) * 8u;
        i2v w = ldg32<i2v>(PA(EV2G_PS_WIN), g8), sc = ldg32<i2v>(PA(EV2G_PS_SC), g8);
        int lut0 = ldg32<int>(PA(EV2G_PS_LUT), g8 >> 1);
        const d2v k_max = ldg32<d2v>(wa.cs_pack, c8 * 6u);
        d2v k_min = {0.0, 0.0}, k_pow = {0.0, 0.0};
        if (tid < 64) { k_min = ldg32<d2v>(wa.cs_pack, cp8 * 6u + 16u); k_pow = ldg32<d2v>(wa.cs_pack, cp8 * 6u + 32u); }
        a_next = ldg32<double>(io.actions, (unsigned)gc * 8u);
        double l_pot = ldg32<double>(slabH + HS8, ((unsigned)min(t0, T - 1) * (unsigned)E + (unsigned)ec) * 8u);
        double l_pot2 = 0.0;
        if (RK == 3) l_pot2 = ldg32<double>(slabH + HS8, ((unsigned)min(max(t0 - 1, 0), T - 1) * (unsigned)E + (unsigned)ec) * 8u);
        d2v acc01 = ldg32<d2v>(env_acc, (unsigned)ec * 64u), acc23 = ldg32<d2v>(env_acc, (unsigned)ec * 64u + 16u);
        double acc4 = ldg32<double>(env_acc, (unsigned)ec * 64u + 32u);
        d2v k_max_w = k_max;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(w), "+v"(sc), "+v"(lut0), "+v"(k_max_w), "+v"(k_min), "+v"(k_pow),
                     "+v"(a_next), "+v"(l_pot), "+v"(l_pot2), "+v"(acc01), "+v"(acc23), "+v"(acc4));
        c_imax = k_max_w.x; c_dmaxabs = k_max_w.y;
        ta = valid ? w.x : EV2G_INT_MAX; td = valid ? w.y : -1;
        if (tid < P) {
            s_cst[0 * 64 + tid] = k_min.x - 0.01; s_cst[1 * 64 + tid] = k_min.y;
            s_cst[2 * 64 + tid] = k_pow.x; s_cst[3 * 64 + tid] = k_pow.y;
        }
        if (valid) {
            s_ss[tid] = sc.x; s_cyc[tid] = sc.y;
            s_dirty[tid] = (lut0 + 1) << 8;   // bits 0,1: write-back flags; bit 2: emergency crossing of the step; bits 8..: 1 + efficiency-table id
            if (w.x <= t0 && t0 <= w.y) {
                s_cap[tid] = ldg32<double>(PA(EV2G_PS_CAP), g8); s_tot[tid] = ldg32<double>(PA(EV2G_PS_TOT), g8);
                s_prev[tid] = ldg32<double>(PA(EV2G_PS_PREV), g8);
                bcap = ldg32<double>(PA(EV2G_PS_BCAP), g8); potc = ldg32<double>(PA(EV2G_PS_POTC), g8);
                s_abse[tid] = log_soc ? ldg32<double>(PA(EV2G_PS_ABSE), g8) : 0.0;
            } else {
                s_cap[tid] = 0.0; s_tot[tid] = 0.0; s_prev[tid] = 0.0; s_abse[tid] = 0.0;
            }
        }
        if (head) {
            double *ea = eacc + elg * 7;
            ea[0] = acc01.x; ea[1] = acc01.y; ea[2] = acc23.x; ea[3] = acc23.y; ea[4] = acc4;
            ea[5] = (t0 < T) ? l_pot : 0.0;
            ea[6] = (t0 > 0 && t0 <= T) ? l_pot2 : 0.0;
        }
    }
    __syncthreads();

    // ---- what one iteration hands to the next --------------------------------------------------------------------------------
    // from A(kk), consumed by grab / Cs / Co of the same step one iteration later
    bool occ = false;            // an EV was attached during the step
    int isgn = 0;                // +1 / -1: a charge / discharge item was posted for this port, 0: none
    double cap_before = 0.0;     // capacity before EV.step (SoC log)
    // prefetch group 1, issued after X(kk): the next action and the session-record tail for the departure / arrival of step kk
    double pf_B = 0.0, pf_des = 0.0, pf_cap0 = 0.0;
    d2v pf_r5 = {0.0, 0.0};      // (pacmax, v)
    int pf_nta = 0, pf_ntd = 0, pf_lut = 0;
    // prefetch group 2, issued before Y(kk): what the output side of step kk reads
    double pf_pch = 0.0, pf_pdis = 0.0, pf_ob0 = 0.0;
    d2v pf_tr = {0.0, 0.0}, pf_h0 = {0.0, 0.0}, pf_h1 = {0.0, 0.0};
    constexpr int NHEAD = (SK == 1) ? 0 : (SK == 0 ? 60 : 20);   // 20 prices (+ 40 window columns)
    constexpr int NPAIR = NHEAD / 2;
    const unsigned eT64 = (unsigned)(scn * T) * 64u;  // this env's rows in the [M,T,8] step table

    PP_DECL
    for (int kk = 0; kk <= k_steps; kk++) {
        asm volatile("" : "+s"(S));
        PP_MARK(8)
        // per-lane identity: three opaque registers per iteration (fenced from loop-invariant hoisting like ev2g_step_wave's),
        // everything else re-derived from them where it is used -- this kernel has no registers to park invariants in
        int tid_l = tid, e_l = e, qe_l = q | (elw << 8);
        asm volatile("" : "+v"(tid_l), "+v"(e_l), "+v"(qe_l));
        const int q_l = qe_l & 255, lane_l = tid_l & 63;
        const int g_l = valid ? e_l * P + q_l : 0;
        const unsigned g8 = (unsigned)g_l * 8u;
        const bool fin = kk > 0;              // step kk-1 is to be finished
        const bool start = kk < k_steps;      // step kk is to be started
        const int t = t0 + kk;                // the step to start; the step to finish is tp = t - 1 and its sstep is t
        const int tp = t - 1;
        const bool last_step = !start;        // (for the step being finished)

        // what the output side of step tp needs from its state side
        double f_energy = 0.0, f_current = 0.0, f_cap = 0.0, f_capd = 0.0, f_tot = 0.0, f_des = 0.0, f_abse = 0.0, f_capb = cap_before;
        int f_ss = 0, f_isgn = isgn, f_emerg = 0;
        bool f_occ = occ, f_dep = false;
        if (fin) {
            // ---- collect the prefetches of the previous iteration (and whatever is left of its stores) ----
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
            asm volatile("" : "+v"(a_next), "+v"(pf_B), "+v"(pf_des), "+v"(pf_cap0), "+v"(pf_r5), "+v"(pf_nta), "+v"(pf_ntd), "+v"(pf_lut));
            asm volatile("" : "+v"(pf_pch), "+v"(pf_pdis), "+v"(pf_tr), "+v"(pf_ob0), "+v"(pf_h0), "+v"(pf_h1));
            PP_MARK(0)
            if (valid) {
                // ---- grab: what B(tp) left for this port ----
                if (occ) {
                    f_cap = s_cap[tid_l];
                    if (SK == 1) f_tot = s_tot[tid_l];
                    if (isgn != 0) { f_energy = s_amps[tid_l]; f_current = s_cur[tid_l]; f_emerg = (s_dirty[tid_l] >> 2) & 1; }
                    // ---- Cs: departure (ev_charger.py:209-229, ev.py:191-214): state side ----
                    if (tp >= td) {
                        f_dep = true;
                        f_capd = f_cap;        // the capacity the EV leaves with
                        f_ss = s_ss[tid_l];
                        f_des = pf_des;
                        if (log_soc) f_abse = s_abse[tid_l];
                        ta = pf_nta; td = pf_ntd;   // window of the port's next session
                        s_ss[tid_l] = (ta != EV2G_INT_MAX) ? f_ss + 1 : -1;
                        s_cyc[tid_l] = 0;
                        s_dirty[tid_l] |= 2;
                    }
                }
                // ---- Cs: arrival at the end of step tp (ev2gym_env.py:399-417, ev.py:115-136) ----
                if (ta == t) {
                    if (f_dep) {   // the next session arrives right behind a departure of this very step (replayed scenarios only): its
                                   // record was not the one prefetched
                        const unsigned r8 = (unsigned)s_ss[tid_l] * (unsigned)sizeof(SessRec);
                        pf_B = ldg32<double>(S->rec, r8 + 72u); pf_r5 = ldg32<d2v>(S->rec, r8 + 80u);
                        pf_cap0 = ldg32<double>(S->rec, r8 + 96u);
                        const i4v r7 = ldg32<i4v>(S->rec, r8 + 112u);
                        pf_lut = r7.z;
                    }
                    f_cap = pf_cap0;
                    const double B = pf_B;
                    const double v = pf_r5.y;
                    const double evc = pf_r5.x * 1000.0 / v;            // utils.py:773-777
                    potc = v * ((evc < c_imax) ? evc : c_imax) / 1000.0;
                    bcap = B;
                    s_cap[tid_l] = f_cap; s_tot[tid_l] = 0.0; s_prev[tid_l] = 0.0; s_cyc[tid_l] = 0;
                    s_abse[tid_l] = 0.0;
                    f_tot = 0.0;
                    const int lut_new = pf_lut;
                    stg32<int>(PA(EV2G_PS_LUT), g8 >> 1, lut_new);
                    stg32<double>(PA(EV2G_PS_BCAP), g8, B);
                    stg32<double>(PA(EV2G_PS_POTC), g8, potc);
                    stg32<double>(PA(EV2G_PS_PENERGY), g8, 0.0);
                    stg32<double>(PA(EV2G_PS_PCURRENT), g8, 0.0);
                    s_dirty[tid_l] = (s_dirty[tid_l] & 3) | 1 | ((lut_new + 1) << 8);
                }
            }
        }

        PP_MARK(1)
        if (start) {
            // ---------------- A: charger level of step t (ev_charger.py:137-186) ----------------
            int *cntk = cnt + 2 * (kk & 1);
            occ = false; isgn = 0;
            if (valid) {
                occ = (ta <= t) && (t <= td);
                if (log_soc && occ) cap_before = s_cap[tid_l];
                double a = occ ? a_next : 0.0;
                // one port per charger: a / sum(a) = a / a, which is exactly +-1 for every finite action (ev_charger.py:143-149)
                if (a > 1.0) a = 1.0;
                else if (a < -1.0) a = -1.0;
                double amps = 0.0;
                if (occ) {
                    const double x = rnd5_x(a);
                    if (x > 0.0) { amps = x * c_imax; if (amps < s_cst[0 * 64 + q_l]) amps = 0.0; }
                    else if (x < 0.0) { const double c_dmin = s_cst[1 * 64 + q_l]; amps = x * c_dmaxabs; if (amps > c_dmin - 0.01) amps = c_dmin; }
                }
                if (amps != 0.0) {
                    s_amps[tid_l] = amps;
                    isgn = (amps > 0.0) ? 1 : -1;
                    items[atomicAdd(&cntk[0], 1)] = tid_l;
                }
            }
            PP_MARK(2)
            lds_barrier();   // X: the workers start B(t)
            PP_MARK(3)

            // ---- prefetch group 1: the action of step t+1, the record tail for this step's departure / arrival ----
            const bool more = kk + 1 < k_steps;
            a_next = ldg32_nt<double>(io.actions + (long long)(more ? kk + 1 : kk) * io.a_stride, (unsigned)gc * 8u);
            const bool ev_dep = occ && t >= td, ev_arr = (ta == t + 1);
            if (__ballot(ev_dep || ev_arr) != 0ull) {   // (uniform)
                const unsigned r8 = (ev_dep || ev_arr) ? (unsigned)s_ss[tid_l] * (unsigned)sizeof(SessRec) : 0u;
                static_assert(offsetof(SessRec, B) == 72 && offsetof(SessRec, pacmax) == 80 && offsetof(SessRec, v) == 88 && offsetof(SessRec, cap0) == 96 &&
                              offsetof(SessRec, des) == 104 && offsetof(SessRec, nt_arr) == 112 && offsetof(SessRec, lut) == 120, "SessRec tail layout");
                pf_B = ldg32<double>(S->rec, r8 + 72u); pf_r5 = ldg32<d2v>(S->rec, r8 + 80u);
                const d2v r6 = ldg32<d2v>(S->rec, r8 + 96u);
                const i4v r7 = ldg32<i4v>(S->rec, r8 + 112u);
                pf_cap0 = r6.x; pf_des = r6.y; pf_nta = r7.x; pf_ntd = r7.y; pf_lut = r7.z;
            }
        }

        if (fin) {
            // ---------------- Co: output side of phase C for step tp (sstep = t) ----------------
            const int sstep = t;
            double *obs = io.obs ? io.obs + (long long)(kk - 1) * io.o_stride : nullptr;       // uniform bases (scalar arithmetic)
            float *obs32 = S->x_obs32 ? (float *)S->x_obs32 + (long long)(io.step0 + kk - 1) * S->x_o32_stride : nullptr;
            uint8_t *mask = io.mask ? io.mask + (long long)(kk - 1) * io.m_stride : nullptr;
            bool occ_any = false;   // an EV on this port before or after the step
            if (valid) {
                double profit = 0.0, satpen = 0.0, pot = 0.0;
                double r_pow = 0.0, r_ch = 0.0, r_dis = 0.0;
                if (f_occ) {
                    if (f_isgn != 0) {
                        const double dtd = (double)S->dt;   // (re-derived per use: a loop-carried copy costs a VGPR pair this kernel does not have)
                        r_pow = f_energy * 60.0 / dtd;
                        if (f_isgn > 0) r_ch = fabs(f_energy); else r_dis = fabs(f_energy);
                        if (f_energy != 0.0) profit = (f_isgn > 0) ? r_ch * pf_pch : r_dis * pf_pdis;   // by the sign of the ACTION (ev_charger.py:178,194)
                    }
                    if (f_current - 0.0001 > c_imax) stg32<int>(S->env_fault, (unsigned)e_l * 4u, 1);  // ev_charger.py:203-205
                    if (last_step) { stg32<double>(PA(EV2G_PS_PENERGY), g8, f_energy); stg32<double>(PA(EV2G_PS_PCURRENT), g8, f_current); }
                    if (log_soc) stg32<double>(S->soc_log + (long long)tp * P, g8 + (unsigned)e_l * (unsigned)((T - 1) * P * 8), (f_current != 0.0) ? f_capb : -f_capb);
                    if (f_dep) {  // departure (ev_charger.py:209-229, ev.py:191-214): the output side
                        const double score = (f_capd < f_des - 0.001) ? f_capd / f_des : 1.0;
                        if (RK == 3) satpen = ev2g_departure_term(S->reward_kind, S->cost_kind, score, f_capd, f_des);
                        else if (RK != 1 || S->cost_kind == 1) satpen = 100.0 * exp(-10.0 * score);
                        __hip_atomic_fetch_add((int __attribute__((address_space(1))) *)(PA(EV2G_PS_SERVED) + (g8 >> 1)), 1,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_fetch_add((double __attribute__((address_space(1))) *)(PA(EV2G_PS_SATSUM) + g8), score,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        stg32<double>(slabS, (unsigned)f_ss * 8u, f_capd);
                        if (log_soc) stg32<double>((slabS + SS8), (unsigned)f_ss * 8u, f_abse);
                    }
                }
                const bool occ_after = (ta <= sstep) && (sstep <= td);
                if (RK == 3 && occ_after && S->reward_kind >= 9) {   // (pst_)V2G_profitmaxV2: every connected EV (reward.py:173-195)
                    const unsigned r8 = (unsigned)s_ss[tid_l] * (unsigned)sizeof(SessRec);
                    satpen += ev2g_connected_term(ldg32<double>(S->rec, r8 + (unsigned)offsetof(SessRec, des)), f_cap,
                                                  ldg32<double>(S->rec, r8 + (unsigned)offsetof(SessRec, pacmax)), S->sixty_over_dt, td, sstep);
                }
                occ_any = f_occ || occ_after;
                if (mask) stg32<uint8_t>(mask, (unsigned)g_l, occ_after ? 1 : 0);
                double o0 = 0.0, o1 = 0.0, o2 = 0.0;
                if (occ_after) {
                    const double soc = f_cap / bcap;
                    if (SK == 1) { o0 = (soc == 1.0) ? 1.0 : 0.5; o1 = f_tot; o2 = (double)(sstep - ta); }
                    else { o0 = soc; o1 = (double)(td - sstep); }
                    if (soc < 1.0 && td > sstep) pot = potc;  // utils.py:771
                }
                {   // per-charger clamp (utils.py:779-789)
                    const double c_maxp = s_cst[2 * 64 + q_l], c_minp = s_cst[3 * 64 + q_l];
                    pot = (pot > c_maxp) ? c_maxp : ((pot < c_minp) ? 0.0 : pot);
                }
                if (obs) {
                    const unsigned o8 = (unsigned)(e_l * D + ocol) * 8u;
                    stg32<d2v>(obs, o8, (d2v){o0, o1});
                    if (SK == 1) stg32<double>(obs, o8 + 16u, o2);
                }
                if (obs32) {
                    const unsigned o4 = (unsigned)(e_l * D + ocol) * 4u;
                    if (SK == 1) { stg32<float>(obs32, o4, (float)o0); stg32<float>(obs32, o4 + 4u, (float)o1); stg32<float>(obs32, o4 + 8u, (float)o2); }
                    else stg32<f2v>(obs32, o4, (f2v){(float)o0, (float)o1});
                }
                stage[0 * RS + tid_l] = r_pow;
                stage[1 * RS + tid_l] = profit;
                stage[2 * RS + tid_l] = satpen;
                stage[3 * RS + tid_l] = pot;
                stage[4 * RS + tid_l] = r_ch;
                stage[5 * RS + tid_l] = r_dis;
                stage[6 * RS + tid_l] = (double)f_emerg;
                stage[7 * RS + tid_l] = f_current;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wavefront's LDS writes are visible to itself
            PP_MARK(4)

            // ---------------- D: per-env reduction inside the wavefront (ev2g_step_wave's fixed tree) ----------------
            double esum[EV2G_NQ];
#pragma unroll
            for (int kq = 0; kq < EV2G_NQ; kq++) esum[kq] = 0.0;
            if (__ballot(occ_any) != 0ull) {   // (uniform)
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
                        if (j == 0) stage[k * RS + a] = acc;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int kq = 0; kq < EV2G_NQ; kq++) esum[kq] = stage[kq * RS + (tid_l - q_l)];   // the env's sums (its head slot), every lane
            }

            PP_MARK(5)
            // ---------------- E: per env (head lane) + observation head (the env's lanes) ----------------
            const double usage = esum[0];
            const double pf_base = pf_tr.x, pf_maxp = pf_tr.y;
            const double pf_minp = dpp_mov_f64<0x130>(pf_tr.x), pf_sp = dpp_mov_f64<0x130>(pf_tr.y);
            const double tr_power = pf_base + usage;   // inflexible_load[t] + solar_power[t] + sum of the charger powers (transformer.py:258-302)
            const double over = (tr_power > pf_maxp + 0.0001 || tr_power < pf_minp - 0.0001) ? fabs(tr_power - pf_maxp) : 0.0;
            if (P >= 3) {
                const double over_n = dpp_mov_f64<0x138>(over);   // wave_shr:1: lane L takes lane L-1's value
                if (valid && q_l < 3 && (q_l != 2 || sstep < T)) {
                    const double hv = (q_l == 0) ? usage : ((q_l == 1) ? over_n : esum[3]);
                    const unsigned hoff = (q_l == 0) ? 0u : ((q_l == 1) ? 2u * (unsigned)HS8 : (unsigned)HS8);
                    const unsigned hstep = (q_l == 2) ? (unsigned)sstep : (unsigned)tp;
                    stg32<double>(slabH, hoff + (hstep * (unsigned)E + (unsigned)e_l) * 8u, hv);
                }
            }
            if (head) {
                double *ea = eacc + ((tid_l >> 6) * EPW + (qe_l >> 8)) * 7;
                const double ea0 = ea[0], ea1 = ea[1], ea2 = ea[2], ea3 = ea[3], ea4 = ea[4], ea5 = ea[5];
                const double ea6 = (RK == 3) ? ea[6] : 0.0;
                const unsigned e8 = (unsigned)e_l * 8u;
                double over100 = 0.0;
                if (RK == 0 || RK == 3) over100 = 100.0 * over;
                if (last_step) stg32<double>(S->tr_power_now, e8, tr_power);
                const double potn = esum[3];
                if (P < 3) {   // two-port envs: no third lane to share the history stores with
                    stg32<double>((slabH + 2 * HS8) + (long long)tp * E * 8, e8, over);
                    stg32<double>(slabH + (long long)tp * E * 8, e8, usage);
                    if (sstep < T) stg32<double>((slabH + HS8) + (long long)sstep * E * 8, e8, potn);
                }
                const double costs = esum[1];
                double reward;
                if (RK == 1) {  // SquaredTrackingErrorReward reward.py:7-14
                    const double pp = ea5;
                    const double m = (pp < pf_sp) ? pp : pf_sp;
                    const double d = m - usage;
                    reward = -(d * d);
                } else if (RK == 2) {  // profit_maximization reward.py:78-87
                    reward = costs - esum[2];
                } else if (RK == 3) {  // the other fused rewards, by V2P::reward_kind
                    RewardIn ri;
                    ri.costs = costs; ri.usage = usage; ri.sp = pf_sp; ri.pot_t = ea5; ri.pot_tm1 = ea6; ri.over100 = over100;
                    ri.user = esum[2]; ri.tr0_maxp = pf_maxp;
                    reward = ev2g_reward(S->reward_kind, ri);
                } else {  // ProfitMax_TrPenalty_UserIncentives reward.py:34-44
                    reward = costs - over100 - esum[2];
                }
                const double n0 = ea0 + reward, n1 = ea1 + costs, n2 = ea2 + esum[4], n3 = ea3 + esum[5], n4 = ea4 + esum[6];
                ea[0] = n0; ea[1] = n1; ea[2] = n2; ea[3] = n3; ea[4] = n4; ea[5] = potn;
                if (RK == 3) ea[6] = ea5;
                if (io.reward) stg32<double>(io.reward + (long long)(kk - 1) * io.r_stride, e8, reward);
                if (io.done) stg32<uint8_t>(io.done + (long long)(kk - 1) * io.d_stride, (unsigned)e_l, (sstep >= T) ? 1 : 0);
                if (S->x_cost)   // cost_function (rl_agent/cost.py:8-27)
                    stg32<double>(S->x_cost + (long long)(io.step0 + kk - 1) * S->x_c_stride, e8, (S->cost_kind == 2) ? costs : 100.0 * over + esum[2]);
                if (sstep >= T || last_step) {  // publish the running episode totals (get_statistics reads them)
                    const unsigned a8 = (unsigned)e_l * 64u;
                    stg32<d2v>(env_acc, a8, (d2v){n0, n1});
                    stg32<d2v>(env_acc, a8 + 16u, (d2v){n2, n3});
                    stg32<double>(env_acc, a8 + 32u, n4);
                }
            }
            if (valid && obs32) {
                const unsigned o4 = (unsigned)(e_l * D) * 4u;
                if (SK == 1) {
                    if (q_l == 0) {
                        stg32<float>(obs32, o4, (float)((double)sstep / (double)T));
                        stg32<float>(obs32, o4 + 4u, (float)((sstep < T) ? pf_ob0 : 0.0));
                        stg32<float>(obs32, o4 + 8u, (float)usage);
                    }
                } else {
                    if (q_l == 0) stg32<f2v>(obs32, o4, (f2v){(float)sstep, (float)usage});
                    if (q_l < NPAIR) stg32<f2v>(obs32, o4 + 8u + (unsigned)q_l * 8u, (f2v){(float)pf_h0.x, (float)pf_h0.y});
                    if (q_l + P < NPAIR) stg32<f2v>(obs32, o4 + 8u + (unsigned)(q_l + P) * 8u, (f2v){(float)pf_h1.x, (float)pf_h1.y});
                    const unsigned h8 = (unsigned)((scn * (T + 1) + sstep) * NHEAD) * 8u;
                    for (int pi = q_l + 2 * P; pi < NPAIR; pi += P) {
                        const d2v hv = ldg32<d2v>(S->head_tab, h8 + (unsigned)pi * 16u);
                        stg32<f2v>(obs32, o4 + 8u + (unsigned)pi * 8u, (f2v){(float)hv.x, (float)hv.y});
                    }
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
                    if (q_l == 0) stg32<d2v>(obs, o8, (d2v){(double)sstep, usage});
                    if (q_l < NPAIR) stg32<d2v>(obs, o8 + 16u + (unsigned)q_l * 16u, pf_h0);
                    if (q_l + P < NPAIR) stg32<d2v>(obs, o8 + 16u + (unsigned)(q_l + P) * 16u, pf_h1);
                    const unsigned h8 = (unsigned)((scn * (T + 1) + sstep) * NHEAD) * 8u;
                    for (int pi = q_l + 2 * P; pi < NPAIR; pi += P)    // tiny envs (P < 15): the remaining pairs, unprefetched
                        stg32<d2v>(obs, o8 + 16u + (unsigned)pi * 16u, ldg32<d2v>(S->head_tab, h8 + (unsigned)pi * 16u));
                }
            }
        }

        PP_MARK(6)
        if (start) {
            // ---- prefetch group 2: what the output side of step t reads (consumed one iteration later, after X) ----
            const int sstep_n = t + 1;
            const unsigned et64 = eT64 + (unsigned)t * 64u;
            const d2v st0 = ldg32_nt<d2v>(S->step_tab, et64);                                  // charge price, discharge price
            pf_pch = st0.x; pf_pdis = st0.y;
            // the head lane (q == 0) takes {inflexible + solar, max_power}, its neighbour {min_power, setpoint} (DPP wave shift in E)
            pf_tr = ldg32_nt<d2v>(S->step_tab, et64 + ((q_l == 0) ? 16u : 32u));
            if (SK == 1) {
                pf_ob0 = ldg32_nt<double>(S->step_tab, eT64 + (unsigned)min(sstep_n, T - 1) * 64u + 40u);   // next setpoint; head lane only, masked by sstep < T
            } else {
                const unsigned h8 = (unsigned)((scn * (T + 1) + sstep_n) * NHEAD) * 8u;
                pf_h0 = ldg32_nt<d2v>(S->head_tab, h8 + (unsigned)min(q_l, NPAIR - 1) * 16u);
                if (P < NPAIR) pf_h1 = ldg32_nt<d2v>(S->head_tab, h8 + (unsigned)min(q_l + P, NPAIR - 1) * 16u);   // (uniform)
            }
            PP_MARK(7)
            lds_barrier();   // Y: the workers finished B(t)
        }
    }
    if (tid == 0) { PP_FLUSH(0, 9) }
    __syncthreads();
    if (valid) {
        const int d = s_dirty[tid];
        const unsigned g8 = (unsigned)g * 8u;
        if (d & 2) stg32<i2v>(PA(EV2G_PS_WIN), g8, (i2v){ta, td});
        if (d & 3) stg32<i2v>(PA(EV2G_PS_SC), g8, (i2v){s_ss[tid], s_cyc[tid]});
        if (d & 1) {
            stg32<double>(PA(EV2G_PS_CAP), g8, s_cap[tid]); stg32<double>(PA(EV2G_PS_TOT), g8, s_tot[tid]); stg32<double>(PA(EV2G_PS_PREV), g8, s_prev[tid]);
            if (log_soc) stg32<double>(PA(EV2G_PS_ABSE), g8, s_abse[tid]);
        }
    }
}
