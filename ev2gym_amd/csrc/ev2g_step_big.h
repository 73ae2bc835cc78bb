// ev2g_step_big.h -- the step kernel for BIG envs (512 < P <= 1024 ports, one env per workgroup; BASELINE configs[3]: 1000 chargers on
// 50 transformers), launched the way ev2g_step_v2<.., SPEC = 1> is (default plugin pair, every float64 output with step stride 0, SoC log on).
//
// Why it exists (round 6).  ev2g_step_v2<1024> needs one 1024-thread workgroup per env and 148 bytes of LDS per port: ONE workgroup per CU, by
// registers (16 wavefronts x 128 VGPRs) and by LDS (148 KB) alike.  Its step is a chain of phases separated by barriers -- home lanes, then the
// battery maths on a fifth of the lanes, then home lanes again, a reduction, one wavefront of env-level work -- and with a single workgroup
// resident nothing overlaps that chain: the CU streams its 91 KB per env-step at about half of what it could (0.41 of the roofline, traffic 1.05 x
// algorithmic: the one BASELINE shape that really streams).  This kernel runs the same step with 512 threads, TWO PORTS PER HOME LANE, and 70
// bytes of LDS per port, so that TWO workgroups (two envs) are resident per CU and one's memory phases run under the other's compute chain:
//   * LDS per port: capacity, total / previous energy, abs-energy, battery size (40 B), two 8-byte hand-over words (amps -> energy, power), the
//     window as two 16-bit step numbers, session index, one word of cycles | flags | charger class (12 B), a 2-byte work-list entry.  No per-port
//     staging rows: the seven env-level sums are reduced from REGISTERS (DPP butterflies per wavefront, eight partials per quantity through LDS),
//     only the port powers -- needed per transformer -- are staged, in the hand-over word the battery maths already writes;
//   * what only the battery maths can know (the real current: over-current fault, the SoC log's activity sign, the last step's port readings) is
//     written by the worker lane itself; the home lanes never read it back;
//   * the charge-power-potential term of an attached EV is not kept in LDS as a value: it takes a handful of distinct values (car model x charger),
//     the launch holds them in a 15-entry LDS table and the port's word carries a 4-bit index (15: not in the table -- a device refill's new
//     model -- fetch the port's own state line);
//   * departures and arrivals are known before the step: their operands are requested in phase A, two barriers ahead of their use;
//   * charger constants come from a per-launch class table in LDS (<= 16 distinct charger tuples; the shipped configs have one);
//   * one env per workgroup makes everything per-env wave-uniform: prices and scenario rows are scalar loads;
//   * the 2000 window columns of the observation head (a copy of the scenario's window table row, state.py:128-151) are requested at the end of
//     phase A and stored behind the battery maths by every wavefront: phase E is wavefront 0 alone, the others go straight into the next step;
//   * three workgroup barriers per step (A | B | C + D | E), v2's one-env scheme has four.
// Arithmetic: the same functions (ev_math_charge / ev_math_discharge, rnd5_x, ceil2_x) in the same order.  Transformer powers use v2's tree
// (8 lanes per segment, two chains... one chain here as there: identical), so overloads agree bit for bit; the six env-level sums (profit, user term,
// potential, charged / discharged energy, violations) use a different fixed tree than v2's (lane pair -> wavefront butterfly -> eight partials):
// deterministic, but the last bit of reward / cost sums may differ from the general instantiation's (tests hold them to 1e-12, ints exactly).
#pragma once
#include "ev2g_step_v2.h"

#define EV2G_BIG_BLOCK 512
#ifndef EV2G_BIG_PRIO
#define EV2G_BIG_PRIO 3
#endif
#define EV2G_BIG_NCC 16          // charger classes (distinct constant tuples) the LDS table holds
#define EV2G_BIG_TMAX 32766      // windows are kept as 16-bit step numbers (0x7fff = none)

struct BigArgs {
    const unsigned char *slot_ccls;   // [P] charger class of every port slot
    const double *ccls_tab;           // [ncc][6] imax, imin, dmin, |dmax|, max power, min power
    const double *potc_tab;           // [15] the distinct charge-power-potential terms of the loaded sessions (unused entries: NaN)
    int ncc;
};

__host__ __device__ inline size_t ev2g_big_lds_bytes(int P, int R) {
    const size_t NP = ((size_t)P + 1) & ~(size_t)1;
    return 8 * (7 * NP + (size_t)R + 5 * 8 + 8 + 8 + (size_t)EV2G_BIG_NCC * 6 + 16) + 4 * (3 * NP + 2 * (size_t)R + 4 + 8) + 2 * (2 * NP + 2 * EV2G_BIG_BLOCK);
}

__device__ __forceinline__ int big_pack16(int v) { return (v == EV2G_INT_MAX) ? 0x7fff : (v & 0xffff); }
__device__ __forceinline__ int big_unpack16(int v16) { const int v = (int)(short)v16; return (v == 0x7fff) ? EV2G_INT_MAX : v; }

// uniform (scalar-cache) load of a read-only scenario value
template <class T> __device__ __forceinline__ T big_sld(EV2G_GP(const T) p, long long i) {
    return *((const T __attribute__((address_space(4))) *)(unsigned long long)p + i);
}

template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, 4) ev2g_step_big(const V2P *__restrict__ params, StepIO io, int t0, int k_steps, BigArgs ba) {
    extern __shared__ double lds[];
    typedef const V2P __attribute__((address_space(4))) *ParamPtr;
    ParamPtr S = (ParamPtr)(unsigned long long)params;
    constexpr int NW = BLOCK / 64;
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const int P = S->P, R = S->R, T = S->T, C = S->C, E = S->E, D = S->D, M = S->M;
    int e;
    {   // XCD-aware mapping: workgroup b runs on XCD b % 8; give each XCD a contiguous range of envs
        const int nb = gridDim.x, b = blockIdx.x, per = nb >> 3;
        e = (nb & 7) == 0 ? (b & 7) * per + (b >> 3) : b;
    }
    if (e >= E) return;
    const int scn = ev2g_scn(e, io.scn_off, M);
    const int NP = (P + 1) & ~1;
    double *s_cap = lds, *s_tot = s_cap + NP, *s_prev = s_tot + NP, *s_abse = s_prev + NP, *s_bcap = s_abse + NP;
    double *s_x = s_bcap + NP;        // phase A: amps; phase B: EV.current_energy of the step
    double *s_y = s_x + NP;           // the port's power this step (0 unless the battery maths writes it): what the transformers add up
    double *tsum = s_y + NP;          // [R] power of every transformer's chargers
    double *wsum = tsum + R;          // [5][NW] per-wavefront partial sums: profit, user term, potential, charged, discharged
    double *eacc = wsum + 5 * NW;     // [5] episode accumulators (+ 3 pad)
    double *emg = eacc + 8;           // [NW] emergency-capacity violations of the step, per wavefront (counts)
    double *ctab = emg + 8;           // [NCC][6] charger classes
    double *ptab = ctab + EV2G_BIG_NCC * 6;   // [16] distinct potential terms (entry 15 unused: index 15 = "not in the table, fetch the port's state line")
    int *s_tatd = (int *)(ptab + 16);   // t_arr | t_dep << 16 of the attached-or-next session
    int *s_ss = s_tatd + NP;
    int *s_cycd = s_ss + NP;          // bit 0: cap/tot/prev/cycles changed, 1: window changed, 2: violation this step, 3: this step's item charged (else discharged);
                                      // bits 8..23 charging cycles; bits 24..27 charger class; bits 28..31 potential-term index of the attached EV
    int *seg = s_cycd + NP, *trobs = seg + R + 1, *cnt = trobs + R;
    int *cntev = cnt + 4;             // [NW] ports with a departure or an arrival in this step, per wavefront
    unsigned short *items = (unsigned short *)(cntev + NW);
    unsigned short *s_lut = items + NP;   // efficiency-table id + 1 of the attached EV (0: none), so that the battery maths issues the table look-up WITH the
                                          // record loads, not behind a dependent fetch of the id
    unsigned short *evl = s_lut + NP;     // [NW][128] the step's event ports, wavefront w's in lane order at evl + 128 w (a fixed place per wavefront: the
                                          // order in which the LAST wavefront works them off -- and adds their terms up -- does not depend on timing)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double dtd = (double)S->dt, sixty_over_dt = S->sixty_over_dt, dt_over_60 = S->dt_over_60;

    // ---- launch prologue: this lane's two ports, global state -> LDS ----
    int pk[2];          // action / mask index of the slot (low 16 bits) | its first observation column (high 16 bits)
    double a_next[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int q = tid + u * BLOCK;
        const bool valid = q < P;
        const int qc = valid ? q : 0;
        const int pref = S->slot_port[qc];
        pk[u] = pref | (S->slot_obs[qc] << 16);
        a_next[u] = io.actions[e * P + pref];
        if (valid) {
            const EV2G_GP(PortLine) ln = S->line + (e * P + q);
            const int ta = ln->ta, td = ln->td;
            s_tatd[q] = (int)((unsigned)big_pack16(ta) | ((unsigned)big_pack16(td) << 16));
            s_ss[q] = ln->ss;
            s_lut[q] = (unsigned short)(ev2g_line_lut(ln->cyc_lut) + 1);
            const bool body = (ta <= t0) && (t0 <= td);
            unsigned pidx = 15u;
            if (body) { const double pc = ln->potc; for (int i = 14; i >= 0; i--) if (ba.potc_tab[i] == pc) pidx = (unsigned)i; }
            s_cycd[q] = (int)((unsigned)(ev2g_line_cycles(ln->cyc_lut) << 8) | ((unsigned)ba.slot_ccls[q] << 24) | (pidx << 28));
            s_cap[q] = body ? ln->cap : 0.0; s_tot[q] = body ? ln->tot : 0.0; s_prev[q] = body ? ln->prev : 0.0;
            s_abse[q] = body ? ln->abse : 0.0; s_bcap[q] = body ? ln->bcap : 1.0;
        }
    }
    // the first step's actions are collected HERE: a request still pending at the loop's entry would make every iteration's first use of the
    // register a conservative vmcnt(0) -- a drain of the previous step's stores in the middle of phase A
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a_next[0]), "+v"(a_next[1]));
    if (tid < 3) cnt[tid] = 0;
    for (int i = tid; i <= R; i += BLOCK) seg[i] = S->tr_seg[i];
    for (int i = tid; i < R; i += BLOCK) trobs[i] = S->tr_obs[i];
    if (tid < 8) eacc[tid] = 0.0;
    for (int i = tid; i < EV2G_BIG_NCC * 6; i += BLOCK) ctab[i] = (i < ba.ncc * 6) ? ba.ccls_tab[i] : 0.0;
    if (tid < 16) ptab[tid] = (tid < 15) ? ba.potc_tab[tid] : 0.0;
    __syncthreads();
    // the observation-head pairs this lane copies every step: pair pi = tid + u * BLOCK of the 20 R window-column pairs; transformer r = pi / 20, pair jj = pi % 20
    // ... and the 20 |charge price| columns (state.py:121-129) ride in pair slots 20 R .. 20 R + 19 (R <= 50: 1020 slots of 1024): hp_dst = -2 - column
#define EV2G_BIG_HP_ROLE(u, src, dst)                                                                                                    \
    {                                                                                                                                    \
        const int pi_ = tid_l + (u) * BLOCK;                                                                                             \
        if (pi_ < 20 * R) { const int r_ = pi_ / 20, jj_ = pi_ - r_ * 20; src = r_ * (T + 1) * 40 + 2 * jj_; dst = trobs[r_] + 2 * jj_; } \
        else if (pi_ < 20 * R + 20) { src = 0; dst = -2 - (pi_ - 20 * R); }                                                              \
        else { src = 0; dst = -1; }                                                                                                      \
    }
#define EV2G_BIG_HP_ROLE2(u, src, dst, tb)                                                                                                \
    {                                                                                                                                    \
        const int pi_ = tid_l + (u) * BLOCK;                                                                                             \
        if (pi_ < 20 * R) { const int r_ = pi_ / 20, jj_ = pi_ - r_ * 20; src = r_ * (T + 1) * 40 + 2 * jj_; dst = (tb) + 2 * jj_; }       \
        else if (pi_ < 20 * R + 20) { src = 0; dst = -2 - (pi_ - 20 * R); }                                                              \
        else { src = 0; dst = -1; }                                                                                                      \
    }
    int hp_src[2], hp_dst[2];   // double offsets inside the scenario's window block (without the step term) / inside the env's observation row; -1: none
    {
        const int tid_l = tid;
        EV2G_BIG_HP_ROLE(0, hp_src[0], hp_dst[0])
        EV2G_BIG_HP_ROLE(1, hp_src[1], hp_dst[1])
    }
    const long long scnT = (long long)scn * T;
    EV2G_GP(const double) win_base = S->win_tab + (long long)scn * R * (T + 1) * 40;
    // The pointers every step uses, as per-env bases fetched ONCE: read through the parameter block where they are used (ev2g_step_v2's scheme, which
    // keeps ~70 values out of the scalar registers) each costs a scalar-cache round trip on the step's chain -- phase A alone waited for four of them.
    // One env per workgroup makes the bases uniform; what the register allocator cannot keep in SGPRs it parks in VGPR lanes (v_writelane: no latency).
    EV2G_GP(const double) b_prch = S->price_ch + scnT;
    EV2G_GP(const double) b_prdis = S->price_dis + scnT;
    EV2G_GP(const double) b_trb = S->tr_base + (long long)scn * R * T;
    EV2G_GP(const double) b_trx = S->tr_maxp + (long long)scn * R * T;
    EV2G_GP(const double) b_trn = S->tr_minp + (long long)scn * R * T;
    EV2G_GP(double) b_soc = S->soc_log + (long long)e * T * P;
    EV2G_GP(double) b_hist = S->hist + EV2G_HIST(e, 0, T, R);
    EV2G_GP(const SessRec) b_rec = S->rec;
    EV2G_GP(const double) b_lut = S->lut;
    double *const obs_e = io.obs + (long long)e * D;
    uint8_t *const mask_e = io.mask + (long long)e * P;
    // ---------------- E: the env-level results of a finished step `te` (the last wavefront but one, all 64 lanes): transformers (transformer.py:258-302), reward
    // (reward.py:34-44), histories, the observation's first two columns.  It runs one phase late -- in the battery-maths slot of the NEXT step, where this
    // wavefront has no items as a rule, or behind the loop for the launch's last step -- so that no wavefront waits for it: tsum / wsum / emg of step te
    // stay untouched until the next step's phase C.  It fetches its own transformer series (inflexible load + solar power as their precomputed sum:
    // Transformer.reset, transformer.py:262-263, the same addition done once at load).
#define EV2G_BIG_PHASE_E(te, last_)                                                                                                            \
    {                                                                                                                                         \
        const int erT_ = min(lane, R - 1) * T + (te);                                                                                         \
        const double pf_base_ = b_trb[erT_], pf_maxp_ = b_trx[erT_], pf_minp_ = b_trn[erT_];                                                  \
        double over100_ = 0.0, trp_ = 0.0;                                                                                                    \
        if (lane < R) {                                                                                                                       \
            trp_ = tsum[lane];                                                                                                                \
            double ptr_ = pf_base_;                                                                                                           \
            ptr_ += trp_;                                                                                                                     \
            const double over_ = (ptr_ > pf_maxp_ + 0.0001 || ptr_ < pf_minp_ - 0.0001) ? fabs(ptr_ - pf_maxp_) : 0.0;                        \
            b_hist[(te) * (2 + R) + 2 + lane] = over_;                                                                                        \
            if (last_) S->tr_power_now[e * R + lane] = ptr_;                                                                                  \
            over100_ = 100.0 * over_;                                                                                                         \
        }                                                                                                                                     \
        const double q_over_ = wave_sum_dpp(over100_);                                                                                        \
        const double usage_ = wave_sum_dpp(trp_);                                                                                             \
        double tot_ = 0.0;                                                                                                                    \
        if (lane < 6 && ((last_) || lane < 3)) {                                                                                              \
            const double *wp_ = (lane < 5) ? wsum + lane * NW : emg;                                                                          \
            _Pragma("unroll") for (int w_ = 0; w_ < NW; w_++) tot_ += wp_[w_];                                                                \
        }                                                                                                                                     \
        const double costs_ = readlane_f64(tot_, 0), q_sat_ = readlane_f64(tot_, 1), potn_ = readlane_f64(tot_, 2);                           \
        const double q_ech_ = readlane_f64(tot_, 3), q_edis_ = readlane_f64(tot_, 4), q_emerg_ = readlane_f64(tot_, 5);   /* (the launch's totals) */ \
        if (lane == 0) {                                                                                                                      \
            b_hist[(te) * (2 + R)] = usage_;                                                                                                  \
            if ((te) + 1 < T) b_hist[((te) + 1) * (2 + R) + 1] = potn_;                                                                       \
            const double reward_ = costs_ - q_over_ - q_sat_;   /* ProfitMax_TrPenalty_UserIncentives (reward.py:34-44) */                    \
            const double a0_ = eacc[0] + reward_, a1_ = eacc[1] + costs_;                                                                     \
            io.reward[e] = reward_;                                                                                                           \
            io.done[e] = ((te) + 1 >= T) ? 1 : 0;                                                                                             \
            obs_e[0] = (double)((te) + 1);                                                                                                    \
            obs_e[1] = usage_;                                                                                                                \
            if (last_) {  /* flush the accumulators (get_statistics reads them): a launch of this kernel ends inside the episode */            \
                double *ga_ = (double *)S->env_acc + e * 8;                                                                                   \
                ga_[0] += a0_; ga_[1] += a1_; ga_[2] += q_ech_; ga_[3] += q_edis_; ga_[4] += q_emerg_;                                        \
                eacc[0] = 0.0; eacc[1] = 0.0;                                                                                                 \
            } else { eacc[0] = a0_; eacc[1] = a1_; }                                                                                          \
        }                                                                                                                                     \
    }
    int t = t0;
    // charged / discharged energy and emergency-capacity violations feed nothing but the episode totals (get_statistics): each lane adds its ports'
    // terms up over the LAUNCH and the lanes are summed once, at its last step -- two wavefront reductions and two ballots per wavefront-step less.
    // (A launch's totals are a fixed function of its steps; launches of different lengths group the additions differently: last-bit differences
    // between a 112-step launch and 112 single-step launches in these three statistics, nowhere else.)
    double l_ech = 0.0, l_edis = 0.0;
    int n_em = 0;   // (a count per wavefront, kept in a scalar register)

    PT_DECL
    for (int kk = 0; kk < k_steps; kk++) {
        PT_MARK(7)
        asm volatile("" : "+s"(S));
        int tid_l = tid;
        asm volatile("" : "+v"(tid_l));
        const int sstep = t + 1;
        const bool last_step = (kk == k_steps - 1);
        const int eP = e * P;

        // ---------------- A: home lanes, charger level (ev_charger.py:137-186) ----------------
        int tatd[2];
        bool occ[2];
        double amps[2];
        double capb[2];   // capacity before the step: what the SoC log records (phase C writes it; the battery maths overwrites the LDS copy)
        {
            int cw[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int q = tid_l + u * BLOCK, qc = (q < P) ? q : 0;
                tatd[u] = s_tatd[qc]; cw[u] = s_cycd[qc]; capb[u] = s_cap[qc];
            }
#ifdef EV2G_PT_ASPLIT
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PT_MARK(7)
#endif
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int q = tid_l + u * BLOCK;
                const bool valid = q < P;
                const int ta = (int)(short)(tatd[u] & 0xffff), td = tatd[u] >> 16;   // (0x7fff = none: never reached by t)
                occ[u] = valid && (ta <= t) && (t <= td);
                amps[u] = 0.0;
                if (occ[u]) {
                    double a = a_next[u];
                    // one port per charger: a / sum(a) = a / a and -a / a, exactly +-1 for every finite action (ev_charger.py:143-149)
                    if (a > 1.0) a = 1.0;
                    else if (a < -1.0) a = -1.0;
                    // rnd5 (ev_charger.py:157): |a| <= 1, so rint(a * 1e5) is inside the range in which the two-FMA form of the division by 1e5 is
                    // exact (div_int_by_const, tests/test_fma_division.py): no fallback division to compile in
                    const double n5 = rint(a * 100000.0), q5 = n5 * (1.0 / 100000.0);
                    const double x = fma(fma(-q5, 100000.0, n5), 1.0 / 100000.0, q5);
                    const double *ct = ctab + ((cw[u] >> 24) & 15) * 6;
                    if (x > 0.0) { amps[u] = x * ct[0]; if (amps[u] < ct[1] - 0.01) amps[u] = 0.0; }
                    else if (x < 0.0) { amps[u] = x * ct[3]; if (amps[u] > ct[2] - 0.01) amps[u] = ct[2]; }
                }
                if (valid) { s_x[q] = amps[u]; s_y[q] = 0.0; }
            }
        }
#ifdef EV2G_PT_ASPLIT
        PT_MARK(6)
#endif
        {   // compact the ports that have battery maths to do: charging items from the front of `items`, discharging ones from its back;
            // one LDS atomic per wavefront and list (ballot + lane prefix count)
            const unsigned long long mc0 = __ballot(amps[0] > 0.0), mc1 = __ballot(amps[1] > 0.0);
            const unsigned long long md0 = __ballot(amps[0] < 0.0), md1 = __ballot(amps[1] < 0.0);
            int bch = 0, bdis = 0;
            if (lane == 0) {
                if (mc0 | mc1) bch = atomicAdd(&cnt[0], __popcll(mc0) + __popcll(mc1));
                if (md0 | md1) bdis = atomicAdd(&cnt[1], __popcll(md0) + __popcll(md1));
            }
            bch = __builtin_amdgcn_readfirstlane(bch);
            bdis = __builtin_amdgcn_readfirstlane(bdis);
#define EV2G_MBCNT(m) ((int)__builtin_amdgcn_mbcnt_hi((unsigned)((m) >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)(m), 0u)))
            if (amps[0] > 0.0) items[bch + EV2G_MBCNT(mc0)] = (unsigned short)tid_l;
            else if (amps[0] < 0.0) items[NP - 1 - (bdis + EV2G_MBCNT(md0))] = (unsigned short)tid_l;
            if (amps[1] > 0.0) items[bch + __popcll(mc0) + EV2G_MBCNT(mc1)] = (unsigned short)(tid_l + BLOCK);
            else if (amps[1] < 0.0) items[NP - 1 - (bdis + __popcll(md0) + EV2G_MBCNT(md1))] = (unsigned short)(tid_l + BLOCK);
#undef EV2G_MBCNT
        }
#ifdef EV2G_PT_ASPLIT
        PT_MARK(0)
#endif
        // ---- requests whose answers are consumed behind the battery maths: the next step's actions, this step's observation-head pairs ----
        {
            const bool more = (kk + 1 < k_steps);
            const double *an = io.actions + (long long)(more ? kk + 1 : kk) * io.a_stride + eP;
#pragma unroll
            for (int u = 0; u < 2; u++) a_next[u] = __builtin_nontemporal_load(an + (pk[u] & 0xffff));
        }
        d2_t hp[2];
        typedef double d2a8_t __attribute__((ext_vector_type(2), aligned(8)));
#pragma unroll
        for (int u = 0; u < 2; u++) {   // ONE unconditional 16-byte load per slot from a selected, always valid address (a load in a branch that merges with a
                                        // default costs a vmcnt(0) drain, ev2g_step_v2.h); a price lane reads the pair that holds its column (clamped inside the row)
#ifdef EV2G_BIG_ABL_HP0   /* ablation (wrong results): the head rows of step 0 every step -- cache-resident instead of streamed */
            EV2G_GP(const double) pw = win_base + hp_src[u];
#else
            EV2G_GP(const double) pw = win_base + (long long)sstep * 40 + hp_src[u];
#endif
            EV2G_GP(const double) pp = b_prch + min(sstep + (-2 - hp_dst[u]), T - 2);
            hp[u] = __builtin_nontemporal_load((const d2a8_t __attribute__((address_space(1))) *)((hp_dst[u] < -1) ? pp : pw));
        }
        // Departures and arrivals are known before the step (occupancy does not depend on the actions) and they are few -- a dozen of each per step
        // at 1000 ports -- but the code that handles one (an exp, a division, atomics, the arriving session's operands) is long, and a wavefront
        // executes it whenever ONE of its lanes needs it: every wavefront, nearly every step.  So the home lanes only LIST the ports that have an
        // event (per wavefront, in lane order, at a fixed place) and the last wavefront works the whole list off, once per step.
        bool ev[2];
        {
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int ta = (int)(short)(tatd[u] & 0xffff), td = tatd[u] >> 16;
                ev[u] = (occ[u] && t >= td) || ((tid_l + u * BLOCK < P) && ta == sstep);
            }
            const unsigned long long me0 = __ballot(ev[0]), me1 = __ballot(ev[1]);
            if ((me0 | me1) != 0ull) {   // (uniform)
#define EV2G_MBCNT(m) ((int)__builtin_amdgcn_mbcnt_hi((unsigned)((m) >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)(m), 0u)))
                if (ev[0]) evl[wv * 128 + EV2G_MBCNT(me0)] = (unsigned short)tid_l;
                if (ev[1]) evl[wv * 128 + __popcll(me0) + EV2G_MBCNT(me1)] = (unsigned short)(tid_l + BLOCK);
#undef EV2G_MBCNT
            }
            if (lane == 0) cntev[wv] = __popcll(me0) + __popcll(me1);
        }
#ifdef EV2G_PT_ASPLIT
        PT_MARK(4)
#else
        PT_MARK(0)
#endif
        lds_barrier();
        PT_MARK(1)

        // ---------------- B: worker lanes, battery maths on the compact list ----------------
#ifndef EV2G_BIG_NOPRIO
        __builtin_amdgcn_s_setprio(EV2G_BIG_PRIO);
#endif
        {
            const int nch = cnt[0], ndis = cnt[1];
            const int nchp = (nch + 63) & ~63;  // discharge items start on a wavefront boundary
            for (int i = tid_l; i < nchp + ndis; i += BLOCK) {
                int h = -1;
                if (i < nch) h = items[i];
                else if (i >= nchp) h = items[NP - 1 - (i - nchp)];
                if (h >= 0) {
                    const int ssh = s_ss[h];
                    const char __attribute__((address_space(1))) *rp = (const char __attribute__((address_space(1))) *)(b_rec + ssh);
                    union { SessRec r; d2_t v[8]; } rr;
                    const int r_lut = (int)s_lut[h] - 1;
                    const double cap0 = s_cap[h], prev0 = s_prev[h], tot0 = s_tot[h];
                    const int cw0 = s_cycd[h];
                    const int cyc0 = (cw0 >> 8) & 0xffff;
                    const double amps_h = s_x[h];
                    const double imax_h = ctab[((cw0 >> 24) & 15) * 6];
                    // table entry and session record are independent loads: one memory round trip, not two.  The look-up is unconditional (clamped
                    // index); whether it applies is decided where it is used.
                    const int li = (r_lut >= 0) ? ev_lut_index(r_lut, amps_h) : -1;
                    double lut_raw = b_lut[max(li, 0)];
                    EvRes o;
                    if (i < nchp) {   // (uniform)
#pragma unroll
                        for (int c = 0; c < 5; c++) rr.v[c] = *(const d2_t __attribute__((address_space(1))) *)(rp + 16 * c);
                        asm volatile("" : "+v"(lut_raw), "+v"(rr.v[0]), "+v"(rr.v[1]), "+v"(rr.v[2]), "+v"(rr.v[3]), "+v"(rr.v[4]));
                        const double lutv = (li >= 0) ? lut_raw : 1.0 / 100.0;
                        o = ev_math_charge(rr.r, lutv, amps_h, cap0, prev0, tot0, cyc0, sixty_over_dt, dt_over_60, true, r_lut >= 0);
                    } else {
#pragma unroll
                        for (int c = 3; c < 7; c++) rr.v[c] = *(const d2_t __attribute__((address_space(1))) *)(rp + 16 * c);
                        asm volatile("" : "+v"(lut_raw), "+v"(rr.v[3]), "+v"(rr.v[4]), "+v"(rr.v[5]), "+v"(rr.v[6]));
                        const double lutv = (li >= 0) ? lut_raw : 1.0 / 100.0;
                        o = ev_math_discharge(rr.r, lutv, amps_h, cap0, prev0, tot0, cyc0, dtd, r_lut >= 0, S->rdt, S->dt_fdiv != 0);
                    }
                    const bool changed = o.cycles != cyc0 || o.energy != 0.0 || o.cap != cap0 || o.prev_power != prev0;
                    s_cap[h] = o.cap;
                    s_prev[h] = o.prev_power;
                    s_tot[h] = o.tot_e;
                    s_cycd[h] = (int)((unsigned)cw0 & 0xff000003u) | (o.cycles << 8) | (changed ? 1 : 0) | (o.emerg ? 4 : 0) | ((i < nch) ? 8 : 0) | ((o.current != 0.0) ? 16 : 0);
                    s_x[h] = o.energy;
                    s_abse[h] += fabs(o.energy);
                    s_y[h] = o.energy * 60.0 / dtd;
                    // what only this lane knows -- the real current: over-current fault (ev_charger.py:203-205), the last step's reading, and whether the
                    // step was active (flag 16: the SoC log's sign, written by the home lane in phase C).  No global store on this path in an
                    // ordinary step: a store in flight here would have to drain before phase C may collect its requests
                    if (o.current - 0.0001 > imax_h) S->env_fault[e] = 1;
                    if (last_step) S->port_current[eP + h] = o.current;
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        // the last wavefront: event i of the step (wavefront-major, lane order) -> lane i; its operands are requested here, behind this wavefront's own
        // battery maths (if it has any) and a barrier ahead of their use: an arrival takes {B, cap0, potc, table id} from the session's record, a departure
        // {des, next window} from its tail entry.  Lanes without an event read session 0 (clamped addresses, unconditional loads).
        int ev_q = -1, ev_n = 0;
        double pf_ra = 0.0, pf_rb = 0.0, pf_rc = 0.0;
        int pf_lut = -1;
        if (wv == NW - 2 && kk > 0) EV2G_BIG_PHASE_E(t - 1, false)   // (uniform) the step before: its env-level results, off every other wavefront's path
        int pf_pk = 0;   // the event port's action / mask index | observation column (what its home lane keeps in `pk`)
        if (wv == NW - 1) {   // (uniform)
            int base = 0, w_of = -1, b_of = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const int c = cntev[w];
                if (lane >= base && lane < base + c) { w_of = w; b_of = base; }
                base += c;
            }
            ev_n = base;
            if (w_of >= 0) ev_q = evl[w_of * 128 + (lane - b_of)];
            const int qe = max(ev_q, 0);
            const int tde = s_tatd[qe] >> 16;
            const int sse = (ev_q >= 0) ? s_ss[qe] : 0;
            const bool ea = !(((int)(short)(s_tatd[qe] & 0xffff) <= t) && t >= tde);   // not a departure: an arrival
            pf_lut = S->ss_lut[sse];
            pf_pk = S->slot_port[qe] | (S->slot_obs[qe] << 16);
            const char *rp = (const char *)((const SessRec *)S->rec + sse), *tp = (const char *)((const SessTail *)S->tail + sse);
            pf_ra = *(const double *)(ea ? rp + offsetof(SessRec, B) : tp + offsetof(SessTail, des));
            pf_rb = *(const double *)(ea ? rp + offsetof(SessRec, cap0) : tp + offsetof(SessTail, nt_arr));   // (the window: two ints)
            pf_rc = *(const double *)(rp + offsetof(SessRec, potc));
        }
        PT_MARK(2)
        lds_barrier();
        PT_MARK(1)
        if (tid_l < 2) cnt[tid_l] = 0;
        const double pf_pch = big_sld<double>(b_prch, t), pf_pdis = big_sld<double>(b_prdis, t);
        // ONE collection point for everything requested in phase A, BEFORE this step's first global store: on gfx9-family ISAs vmcnt counts loads and
        // stores together and they retire out of order with respect to each other, so a load consumed while younger stores are pending costs a
        // full drain of those stores (ev2g_step_v2.h).  s_waitcnt vmcnt(0) expcnt(7) lgkmcnt(15); as outputs of the empty asm the registers are
        // plain values from here on (also across the loop's back edge: the next step's actions).
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("" : "+v"(a_next[0]), "+v"(a_next[1]), "+v"(hp[0]), "+v"(hp[1]));
        asm volatile("" : "+v"(pf_ra), "+v"(pf_rb), "+v"(pf_rc), "+v"(pf_lut), "+v"(pf_pk));
        // the observation head: |charge price| window and the transformers' load / PV / limit windows, copied from the scenario's tables
#pragma unroll
        for (int u = 0; u < 2; u++) {
#ifdef EV2G_BIG_ABL_NOHS  /* ablation (wrong results): no head stores */
            if (hp_dst[u] >= 0) asm volatile("" :: "v"(hp[u]));
#else
            if (hp_dst[u] >= 0) *(d2_t *)(obs_e + hp_dst[u]) = hp[u];
#endif
            else if (hp_dst[u] < -1) {   // a price column: zero past the horizon
                const int c = -2 - hp_dst[u], k = sstep + c;
                obs_e[2 + c] = (k < T) ? fabs((k > T - 2) ? hp[u].y : hp[u].x) : 0.0;
            }
        }

        // ---------------- C: home lanes: the step's port-level results; the last wavefront: the step's departures and arrivals ----------------
        double v_profit = 0.0, v_sat = 0.0, v_pot = 0.0;
        bool em[2] = {false, false};
        // what a port contributes once its occupancy after the step is known (ev2gym_env.py:452-457, rl_agent/state.py:136-151, utils.py:760-791):
        // action mask, its two observation columns, its charge-power-potential term.  `cwq`: the port's word (charger class, potential-term index)
#define EV2G_BIG_PORT_OUT(q_, pk_, occ_after_, cap_, bcap_, td_, cwq_)                                                                        \
        {                                                                                                                                    \
            mask_e[(pk_) & 0xffff] = (occ_after_) ? 1 : 0;                                                                                   \
            d2_t ov_ = {0.0, 0.0};                                                                                                           \
            double pot_ = 0.0;                                                                                                               \
            if (occ_after_) {                                                                                                                \
                const double soc_ = (cap_) / (bcap_);                                                                                        \
                ov_.x = soc_; ov_.y = (double)((td_) - sstep);                                                                               \
                if (soc_ < 1.0 && (td_) > sstep) {  /* utils.py:771 */                                                                       \
                    const unsigned pidx_ = (unsigned)(cwq_) >> 28;                                                                           \
                    if (pidx_ < 15u) pot_ = ptab[pidx_];                                                                                     \
                    else {   /* a value the loaded pool did not hold (a device refill's new car model): the state line has it */              \
                        pot_ = S->line[eP + (q_)].potc;                                                                                      \
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pot_));                                                                     \
                    }                                                                                                                        \
                }                                                                                                                            \
                const double *ct_ = ctab + (((cwq_) >> 24) & 15) * 6;   /* per-charger clamp (utils.py:779-789) */                            \
                const double mx_ = ct_[4], mn_ = ct_[5];                                                                                     \
                pot_ = (pot_ > mx_) ? mx_ : ((pot_ < mn_) ? 0.0 : pot_);                                                                     \
            }                                                                                                                                \
            *(d2_t *)(obs_e + ((unsigned)(pk_) >> 16)) = ov_;                                                                                \
            v_pot += pot_;                                                                                                                   \
        }
        {
            int cw[2];
            double capq[2], enq[2], bcq[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int q = tid_l + u * BLOCK, qc = (q < P) ? q : 0;
                cw[u] = s_cycd[qc]; capq[u] = s_cap[qc]; enq[u] = s_x[qc]; bcq[u] = s_bcap[qc];
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int q = tid_l + u * BLOCK;
                if (occ[u]) {
                    const bool item = amps[u] != 0.0;       // the battery maths ran for this port: its flags in the port's word are this step's
                    const double energy = enq[u];           // 0 for idle EVs (phase A stored amps == 0)
                    const bool active = item && (cw[u] & 16);
                    // historic_soc / active_steps (ev.py:156,162,185): capacity before the step, negated if the step was inactive
                    b_soc[t * P + q] = active ? capb[u] : -capb[u];
                    if (last_step) { S->port_energy[eP + q] = energy; if (!active) S->port_current[eP + q] = 0.0; }
                    if (energy != 0.0) {  // profit += |E| * price, by the sign of the ACTION (ev_charger.py:178,194); a charge step can return a
                                          // tiny negative energy when ceil2 left the capacity above the battery size
                        const double ae = fabs(energy);
                        if (cw[u] & 8) { v_profit += ae * pf_pch; l_ech += ae; } else { v_profit += ae * pf_pdis; l_edis += ae; }
                    }
                    em[u] = item && (cw[u] & 4);
                }
                if (q < P && !ev[u]) {   // no departure, no arrival: the port stays as it is (an event port is finished by the last wavefront, below)
                    const int td = tatd[u] >> 16;
                    EV2G_BIG_PORT_OUT(q, pk[u], occ[u], capq[u], bcq[u], td, cw[u])
                }
            }
        }
        // ---- the step's events (ev_charger.py:209-229, ev2gym_env.py:399-417, ev.py:115-136,191-214): lane i of the last wavefront takes event i ----
        if (wv == NW - 1 && ev_n > 0) {   // (uniform)
            for (int i0 = 0; i0 < ev_n; i0 += 64) {
                if (i0 > 0) {   // more than 64 events in one step (rare): the next 64, their operands fetched here
                    int base = 0, w_of = -1, b_of = 0;
                    for (int w = 0; w < NW; w++) {
                        const int c = cntev[w];
                        if (i0 + lane >= base && i0 + lane < base + c) { w_of = w; b_of = base; }
                        base += c;
                    }
                    ev_q = (w_of >= 0) ? (int)evl[w_of * 128 + (i0 + lane - b_of)] : -1;
                    const int qe = max(ev_q, 0);
                    const int w0 = s_tatd[qe];
                    const int sse = (ev_q >= 0) ? s_ss[qe] : 0;
                    const bool ea = !(((int)(short)(w0 & 0xffff) <= t) && t >= (w0 >> 16));
                    const char *rp = (const char *)((const SessRec *)S->rec + sse), *tp = (const char *)((const SessTail *)S->tail + sse);
                    pf_lut = S->ss_lut[sse];
                    pf_ra = *(const double *)(ea ? rp + offsetof(SessRec, B) : tp + offsetof(SessTail, des));
                    pf_rb = *(const double *)(ea ? rp + offsetof(SessRec, cap0) : tp + offsetof(SessTail, nt_arr));
                    pf_rc = *(const double *)(rp + offsetof(SessRec, potc));
                    pf_pk = S->slot_port[qe] | (S->slot_obs[qe] << 16);
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_ra), "+v"(pf_rb), "+v"(pf_rc), "+v"(pf_lut), "+v"(pf_pk));
                }
                if (ev_q >= 0) {
                    const int q = ev_q;
                    const int w0 = s_tatd[q];
                    int ta = (int)(short)(w0 & 0xffff), td = w0 >> 16;
                    int cwn = s_cycd[q], ssn = s_ss[q];
                    double cap = s_cap[q], bcap = s_bcap[q];
                    const int pkq = pf_pk;
                    bool departed = false;
                    if (ta <= t && t >= td) {  // departure (ev_charger.py:209-229, ev.py:191-214)
                        const int ss = ssn;
                        const double des = pf_ra;
                        const double score = (cap < des - 0.001) ? cap / des : 1.0;
                        v_sat += 100.0 * exp(-10.0 * score);   // ProfitMax_TrPenalty_UserIncentives (reward.py:41-42)
                        const int gc = e * C + (pkq & 0xffff);   // single-port chargers: the charger's index is the port's
                        __hip_atomic_fetch_add(&S->cs_served[gc], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_fetch_add(&S->cs_sat_sum[gc], score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        S->sess_final_cap[ss] = cap;
                        S->sess_abs_e[ss] = s_abse[q];
                        const int nta = __double2loint(pf_rb), ntd = __double2hiint(pf_rb);   // window of the port's next session
                        departed = true;
                        ta = (nta == EV2G_INT_MAX) ? 0x7fff : nta; td = (ntd == EV2G_INT_MAX) ? 0x7fff : ntd;
                        s_tatd[q] = (int)((unsigned)(ta & 0xffff) | ((unsigned)td << 16));
                        ssn = (nta != EV2G_INT_MAX) ? ss + 1 : -1;
                        s_ss[q] = ssn;
                        cwn = (cwn & 0x0f00001f) | 2;   // cycles = 0, no EV (the step's flags stay: the port's home lane may still be reading them)
                    }
                    if (ta == sstep) {  // arrival at the end of this step (ev2gym_env.py:399-417, ev.py:115-136)
                        double B = pf_ra, c0 = pf_rb, potc = pf_rc;
                        int lutn = pf_lut;
                        if (departed) {   // the next session arrives right behind a departure of this very step: its record was not the one requested
                            const SessRec &r = *(const SessRec *)(S->rec + ssn);
                            B = r.B; c0 = r.cap0; potc = r.potc; lutn = S->ss_lut[ssn];
                            asm volatile("s_waitcnt vmcnt(0)" : "+v"(B), "+v"(c0), "+v"(potc), "+v"(lutn));
                        }
                        s_lut[q] = (unsigned short)(lutn + 1);
                        unsigned pidx = 15u;
                        for (int i = 14; i >= 0; i--) if (ptab[i] == potc) pidx = (unsigned)i;
                        cap = c0; bcap = B;
                        s_cap[q] = cap; s_tot[q] = 0.0; s_prev[q] = 0.0; s_bcap[q] = B; s_abse[q] = 0.0;
                        cwn = (int)(((unsigned)cwn & 0x0f00001fu) | 1u | (pidx << 28));   // cycles = 0
                        S->line[eP + q].bcap = B;
                        S->line[eP + q].potc = potc;
                        S->port_energy[eP + q] = 0.0;
                        S->port_current[eP + q] = 0.0;
                    }
                    s_cycd[q] = cwn;
                    const bool occ_after = (ta <= sstep) && (sstep <= td);
                    EV2G_BIG_PORT_OUT(q, pkq, occ_after, cap, bcap, td, cwn)
                }
            }
        }
#undef EV2G_BIG_PORT_OUT
        // per-wavefront partial sums of the step's env-level quantities, from registers (fixed tree: lane pair, DPP butterflies, one partial per wavefront)
        {
            const double w_profit = wave_sum_dpp(v_profit), w_pot = wave_sum_dpp(v_pot);
            double w_sat = 0.0;
            if (__ballot(v_sat != 0.0) != 0ull) w_sat = wave_sum_dpp(v_sat);   // (uniform; departures are rare and only the last wavefront has them)
            if (lane == 0) { wsum[0 * NW + wv] = w_profit; wsum[1 * NW + wv] = w_sat; wsum[2 * NW + wv] = w_pot; }
            n_em += __popcll(__ballot(em[0])) + __popcll(__ballot(em[1]));
            if (last_step) {   // (uniform) the launch's energy totals and violation count
                const double w_ech = wave_sum_dpp(l_ech), w_edis = wave_sum_dpp(l_edis);
                if (lane == 0) { wsum[3 * NW + wv] = w_ech; wsum[4 * NW + wv] = w_edis; emg[wv] = (double)n_em; }
            }
        }
#ifndef EV2G_PT_ASPLIT
        PT_MARK(3)
#endif
        // ---------------- D: power per transformer: 8 lanes per segment, DPP butterfly (the tree of ev2g_step_v2's one-env scheme) ----------------
        if (tid_l < R * 8) {
            const int r = tid_l >> 3, j = tid_l & 7;
            const int b = seg[r + 1];
            double acc = 0.0;
            for (int i = seg[r] + j; i < b; i += 8) acc += s_y[i];
            acc += xor1_f64(acc);
            acc += xor2_f64(acc);
            acc += xor4_f64(acc);
            if (j == 0) tsum[r] = acc;
        }
#ifdef EV2G_PT_ASPLIT
        PT_MARK(3)
#else
        PT_MARK(4)
#endif
        lds_barrier();
        PT_MARK(1)

        PT_MARK(5)
        PT_STEP_END(false)
        t += 1;
    }
    PT_FLUSH
    if (wv == NW - 2 && k_steps > 0) EV2G_BIG_PHASE_E(t - 1, true)   // the launch's last step
#undef EV2G_BIG_PHASE_E
    // ---- write the LDS-resident port state back ----
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int q = tid + u * BLOCK;
        if (q < P) {
            const int cw = s_cycd[q];
            EV2G_GP(PortLine) ln = S->line + (e * P + q);
            if (cw & 2) { const int w = s_tatd[q]; ln->ta = big_unpack16(w & 0xffff); ln->td = big_unpack16((w >> 16) & 0xffff); }
            if (cw & 3) {   // the line carries the efficiency-table id of the attached EV next to its cycle count (the fast path reads it from there)
                const int ssd = s_ss[q];
                ln->ss = ssd; ln->cyc_lut = ev2g_line_pack((cw >> 8) & 0xffff, ssd >= 0 ? S->ss_lut[ssd] : -1);
            }
            if (cw & 1) { ln->cap = s_cap[q]; ln->tot = s_tot[q]; ln->prev = s_prev[q]; ln->abse = s_abse[q]; }
        }
    }
}
