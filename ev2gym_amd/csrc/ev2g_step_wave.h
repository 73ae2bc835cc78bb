// ev2g_step_wave.h -- fast path of the step kernel for the common shape: P <= 64 ports per env, one transformer,
// single-port chargers (BASELINE cfg2 / cfg3 / cfg5 and the reference's shipped YAML files).
//
// Same phases and the same arithmetic as ev2g_step_v2 (ev2g_step_v2.h), with two structural differences:
//   * WAVE-ALIGNED envs: a wavefront owns EPW = 64 / P whole envs (lanes [0, EPW*P)).  Everything that concerns one
//     env after the battery maths -- departures / arrivals, observation columns, the per-env reduction, transformer
//     overload, reward, observation head -- then happens inside ONE wavefront, ordered by s_waitcnt only.  Only the
//     compaction hand-off (home lanes -> worker lanes -> home lanes) crosses wavefronts, so a step has TWO workgroup
//     barriers instead of five or six.
//   * specialised at compile time on the fused (state, reward) plugin pair, so the plugin branches, their loads and
//     their registers disappear.
// The reduction is still LDS-staged and fixed-order (8 quantities x 8 lanes, two chains, 3 xor steps), i.e.
// bit-reproducible, and identical in value to the generic kernel's (same tree).
#pragma once
#include <type_traits>
#include "ev2g_step_v2.h"
#include "ev2g_mlp.h"

#ifndef EV2G_WAVE_BLOCK
#define EV2G_WAVE_BLOCK 256
#endif

__host__ __device__ inline size_t ev2g_wave_lds_bytes(int envs_per_group, int block = EV2G_WAVE_BLOCK) {
    const size_t NS = (size_t)block;
    return sizeof(double) * (EV2G_NQ * (NS + 8) + 7 * NS + 7 * (size_t)envs_per_group + 4 * 64) + sizeof(int) * (6 * NS + 8);
}
// the fused actor + step instantiation (ACT, below): 16 envs and 16 wavefronts per workgroup, plus the policy's input rows (bf16) and its actions (float) in LDS
#define EV2G_FUSED_BLOCK 1024
#define EV2G_FUSED_SX 200     // MlpS16<6, ..>::SX: bf16 elements per observation row in LDS
#ifndef EV2G_FUSED_RING
#define EV2G_FUSED_RING 10    // weight fragments a wavefront keeps in flight (ev2g_mlp3_inline)
#endif
#ifndef EV2G_FUSED_RING2
#define EV2G_FUSED_RING2 7    // the same with two envs per wavefront (AE = 2): the second row block's accumulators and operand take 12 registers
#endif

#define EV2G_FUSED_SXF 196    // the float32 policy (NWF = 2): floats per observation row in LDS (6 k-steps of 32 + 16 bytes against bank conflicts)
#ifndef EV2G_FUSED_RINGF
#define EV2G_FUSED_RINGF 4    // ... and its weight fragments in flight per wavefront (two terms per k-step; 6 / 8 measure the same as 4 and leave no registers for EV2G_F32_GROUP = 2)
#endif
__host__ __device__ inline size_t ev2g_fused_lds_bytes(int envs_per_wave = 1, int weight_terms = 1) {
    if (weight_terms > 1)   // float32 input rows (H2's third copy lies over them between layers 1 and 3), biases from global memory: 163 488 of the 163 840 bytes
        return ev2g_wave_lds_bytes(EV2G_FUSED_BLOCK / 64 * envs_per_wave, EV2G_FUSED_BLOCK) + (size_t)16 * EV2G_FUSED_SXF * 4;
    return ev2g_wave_lds_bytes(EV2G_FUSED_BLOCK / 64 * envs_per_wave, EV2G_FUSED_BLOCK) + (size_t)16 * EV2G_FUSED_SX * 2 + (size_t)(25 + 19 + 4) * 16 * 4;   // + input rows, biases
}
// What the fused instantiation needs besides the step's own arguments: the policy, and the observation rows its first forward reads.
// StepIO then carries the OUTPUT blocks: obs32 = the rows the steps write (row of the launch's first step; o_stride floats between steps, 0: one row
// overwritten, and then obs0 == obs32), act32 = the action rows the policy writes (a_stride floats between steps), reward / done / mask with their strides.
struct FusedArgs {
    MlpDev m;
    const float *obs0;
};

// Global accesses as  uniform base (an SGPR pair) + 32-bit unsigned BYTE offset (one VGPR):  the
// `global_load/store v, v_off, s[base:base+1]` form.  Indexing a pointer with a (sign-extended) int instead makes every
// access carry a 64-bit VALU address computation (v_ashrrev + v_lshl_add_u64, an address VGPR pair each).  The host
// selects this kernel only when every array it addresses is smaller than 4 GiB (ev2g_host.hip).
typedef const char __attribute__((address_space(1))) *gcptr;
typedef char __attribute__((address_space(1))) *gptr;
template <class T, class B> __device__ __forceinline__ T ldg32(B base, unsigned boff) {
    return *(const T __attribute__((address_space(1))) *)((gcptr)base + boff);
}
// streaming variant: read once per step and never again by this CU -- keep it from displacing the session records and
// efficiency tables (re-read every step) in the vector L1
template <class T, class B> __device__ __forceinline__ T ldg32_nt(B base, unsigned boff) {
    return __builtin_nontemporal_load((const T __attribute__((address_space(1))) *)((gcptr)base + boff));
}
template <class T, class B> __device__ __forceinline__ void stg32(B base, unsigned boff, T v) {
    *(T __attribute__((address_space(1))) *)((gptr)base + boff) = v;
}
template <class T, class B> __device__ __forceinline__ void stg32_nt(B base, unsigned boff, T v) {   // streaming store: written once, read by another kernel if at all
    __builtin_nontemporal_store(v, (T __attribute__((address_space(1))) *)((gptr)base + boff));
}
typedef double d2v __attribute__((ext_vector_type(2)));   // builtin vectors: loadable from any address space
typedef double d2v_a8 __attribute__((ext_vector_type(2), aligned(8)));   // the same at an 8-byte aligned address (global_load_dwordx4 needs dword alignment only)
typedef int i2v __attribute__((ext_vector_type(2)));
typedef int i4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
// The battery-maths part of the record in one memory round trip: the chunks this wavefront's kind of step reads (charging 0..4, discharging
// 3..6; uniform -- the two lists start on wavefront boundaries), issued back to back, then pinned by an empty asm so that the compiler cannot
// sink the ones a later branch does not need behind that branch (it otherwise loads the gate field first and the rest only after testing
// it: two dependent round trips).
template <class B> __device__ __forceinline__ SessRec ldg32_rec_charge(B base, unsigned boff) {
    union { SessRec r; d2v v[sizeof(SessRec) / 16]; } u;
    static_assert(sizeof(SessRec) == 128 && offsetof(SessRec, rB) == 48 && offsetof(SessRec, rv) == 64 && offsetof(SessRec, minB) == 80 && offsetof(SessRec, cap0) == 112,
                  "charging reads chunks 0..4, discharging chunks 3..6");
#pragma unroll
    for (int i = 0; i < 5; i++) u.v[i] = ldg32<d2v>(base, boff + 16u * i);
    asm volatile("" : "+v"(u.v[0]), "+v"(u.v[1]), "+v"(u.v[2]), "+v"(u.v[3]), "+v"(u.v[4]));
    return u.r;
}
template <class B> __device__ __forceinline__ SessRec ldg32_rec_discharge(B base, unsigned boff) {
    union { SessRec r; d2v v[sizeof(SessRec) / 16]; } u;
#pragma unroll
    for (int i = 3; i < 7; i++) u.v[i] = ldg32<d2v>(base, boff + 16u * i);
    asm volatile("" : "+v"(u.v[3]), "+v"(u.v[4]), "+v"(u.v[5]), "+v"(u.v[6]));
    return u.r;
}

// Round 5 (full kernels): the same operands from the dictionary entry (ClsRec, ev2g_device.h) -- four chunks for a charging step, three for a
// discharging one -- with the per-session ones (transition_soc, the efficiencies) supplied by the caller from the port's LDS state.
template <class B> __device__ __forceinline__ SessRec ldg32_cls_charge(B base, unsigned boff) {
    static_assert(sizeof(ClsRec) == 128 && offsetof(ClsRec, gate_ch) == 16 && offsetof(ClsRec, rB) == 32 && offsetof(ClsRec, rv) == 48 && offsetof(ClsRec, v_d) == 64 &&
                  offsetof(ClsRec, gate_dis) == 80 && offsetof(ClsRec, emerg) == 96, "ClsRec layout");
    d2v c[4];
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = ldg32<d2v>(base, boff + 16u * i);
    asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
    SessRec r;
    r.pacmax = c[0].x; r.tsm = c[0].y; r.gate_ch = c[1].x; r.B = c[1].y; r.rB = c[2].x; r.v = c[2].y; r.rv = c[3].x;
    return r;
}
template <class B> __device__ __forceinline__ SessRec ldg32_cls_discharge(B base, unsigned boff) {
    d2v c[3];
#pragma unroll
    for (int i = 0; i < 3; i++) c[i] = ldg32<d2v>(base, boff + 64u + 16u * i);
    asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]));
    SessRec r;
    r.v = c[0].x; r.rv = c[0].y; r.gate_dis = c[1].x; r.minB = c[1].y; r.emerg = c[2].x; r.pdismax = c[2].y;
    return r;
}

// What the launch prologue needs, BY VALUE: with it the first data loads depend on one fetch (the kernarg segment)
// instead of two (kernarg -> parameter block).  A launch starts with cold caches, so every dependent fetch in the
// prologue is a full memory round trip -- paid per step by single-step launches.
struct WaveArgs {
    int P, T, E, D, M;
    char *slab_port; unsigned long long slab_port_slice;
    double *hist;            // [E, T, 3] usage | potential | overload per (env, step) (one transformer)
    double *env_acc;
    const double *cs_pack;   // [C][6] imax, |dmax|, imin, dmin, max power, min power
    char *lines;             // [E*P] PortLine
    const double *step_tab;  // [M, T, 8]; slots 6, 7: the scenario's occupancy / arrival masks of the step
    char *port_dyn;          // [E*P] PortDyn: transition_soc, efficiencies, table id and dictionary entry of the attached EV
    int dict;                // DevScn::dict: V2P::cls_rec is a dictionary (entry = bits 20..31 of the port's LDS word); 0: one ClsRec per session
    int epw, es;             // envs per wavefront (<= 64 / P) and the lane stride between them (>= P; P: packed, 64 / epw: aligned to lane rows)
};

// IO32: the actions are float32 (StepIO::act32) -- the policy-network interface; float32 observations (StepIO::obs32) are
// written whenever that pointer is set, in either instantiation.
// RK: 0..2 = the reward of the three shipped configs, compiled in; 3 = any other fused reward, selected at run time by
// V2P::reward_kind (ev2g_reward / ev2g_departure_term in ev2g_device.h).
// FULL (with IO32: the same for the policy network's hand-over -- float32 actions in, the float32 observation out INSTEAD of the float64
// one; what ev2g_rollout launches between two actor forwards):
// the launch writes all four float64 outputs (observation, reward, done, mask) with step stride 0 (each step overwrites the last:
// what a loop that consumes them step by step, or a benchmark, passes), takes float64 actions, uses no extras (fused cost, float32
// observations, charger histories) and contains no in-launch reset.  The null checks, the extras' parameter fetches, the outputs'
// per-step pointer arithmetic and the reset branch -- dozens of scalar instructions a step, issued by every wavefront -- are compiled
// out: measured 4.53 -> 4.28 us/step at cfg2, 5.40 -> 5.08 at cfg3 (the kernel's time is its instruction count, SURVEY par.8d / DESIGN par.3).
// FULLK = 2 additionally knows at compile time that the SoC log is on (EV2G_FLAG_LOG_SOC: the Python surface's and the benchmark's default)
// and that the env is wide enough for one observation-head column pair per lane (P >= 30 for the 60-column head, P >= 10 for the
// 20-column one): the second pair's prefetch and stores and the tail loop of narrow envs go too.
// FULLK = 3 (round 5) is FULLK = 2 for outputs with STEP STRIDES -- the [K, E, *] blocks of a persistent launch whose every observation, reward, done
// flag and mask is kept (generate_trajectories.py:69-83 style use; a replay block): four running pointers advanced with scalar adds per step,
// everything else as compiled-out as in 2.
// ACT (round 5; BLOCK = 1024, FULLK = 2, IO32): the policy network (FusedArgs::m, obs -> 400 -> 300 -> ports on the bf16 matrix cores) is evaluated INSIDE
// the launch, between the steps, by the workgroup that steps the 16 envs whose rows it reads: one launch per rollout segment of k steps (ev2g_collect /
// ev2g_rollout) instead of two per step -- no kernel boundary, so no cold start of either kernel, the port state stays in LDS across the segment
// like in any persistent launch, observations reach the policy and actions reach the step through LDS, and the weights are streamed once per CU and step.
// The outputs (observation / action / reward / done / mask rows of every step) go to the caller's blocks through running pointers.
// AE (round 6; ACT with PublicPST, P <= 32): TWO envs per wavefront in the fused instantiation (lanes 0..31 / 32..63), 32 policy rows per workgroup -- every weight
// fragment feeds two MFMAs, the weight stream per env halves, and 8192 envs are one round of 256 workgroups instead of two rounds of 512.
// NWF (round 6, last session; ACT with a head-table state, AE = 1): the FLOAT32 policy (EV2G_MLP_F32: two bf16 terms per weight, three per activation)
// inside the launch -- ev2g_mlp3_inline_f32 (ev2g_mlp.h): the input rows stay float32 in LDS, the hidden activations' three copies fill the staging rows.
template <int SK, int RK, bool IO32, int FULLK = 0, int BLOCK = EV2G_WAVE_BLOCK, bool ACT = false, int AE = 1, int NWF = 1>
__global__ void __launch_bounds__(BLOCK, 4) ev2g_step_wave(const V2P *__restrict__ params, StepIO io, int t0,
                                                           int k_steps, int auto_reset, WaveArgs wa, FusedArgs fa) {
    extern __shared__ double lds[];
    static_assert(!ACT || (IO32 && FULLK == 2 && BLOCK == EV2G_FUSED_BLOCK), "the fused actor + step instantiation");
    static_assert(AE == 1 || (ACT && SK == 1 && AE == 2), "two envs per wavefront in the fused instantiation: PublicPST only");
    static_assert(NWF == 1 || (ACT && AE == 1 && NWF == 2), "the float32 policy inside the launch: one env per wavefront (16 policy rows per workgroup)");
    constexpr bool FULL = FULLK >= 1, WIDE = FULLK >= 2, STR = FULLK >= 3 || ACT;
#ifdef EV2G_STR_NT_OFF   // (A/B: the kept rows as ordinary stores)
    constexpr bool STR_NT = false;
#else
    constexpr bool STR_NT = FULLK >= 3;
#endif   // the kept observation rows (0.6 GB per cfg2 launch) as streaming stores: they should not displace the state lines in L2 (-2 %, profiles/r05_ab_strided_nt.txt)
    constexpr bool F64 = FULL && !IO32, F32 = FULL && IO32;   // full with float64 actions in / observations out, or with the float32 hand-over
#if defined(EV2G_PHASE_TIMING) && defined(EV2G_PT_OUTER)
    const unsigned long long pt_k0 = __builtin_readcyclecounter();   // slot 7 := prologue, slot 6 := epilogue (tools/phase_timing.py --outer)
#endif
    typedef const V2P __attribute__((address_space(4))) *ParamPtr;
    ParamPtr S = (ParamPtr)(unsigned long long)params;
    constexpr int NS = BLOCK;
    constexpr int RS = NS + 8;   // stage row stride: +16 banks per row, so the 8 rows one reduction read touches spread over all banks
    const int P = wa.P, T = wa.T, E = wa.E, D = wa.D, M = wa.M;
    int off = io.scn_off;   // scenario-pool window: env e runs scenario (e + off) mod M
    // state slabs (ev2g_device.h): every [E*P] array is slabP + k * PS8, the three [T,E] histories slabH + k * HS8, the
    // two per-session result arrays slabS + k * SS8 -- scalar adds on three base pointers instead of one pointer fetch
    // from the parameter block per array and use
    const gptr slabP = (gptr)wa.slab_port, hist = (gptr)wa.hist, slabS = (gptr)S->slab_sess;
    const unsigned long long PS8 = wa.slab_port_slice, SS8 = S->sess_slice;
    const gptr env_acc = (gptr)wa.env_acc;
#define PA(k) (slabP + PS8 * (unsigned long long)(k))
    // envs per wavefront.  The fused instantiation gives EVERY env a wavefront of its own, whatever its width (a policy row is an env: 16 rows per
    // workgroup; the step's time is a chain of latencies, not lanes): the lanes behind the env's last port only copy observation-head pairs
    const int EPW = ACT ? AE : wa.epw;
    const int ES = ACT ? (AE == 1 ? P : 32) : wa.es;      // lanes from one env of the wavefront to the next
    const int G = (BLOCK / 64) * EPW;    // envs per workgroup
    int grp;
    {   // XCD-aware mapping: workgroup b runs on XCD b % 8; give each XCD a contiguous range of env groups
        const int nb = gridDim.x, b = blockIdx.x, per = nb >> 3;
        grp = (nb & 7) == 0 ? (b & 7) * per + (b >> 3) : b;
    }
    const int e0 = grp * G;
    double *stage = lds;                                   // [NQ][RS] per-port step results, by home index (= tid)
    double *s_cap = stage + (size_t)EV2G_NQ * RS;
    double *s_tot = s_cap + NS, *s_prev = s_tot + NS, *s_bcap = s_prev + NS, *s_potc = s_bcap + NS;
    double *s_amps = s_potc + NS, *s_abse = s_amps + NS;
    double *eacc = s_abse + NS;                            // [G][7] episode accumulators + charge_power_potential[t], [t-1], per env
    double *s_cst = eacc + 7 * G;                          // [4][64] per-charger gates and clamps (rarely changing operands
                                                           // kept out of the register file): imin-0.01, dmin, max power, min power
    int *s_ta = (int *)(s_cst + 4 * 64);
    int *s_td = s_ta + NS, *s_ss = s_td + NS, *s_cyc = s_ss + NS, *s_dirty = s_cyc + NS, *items = s_dirty + NS;
    int *cnt = items + NS;  // cnt[2*(kk&1) + {0 charge, 1 discharge}]
    // Full kernels keep window, battery size and potential term in registers (below): the LDS behind s_bcap / s_potc / s_ta + s_td holds the attached
    // EV's transition_soc and efficiencies instead -- what the battery maths used to fetch from the session's own record -- and bits 20..31 of
    // s_dirty its dictionary entry (bits 8..19 the efficiency-table id + 1: a full kernel needs n_lut <= 4094, checked by the host)
    double *s_ts = s_bcap, *s_etac = s_potc, *s_etad = (double *)s_ta;
    constexpr int LUTMASK = FULL ? 0xfff : 0xffff;
    // ACT: the policy's input rows (bf16; env w of the workgroup = wavefront w = row w) and its actions behind the step's own LDS; its hidden
    // activations in the staging rows, which are dead between the end of a step and the next phase A (idle lanes re-zero their slots afterwards)
    // (round 6: PublicPST -- 3 + 3 P <= 63 inputs, P <= 20 outputs -- rides the 64 -> 400 -> 300 -> 32 packing: a third of layer 1's weight stream)
    typedef typename std::conditional<SK == 1, MlpS16<2, 25, 19, 2, 1, 4, AE>, MlpS16<6, 25, 19, 4, 1, 4, AE>>::type MC;
    constexpr int FSX = MC::SX;          // bf16 elements per observation row in LDS
    constexpr int FKS1 = (SK == 1) ? 2 : 6, FNT3 = (SK == 1) ? 2 : 4;
    uint16_t *bufX = (uint16_t *)(cnt + 8);
    float *lbias = (float *)(bufX + 16 * EV2G_FUSED_SX);   // the three bias vectors (MC::NB floats), staged once per launch
    // the actions of env w: the first 64 floats of wavefront w's own slice of s_amps (dead between a step's phase C and the next phase A; the
    // wavefront reads its actions -- one instruction, all lanes -- before it writes the amps over them)
    float *act_lds = (float *)s_amps;
    uint16_t *bufH1 = (uint16_t *)stage, *bufH2 = bufH1 + 16 * AE * MC::SH1;
    // NWF = 2: float32 input rows in bufX's place (no staged biases); H1's three copies, then two of H2's in the staging rows, H2's third over the input rows
    constexpr int FSXF = EV2G_FUSED_SXF;
    float *bufXf = (float *)(cnt + 8);
    static_assert(NWF == 1 || (3 * 16 * MC::SH1 * 2 <= 5 * RS * 8 && EV2G_NQ == 8 && FKS1 * 32 <= 256), "the input rows' bf16 terms live in staging rows 5..7, behind H1's three copies");
    static_assert(NWF == 1 || (FKS1 * 32 + 4 <= FSXF && (3 * MC::SH1 + 2 * MC::SH2) * 16 * 2 <= EV2G_NQ * (EV2G_FUSED_BLOCK + 8) * 8 && MC::SH2 * 16 * 2 <= 16 * FSXF * 4), "float32 policy buffers");
    static_assert(AE * MC::SX <= EV2G_FUSED_SX && MC::NB <= (25 + 19 + 4) * 16 && 16 * AE * (MC::SH1 + MC::SH2) * 2 <= EV2G_NQ * (EV2G_FUSED_BLOCK + 8) * 8, "policy buffers");
    constexpr int ACT_AS = (AE == 1) ? 128 : 64;   // floats between two envs' action rows in s_amps (a wavefront's slice holds its envs' rows)
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const bool log_soc = WIDE ? true : (S->soc_log != nullptr);
    const bool log_cs = FULL ? false : (S->cs_profits != nullptr);   // EV2G_FLAG_LOG_CS_HISTORY: charger-level accumulators and histories (ev2gym_env.py:533-535)
    const double dtd = (double)S->dt, sixty_over_dt = S->sixty_over_dt, dt_over_60 = S->dt_over_60;
    const bool pow2_dt = S->pow2_dt != 0;

    // ---- home lane set-up ----
    const int elw = lane / ES;           // env inside the wavefront
    const int q = (ACT && AE == 1) ? lane : lane - elw * ES;       // port slot (== reference port: one transformer, single-port chargers)
    const int e = (ACT && AE == 1) ? e0 + wv : e0 + wv * EPW + elw;
    const bool valid = (elw < EPW) && (q < P) && (e < E);
    const bool hcopy = ACT && AE == 1 && !valid && e < E && lane < 32;   // (ACT, envs narrower than their head pairs: lanes P .. NPAIR-1 copy the pairs the ports cannot)
    const int g = valid ? e * P + q : 0;
    const int ocol = (SK == 1) ? 3 + 3 * q : (SK == 0 ? 62 + 2 * q : 22 + 2 * q);
    const int cs = valid ? q : 0;
    int t = t0;
    const bool head = valid && q == 0;   // one lane per env: env-level scalars
    const int elg = wv * EPW + elw;      // env inside the workgroup
    const int arow = (AE == 1) ? wv : min(elg, 16 * AE - 1);   // (ACT) this lane's env as a row of the policy's buffers
    // the policy's copy of an observation column pair (even column): packed bf16, or the float32 values themselves
    auto xput2 = [&](int col, float a, float b) __attribute__((always_inline)) {
        if (NWF > 1) *(float2 *)(bufXf + arow * FSXF + col) = make_float2(a, b);
        else *(uint32_t *)(bufX + arow * FSX + col) = ev2g_pack_bf16(a, b);
    };
    // ---- launch prologue.  A single-step launch (the RL loop with a policy between steps) pays it every step, with cold caches.
    // Round trip 1: what does not depend on data -- charger constants, the first action, the env accumulators and, for a single-step launch,
    // the scenario's occupancy / arrival masks of this step (step table slots 6, 7: occupancy does not depend on the actions); for a longer
    // launch the head chunk of every port's state line instead.  Round trip 2: the state lines (PortLine, 64 bytes = one sector) of the
    // ports that hold an EV -- in a single-step launch also the head chunk, and only for ports that hold an EV or receive one this step.
    double c_imax, c_dmaxabs, a_next;
    // FULL kernels keep what only the port's own lane touches -- its occupancy window, the attached EV's battery size and potential
    // term -- in registers instead of LDS (no in-launch reset rewrites them from outside): six LDS instructions a step less
    int r_ta = EV2G_INT_MAX, r_td = -1;
    double r_bcap = 1.0, r_potc = 0.0;
    double r_rb = 1.0;   // RN(1 / battery size) of the attached EV: the observation's cap / B goes through it (ev2g_fdiv2) in the full kernels
    const unsigned l64 = (unsigned)g * (unsigned)sizeof(PortLine);   // this port's state line
    static_assert(sizeof(PortLine) == 64 && offsetof(PortLine, cap) == 16 && offsetof(PortLine, prev) == 32 && offsetof(PortLine, bcap) == 48, "PortLine layout");
    {
        const unsigned c8 = (unsigned)cs * 8u, cp8 = (unsigned)min(tid, P - 1) * 8u;
        const unsigned ec = (unsigned)(valid ? e : e0);
        const bool by_mask = (k_steps == 1) && (t < T);   // (uniform)
        i4v hd = {EV2G_INT_MAX, -1, -1, 0};                // {t_arr, t_dep, session, cycles | table id + 1}
        d2v mk = {0.0, 0.0};
        if (by_mask) mk = ldg32<d2v>(wa.step_tab, (unsigned)(ev2g_scn((int)ec, off, M) * T + t) * 64u + 48u);
        else hd = ldg32<i4v>(wa.lines, l64);
        const d2v k_max = ldg32<d2v>(wa.cs_pack, c8 * 6u);   // (imax, |dmax|) of this lane's charger
        d2v k_min = {0.0, 0.0}, k_pow = {0.0, 0.0};            // gates and clamps, staged in LDS by the first P lanes of the workgroup
        if (tid < 64) { k_min = ldg32<d2v>(wa.cs_pack, cp8 * 6u + 16u); k_pow = ldg32<d2v>(wa.cs_pack, cp8 * 6u + 32u); }   // (wavefront 0 only)
        if (ACT) a_next = 0.0;   // (the policy runs inside the launch: its actions arrive through LDS)
        else
        a_next = IO32 ? (double)ldg32<float>(io.act32 + (long long)io.step0 * io.a_stride, (unsigned)(valid ? g : e0 * P) * 4u)
                      : ldg32<double>(io.actions, (unsigned)(valid ? g : e0 * P) * 8u);
        double l_pot = ldg32<double>(hist, ((ec * (unsigned)T + (unsigned)min(t, T - 1)) * 3u + 1u) * 8u);
        double l_pot2 = 0.0;   // charge_power_potential[t-1]: only SquaredTrackingErrorRewardWithPenalty (a run-time reward) reads it
        if (RK == 3) l_pot2 = ldg32<double>(hist, ((ec * (unsigned)T + (unsigned)min(max(t - 1, 0), T - 1)) * 3u + 1u) * 8u);
        d2v acc01 = ldg32<d2v>(env_acc, ec * 64u), acc23 = ldg32<d2v>(env_acc, ec * 64u + 16u);
        double acc4 = ldg32<double>(env_acc, ec * 64u + 32u);
        d2v k_max_w = k_max;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(hd), "+v"(mk), "+v"(k_max_w), "+v"(k_min), "+v"(k_pow),
                     "+v"(a_next), "+v"(l_pot), "+v"(l_pot2), "+v"(acc01), "+v"(acc23), "+v"(acc4));
        c_imax = k_max_w.x; c_dmaxabs = k_max_w.y;
        if (tid < P) {
            s_cst[0 * 64 + tid] = k_min.x - 0.01; s_cst[1 * 64 + tid] = k_min.y;
            s_cst[2 * 64 + tid] = k_pow.x; s_cst[3 * 64 + tid] = k_pow.y;
        }
        if (valid) {
            bool body;   // an EV is attached in the launch's first step: the rest of the line is needed
            if (by_mask) {
                const unsigned long long m_occ = (unsigned long long)__double_as_longlong(mk.x), m_arr = (unsigned long long)__double_as_longlong(mk.y);
                body = (m_occ >> q) & 1ull;
                if (body || ((m_arr >> q) & 1ull)) hd = ldg32<i4v>(wa.lines, l64);   // (a port without either keeps the defaults: no event can touch it this step)
            } else body = (hd.x <= t) && (t <= hd.y);
            d2v b1 = {0.0, 0.0}, b2 = {0.0, 0.0}, b3 = {1.0, 0.0};
            d2v pd0 = {1.0, 1.0}; i4v pd1 = {0, 0, -1, 0};
            static_assert(sizeof(PortDyn) == 32 && offsetof(PortDyn, eta_dis) == 16 && offsetof(PortDyn, lut) == 24 && offsetof(PortDyn, cls) == 28, "PortDyn layout");
            if (body) {
                b1 = ldg32<d2v>(wa.lines, l64 + 16u); b2 = ldg32<d2v>(wa.lines, l64 + 32u); b3 = ldg32<d2v>(wa.lines, l64 + 48u);
                if (FULL) { pd0 = ldg32<d2v>(wa.port_dyn, (unsigned)g * 32u); pd1 = ldg32<i4v>(wa.port_dyn, (unsigned)g * 32u + 16u); }
            }
            r_ta = hd.x; r_td = hd.y;
            if (!FULL) { s_ta[tid] = hd.x; s_td[tid] = hd.y; }
            s_ss[tid] = hd.z; s_cyc[tid] = ev2g_line_cycles(hd.w);
            // s_dirty: bits 0,1 = what the epilogue must write back; bits 8..23 = 1 + efficiency-table id of the attached EV,
            // so that the battery maths can issue the table look-up together with (not behind) the session-record load
            s_dirty[tid] = (int)(((unsigned)hd.w >> 16) << 8 | ((FULL && wa.dict) ? (unsigned)pd1.w << 20 : 0u));
            if (FULL) { s_ts[tid] = pd0.x; s_etac[tid] = pd0.y; s_etad[tid] = __hiloint2double(pd1.y, pd1.x); }
            s_cap[tid] = b1.x; s_tot[tid] = b1.y; s_prev[tid] = b2.x; s_abse[tid] = log_soc ? b2.y : 0.0;
            r_bcap = b3.x; r_potc = b3.y;
            if (FULL) { if (body) r_rb = 1.0 / r_bcap; }
            else { s_bcap[tid] = r_bcap; s_potc[tid] = r_potc; }
        }
        if (head) {   // episode accumulators (continued from global memory) and charge_power_potential[t], in LDS
            double *ea = eacc + elg * 7;
            ea[0] = acc01.x; ea[1] = acc01.y; ea[2] = acc23.x; ea[3] = acc23.y; ea[4] = acc4;
            ea[5] = (t < T) ? l_pot : 0.0;
            ea[6] = (t > 0 && t <= T) ? l_pot2 : 0.0;
        }
    }
    if (tid < 4) cnt[tid] = 0;
    for (int k = 0; k < EV2G_NQ; k++) stage[k * RS + tid] = 0.0;
    if (ACT) {   // the observation the first forward reads (the reset observation, or the last step's of an earlier segment): this wavefront's env row -> bf16
        const bool env_ok = (e0 + arow) < E;   // (one env per wavefront; AE = 2: one per half)
        const float *xr = fa.obs0 + (long long)(env_ok ? e0 + arow : e0) * D;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int c = (AE == 1) ? 2 * lane + 128 * j : 2 * (lane & 31) + 64 * j;
            float2 v = make_float2(0.f, 0.f);
            if (SK == 1) {   // (D = 3 + 3 P may be odd: rows are 4-byte aligned only)
                if (env_ok && c < D) v.x = xr[c];
                if (env_ok && c + 1 < D) v.y = xr[c + 1];
            } else {
                if (env_ok && c + 1 < D) v = *(const float2 *)(xr + c);   // (D even: 22 + 40 + 2 P, 22 + 2 P)
                else if (env_ok && c < D) v.x = xr[c];
            }
            if (NWF > 1) { if (c < FKS1 * 32) xput2(c, v.x, v.y); }
            else
            if (c < FSX) *(uint32_t *)(bufX + arow * FSX + c) = ev2g_pack_bf16(v.x, v.y);   // columns D .. : zeros (the k-steps' padding)
        }
        if (NWF == 1 && tid < MC::NB) lbias[tid] = fa.m.b1[tid];   // (b1 | b2 | b3 are one array on this path, each padded to its tiles)
    }
    __syncthreads();

    // FULL kernels have registers to spare (no extras, no second head pair): what the step loop otherwise re-derives every step with
    // quarter-rate integer multiplies -- this lane's rows in the step / head tables (the scenario does not change: no in-launch reset)
    // and its places in the observation -- is computed once and kept
    unsigned hb_step = 0, hb_head = 0, hb_obs_port = 0, hb_obs_env = 0;
    const double *act_run = io.actions;   // the actions of the step in work
    const float *act32_run = (F32 && !ACT) ? io.act32 + (long long)io.step0 * io.a_stride : nullptr;
    double *obs_run = io.obs, *rew_run = io.reward;   // the outputs of the step in work (STR: advanced by their step strides; else the launch's one row)
    uint8_t *done_run = io.done, *mask_run = io.mask;
    float *obs32_run = io.obs32, *act_out = ACT ? const_cast<float *>(io.act32) + (long long)e0 * P : nullptr;   // (ACT: the rows the policy writes, this workgroup's first env)
    constexpr unsigned OB = F32 ? 4u : 8u;    // bytes per observation element the full kernel writes
    if (FULL) {
        const int scn0 = ev2g_scn((valid || hcopy) ? e : e0, off, M);
        hb_step = (unsigned)(scn0 * T) * 64u;
        hb_head = (unsigned)(scn0 * (T + 1)) * (unsigned)(((SK == 1) ? 0 : (SK == 0 ? 60 : 20)) * 8);
        hb_obs_env = (unsigned)(e * D) * OB;
        if (WIDE) hb_obs_port = hb_obs_env + (unsigned)ocol * OB;
    }
    PT_DECL
#if defined(EV2G_PHASE_TIMING) && defined(EV2G_PT_OUTER)
    pt_cur[7] += pt_last - pt_k0;
#endif
    for (int kk = 0; kk < k_steps; kk++) {
#if defined(EV2G_PHASE_TIMING) && defined(EV2G_PT_OUTER)
        PT_MARK(5)
#elif !defined(EV2G_PT_BSPLIT)   // (tools/phase_timing.py -DEV2G_PT_BSPLIT: slot 7 := the battery maths' operand wait, the loop top counts as phase A)
        PT_MARK(7)
#endif
        asm volatile("" : "+s"(S));
        int tid_l = tid, g_l = g, e_l = e, q_l = q, lane_l = lane;
        asm volatile("" : "+v"(tid_l), "+v"(g_l), "+v"(e_l), "+v"(q_l), "+v"(lane_l));
        const unsigned g8 = (unsigned)g_l * 8u;   // byte offset of this port in every [E*P] float64 / int2 array
        if (!FULL && t >= T) {  // episode finished inside a fused run: in-kernel ev2g_reset for this workgroup
            if (!auto_reset) break;
            off = ev2g_scn(off, io.scn_stride, M);
            if (valid) {
                const unsigned gs8 = (unsigned)(ev2g_scn(e_l, off, M) * P + q_l) * 8u;   // this port in the scenario pool
                const i2v w = ldg32<i2v>(S->port_first_win, gs8);
                s_ta[tid_l] = w.x; s_td[tid_l] = w.y; s_ss[tid_l] = ldg32<int>(S->port_first, gs8 >> 1); s_cyc[tid_l] = 0;
                s_cap[tid_l] = 0.0; s_tot[tid_l] = 0.0; s_prev[tid_l] = 0.0; s_abse[tid_l] = 0.0; s_dirty[tid_l] = 3;
                stg32<double>(PA(EV2G_PS_PENERGY), g8, 0.0);
                stg32<double>(PA(EV2G_PS_PCURRENT), g8, 0.0);
                stg32<double>(PA(EV2G_PS_SATSUM), g8, 0.0);   // single-port chargers: charger index == port index
                stg32<int>(PA(EV2G_PS_SERVED), g8 >> 1, 0);
                if (log_cs) {   // single-port chargers: charger index == port index
                    stg32<double>(S->cs_profits, g8, 0.0); stg32<double>(S->cs_e_ch, g8, 0.0); stg32<double>(S->cs_e_dis, g8, 0.0);
                }
            }
            if (head) {
                for (int i = 0; i < 8; i++) stg32<double>(env_acc, (unsigned)e_l * 64u + (unsigned)i * 8u, 0.0);
                for (int i = 0; i < 7; i++) eacc[elg * 7 + i] = 0.0;
            }
            t = 0;
        }
        double *obs = F64 ? obs_run : ((!FULL && io.obs) ? io.obs + (long long)kk * io.o_stride : nullptr);       // uniform bases (scalar arithmetic)
        float *obs32 = F32 ? obs32_run : ((!FULL && S->x_obs32) ? (float *)S->x_obs32 + (long long)(io.step0 + kk) * S->x_o32_stride : nullptr);
        uint8_t *mask = FULL ? mask_run : (io.mask ? io.mask + (long long)kk * io.m_stride : nullptr);
        const int sstep = t + 1;
        const bool last_step = (kk == k_steps - 1) || (!FULL && sstep >= T && !auto_reset);
        int *cntk = cnt + 2 * (kk & 1);
        // next step's counters, cleared BEFORE this step's first barrier: they were last read in the battery-maths phase of the step
        // before, which a busy step closes with a barrier and a quiet step leaves at zero
        if (tid_l < 2) cnt[2 * ((kk + 1) & 1) + tid_l] = 0;

        double a_cur = a_next;   // this step's action; the prefetch below replaces a_next by the next step's
        if (ACT) {
            // ---- the policy, on the 16 observation rows of this workgroup's envs (ev2g_mlp3_inline, ev2g_mlp.h) ----
            if (NWF > 1)
                ev2g_mlp3_inline_f32<FKS1, 25, 19, FNT3, BLOCK / 64, EV2G_FUSED_RINGF>(fa.m, bufXf, FSXF, bufH1, bufH1 + 3 * 16 * MC::SH1, (uint16_t *)bufXf, (uint16_t *)(stage + 5 * RS), RS * 4, act_lds, ACT_AS, act_out, min(16, E - e0), tid_l);
            else
            ev2g_mlp3_inline<FKS1, 25, 19, FNT3, BLOCK / 64, (AE == 1 ? EV2G_FUSED_RING : EV2G_FUSED_RING2), AE>(fa.m, bufX, bufH1, bufH2, lbias, act_lds, ACT_AS, act_out, min(16 * AE, E - e0), tid_l);   // (starts and ends with a barrier: the actions are in LDS)
            act_out += io.a_stride;
            a_cur = valid ? (double)act_lds[arow * ACT_AS + q_l] : 0.0;
            // the staging slots of idle lanes must read as +0.0 in the per-env reduction (the lanes behind an env's last port never write them)
            if (!valid) {
#pragma unroll
                for (int k = 0; k < EV2G_NQ; k++) stage[k * RS + tid_l] = 0.0;
            }
            PT_MARK(7)   // (phase-timing builds: slot 7 = the policy, including the wait at its first barrier)
        }
        // ---- prefetch what the rest of this step needs (collected before the stores of phase C) ----
        // Every prefetch is ONE unconditional load from a clamped (always valid) address; the conditions are applied
        // where the value is consumed.  A load inside a divergent branch whose result merges with a default makes the
        // compiler serialise on the destination register (write-after-write) with a full vmcnt(0) drain.
        const bool more = (kk + 1 < k_steps) && (FULL || sstep < T || auto_reset);
        const int ec = valid ? e_l : e0;   // clamped env for idle lanes
        const int gc = valid ? g_l : e0 * P;
        if (ACT) {
        } else if (F64) {   // a running pointer instead of a 64-bit scalar product per step
            a_next = ldg32_nt<double>(act_run + (more ? io.a_stride : 0), (unsigned)gc * 8u);
            act_run += io.a_stride;
        } else if (F32) {
            a_next = (double)ldg32_nt<float>(act32_run + (more ? io.a_stride : 0), (unsigned)gc * 4u);
            act32_run += io.a_stride;
        } else
        a_next = IO32 ? (double)ldg32_nt<float>(io.act32 + (long long)(io.step0 + (more ? kk + 1 : kk)) * io.a_stride, (unsigned)gc * 4u)
                      : ldg32_nt<double>(io.actions + (long long)(more ? kk + 1 : kk) * io.a_stride, (unsigned)gc * 8u);
        const int scn = ev2g_scn(ec, off, M);             // this env's scenario in the resident pool
        const unsigned eT64 = FULL ? hb_step : (unsigned)(scn * T) * 64u;  // its rows in the [M,T,8] step table
        const unsigned et64 = eT64 + (unsigned)t * 64u;
        const d2v st0 = ldg32_nt<d2v>(S->step_tab, et64);                                  // charge price, discharge price
        double pf_pch = st0.x, pf_pdis = st0.y;
        // the transformer scalars only the head lane needs: ONE more load in which the head lane (q == 0) takes
        // {inflexible + solar, max_power} and its neighbour takes {min_power, setpoint}; the head lane picks the
        // neighbour's pair up with a DPP wave shift in phase E
        d2v pf_tr = ldg32_nt<d2v>(S->step_tab, et64 + ((q_l == 0) ? 16u : 32u));
        // observation head columns of this env, distributed over its P lanes as 16-byte column PAIRS: pair q, q+P
        double pf_ob0 = 0.0;
        d2v pf_h0 = {0.0, 0.0}, pf_h1 = {0.0, 0.0};
        constexpr int NHEAD = (SK == 1) ? 0 : (SK == 0 ? 60 : 20);   // 20 prices (+ 40 window columns)
        constexpr int NPAIR = NHEAD / 2;
        if (SK == 1) {
            pf_ob0 = ldg32_nt<double>(S->step_tab, eT64 + (unsigned)min(sstep, T - 1) * 64u + 40u);   // next setpoint; head lane only, masked by sstep < T
        } else {
            // observation head table [E, T+1, NHEAD]: |charge price| window (zero-padded) + load/PV/limit window, exactly
            // the values of columns 2..2+NHEAD of the observation emitted at the end of step sstep-1 (state.py:65-83, :108-135)
            const unsigned h8 = FULL ? hb_head + (unsigned)sstep * (unsigned)(NHEAD * 8) : (unsigned)((scn * (T + 1) + sstep) * NHEAD) * 8u;
            pf_h0 = ldg32_nt<d2v>(S->head_tab, h8 + (unsigned)min(q_l, NPAIR - 1) * 16u);
            if (!WIDE && P < NPAIR) pf_h1 = ldg32_nt<d2v>(S->head_tab, h8 + (unsigned)min(q_l + P, NPAIR - 1) * 16u);   // (uniform)
        }
        // ---------------- A: home lanes, charger level (ev_charger.py:137-186) ----------------
        bool occ = false;
        double cap_before = 0.0;
        int ta_a = EV2G_INT_MAX, td_a = -1;   // this port's window as phase A saw it (idle lanes: no event)
        // A wavefront none of whose ports holds an EV in this step or receives one at its end (workplace nights, early mornings: a third of the
        // wavefront-steps at cfg2) has nothing to decide in phases A and C: every action is masked, every per-port observation column and mask entry
        // is zero, nothing is staged (phase D skips such a wavefront already).  The full kernels know it from the windows in their registers.
#ifdef EV2G_NO_EMPTY_WAVE_PATH
        const bool wave_live = true;
#else
        // (one env per wavefront only: with two or three a wavefront is rarely empty and the test costs more than it saves -- cfg3 +0.8 %, cfg2 -3 %,
        // profiles/r05_ab_empty_wavefront_path.txt)
        const bool wave_live = !FULL || EPW != 1 || __ballot(valid && ((r_ta <= t && t <= r_td) || r_ta == sstep)) != 0ull;   // (uniform)
#endif
        if (valid && wave_live) {
            // every LDS operand of the phase in one batch (one wait) instead of one round trip per branch
            int ta = FULL ? r_ta : s_ta[tid_l], td = FULL ? r_td : s_td[tid_l];
            double cap_b = s_cap[tid_l], c_thr = s_cst[0 * 64 + q_l], c_dmin = s_cst[1 * 64 + q_l];
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta), "+v"(td), "+v"(cap_b), "+v"(c_thr), "+v"(c_dmin));
            ta_a = ta; td_a = td;
            occ = (ta <= t) && (t <= td);
            if (log_soc && occ) cap_before = cap_b;
            double a = occ ? a_cur : 0.0;
            // one port per charger: a / sum(a) = a / a, which is exactly +-1 for every finite action (ev_charger.py:143-149)
            a = (a > 1.0) ? 1.0 : ((a < -1.0) ? -1.0 : a);
            // Straight-line selects instead of nested branches (each divergent `if` is three scalar instructions around a handful of vector
            // ones): an empty port has a == 0, hence x == 0 and amps == 0, like the branch it replaces.  |a| <= 1, so rint(a * 1e5) is
            // within the range in which the two-FMA form of the division by 1e5 is exact (div_int_by_const, tests/test_fma_division.py):
            // no fallback division.
            const double n5 = rint(a * 100000.0), q5 = n5 * (1.0 / 100000.0);
            const double x = fma(fma(-q5, 100000.0, n5), 1.0 / 100000.0, q5);   // rnd5 (ev_charger.py:157)
            const double ac = x * c_imax, ad = x * c_dmaxabs;
            const double amps_c = (ac < c_thr) ? 0.0 : ac, amps_d = (ad > c_dmin - 0.01) ? c_dmin : ad;
            const double amps = (x > 0.0) ? amps_c : ((x < 0.0) ? amps_d : 0.0);
            s_amps[tid_l] = amps;
            stage[0 * RS + tid_l] = 0.0;
            stage[4 * RS + tid_l] = 0.0;
            stage[5 * RS + tid_l] = 0.0;
            stage[6 * RS + tid_l] = 0.0;
            stage[7 * RS + tid_l] = 0.0;
            if (amps != 0.0) items[(amps > 0.0) ? atomicAdd(&cntk[0], 1) : NS - 1 - atomicAdd(&cntk[1], 1)] = tid_l;
        }
        PT_MARK(0)
        // Departures and arrivals are known before the step (occupancy does not depend on the actions): the fields phase C needs
        // from the session record -- its last four 16-byte chunks -- travel with the other prefetches instead of being fetched,
        // dependently, inside that phase's branches.  Only wavefront-steps that have such an event issue them (about half at cfg2);
        // the other lanes of such a wavefront read record 0.
        const bool ev_dep = occ && t >= td_a, ev_arr = (ta_a == sstep);
        d2v pf_c2 = {0.0, 0.0}, pf_c7 = {0.0, 0.0};   // an arrival: {B, RN(1/B)}, {cap0, potc} of the arriving session's record ...
        d2v pf_d0 = {1.0, 1.0};                       // ... and its SessDyn: {transition_soc, charge efficiency},
        i4v pf_d1 = {0, 0, -1, 0};                    //     {discharge efficiency, efficiency-table id, dictionary entry}
        i4v pf_tl = {0, 0, 0, 0};                     // a departure: {des (two words), next window} of the leaving session's tail entry
        if (__ballot(ev_dep || ev_arr) != 0ull) {   // (uniform)
            const unsigned sse = (ev_dep || ev_arr) ? (unsigned)s_ss[tid_l] : 0u;
            static_assert(offsetof(SessRec, B) == 40 && offsetof(SessRec, rB) == 48 && offsetof(SessRec, cap0) == 112 && offsetof(SessRec, potc) == 120 && sizeof(SessTail) == 16 &&
                          offsetof(SessTail, nt_arr) == 8, "SessRec / SessTail layout");
            pf_c2 = ldg32<d2v_a8>(S->rec, sse * (unsigned)sizeof(SessRec) + 40u); pf_c7 = ldg32<d2v>(S->rec, sse * (unsigned)sizeof(SessRec) + 112u);
            pf_tl = ldg32<i4v>(S->tail, sse * 16u);
            pf_d0 = ldg32<d2v>(S->sess_dyn, sse * 32u); pf_d1 = ldg32<i4v>(S->sess_dyn, sse * 32u + 16u);
        }
#if defined(EV2G_PHASE_TIMING) && defined(EV2G_PT_OUTER)
        PT_MARK(0)
#else
        PT_MARK(6)
#endif
        lds_barrier();
        PT_MARK(1)

        // ---------------- B: worker lanes, battery maths on the compact list ----------------
        // The wavefronts that hold list items are the critical path of the whole workgroup (the others wait at the next
        // barrier): they issue at raised priority until their items are done.
        const int nch = cntk[0], ndis = cntk[1];
        // A step in which no port of the workgroup has an EV to integrate (the night half of a workplace episode, the early morning: 40 % of the
        // workgroup-steps at cfg2) has no battery-maths phase to fence: the second barrier is skipped.  Every wavefront reads the same counts.
        if (nch + ndis != 0) {
        __builtin_amdgcn_s_setprio(3);
        {
            const int nchp = (nch + 63) & ~63;
            for (int i = tid_l; i < nchp + ndis; i += BLOCK) {
                int h = -1;
                if (i < nch) h = items[i];
                else if (i >= nchp) h = items[NS - 1 - (i - nchp)];
                if (h >= 0) {
                    const double amps_h = s_amps[h];
                    const int dw_h = s_dirty[h];
                    const int lut_id = ((dw_h >> 8) & LUTMASK) - 1;
                    // table entry and session record are independent loads: one memory round trip, not two.  The look-up
                    // is unconditional (clamped index); whether it applies is decided where it is used.
                    const int li = (lut_id >= 0) ? ev_lut_index(lut_id, amps_h) : -1;
                    double lut_raw = ldg32<double>(S->lut, (unsigned)max(li, 0) * 8u);
                    // full kernels: the operands of the car model from the dictionary (an L1 hit), transition_soc / efficiency from LDS
                    const unsigned r8 = ((FULL && wa.dict) ? (unsigned)dw_h >> 20 : (unsigned)s_ss[h]) * (unsigned)sizeof(SessRec);
                    const double cap0 = s_cap[h], prev0 = s_prev[h];
                    const int cyc0 = s_cyc[h];
                    EvRes o;
                    if (i < nchp) {   // (uniform: the discharge items start on a wavefront boundary)
                        SessRec r;
                        if (FULL) { const double ts_h = s_ts[h], eta_h = s_etac[h]; r = ldg32_cls_charge(S->cls_rec, r8); r.ts = ts_h; r.eta_ch = eta_h; }
                        else r = ldg32_rec_charge(S->rec, r8);
                        asm volatile("" : "+v"(lut_raw));
#if defined(EV2G_PHASE_TIMING) && defined(EV2G_PT_BSPLIT)
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        PT_MARK(7)
#endif
                        const double lutv = (li >= 0) ? lut_raw : 1.0 / 100.0;
                        o = ev_math_charge(r, lutv, amps_h, cap0, prev0, s_tot[h], cyc0, sixty_over_dt, dt_over_60, pow2_dt, lut_id >= 0);
                    } else {
                        SessRec r;
                        if (FULL) { const double eta_h = s_etad[h]; r = ldg32_cls_discharge(S->cls_rec, r8); r.eta_dis = eta_h; }
                        else r = ldg32_rec_discharge(S->rec, r8);
                        asm volatile("" : "+v"(lut_raw));
                        const double lutv = (li >= 0) ? lut_raw : 1.0 / 100.0;
                        o = ev_math_discharge(r, lutv, amps_h, cap0, prev0, s_tot[h], cyc0, dtd, lut_id >= 0, S->rdt, S->dt_fdiv != 0);   // (fetched here, by the discharging wavefronts only: not kept across the step loop)
                    }
                    if (o.cycles != cyc0 || o.energy != 0.0 || o.cap != cap0 || o.prev_power != prev0) s_dirty[h] |= 1;
                    s_cap[h] = o.cap;
                    s_prev[h] = o.prev_power;
                    s_tot[h] = o.tot_e;
                    s_cyc[h] = o.cycles;
                    s_amps[h] = o.energy;
                    if (log_soc) s_abse[h] += fabs(o.energy);
                    stage[0 * RS + h] = o.energy * 60.0 / dtd;
                    stage[(i < nch ? 4 : 5) * RS + h] = fabs(o.energy);
                    stage[6 * RS + h] = (double)o.emerg;
                    stage[7 * RS + h] = o.current;
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        PT_MARK(2)
        lds_barrier();
        }
        PT_MARK(1)

        // ---------------- C: home lanes (from here on everything of one env lives in one wavefront) ----------------
        __builtin_amdgcn_s_waitcnt(0x0F70);   // collect the prefetches before this phase issues stores (vmcnt(0))
        // ... and make that visible to the compiler's wait-count tracking: as outputs of this (empty) asm the
        // prefetched registers are plain values from here on, so their later uses -- after this phase's stores, and
        // across the loop back-edge for the next action -- no longer cost a conservative vmcnt(0) drain
        asm volatile("" : "+v"(a_next), "+v"(pf_pch), "+v"(pf_pdis), "+v"(pf_tr), "+v"(pf_ob0), "+v"(pf_h0), "+v"(pf_h1));
        asm volatile("" : "+v"(pf_c2), "+v"(pf_c7), "+v"(pf_tl), "+v"(pf_d0), "+v"(pf_d1));
        bool occ_any = false;   // an EV on this port before or after the step
        if (FULL && valid && !wave_live) {   // the empty wavefront's outputs: zeros
            stg32<uint8_t>(mask, (unsigned)g_l, (uint8_t)0);
            const unsigned ocol_l = (unsigned)((SK == 1) ? 3 + 3 * q_l : (SK == 0 ? 62 + 2 * q_l : 22 + 2 * q_l));
            if (F64) {
                const unsigned o8 = WIDE ? hb_obs_port : hb_obs_env + ocol_l * 8u;
                if (STR_NT) { stg32_nt<d2v>(obs, o8, (d2v){0.0, 0.0}); if (SK == 1) stg32_nt<double>(obs, o8 + 16u, 0.0); }
                else { stg32<d2v>(obs, o8, (d2v){0.0, 0.0}); if (SK == 1) stg32<double>(obs, o8 + 16u, 0.0); }
            }
            if (F32) {
                const unsigned o4 = WIDE ? hb_obs_port : hb_obs_env + ocol_l * 4u;
                if (SK == 1) { stg32<float>(obs32, o4, 0.f); stg32<float>(obs32, o4 + 4u, 0.f); stg32<float>(obs32, o4 + 8u, 0.f); }
                else stg32<f2v>(obs32, o4, (f2v){0.f, 0.f});
                if (ACT) {
                    if (SK == 1 && NWF > 1) { bufXf[arow * FSXF + ocol_l] = 0.f; bufXf[arow * FSXF + ocol_l + 1] = 0.f; bufXf[arow * FSXF + ocol_l + 2] = 0.f; }
                    else
                    if (SK == 1) { bufX[arow * FSX + ocol_l] = 0; bufX[arow * FSX + ocol_l + 1] = 0; bufX[arow * FSX + ocol_l + 2] = 0; }   // (odd columns: three 16-bit words)
                    else xput2((int)ocol_l, 0.f, 0.f);
                }
            }
        }
        if (valid && wave_live) {
            double profit = 0.0, satpen = 0.0, pot = 0.0;
            // every LDS operand of this phase in ONE batch (one wait), whichever branch consumes it: read one by one behind the
            // branches below, each of them was its own LDS round trip on the workgroup-step chain
            int ta = FULL ? r_ta : s_ta[tid_l], td = FULL ? r_td : s_td[tid_l], ss_now = s_ss[tid_l];
            double cap = s_cap[tid_l];
            double b_energy = s_amps[tid_l], b_cur = stage[7 * RS + tid_l], b_ech = stage[4 * RS + tid_l], b_edis = stage[5 * RS + tid_l];
            double b_bcap = FULL ? r_bcap : s_bcap[tid_l], b_potc = FULL ? r_potc : s_potc[tid_l], b_tot = (SK == 1) ? s_tot[tid_l] : 0.0;
            double c_maxp = s_cst[2 * 64 + q_l], c_minp = s_cst[3 * 64 + q_l];
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta), "+v"(td), "+v"(ss_now), "+v"(cap), "+v"(b_energy), "+v"(b_cur), "+v"(b_ech), "+v"(b_edis),
                         "+v"(b_bcap), "+v"(b_potc), "+v"(b_tot), "+v"(c_maxp), "+v"(c_minp));
            bool departed = false;
            if (occ) {
                const double energy = b_energy;
                const double current = b_cur;
                if (energy != 0.0) {  // profit by the sign of the ACTION (ev_charger.py:178,194), staged under 4 / 5
                    const double ech = b_ech;
                    profit = (ech != 0.0) ? ech * pf_pch : b_edis * pf_pdis;
                }
                if (log_cs && energy != 0.0) {   // charger accumulators (ev_charger.py:178-181,194-197): one lane per charger and step
                    __hip_atomic_fetch_add((double __attribute__((address_space(1))) *)((gptr)S->cs_profits + g8), profit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add((double __attribute__((address_space(1))) *)((gptr)S->cs_e_ch + g8), b_ech, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add((double __attribute__((address_space(1))) *)((gptr)S->cs_e_dis + g8), b_edis, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (current - 0.0001 > c_imax) stg32<int>(S->env_fault, (unsigned)e_l * 4u, 1);  // ev_charger.py:203-205
                if (last_step) { stg32<double>(PA(EV2G_PS_PENERGY), g8, energy); stg32<double>(PA(EV2G_PS_PCURRENT), g8, current); }
                if (log_soc) stg32<double>(S->soc_log + (long long)t * P, g8 + (unsigned)e_l * (unsigned)((T - 1) * P * 8), (current != 0.0) ? cap_before : -cap_before);
                if (t >= td) {  // departure (ev_charger.py:209-229, ev.py:191-214)
                    const int ss = ss_now;
                    const double des = __hiloint2double(pf_tl.y, pf_tl.x);
                    const double score = (cap < des - 0.001) ? cap / des : 1.0;
                    if (RK == 3) satpen = ev2g_departure_term(S->reward_kind, S->cost_kind, score, cap, des);
                    else if (RK != 1 || S->cost_kind == 1) satpen = 100.0 * exp(-10.0 * score);
                    // fire-and-forget device atomics (no returned value => no memory round trip on this path); exactly one
                    // lane updates a given charger per step, so the result does not depend on any ordering
                    __hip_atomic_fetch_add((int __attribute__((address_space(1))) *)(PA(EV2G_PS_SERVED) + (g8 >> 1)), 1,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add((double __attribute__((address_space(1))) *)(PA(EV2G_PS_SATSUM) + g8), score,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    stg32<double>(slabS, (unsigned)ss * 8u, cap);
                    if (log_soc) stg32<double>((slabS + SS8), (unsigned)ss * 8u, s_abse[tid_l]);
                    ta = pf_tl.z; td = pf_tl.w;   // window of the port's next session
                    departed = true;
                    if (FULL) { r_ta = ta; r_td = td; } else { s_ta[tid_l] = ta; s_td[tid_l] = td; }
                    ss_now = (ta != EV2G_INT_MAX) ? ss + 1 : -1;
                    s_ss[tid_l] = ss_now;
                    s_cyc[tid_l] = 0;
                    s_dirty[tid_l] |= 2;
                }
            }
            if (ta == sstep) {  // arrival at the end of this step (ev2gym_env.py:399-417, ev.py:115-136)
                if (departed) {   // the next session arrives right behind a departure of this very step (the reference's spawner leaves a
                                  // gap, replayed scenarios need not): its record was not the one prefetched
                    const unsigned r8 = (unsigned)ss_now * (unsigned)sizeof(SessRec);
                    pf_c2 = ldg32<d2v_a8>(S->rec, r8 + 40u); pf_c7 = ldg32<d2v>(S->rec, r8 + 112u);
                    pf_d0 = ldg32<d2v>(S->sess_dyn, (unsigned)ss_now * 32u); pf_d1 = ldg32<i4v>(S->sess_dyn, (unsigned)ss_now * 32u + 16u);
                }
                cap = pf_c7.x;
                const double B = pf_c2.x;
                const double potc = pf_c7.y;   // v * min(pacmax*1000/v, charger max current) / 1000 (utils.py:773-777), evaluated when the session was loaded
                s_cap[tid_l] = cap; s_tot[tid_l] = 0.0; s_prev[tid_l] = 0.0; s_cyc[tid_l] = 0;
                if (FULL) { r_bcap = B; r_potc = potc; r_rb = pf_c2.y; } else { s_bcap[tid_l] = B; s_potc[tid_l] = potc; }
                s_abse[tid_l] = 0.0;
                b_bcap = B; b_potc = potc; b_tot = 0.0;
                const int lut_new = pf_d1.z;
                stg32<d2v>(wa.lines, l64 + 48u, (d2v){B, potc});   // (the table id travels in s_dirty and reaches the line's head chunk in the epilogue)
                // the attached EV's SessDyn next to its line: what the prologue of a later launch reads (every instantiation writes it)
                stg32<d2v>(wa.port_dyn, (unsigned)g_l * 32u, pf_d0); stg32<i4v>(wa.port_dyn, (unsigned)g_l * 32u + 16u, pf_d1);
                if (FULL) { s_ts[tid_l] = pf_d0.x; s_etac[tid_l] = pf_d0.y; s_etad[tid_l] = __hiloint2double(pf_d1.y, pf_d1.x); }
                stg32<double>(PA(EV2G_PS_PENERGY), g8, 0.0);
                stg32<double>(PA(EV2G_PS_PCURRENT), g8, 0.0);
                s_dirty[tid_l] = (int)((unsigned)((s_dirty[tid_l] & 3) | 1 | ((lut_new + 1) << 8)) | ((FULL && wa.dict) ? (unsigned)pf_d1.w << 20 : 0u));
            }
            const bool occ_after = (ta <= sstep) && (sstep <= td);
            if (RK == 3 && occ_after && S->reward_kind >= 9) {   // (pst_)V2G_profitmaxV2: every connected EV (reward.py:173-195)
                const unsigned r8 = (unsigned)ss_now * (unsigned)sizeof(SessRec);
                satpen += ev2g_connected_term(ldg32<double>(S->tail, (unsigned)ss_now * 16u), cap,
                                              ldg32<double>(S->rec, r8 + (unsigned)offsetof(SessRec, pacmax)), sixty_over_dt, td, sstep);
            }
            occ_any = occ || occ_after;
            if (FULL || mask) stg32<uint8_t>(mask, (unsigned)g_l, occ_after ? 1 : 0);
            double o0 = 0.0, o1 = 0.0, o2 = 0.0;
            if (occ_after) {
                const double soc = FULL ? ev2g_fdiv2(cap, b_bcap, r_rb) : cap / b_bcap;   // bit-identical (ev2g_device.h)
                if (SK == 1) { o0 = (soc == 1.0) ? 1.0 : 0.5; o1 = b_tot; o2 = (double)(sstep - ta); }
                else { o0 = soc; o1 = (double)(td - sstep); }
                if (soc < 1.0 && td > sstep) pot = b_potc;  // utils.py:771
            }
            pot = (pot > c_maxp) ? c_maxp : ((pot < c_minp) ? 0.0 : pot);   // per-charger clamp (utils.py:779-789)
            // (the narrow full kernels sit at the register limit: they re-derive the port's column from the env's base)
            const unsigned ocol_l = (unsigned)((SK == 1) ? 3 + 3 * q_l : (SK == 0 ? 62 + 2 * q_l : 22 + 2 * q_l));
            if (F64 || (!FULL && obs)) {
                const unsigned o8 = WIDE ? hb_obs_port : (FULL ? hb_obs_env + ocol_l * 8u : (unsigned)(e_l * D + ocol) * 8u);
                if (STR_NT) { stg32_nt<d2v>(obs, o8, (d2v){o0, o1}); if (SK == 1) stg32_nt<double>(obs, o8 + 16u, o2); }
                else {
                stg32<d2v>(obs, o8, (d2v){o0, o1});   // one 16-byte store (D and the column offset are even for SK != 1)
                if (SK == 1) stg32<double>(obs, o8 + 16u, o2);
                }
            }
            if (F32 || (!FULL && obs32)) {
                const unsigned o4 = WIDE ? hb_obs_port : (FULL ? hb_obs_env + ocol_l * 4u : (unsigned)(e_l * D + ocol) * 4u);
                if (SK == 1) { stg32<float>(obs32, o4, (float)o0); stg32<float>(obs32, o4 + 4u, (float)o1); stg32<float>(obs32, o4 + 8u, (float)o2); }
                else stg32<f2v>(obs32, o4, (f2v){(float)o0, (float)o1});
                if (ACT) {   // the policy's copy of the same columns
                    if (SK == 1 && NWF > 1) { bufXf[arow * FSXF + ocol_l] = (float)o0; bufXf[arow * FSXF + ocol_l + 1] = (float)o1; bufXf[arow * FSXF + ocol_l + 2] = (float)o2; }
                    else
                    if (SK == 1) {
                        const uint32_t w01 = ev2g_pack_bf16((float)o0, (float)o1), w2 = ev2g_pack_bf16((float)o2, 0.f);
                        bufX[arow * FSX + ocol_l] = (uint16_t)w01; bufX[arow * FSX + ocol_l + 1] = (uint16_t)(w01 >> 16); bufX[arow * FSX + ocol_l + 2] = (uint16_t)w2;
                    } else xput2((int)ocol_l, (float)o0, (float)o1);
                }
            }
            if (log_cs) {   // cs_power / cs_current of the step (ev2gym_env.py:533-535) and the chargers' current_power_output / current_total_amps
                const double pw = occ ? stage[0 * RS + tid_l] : 0.0, cur = occ ? b_cur : 0.0;
                const unsigned hc8 = ((unsigned)(t * E + e_l) * (unsigned)P + (unsigned)q_l) * 8u;
                stg32<double>(S->cs_power_hist, hc8, pw); stg32<double>(S->cs_cur_hist, hc8, cur);
                if (last_step) { stg32<double>(S->cs_power_now, g8, pw); stg32<double>(S->cs_cur_now, g8, cur); }
            }
            stage[1 * RS + tid_l] = profit;
            stage[2 * RS + tid_l] = satpen;
            stage[3 * RS + tid_l] = pot;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wavefront's LDS writes are visible to itself
        PT_MARK(3)

        // ---------------- D: per-env reduction inside the wavefront (fixed tree: bit-reproducible) ----------------
        // lane = k*8 + j: quantity k, chain j.  Every lane issues its 8 LDS reads unconditionally (clamped index, masked
        // by a select; the slots of idle lanes hold zeros), sums two chains, and three DPP xor-butterflies leave env w's
        // sum of quantity k in lane k*8.  That lane parks it in the env's head slot of the stage row it just reduced
        // (stage[k][first port of env w]; only this lane ever reads that slot in this phase), where the head lane --
        // the only consumer of env-level sums -- picks all eight up below.  LDS operations of one wavefront execute in
        // order, so no barrier is involved.
        // A wavefront whose envs hold no EV before or after this step (the night half of a workplace episode, the early
        // morning) has nothing to add up: every staged value of its ports is an exact +0.0.
        double esum[EV2G_NQ];
#pragma unroll
        for (int kq = 0; kq < EV2G_NQ; kq++) esum[kq] = 0.0;
        // SquaredTrackingErrorRewardWithPenalty only (a run-time reward): the port powers once more, charger by charger like the reference
        // (RewardIn::usage_seq, ev2g_device.h) -- by the env's head lane, before the reduction parks the env's sum in its first slot
        double useq = 0.0;
        if (RK == 3 && S->reward_kind == 4 && head) {
            const double *prow = stage + tid_l;   // slot == reference port on this path (one transformer, single-port chargers)
            for (int c0 = 0; c0 < P; c0 += 8) {
                double x[8];
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = prow[min(c0 + i, P - 1)];
#pragma unroll
                for (int i = 0; i < 8; i++) if (c0 + i < P) useq += x[i];
            }
        }
        if (__ballot(occ_any) != 0ull) {   // (uniform)
        {
            const int k = lane_l >> 3, j = lane_l & 7;
            const int wbase = (tid_l & ~63);
            const double *row = stage + k * RS;
            // The order of the additions is the same in every branch (chains j, j+16, j+32, j+48 and j+8, j+24, j+40, j+56, then the
            // butterfly); the branches differ in which of the operands they know to be zero without looking.
            if (EPW == 1) {   // (uniform) one env per wavefront: the slots behind its last port belong to idle lanes and still hold the
                              // zeros they were initialised with -- eight reads at constant offsets, no masks, no index clamps
                const double *r0 = row + wbase + j;
                double xa[4], xb[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { xa[u] = r0[16 * u]; xb[u] = r0[16 * u + 8]; }
                double acc = 0.0, accb = 0.0;
#pragma unroll
                for (int u = 0; u < 4; u++) { acc += xa[u]; accb += xb[u]; }
                acc += accb;
                acc += xor1_f64(acc);
                acc += xor2_f64(acc);
                acc += xor4_f64(acc);
                // lane 8k now holds the env's sum of quantity k: with ONE env per wavefront the sums are wave-uniform, so they are
                // read out with v_readlane (scalar registers) instead of being parked in LDS and read back by every lane --
                // two LDS round trips less on the step's chain.  (Quantity 7, the summed current, has no consumer on this path.)
                const int acc_lo = __double2loint(acc), acc_hi = __double2hiint(acc);
#pragma unroll
                for (int kq = 0; kq < EV2G_NQ - 1; kq++)
                    esum[kq] = __hiloint2double(__builtin_amdgcn_readlane(acc_hi, kq * 8), __builtin_amdgcn_readlane(acc_lo, kq * 8));
            } else {          // several envs per wavefront: P <= 32, so ports j+32 .. j+56 do not exist (their terms were exact zeros);
                              // an unclamped index past the env reads a neighbour's slot (or, behind the last row, the array that
                              // follows `stage` in LDS) and is masked
                const bool upper = P > 16;   // (uniform)
#ifndef EV2G_NO_DPAR
                // two or three envs per wavefront (the PublicPST benchmark shape: 3 x 20 ports): their reductions are independent -- all LDS
                // reads first, then the three add / butterfly chains side by side -- instead of one env after the other, each behind its own
                // LDS round trip (1.9 k of a workgroup-step's 10.8 k ticks at cfg3)
                auto envs_at_once = [&](auto NEc) __attribute__((always_inline)) {
                    constexpr int NE = decltype(NEc)::value;
                    double ra0[NE], rb0[NE], ra1[NE], rb1[NE];
#pragma unroll
                    for (int w = 0; w < NE; w++) {
                        const double *r0 = row + wbase + w * ES + j;
                        ra0[w] = r0[0]; rb0[w] = r0[8]; ra1[w] = 0.0; rb1[w] = 0.0;
                        if (upper) { ra1[w] = r0[16]; rb1[w] = r0[24]; }
                    }
                    double acc[NE];
#pragma unroll
                    for (int w = 0; w < NE; w++) {
                        const int a = wbase + w * ES, b = a + P, i = a + j;
                        double ac = 0.0, accb = 0.0;
                        ac += (i < b) ? ra0[w] : 0.0;
                        accb += (i + 8 < b) ? rb0[w] : 0.0;
                        if (upper) { ac += (i + 16 < b) ? ra1[w] : 0.0; accb += (i + 24 < b) ? rb1[w] : 0.0; }
                        acc[w] = ac + accb;
                    }
#pragma unroll
                    for (int w = 0; w < NE; w++) acc[w] += xor1_f64(acc[w]);
#pragma unroll
                    for (int w = 0; w < NE; w++) acc[w] += xor2_f64(acc[w]);
#pragma unroll
                    for (int w = 0; w < NE; w++) acc[w] += xor4_f64(acc[w]);
#pragma unroll
                    for (int w = 0; w < NE; w++) if (j == 0) stage[k * RS + wbase + w * ES] = acc[w];
                };
                if (EPW == 3) envs_at_once(std::integral_constant<int, 3>{});
                else if (EPW == 2) envs_at_once(std::integral_constant<int, 2>{});
                else
#endif
#pragma unroll 1
                for (int w = 0; w < EPW; w++) {
                    const int a = wbase + w * ES, b = a + P;
                    const double *r0 = row + a + j;
                    const double ra0 = r0[0], rb0 = r0[8];
                    double ra1 = 0.0, rb1 = 0.0;
                    if (upper) { ra1 = r0[16]; rb1 = r0[24]; }
                    const int i = a + j;
                    double acc = 0.0, accb = 0.0;
                    acc += (i < b) ? ra0 : 0.0;
                    accb += (i + 8 < b) ? rb0 : 0.0;
                    if (upper) { acc += (i + 16 < b) ? ra1 : 0.0; accb += (i + 24 < b) ? rb1 : 0.0; }
                    acc += accb;
                    acc += xor1_f64(acc);
                    acc += xor2_f64(acc);
                    acc += xor4_f64(acc);
                    if (j == 0) stage[k * RS + a] = acc;
                }
            }
        }
        if (EPW != 1) {   // (uniform)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int kq = 0; kq < EV2G_NQ - 1; kq++) esum[kq] = stage[kq * RS + (tid_l - q_l)];   // the env's sums (its head slot), every lane (quantity 7, the summed current, has no consumer on this path)
        }
        }

        PT_MARK(4)
        // ---------------- E: per env (head lane) + observation head (the env's lanes) ----------------
        const double usage = esum[0];
        // {min_power, setpoint} from the next lane (wave_shl:1 -- the head lane's neighbour always belongs to the same env)
        const double pf_base = pf_tr.x, pf_maxp = pf_tr.y;
        const double pf_minp = dpp_mov_f64<0x130>(pf_tr.x), pf_sp = dpp_mov_f64<0x130>(pf_tr.y);
        // Transformer.reset + step + get_how_overloaded (transformer.py:258-302), evaluated wave-wide (meaningful in head lanes)
        const double tr_power = pf_base + usage;   // inflexible_load[t] + solar_power[t] + sum of the charger powers
        const double over = (tr_power > pf_maxp + 0.0001 || tr_power < pf_minp - 0.0001) ? fabs(tr_power - pf_maxp) : 0.0;
        if (WIDE || P >= 3) {
            // the three history entries of the step in ONE store: lane 0 of the env writes usage[t], lane 1 the overload
            // (handed over by a DPP wave shift), lane 2 potential[t+1] -- not three stores with one active lane each
            const double over_n = dpp_mov_f64<0x138>(over);   // wave_shr:1: lane L takes lane L-1's value
            if (valid && q_l < 3 && (q_l != 2 || sstep < T)) {
                const double hv = (q_l == 0) ? usage : ((q_l == 1) ? over_n : esum[3]);
                // row (e, t) of the history array [E, T, 3]: words {usage, potential, overload}; lane 2 writes into row t + 1 -- 40 bytes in all
                const unsigned hword = (q_l == 0) ? 0u : ((q_l == 1) ? 16u : 32u);
                stg32<double>(hist + (long long)t * 24, (unsigned)(e_l * T) * 24u + hword, hv);
            }
        }
        if (head) {
            double *ea = eacc + elg * 7;
            // all six accumulator words are read up front (one LDS wait) and written back together at the end; reading
            // each one next to its update made every update wait for its own LDS round trip
            const double ea0 = ea[0], ea1 = ea[1], ea2 = ea[2], ea3 = ea[3], ea4 = ea[4], ea5 = ea[5];
            const double ea6 = (RK == 3) ? ea[6] : 0.0;
            const unsigned e8 = (unsigned)e_l * 8u;
            double over100 = 0.0;
            if (RK == 0 || RK == 3) over100 = 100.0 * over;
            if (last_step) stg32<double>(S->tr_power_now, e8, tr_power);
            const double potn = esum[3];
            if (!WIDE && P < 3) {   // two-port envs: no third lane to share the history stores with
                const unsigned h8 = (unsigned)(e_l * T + t) * 24u;
                stg32<double>(hist, h8 + 16u, over);
                stg32<double>(hist, h8, usage);
                if (sstep < T) stg32<double>(hist, h8 + 32u, potn);
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
                ri.costs = costs; ri.usage = usage; ri.usage_seq = (S->reward_kind == 4) ? useq : usage; ri.sp = pf_sp; ri.pot_t = ea5; ri.pot_tm1 = ea6; ri.over100 = over100;
                ri.user = esum[2]; ri.tr0_maxp = pf_maxp;
                reward = ev2g_reward(S->reward_kind, ri);
            } else {  // ProfitMax_TrPenalty_UserIncentives reward.py:34-44
                reward = costs - over100 - esum[2];
            }
            const double n0 = ea0 + reward, n1 = ea1 + costs, n2 = ea2 + esum[4], n3 = ea3 + esum[5], n4 = ea4 + esum[6];
            ea[0] = n0; ea[1] = n1; ea[2] = n2; ea[3] = n3; ea[4] = n4; ea[5] = potn;
            if (RK == 3) ea[6] = ea5;
            if (FULL) { stg32<double>(rew_run, e8, reward); stg32<uint8_t>(done_run, (unsigned)e_l, (sstep >= T) ? 1 : 0); }
            if (!FULL && io.reward) stg32<double>(io.reward + (long long)kk * io.r_stride, e8, reward);
            if (!FULL && io.done) stg32<uint8_t>(io.done + (long long)kk * io.d_stride, (unsigned)e_l, (sstep >= T) ? 1 : 0);
            if (!FULL && S->x_cost)   // cost_function (rl_agent/cost.py:8-27); the overload weight is applied here when the reward does not carry it
                stg32<double>(S->x_cost + (long long)(io.step0 + kk) * S->x_c_stride, e8, (S->cost_kind == 2) ? costs : 100.0 * over + esum[2]);
            if (sstep >= T || last_step) {  // publish the running episode totals (get_statistics reads them)
                const unsigned a8 = (unsigned)e_l * 64u;
                stg32<d2v>(env_acc, a8, (d2v){n0, n1});
                stg32<d2v>(env_acc, a8 + 16u, (d2v){n2, n3});
                stg32<double>(env_acc, a8 + 32u, n4);
            }
        }
        if ((valid || hcopy) && (F32 || (!FULL && obs32))) {
            const unsigned o4 = FULL ? hb_obs_env : (unsigned)(e_l * D) * 4u;
            if (SK == 1) {
                if (q_l == 0) {
                    const float h0 = (float)((double)sstep / (double)T), h1 = (float)((sstep < T) ? pf_ob0 : 0.0), h2 = (float)usage;
                    stg32<float>(obs32, o4, h0);
                    stg32<float>(obs32, o4 + 4u, h1);
                    stg32<float>(obs32, o4 + 8u, h2);
                    if (ACT && NWF > 1) { bufXf[arow * FSXF] = h0; bufXf[arow * FSXF + 1] = h1; bufXf[arow * FSXF + 2] = h2; }
                    else
                    if (ACT) {   // the policy's copy: columns 0, 1 as one word, column 2 on its own (column 3 belongs to port 0)
                        *(uint32_t *)(bufX + arow * FSX) = ev2g_pack_bf16(h0, h1);
                        bufX[arow * FSX + 2] = (uint16_t)ev2g_pack_bf16(h2, 0.f);
                    }
                }
            } else {
                if (q_l == 0) stg32<f2v>(obs32, o4, (f2v){(float)sstep, (float)usage});
                if (q_l < NPAIR) stg32<f2v>(obs32, o4 + 8u + (unsigned)q_l * 8u, (f2v){(float)pf_h0.x, (float)pf_h0.y});
                if (ACT) {
                    if (q_l == 0) xput2(0, (float)sstep, (float)usage);
                    if (q_l < NPAIR) xput2(2 + 2 * q_l, (float)pf_h0.x, (float)pf_h0.y);
                }
                if (!WIDE) {
                    if (q_l + P < NPAIR) stg32<f2v>(obs32, o4 + 8u + (unsigned)(q_l + P) * 8u, (f2v){(float)pf_h1.x, (float)pf_h1.y});
                    const unsigned h8 = (unsigned)((scn * (T + 1) + sstep) * NHEAD) * 8u;
                    for (int pi = q_l + 2 * P; pi < NPAIR; pi += P) {
                        const d2v hv = ldg32<d2v>(S->head_tab, h8 + (unsigned)pi * 16u);
                        stg32<f2v>(obs32, o4 + 8u + (unsigned)pi * 8u, (f2v){(float)hv.x, (float)hv.y});
                    }
                }
            }
        }
        if (valid && (F64 || (!FULL && obs))) {
            const unsigned o8 = FULL ? hb_obs_env : (unsigned)(e_l * D) * 8u;
            if (SK == 1) {  // PublicPST state.py:6-35
                if (q_l == 0) {
                    stg32<double>(obs, o8, (double)sstep / (double)T);
                    stg32<double>(obs, o8 + 8u, (sstep < T) ? pf_ob0 : 0.0);
                    stg32<double>(obs, o8 + 16u, usage);
                }
            } else {  // V2G_profit_max(_loads) state.py:65-83, :108-135: columns 2.. are a copy of the head table row
                if (STR_NT) {
                    if (q_l == 0) stg32_nt<d2v>(obs, o8, (d2v){(double)sstep, usage});
                    if (q_l < NPAIR) stg32_nt<d2v>(obs, o8 + 16u + (unsigned)q_l * 16u, pf_h0);
                } else {
                if (q_l == 0) stg32<d2v>(obs, o8, (d2v){(double)sstep, usage});
                if (q_l < NPAIR) stg32<d2v>(obs, o8 + 16u + (unsigned)q_l * 16u, pf_h0);
                }
                if (!WIDE) {
                    if (q_l + P < NPAIR) stg32<d2v>(obs, o8 + 16u + (unsigned)(q_l + P) * 16u, pf_h1);
                    const unsigned h8 = (unsigned)((scn * (T + 1) + sstep) * NHEAD) * 8u;
                    for (int pi = q_l + 2 * P; pi < NPAIR; pi += P)    // tiny envs (P < 15): the remaining pairs, unprefetched
                        stg32<d2v>(obs, o8 + 16u + (unsigned)pi * 16u, ldg32<d2v>(S->head_tab, h8 + (unsigned)pi * 16u));
                }
            }
        }
        PT_MARK(5)
        PT_STEP_END(cntk[0] + cntk[1] == 0)
        t += 1;
        if (STR) { obs_run += io.o_stride; rew_run += io.r_stride; done_run += io.d_stride; mask_run += io.m_stride; }
        if (ACT) obs32_run += io.o_stride;
        // The next step's phase A rewrites stage[0,4..7] / s_amps of this wavefront's own lanes only after this
        // wavefront finished reading them (program order); other wavefronts never touch these slots outside phase B,
        // which is fenced by the two barriers.
    }
    __syncthreads();
    if (valid) {
        const int d = s_dirty[tid];
        const unsigned g8 = (unsigned)g * 8u;
        if (d & 3) stg32<i4v>(wa.lines, l64, (i4v){FULL ? r_ta : s_ta[tid], FULL ? r_td : s_td[tid], s_ss[tid], ev2g_line_pack(s_cyc[tid], ((d >> 8) & LUTMASK) - 1)});
        if (d & 1) {
            stg32<d2v>(wa.lines, l64 + 16u, (d2v){s_cap[tid], s_tot[tid]});
            stg32<d2v>(wa.lines, l64 + 32u, (d2v){s_prev[tid], s_abse[tid]});
        }
    }
#if defined(EV2G_PHASE_TIMING) && defined(EV2G_PT_OUTER)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PT_MARK(6)
#endif
    PT_FLUSH
}
