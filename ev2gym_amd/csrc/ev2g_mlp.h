// ev2g_mlp.h -- the policy head of a rollout: obs[E,D] (float32) -> actions[E,P] (float32) through a three-layer MLP
// (D -> H1 -> H2 -> P, ReLU, ReLU, tanh: the SB3 MlpPolicy actor shape of BASELINE configs[4]), as ONE kernel between two
// env steps.
//
// The reference trains SB3 agents against one CPU env (train_stable_baselines.py:62-130); with thousands of envs resident
// on the GPU a step of the engine takes ~10 us, so a policy evaluated through a dozen framework launches (GEMM, bias,
// activation, casts ...) would be 90 % of a rollout step.  The kernels here keep the whole forward on chip.  Three generations live in
// this file (newest last, each described where it is defined):
//   * ev2g_mlp3_s16 (round 4; what the shipped layer widths run): built around the weight stream -- 16 (or 32) env rows per workgroup on
//     every CU, v_mfma_f32_16x16x32_bf16 with the weights as the A operand, one static fragment sequence per wavefront through a register
//     ring; also the FLOAT32 network as split bf16 operands (two / three terms per weight);
//   * ev2g_mlp3_any / ev2g_mlp3_f32: any layer widths (bf16 operands / float32 operands on v_mfma_f32_32x32x2_f32), 32 rows per workgroup;
//   * ev2g_mlp3_fixed (rounds 2-3; EV2G_MLP_OLD=1, kept for A/B runs): 32-row workgroups, v_mfma_f32_32x32x16_bf16, weights as the B operand.
// Common to all: weights are packed ONCE on the host into the MFMA fragment order of the kernel that will read them (bf16, or float32 for
// ev2g_mlp3_f32), so a lane's operand is one coalesced 16-byte load straight from L2; activations live in LDS; bias + activation are
// applied to the accumulators in registers.  Precision: bf16 operands with fp32 accumulation unless a float32 mode is chosen
// (a policy network, not part of the simulator's float64 path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#define EV2G_MLP_ROWS 32
#define EV2G_MLP_BLOCK 256

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MlpDev {
    int d_in, h1, h2, d_out;          // logical sizes
    int k1, n1, n2, n3;               // padded: k1 = ceil16(d_in), n1 = ceil32(h1) (= K of layer 2), n2 = ceil32(h2), n3 = ceil32(d_out)
    const uint16_t *w1, *w2, *w3;     // bf16, fragment order [n_tile][k_step][lane][8]
    const float *b1, *b2, *b3;        // padded with zeros
    float out_lo;                     // actions in [out_lo, 1]: tanh for -1, (tanh + 1) / 2 for 0
    unsigned long long *dbg;          // cycle stamps of workgroup 0 (EV2G_MLP_TIMING builds only), else nullptr
};
#ifdef EV2G_MLP_TIMING
#define MLP_STAMP(i) if (m.dbg && blockIdx.x == 0 && threadIdx.x == 0) m.dbg[i] = __builtin_readcyclecounter();
#else
#define MLP_STAMP(i)
#endif
// the inline float32 policy's k-step (ev2g_mlp3_inline_f32): 0 = the compiler's interleaving of LDS reads, weight requests and MFMAs; 1 = the five MFMAs as one
// back-to-back group behind one wait; 2 = 1 + the next k-step's operand terms read from LDS ahead of the group.  Measured at cfg2 (profiles/r06_fused_float32_policy.txt):
// 2 with a ring of 4 fragments (no scratch) +2.5 % over 0 with a ring of 8; 1 and 2 with deeper rings spill 20..68 bytes per lane and lose 1..2.5 %.
#ifndef EV2G_F32_GROUP
#define EV2G_F32_GROUP 2
#endif
// the bf16 inline policy (ev2g_mlp3_inline), layers in which a wavefront may hold two tiles: bit 0 = layer 1, bit 1 = layer 2 walk k-step, tile slot (one operand read, two
// independent MFMAs per k-step).  Measured at cfg2 (profiles/r06_fused_bf16_kstep_outer.txt): 0: 502 M, 1: 505 M, 2: 509 M, 3: 502 M env-steps/s -- the policy phase is bound
// by the weight stream (437 KB per workgroup and forward at ~51 B/clk/CU): what one layer gains the other gives back (stamps); layer 2 alone keeps +1.3 %.
#ifndef EV2G_BF16_KSOUTER
#define EV2G_BF16_KSOUTER 2
#endif
// tuning builds (-DEV2G_F32_STAMPS, tools/r6/f32_stamps.py): every wavefront of the first 8 workgroups stamps the layers of the inline float32 policy (the last forward's stay)
#ifdef EV2G_F32_STAMPS
#define F32_STAMP(i) if (m.dbg && blockIdx.x < 8 && (threadIdx.x & 63) == 0) m.dbg[((blockIdx.x * 16 + (threadIdx.x >> 6)) * 8) + i] = __builtin_readcyclecounter();
#define F32_STAMP2(i) if (m.dbg && blockIdx.x < 8 && (threadIdx.x & 63) == 0) m.dbg[1024 + ((blockIdx.x * 16 + (threadIdx.x >> 6)) * 16) + (i)] = __builtin_readcyclecounter();
#else
#define F32_STAMP(i)
#define F32_STAMP2(i)
#endif

__host__ __device__ inline int ev2g_mlp_lds_stride(int k) { return k + 8; }   // bf16 elements per LDS row: +16 bytes against bank conflicts
__host__ __device__ inline size_t ev2g_mlp_lds_bytes(const MlpDev &m) {
    const int a = ev2g_mlp_lds_stride(m.k1 > m.n2 ? m.k1 : m.n2), b = ev2g_mlp_lds_stride(m.n1);
    return (size_t)EV2G_MLP_ROWS * (a + b) * sizeof(uint16_t) + (size_t)(m.n1 + m.n2 + m.n3) * sizeof(float);   // + the staged biases
}

__device__ __forceinline__ uint16_t ev2g_f32_to_bf16(float f) {   // round to nearest even
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// two float32 -> two bf16 in one word (v_cvt_pk_bf16_f32: round to nearest even, like ev2g_f32_to_bf16 and the host's packing)
typedef __bf16 ev2g_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ev2g_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t ev2g_pack_bf16(float a, float b) {
    const ev2g_f32x2 f = {a, b};
    const ev2g_bf16x2 h = __builtin_convertvector(f, ev2g_bf16x2);
    uint32_t u;
    __builtin_memcpy(&u, &h, 4);
    return u;
}

// tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware exp2 / rcp (relative error ~1e-6, saturates cleanly at +-1)
__device__ __forceinline__ float ev2g_fast_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);   // exp(2x) = 2^(2x / ln 2)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// tanh of the float32 policies' output layer (EV2G_MLP_F32 / F32X3 in ev2g_mlp3_s16 and the fused launch's ev2g_mlp3_inline_f32: one function, so their actions stay
// bit-identical): t = exp(-2|x|) on the hardware exp2 (1 ulp), then (1 - t) / (1 + t) by an IEEE division -- absolute error below 1.5e-7 everywhere (near 0 the
// difference 1 - t carries t's rounding, 6e-8; beyond |x| = 9 the result is exactly 1).  The library tanhf it replaces cost the fused launch ~1.3 k cycles per
// step on the critical path of layer 3's four wavefronts (profiles/r06_fused_float32_policy.txt); EV2G_F32_TANH=0 builds keep it.
// the inline float32 policy's input rows: 1 = every wavefront splits ITS OWN row into the three bf16 terms at the policy's entry (22 VALU operations), into staging chunks
// that are dead in layer 1; 0 = every wavefront splits the operand fragments it reads (6 x 44 operations per wavefront and forward: layer 1 was VALU-issue-bound, 9.5 k cycles)
// The inline policies' weight requests through a SCALAR base: the wavefront's number read as a scalar (`tid >> 6` alone is a vector value to the compiler) makes the tile guards
// scalar branches and a request `global_load_dwordx4 v, v_lane16, s[base]` behind scalar adds instead of three vector instructions and an address register pair.  Measured at
// cfg2 (profiles/r06_fused_float32_policy.txt, r06_fused_bf16_kstep_outer.txt): float32 policy +1.4 % (324 -> 329 M: its layer 2 is MFMA / issue-bound), bf16 policy +0.2 %
// (weight-stream-bound: off).
#ifndef EV2G_S16_SADDR   // the same in the stand-alone streaming actor (ev2g_mlp3_s16)
#define EV2G_S16_SADDR 1
#endif
#ifndef EV2G_BF16_SADDR
#define EV2G_BF16_SADDR 0
#endif
#ifndef EV2G_F32_SADDR
#define EV2G_F32_SADDR 1
#endif
#ifndef EV2G_F32_XSPLIT
#define EV2G_F32_XSPLIT 1
#endif
#ifndef EV2G_F32_L1AHEAD   // layer 1 (two tiles' accumulators per wavefront): the next k-step's operand terms read ahead of the MFMA group (12 registers)
#define EV2G_F32_L1AHEAD 0
#endif
#ifndef EV2G_F32_TANH
#define EV2G_F32_TANH 1
#endif
__device__ __forceinline__ float ev2g_tanh_f32(float x) {
#if EV2G_F32_TANH
    const float t = __builtin_amdgcn_exp2f(fabsf(x) * -2.885390081777927f);   // exp(-2|x|)
    return copysignf((1.0f - t) / (1.0f + t), x);
#else
    return tanhf(x);
#endif
}

// Epilogue of one 32 x 32 output tile held in MFMA accumulators: bias, activation, and either the next layer's A matrix
// (bf16, LDS) or -- FINAL -- the float32 action rows.  C/D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
// `bias` may be an LDS copy (the fixed-shape kernel stages all biases at its start: no global round trip in an epilogue).
template <bool FINAL>
__device__ __forceinline__ void ev2g_mlp_tile_out(const f32x16 acc, int nt, const float *__restrict__ bias, uint16_t *__restrict__ out,
                                                  int so, float *__restrict__ gout, int row0, int n_rows, int d_out, float out_lo) {
    const int lane = threadIdx.x & 63;
    const int col = nt * 32 + (lane & 31);
    const float bv = bias[col];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float v = acc[r] + bv;
        if (FINAL) {
            v = ev2g_fast_tanh(v);
            if (out_lo == 0.0f) v = v * 0.5f + 0.5f;
            if (col < d_out && row0 + row < n_rows) gout[(size_t)(row0 + row) * d_out + col] = v;
        } else {
            out[row * so + col] = ev2g_f32_to_bf16(v > 0.0f ? v : 0.0f);
        }
    }
}

// ---- fixed shapes (K/16 of every layer known at compile time: the shipped configs) --------------------------------------
// A layer for one wavefront: output-column tiles of 32 round-robin over the four wavefronts, a tile = one chain of KS MFMAs
// on one accumulator.  What bounds the kernel is the weight stream (0.45 MB per workgroup through a 64 B/clk CU port and a
// shared L2: a fragment load takes ~1 us to come back when every workgroup streams at once), not the MFMAs -- so weight
// fragments are requested as far ahead as the register file allows (one 256-thread workgroup per CU leaves each wavefront
// all 512 registers): a layer keeps NB tiles of fragments in flight, its first NB tiles are requested before the layer
// before it starts computing (MlpFrags::first), and a buffer is re-requested as soon as its tile is consumed.  The A
// fragments (the activations) do not depend on the tile: read from LDS once per layer.
template <int KS, int NB> struct MlpFrags {
    uint4 b[NB][KS];
    int rot_t, ksr[KS];
    const uint4 *wl;
    int NT;
    // Every workgroup streams the SAME weights; in lockstep they would all ask the same L2 channel for the same line at the
    // same time.  Each workgroup therefore walks the tiles and the K steps in its own rotation (sums in a different order:
    // fp32 accumulation, differences at rounding level).
    __device__ __forceinline__ int tile_of(int it) const { return (min(it, NT - 1) + rot_t) % NT; }   // clamped: past the end re-reads a valid tile
    __device__ __forceinline__ uint4 frag(int tile, int u) const { return wl[(unsigned)((tile * KS + ksr[u]) * 64)]; }
    __device__ __forceinline__ void request(int I, int it) {   // (I is a constant wherever this is called, after unrolling)
        const int tile = tile_of(it);
#pragma unroll
        for (int u = 0; u < KS; u++) b[I][u] = frag(tile, u);
    }
    __device__ __forceinline__ void first(const uint16_t *W, int N) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        NT = N >> 5;
        wl = (const uint4 *)W + lane;
        rot_t = (int)(blockIdx.x % (unsigned)NT);
        const int rot_k = (int)((blockIdx.x * 5u) % (unsigned)KS);
#pragma unroll
        for (int u = 0; u < KS; u++) { ksr[u] = u + rot_k; if (ksr[u] >= KS) ksr[u] -= KS; }
        constexpr int W4 = EV2G_MLP_BLOCK / 64;
#pragma unroll
        for (int i = 0; i < NB; i++) request(i, wave + i * W4);
    }
};

// MAXT: the most tiles one wavefront can own (ceil(NT / 4)); the tile loop is unrolled over it so that buffer indices are static
template <bool FINAL, int KS, int NB, int MAXT>
__device__ __forceinline__ void ev2g_mlp_layer_fixed(MlpFrags<KS, NB> &F, const uint16_t *__restrict__ A, int sa, const float *__restrict__ bias,
                                                     uint16_t *__restrict__ out, int so, float *__restrict__ gout, int row0, int n_rows,
                                                     int d_out, float out_lo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NT = F.NT;
    constexpr int W4 = EV2G_MLP_BLOCK / 64;
    const uint16_t *arow = A + (lane & 31) * sa + (lane >> 5) * 8;
    uint4 afr[KS];
#pragma unroll
    for (int u = 0; u < KS; u++) afr[u] = *(const uint4 *)(arow + F.ksr[u] * 16);
#pragma unroll
    for (int i = 0; i < MAXT; i++) {
        const int it = wave + i * W4;
        if (it < NT) {   // (uniform)
            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < KS; u++) {
                bf16x8 a, bq;
                __builtin_memcpy(&a, &afr[u], 16); __builtin_memcpy(&bq, &F.b[i % NB][u], 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq, acc, 0, 0, 0);
            }
            if (i + NB < MAXT && it + NB * W4 < NT) F.request(i % NB, it + NB * W4);   // this buffer is free again: the tile NB ahead
            ev2g_mlp_tile_out<FINAL>(acc, F.tile_of(it), bias, out, so, gout, row0, n_rows, d_out, out_lo);
        }
    }
}

// The same for any K (shapes other than the shipped ones): a ring of EV2G_MLP_DEPTH weight fragments per lane stays in flight
// ahead of the MFMAs, refilled slot by slot.
#define EV2G_MLP_DEPTH 12
template <bool FINAL>
__device__ __forceinline__ void ev2g_mlp_layer_any(const uint16_t *__restrict__ A, int sa, int K, int N, const uint16_t *__restrict__ W,
                                                   const float *__restrict__ bias, uint16_t *__restrict__ out, int so,
                                                   float *__restrict__ gout, int row0, int n_rows, int d_out, float out_lo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int KS = K >> 4, NT = N >> 5;
    const uint16_t *arow = A + (lane & 31) * sa + (lane >> 5) * 8;
    for (int nt = wave; nt < NT; nt += EV2G_MLP_BLOCK / 64) {
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const uint4 *w = (const uint4 *)W + ((size_t)nt * KS) * 64 + lane;
        uint4 ring[EV2G_MLP_DEPTH];
#pragma unroll
        for (int u = 0; u < EV2G_MLP_DEPTH; u++) ring[u] = w[(size_t)min(u, KS - 1) * 64];
        for (int ks0 = 0; ks0 < KS; ks0 += EV2G_MLP_DEPTH) {
#pragma unroll
            for (int u = 0; u < EV2G_MLP_DEPTH; u++) {
                const int ks = ks0 + u;
                if (ks < KS) {   // (uniform)
                    const uint4 aq = *(const uint4 *)(arow + ks * 16);
                    bf16x8 a, b;
                    __builtin_memcpy(&a, &aq, 16); __builtin_memcpy(&b, &ring[u], 16);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                    ring[u] = w[(size_t)min(ks + EV2G_MLP_DEPTH, KS - 1) * 64];
                }
            }
        }
        ev2g_mlp_tile_out<FINAL>(acc, nt, bias, out, so, gout, row0, n_rows, d_out, out_lo);
    }
}

// input rows -> bf16 A matrix in LDS (zero-padded in K and past the last row).  The 32 rows are one contiguous float32 range:
// every lane issues its loads back to back (no dependent round trips), then converts and scatters into the padded rows.
__device__ __forceinline__ void ev2g_mlp_stage_input(const MlpDev &m, const float *__restrict__ x, int row0, int n_rows, uint16_t *bufA, int sA) {
    const int nr = min(EV2G_MLP_ROWS, n_rows - row0), total = nr * m.d_in;
    const float *xs = x + (size_t)row0 * m.d_in;
    const float rdin = 1.0f / (float)m.d_in;
    constexpr int NL = 6;   // 6 x 256 lanes x 4 floats = 6144 >= 32 rows x 192 columns: one pass for the shipped shapes
    const bool vec = (((size_t)xs) & 15) == 0;   // (uniform) the range starts 16-byte aligned: 16-byte loads
    for (int base = 0; base < total; base += NL * EV2G_MLP_BLOCK * 4) {
        float4 v[NL];
#pragma unroll
        for (int it = 0; it < NL; it++) {
            const int f = base + (it * EV2G_MLP_BLOCK + (int)threadIdx.x) * 4;
            if (vec && f + 3 < total) v[it] = *(const float4 *)(xs + f);
            else {
                v[it].x = f < total ? xs[f] : 0.f; v[it].y = f + 1 < total ? xs[f + 1] : 0.f;
                v[it].z = f + 2 < total ? xs[f + 2] : 0.f; v[it].w = f + 3 < total ? xs[f + 3] : 0.f;
            }
        }
#pragma unroll
        for (int it = 0; it < NL; it++) {
            const int f = base + (it * EV2G_MLP_BLOCK + (int)threadIdx.x) * 4;
            const float e[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
            int r = (int)((float)f * rdin), c = f - r * m.d_in;   // exact for f < 2^23 after the one-step correction
            if (c < 0) { r--; c += m.d_in; } else if (c >= m.d_in) { r++; c -= m.d_in; }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (f + j < total) bufA[r * sA + c] = ev2g_f32_to_bf16(e[j]);
                if (++c == m.d_in) { c = 0; r++; }
            }
        }
    }
    // zero padding: columns d_in..k1 of every row, and the rows past n_rows
    const int padc = m.k1 - m.d_in;
    for (int i = threadIdx.x; i < EV2G_MLP_ROWS * padc; i += EV2G_MLP_BLOCK) { const int r = i / padc; bufA[r * sA + m.d_in + (i - r * padc)] = 0; }
    for (int i = threadIdx.x; i < (EV2G_MLP_ROWS - nr) * m.d_in; i += EV2G_MLP_BLOCK) { const int r = nr + i / m.d_in; bufA[r * sA + (i - (r - nr) * m.d_in)] = 0; }
}

// The same in two halves for the fixed-shape kernel (one pass: 32 rows x d_in <= EV2G_MLP_NL x 256 lanes x 4 floats, checked by the
// host when it picks that kernel): request the rows, ... , convert and scatter them.
#define EV2G_MLP_NL 6
__device__ __forceinline__ void ev2g_mlp_input_request(const MlpDev &m, const float *__restrict__ x, int row0, int n_rows, float4 (&v)[EV2G_MLP_NL]) {
    const int nr = min(EV2G_MLP_ROWS, n_rows - row0), total = nr * m.d_in;
    const float *xs = x + (size_t)row0 * m.d_in;
    const bool vec = (((size_t)xs) & 15) == 0;
#pragma unroll
    for (int it = 0; it < EV2G_MLP_NL; it++) {
        const int f = (it * EV2G_MLP_BLOCK + (int)threadIdx.x) * 4;
        if (vec && f + 3 < total) v[it] = *(const float4 *)(xs + f);
        else {
            v[it].x = f < total ? xs[f] : 0.f; v[it].y = f + 1 < total ? xs[f + 1] : 0.f;
            v[it].z = f + 2 < total ? xs[f + 2] : 0.f; v[it].w = f + 3 < total ? xs[f + 3] : 0.f;
        }
    }
}
__device__ __forceinline__ void ev2g_mlp_input_store(const MlpDev &m, int row0, int n_rows, const float4 (&v)[EV2G_MLP_NL], uint16_t *bufA, int sA) {
    const int nr = min(EV2G_MLP_ROWS, n_rows - row0), total = nr * m.d_in;
    const float rdin = 1.0f / (float)m.d_in;
#pragma unroll
    for (int it = 0; it < EV2G_MLP_NL; it++) {
        const int f = (it * EV2G_MLP_BLOCK + (int)threadIdx.x) * 4;
        const float e[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
        int r = (int)((float)f * rdin), c = f - r * m.d_in;
        if (c < 0) { r--; c += m.d_in; } else if (c >= m.d_in) { r++; c -= m.d_in; }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (f + j < total) bufA[r * sA + c] = ev2g_f32_to_bf16(e[j]);
            if (++c == m.d_in) { c = 0; r++; }
        }
    }
    const int padc = m.k1 - m.d_in;
    for (int i = threadIdx.x; i < EV2G_MLP_ROWS * padc; i += EV2G_MLP_BLOCK) { const int r = i / padc; bufA[r * sA + m.d_in + (i - r * padc)] = 0; }
    for (int i = threadIdx.x; i < (EV2G_MLP_ROWS - nr) * m.d_in; i += EV2G_MLP_BLOCK) { const int r = nr + i / m.d_in; bufA[r * sA + (i - (r - nr) * m.d_in)] = 0; }
}

// LDS: bufA [32][sA] bf16 (input, then layer-2 output) | bufB [32][sB] bf16 (layer-1 output) | biases (fixed-shape kernel)
template <int KS1, int KS2, int KS3>
__global__ void __launch_bounds__(EV2G_MLP_BLOCK) ev2g_mlp3_fixed(MlpDev m, const float *__restrict__ x, float *__restrict__ y, int n_rows) {
    extern __shared__ __attribute__((aligned(16))) uint16_t mlds[];
    const int sA = ev2g_mlp_lds_stride(m.k1 > m.n2 ? m.k1 : m.n2), sB = ev2g_mlp_lds_stride(m.n1);
    uint16_t *bufA = mlds, *bufB = mlds + EV2G_MLP_ROWS * sA;
    float *lb1 = (float *)(bufB + EV2G_MLP_ROWS * sB), *lb2 = lb1 + m.n1, *lb3 = lb2 + m.n2;
    const int row0 = blockIdx.x * EV2G_MLP_ROWS;
    MLP_STAMP(0)
    // register budget per lane (512): F1 4 x KS1 x 4 | F2 2 x KS2 x 4 requested while layer 1 runs | F3 KS3 x 4 while layer 2 runs
    MlpFrags<KS1, 4> F1;
    MlpFrags<KS2, 2> F2;
    // vmcnt retires in order: the input rows and the biases are requested FIRST, so that their conversion into LDS overlaps the arrival
    // of the weight fragments instead of waiting behind all of them
    float4 xin[EV2G_MLP_NL];
    ev2g_mlp_input_request(m, x, row0, n_rows, xin);
    float bv[4];   // n1 + n2 + n3 <= 416 + 320 + 128 <= 4 x 256 for the fixed shapes
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = (int)threadIdx.x + j * EV2G_MLP_BLOCK;
        bv[j] = i < m.n1 ? m.b1[i] : (i < m.n1 + m.n2 ? m.b2[i - m.n1] : (i < m.n1 + m.n2 + m.n3 ? m.b3[i - m.n1 - m.n2] : 0.f));
    }
    F1.first(m.w1, m.n1);   // ALL of this wavefront's layer-1 weight tiles
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = (int)threadIdx.x + j * EV2G_MLP_BLOCK;
        if (i < m.n1 + m.n2 + m.n3) lb1[i] = bv[j];
    }
    ev2g_mlp_input_store(m, row0, n_rows, xin, bufA, sA);
    MLP_STAMP(1)
    F2.first(m.w2, m.n2);   // layer 2's first two tiles: in flight while layer 1 computes
    __syncthreads();
    MLP_STAMP(2)
    ev2g_mlp_layer_fixed<false, KS1, 4, 4>(F1, bufA, sA, lb1, bufB, sB, nullptr, row0, n_rows, 0, 0.f);
    MLP_STAMP(3)
    MlpFrags<KS3, 1> F3;
    F3.first(m.w3, m.n3);
    __syncthreads();
    MLP_STAMP(4)
    ev2g_mlp_layer_fixed<false, KS2, 2, 3>(F2, bufB, sB, lb2, bufA, sA, nullptr, row0, n_rows, 0, 0.f);
    MLP_STAMP(5)
    __syncthreads();
    MLP_STAMP(6)
    ev2g_mlp_layer_fixed<true, KS3, 1, 1>(F3, bufA, sA, lb3, nullptr, 0, y, row0, n_rows, m.d_out, m.out_lo);
    MLP_STAMP(7)
}

// ---- fixed shapes, round 4: 16-row workgroups on every CU, transposed product, one continuous weight stream ------------------------
// What a forward costs is streaming the 0.44 MB of weights into each CU (64 B/clk per CU: ~3 us, `tools/micro/weight_stream.hip` -- the same
// whether 64, 128 or 256 workgroups do it) plus whatever of the rest is NOT hidden under that stream.  Round 2-3's kernel (ev2g_mlp3_fixed) hid
// little: 28 k cycles against the stream's 7 k.  This one is built around the stream:
//   * a workgroup owns 16 env rows (256 workgroups at 4096 envs: every CU; half the input rows to fetch cold, half the MFMA and epilogue
//     work on the chain behind the stream);
//   * v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A operand and the activations as B: D[m][n] = sum_k W[n0 + m][k] X[n][k], so a lane
//     ends up with four CONSECUTIVE output columns of one env row -- the next layer's operand is written with one 8-byte LDS store per
//     tile (the 32x32 form left 16 two-byte stores), the actions with two 8-byte global stores;
//   * column tiles of 16 go round-robin over the four wavefronts (25 / 19 / 4 tiles: 7 + 5 + 1 per wavefront at most); the bias is the
//     accumulator's initial value; two accumulators per tile (even / odd k-steps) so that no MFMA waits for the one before it;
//   * every wavefront walks ONE static sequence of weight fragments -- layer 1's tiles, layer 2's, layer 3's, in the order it consumes
//     them -- through a ring of EV2G_MLPS_RING register slots: the first RING fragments are requested at kernel start (before the input
//     rows have arrived), and fragment s + RING is requested the moment fragment s has been consumed, across layer boundaries and barriers:
//     the stream never waits for the compute, only the other way round.
// Weights are packed per layer as [tile][k-step][lane] 16-byte fragments: lane l holds W[tile*16 + (l & 15)][ks*32 + 8*(l >> 4) + 0..7].
#define EV2G_MLPS_ROWS 16
#define EV2G_MLPS_RING 52
#ifndef EV2G_MLPS_HEAD
#define EV2G_MLPS_HEAD 28   // fragments requested before the input rows are converted
#endif
typedef float f32x4m __attribute__((ext_vector_type(4)));

// NW > 1: the FLOAT32 network on the bf16 matrix cores (precision = EV2G_MLP_F32).  A float32 weight is stored as NW bf16 terms
// (w = w0 + w1 (+ w2), each the bf16 rounding of what the ones before it left: 16 or 24 significant bits), an activation is split into three
// terms when it is written to LDS, and a k-step accumulates the products w_i x_j with i + j <= 2 (5 MFMAs for NW = 2, 6 for NW = 3) into the
// float32 accumulator: every product of two bf16 values is exact in float32, so what is lost is the dropped terms (2^-24 relative and
// below for NW = 3: the same level as float32 operands) and the accumulation order.  The float32 MFMA (v_mfma_f32_32x32x2_f32, ev2g_mlp3_f32)
// runs at 1/16 of the bf16 rate; here the cost is the weight stream, NW times the bf16 kernel's.
template <int KS1, int NT1, int NT2, int NT3, int NW = 1, int WV = 4, int RB = 1> struct MlpS16 {   // WV: wavefronts per workgroup (4: one per SIMD; 8: two); RB: blocks of 16 env rows per workgroup
    static constexpr int ROWS = EV2G_MLPS_ROWS * RB;
    static constexpr int NX = NW == 1 ? 1 : 3;   // terms of an activation
    static constexpr int KS2 = (NT1 * 16 + 31) / 32, KS3 = (NT2 * 16 + 31) / 32;
    static constexpr int NTH = WV * 64;
    static constexpr int MT1 = (NT1 + WV - 1) / WV, MT2 = (NT2 + WV - 1) / WV, MT3 = (NT3 + WV - 1) / WV;   // tile slots per wavefront
    static constexpr int S1 = MT1 * KS1 * NW, S2 = MT2 * KS2 * NW, S3 = MT3 * KS3 * NW, STOT = S1 + S2 + S3;   // fragments of the sequence, per layer
    static constexpr int SX = KS1 * 32 + 8, SH1 = KS2 * 32 + 8, SH2 = KS3 * 32 + 8;             // LDS row strides (bf16 elements; +16 bytes against bank conflicts)
    static constexpr int NB = (NT1 + NT2 + NT3) * 16;                                          // staged biases (floats)
    static constexpr int RING = (NW == 1 ? EV2G_MLPS_RING : 36) * 4 / WV;                                 // (three operand copies per k-step take the registers)
    static constexpr size_t lds_bytes = (size_t)ROWS * (SX + SH1 + SH2) * 2 * NX + (size_t)NB * 4;
};

// float32 -> NX bf16 terms, two values at a time (packed words); term k is the bf16 rounding of what terms 0..k-1 left
template <int NX> __device__ __forceinline__ void ev2g_split_bf16(float a, float b, uint32_t (&w)[NX]) {
#pragma unroll
    for (int k = 0; k < NX; k++) {
        w[k] = ev2g_pack_bf16(a, b);
        if (k + 1 < NX) { a -= __uint_as_float(w[k] << 16); b -= __uint_as_float(w[k] & 0xffff0000u); }   // (exact: the term is a prefix of the value's bits)
    }
}

template <int KS1, int NT1, int NT2, int NT3, int NW = 1, int WV = 4, int RB = 1>
__global__ void __launch_bounds__(WV * 64) ev2g_mlp3_s16(MlpDev m, const float *__restrict__ x, float *__restrict__ y, int n_rows) {
    typedef MlpS16<KS1, NT1, NT2, NT3, NW, WV, RB> C;
    constexpr int ROWS = C::ROWS;
    constexpr int NTH = C::NTH;
    constexpr int RING = C::RING, NX = C::NX;
    extern __shared__ __attribute__((aligned(16))) uint16_t mlds[];
    // operand buffers: NX copies (terms) of each, one behind the other
    constexpr int BX = ROWS * C::SX, BH1 = ROWS * C::SH1, BH2 = ROWS * C::SH2;
    uint16_t *bufX = mlds, *bufH1 = bufX + NX * BX, *bufH2 = bufH1 + NX * BH1;
    float *lb = (float *)(bufH2 + NX * BH2);   // biases: layer 1 | layer 2 | layer 3
#if EV2G_S16_SADDR
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane((tid >> 6) & (WV - 1));   // (uniform by construction: a scalar for the tile guards and the weight bases)
#else
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & (WV - 1);
#endif
    const int row0 = blockIdx.x * ROWS;
    const int nr = min(ROWS, n_rows - row0);
    MLP_STAMP(0)
    // ---- requests, oldest first (vmcnt retires in order): input rows, biases, then the head of the weight sequence ----
    const int d_in = m.d_in, total = nr * d_in;
    const float *xs = x + (size_t)row0 * d_in;
    constexpr int NL2 = (ROWS * KS1 * 32 / 2 + NTH - 1) / NTH;   // 8-byte pieces per lane (16 rows of at most KS1*32 columns)
    const bool pairs = (d_in & 1) == 0 && (((size_t)xs) & 7) == 0;    // (uniform) an even row length: two neighbours never straddle a row
    // (unconditional loads from clamped addresses: a load inside a per-lane branch whose result merges with a default makes the compiler
    // drain vmcnt before the next one -- six round trips in a row instead of one)
    float2 xin[NL2];
    if (pairs) {
#pragma unroll
        for (int it = 0; it < NL2; it++) xin[it] = *(const float2 *)(xs + min((it * NTH + tid) * 2, total - 2));
    } else {
#pragma unroll
        for (int it = 0; it < NL2; it++) { const int f = (it * NTH + tid) * 2; xin[it].x = xs[min(f, total - 1)]; xin[it].y = xs[min(f + 1, total - 1)]; }
    }
    float bv[(C::NB + NTH - 1) / NTH];   // the three bias vectors are ONE array on this path (ev2g_mlp_create_ex): b1 | b2 | b3, each padded to its tiles
#pragma unroll
    for (int j = 0; j < (C::NB + NTH - 1) / NTH; j++) bv[j] = m.b1[min(tid + j * NTH, C::NB - 1)];
#if EV2G_S16_SADDR
    const unsigned lane16 = (unsigned)lane * 16u;
    typedef const char __attribute__((address_space(1))) *wgptr;
    typedef unsigned u32x4g __attribute__((ext_vector_type(4)));
    const wgptr w1 = (wgptr)(unsigned long long)m.w1, w2 = (wgptr)(unsigned long long)m.w2, w3 = (wgptr)(unsigned long long)m.w3;
    auto frag = [&](wgptr w, int idx) __attribute__((always_inline)) -> uint4 {   // (ev2g_mlp3_inline_f32: scalar base + zero-extended lane offset = the `saddr` form)
        wgptr fb = w + (unsigned long long)(unsigned)idx * 1024ull;
        unsigned l16 = lane16;
        asm volatile("" : "+s"(fb), "+v"(l16));
        const u32x4g v = *(const u32x4g __attribute__((address_space(1))) *)(fb + l16);
        uint4 r;
        __builtin_memcpy(&r, &v, 16);
        return r;
    };
#else
    const uint4 *w1 = (const uint4 *)m.w1 + lane, *w2 = (const uint4 *)m.w2 + lane, *w3 = (const uint4 *)m.w3 + lane;
    auto frag = [&](const uint4 *w, int idx) __attribute__((always_inline)) -> uint4 { return w[(unsigned)(idx * 64)]; };
#endif
    uint4 ring[RING];
    // fragment `seq` of this wavefront's sequence -> ring slot seq % RING (seq is a constant wherever this is called, after unrolling; the
    // tile guard is a compile-time `true` except in a layer's last tile slot).  Sequence order inside a layer: tile slot, k-step, weight term.
    auto request = [&](int seq) __attribute__((always_inline)) {
        if (seq >= C::STOT) return;
        const int L = seq < C::S1 ? 0 : (seq < C::S1 + C::S2 ? 1 : 2);
        const int r = seq - (L == 0 ? 0 : (L == 1 ? C::S1 : C::S1 + C::S2));
        const int KS = L == 0 ? KS1 : (L == 1 ? C::KS2 : C::KS3), NT = L == 0 ? NT1 : (L == 1 ? NT2 : NT3);
        const int i = r / (KS * NW), rem = r - i * (KS * NW);   // rem = ks * NW + term: the fragment's place inside its tile
        const auto w = L == 0 ? w1 : (L == 1 ? w2 : w3);
        if (WV * i + WV - 1 < NT || wave + WV * i < NT) ring[seq % RING] = frag(w, (wave + WV * i) * (KS * NW) + rem);
    };
    MLP_STAMP(8)
    // The CU's vector-memory port takes ~64 cycles per wavefront and fragment with four wavefronts asking (3.3 k cycles for the whole ring):
    // the input rows arrive while the first fragments are being requested.  HEAD of them go out first, then the rows are converted (the
    // port works the queue off meanwhile), the rest of the ring between the conversion steps; padding and biases come last.
    constexpr int HEAD = (EV2G_MLPS_HEAD * 4 / WV) < RING ? (EV2G_MLPS_HEAD * 4 / WV) : RING;
    constexpr int PER = (RING - HEAD + NL2 - 1) / NL2;   // requests per conversion step
#pragma unroll
    for (int sq = 0; sq < HEAD; sq++) request(sq);
    MLP_STAMP(9)
    // ---- input rows -> bf16 operand rows in LDS ----
    if (pairs) {   // element pair p = it * NTH + tid sits at (row, column) = divmod(2 p, d_in): one division, then steps of 2 NTH elements
        const int q512 = (2 * NTH) / d_in, r512 = 2 * NTH - q512 * d_in;   // (uniform)
        int f = tid * 2;
        int r = (int)((float)f * (1.0f / (float)d_in)), c = f - r * d_in;   // exact after the one-step correction (f < 2^23)
        if (c < 0) { r--; c += d_in; } else if (c >= d_in) { r++; c -= d_in; }
#pragma unroll
        for (int it = 0; it < NL2; it++) {
            uint32_t wd[NX];
            ev2g_split_bf16<NX>(xin[it].x, xin[it].y, wd);
            if (f < total) {
#pragma unroll
                for (int k = 0; k < NX; k++) *(uint32_t *)(bufX + k * BX + r * C::SX + c) = wd[k];
            }
            f += 2 * NTH; r += q512; c += r512;
            if (c >= d_in) { c -= d_in; r++; }
            // the rest of the ring goes out BETWEEN the conversion steps: a request blocks its wavefront while the port is busy, the conversion
            // of the other wavefronts fills that time (and their requests, this one's conversion)
#pragma unroll
            for (int u = 0; u < PER; u++) request(HEAD + it * PER + u < RING ? HEAD + it * PER + u : C::STOT);
        }
    } else {
        const float rdin = 1.0f / (float)d_in;
#pragma unroll
        for (int it = 0; it < NL2; it++) {
            const int f = (it * NTH + tid) * 2;
            int r = (int)((float)f * rdin), c = f - r * d_in;
            if (c < 0) { r--; c += d_in; } else if (c >= d_in) { r++; c -= d_in; }
            uint32_t wd[NX];
            ev2g_split_bf16<NX>(xin[it].x, xin[it].y, wd);
#pragma unroll
            for (int k = 0; k < NX; k++) if (f < total) bufX[k * BX + r * C::SX + c] = (uint16_t)wd[k];
            if (++c == d_in) { c = 0; r++; }
#pragma unroll
            for (int k = 0; k < NX; k++) if (f + 1 < total) bufX[k * BX + r * C::SX + c] = (uint16_t)(wd[k] >> 16);
        }
    }
    MLP_STAMP(10)
    if (!pairs) {
#pragma unroll
        for (int sq = HEAD; sq < RING; sq++) request(sq);
    }
    if (tid < 256) {   // zero padding: thread (row = tid / 16, j = tid % 16) clears columns j, j + 16, ... of its row's tail in every operand buffer
        const int pj = tid & 15;   // (the first 256 threads)
#pragma unroll
        for (int rbp = 0; rbp < RB; rbp++) {
        const int pr = ((tid >> 4) & 15) + 16 * rbp;
        constexpr int P1 = C::KS2 * 32 - NT1 * 16, P2 = C::KS3 * 32 - NT2 * 16;                       // columns no tile writes
        static_assert(P1 <= 16 && P2 <= 16, "tail columns");
#pragma unroll
        for (int k = 0; k < NX; k++) {
            for (int cc = d_in + pj; cc < KS1 * 32; cc += 16) bufX[k * BX + pr * C::SX + cc] = 0;             // columns d_in .. KS1*32
            if (pr >= nr) for (int cc = pj; cc < d_in; cc += 16) bufX[k * BX + pr * C::SX + cc] = 0;         // rows past the batch
            if (pj < P1) bufH1[k * BH1 + pr * C::SH1 + NT1 * 16 + pj] = 0;
            if (pj < P2) bufH2[k * BH2 + pr * C::SH2 + NT2 * 16 + pj] = 0;
        }
        }
    }
#pragma unroll
    for (int j = 0; j < (C::NB + NTH - 1) / NTH; j++) { const int i = tid + j * NTH; if (i < C::NB) lb[i] = bv[j]; }
    MLP_STAMP(1)
    __syncthreads();
    MLP_STAMP(2)
    const int brow = (lane & 15), kq = (lane >> 4);
    // one layer for this wavefront: operand fragments of the 16 rows from LDS (once), then its tile slots
    auto layer = [&](auto Lc, const uint16_t *A, int sa, int ba, const float *bias, uint16_t *out, int so, int bo) __attribute__((always_inline)) {
        constexpr int L = decltype(Lc)::value;
        constexpr int KS = L == 0 ? KS1 : (L == 1 ? C::KS2 : C::KS3), NT = L == 0 ? NT1 : (L == 1 ? NT2 : NT3), MT = (NT + WV - 1) / WV;
        constexpr int base = L == 0 ? 0 : (L == 1 ? C::S1 : C::S1 + C::S2);
        uint4 bfr[RB][NX][KS];
#pragma unroll
        for (int rb = 0; rb < RB; rb++)
#pragma unroll
            for (int k = 0; k < NX; k++)
#pragma unroll
                for (int ks = 0; ks < KS; ks++) bfr[rb][k][ks] = *(const uint4 *)(A + k * ba + (rb * 16 + brow) * sa + ks * 32 + kq * 8);
        f32x4m bini[MT];
#pragma unroll
        for (int i = 0; i < MT; i++) bini[i] = *(const f32x4m *)(bias + min(wave + WV * i, NT - 1) * 16 + kq * 4);
#pragma unroll
        for (int i = 0; i < MT; i++) {
            const int tile = wave + WV * i;
            if (WV * i + WV - 1 < NT || tile < NT) {   // (uniform; a constant but for the last slot)
                f32x4m acc0[RB], acc1[RB];
#pragma unroll
                for (int rb = 0; rb < RB; rb++) { acc0[rb] = bini[i]; acc1[rb] = f32x4m{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
#pragma unroll
                    for (int p = NW - 1; p >= 0; p--) {   // (the small terms first)
                        const int sq = base + (i * KS + ks) * NW + p;
                        bf16x8 a;
                        __builtin_memcpy(&a, &ring[sq % RING], 16);
#pragma unroll
                        for (int xq = NX - 1; xq >= 0; xq--) {
                            if (NW == 1 || p + xq <= 2) {
#pragma unroll
                                for (int rb = 0; rb < RB; rb++) {   // (one weight fragment, RB blocks of rows)
                                    bf16x8 b;
                                    __builtin_memcpy(&b, &bfr[rb][xq][ks], 16);
                                    if ((ks + p + xq) & 1) acc1[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1[rb], 0, 0, 0);
                                    else acc0[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0[rb], 0, 0, 0);
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int p = 0; p < NW; p++) request(base + (i * KS + ks) * NW + p + RING);   // these slots are free again
                }
                const int col = tile * 16 + kq * 4;   // this lane: columns col .. col + 3 of env rows `brow`, `brow + 16` ...
#pragma unroll
                for (int rb = 0; rb < RB; rb++) {
                const f32x4m acc = acc0[rb] + acc1[rb];
                const int erow = rb * 16 + brow;
                if (L < 2) {
                    uint32_t lo[NX], hi[NX];
                    ev2g_split_bf16<NX>(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), lo);
                    ev2g_split_bf16<NX>(fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f), hi);
#pragma unroll
                    for (int k = 0; k < NX; k++) *(uint2 *)(out + k * bo + erow * so + col) = make_uint2(lo[k], hi[k]);
                } else {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) { v[r] = NW == 1 ? ev2g_fast_tanh(acc[r]) : ev2g_tanh_f32(acc[r]); if (m.out_lo == 0.0f) v[r] = v[r] * 0.5f + 0.5f; }
                    const int d_out = m.d_out;
                    if (erow < nr) {
                        float *yr = y + (size_t)(row0 + erow) * d_out + col;
                        if ((d_out & 1) == 0) {   // (uniform) even row length: the pairs are 8-byte aligned
                            if (col + 1 < d_out) *(float2 *)yr = make_float2(v[0], v[1]);
                            if (col + 3 < d_out) *(float2 *)(yr + 2) = make_float2(v[2], v[3]);
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; r++) if (col + r < d_out) yr[r] = v[r];
                        }
                    }
                }
                }
            } else {
#pragma unroll
                for (int u = 0; u < KS * NW; u++) request(base + i * KS * NW + u + RING);   // no tile in this slot: the sequence moves on all the same
            }
        }
    };
    layer(std::integral_constant<int, 0>{}, bufX, C::SX, BX, lb, bufH1, C::SH1, BH1);
    MLP_STAMP(3)
    __syncthreads();
    MLP_STAMP(4)
    layer(std::integral_constant<int, 1>{}, bufH1, C::SH1, BH1, lb + NT1 * 16, bufH2, C::SH2, BH2);
    MLP_STAMP(5)
    __syncthreads();
    MLP_STAMP(6)
    layer(std::integral_constant<int, 2>{}, bufH2, C::SH2, BH2, lb + (NT1 + NT2) * 16, nullptr, 0, 0);
    MLP_STAMP(7)
}

// ---- the same network as a DEVICE FUNCTION of a 16-wavefront workgroup that also steps the 16 envs whose rows these are (ev2g_step_wave<.., 1024,
// true>, ev2g_step_wave.h; round 5: one launch per rollout segment instead of two per step).  Same tiles, same fragment packing, same MFMA chain per
// tile (two accumulators by k-step parity, bias as the first one's initial value, ReLU / tanh epilogues) as ev2g_mlp3_s16: the actions are bit-identical
// to that kernel's.  What differs is where things live: the input rows are already bf16 in LDS (the step kernel writes every observation column there
// next to its global store), the actions also go to LDS (the step's phase A reads them), the hidden activations use LDS that the step leaves free
// between two steps, and a wavefront holds one tile's fragments at a time (128 registers per lane with four wavefronts per SIMD).
//   X [16][SX] bf16 -> H1 [16][SH1] -> H2 [16][SH2] -> act [16][64] float (LDS) + y rows (global, d_out floats each, rows < nr)
// barrier that orders LDS traffic only (the weight requests in flight and the action stores need no ordering inside the workgroup)
__device__ __forceinline__ void ev2g_mlp_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Every wavefront walks ONE static sequence of weight fragments -- its tiles of layer 1, of layer 2, of layer 3 -- through a ring of RING register
// slots: the head of the sequence is requested before the first barrier, fragment s + RING the moment fragment s has been consumed, across tile and
// layer boundaries and barriers (weights do not depend on activations), so the CU's vector-memory port works from the first cycle on and only the
// compute waits (the scheme of ev2g_mlp3_s16; a wavefront here holds 13 fragments instead of 26).  `act` rows are `as` floats apart.
// RB (round 6): blocks of 16 env rows per workgroup -- a weight fragment feeds RB MFMAs (rows 0..15, 16..31), each row's chain as in the one-block form.
template <int KS1, int NT1, int NT2, int NT3, int WVS, int RING, int RB = 1>
__device__ __forceinline__ void ev2g_mlp3_inline(const MlpDev &m, const uint16_t *bufX, uint16_t *bufH1, uint16_t *bufH2, const float *lb, float *act, int as, float *y, int nr, int tid) {
    typedef MlpS16<KS1, NT1, NT2, NT3, 1, 4, RB> C;
    constexpr int KS2 = C::KS2, KS3 = C::KS3;
    constexpr int MT1 = (NT1 + WVS - 1) / WVS, MT2 = (NT2 + WVS - 1) / WVS, MT3 = (NT3 + WVS - 1) / WVS;
    constexpr int S1 = MT1 * KS1, S2 = MT2 * KS2, S3 = MT3 * KS3, STOT = S1 + S2 + S3;
#if EV2G_BF16_SADDR   // (as in ev2g_mlp3_inline_f32 below: the wavefront's number as a scalar, weight requests through a scalar base)
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)lane * 16u;
    typedef const char __attribute__((address_space(1))) *wgptr;
    typedef unsigned u32x4g __attribute__((ext_vector_type(4)));
    const wgptr w1 = (wgptr)(unsigned long long)m.w1, w2 = (wgptr)(unsigned long long)m.w2, w3 = (wgptr)(unsigned long long)m.w3;
    auto frag = [&](wgptr w, int idx) __attribute__((always_inline)) -> uint4 {
        wgptr fb = w + (unsigned long long)(unsigned)idx * 1024ull;
        unsigned l16 = lane16;
        asm volatile("" : "+s"(fb), "+v"(l16));
        const u32x4g v = *(const u32x4g __attribute__((address_space(1))) *)(fb + l16);
        uint4 r;
        __builtin_memcpy(&r, &v, 16);
        return r;
    };
#else
    const int lane = tid & 63, wave = tid >> 6;
    const uint4 *w1 = (const uint4 *)m.w1 + lane, *w2 = (const uint4 *)m.w2 + lane, *w3 = (const uint4 *)m.w3 + lane;
    auto frag = [&](const uint4 *w, int idx) __attribute__((always_inline)) -> uint4 { return w[(unsigned)(idx * 64)]; };
#endif
    uint4 ring[RING];
    auto request = [&](int seq) __attribute__((always_inline)) {   // (seq is a constant wherever this is called, after unrolling)
        if (seq >= STOT) return;
        const int L = seq < S1 ? 0 : (seq < S1 + S2 ? 1 : 2);
        const int r = seq - (L == 0 ? 0 : (L == 1 ? S1 : S1 + S2));
        const int KS = L == 0 ? KS1 : (L == 1 ? KS2 : KS3), NT = L == 0 ? NT1 : (L == 1 ? NT2 : NT3);
        const int MTL = L == 0 ? MT1 : (L == 1 ? MT2 : MT3);
        const bool kso = RB == 1 && MTL == 2 && ((L == 0 && (EV2G_BF16_KSOUTER & 1)) || (L == 1 && (EV2G_BF16_KSOUTER & 2)));   // k-step, tile slot (below)
        const int i = kso ? r % MTL : r / KS, ks = kso ? r / MTL : r - i * KS;
        const auto w = L == 0 ? w1 : (L == 1 ? w2 : w3);
        if (WVS * i + WVS - 1 < NT || wave + WVS * i < NT) ring[seq % RING] = frag(w, (wave + WVS * i) * KS + ks);
    };
    // (requesting the head of the sequence a phase EARLIER in the step kernel -- behind phase E, D or C of the step before -- was tried: the whole ring
    // then lives across the step loop's back edge and the register allocator spills 40..119 registers; only layer 1's first tile (6 fragments)
    // requested early fits -- and measured 1-2 % SLOWER than this, docs/history/experiments/round5/gpu_r5e.sh: the head's latency is not what the policy phase waits for)
    F32_STAMP(0)
#pragma unroll
    for (int sq = 0; sq < RING; sq++) request(sq);
    // first barrier: every wavefront's observation columns of the step before (or the prologue's rows) are in bufX, and nobody still reads the
    // staging rows the hidden activations are about to use
    ev2g_mlp_lds_barrier();
    F32_STAMP(1)
    if (tid < 256 * RB) {   // columns no tile writes (the next layer's k-steps read them): zeros
        const int pj = tid & 15, pr = tid >> 4;
        constexpr int P1 = KS2 * 32 - NT1 * 16, P2 = KS3 * 32 - NT2 * 16;
        if (pj < P1) bufH1[pr * C::SH1 + NT1 * 16 + pj] = 0;
        if (pj < P2) bufH2[pr * C::SH2 + NT2 * 16 + pj] = 0;
    }
    const int brow = lane & 15, kq = lane >> 4;
    auto layer = [&](auto Lc, const uint16_t *A, int sa, const float *bias, uint16_t *out, int so) __attribute__((always_inline)) {
        constexpr int L = decltype(Lc)::value;
        constexpr int KS = L == 0 ? KS1 : (L == 1 ? KS2 : KS3), NT = L == 0 ? NT1 : (L == 1 ? NT2 : NT3), MT = (NT + WVS - 1) / WVS;
        constexpr int base = L == 0 ? 0 : (L == 1 ? S1 : S1 + S2);
        // EV2G_BF16_KSOUTER (bit 0: layer 1, bit 1: layer 2; one block of rows): a two-tile wavefront walks k-step, tile slot -- ONE operand read and two independent
        // MFMAs per k-step -- instead of one tile after the other: its chain of (LDS read -> MFMA -> request) round trips halves (same chain per tile: bit-identical)
        constexpr bool KSO = RB == 1 && MT == 2 && L < 2 && ((L == 0 && (EV2G_BF16_KSOUTER & 1)) || (L == 1 && (EV2G_BF16_KSOUTER & 2)));
        if (KSO) {
            f32x4m acc0[MT], acc1[MT];
#pragma unroll
            for (int i = 0; i < MT; i++) { acc0[i] = *(const f32x4m *)(bias + min(wave + WVS * i, NT - 1) * 16 + kq * 4); acc1[i] = f32x4m{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                bf16x8 b;
                const uint4 bw = *(const uint4 *)(A + brow * sa + ks * 32 + kq * 8);
                __builtin_memcpy(&b, &bw, 16);
#pragma unroll
                for (int i = 0; i < MT; i++) {
                    const int sq = base + ks * MT + i;
                    if (WVS * i + WVS - 1 < NT || wave + WVS * i < NT) {   // (uniform; a constant but for the last slot)
                        bf16x8 a;
                        __builtin_memcpy(&a, &ring[sq % RING], 16);
                        if (ks & 1) acc1[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1[i], 0, 0, 0);
                        else acc0[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0[i], 0, 0, 0);
                    }
                    request(sq + RING);
                }
            }
#pragma unroll
            for (int i = 0; i < MT; i++) {
                const int tile = wave + WVS * i;
                if (WVS * i + WVS - 1 < NT || tile < NT) {
                    const f32x4m acc = acc0[i] + acc1[i];
                    const uint32_t lo = ev2g_pack_bf16(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f)), hi = ev2g_pack_bf16(fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
                    *(uint2 *)(out + brow * so + tile * 16 + kq * 4) = make_uint2(lo, hi);
                }
            }
        } else
#pragma unroll
        for (int i = 0; i < MT; i++) {
            const int tile = wave + WVS * i;
            if (WVS * i + WVS - 1 < NT || tile < NT) {   // (uniform; a constant but for the last slot)
                f32x4m acc0[RB], acc1[RB];
#pragma unroll
                for (int rb = 0; rb < RB; rb++) { acc0[rb] = *(const f32x4m *)(bias + tile * 16 + kq * 4); acc1[rb] = f32x4m{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    const int sq = base + i * KS + ks;
                    bf16x8 a;
                    __builtin_memcpy(&a, &ring[sq % RING], 16);
#pragma unroll
                    for (int rb = 0; rb < RB; rb++) {   // (one weight fragment, RB blocks of rows)
                        bf16x8 b;
                        const uint4 bw = *(const uint4 *)(A + (rb * 16 + brow) * sa + ks * 32 + kq * 8);
                        __builtin_memcpy(&b, &bw, 16);
                        if (ks & 1) acc1[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1[rb], 0, 0, 0);
                        else acc0[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0[rb], 0, 0, 0);
                    }
                    request(sq + RING);   // this slot is free again
                }
                const int col = tile * 16 + kq * 4;   // this lane: columns col .. col + 3 of env rows `brow`, `brow + 16`
#pragma unroll
                for (int rb = 0; rb < RB; rb++) {
                    const f32x4m acc = acc0[rb] + acc1[rb];
                    const int erow = rb * 16 + brow;
                    if (L < 2) {
                        const uint32_t lo = ev2g_pack_bf16(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f)), hi = ev2g_pack_bf16(fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
                        *(uint2 *)(out + erow * so + col) = make_uint2(lo, hi);
                    } else {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) { v[r] = ev2g_fast_tanh(acc[r]); if (m.out_lo == 0.0f) v[r] = v[r] * 0.5f + 0.5f; }
                        *(float4 *)(act + erow * as + col) = make_float4(v[0], v[1], v[2], v[3]);   // (NT3 <= 4: columns 0..63)
                        const int d_out = m.d_out;
                        if (erow < nr) {
                            float *yr = y + (size_t)erow * d_out + col;
                            if ((d_out & 1) == 0) {
                                if (col + 1 < d_out) *(float2 *)yr = make_float2(v[0], v[1]);
                                if (col + 3 < d_out) *(float2 *)(yr + 2) = make_float2(v[2], v[3]);
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; r++) if (col + r < d_out) yr[r] = v[r];
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < KS; u++) request(base + i * KS + u + RING);   // no tile in this slot: the sequence moves on all the same
            }
        }
    };
    layer(std::integral_constant<int, 0>{}, bufX, C::SX, lb, bufH1, C::SH1);
    F32_STAMP(2)
    ev2g_mlp_lds_barrier();
    F32_STAMP(3)
    layer(std::integral_constant<int, 1>{}, bufH1, C::SH1, lb + NT1 * 16, bufH2, C::SH2);
    F32_STAMP(4)
    ev2g_mlp_lds_barrier();
    F32_STAMP(5)
    layer(std::integral_constant<int, 2>{}, bufH2, C::SH2, lb + (NT1 + NT2) * 16, nullptr, 0);
    F32_STAMP(6)
    ev2g_mlp_lds_barrier();
    F32_STAMP(7)
}

// ---- the FLOAT32 network (EV2G_MLP_F32: two bf16 terms per weight, three per activation, five MFMA products per k-step) as a device function of the
// same 16-wavefront workgroup (ev2g_step_wave<.., ACT, 1, 2>).  Same tiles, same fragment packing, same per-tile MFMA chain -- terms, order, accumulator by
// parity -- and the same epilogues as ev2g_mlp3_s16<.., NW = 2>: the actions are bit-identical to that kernel's.  What the 160 KB of LDS next to the step's
// state dictate:
//   * the input rows stay FLOAT32 in LDS (16 x (KS1*32 + 4) floats: two thirds of the three bf16 copies) and a wavefront splits its operand fragment into
//     the three terms when it reads it -- ONCE per k-step for both of its layer-1 tiles (layer 1 walks k-step, tile, term; the other layers tile, k-step,
//     term), so the split costs each wavefront 6 x 44 VALU operations per forward;
//   * the hidden activations are split where they are produced (three bf16 copies each, like the stand-alone kernel): H1's three copies and two of H2's
//     in the step's staging rows (61.7 of 66 KB), H2's third copy over the input rows, which are dead after layer 1 -- the padding columns behind d_in
//     are re-zeroed at the end (the step rewrites every real column before the next forward);
//   * the biases are read from global memory (L1 / L2 hits, requested a layer ahead), not staged.
template <int KS1, int NT1, int NT2, int NT3, int WVS, int RING>
__device__ __forceinline__ void ev2g_mlp3_inline_f32(const MlpDev &m, float *bufX, int sxf, uint16_t *h1, uint16_t *h2ab, uint16_t *h2c, uint16_t *xs, int xcs, float *act, int as, float *y, int nr, int tid) {
    constexpr int NW = 2, NX = 3;
    typedef MlpS16<KS1, NT1, NT2, NT3, 1, 4, 1> C;
    constexpr int KS2 = C::KS2, KS3 = C::KS3;
    constexpr int MT1 = (NT1 + WVS - 1) / WVS, MT2 = (NT2 + WVS - 1) / WVS, MT3 = (NT3 + WVS - 1) / WVS;
    // Layer 3 (NT3 tiles: one wavefront each, the other wavefronts idle) through a ring of ITS OWN: with the shared ring's few slots a lone wavefront pays a
    // memory round trip per two k-steps (stamps: 330 cycles per k-step at RING = 4); the registers of layers 1 / 2 (a second tile's accumulators, the split
    // temporaries) are free there.  R3 fragments are requested when the wavefront leaves layer 2 -- they fly under the wait at that layer's closing barrier.
#ifndef EV2G_F32_RING3
#define EV2G_F32_RING3 8
#endif
    constexpr int R3 = (MT3 == 1) ? EV2G_F32_RING3 : 0, R3M = R3 > 0 ? R3 : 1;
    constexpr int S1 = MT1 * KS1 * NW, S2 = MT2 * KS2 * NW, S3 = R3 > 0 ? 0 : MT3 * KS3 * NW, STOT = S1 + S2 + S3;
    constexpr int BH1 = 16 * C::SH1, BH2 = 16 * C::SH2;
    static_assert(RING % NW == 0 && RING >= 2 * NW && R3 % NW == 0, "ring slots come in pairs of weight terms");
    // (A/B switch: layer 3's few tiles on wavefronts W3OFF .. W3OFF + NT3 - 1.  Wavefronts 4..7 reach layer 2's closing barrier ~3 k cycles before the two-tile ones
    // (0 .. 2), so their ring is full of layer-3 fragments by then -- and layer 3 takes the same 3.5 k cycles: it waits for its own LDS-read -> MFMA chain per k-step,
    // not for weights; profiles/r06_fused_float32_policy.txt.  Default 0.)
#ifndef EV2G_F32_W3OFF
#define EV2G_F32_W3OFF 0
#endif
    constexpr int W3OFF = (MT3 == 1 && NT3 + EV2G_F32_W3OFF <= WVS) ? EV2G_F32_W3OFF : 0;
#if EV2G_F32_SADDR
    // the wavefront's number as a SCALAR (it is uniform by construction; `tid >> 6` alone is a vector value to the compiler): the tile guards become scalar branches and a
    // weight request is `global_load_dwordx4 v, v_lane16, s[base]` behind two scalar adds instead of three vector instructions and an address register pair per request
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)lane * 16u;
    typedef const char __attribute__((address_space(1))) *wgptr;
    typedef unsigned u32x4g __attribute__((ext_vector_type(4)));
    const wgptr w1 = (wgptr)(unsigned long long)m.w1, w2 = (wgptr)(unsigned long long)m.w2, w3 = (wgptr)(unsigned long long)m.w3;
    auto frag = [&](wgptr w, int idx) __attribute__((always_inline)) -> uint4 {   // fragment idx (1 KB each) of a layer's packed weights, this lane's 16 bytes
        wgptr fb = w + (unsigned long long)(unsigned)idx * 1024ull;
        unsigned l16 = lane16;
        asm volatile("" : "+s"(fb), "+v"(l16));   // (the scalar base stays a scalar pair and the lane offset a 32-bit value extended HERE: base + zext(offset) is the `saddr` form of the load)
        const u32x4g v = *(const u32x4g __attribute__((address_space(1))) *)(fb + l16);
        uint4 r;
        __builtin_memcpy(&r, &v, 16);
        return r;
    };
#else
    const int lane = tid & 63, wave = tid >> 6;
    const uint4 *w1 = (const uint4 *)m.w1 + lane, *w2 = (const uint4 *)m.w2 + lane, *w3 = (const uint4 *)m.w3 + lane;
    auto frag = [&](const uint4 *w, int idx) __attribute__((always_inline)) -> uint4 { return w[(unsigned)(idx * 64)]; };
#endif
    const float *ball = m.b1;   // b1 | b2 | b3, each padded to its tiles
    uint4 ring[RING];
    auto request = [&](int seq) __attribute__((always_inline)) {   // (seq is a constant wherever this is called, after unrolling)
        if (seq >= STOT) return;
        const int L = seq < S1 ? 0 : (seq < S1 + S2 ? 1 : 2);
        const int r = seq - (L == 0 ? 0 : (L == 1 ? S1 : S1 + S2));
        const int KS = L == 0 ? KS1 : (L == 1 ? KS2 : KS3), NT = L == 0 ? NT1 : (L == 1 ? NT2 : NT3);
        int i, rem;   // tile slot; ks * NW + term
        if (L == 0) { const int ks = r / (MT1 * NW), r2 = r - ks * (MT1 * NW); i = r2 / NW; rem = ks * NW + (r2 - i * NW); }
        else { i = r / (KS * NW); rem = r - i * (KS * NW); }
        const auto w = L == 0 ? w1 : (L == 1 ? w2 : w3);
        if (L == 2) { if ((unsigned)(wave - W3OFF) < (unsigned)NT3) ring[seq % RING] = frag(w, (wave - W3OFF) * (KS * NW) + rem); }
        else
        if (WVS * i + WVS - 1 < NT || wave + WVS * i < NT) ring[seq % RING] = frag(w, (wave + WVS * i) * (KS * NW) + rem);
    };
    F32_STAMP(0)
    uint4 ring3[R3M];
    const int tile3 = wave - W3OFF;   // this wavefront's layer-3 tile, if 0 <= tile3 < NT3
    auto request3 = [&](int r) __attribute__((always_inline)) {   // fragment r = ks * NW + term of that tile (callers: wavefronts that own one)
        if (R3 > 0 && r < KS3 * NW) ring3[r % R3M] = frag(w3, tile3 * (KS3 * NW) + r);
    };
    const int brow = lane & 15, kq = lane >> 4;
    f32x4m bias1[MT1];
#pragma unroll
    for (int i = 0; i < MT1; i++) bias1[i] = *(const f32x4m *)(ball + min(wave + WVS * i, NT1 - 1) * 16 + kq * 4);
#pragma unroll
    for (int sq = 0; sq < RING; sq++) request(sq);
#if EV2G_F32_XSPLIT
    // This wavefront's OWN input row (its env's observation: its own LDS writes, in program order) -> the three bf16 terms, into its own 512-byte chunks of three
    // staging rows (xs + c * xcs + wave * 256 elements; chunks no other wavefront touches outside the battery-maths phase, and dead in layer 1: H2's first two
    // copies land there in layer 2).  The 24 16-byte pieces of a row sit at piece (j ^ row): the rows are 512 bytes apart -- all on the same banks -- and a
    // ds_read_b128 lane group holds the rows {0..3, 12..15} at one k-quarter and {4..11} at the next, whose piece numbers differ in bit 0 only: no conflicts.
    {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int c = h == 0 ? 2 * lane : 128 + 2 * lane;   // columns 0..127, then 128..191 (lanes 0..31)
            if (c < KS1 * 32) {
                const float2 v = *(const float2 *)(bufX + wave * sxf + c);
                uint32_t wd[NX];
                ev2g_split_bf16<NX>(v.x, v.y, wd);
                const int at = wave * 256 + (((c >> 3) ^ wave) << 3) + (c & 7);
#pragma unroll
                for (int k = 0; k < NX; k++) *(uint32_t *)(xs + k * xcs + at) = wd[k];
            }
        }
    }
#endif
    ev2g_mlp_lds_barrier();   // the step's observation columns are in bufX; nobody reads the staging rows any more
    F32_STAMP(1)
    if (tid < 256) {   // H1's columns no tile writes (layer 2's last k-step reads them): zeros, in every copy
        const int pj = tid & 15, pr = tid >> 4;
        constexpr int P1 = KS2 * 32 - NT1 * 16;
        if (pj < P1) {
#pragma unroll
            for (int k = 0; k < NX; k++) h1[k * BH1 + pr * C::SH1 + NT1 * 16 + pj] = 0;
        }
    }
    // the MFMA chain of one (tile, k-step): the small terms first, accumulator by the parity of k-step + terms (ev2g_mlp3_s16)
    auto chain = [&](int ks, const uint4 &w_t0, const uint4 &w_t1, const uint4 (&bt)[NX], f32x4m &acc0, f32x4m &acc1) __attribute__((always_inline)) {
        typedef unsigned u32x4m __attribute__((ext_vector_type(4)));
        u32x4m av[NW], bv[NX];
        __builtin_memcpy(&av[0], &w_t0, 16);
        __builtin_memcpy(&av[1], &w_t1, 16);
#pragma unroll
        for (int k = 0; k < NX; k++) __builtin_memcpy(&bv[k], &bt[k], 16);
#if EV2G_F32_GROUP
        // every operand of the k-step collected BEFORE its first MFMA, then the five MFMAs back to back: a wait or a load between two MFMAs on the same accumulator
        // costs ~40 cycles each (MI355X_MICROARCH.md, per-instruction constants)
        asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]));
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int p = NW - 1; p >= 0; p--) {
            bf16x8 a;
            __builtin_memcpy(&a, &av[p], 16);
#pragma unroll
            for (int xq = NX - 1; xq >= 0; xq--) {
                if (p + xq <= 2) {
                    bf16x8 b;
                    __builtin_memcpy(&b, &bv[xq], 16);
                    if ((ks + p + xq) & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
                }
            }
        }
#if EV2G_F32_GROUP
        __builtin_amdgcn_sched_barrier(0);
#endif
    };
    auto hidden_out = [&](const f32x4m &acc, uint16_t *o0, uint16_t *o1, uint16_t *o2, int so, int col) __attribute__((always_inline)) {
        uint32_t lo[NX], hi[NX];
        ev2g_split_bf16<NX>(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), lo);
        ev2g_split_bf16<NX>(fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f), hi);
        *(uint2 *)(o0 + brow * so + col) = make_uint2(lo[0], hi[0]);
        *(uint2 *)(o1 + brow * so + col) = make_uint2(lo[1], hi[1]);
        *(uint2 *)(o2 + brow * so + col) = make_uint2(lo[2], hi[2]);
    };
    // ---- layer 1: float32 rows -> three terms per operand fragment, once per k-step for both tile slots ----
    f32x4m bias2[MT2];
#pragma unroll
    for (int i = 0; i < MT2; i++) bias2[i] = *(const f32x4m *)(ball + NT1 * 16 + min(wave + WVS * i, NT2 - 1) * 16 + kq * 4);   // (a layer ahead)
    {
        f32x4m acc0[MT1], acc1[MT1];
#pragma unroll
        for (int i = 0; i < MT1; i++) { acc0[i] = bias1[i]; acc1[i] = f32x4m{0.f, 0.f, 0.f, 0.f}; }
#if EV2G_F32_XSPLIT && EV2G_F32_L1AHEAD
        uint4 btn[NX];
#pragma unroll
        for (int k = 0; k < NX; k++) btn[k] = *(const uint4 *)(xs + k * xcs + brow * 256 + ((kq ^ brow) << 3));
#endif
#pragma unroll
        for (int ks = 0; ks < KS1; ks++) {
            uint4 bt[NX];
#if EV2G_F32_XSPLIT && EV2G_F32_L1AHEAD
#pragma unroll
            for (int k = 0; k < NX; k++) bt[k] = btn[k];
            if (ks + 1 < KS1) {
#pragma unroll
                for (int k = 0; k < NX; k++) btn[k] = *(const uint4 *)(xs + k * xcs + brow * 256 + ((((ks + 1) * 4 + kq) ^ brow) << 3));
            }
#elif EV2G_F32_XSPLIT
#pragma unroll
            for (int k = 0; k < NX; k++) bt[k] = *(const uint4 *)(xs + k * xcs + brow * 256 + (((ks * 4 + kq) ^ brow) << 3));
#else
            const float4 xa = *(const float4 *)(bufX + brow * sxf + ks * 32 + kq * 8), xb = *(const float4 *)(bufX + brow * sxf + ks * 32 + kq * 8 + 4);
            uint32_t t0[NX], t1[NX], t2[NX], t3[NX];
            ev2g_split_bf16<NX>(xa.x, xa.y, t0); ev2g_split_bf16<NX>(xa.z, xa.w, t1); ev2g_split_bf16<NX>(xb.x, xb.y, t2); ev2g_split_bf16<NX>(xb.z, xb.w, t3);
#pragma unroll
            for (int k = 0; k < NX; k++) bt[k] = make_uint4(t0[k], t1[k], t2[k], t3[k]);
#endif
#pragma unroll
            for (int i = 0; i < MT1; i++) {
                const int sq0 = (ks * MT1 + i) * NW;
                if (WVS * i + WVS - 1 < NT1 || wave + WVS * i < NT1) chain(ks, ring[sq0 % RING], ring[(sq0 + 1) % RING], bt, acc0[i], acc1[i]);   // (uniform; a constant but for the last slot)
#pragma unroll
                for (int p = 0; p < NW; p++) request(sq0 + p + RING);
            }
        }
#pragma unroll
        for (int i = 0; i < MT1; i++) {
            const int tile = wave + WVS * i;
            if (WVS * i + WVS - 1 < NT1 || tile < NT1) hidden_out(acc0[i] + acc1[i], h1, h1 + BH1, h1 + 2 * BH1, C::SH1, tile * 16 + kq * 4);
        }
    }
    F32_STAMP(2)
    ev2g_mlp_lds_barrier();
    F32_STAMP(3)
    if (tid < 256) {   // H2's tail columns (its third copy lies over the input rows: only now)
        const int pj = tid & 15, pr = tid >> 4;
        constexpr int P2 = KS3 * 32 - NT2 * 16;
        if (pj < P2) { h2ab[pr * C::SH2 + NT2 * 16 + pj] = 0; h2ab[BH2 + pr * C::SH2 + NT2 * 16 + pj] = 0; h2c[pr * C::SH2 + NT2 * 16 + pj] = 0; }
    }
    // ---- layers 2 and 3: tile slot, k-step, term ----
    auto layer = [&](auto Lc, const uint16_t *a0, const uint16_t *a1, const uint16_t *a2, int sa, const f32x4m *bias) __attribute__((always_inline)) {
        constexpr int L = decltype(Lc)::value;
        constexpr int KS = L == 1 ? KS2 : KS3, NT = L == 1 ? NT2 : NT3, MT = (NT + WVS - 1) / WVS;
        constexpr int base = L == 1 ? S1 : S1 + S2;
#pragma unroll
        for (int i = 0; i < MT; i++) {
            const int tile = L == 2 ? wave - W3OFF : wave + WVS * i;
            if (L == 2 ? (unsigned)tile < (unsigned)NT : (WVS * i + WVS - 1 < NT || tile < NT)) {   // (uniform)
                f32x4m acc0 = bias[i], acc1 = f32x4m{0.f, 0.f, 0.f, 0.f};
#if EV2G_F32_GROUP >= 2   // the next k-step's operand terms are read from LDS before this one's MFMA group (12 registers more)
                uint4 btn[NX];
                btn[0] = *(const uint4 *)(a0 + brow * sa + kq * 8); btn[1] = *(const uint4 *)(a1 + brow * sa + kq * 8); btn[2] = *(const uint4 *)(a2 + brow * sa + kq * 8);
#endif
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    const int sq0 = base + (i * KS + ks) * NW;
                    uint4 bt[NX];
#if EV2G_F32_GROUP >= 2
                    bt[0] = btn[0]; bt[1] = btn[1]; bt[2] = btn[2];
                    if (ks + 1 < KS) {
                        btn[0] = *(const uint4 *)(a0 + brow * sa + (ks + 1) * 32 + kq * 8);
                        btn[1] = *(const uint4 *)(a1 + brow * sa + (ks + 1) * 32 + kq * 8);
                        btn[2] = *(const uint4 *)(a2 + brow * sa + (ks + 1) * 32 + kq * 8);
                    }
#else
                    bt[0] = *(const uint4 *)(a0 + brow * sa + ks * 32 + kq * 8);
                    bt[1] = *(const uint4 *)(a1 + brow * sa + ks * 32 + kq * 8);
                    bt[2] = *(const uint4 *)(a2 + brow * sa + ks * 32 + kq * 8);
#endif
                    if (L == 2 && R3 > 0) {
                        chain(ks, ring3[(ks * NW) % R3M], ring3[(ks * NW + 1) % R3M], bt, acc0, acc1);
#pragma unroll
                        for (int p = 0; p < NW; p++) request3(ks * NW + p + R3);
                    } else {
                    chain(ks, ring[sq0 % RING], ring[(sq0 + 1) % RING], bt, acc0, acc1);
#pragma unroll
                    for (int p = 0; p < NW; p++) request(sq0 + p + RING);
                    }
                    if (L == 2) { F32_STAMP2(ks) }
                }
                if (L == 2) { F32_STAMP2(15) }
                const f32x4m acc = acc0 + acc1;
                const int col = tile * 16 + kq * 4;
                if (L == 1) hidden_out(acc, h2ab, h2ab + BH2, h2c, C::SH2, col);
                else {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) { v[r] = ev2g_tanh_f32(acc[r]); if (m.out_lo == 0.0f) v[r] = v[r] * 0.5f + 0.5f; }
                    *(float4 *)(act + brow * as + col) = make_float4(v[0], v[1], v[2], v[3]);   // (NT3 <= 4: columns 0..63)
                    const int d_out = m.d_out;
                    if (brow < nr) {
                        float *yr = y + (size_t)brow * d_out + col;
                        if ((d_out & 1) == 0) {
                            if (col + 1 < d_out) *(float2 *)yr = make_float2(v[0], v[1]);
                            if (col + 3 < d_out) *(float2 *)(yr + 2) = make_float2(v[2], v[3]);
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; r++) if (col + r < d_out) yr[r] = v[r];
                        }
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < KS * NW; u++) request(base + i * KS * NW + u + RING);   // no tile in this slot: the sequence moves on all the same
            }
        }
    };
    f32x4m bias3[MT3];
#pragma unroll
    for (int i = 0; i < MT3; i++) bias3[i] = *(const f32x4m *)(ball + (NT1 + NT2) * 16 + min(max(wave - W3OFF, 0) + WVS * i, NT3 - 1) * 16 + kq * 4);
    layer(std::integral_constant<int, 1>{}, h1, h1 + BH1, h1 + 2 * BH1, C::SH1, bias2);
    if (R3 > 0 && (unsigned)tile3 < (unsigned)NT3) {   // (uniform)
#pragma unroll
        for (int r = 0; r < R3; r++) request3(r);
    }
    F32_STAMP(4)
    ev2g_mlp_lds_barrier();
    F32_STAMP(5)
    layer(std::integral_constant<int, 2>{}, h2ab, h2ab + BH2, h2c, C::SH2, bias3);
    F32_STAMP(6)
    ev2g_mlp_lds_barrier();
    F32_STAMP(7)
    {   // the input rows' padding columns, which H2's third copy overwrote
        const int d_in = m.d_in, pr = tid >> 6;
        for (int cc = d_in + lane; cc < KS1 * 32; cc += 64) bufX[pr * sxf + cc] = 0.f;
    }
}

__global__ void __launch_bounds__(EV2G_MLP_BLOCK) ev2g_mlp3_any(MlpDev m, const float *__restrict__ x, float *__restrict__ y, int n_rows) {
    extern __shared__ __attribute__((aligned(16))) uint16_t mlds[];
    const int sA = ev2g_mlp_lds_stride(m.k1 > m.n2 ? m.k1 : m.n2), sB = ev2g_mlp_lds_stride(m.n1);
    uint16_t *bufA = mlds, *bufB = mlds + EV2G_MLP_ROWS * sA;
    const int row0 = blockIdx.x * EV2G_MLP_ROWS;
    ev2g_mlp_stage_input(m, x, row0, n_rows, bufA, sA);
    __syncthreads();
    ev2g_mlp_layer_any<false>(bufA, sA, m.k1, m.n1, m.w1, m.b1, bufB, sB, nullptr, row0, n_rows, 0, 0.f);
    __syncthreads();
    ev2g_mlp_layer_any<false>(bufB, sB, m.n1, m.n2, m.w2, m.b2, bufA, sA, nullptr, row0, n_rows, 0, 0.f);
    __syncthreads();
    ev2g_mlp_layer_any<true>(bufA, sA, m.n2, m.n3, m.w3, m.b3, nullptr, 0, y, row0, n_rows, m.d_out, m.out_lo);
}


// ---- float32 variant: the SAME network with float32 operands (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation) ----
// SB3 policies are float32 (train_stable_baselines.py:62-130): with this kernel a network trained there produces, on the device,
// the actions its own framework would (agreement with a float32 numpy forward at the 1e-6 level: only the summation order
// differs), at roughly twice the time of the bf16 kernel -- weights are twice the bytes and the f32 MFMA runs at 1/8 of the bf16 rate.
// Operand layout of the 32x32x2 MFMA: lane l supplies A[row = l & 31][k(l)] and B[k(l)][col = l & 31], the instruction sums its two
// k's.  Here MFMA j of a group of eight k's uses k = 8 g + 4 (l >> 5) + j, so a lane's four A values and four weights are contiguous:
// one 16-byte LDS read and one 16-byte (pre-packed, coalesced) global load feed four MFMAs.  Activations stay in LDS as float32.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__host__ __device__ inline int ev2g_mlp32_lds_stride(int k) { return k + 4; }   // floats per LDS row (+16 bytes against bank conflicts)
__host__ __device__ inline size_t ev2g_mlp32_lds_bytes(const MlpDev &m) {
    const int a = ev2g_mlp32_lds_stride(m.k1 > m.n2 ? m.k1 : m.n2), b = ev2g_mlp32_lds_stride(m.n1);
    return (size_t)EV2G_MLP_ROWS * (a + b) * sizeof(float);
}
#define EV2G_MLP32_DEPTH 8
template <bool FINAL>
__device__ __forceinline__ void ev2g_mlp32_layer(const float *__restrict__ A, int sa, int K, int N, const float *__restrict__ W,
                                                 const float *__restrict__ bias, float *__restrict__ out, int so, float *__restrict__ gout,
                                                 int row0, int n_rows, int d_out, float out_lo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int KG = K >> 3, NT = N >> 5;
    const float *arow = A + (lane & 31) * sa + 4 * (lane >> 5);
    for (int nt = wave; nt < NT; nt += EV2G_MLP_BLOCK / 64) {
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const f32x4 *w = (const f32x4 *)W + ((size_t)nt * KG) * 64 + lane;
        f32x4 ring[EV2G_MLP32_DEPTH];
#pragma unroll
        for (int u = 0; u < EV2G_MLP32_DEPTH; u++) ring[u] = w[(size_t)min(u, KG - 1) * 64];
        for (int g0 = 0; g0 < KG; g0 += EV2G_MLP32_DEPTH) {
#pragma unroll
            for (int u = 0; u < EV2G_MLP32_DEPTH; u++) {
                const int g = g0 + u;
                if (g < KG) {   // (uniform)
                    const f32x4 a = *(const f32x4 *)(arow + g * 8), b = ring[u];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
                    ring[u] = w[(size_t)min(g + EV2G_MLP32_DEPTH, KG - 1) * 64];
                }
            }
        }
        const int col = nt * 32 + (lane & 31);
        const float bv = bias[col];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float v = acc[r] + bv;
            if (FINAL) {
                v = tanhf(v);
                if (out_lo == 0.0f) v = v * 0.5f + 0.5f;
                if (col < d_out && row0 + row < n_rows) gout[(size_t)(row0 + row) * d_out + col] = v;
            } else {
                out[row * so + col] = v > 0.0f ? v : 0.0f;
            }
        }
    }
}

__global__ void __launch_bounds__(EV2G_MLP_BLOCK) ev2g_mlp3_f32(MlpDev m, const float *__restrict__ x, float *__restrict__ y, int n_rows) {
    extern __shared__ __attribute__((aligned(16))) float mlds32[];
    const int sA = ev2g_mlp32_lds_stride(m.k1 > m.n2 ? m.k1 : m.n2), sB = ev2g_mlp32_lds_stride(m.n1);
    float *bufA = mlds32, *bufB = mlds32 + EV2G_MLP_ROWS * sA;
    const int row0 = blockIdx.x * EV2G_MLP_ROWS;
    // input rows -> LDS (zero-padded in K and past the last row): the 32 rows are one contiguous float32 range
    const int nr = min(EV2G_MLP_ROWS, n_rows - row0);
    for (int i = threadIdx.x; i < EV2G_MLP_ROWS * m.k1; i += EV2G_MLP_BLOCK) {
        const int r = i / m.k1, c = i - r * m.k1;
        bufA[r * sA + c] = (r < nr && c < m.d_in) ? x[(size_t)(row0 + r) * m.d_in + c] : 0.0f;
    }
    __syncthreads();
    ev2g_mlp32_layer<false>(bufA, sA, m.k1, m.n1, (const float *)m.w1, m.b1, bufB, sB, nullptr, row0, n_rows, 0, 0.f);
    __syncthreads();
    ev2g_mlp32_layer<false>(bufB, sB, m.n1, m.n2, (const float *)m.w2, m.b2, bufA, sA, nullptr, row0, n_rows, 0, 0.f);
    __syncthreads();
    ev2g_mlp32_layer<true>(bufA, sA, m.n2, m.n3, (const float *)m.w3, m.b3, nullptr, 0, y, row0, n_rows, m.d_out, m.out_lo);
}
