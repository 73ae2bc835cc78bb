// ev2g_mlp.h -- the policy head of a rollout: obs[E,D] (float32) -> actions[E,P] (float32) through a three-layer MLP
// (D -> H1 -> H2 -> P, ReLU, ReLU, tanh: the SB3 MlpPolicy actor shape of BASELINE configs[4]), as ONE kernel between two
// env steps.
//
// The reference trains SB3 agents against one CPU env (train_stable_baselines.py:62-130); with thousands of envs resident
// on the GPU a step of the engine takes ~12 us, so a policy evaluated through a dozen framework launches (GEMM, bias,
// activation, casts ...) would be 90 % of a rollout step.  This kernel keeps the whole forward on chip:
//   * one workgroup (4 wavefronts) per 32 env rows; activations live in LDS as bf16, accumulators in registers (fp32);
//   * v_mfma_f32_32x32x16_bf16: output-column tiles of 32 go round-robin over the wavefronts, a tile is one MFMA chain over K;
//   * weights are packed ONCE on the host into MFMA B-fragment order (bf16), so a lane's operand is one coalesced 16-byte
//     load straight from L2 -- no LDS staging: with a single 32-row tile per workgroup a weight is used exactly once per
//     workgroup, and the 0.45 MB of weights stay L2-resident across workgroups;
//   * bias + activation are applied to the accumulators in registers, the next layer's A matrix is written back to LDS.
// Precision: bf16 operands, fp32 accumulation (a policy network, not part of the simulator's float64 path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware exp2 / rcp (relative error ~1e-6, saturates cleanly at +-1)
__device__ __forceinline__ float ev2g_fast_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);   // exp(2x) = 2^(2x / ln 2)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
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
