// How fast can every workgroup of a launch stream the SAME 0.44 MB (the policy network's bf16 weights) out of a cold L2?  The floor of the fused
// actor (ev2g_mlp.h): its MFMAs are ~10 % of its time, the weight stream through each CU's 64 B/clk vector-memory port is the rest.
// Variants: workgroups x wavefronts per workgroup, 16-byte loads in flight per wavefront (<= 63: vmcnt is six bits), every workgroup
// walking the stream from its own rotation or all from the start.  Between two timed launches a trasher kernel rewrites 64 MB (what the
// env step does to the L2s between two forwards).
//   hipcc --offload-arch=gfx950 -O3 -o build_variants/wstream tools/micro/weight_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define NFRAG 443   // 1 KB fragments (64 lanes x 16 bytes)
template <int WAVES, int DEPTH, bool ROT>
__global__ void __launch_bounds__(WAVES * 64) stream(const uint4 *__restrict__ w, unsigned *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rot = ROT ? (int)((blockIdx.x * 37u) % NFRAG) : 0;
    constexpr int PER = (NFRAG + WAVES - 1) / WAVES;   // fragments per wavefront
    uint4 ring[DEPTH];
    unsigned acc = 0;
    auto frag = [&](int i) { int f = wave + min(i, PER - 1) * WAVES; f = min(f, NFRAG - 1) + rot; if (f >= NFRAG) f -= NFRAG; return w[(unsigned)f * 64u + lane]; };
#pragma unroll
    for (int u = 0; u < DEPTH; u++) ring[u] = frag(u);
    for (int i0 = 0; i0 < PER; i0 += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; u++) {
            if (i0 + u < PER) {   // (uniform)
                acc ^= ring[u].x ^ ring[u].y ^ ring[u].z ^ ring[u].w;
                ring[u] = frag(i0 + u + DEPTH);
            }
        }
    }
    if (acc == 0x12345u) out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) trash(double2 *a, int n, int k) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) { a[i].x = k; a[i].y = i; }
}
template <int WAVES, int DEPTH, bool ROT> void run(const char *name, int nwg, const uint4 *w, unsigned *out, double2 *big, int nbig) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int K = 300;
    float ms_pair, ms_trash, ms_hot;
    for (int k = 0; k < 20; k++) { trash<<<1024, 256>>>(big, nbig, k); stream<WAVES, DEPTH, ROT><<<nwg, WAVES * 64>>>(w, out); }
    hipEventRecord(e0);
    for (int k = 0; k < K; k++) { trash<<<1024, 256>>>(big, nbig, k); stream<WAVES, DEPTH, ROT><<<nwg, WAVES * 64>>>(w, out); }
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_pair, e0, e1);
    hipEventRecord(e0);
    for (int k = 0; k < K; k++) trash<<<1024, 256>>>(big, nbig, k);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_trash, e0, e1);
    hipEventRecord(e0);
    for (int k = 0; k < K; k++) stream<WAVES, DEPTH, ROT><<<nwg, WAVES * 64>>>(w, out);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_hot, e0, e1);
    printf("%-34s wg %3d: after a 64 MB rewrite %6.2f us   back to back (warm L2) %6.2f us\n", name, nwg, (ms_pair - ms_trash) * 1e3 / K, ms_hot * 1e3 / K);
}
int main() {
    uint4 *w; unsigned *out; double2 *big;
    const int nbig = 4 << 20;   // 64 MB
    hipMalloc(&w, NFRAG * 1024); hipMalloc(&out, 1 << 22); hipMalloc(&big, (size_t)nbig * 16);
    hipMemset(w, 1, NFRAG * 1024);
    for (int rep = 0; rep < 2; rep++) {
        run<4, 16, true>("4 waves, 16 in flight, rotated", 128, w, out, big, nbig);
        run<4, 32, true>("4 waves, 32 in flight, rotated", 128, w, out, big, nbig);
        run<4, 56, true>("4 waves, 56 in flight, rotated", 128, w, out, big, nbig);
        run<4, 56, false>("4 waves, 56 in flight, lockstep", 128, w, out, big, nbig);
        run<8, 28, true>("8 waves, 28 in flight, rotated", 128, w, out, big, nbig);
        run<8, 56, true>("8 waves, 56 in flight, rotated", 128, w, out, big, nbig);
        run<16, 28, true>("16 waves, 28 in flight, rotated", 128, w, out, big, nbig);
        run<4, 56, true>("4 waves, 56 in flight, rotated", 256, w, out, big, nbig);
        run<8, 56, true>("8 waves, 56 in flight, rotated", 256, w, out, big, nbig);
        run<4, 56, true>("4 waves, 56 in flight, rotated", 64, w, out, big, nbig);
        run<4, 56, true>("4 waves, 56 in flight, rotated", 1, w, out, big, nbig);
    }
    return 0;
}
