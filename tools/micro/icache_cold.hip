// What does a COLD instruction cache cost a short kernel?  (The dispatcher invalidates the instruction and scalar caches at every kernel start.)
// A block of ~10 KB of straight-line code (the size of ev2g_step_wave / ev2g_mlp3_s16) is executed n = 1, 2, 3 times by every wavefront of a
// 1024 x 256 grid: time(n = 2) - time(n = 1) is the block warm, time(n = 1) - launch floor - that is what the first, cold pass adds.
//   hipcc --offload-arch=gfx950 -O3 -o build_variants/icache tools/micro/icache_cold.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define F4(x) x = __builtin_fmaf(x, a, b); x = __builtin_fmaf(x, b, a); x = __builtin_fmaf(x, a, a); x = __builtin_fmaf(x, b, b);
#define F16(x) F4(x) F4(x) F4(x) F4(x)
#define F64(x) F16(x) F16(x) F16(x) F16(x)
#define F256(x) F64(x) F64(x) F64(x) F64(x)
template <int KB> __global__ void __launch_bounds__(256) block(float *out, float a, float b, int n) {
    float x = threadIdx.x * 1e-3f, y = x + 1.f, z = x + 2.f, w = x + 3.f;
    for (int it = 0; it < n; it++) {
        // four independent chains: the block is issue-bound, not latency-bound; 1024 VOP3 FMAs = 8 KB per F256 x 4
        if (KB >= 8) { F256(x) F256(y) F256(z) F256(w) }
        if (KB >= 16) { F256(x) F256(y) F256(z) F256(w) }
        if (KB >= 32) { F256(x) F256(y) F256(z) F256(w) F256(x) F256(y) F256(z) F256(w) }
        asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));
    }
    if (x + y + z + w == 12345.f) out[blockIdx.x * 256 + threadIdx.x] = x;
}
__global__ void __launch_bounds__(256) empty(float *out, int n) { if (n == 12345) out[threadIdx.x] = 1.f; }
template <int KB> void run(float *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int K = 500;
    float t[4];
    for (int n = 0; n <= 3; n++) {
        for (int k = 0; k < 20; k++) block<KB><<<1024, 256>>>(out, 1.0001f, 0.5f, n);
        hipEventRecord(e0);
        for (int k = 0; k < K; k++) block<KB><<<1024, 256>>>(out, 1.0001f, 0.5f, n);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&t[n], e0, e1);
        t[n] *= 1e3f / K;
    }
    printf("%2d KB block: n=0 %.2f us  n=1 %.2f  n=2 %.2f  n=3 %.2f   warm pass %.2f us, first pass %.2f us => cold instruction fetch adds %.2f us\n", KB, t[0], t[1], t[2], t[3],
           t[3] - t[2], t[1] - t[0], (t[1] - t[0]) - (t[3] - t[2]));
}
int main() {
    float *out; hipMalloc(&out, 1 << 22);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int k = 0; k < 20; k++) empty<<<1024, 256>>>(out, 0);
    hipEventRecord(e0);
    for (int k = 0; k < 500; k++) empty<<<1024, 256>>>(out, 0);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("empty kernel, 1024 x 256: %.2f us per launch\n", ms * 1e3 / 500);
    for (int rep = 0; rep < 2; rep++) { run<8>(out); run<16>(out); run<32>(out); }
    return 0;
}
