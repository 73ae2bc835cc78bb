// What does the kernel-argument fetch cost a short kernel after a kernel boundary?  The same kernel (one cold 16-byte load per lane of a
// buffer another kernel has just rewritten, one store) with its arguments fetched by the wavefront's first s_load (default) and with the
// leading arguments preloaded into SGPRs by the dispatcher (-mllvm -amdgpu-kernarg-preload-count=N): build twice, compare.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/kp0 tools/micro/kernarg_preload.hip
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=12 -o /tmp/kp1 tools/micro/kernarg_preload.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Tail { const double *x; int pad[8]; };
__global__ void __launch_bounds__(256) touch(const double2 *__restrict__ a, double2 *__restrict__ b, const double2 *__restrict__ c, int n, int mode, Tail tl) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double2 v = a[i];
    if (mode == 1) {   // a dependent second round trip, like the step kernel's prologue (index from the first load)
        const int j = (int)(((unsigned long long)__double_as_longlong(v.x)) % (unsigned)n);
        const double2 w = c[j];
        v.x += w.x; v.y += w.y;
    }
    b[i] = v;
}
__global__ void __launch_bounds__(256) rewrite(double2 *a, int n, int k) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { a[i].x = __longlong_as_double((long long)((i * 2654435761u + k) % (unsigned)n)); a[i].y = k; }
}
int main() {
    const int n = 1024 * 256;   // the step kernel's grid: 1024 workgroups of 256
    double2 *a, *b, *c;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&c, n * 16);
    hipMemset(c, 0, n * 16);
    Tail tl{}; tl.x = nullptr;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            const int K = 2000;
            for (int k = 0; k < 50; k++) { rewrite<<<1024, 256>>>(a, n, k); touch<<<1024, 256>>>((const double2 *)a, b, (const double2 *)c, n, mode, tl); }
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int k = 0; k < K; k++) { rewrite<<<1024, 256>>>(a, n, k); touch<<<1024, 256>>>((const double2 *)a, b, (const double2 *)c, n, mode, tl); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            float ms2;
            hipEventRecord(e0);
            for (int k = 0; k < K; k++) rewrite<<<1024, 256>>>(a, n, k);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms2, e0, e1);
            printf("mode %d rep %d: pair %.3f us, rewrite alone %.3f us, touch ~ %.3f us\n", mode, rep, ms * 1e3 / K, ms2 * 1e3 / K, (ms - ms2) * 1e3 / K);
        }
    }
    return 0;
}
