// Does a fresh kernel pay per distinct ARRAY it touches (address translation), or per byte?  (development probe)
// N loads per lane, issued together, either from N arrays far apart or from one array; cycles until all have arrived.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int N>
__global__ void probe(const double *base, size_t stride_elems, unsigned long long *out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    double v[N];
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = base[(size_t)k * stride_elems + i];
    double s = 0;
#pragma unroll
    for (int k = 0; k < N; k++) s += v[k];
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (s == 12345.678 ? 1 : 0);
}
__global__ void dirty(double *p, size_t n) { for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0; }
int main() {
    const int G = 1024, B = 256; const size_t n = (size_t)G * B;
    const size_t far = 8u << 20;   // 64 MB apart (in doubles)
    double *buf; hipMalloc(&buf, sizeof(double) * (far * 16 + n)); unsigned long long *out; hipMalloc(&out, 8 * G);
    double *other; hipMalloc(&other, 256u << 20);
    std::vector<unsigned long long> h(G);
    auto run = [&](const char *name, auto kern, size_t stride) {
        double med[5];
        for (int r = 0; r < 5; r++) {
            hipLaunchKernelGGL(dirty, dim3(2048), dim3(256), 0, 0, other, (size_t)(256u << 20) / 8);   // another kernel in between: caches / TLBs turn over
            hipLaunchKernelGGL(kern, dim3(G), dim3(B), 0, 0, (const double *)buf, stride, out);
            hipMemcpy(h.data(), out, 8 * G, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end()); med[r] = (double)h[G / 2];
        }
        std::sort(med, med + 5);
        printf("%-44s median cycles to collect: %8.0f   (p90 of last run %llu)\n", name, med[2], h[G * 9 / 10]);
    };
    hipLaunchKernelGGL(dirty, dim3(2048), dim3(256), 0, 0, buf, far * 16 + n);
    run("1 load", probe<1>, far);
    run("8 loads, one array (contiguous 8 x 2 MB)", probe<8>, n);
    run("8 loads, eight arrays 64 MB apart", probe<8>, far);
    run("14 loads, one array", probe<14>, n);
    run("14 loads, fourteen arrays 64 MB apart", probe<14>, far);
    return 0;
}
