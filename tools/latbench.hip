// Development microbenchmark: dependent-load latency (pointer chase) of one lane for several footprints.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/latbench tools/latbench.hip && /tmp/latbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ void k_chase(const unsigned *next, int hops, unsigned *out, long long *cycles)
{
    unsigned p = out[0];   // continue where the previous launch stopped (untouched lines)
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < hops; ++i) p = next[(size_t)p * 32];   // one 128-B line per element
    long long t1 = __builtin_readcyclecounter();
    out[0] = p;
    cycles[0] = t1 - t0;
}

int main()
{
    std::mt19937 rng(1);
    for (size_t mb : {1, 16, 128, 1024, 4096}) {
        const size_t n = mb * 1024 * 1024 / 128;   // lines
        std::vector<unsigned> perm(n), nxt(n);
        std::iota(perm.begin(), perm.end(), 0u);
        std::shuffle(perm.begin() + 1, perm.end(), rng);
        for (size_t i = 0; i < n; ++i) nxt[perm[i]] = perm[(i + 1) % n];
        std::vector<unsigned> host(n * 32, 0);
        for (size_t i = 0; i < n; ++i) host[i * 32] = nxt[i];
        unsigned *d, *out;
        long long *cyc;
        hipMalloc(&d, n * 128);
        hipMalloc(&out, 4);
        hipMalloc(&cyc, 8);
        hipMemcpy(d, host.data(), n * 128, hipMemcpyHostToDevice);
        hipMemset(out, 0, 4);
        const int hops = 20000;
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, d, hops, out, cyc);
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, d, hops, out, cyc);
        long long c;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("footprint %5zu MiB: %7.1f cycles / dependent load\n", mb, (double)c / hops);
        hipFree(d); hipFree(out); hipFree(cyc);
    }
    return 0;
}
