// Development microbenchmark: HBM read rate of the A-operand access pattern of the layer-0 backward GEMM
// (256-row tiles, 128 B per row per k stage) for two layouts of the [n][M*H1] gradient buffer.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/membench tools/membench.hip && /tmp/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));

// mode 0: row-major [n][2048], stage kt reads cols 32kt..32kt+31 of 256 rows
// mode 1: member-major [8][n][256], stage kt -> member kt/8, cols 32(kt%8)
// mode 2: row-major but 64 floats (256 B) per row per stage
template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void k_read(const float *A, int64_t n, float *out)
{
    const int tid = threadIdx.x, srow = tid >> 2, piece = tid & 3;
    const int64_t r0 = (int64_t)blockIdx.x * 256 + srow, r1 = r0 + 128;
    v4f acc = {0, 0, 0, 0};
    const int NK = MODE == 2 ? 32 : 64;
    for (int k0 = 0; k0 < NK; k0 += DEPTH) {
        v4f v[DEPTH][4];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int kt = k0 + d;
            const float *p0, *p1;
            if (MODE == 0) {
                p0 = A + r0 * 2048 + kt * 32 + piece * 8;
                p1 = A + r1 * 2048 + kt * 32 + piece * 8;
            } else if (MODE == 1) {
                p0 = A + (int64_t)(kt >> 3) * n * 256 + r0 * 256 + (kt & 7) * 32 + piece * 8;
                p1 = A + (int64_t)(kt >> 3) * n * 256 + r1 * 256 + (kt & 7) * 32 + piece * 8;
            } else {
                p0 = A + r0 * 2048 + kt * 64 + piece * 16;
                p1 = A + r1 * 2048 + kt * 64 + piece * 16;
            }
            v[d][0] = *(const v4f *)p0;
            v[d][1] = *(const v4f *)(p0 + 4);
            v[d][2] = *(const v4f *)p1;
            v[d][3] = *(const v4f *)(p1 + 4);
            if (MODE == 2) {
                v[d][0] += *(const v4f *)(p0 + 8);
                v[d][1] += *(const v4f *)(p0 + 12);
                v[d][2] += *(const v4f *)(p1 + 8);
                v[d][3] += *(const v4f *)(p1 + 12);
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d][0] + v[d][1] + v[d][2] + v[d][3];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int MODE, int DEPTH>
static void run(const char *name, const float *A, int64_t n, float *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = (int)(n / 256);
    hipLaunchKernelGGL((k_read<MODE, DEPTH>), dim3(blocks), dim3(512), 0, 0, A, n, out);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_read<MODE, DEPTH>), dim3(blocks), dim3(512), 0, 0, A, n, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    printf("%-40s %8.3f ms  %7.1f GB/s\n", name, ms, (double)n * 2048 * 4 / ms / 1e6);
}

int main()
{
    const int64_t n = 262144;  // one MLP chunk: 2 GiB
    float *A, *out;
    hipMalloc(&A, (size_t)n * 2048 * 4);
    hipMalloc(&out, 64);
    hipMemset(A, 0, (size_t)n * 2048 * 4);
    run<0, 2>("row-major 128B/row/stage depth2", A, n, out);
    run<0, 4>("row-major 128B/row/stage depth4", A, n, out);
    run<0, 8>("row-major 128B/row/stage depth8", A, n, out);
    run<1, 2>("member-major 128B depth2", A, n, out);
    run<1, 4>("member-major 128B depth4", A, n, out);
    run<1, 8>("member-major 128B depth8", A, n, out);
    run<2, 2>("row-major 256B/row/stage depth2", A, n, out);
    run<2, 4>("row-major 256B/row/stage depth4", A, n, out);
    hipFree(A);
    hipFree(out);
    return 0;
}
