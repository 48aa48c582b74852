// Development check: does v_mfma_f32_32x32x16_f16 flush fp16 subnormal inputs?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float a_val, float b_val, float *out)
{
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    a[0] = (_Float16)a_val;   // every lane: A[row][k0] (k0 = 0 or 8)
    b[0] = (_Float16)b_val;
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
}
int main()
{
    float *d; hipMalloc(&d, 8);
    for (float av : {1.0f, 9.5367431640625e-07f /* 2^-20 */, 3.0517578125e-05f /* 2^-15 */}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, av, 1024.0f, d);
        float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a = %g (as f16 %g) x 1024 summed over 2 k-groups -> %g (expected %g)\n", av, h[1], h[0], 2 * 1024.0 * h[1]);
    }
    return 0;
}
