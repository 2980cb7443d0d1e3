// Developer microbenchmark: does fp32 MFMA (16x16x4) issued by the same waves overlap with their
// packed-fp32 VALU work on gfx950?  Per iteration a wave runs NM MFMAs (two accumulators) and NV
// v_pk_fma_f32.  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize mfma_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int NM, int NV> __global__ void __launch_bounds__(512) k(float *out, int iters, float a, float b)
{
    f4 d0 = {0, 0, 0, 0}, d1 = {1, 1, 1, 1};
    f2 p[8]; for (int i = 0; i < 8; ++i) p[i] = (f2){(float)threadIdx.x + i, 1.0f};
    const f2 pa = {a, a}, pb = {b, b};
    float av = a + threadIdx.x, bv = b - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < (NM > NV / 4 ? NM : NV / 4); ++i) {
            if (i < NM) {
                if (i & 1) d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, d1, 0, 0, 0);
                else d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, d0, 0, 0, 0);
            }
            if (4 * i < NV) {
#pragma unroll
                for (int j = 0; j < 4; ++j) p[(4 * i + j) & 7] = __builtin_elementwise_fma(p[(4 * i + j) & 7], pa, pb);
            }
        }
    }
    float s = d0[0] + d0[1] + d0[2] + d0[3] + d1[0] + d1[1] + d1[2] + d1[3];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV> void run(const char *name)
{
    float *d; hipMalloc(&d, 512 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    k<NM, NV><<<512, 512>>>(d, 10, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NM, NV><<<512, 512>>>(d, iters, 1.0001f, 0.5f);   // 2 WGs of 8 waves per CU = 4 waves/SIMD
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves x iters iterations
    printf("%-28s %.3f ms  -> %.0f cycles per wave-iteration per SIMD-slot (@2.4 GHz)\n", name, ms,
           ms * 1e-3 * 2.4e9 / (iters * 4.0));
    hipFree(d);
}

int main()
{
    run<0, 280>("280 pk_fma");
    run<70, 0>("70 mfma");
    run<70, 280>("70 mfma + 280 pk_fma");
    run<70, 1120>("70 mfma + 1120 pk_fma");
    run<0, 1120>("1120 pk_fma");
    return 0;
}
