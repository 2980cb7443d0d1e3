// Developer microbenchmark: does bf16 MFMA (v_mfma_f32_32x32x16_bf16, the dense matrix pipe of
// gfx950) overlap with packed-fp32 / integer VALU work?  Three arrangements per SIMD (4 waves):
//   same-wave : every wave interleaves NM MFMAs with NV VALU instructions
//   split     : waves 0,1 of each SIMD run only MFMAs, waves 2,3 only VALU (different waves)
// hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize mfma_bf16_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

// MODE 0: pk_fma VALU, MODE 1: v_mad_u64_u32 (Philox-like) VALU, MODE 2: scalar v_fma_f32
template <int NM, int NV, int MODE, bool SPLIT>
__global__ void __launch_bounds__(512) k(float *out, int iters, float a, float b, unsigned long long seed)
{
    f16v d0 = {}, d1 = {};
    for (int i = 0; i < 16; ++i) { d0[i] = (float)i; d1[i] = 1.0f; }
    bf8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (__bf16)(a + threadIdx.x + i); bv[i] = (__bf16)(b - i); }
    f2 p[8]; for (int i = 0; i < 8; ++i) p[i] = (f2){(float)threadIdx.x + i, 1.0f};
    const f2 pa = {a, a}, pb = {b, b};
    unsigned long long q[4] = {seed + threadIdx.x, seed * 3 + 1, seed ^ 0x9E3779B9ull, seed + 77};
    const int wave = threadIdx.x >> 6;                      // 8 waves per workgroup, 2 WGs per CU
    const bool do_m = !SPLIT || (wave & 1) == 0, do_v = !SPLIT || (wave & 1) == 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < (NM > NV / 4 ? NM : NV / 4); ++i) {
            if (i < NM && do_m) {
                if (i & 1) d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, d1, 0, 0, 0);
                else d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, d0, 0, 0, 0);
            }
            if (4 * i < NV && do_v) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (MODE == 0) p[(4 * i + j) & 7] = __builtin_elementwise_fma(p[(4 * i + j) & 7], pa, pb);
                    else if (MODE == 2) p[(4 * i + j) & 7].x = __builtin_fmaf(p[(4 * i + j) & 7].x, a, b);
                    else q[j] = (unsigned long long)(unsigned)q[j] * 0xD2511F53ull + (q[j] >> 32);
                }
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += d0[i] + d1[i];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    for (int i = 0; i < 4; ++i) s += (float)(q[i] & 0xFFFF);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV, int MODE, bool SPLIT> void run(const char *name)
{
    float *d; hipMalloc(&d, 512 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    k<NM, NV, MODE, SPLIT><<<512, 512>>>(d, 10, 1.0001f, 0.5f, 12345ull);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NM, NV, MODE, SPLIT><<<512, 512>>>(d, iters, 1.0001f, 0.5f, 12345ull);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s %8.3f ms -> %7.0f cycles per iteration per SIMD (4 waves, @2.4 GHz)\n", name, ms,
           ms * 1e-3 * 2.4e9 / iters);
    hipFree(d);
}

int main()
{
    run<0, 280, 0, false>("280 pk_fma / wave");
    run<0, 280, 1, false>("280 mad_u64 / wave");
    run<32, 0, 0, false>("32 mfma_bf16_32x32x16 / wave");
    run<32, 280, 0, false>("same wave: 32 mfma + 280 pk_fma");
    run<32, 280, 1, false>("same wave: 32 mfma + 280 mad_u64");
    run<0, 280, 2, false>("280 v_fma_f32 / wave");
    run<32, 280, 2, false>("same wave: 32 mfma + 280 v_fma_f32");
    run<64, 560, 1, true>("split waves: 2x(64 mfma) | 2x(560 mad_u64)");
    return 0;
}
