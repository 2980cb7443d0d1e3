// Developer microbenchmark: VALU issue rate on gfx950 for the instruction kinds the frame
// kernel is made of, at 1/2/4 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP 256
template <int KIND> __global__ void __launch_bounds__(1024) k(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    uint32_t u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (KIND == 0) {          // v_fma_f32, 8 independent chains
                x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
                x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
            } else if (KIND == 1) {   // v_pk_fma_f32, 4 chains x 2 (same 8 fma's worth of flops in 4 instr)
                p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
                p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
                p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
                p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
            } else if (KIND == 2) {   // v_mad_u64_u32 (Philox multiply)
                uint64_t m0 = (uint64_t)u0 * 0xD2511F53u, m1 = (uint64_t)u1 * 0xCD9E8D57u;
                uint64_t m2 = (uint64_t)u2 * 0xD2511F53u, m3 = (uint64_t)u3 * 0xCD9E8D57u;
                u0 = (uint32_t)(m0 >> 32) ^ (uint32_t)m1; u1 = (uint32_t)(m1 >> 32) ^ (uint32_t)m2;
                u2 = (uint32_t)(m2 >> 32) ^ (uint32_t)m3; u3 = (uint32_t)(m3 >> 32) ^ (uint32_t)m0;
            } else if (KIND == 3) {   // v_xor_b32
                u0 ^= u1; u1 ^= u2; u2 ^= u3; u3 ^= u0; u0 ^= u2; u1 ^= u3; u2 ^= u0; u3 ^= u1;
            } else if (KIND == 5) {   // v_mul_hi_u32
                u0 = __umulhi(u0, 0xD2511F53u) ^ u1; u1 = __umulhi(u1, 0xCD9E8D57u) ^ u2;
                u2 = __umulhi(u2, 0xD2511F53u) ^ u3; u3 = __umulhi(u3, 0xCD9E8D57u) ^ u0;
            } else if (KIND == 6) {   // v_mul_lo_u32
                u0 = (u0 * 0xD2511F53u) ^ u1; u1 = (u1 * 0xCD9E8D57u) ^ u2;
                u2 = (u2 * 0xD2511F53u) ^ u3; u3 = (u3 * 0xCD9E8D57u) ^ u0;
            } else if (KIND == 7) {   // v_pk_add_f32
                p0 = p0 + pa; p1 = p1 + pb; p2 = p2 + pa; p3 = p3 + pb;
                p0 = p0 + pb; p1 = p1 + pa; p2 = p2 + pb; p3 = p3 + pa;
            } else if (KIND == 8) {   // v_cvt_f32_u32 + v_add
                x0 += (float)u0; x1 += (float)u1; x2 += (float)u2; x3 += (float)u3;
                u0 += 3; u1 += 5; u2 += 7; u3 += 11;
            } else if (KIND == 4) {   // transcendental v_sin_f32
                x0 = __builtin_amdgcn_sinf(x0); x1 = __builtin_amdgcn_sinf(x1); x2 = __builtin_amdgcn_sinf(x2); x3 = __builtin_amdgcn_sinf(x3);
                x4 = __builtin_amdgcn_sinf(x4); x5 = __builtin_amdgcn_sinf(x5); x6 = __builtin_amdgcn_sinf(x6); x7 = __builtin_amdgcn_sinf(x7);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + u0 + u1 + u2 + u3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int KIND> void run(const char *name, int threads, int instr_per_rep)
{
    float *d; hipMalloc(&d, 256 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 2000;
    k<KIND><<<256, threads>>>(d, 10, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<256, threads>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waves_per_simd = threads / 64 / 4.0;
    double instr_per_wave = (double)iters * (REP / 8) * instr_per_rep;
    double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * waves_per_simd);
    printf("%-14s waves/SIMD=%.0f  %.3f ms  %.2f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz)\n", name,
           waves_per_simd, ms, ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4);
    hipFree(d);
}

int main()
{
    for (int t : {256, 512, 1024}) {
        if (t == 256) { run<0>("v_fma_f32", 256, 8); run<1>("v_pk_fma_f32", 256, 8); run<2>("v_mad_u64_u32", 256, 4); run<3>("v_xor_b32", 256, 8); run<4>("v_sin_f32", 256, 8); }
        if (t == 512) { run<0>("v_fma_f32", 512, 8); run<1>("v_pk_fma_f32", 512, 8); run<2>("v_mad_u64_u32", 512, 4); run<3>("v_xor_b32", 512, 8); run<4>("v_sin_f32", 512, 8); }
        if (t == 1024) { run<5>("mul_hi+xor", 1024, 8); run<6>("mul_lo+xor", 1024, 8); run<7>("v_pk_add_f32", 1024, 8); run<8>("cvt+add+iadd", 1024, 12); run<0>("v_fma_f32", 1024, 8); run<1>("v_pk_fma_f32", 1024, 8); run<2>("v_mad_u64_u32", 1024, 4); run<3>("v_xor_b32", 1024, 8); run<4>("v_sin_f32", 1024, 8); }
    }
    return 0;
}
