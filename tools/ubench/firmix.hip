// Developer microbenchmark: instruction mix of phase B of the frame kernel with the FIR on the
// VALU (420 v_pk_fma_f32) versus on the bf16 matrix pipe (60 v_mfma_f32_16x16x32_bf16 + ~110
// integer VALU for the bf16 split + 38 16-byte LDS operations), both next to the RNG of the same
// phase (100 mad_u64, 120 xor3, 44 transcendentals, ~200 scalar fp32) and to a filler of the other
// phases' mix (350 packed + 400 scalar/int).  4 waves per SIMD, 2 workgroups of 8 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int MODE> __global__ void __launch_bounds__(512) k(float *out, int iters, float a, float b, unsigned seed)
{
    __shared__ __attribute__((aligned(16))) float lds[512 * 12];
    f2 p[10]; for (int i = 0; i < 10; ++i) p[i] = (f2){(float)threadIdx.x + i, 1.0f};
    f4 d[5]; for (int i = 0; i < 5; ++i) d[i] = (f4){0, 0, 0, 0};
    float sc[6] = {a, b, a + b, a - b, 1.f, 2.f};
    unsigned q[4] = {seed + threadIdx.x, seed * 3 + 1, seed ^ 0x9E3779B9u, seed + 77};
    const f2 pa = {a, a}, pb = {b, b};
    f4 *l4 = reinterpret_cast<f4 *>(lds) + threadIdx.x;
    for (int i = threadIdx.x; i < 512 * 12; i += 512) lds[i] = (float)i;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        // ---- RNG part of phase B (both variants)
#pragma unroll
        for (int i = 0; i < 100; ++i) {
            const unsigned long long m = (unsigned long long)q[i & 3] * 0xD2511F53ull;
            q[i & 3] = (unsigned)(m >> 32) ^ q[(i + 1) & 3] ^ (unsigned)m;
        }
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            sc[i % 6] = __builtin_amdgcn_sqrtf(__builtin_amdgcn_logf(sc[i % 6] * sc[i % 6] + 1.5f));
            sc[(i + 1) % 6] += __builtin_amdgcn_sinf(sc[i % 6]) * __builtin_amdgcn_cosf(sc[(i + 2) % 6]);
        }
#pragma unroll
        for (int i = 0; i < 200; ++i) sc[i % 6] = __builtin_fmaf(sc[i % 6], a, b);
        // ---- FIR
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 420; ++i) p[i % 10] = __builtin_elementwise_fma(p[i % 10], pa, pb);
        } else {
#pragma unroll
            for (int i = 0; i < 110; ++i) q[i & 3] = (q[i & 3] & 0xFFFF0000u) + (q[(i + 1) & 3] >> 3);
#pragma unroll
            for (int i = 0; i < 8; ++i) l4[512 * (i % 3)] = (f4){sc[0], sc[1], (float)q[0], (float)q[1]};
            __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                bf8 A[2][3], B[2][3];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const f4 x = l4[512 * pl + 64 * ((t + ks) % 4)];
                        B[ks][pl] = __builtin_bit_cast(bf8, x);
                        A[ks][pl] = __builtin_bit_cast(bf8, (f4){x.y, x.x, x.w, x.z});
                    }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    d[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ks][0], B[ks][0], d[t], 0, 0, 0);
                    d[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ks][0], B[ks][1], d[t], 0, 0, 0);
                    d[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ks][1], B[ks][0], d[t], 0, 0, 0);
                    d[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ks][1], B[ks][1], d[t], 0, 0, 0);
                    d[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ks][0], B[ks][2], d[t], 0, 0, 0);
                    d[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ks][2], B[ks][0], d[t], 0, 0, 0);
                }
            }
        }
        // ---- the other phases: packed (FFT) + scalar/int mix
#pragma unroll
        for (int i = 0; i < 350; ++i) p[i % 10] = __builtin_elementwise_fma(p[i % 10], pb, pa);
#pragma unroll
        for (int i = 0; i < 200; ++i) sc[i % 6] = __builtin_fmaf(sc[i % 6], b, a);
#pragma unroll
        for (int i = 0; i < 200; ++i) q[i & 3] = (q[i & 3] << 1) ^ (q[(i + 1) & 3] + 0x9E3779B9u);
    }
    float s = 0.f;
    for (int i = 0; i < 10; ++i) s += p[i].x + p[i].y;
    for (int i = 0; i < 5; ++i) s += d[i].x + d[i].y + d[i].z + d[i].w;
    for (int i = 0; i < 6; ++i) s += sc[i];
    for (int i = 0; i < 4; ++i) s += (float)(q[i] & 0xFFFF);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char *name)
{
    float *d; hipMalloc(&d, 512 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400;
    k<MODE><<<512, 512>>>(d, 4, 1.0001f, 0.5f, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<512, 512>>>(d, iters, 1.0001f, 0.5f, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %8.3f ms -> %7.0f cycles per wave-iteration slot (per SIMD / 4)\n", name, ms,
           ms * 1e-3 * 2.4e9 / iters / 4);
    hipFree(d);
}

int main()
{
    run<0>("FIR on VALU (420 pk_fma)");
    run<1>("FIR on bf16 MFMA (60 mfma + split + LDS)");
    return 0;
}
