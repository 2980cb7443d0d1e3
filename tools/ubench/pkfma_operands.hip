// Developer microbenchmark: issue cost of v_pk_fma_f32 with a scalar (SGPR pair) multiplicand --
// the FIR's form, taps as scalars -- versus all-vector operands, and with the op_sel broadcast
// the FIR uses.  4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE> __global__ void __launch_bounds__(512) k(float *out, int iters, f2 sa, f2 sb)
{
    f2 p[10]; for (int i = 0; i < 10; ++i) p[i] = (f2){(float)threadIdx.x + i, 1.0f};
    f2 va = sa + (f2){(float)threadIdx.x * 1e-9f, 0.f}, vb = sb;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 400; ++i) {
            if (MODE == 0) p[i % 10] = __builtin_elementwise_fma(p[i % 10], va, vb);            // vector x vector
            else if (MODE == 1) p[i % 10] = __builtin_elementwise_fma(p[i % 10], sa, vb);       // scalar pair
            else p[i % 10] = __builtin_elementwise_fma(sa.xx, p[(i + 1) % 10], p[i % 10]);       // scalar, broadcast
        }
    }
    float s = 0.f;
    for (int i = 0; i < 10; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char *name)
{
    float *d; hipMalloc(&d, 512 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    k<MODE><<<512, 512>>>(d, 4, (f2){1.0001f, 0.9999f}, (f2){0.5f, 0.25f});
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<512, 512>>>(d, iters, (f2){1.0001f, 0.9999f}, (f2){0.5f, 0.25f});
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms -> %.2f cycles per instruction per SIMD\n", name, ms,
           ms * 1e-3 * 2.4e9 / iters / (4.0 * 400));
    hipFree(d);
}

int main()
{
    run<0>("v_pk_fma_f32 v, v, v");
    run<1>("v_pk_fma_f32 v, s[pair], v");
    run<2>("v_pk_fma_f32 s[pair] (broadcast), v, v");
    return 0;
}
