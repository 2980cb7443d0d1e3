// Developer microbenchmark: does v_mfma_f32_16x16x32_f16 see an operand word that a VALU instruction wrote RIGHT in front of it?
// The matrix-pipe transforms build their operands with split_h (v_cvt_pk_f16_f32 + v_fma_mixlo_f16 + v_fma_mixhi_f16, the mix
// instructions as inline asm -- which the compiler's hazard recognizer cannot look into) and the compiler may place the first MFMA of a
// chain one instruction behind the last v_fma_mixhi_f16.  Here: word 3 of the B (or A) operand holds STALE bits, is rewritten by
// cvt / mixlo / mixhi (or a plain v_mov_b32), GAP independent VALU instructions follow, then the MFMA; the result is compared with
// the same product computed with 16 wait states in between.  One wave per workgroup (alone on its SIMD) or sixteen (four per SIMD).
//   hipcc --offload-arch=gfx950 -O3 mfma_after_mix.hip -o mfma_after_mix && ./mfma_after_mix [--quick]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <stdint.h>
#include <string>
typedef float f4 __attribute__((ext_vector_type(4)));

// KIND 0: the rewritten word is word 3 of B, written by mixlo + mixhi;  1: word 3 of A likewise;  2: word 3 of B by v_mov_b32;
// 3: word 3 of B by v_cvt_pk_f16_f32 alone.  GAP = independent v_add_u32 between the write and the MFMA.
template <int KIND, int GAP> __device__ __forceinline__ f4 probe(float x, float y, uint32_t stale, uint32_t w0, uint32_t w1, uint32_t w2,
                                                                 uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, bool safe)
{
    f4 d;
    uint32_t junk = 0;
    // fixed registers: v[40:43] = the operand under test, v[44:47] = the other operand, v48 = f16 pair of (x, y)
    if (safe) {
        asm volatile("v_mov_b32 v40, %2\n\tv_mov_b32 v41, %3\n\tv_mov_b32 v42, %4\n\tv_mov_b32 v43, %5\n\t"
                     "v_mov_b32 v44, %6\n\tv_mov_b32 v45, %7\n\tv_mov_b32 v46, %8\n\tv_mov_b32 v47, %9\n\t"
                     "s_nop 7\n\t"
                     "v_cvt_pk_f16_f32 v48, %10, %11\n\t"
                     ".if %12 == 2\n\tv_mov_b32 v43, v48\n\t.endif\n\t"
                     ".if %12 == 3\n\tv_cvt_pk_f16_f32 v43, %10, %11\n\t.endif\n\t"
                     ".if %12 < 2\n\t"
                     "v_fma_mixlo_f16 v43, v48, -1.0, %10 op_sel_hi:[1,0,0]\n\t"
                     "v_fma_mixhi_f16 v43, v48, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                     ".endif\n\t"
                     "s_nop 7\n\ts_nop 7\n\t"
                     ".if %12 == 1\n\tv_mfma_f32_16x16x32_f16 %0, v[40:43], v[44:47], 0\n\t.else\n\tv_mfma_f32_16x16x32_f16 %0, v[44:47], v[40:43], 0\n\t.endif\n\t"
                     "s_nop 7\n\ts_nop 7\n\ts_nop 3"
                     : "=&v"(d), "+v"(junk)
                     : "v"(w0), "v"(w1), "v"(w2), "v"(stale), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(x), "v"(y), "n"(KIND)
                     : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48");
    } else {
        asm volatile("v_mov_b32 v40, %2\n\tv_mov_b32 v41, %3\n\tv_mov_b32 v42, %4\n\tv_mov_b32 v43, %5\n\t"
                     "v_mov_b32 v44, %6\n\tv_mov_b32 v45, %7\n\tv_mov_b32 v46, %8\n\tv_mov_b32 v47, %9\n\t"
                     "s_nop 7\n\t"
                     "v_cvt_pk_f16_f32 v48, %10, %11\n\t"
                     ".if %12 == 2\n\tv_mov_b32 v43, v48\n\t.endif\n\t"
                     ".if %12 == 3\n\tv_cvt_pk_f16_f32 v43, %10, %11\n\t.endif\n\t"
                     ".if %12 < 2\n\t"
                     "v_fma_mixlo_f16 v43, v48, -1.0, %10 op_sel_hi:[1,0,0]\n\t"
                     "v_fma_mixhi_f16 v43, v48, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                     ".endif\n\t"
                     ".rept %13\n\tv_add_u32 %1, 1, %1\n\t.endr\n\t"
                     ".if %12 == 1\n\tv_mfma_f32_16x16x32_f16 %0, v[40:43], v[44:47], 0\n\t.else\n\tv_mfma_f32_16x16x32_f16 %0, v[44:47], v[40:43], 0\n\t.endif\n\t"
                     "s_nop 7\n\ts_nop 7\n\ts_nop 3"
                     : "=&v"(d), "+v"(junk)
                     : "v"(w0), "v"(w1), "v"(w2), "v"(stale), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(x), "v"(y), "n"(KIND), "n"(GAP)
                     : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48");
    }
    return d;
}

__device__ __forceinline__ uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s; }
// a packed pair of moderate f16 values from random bits (exponents 12..17: 2^-3 .. 2^2)
__device__ __forceinline__ uint32_t h2rand(uint32_t &s)
{
    const uint32_t r = lcg(s);
    const uint32_t lo = ((r & 0x8000u) | ((12u + ((r >> 10) & 3u)) << 10) | (r & 0x3FFu));
    const uint32_t hi = (((r >> 16) & 0x8000u) | ((12u + ((r >> 26) & 3u)) << 10) | ((r >> 16) & 0x3FFu));
    return lo | (hi << 16);
}

template <int KIND, int GAP> __global__ void k(unsigned long long *bad, int iters)
{
    uint32_t s = 12345u + 977u * (blockIdx.x * blockDim.x + threadIdx.x);
    unsigned long long nb = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t w0 = h2rand(s), w1 = h2rand(s), w2 = h2rand(s), stale = h2rand(s) | 0x40004000u;   // stale: large values
        const uint32_t a0 = h2rand(s), a1 = h2rand(s), a2 = h2rand(s), a3 = h2rand(s);
        const float x = (float)(int)(lcg(s) >> 8) * (1.0f / 8388608.0f), y = (float)(int)(lcg(s) >> 8) * (1.0f / 4194304.0f);
        const f4 ref = probe<KIND, GAP>(x, y, stale, w0, w1, w2, a0, a1, a2, a3, true);
        const f4 got = probe<KIND, GAP>(x, y, stale, w0, w1, w2, a0, a1, a2, a3, false);
        nb += (__float_as_uint(ref.x) != __float_as_uint(got.x)) | (__float_as_uint(ref.y) != __float_as_uint(got.y))
              | (__float_as_uint(ref.z) != __float_as_uint(got.z)) | (__float_as_uint(ref.w) != __float_as_uint(got.w));
    }
    if (nb) atomicAdd(bad, nb);
}

template <int KIND, int GAP> static unsigned long long run(int threads, int iters)
{
    unsigned long long *d, h = 0;
    hipMalloc(&d, 8);
    hipMemset(d, 0, 8);
    k<KIND, GAP><<<256, threads>>>(d, iters);
    hipDeviceSynchronize();
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    hipFree(d);
    return h;
}

template <int KIND> static int kind(const char *what, int iters)
{
    int fails = 0;
    for (int threads : {64, 1024}) {
        const unsigned long long b0 = run<KIND, 0>(threads, iters), b1 = run<KIND, 1>(threads, iters), b2 = run<KIND, 2>(threads, iters),
                                 b4 = run<KIND, 4>(threads, iters), b8 = run<KIND, 8>(threads, iters);
        printf("%-58s %2d waves/CU: wrong results with 0 / 1 / 2 / 4 / 8 vector instructions in between: %llu / %llu / %llu / %llu / %llu of %llu\n",
               what, threads / 64, b0, b1, b2, b4, b8, (unsigned long long)256 * threads * iters);
        printf("RESULT kind=%d waves=%d total=%llu gap0=%llu gap1=%llu gap2=%llu gap4=%llu gap8=%llu\n", KIND, threads / 64,
               (unsigned long long)256 * threads * iters, b0, b1, b2, b4, b8);
        fails += (b0 | b1 | b2 | b4 | b8) != 0;
    }
    return fails;
}

int main(int argc, char **argv)
{
    const bool quick = argc > 1 && std::string(argv[1]) == "--quick";
    const int iters = quick ? 2000 : 40000;
    int fails = 0;
    fails += kind<0>("B word 3 <- v_fma_mixlo_f16 + v_fma_mixhi_f16, then MFMA", iters);
    fails += kind<1>("A word 3 <- v_fma_mixlo_f16 + v_fma_mixhi_f16, then MFMA", iters);
    fails += kind<2>("B word 3 <- v_mov_b32, then MFMA", iters);
    fails += kind<3>("B word 3 <- v_cvt_pk_f16_f32, then MFMA", iters);
    printf(fails ? "HAZARD SEEN: a VALU write right in front of the MFMA is not always seen by it\n" : "no hazard seen\n");
    return 0;
}
