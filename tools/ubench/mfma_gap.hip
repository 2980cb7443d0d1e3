// Developer microbenchmark: is a chain of dependent in-place MFMAs (vDst = SrcC, the FIR's fir_mma block) exact
// when the chain is NOT issued back to back?  Between the third and the fourth of six v_mfma_f32_16x16x32_f16 a
// gap of G instructions is inserted: s_nop 0, or VALU moves of unrelated registers, or nothing but a 64-byte
// line / 4 KB page boundary of the code (padding in front of the block).  Every element of the result must be
// sum_i 32 * a_i * b_i.  (hazards 1 and 3.)
//   hipcc --offload-arch=gfx950 -O3 mfma_gap.hip -o mfma_gap && ./mfma_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define STR_(x) #x
#define STR(x) STR_(x)

template <int KIND, int G> __device__ __forceinline__ f4 chain(h8 a, h8 b0, h8 b1, h8 b2, h8 b3, h8 b4, h8 b5)
{
    f4 d;
    float t0 = 1.f, t1 = 2.f;
    if constexpr (KIND == 0) {          // gap of G s_nop 0
        asm volatile("s_nop 1\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %1, %2, 0\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %1, %3, %0\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %1, %4, %0\n\t"
                     ".rept %8\n\ts_nop 0\n\t.endr\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %1, %5, %0\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %1, %6, %0\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %1, %7, %0\n\t"
                     "s_nop 7\n\ts_nop 7"
                     : "=&v"(d) : "v"(a), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "n"(G));
    } else if constexpr (KIND == 2) {   // gap of one s_sleep G (64 G cycles)
        asm volatile("s_nop 1\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %1, %2, 0\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %1, %3, %0\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %1, %4, %0\n\t"
                     "s_sleep %8\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %1, %5, %0\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %1, %6, %0\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %1, %7, %0\n\t"
                     "s_nop 7\n\ts_nop 7"
                     : "=&v"(d) : "v"(a), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "n"(G));
    } else {                            // gap of G VALU instructions on unrelated registers
        asm volatile("s_nop 1\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %3, %4, 0\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %3, %5, %0\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %3, %6, %0\n\t"
                     ".rept %10\n\tv_fma_f32 %1, %1, %2, %2\n\t.endr\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %3, %7, %0\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %3, %8, %0\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %3, %9, %0\n\t"
                     "s_nop 7\n\ts_nop 7"
                     : "=&v"(d), "+v"(t0), "+v"(t1) : "v"(a), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "n"(G));
        if (t0 == 12345.f) d.x += t1;
    }
    return d;
}

template <int KIND, int G> __global__ void __launch_bounds__(256) k(unsigned *bad, int iters)
{
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        // small integers: every product and sum is exact in f16 / f32
        const int s = (it * 7 + threadIdx.x / 64) & 7;
        h8 a, b[6];
        for (int i = 0; i < 8; ++i) a[i] = (_Float16)1.0f;
        float want = 0.f;
        for (int m = 0; m < 6; ++m) {
            const float v = (float)(((s + 3 * m) & 7) + 1);
            for (int i = 0; i < 8; ++i) b[m][i] = (_Float16)v;
            want += 32.f * v;
        }
        const f4 d = chain<KIND, G>(a, b[0], b[1], b[2], b[3], b[4], b[5]);
        nbad += (d.x != want) + (d.y != want) + (d.z != want) + (d.w != want);
    }
    if (nbad) atomicAdd(bad, nbad);
}


// KIND 2: no instruction in the gap at all -- the chain's code crosses a 64-byte line P bytes into it, 640 copies
// in a row (80-120 KB of code, more than the 64 KB instruction cache), so that every pass misses at every line.
template <int P> __global__ void __launch_bounds__(256) kline(unsigned *bad, int iters)
{
    const unsigned one = 0x3C003C00u;                        // (1.0, 1.0) in f16
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        const int s = (it * 7 + threadIdx.x / 64) & 7;
        unsigned w[6];
        float want = 0.f;
        for (int m = 0; m < 6; ++m) {
            const int v = ((s + 3 * m) & 7) + 1;
            const _Float16 h = (_Float16)(float)v;
            unsigned short us = __builtin_bit_cast(unsigned short, h);
            w[m] = us | ((unsigned)us << 16);
            want += 32.f * v;
        }
        float t0, t1, t2, t3;
        asm volatile(
            "v_mov_b32 v108, %4\n\tv_mov_b32 v109, %4\n\tv_mov_b32 v110, %4\n\tv_mov_b32 v111, %4\n\t"
            "v_mov_b32 v112, %5\n\tv_mov_b32 v113, %5\n\tv_mov_b32 v114, %5\n\tv_mov_b32 v115, %5\n\t"
            "v_mov_b32 v116, %6\n\tv_mov_b32 v117, %6\n\tv_mov_b32 v118, %6\n\tv_mov_b32 v119, %6\n\t"
            "v_mov_b32 v120, %7\n\tv_mov_b32 v121, %7\n\tv_mov_b32 v122, %7\n\tv_mov_b32 v123, %7\n\t"
            "v_mov_b32 v124, %8\n\tv_mov_b32 v125, %8\n\tv_mov_b32 v126, %8\n\tv_mov_b32 v127, %8\n\t"
            "v_mov_b32 v128, %9\n\tv_mov_b32 v129, %9\n\tv_mov_b32 v130, %9\n\tv_mov_b32 v131, %9\n\t"
            "v_mov_b32 v132, %10\n\tv_mov_b32 v133, %10\n\tv_mov_b32 v134, %10\n\tv_mov_b32 v135, %10\n\t"
            "v_mov_b32 v104, 0\n\tv_mov_b32 v105, 0\n\tv_mov_b32 v106, 0\n\tv_mov_b32 v107, 0\n\t"
            ".rept 640\n\t"
            ".p2align 6\n\t"
            ".rept %11\n\ts_nop 0\n\t.endr\n\t"
            "s_nop 1\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[112:115], 0\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[116:119], v[100:103]\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[120:123], v[100:103]\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[124:127], v[100:103]\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[128:131], v[100:103]\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[132:135], v[100:103]\n\t"
            "s_nop 7\n\ts_nop 7\n\t"
            "v_add_f32 v104, v104, v100\n\tv_add_f32 v105, v105, v101\n\tv_add_f32 v106, v106, v102\n\tv_add_f32 v107, v107, v103\n\t"
            ".endr\n\t"
            "v_mov_b32 %0, v104\n\tv_mov_b32 %1, v105\n\tv_mov_b32 %2, v106\n\tv_mov_b32 %3, v107"
            : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3)
            : "v"(one), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "n"(P / 4)
            : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135");
        const float tot = 640.f * want;
        nbad += (t0 != tot) + (t1 != tot) + (t2 != tot) + (t3 != tot);
    }
    if (nbad) atomicAdd(bad, nbad);
}
template <int P> void runline(unsigned *d_bad)
{
    hipMemset(d_bad, 0, 4);
    kline<P><<<1024, 256>>>(d_bad, 20);
    unsigned h = 0;
    hipMemcpy(&h, d_bad, 4, hipMemcpyDeviceToHost);
    printf("code line boundary %2d bytes into the block (s_nop at 0, MFMA i at 4 + 8 i): %u wrong sums of %.1e%s\n", 64 - P, h,
           1024.0 * 256 * 20 * 4, h ? "   <--" : "");
}
template <int P0, int P1> void sweepline(unsigned *d_bad)
{
    if constexpr (P0 <= P1) { runline<P0>(d_bad); sweepline<P0 + 4, P1>(d_bad); }
}


// KIND 4: the block crosses a 4 KB page of the code P bytes into it; one launch per kernel, so that every wave
// meets the second page for the first time inside the chain.
template <int P> __global__ void __launch_bounds__(256) kpage(unsigned *bad)
{
    const unsigned one = 0x3C003C00u;
    const int s = (blockIdx.x + threadIdx.x / 64) & 7;
    unsigned w[6];
    float want = 0.f;
    for (int m = 0; m < 6; ++m) {
        const int v = ((s + 3 * m) & 7) + 1;
        const _Float16 h = (_Float16)(float)v;
        unsigned short us = __builtin_bit_cast(unsigned short, h);
        w[m] = us | ((unsigned)us << 16);
        want += 32.f * v;
    }
    float t0, t1, t2, t3;
    asm volatile(
        "v_mov_b32 v108, %4\n\tv_mov_b32 v109, %4\n\tv_mov_b32 v110, %4\n\tv_mov_b32 v111, %4\n\t"
        "v_mov_b32 v112, %5\n\tv_mov_b32 v113, %5\n\tv_mov_b32 v114, %5\n\tv_mov_b32 v115, %5\n\t"
        "v_mov_b32 v116, %6\n\tv_mov_b32 v117, %6\n\tv_mov_b32 v118, %6\n\tv_mov_b32 v119, %6\n\t"
        "v_mov_b32 v120, %7\n\tv_mov_b32 v121, %7\n\tv_mov_b32 v122, %7\n\tv_mov_b32 v123, %7\n\t"
        "v_mov_b32 v124, %8\n\tv_mov_b32 v125, %8\n\tv_mov_b32 v126, %8\n\tv_mov_b32 v127, %8\n\t"
        "v_mov_b32 v128, %9\n\tv_mov_b32 v129, %9\n\tv_mov_b32 v130, %9\n\tv_mov_b32 v131, %9\n\t"
        "v_mov_b32 v132, %10\n\tv_mov_b32 v133, %10\n\tv_mov_b32 v134, %10\n\tv_mov_b32 v135, %10\n\t"
        "s_branch 1f\n\t"
        ".p2align 12\n\t"
        "1:\n\t"
        ".rept %11\n\ts_nop 0\n\t.endr\n\t"
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[112:115], 0\n\t"
        "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[116:119], v[100:103]\n\t"
        "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[120:123], v[100:103]\n\t"
        "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[124:127], v[100:103]\n\t"
        "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[128:131], v[100:103]\n\t"
        "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[132:135], v[100:103]\n\t"
        "s_nop 7\n\ts_nop 7\n\t"
        "v_mov_b32 %0, v100\n\tv_mov_b32 %1, v101\n\tv_mov_b32 %2, v102\n\tv_mov_b32 %3, v103"
        : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3)
        : "v"(one), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "n"((4096 - P) / 4)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135");
    const unsigned nbad = (t0 != want) + (t1 != want) + (t2 != want) + (t3 != want);
    if (nbad) atomicAdd(bad, nbad);
}
template <int P> void runpage(unsigned *d_bad)
{
    hipMemset(d_bad, 0, 4);
    kpage<P><<<1024, 256>>>(d_bad);
    unsigned h = 0;
    hipMemcpy(&h, d_bad, 4, hipMemcpyDeviceToHost);
    printf("code page boundary %2d bytes into the block, first launch: %u wrong sums of %.1e%s\n", P, h, 1024.0 * 256 * 4, h ? "   <--" : "");
    hipMemset(d_bad, 0, 4);
    kpage<P><<<1024, 256>>>(d_bad);
    hipMemcpy(&h, d_bad, 4, hipMemcpyDeviceToHost);
    if (h) printf("        ... second launch: %u wrong sums   <--\n", h);
}
template <int P0, int P1> void sweeppage(unsigned *d_bad)
{
    if constexpr (P0 <= P1) { runpage<P0>(d_bad); sweeppage<P0 + 4, P1>(d_bad); }
}


// KIND 5: write-after-read on the chain's operands.  The B operand of MFMA number WHICH (0..5) is overwritten with
// zeros G wait states behind the LAST MFMA of the chain: if a queued dependent MFMA fetched its operands only when
// it starts, that term would be missing from the sum.
template <int WHICH, int G> __global__ void __launch_bounds__(256) kwar(unsigned *bad, int iters)
{
    const unsigned one = 0x3C003C00u;
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        const int s = (it * 7 + threadIdx.x / 64) & 7;
        unsigned w[6];
        float want = 0.f;
        for (int m = 0; m < 6; ++m) {
            const int v = ((s + 3 * m) & 7) + 1;
            const _Float16 h = (_Float16)(float)v;
            unsigned short us = __builtin_bit_cast(unsigned short, h);
            w[m] = us | ((unsigned)us << 16);
            want += 32.f * v;
        }
        float t0, t1, t2, t3;
        asm volatile(
            "v_mov_b32 v108, %4\n\tv_mov_b32 v109, %4\n\tv_mov_b32 v110, %4\n\tv_mov_b32 v111, %4\n\t"
            "v_mov_b32 v112, %5\n\tv_mov_b32 v113, %5\n\tv_mov_b32 v114, %5\n\tv_mov_b32 v115, %5\n\t"
            "v_mov_b32 v116, %6\n\tv_mov_b32 v117, %6\n\tv_mov_b32 v118, %6\n\tv_mov_b32 v119, %6\n\t"
            "v_mov_b32 v120, %7\n\tv_mov_b32 v121, %7\n\tv_mov_b32 v122, %7\n\tv_mov_b32 v123, %7\n\t"
            "v_mov_b32 v124, %8\n\tv_mov_b32 v125, %8\n\tv_mov_b32 v126, %8\n\tv_mov_b32 v127, %8\n\t"
            "v_mov_b32 v128, %9\n\tv_mov_b32 v129, %9\n\tv_mov_b32 v130, %9\n\tv_mov_b32 v131, %9\n\t"
            "v_mov_b32 v132, %10\n\tv_mov_b32 v133, %10\n\tv_mov_b32 v134, %10\n\tv_mov_b32 v135, %10\n\t"
            "s_nop 7\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[112:115], 0\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[116:119], v[100:103]\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[120:123], v[100:103]\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[124:127], v[100:103]\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[128:131], v[100:103]\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[132:135], v[100:103]\n\t"
            ".rept %11\n\ts_nop 0\n\t.endr\n\t"
            "v_mov_b32 v[%12], 0\n\tv_mov_b32 v[%12 + 1], 0\n\tv_mov_b32 v[%12 + 2], 0\n\tv_mov_b32 v[%12 + 3], 0\n\t"
            "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
            "v_mov_b32 %0, v100\n\tv_mov_b32 %1, v101\n\tv_mov_b32 %2, v102\n\tv_mov_b32 %3, v103"
            : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3)
            : "v"(one), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "n"(G), "n"(112 + 4 * WHICH)
            : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135");
        nbad += (t0 != want) + (t1 != want) + (t2 != want) + (t3 != want);
    }
    if (nbad) atomicAdd(bad, nbad);
}
template <int WHICH, int G> void runwar(unsigned *d_bad)
{
    hipMemset(d_bad, 0, 4);
    kwar<WHICH, G><<<1024, 256>>>(d_bad, 200);
    unsigned h = 0;
    hipMemcpy(&h, d_bad, 4, hipMemcpyDeviceToHost);
    printf("B operand of MFMA %d zeroed %2d wait states behind the chain: %u wrong sums of %.1e%s\n", WHICH, G, h, 1024.0 * 256 * 200 * 4, h ? "   <--" : "");
}


// KIND 6: the accumulator hops: d2 = a b1 + d1 with d1 = a b0 in DIFFERENT registers (vDst != SrcC), G wait states
// between the two MFMAs -- what the compiler's own scheduling of a chain of builtins produces.
template <int G> __global__ void __launch_bounds__(256) khop(unsigned *bad, int iters)
{
    const unsigned one = 0x3C003C00u;
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        const int s = (it * 7 + threadIdx.x / 64) & 7;
        unsigned w[2];
        float want = 0.f;
        for (int m = 0; m < 2; ++m) {
            const int v = ((s + 3 * m) & 7) + 1;
            const _Float16 h = (_Float16)(float)v;
            unsigned short us = __builtin_bit_cast(unsigned short, h);
            w[m] = us | ((unsigned)us << 16);
            want += 32.f * v;
        }
        float t0, t1, t2, t3;
        asm volatile(
            "v_mov_b32 v108, %4\n\tv_mov_b32 v109, %4\n\tv_mov_b32 v110, %4\n\tv_mov_b32 v111, %4\n\t"
            "v_mov_b32 v112, %5\n\tv_mov_b32 v113, %5\n\tv_mov_b32 v114, %5\n\tv_mov_b32 v115, %5\n\t"
            "v_mov_b32 v116, %6\n\tv_mov_b32 v117, %6\n\tv_mov_b32 v118, %6\n\tv_mov_b32 v119, %6\n\t"
            "v_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
            "s_nop 7\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[112:115], 0\n\t"
            ".rept %7\n\ts_nop 0\n\t.endr\n\t"
            "v_mfma_f32_16x16x32_f16 v[96:99], v[108:111], v[116:119], v[100:103]\n\t"
            "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
            "v_mov_b32 %0, v96\n\tv_mov_b32 %1, v97\n\tv_mov_b32 %2, v98\n\tv_mov_b32 %3, v99"
            : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3)
            : "v"(one), "v"(w[0]), "v"(w[1]), "n"(G)
            : "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135");
        nbad += (t0 != want) + (t1 != want) + (t2 != want) + (t3 != want);
    }
    if (nbad) atomicAdd(bad, nbad);
}
template <int G> void runhop(unsigned *d_bad)
{
    hipMemset(d_bad, 0, 4);
    khop<G><<<1024, 256>>>(d_bad, 200);
    unsigned h = 0;
    hipMemcpy(&h, d_bad, 4, hipMemcpyDeviceToHost);
    printf("accumulator hop (vDst != SrcC), %2d wait states between the MFMAs: %u wrong sums of %.1e%s\n", G, h, 1024.0 * 256 * 200 * 4, h ? "   <--" : "");
}
template <int G0, int G1> void sweephop(unsigned *d_bad)
{
    if constexpr (G0 <= G1) { runhop<G0>(d_bad); sweephop<G0 + 1, G1>(d_bad); }
}


// KIND 7: the result of the chain's last MFMA read by a VALU instruction G wait states behind it (16 waves per CU
// all doing the same, so that the matrix pipe is contended).
template <int G> __global__ void __launch_bounds__(256) kraw(unsigned *bad, int iters)
{
    const unsigned one = 0x3C003C00u;
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        const int s = (it * 7 + threadIdx.x / 64) & 7;
        unsigned w[6];
        float want = 0.f;
        for (int m = 0; m < 6; ++m) {
            const int v = ((s + 3 * m) & 7) + 1;
            const _Float16 h = (_Float16)(float)v;
            unsigned short us = __builtin_bit_cast(unsigned short, h);
            w[m] = us | ((unsigned)us << 16);
            want += 32.f * v;
        }
        float t0, t1, t2, t3;
        asm volatile(
            "v_mov_b32 v108, %4\n\tv_mov_b32 v109, %4\n\tv_mov_b32 v110, %4\n\tv_mov_b32 v111, %4\n\t"
            "v_mov_b32 v112, %5\n\tv_mov_b32 v113, %5\n\tv_mov_b32 v114, %5\n\tv_mov_b32 v115, %5\n\t"
            "v_mov_b32 v116, %6\n\tv_mov_b32 v117, %6\n\tv_mov_b32 v118, %6\n\tv_mov_b32 v119, %6\n\t"
            "v_mov_b32 v120, %7\n\tv_mov_b32 v121, %7\n\tv_mov_b32 v122, %7\n\tv_mov_b32 v123, %7\n\t"
            "v_mov_b32 v124, %8\n\tv_mov_b32 v125, %8\n\tv_mov_b32 v126, %8\n\tv_mov_b32 v127, %8\n\t"
            "v_mov_b32 v128, %9\n\tv_mov_b32 v129, %9\n\tv_mov_b32 v130, %9\n\tv_mov_b32 v131, %9\n\t"
            "v_mov_b32 v132, %10\n\tv_mov_b32 v133, %10\n\tv_mov_b32 v134, %10\n\tv_mov_b32 v135, %10\n\t"
            "v_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
            "s_nop 7\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[112:115], 0\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[116:119], v[100:103]\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[120:123], v[100:103]\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[124:127], v[100:103]\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[128:131], v[100:103]\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], v[108:111], v[132:135], v[100:103]\n\t"
            ".rept %11\n\ts_nop 0\n\t.endr\n\t"
            "v_mov_b32 %0, v100\n\tv_mov_b32 %1, v101\n\tv_mov_b32 %2, v102\n\tv_mov_b32 %3, v103\n\t"
            "s_nop 7\n\ts_nop 7"
            : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
            : "v"(one), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "n"(G)
            : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135");
        nbad += (t0 != want) + (t1 != want) + (t2 != want) + (t3 != want);
    }
    if (nbad) atomicAdd(bad, nbad);
}
template <int G> void runraw(unsigned *d_bad)
{
    hipMemset(d_bad, 0, 4);
    kraw<G><<<1024, 256>>>(d_bad, 200);
    unsigned h = 0;
    hipMemcpy(&h, d_bad, 4, hipMemcpyDeviceToHost);
    printf("result read by the VALU %2d wait states behind the last MFMA: %u wrong values of %.1e%s\n", G, h, 1024.0 * 256 * 200 * 4, h ? "   <--" : "");
}
template <int G0, int G1> void sweepraw(unsigned *d_bad)
{
    if constexpr (G0 <= G1) { runraw<G0>(d_bad); sweepraw<G0 + 1, G1>(d_bad); }
}

template <int KIND, int G> void run(unsigned *d_bad)
{
    hipMemset(d_bad, 0, 4);
    k<KIND, G><<<1024, 256>>>(d_bad, KIND == 2 ? 200 : 2000);
    unsigned h = 0;
    hipMemcpy(&h, d_bad, 4, hipMemcpyDeviceToHost);
    printf("%s gap %3d: %u wrong elements of %.1e%s\n", KIND == 0 ? "s_nop" : (KIND == 2 ? "s_sleep" : "VALU "), G, h, 1024.0 * 256 * 2000 * 4, h ? "   <--" : "");
}

template <int KIND, int G0, int G1> void sweep(unsigned *d_bad)
{
    if constexpr (G0 <= G1) { run<KIND, G0>(d_bad); sweep<KIND, G0 + 1, G1>(d_bad); }
}

int main()
{
    unsigned *d_bad; hipMalloc(&d_bad, 4);
    sweeppage<4, 60>(d_bad);            // (first: nothing else of this code object has run yet)
    sweepraw<0, 16>(d_bad);
    sweephop<0, 16>(d_bad);
    runwar<5, 0>(d_bad); runwar<5, 1>(d_bad); runwar<5, 2>(d_bad); runwar<5, 4>(d_bad); runwar<5, 8>(d_bad); runwar<5, 16>(d_bad); runwar<5, 32>(d_bad);
    runwar<4, 0>(d_bad); runwar<4, 4>(d_bad); runwar<3, 0>(d_bad); runwar<2, 0>(d_bad); runwar<1, 0>(d_bad);
    sweep<0, 0, 24>(d_bad);
    run<0, 32>(d_bad); run<0, 48>(d_bad); run<0, 64>(d_bad); run<0, 100>(d_bad);
    sweep<1, 1, 16>(d_bad);
    run<1, 24>(d_bad); run<1, 32>(d_bad); run<1, 64>(d_bad);
    run<2, 1>(d_bad); run<2, 2>(d_bad); run<2, 4>(d_bad); run<2, 8>(d_bad); run<2, 16>(d_bad); run<2, 32>(d_bad); run<2, 64>(d_bad); run<2, 127>(d_bad);
    sweepline<0, 60>(d_bad);
    return 0;
}
