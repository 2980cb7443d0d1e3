// Developer microbenchmark (hazard 1): the matrix-pipe DFT of layouts 10 / 11 issues TRAINS of MFMA
// blocks -- six (four) v_mfma_f32_16x16x32_f16 as two alternating in-place chains, each block one asm statement on its own
// 64-byte line, 12 wait states and the alignment padding (s_nop 0 ...) between two blocks.  mfma_stall_victim.hip showed
// that a gap of 7 or more wait states INSIDE a run of four or more MFMAs corrupts op_sel packed arithmetic of the SIMD's
// other waves.  Is the gap BETWEEN two such blocks (20 .. 70 cycles, new accumulators) one of those?  Same set-up: one
// 16-wave workgroup per CU, waves 0..11 run trains of NB blocks with PAD extra s_nop 0 between them (or VALU work as
// the kernel has between its stages) and check their sums, waves 12..15 run swizzled packed adds / fmas and check
// every lane.  Also the merged form: two four-MFMA blocks in ONE line (eight back to back).
//   hipcc --offload-arch=gfx950 -O3 mfma_block_train.hip -o mfma_block_train && ./mfma_block_train [--quick]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define MF(d, a, b, c) "v_mfma_f32_16x16x32_f16 " d ", " a ", " b ", " c "\n\t"
// one block of the kernel (mma33): re = a0 b0 + a1 b1 + a2 b2, im likewise, alternating; PAD s_nop 0 behind it
template <int PAD> __device__ __forceinline__ void block6(f4 &re, f4 &im, h8 a, h8 b)
{
    asm volatile(".p2align 6\n\ts_nop 1\n\t"
                 MF("%0", "%2", "%3", "0") MF("%1", "%2", "%3", "0") MF("%0", "%2", "%3", "%0")
                 MF("%1", "%2", "%3", "%1") MF("%0", "%2", "%3", "%0") MF("%1", "%2", "%3", "%1")
                 "s_nop 7\n\ts_nop 3\n\t.rept %4\n\ts_nop 0\n\t.endr"
                 : "=&v"(re), "=&v"(im) : "v"(a), "v"(b), "n"(PAD));
}
template <int PAD> __device__ __forceinline__ void block4(f4 &re, f4 &im, h8 a, h8 b)
{
    asm volatile(".p2align 6\n\ts_nop 1\n\t"
                 MF("%0", "%2", "%3", "0") MF("%1", "%2", "%3", "0") MF("%0", "%2", "%3", "%0") MF("%1", "%2", "%3", "%1")
                 "s_nop 7\n\ts_nop 3\n\t.rept %4\n\ts_nop 0\n\t.endr"
                 : "=&v"(re), "=&v"(im) : "v"(a), "v"(b), "n"(PAD));
}
// three MFMAs (one in-place chain), PAD s_nop 0 behind
template <int PAD> __device__ __forceinline__ void block3(f4 &re, h8 a, h8 b)
{
    asm volatile(".p2align 6\n\ts_nop 1\n\t"
                 MF("%0", "%1", "%2", "0") MF("%0", "%1", "%2", "%0") MF("%0", "%1", "%2", "%0")
                 "s_nop 7\n\ts_nop 3\n\t.rept %3\n\ts_nop 0\n\t.endr"
                 : "=&v"(re) : "v"(a), "v"(b), "n"(PAD));
}
// twelve back to back over two lines (48 + 48 bytes: 8 in the first line would leave 4 for the second; 6 + 6 here)
__device__ __forceinline__ void block12(f4 &r0, f4 &i0, f4 &r1, f4 &i1, h8 a, h8 b)
{
    asm volatile("s_nop 1\n\t.p2align 6\n\t"
                 MF("%0", "%4", "%5", "0") MF("%1", "%4", "%5", "0") MF("%0", "%4", "%5", "%0") MF("%1", "%4", "%5", "%1")
                 MF("%0", "%4", "%5", "%0") MF("%1", "%4", "%5", "%1")
                 MF("%2", "%4", "%5", "0") MF("%3", "%4", "%5", "0") MF("%2", "%4", "%5", "%2") MF("%3", "%4", "%5", "%3")
                 MF("%2", "%4", "%5", "%2") MF("%3", "%4", "%5", "%3")
                 "s_nop 7\n\ts_nop 3"
                 : "=&v"(r0), "=&v"(i0), "=&v"(r1), "=&v"(i1) : "v"(a), "v"(b));
}
// two four-MFMA blocks in one 64-byte line: eight back to back (4 + 60 bytes would not fit: the leading s_nop goes in front)
__device__ __forceinline__ void block44(f4 &r0, f4 &i0, f4 &r1, f4 &i1, h8 a, h8 b)
{
    asm volatile("s_nop 1\n\t.p2align 6\n\t"
                 MF("%0", "%4", "%5", "0") MF("%1", "%4", "%5", "0") MF("%0", "%4", "%5", "%0") MF("%1", "%4", "%5", "%1")
                 MF("%2", "%4", "%5", "0") MF("%3", "%4", "%5", "0") MF("%2", "%4", "%5", "%2") MF("%3", "%4", "%5", "%3")
                 "s_nop 7\n\ts_nop 3"
                 : "=&v"(r0), "=&v"(i0), "=&v"(r1), "=&v"(i1) : "v"(a), "v"(b));
}

// FORM 0: four six-blocks, PAD nops between; 1: four four-blocks; 2: four six-blocks with VALU work (a twiddle's worth of
// packed fp32 + converts) between them; 3: two merged 4+4 lines
template <int FORM, int PAD, int PK> __global__ void __launch_bounds__(1024) k(unsigned *out, int iters, unsigned chain_mask, unsigned by_mask)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[16 * 64 * 16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if ((chain_mask >> wv) & 1u) {
        h8 a, t;
        h8 *row = reinterpret_cast<h8 *>(lds) + wv * 64 + lane;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)1.0f; t[i] = (_Float16)(float)(1 + (lane & 3)); }
        *row = t;
        const float per = 32.0f * (float)(1 + (lane & 3));        // one MFMA's contribution
        unsigned nbad = 0;
        for (int it = 0; it < iters; ++it) {
            const h8 b = *row;
            f4 r[4], i[4];
            float want;
            if constexpr (FORM == 0) {
                for (int u = 0; u < 4; ++u) block6<PAD>(r[u], i[u], a, b);
                want = 3.0f * per;
            } else if constexpr (FORM == 1) {
                for (int u = 0; u < 4; ++u) block4<PAD>(r[u], i[u], a, b);
                want = 2.0f * per;
            } else if constexpr (FORM == 2) {
                v2f w = {1.0f, 0.0f};
                asm volatile("" : "+v"(w));
                for (int u = 0; u < 4; ++u) {
                    block6<0>(r[u], i[u], a, b);
                    // (eight packed operations and a few converts, results folded back so that nothing is dropped)
                    v2f x = {r[u].x, i[u].x}, y = {r[u].y, i[u].y};
                    for (int q = 0; q < 4; ++q) { x = x * w.xx; y = y * w.xx; }
                    r[u].x = x.x; i[u].x = x.y; r[u].y = y.x; i[u].y = y.y;
                }
                want = 3.0f * per;
            } else if constexpr (FORM == 3) {
                block44(r[0], i[0], r[1], i[1], a, b);
                block44(r[2], i[2], r[3], i[3], a, b);
                want = 2.0f * per;
            } else if constexpr (FORM == 4) {               // eight three-blocks
                for (int u = 0; u < 4; ++u) { block3<PAD>(r[u], a, b); block3<PAD>(i[u], a, b); }
                want = 3.0f * per;
            } else if constexpr (FORM == 5) {               // ONE six-block per round (the FIR's rhythm)
                block6<0>(r[0], i[0], a, b);
                r[1] = r[2] = r[3] = r[0]; i[1] = i[2] = i[3] = i[0];
                want = 3.0f * per;
            } else if constexpr (FORM == 6) {               // one run of twelve per round
                block12(r[0], i[0], r[1], i[1], a, b);
                r[2] = r[3] = r[0]; i[2] = i[3] = i[0];
                want = 3.0f * per;
            } else {                                        // FORM 7: four six-blocks, PAD x 16 dependent VALU instructions between
                v2f w = {1.0f, 0.0f};
                asm volatile("" : "+v"(w));
                for (int u = 0; u < 4; ++u) {
                    block6<0>(r[u], i[u], a, b);
                    float x = r[u].x;
                    for (int q = 0; q < 16 * PAD; ++q) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(w.x));
                    r[u].x = x;
                }
                want = 3.0f * per;
            }
            for (int u = 0; u < 4; ++u)
                nbad += (r[u].x != want) + (r[u].y != want) + (r[u].z != want) + (r[u].w != want)
                        + (i[u].x != want) + (i[u].y != want) + (i[u].z != want) + (i[u].w != want);
            asm volatile("" : "+v"(a));
            asm volatile("" ::: "memory");
        }
        if (nbad) atomicAdd(&out[0], nbad);
    } else if ((by_mask >> wv) & 1u) {
        v2f x[8];
        for (int j = 0; j < 8; ++j) x[j] = (v2f){(float)(lane + j), (float)(2 * lane + j)};
        const v2f one = {1.0f, 1.0f};
        unsigned nbad = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v2f y;
                    if (PK == 3)      // the kernel's add_mi
                        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(y) : "v"(x[j]), "v"((v2f){-1.0f, 1.0f}));
                    else if (PK == 8)      // broadcast of the low half only (layout 10's window multiply): x + (1, 1) from (1, 7)
                        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(y) : "v"(x[j]), "v"((v2f){1.0f, 7.0f}));
                    else if (PK == 9)      // layout 10's noise scaling: g (broadcast low) * n + c
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(y) : "v"((v2f){1.0f, 7.0f}), "v"(one), "v"(x[j]));
                    else if (PK == 12) {   // split_h's mixed-precision pair (op_sel on the f16 source): y = x + 1 through f16 halves
                        // hi = f16 pair of x (small integers: exact); l = (f16)(x + 1 - hi) = (1, 1); y = x + float(l)
                        uint32_t hi, l;
                        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x[j].x), "v"(x[j].y));
                        const float ax = x[j].x + 1.0f, ay = x[j].y + 1.0f;
                        asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(ax));
                        asm volatile("s_nop 0\n\tv_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(ay));
                        y = x[j];
                        y.x += (l & 0xFFFFu) == 0x3C00u ? 1.0f : 100.0f;
                        y.y += (l >> 16) == 0x3C00u ? 1.0f : 100.0f;
                    }
                    else if (PK == 10)     // no modifier at all
                        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(y) : "v"(x[j]), "v"(one));
                    else if (PK == 11)     // neg only (layout 10's twiddle)
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(y) : "v"(x[j]), "v"(one), "v"((v2f){-1.0f, -1.0f}));
                    else              // cmul's second instruction
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(y) : "v"(x[j]), "v"(one));
                    x[j] = y;
                }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float w0 = (float)(lane + j) + 16.0f, w1 = (float)(2 * lane + j) + 16.0f;
                nbad += (x[j].x != w0) + (x[j].y != w1);
                x[j] = (v2f){(float)(lane + j), (float)(2 * lane + j)};
                asm volatile("" : "+v"(x[j]));
            }
        }
        if (nbad) { atomicAdd(&out[1], nbad); atomicAdd(&out[2 + (lane >> 4)], nbad); }
    }
}

static bool g_quick = false;
template <int FORM, int PAD, int PK> void run(const char *what, unsigned chain_mask = 0x0FFFu, unsigned by_mask = 0xF000u)
{
    unsigned *d;
    hipMalloc(&d, 64 * sizeof(unsigned));
    hipMemset(d, 0, 64 * sizeof(unsigned));
    const int iters = g_quick ? 2000 : 10000;
    hipLaunchKernelGGL((k<FORM, PAD, PK>), dim3(256), dim3(1024), 0, 0, d, iters, chain_mask, by_mask);
    hipDeviceSynchronize();
    std::vector<unsigned> h(64);
    hipMemcpy(h.data(), d, 64 * sizeof(unsigned), hipMemcpyDeviceToHost);
    if (g_quick) printf("RESULT form=%d pad=%d bystander=%d chains_bad=%u bystanders_bad=%u\n", FORM, PAD, PK, h[0], h[1]);
    printf("%-64s blocks' wrong sums %8u of %.2e; bystanders' wrong values %8u of %.2e (by lane row: %u %u %u %u)\n", what, h[0],
           256.0 * __builtin_popcount(chain_mask) * 64 * 32 * iters, h[1], 256.0 * __builtin_popcount(by_mask) * 64 * 16 * iters,
           h[2], h[3], h[4], h[5]);
    hipFree(d);
}

int main(int argc, char **argv)
{
    if (argc > 1 && argv[1][0] == '-' && argv[1][1] == '-' && argv[1][2] == 'q') {
        g_quick = true;
        // (tests/test_gpu_parity.py::test_mfma_trains_only_hit_op_sel_swizzles) the worst train -- three-blocks -- and the
        // kernel's own -- six-blocks, and with a twiddle's VALU work between -- against every packed / mixed form layouts
        // 10 / 11 contain; then the one form they do NOT contain
        run<4, 0, 12>("three-blocks; bystander: split_h");
        run<4, 0, 8>("three-blocks; bystander: v_pk_add_f32 op_sel_hi:[1,0]");
        run<4, 0, 9>("three-blocks; bystander: v_pk_fma_f32 op_sel_hi:[0,1,1]");
        run<4, 0, 11>("three-blocks; bystander: v_pk_fma_f32 neg only");
        run<4, 0, 10>("three-blocks; bystander: v_pk_add_f32 without modifier");
        run<0, 0, 12>("six-blocks; bystander: split_h");
        run<2, 0, 9>("six-blocks + VALU; bystander: v_pk_fma_f32 op_sel_hi:[0,1,1]");
        run<0, 0, 8>("six-blocks; bystanders 0..11 (older waves): op_sel_hi:[1,0]", 0xF000u, 0x0FFFu);
        run<4, 0, 3>("three-blocks; bystander: v_pk_add_f32 op_sel (add_mi)");
        return 0;
    }
    run<0, 0, 3>("six-blocks, aligned (4 s_nop 0 of padding), no extra pad");
    run<5, 0, 3>("ONE six-block per round (the FIR's rhythm)");
    run<6, 0, 3>("one run of twelve per round");
    run<4, 0, 3>("three-blocks, 12 wait states + padding between");
    run<4, 5, 3>("three-blocks, 5 more s_nop 0");
    run<4, 0, 7>("three-blocks, pk_fma bystanders");
    run<4, 0, 3>("three-blocks, bystanders 0..11, chains 12..15", 0xF000u, 0x0FFFu);
    run<7, 1, 3>("six-blocks, 16 dependent v_mul between");
    run<7, 2, 3>("six-blocks, 32 dependent v_mul between");
    run<7, 4, 3>("six-blocks, 64 dependent v_mul between");
    run<7, 8, 3>("six-blocks, 128 dependent v_mul between");
    run<7, 16, 3>("six-blocks, 256 dependent v_mul between");
    // which bystander instructions are hit (six-block trains)
    run<0, 0, 8>("  bystander: v_pk_add_f32 op_sel_hi:[1,0] (low half broadcast)");
    run<0, 0, 9>("  bystander: v_pk_fma_f32 op_sel_hi:[0,1,1]");
    run<0, 0, 10>("  bystander: v_pk_add_f32 without modifier");
    run<0, 0, 11>("  bystander: v_pk_fma_f32 neg only");
    run<0, 0, 7>("  bystander: v_pk_fma_f32 op_sel (cmul)");
    run<0, 0, 12>("  bystander: v_cvt_pk_f16_f32 + v_fma_mixlo/mixhi_f16 (split_h)");
    run<4, 0, 12>("  three-blocks; bystander: split_h");
    run<4, 0, 8>("  three-blocks; bystander: v_pk_add_f32 op_sel_hi:[1,0]");
    run<4, 0, 9>("  three-blocks; bystander: v_pk_fma_f32 op_sel_hi:[0,1,1]");
    run<4, 0, 11>("  three-blocks; bystander: v_pk_fma_f32 neg only");
    run<4, 0, 10>("  three-blocks; bystander: v_pk_add_f32 without modifier");
    return 0;
}
