// Developer microbenchmark: what a wave64 vector instruction costs on gfx950 as a function of instruction kind, of the number of
// INDEPENDENT dependency chains a wave runs (ILP 1, 2, 4) and of the waves per SIMD (1 .. 4).  Every instruction is inline asm, so the
// compiler neither packs nor removes anything.  hipcc --offload-arch=gfx950 -O3 valu_dep.hip -o valu_dep
// Output: cycles of SIMD time per wave-instruction (kernel time x clock / instructions per SIMD) -- at ILP 1 and one wave this is the
// dependent-issue latency, with enough waves / chains the issue cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP 64
enum { K_FMA, K_PKFMA, K_XOR, K_BITOP3, K_MADU64, K_SIN, K_ADDU, K_ALIGN, K_MIX, K_MIXED_PHILOX, K_CVTPK, K_CVTF32, K_CVTSDWA, K_PKADD, K_SUB, K_MIXF32, K_MUL, K_PKMUL, K_LOG, K_SQRT, K_CVTU, K_LSHL, K_ADD3, K_LSHLADD, K_PERM, K_CNDMASK, K_MULU24, K_AND, K_OR, K_MOV, K_MAXF, K_ADDF, K_LSHR, K_BFE, K_ANDOR, K_CVTPKU8, K_RCP, K_CMP, K_CNDS, K_SUBU, K_MULLO, K_MULHI, K_FMAC, K_FMAMK, K_PKMULB, K_COS, K_MED3, K_DPP, K_COUNT };

template <int KIND, int ILP> __global__ void __launch_bounds__(1024) k(uint32_t *out, int iters)
{
    uint32_t a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    uint32_t b0 = a0 + 11, b1 = a1 + 13, b2 = a2 + 17, b3 = a3 + 19;       // second words (64-bit pairs / second chain state)
    const uint32_t m0 = 0xD2511F53u, m1 = 0xCD9E8D57u;
    uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {1.f, 2.f}, p1 = {3.f, 4.f}, p2 = {5.f, 6.f}, p3 = {7.f, 8.f};
    const f2 pc = {1.0001f, 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
#define CH(x, y, q, p)                                                                                            \
    do {                                                                                                          \
        if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(1.0001f));                     \
        if (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(pc));                     \
        if (KIND == K_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(y));                               \
        if (KIND == K_BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x) : "v"(y), "s"(m0)); \
        if (KIND == K_MADU64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q) : "v"((uint32_t)q), "s"(m0) : "vcc"); \
        if (KIND == K_SIN) asm volatile("v_sin_f32 %0, %0" : "+v"(x));                                            \
        if (KIND == K_ADDU) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));                              \
        if (KIND == K_ALIGN) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(x));                              \
        if (KIND == K_MIX) asm volatile("v_fma_mixlo_f16 %0, %0, -1.0, %1 op_sel_hi:[1,0,0]" : "+v"(x) : "v"(y)); \
        if (KIND == K_CVTPK) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x) : "v"(y));                      \
        if (KIND == K_CVTF32) asm volatile("v_cvt_f32_f16_e32 %0, %0" : "+v"(x));                                  \
        if (KIND == K_CVTSDWA) asm volatile("v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(x)); \
        if (KIND == K_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(pc));                          \
        if (KIND == K_SUB) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(x) : "v"(y));                            \
        if (KIND == K_MIXF32) asm volatile("v_fma_mix_f32 %0, %0, -1.0, %1 op_sel_hi:[1,0,0]" : "+v"(x) : "v"(y)); \
        if (KIND == K_MUL) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(x) : "v"(y));                            \
        if (KIND == K_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(pc));                          \
        if (KIND == K_LOG) asm volatile("v_log_f32_e32 %0, %0" : "+v"(x));                                         \
        if (KIND == K_SQRT) asm volatile("v_sqrt_f32_e32 %0, %0" : "+v"(x));                                       \
        if (KIND == K_CVTU) asm volatile("v_cvt_f32_u32_e32 %0, %0" : "+v"(x));                                    \
        if (KIND == K_LSHL) asm volatile("v_lshlrev_b32_e32 %0, 3, %0" : "+v"(x));                                 \
        if (KIND == K_ADD3) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x) : "v"(y));                          \
        if (KIND == K_LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x) : "v"(y));                    \
        if (KIND == K_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "s"(m0));                 \
        if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(y));               \
        if (KIND == K_MULU24) asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(x) : "v"(y));                     \
        if (KIND == K_AND) asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(x) : "v"(y));                            \
        if (KIND == K_OR) asm volatile("v_or_b32_e32 %0, %0, %1" : "+v"(x) : "v"(y));                              \
        if (KIND == K_MOV) asm volatile("v_mov_b32_e32 %0, %1\n\tv_mov_b32_e32 %1, %0" : "+v"(x), "+v"(y));        \
        if (KIND == K_MAXF) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(x) : "v"(y));                           \
        if (KIND == K_ADDF) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(x) : "v"(y));                           \
        if (KIND == K_LSHR) asm volatile("v_lshrrev_b32_e32 %0, 3, %0" : "+v"(x));                                 \
        if (KIND == K_BFE) asm volatile("v_bfe_u32 %0, %0, 3, 28" : "+v"(x));                                      \
        if (KIND == K_ANDOR) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(x) : "v"(y));                       \
        if (KIND == K_CVTPKU8) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(x) : "v"(y));                   \
        if (KIND == K_RCP) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(x));                                         \
        if (KIND == K_CMP) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1\n\tv_add_f32_e32 %0, %0, %1" : "+v"(x) : "v"(y) : "vcc"); \
        if (KIND == K_CNDS) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "s"((uint64_t)0x5555555555555555ull)); \
        if (KIND == K_SUBU) asm volatile("v_sub_u32_e32 %0, %0, %1" : "+v"(x) : "v"(y));                           \
        if (KIND == K_MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(y));                           \
        if (KIND == K_MULHI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(y));                           \
        if (KIND == K_FMAC) asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(x) : "v"(y));                          \
        if (KIND == K_FMAMK) asm volatile("v_fmamk_f32 %0, %0, 0x2f800000, %1" : "+v"(x) : "v"(y));                \
        if (KIND == K_PKMULB) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p) : "v"(pc));         \
        if (KIND == K_COS) asm volatile("v_cos_f32_e32 %0, %0" : "+v"(x));                                         \
        if (KIND == K_MED3) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));                          \
        if (KIND == K_DPP) asm volatile("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x)); \
        if (KIND == K_MIXED_PHILOX) {                                                                             \
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q) : "v"(x), "s"(m0) : "vcc");                 \
            asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(x) : "v"((uint32_t)(q >> 32)), "v"(y), "s"(m1)); \
        }                                                                                                         \
    } while (0)
            CH(a0, b0, q0, p0);
            if (ILP >= 2) CH(a1, b1, q1, p1);
            if (ILP >= 4) { CH(a2, b2, q2, p2); CH(a3, b3, q3, p3); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (uint32_t)q0 + (uint32_t)q1 + (uint32_t)q2 + (uint32_t)q3
                                                 + (uint32_t)(p0.x + p1.x + p2.x + p3.x);
}

static double clock_ghz = 2.4;

template <int KIND, int ILP> void run(const char *name, int waves_per_simd)
{
    uint32_t *d;
    hipMalloc(&d, 256 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = 256 * waves_per_simd, iters = 4000;
    k<KIND, ILP><<<256, threads>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND, ILP><<<256, threads>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_chain = (KIND == K_MIXED_PHILOX) ? 2.0 : 1.0;
    const double instr_per_wave = (double)iters * REP * ILP * per_chain;
    const double cyc_wave = ms * 1e6 * clock_ghz / instr_per_wave;            // cycles per instruction as one wave sees it
    printf("%-14s ILP %d  waves/SIMD %d  %8.3f ms   %6.2f cycles per instr of a wave   %6.2f cycles of SIMD time per instr\n", name, ILP,
           waves_per_simd, ms, cyc_wave, cyc_wave / waves_per_simd);
    hipFree(d);
}

template <int KIND> void sweep(const char *name)
{
    for (int w = 1; w <= 4; ++w) { run<KIND, 1>(name, w); }
    for (int w = 1; w <= 4; ++w) { run<KIND, 2>(name, w); }
    for (int w = 1; w <= 4; w += 1) { run<KIND, 4>(name, w); }
}
// (short form: dependent latency of one wave, and the issue cost with four chains in each of four waves)
template <int KIND> void brief(const char *name)
{
    run<KIND, 1>(name, 1);
    run<KIND, 4>(name, 3);
    run<KIND, 4>(name, 4);
}

int main(int argc, char **argv)
{
    if (argc > 1) clock_ghz = atof(argv[1]);
    const bool only_brief = argc > 2;
    if (!only_brief) {
    sweep<K_FMA>("v_fma_f32");
    sweep<K_PKFMA>("v_pk_fma_f32");
    sweep<K_XOR>("v_xor_b32");
    sweep<K_ADDU>("v_add_u32");
    sweep<K_BITOP3>("v_bitop3_b32");
    sweep<K_ALIGN>("v_alignbit");
    sweep<K_MADU64>("v_mad_u64_u32");
    sweep<K_SIN>("v_sin_f32");
    sweep<K_MIX>("v_fma_mixlo");
    sweep<K_MIXED_PHILOX>("mad+bitop3");
    }
    if (argc > 3) goto more;
    brief<K_CVTPK>("v_cvt_pk_f16");
    brief<K_CVTF32>("v_cvt_f32_f16");
    brief<K_CVTSDWA>("cvt_f32_f16sdwa");
    brief<K_PKADD>("v_pk_add_f32");
    brief<K_SUB>("v_sub_f32");
    brief<K_MIXF32>("v_fma_mix_f32");
    brief<K_MUL>("v_mul_f32");
    brief<K_PKMUL>("v_pk_mul_f32");
    brief<K_LOG>("v_log_f32");
    brief<K_SQRT>("v_sqrt_f32");
    brief<K_CVTU>("v_cvt_f32_u32");
    brief<K_LSHL>("v_lshlrev_b32");
    brief<K_ADD3>("v_add3_u32");
    brief<K_LSHLADD>("v_lshl_add_u32");
    brief<K_PERM>("v_perm_b32");
    brief<K_CNDMASK>("v_cndmask_b32");
    brief<K_MULU24>("v_mul_u32_u24");
more:
    brief<K_AND>("v_and_b32");
    brief<K_OR>("v_or_b32");
    brief<K_MOV>("2x v_mov_b32");
    brief<K_MAXF>("v_max_f32");
    brief<K_ADDF>("v_add_f32");
    brief<K_LSHR>("v_lshrrev_b32");
    brief<K_BFE>("v_bfe_u32");
    brief<K_ANDOR>("v_and_or_b32");
    brief<K_CVTPKU8>("v_cvt_pk_u8_f32");
    brief<K_RCP>("v_rcp_f32");
    brief<K_CMP>("v_cmp+v_add_f32");
    brief<K_CNDS>("v_cndmask_e64");
    brief<K_SUBU>("v_sub_u32");
    brief<K_MULLO>("v_mul_lo_u32");
    brief<K_MULHI>("v_mul_hi_u32");
    brief<K_FMAC>("v_fmac_f32");
    brief<K_FMAMK>("v_fmamk_f32");
    brief<K_PKMULB>("v_pk_mul bcast");
    brief<K_COS>("v_cos_f32");
    brief<K_MED3>("v_med3_f32");
    brief<K_DPP>("v_add_u32_dpp");
    return 0;
}
