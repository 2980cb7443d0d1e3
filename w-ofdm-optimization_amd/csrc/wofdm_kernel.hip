// wofdm_kernel.hip -- the fused w-OFDM frame kernel for gfx950 (MI355X, CDNA4).
//
// One workgroup simulates one frame at a time (persistent over a contiguous run of the
// cell-major (cell, frame) work items); one 64-lane wavefront owns one, two, four, eight or sixteen OFDM symbols of
// the frame (layout ids: wofdm_kernel.h):
//
//   A  Philox bits -> Gray QAM (registers) -> N-point IFFT (layouts 10 ... 15: both 16-point DFT stages as split-f16 products on the
//      matrix pipe, registers to registers; the others: in-register 8/16-point DFT stages with one or two exchanges through the
//      wave's own slice of the LDS frame buffer) -> CP/CS copy x Tx window written straight from the last stage, in the
//      matrix-pipe layouts split into two packed-f16 words per sample; the beta-sample fall tail goes to a side buffer
//      (matlab/main_BER_calculation.m:246-252, 358-376, 419-439)
//   -- "barrier" 1: LDS flag of the predecessor wave --
//   B  add the previous symbol's fall tail onto the own rise tail (overlap-add, m:253-259),
//      21-tap complex FIR over the serialised frame (conv, m:260): on the matrix pipe as a block-Toeplitz
//      product in split f16 with fp32 accumulation (layouts 6 ... 15: six v_mfma_f32_16x16x32_f16 per 128
//      samples), on the VALU from LDS in the others; Philox/Box-Muller unit noise for the same samples,
//      per-wave partial signal/noise powers (add_wgn, m:277-294)
//   -- barrier 2 --
//   C  r = c + g n back into the own slice (truncate + reshape, m:261-263), Rx window / fold fused into the
//      first FFT stage's loads (m:297-355; the circular shift is left to the equaliser, see phase C), FFT,
//      pilot wave publishes X0/Y0 (m:266)
//   -- "barrier" 3: LDS flag of the pilot wave --
//   D  one-tap equalise, hard demap, bit/symbol error popcount in registers (m:267-272)
//
// No HBM traffic inside the loop in generate mode (N = 1024 parks its unit noise in an L2-resident
// scratch row): constants come in once per cell, four 64-bit counters go out once per cell.
// Bound by vector issue and by how the waves' latencies interleave ("What bounds the kernel"); in the default
// layouts the matrix pipe carries the FIR AND both transforms.  Template variants add the
// subcarrier allocation and the per-symbol spectral Tx mask of main_channel_mask.m (VAR), injected
// randomness (INJECT) and stage dumps (DUMP).
#include "wofdm_kernel.h"
#include "philox.h"
#include <type_traits>

// Relaxed synchronisation (default): of the three workgroup barriers of a frame only the one in
// front of the noise scaling is a true all-to-all (total powers).  Barrier 1 is "my predecessor
// wave has written its symbols" and barrier 3 "the pilot wave has published the equaliser": both
// become LDS flags carrying the loop iteration number, so that early waves run on into the next
// phase instead of idling at the end of each one.  -DWOFDM_RELAXED_SYNC=0 restores the barriers.
#ifndef WOFDM_RELAXED_SYNC
#define WOFDM_RELAXED_SYNC 1
#endif
// The flag waits spin a bounded number of times, so a protocol error can never hang the GPU; a wave
// that runs out of budget marks an LDS word, which becomes bit 0 of the plan's status word when the
// workgroup retires: wofdm_plan_status and the synchronous entry points then fail with WOFDM_E_HIP
// instead of returning counters built on stale samples.  (-DWOFDM_CHECKED_SYNC=0 drops the mark.)
#ifndef WOFDM_CHECKED_SYNC
#define WOFDM_CHECKED_SYNC 1
#endif
// Fault injection for tests/test_gpu_parity.py::test_lost_flag_is_reported (libwofdm_hip_fault.so
// only): wave 1 of every workgroup "forgets" to publish its symbols in its third frame.
#ifndef WOFDM_FAULT_SKIP_FLAG
#define WOFDM_FAULT_SKIP_FLAG 0
#endif

// N = 512 / 1024: FFT as 8.8.8 / 16.4.16 with the outer stages in registers (fft_big); 0 = the
// radix-4/2 ladder through LDS used for the small sizes
#ifndef WOFDM_FFT_BIG_RADIX
#define WOFDM_FFT_BIG_RADIX 1
#endif

// Received samples stored swizzled against the four-way bank conflicts of the noise-scaling stores (layouts 10 / 11 / 12, phase C)
#ifndef WOFDM_RX_SWIZZLE
#define WOFDM_RX_SWIZZLE 1
#endif
// N = 1024: tiles whose unit noise stays in registers instead of being parked in the HBM scratch row (see phase B).  All nine since
// the end of round 4: with the transforms on the matrix pipe the kernel holds FIR outputs and noise of all tiles at 128 registers
// with seven of them spilled (28 bytes of scratch per lane), and that beats parking any tile's noise in HBM -- 6 kept: 2.62e8,
// 7: 2.70, 8: 2.72, 9: 2.74e8 symbols/s, C4 at full size 7.65 -> 7.30 s (interleaved A/Bs, profiles/r04_noise_keep_ab.txt).  (Round 2
// parked all nine, round 3 three of nine: those kernels, with their transforms on the vector pipe, spilled dozens.)
#ifndef WOFDM_NOISE_KEEP_TILES
#define WOFDM_NOISE_KEEP_TILES 9
#endif

// Issue priority of a wave by the phase it is in (s_setprio, 0 .. 3; round 4).  The SIMD's arbiter picks by priority, then age.  The
// tile loop of phase B is one long run of vector instructions (Philox, Box-Muller) that is always ready to issue; every other
// phase is short bursts of vector work between LDS round trips, MFMA results, flags and the barrier.  At equal priority a wave
// in such a phase queues behind the dense one for every burst and its chain of latencies stretches; with the tile loop BELOW
// everything else the bursts go out at once and the tile loop takes the slots that are left -- which is all of them whenever
// the others wait.  C2: 12.5 -> 11.3 ms (interleaved A/Bs, profiles/r04_prio_ab.txt).  Three levels: the latency-bound parts (bits,
// labels, overlap-add, power sums and barrier, gain and noise scaling, Rx loads, demapping) 2, the transforms and the Tx write 1, the
// tile loop 0 (the transforms at 0 as well: 12.2 ms; at 2: no better than 1; every level above the tile loop alike: +0.8 %).  The
// layouts with one symbol per wave keep their transforms at the upper level: at the middle one N = 512 / 1024 lose 4 % / 8 %.
#ifndef WOFDM_PRIO_A
#define WOFDM_PRIO_A 2
#endif
#ifndef WOFDM_PRIO_TILES
#define WOFDM_PRIO_TILES 0
#endif
#ifndef WOFDM_PRIO_B3
#define WOFDM_PRIO_B3 2
#endif
#ifndef WOFDM_PRIO_C
#define WOFDM_PRIO_C 2
#endif
#ifndef WOFDM_PRIO_D
#define WOFDM_PRIO_D 2
#endif
#ifndef WOFDM_PRIO_X                                 /* the transforms and the Tx write (layouts with four symbols per wave) */
#define WOFDM_PRIO_X 1
#endif
#ifndef WOFDM_PRIO_ON
#define WOFDM_PRIO_ON 1
#endif
#define WAVE_PRIO(x) do { if (WOFDM_PRIO_ON) __builtin_amdgcn_s_setprio(x); } while (0)
// layouts 10 ... 15, generate mode: the FIR tile as a hand-placed pipeline (MFMAs between the Philox rounds; see phase B)
#ifndef WOFDM_TILE_PIPELINE
#define WOFDM_TILE_PIPELINE 1
#endif

// one symbol per wave: the frame's trailing samples ride in the last wave's last tile instead of a tile of their own (phase B)
#ifndef WOFDM_FOLD_TAIL
#define WOFDM_FOLD_TAIL 1
#endif
// layouts 10, 11: the 256-point transforms as a pipeline over the wave's four symbols (phases A and C)
#ifndef WOFDM_MDFT_PIPELINE
#define WOFDM_MDFT_PIPELINE 1
#endif

#ifndef WOFDM_MIN_WAVES_PER_SIMD
#define WOFDM_MIN_WAVES_PER_SIMD 4      // one 16-wave workgroup per CU -> 128 VGPRs per lane
#endif

namespace {

#ifdef WOFDM_STAMP
// Diagnostic build only (tools/stamp_report.py): per-wave cycle totals of the four phases and of
// the three barrier waits, written behind the counters.  Never defined in the shipped library.
#define STAMP(slot)                                                                            \
    do {                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                          \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                    \
        stamp_acc[slot] += now_ - stamp_t;                                                     \
        stamp_t = now_;                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    } while (0)
#ifdef WOFDM_STAMP_A4                     /* the Tx write split into slots 8..10 instead of A1..A3 (developer probe) */
#define STAMPF(slot) do { if ((slot) > 10) STAMP(slot); } while (0)
#define STAMPX(slot) STAMP(slot)
#else
#define STAMPF(slot) STAMP(slot)          /* finer marks inside the phases (slots 8..15) */
#define STAMPX(slot) do { } while (0)
#endif
#ifdef WOFDM_STAMP_MASK                   /* the Tx mask stage split into slots 13..15 instead of phase C */
#define STAMPC(slot) do { } while (0)
#define STAMPM(slot) STAMP(slot)
#else
#define STAMPC(slot) STAMP(slot)
#define STAMPM(slot) do { } while (0)
#endif
#else
// (a comment in the assembly: tools/isa_mix.py splits the frame loop's instruction mix at these)
#define STAMP(slot) asm volatile("; wofdm_mark " #slot)
#define STAMPX(slot) do { } while (0)
#ifdef WOFDM_MMARK      /* developer builds (hipcc -S): the finer marks as comments too */
#define STAMPF(slot) asm volatile("; wofdm_mmark " #slot)
#define STAMPC(slot) asm volatile("; wofdm_mmark " #slot)
#define STAMPM(slot) asm volatile("; wofdm_mmark " #slot)
#else
#define STAMPF(slot) do { } while (0)
#define STAMPC(slot) do { } while (0)
#define STAMPM(slot) do { } while (0)
#endif
#endif

// The FIR's six MFMAs of a tile (fir_mma) are ONE asm block that starts on a 64-byte boundary: its 60 bytes sit in one
// instruction-cache line, so that no instruction fetch can fall between two of them.  Measured on gfx950
// (tools/ubench/mfma_stall_victim.hip, profiles/r03_mfma_stall_victim.txt): when 7 or more wait
// states pass between the fourth or a later MFMA of such a run and the dependent one behind it -- an instruction-fetch
// miss, an s_sleep, VALU instructions scheduled in between -- VOP3P instructions with op_sel (the packed complex
// arithmetic below) executed meanwhile by OTHER waves of the same SIMD return wrong values in lanes 48..63.  The chain
// itself stays exact; up to 6 wait states, and any delay by issue arbitration alone, are harmless.
#define WOFDM_MMA_ALIGN "6"
// Developer build (-DWOFDM_DELAY, tools/delay_probe.py): chosen waves sleep at a chosen point of the frame,
// so that a hole in the wave-to-wave synchronisation shows on every launch instead of once in a cold process.
#ifdef WOFDM_DELAY
#define DELAY_AT(pt)                                                                           \
    do {                                                                                       \
        if (p.delay_point == (pt) && ((p.delay_waves >> wv) & 1u))                             \
            for (uint32_t d_ = 0; d_ < p.delay_len; ++d_) __builtin_amdgcn_s_sleep(127);       \
    } while (0)
#else
#define DELAY_AT(pt) do { } while (0)
#endif

// Complex samples are 2-wide float vectors: gfx950 issues one wave64 VALU instruction per
// ~4 cycles per SIMD whether it is v_fma_f32 or v_pk_fma_f32 (tools/ubench/valu_rate.hip:
// 4.5 vs 5.1 cycles), so the fp32 peak is only reachable with packed math, and complex
// arithmetic packs naturally as (re, im).
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f mk(float x, float y) { return (v2f){x, y}; }

// Matrix-pipe FIR (layouts 6, 7, 8): samples travel through LDS as two packed-f16 words, y = hi + lo with both
// halves rounded to nearest: two 11-bit significands, |y - hi - lo| <= 2^-22 |y|.  Three f16 MFMA terms
// (h_hi x_hi + h_hi x_lo + h_lo x_hi, fp32 accumulation) drop h_lo x_lo, another 2^-22: a product is good to about
// 2^-21, four times coarser than an fp32 product -- on whole frames, against fp64 arithmetic, conv is off by 1.3e-7
// (rms) of the frame's rms, the fp32 VALU form by 1.1e-7: both at the rounding floor of the fp32 stages around them
// (tests/test_gpu_parity.py::test_fir_precision_matrix_pipe_vs_valu).
#ifndef WOFDM_SPLIT_MIX32
#define WOFDM_SPLIT_MIX32 1
#endif
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef h2 hpair;                  // (phase C has a local named h2)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_h(v2f y, uint32_t &hi, uint32_t &lo)
{
    const h2 h = __builtin_convertvector(y, h2);                       // v_cvt_pk_f16_f32 (RNE)
    hi = __builtin_bit_cast(uint32_t, h);
    // lo = f16(y - float(hi)), one mixed-precision fma per half (the difference is exact in fp32, rounded once):
    // three instructions per sample instead of five (two converts back, a packed subtract, a packed convert)
#if WOFDM_SPLIT_MIX32
    // (round 4) the two differences as v_fma_mix_f32 -- 4.3 cycles of vector issue each, where the f16-destination forms
    // v_fma_mixlo / mixhi_f16 take 8.3 like a transcendental and read their own destination (tools/ubench/valu_dep.hip) --
    // and one more packed convert: four instructions, 17 cycles instead of three and 21, and a dependent chain of three, not
    // of three with two slow links.  Same arithmetic: the difference is exact in fp32 and rounded once.
    float dx, dy;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(dx) : "v"(hi), "v"(y.x));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(dy) : "v"(hi), "v"(y.y));
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(mk(dx, dy), h2));
#else
    uint32_t l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(y.x));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(y.y));
    lo = l;
#endif
}
// A matrix operand built from split_h words must not reach its MFMA straight from the vector instruction that wrote it: an operand
// written inside inline asm needs its wait states spelled out (documented: cdna_hip_programming.md 5.7, item 2; measured:
// tools/ubench/mfma_after_mix.hip -- every result wrong with no instruction in between, none with one).  The compiler keeps the
// distance for the instructions it knows; the mix instructions above are inline asm.  Both operand halves pass through this
// statement; csrc/verify_code_layout.py checks the distance in the built library, as part of the build.
__device__ __forceinline__ void mma_operand_fence(h8 &hi, h8 &lo)
{
    // (not volatile: ordered by its operands alone -- as a volatile statement it cost N = 512 one per cent)
    // Two wait states: the documented figure for an operand written inside inline asm (cdna_hip_programming.md 5.7, item 2:
    // `s_nop 1`); verify_code_layout.py demands them of the built code.
    asm("s_nop 1" : "+v"(hi), "+v"(lo));
}
__device__ __forceinline__ v2f join_h(uint32_t hi, uint32_t lo)
{
    return __builtin_convertvector(__builtin_bit_cast(h2, hi), v2f)
           + __builtin_convertvector(__builtin_bit_cast(h2, lo), v2f);
}

__device__ __forceinline__ void wave_sync()
{
    // LDS traffic between lanes of ONE wave: DS ops execute in issue order, so only the
    // compiler has to be kept from reordering across this point.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// a * w  (2 packed instructions; swizzle and sign live in the VOP3P modifiers)
__device__ __forceinline__ v2f cmul(v2f a, v2f w)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// a * conj(w)
__device__ __forceinline__ v2f cmul_conj(v2f a, v2f w)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
        : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// a + (-i) d = (a.x + d.y, a.y - d.x)   and   a + (+i) d = (a.x - d.y, a.y + d.x)
__device__ __forceinline__ v2f add_mi(v2f a, v2f d)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    return r;
}
__device__ __forceinline__ v2f add_pi(v2f a, v2f d)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    return r;
}
// tw tables hold exp(-2 pi i ...): the forward DFT multiplies by them, the inverse by the conjugate
template <int DIR> __device__ __forceinline__ v2f twid(v2f a, v2f w)
{
    return DIR < 0 ? cmul(a, w) : cmul_conj(a, w);
}

template <int DIR> __device__ __forceinline__ void radix4(v2f (&u)[4])
{
    const v2f a0 = u[0] + u[2], a1 = u[0] - u[2];
    const v2f a2 = u[1] + u[3], d = u[1] - u[3];
    u[0] = a0 + a2;
    u[2] = a0 - a2;
    u[1] = DIR < 0 ? add_mi(a1, d) : add_pi(a1, d);      // a1 + (-+ i) d
    u[3] = DIR < 0 ? add_pi(a1, d) : add_mi(a1, d);      // a1 - (-+ i) d
}

template <int N> struct geo {
    static constexpr int NQ = N / 4;                 // radix-4 butterflies per stage
    static constexpr int BPL = (NQ + 63) / 64;       // ... per lane
    static constexpr int RB = N / 64 + 1;            // FIR outputs per lane
    static constexpr bool FULL = NQ >= 64 * BPL;     // every lane owns BPL butterflies
    // Twiddle tables, one per stage after the first, laid out [k][r-1] so that the three
    // factors of a butterfly are adjacent and lanes hit distinct banks:
    //   radix-4 stage NS: 3*NS entries exp(-2 pi i r k / (4 NS));  radix-2 stage NS: NS entries.
    static constexpr int tw_off(int stage_ns)
    {
        // stages in execution order for this N (after the twiddle-free first stage)
        int off = 0, ns = 4;
        while (ns < stage_ns) {
            const bool r2 = (N == 128 && ns == 4) || (N == 512 && ns == 16);
            off += r2 ? ns : 3 * ns;
            ns *= r2 ? 2 : 4;
        }
        return off;
    }
};

// Stockham autosort stages on the wave's LDS slices.  Lane data v[u][q][r] always means element
// (lane + 64 q) + r N/4 of the wave's u-th symbol, both as the first stage's input and the last
// stage's output.  Every stage handles the wave's SPW symbols together (slices `sb` apart), so
// the independent transforms share one write->read turnaround per stage instead of queueing
// behind each other's fences.
template <int N, int DIR, int SPW>
__device__ __forceinline__ void fft_first(v2f (&v)[SPW][geo<N>::BPL][4], v2f *fb, int sb, int lane)
{
#pragma unroll
    for (int u = 0; u < SPW; ++u) {
#pragma unroll
        for (int q = 0; q < geo<N>::BPL; ++q) {
            const int j = lane + 64 * q;
            if (geo<N>::FULL || j < geo<N>::NQ) {
                radix4<DIR>(v[u][q]);
#pragma unroll
                for (int r = 0; r < 4; ++r) fb[u * sb + 4 * j + r] = v[u][q][r];
            }
        }
    }
    wave_sync();
}

template <int N, int NS, int DIR, int SPW>
__device__ __forceinline__ void fft_mid4(v2f *fb, int sb, const v2f *tw, int lane)
{
    v2f u4[SPW][geo<N>::BPL][4];
    const v2f *t = tw + geo<N>::tw_off(NS);
#pragma unroll
    for (int u = 0; u < SPW; ++u) {
#pragma unroll
        for (int q = 0; q < geo<N>::BPL; ++q) {
            const int j = lane + 64 * q;
            if (geo<N>::FULL || j < geo<N>::NQ) {
                const int k = j & (NS - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) u4[u][q][r] = fb[u * sb + j + r * geo<N>::NQ];
#pragma unroll
                for (int r = 1; r < 4; ++r) u4[u][q][r] = twid<DIR>(u4[u][q][r], t[3 * k + r - 1]);
                radix4<DIR>(u4[u][q]);
            }
        }
    }
    wave_sync();
#pragma unroll
    for (int u = 0; u < SPW; ++u) {
#pragma unroll
        for (int q = 0; q < geo<N>::BPL; ++q) {
            const int j = lane + 64 * q;
            if (geo<N>::FULL || j < geo<N>::NQ) {
                const int k = j & (NS - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) fb[u * sb + ((j - k) << 2) + k + r * NS] = u4[u][q][r];
            }
        }
    }
    wave_sync();
}

template <int N, int NS, int DIR, int SPW>
__device__ __forceinline__ void fft_mid2(v2f *fb, int sb, const v2f *tw, int lane)
{
    constexpr int NB = N / 2, PER = (NB + 63) / 64;
    v2f y0[SPW][PER], y1[SPW][PER];
    const v2f *t = tw + geo<N>::tw_off(NS);
#pragma unroll
    for (int u = 0; u < SPW; ++u) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int j = lane + 64 * q;
            if (j < NB) {
                const int k = j & (NS - 1);
                const v2f a = fb[u * sb + j];
                const v2f b = twid<DIR>(fb[u * sb + j + NB], t[k]);
                y0[u][q] = a + b; y1[u][q] = a - b;
            }
        }
    }
    wave_sync();
#pragma unroll
    for (int u = 0; u < SPW; ++u) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int j = lane + 64 * q;
            if (j < NB) {
                const int k = j & (NS - 1);
                fb[u * sb + ((j - k) << 1) + k] = y0[u][q];
                fb[u * sb + ((j - k) << 1) + k + NS] = y1[u][q];
            }
        }
    }
    wave_sync();
}

template <int N, int DIR, int SPW>
__device__ __forceinline__ void fft_last(v2f (&v)[SPW][geo<N>::BPL][4], const v2f *fb, int sb,
                                         const v2f *tw, int lane)
{
    const v2f *t = tw + geo<N>::tw_off(geo<N>::NQ);
#pragma unroll
    for (int u = 0; u < SPW; ++u) {
#pragma unroll
        for (int q = 0; q < geo<N>::BPL; ++q) {
            const int j = lane + 64 * q;
            if (geo<N>::FULL || j < geo<N>::NQ) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[u][q][r] = fb[u * sb + j + r * geo<N>::NQ];
#pragma unroll
                for (int r = 1; r < 4; ++r) v[u][q][r] = twid<DIR>(v[u][q][r], t[3 * j + r - 1]);
                radix4<DIR>(v[u][q]);
            }
        }
    }
    wave_sync();
}

// ---------------------------------------------------------------------------------------------
// N = 512 / 1024: three Stockham stages  R . R2 . R  with R = 8 / 16 points held by one lane
// (N = 8.8.8 = 16.4.16), i.e. two LDS round trips instead of four.  A lane owns elements
// lane + 64 t, t = q + BPL r -- exactly the inputs of butterfly `lane` of a first stage of radix
// R = 4 BPL (Ns = 1) and the outputs of butterfly `lane` of a last stage of radix R (Ns = 64),
// so natural order in and out survives.  The in-register R-point DFT is a radix-4 pass over r,
// constant twiddles, and a radix-4 (radix-2) pass over q.
//
// Twiddle table (fill_twiddles):  N = 1024: [0,48) stage 2 exp(-2 pi i r k/64) at [3k + r-1];
// [48,1008) stage 3 exp(-2 pi i t j/1024) at [48 + 15 j + t-1].  N = 512: [0,56) stage 2
// exp(-2 pi i t k/64) at [7k + t-1]; [56,504) stage 3 exp(-2 pi i t j/512) at [56 + 7j + t-1].
//
// Stage 1 stores R consecutive outputs per lane (stride R v2f across lanes: every lane on the same
// banks); the position inside each group of R is XOR-swizzled with the group number so that a
// store instruction spreads over all banks, and stage 2 undoes it when it loads.
template <int R> __device__ __forceinline__ int swz(int idx)
{
    constexpr int LG = R == 16 ? 4 : 3;
    const int a = idx >> LG;
    return idx ^ ((a ^ (a >> LG)) & (R - 1));
}

// x[q][r] = x_t, t = q + 4 r   ->   x[q][r] = X_u, u = r + 4 q     (16 points)
template <int DIR> __device__ __forceinline__ void dft16(v2f (&x)[4][4])
{
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
    // exp(-2 pi i m/16) for m = q c
    const v2f w1 = mk(c1, -s1), w2 = mk(h, -h), w3 = mk(s1, -c1), w4 = mk(0.f, -1.f), w6 = mk(-h, -h),
              w9 = mk(-c1, s1);
#pragma unroll
    for (int q = 0; q < 4; ++q) radix4<DIR>(x[q]);
    x[1][1] = twid<DIR>(x[1][1], w1); x[1][2] = twid<DIR>(x[1][2], w2); x[1][3] = twid<DIR>(x[1][3], w3);
    x[2][1] = twid<DIR>(x[2][1], w2); x[2][2] = twid<DIR>(x[2][2], w4); x[2][3] = twid<DIR>(x[2][3], w6);
    x[3][1] = twid<DIR>(x[3][1], w3); x[3][2] = twid<DIR>(x[3][2], w6); x[3][3] = twid<DIR>(x[3][3], w9);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        v2f col[4] = {x[0][c], x[1][c], x[2][c], x[3][c]};
        radix4<DIR>(col);
#pragma unroll
        for (int d = 0; d < 4; ++d) x[d][c] = col[d];
    }
}
// x[q][r] = x_t, t = q + 2 r   ->   x[q][r] = X_u, u = r + 4 q     (8 points)
template <int DIR> __device__ __forceinline__ void dft8(v2f (&x)[2][4])
{
    constexpr float h = 0.70710678118654752f;
    radix4<DIR>(x[0]);
    radix4<DIR>(x[1]);
    x[1][1] = twid<DIR>(x[1][1], mk(h, -h));
    x[1][2] = twid<DIR>(x[1][2], mk(0.f, -1.f));
    x[1][3] = twid<DIR>(x[1][3], mk(-h, -h));
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const v2f a = x[0][c], b = x[1][c];
        x[0][c] = a + b;
        x[1][c] = a - b;
    }
}
template <int N, int DIR> __device__ __forceinline__ void dft_lane(v2f (&x)[geo<N>::BPL][4])
{
    if constexpr (N == 1024) dft16<DIR>(x);
    else dft8<DIR>(x);
}

template <int N, int DIR>
__device__ __forceinline__ void fft_big(v2f (&v)[1][geo<N>::BPL][4], v2f *fb, const v2f *tw, int lane)
{
    static_assert(N == 512 || N == 1024, "fft_big is the 8.8.8 / 16.4.16 scheme");
    constexpr int BPL = geo<N>::BPL, R = 4 * BPL;             // 2, 8  or  4, 16
    constexpr int T2 = N == 1024 ? 48 : 56;                    // start of the stage-3 twiddles
    // ---- stage 1: radix R, Ns = 1, from registers; out[R lane + u]
    dft_lane<N, DIR>(v[0]);
#pragma unroll
    for (int q = 0; q < BPL; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) fb[swz<R>(R * lane + r + 4 * q)] = v[0][q][r];
    wave_sync();
    if constexpr (N == 1024) {
        // ---- stage 2: radix 4, Ns = 16: butterflies j = lane + 64 q
        v2f u4[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = lane + 64 * q, k = j & 15;
#pragma unroll
            for (int r = 0; r < 4; ++r) u4[q][r] = fb[swz<16>(j + 256 * r)];
#pragma unroll
            for (int r = 1; r < 4; ++r) u4[q][r] = twid<DIR>(u4[q][r], tw[3 * k + r - 1]);
            radix4<DIR>(u4[q]);
        }
        wave_sync();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = lane + 64 * q, k = j & 15;
#pragma unroll
            for (int r = 0; r < 4; ++r) fb[((j - k) << 2) + k + 16 * r] = u4[q][r];
        }
    } else {
        // ---- stage 2: radix 8, Ns = 8: butterfly j = lane
        v2f u8[2][4];
        const int k = lane & 7;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = q + 2 * r;
                u8[q][r] = fb[swz<8>(lane + 64 * t)];
                if (t > 0) u8[q][r] = twid<DIR>(u8[q][r], tw[7 * k + t - 1]);
            }
        dft8<DIR>(u8);
        wave_sync();
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) fb[((lane - k) << 3) + k + 8 * (r + 4 * q)] = u8[q][r];
    }
    wave_sync();
    // ---- stage 3: radix R, Ns = 64, to registers: in[lane + 64 t], twiddle^(t lane), out lane + 64 u
    v2f x[BPL][4];
#pragma unroll
    for (int q = 0; q < BPL; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = q + BPL * r;
            x[q][r] = fb[lane + 64 * t];
            if (t > 0) x[q][r] = twid<DIR>(x[q][r], tw[T2 + (R - 1) * lane + t - 1]);
        }
    dft_lane<N, DIR>(x);
    // x[q'][r'] = X_u with u = r' + 4 q';  the lane owns u = q + BPL r
#pragma unroll
    for (int q = 0; q < BPL; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int u = q + BPL * r;
            v[0][q][r] = x[u >> 2][u & 3];
        }
    wave_sync();
}

// The same 1024-point transform with a 512-entry (4 KB) exchange buffer: both exchanges run in two
// halves.  Exchange 1: the 16 stage-1 outputs of lane l go to entries 16 l ... 16 l + 15, so lanes 0..31
// fill entries [0, 512) and every lane fetches its r = 0, 1 operands (entries j, j + 256), then lanes
// 32..63 fill the buffer again for r = 2, 3.  Exchange 2: butterfly j = lane + 64 q writes block
// floor(j / 16) = floor(lane / 16) + 4 q of 64 entries, and the last stage reads one entry per block:
// q = 0, 1 (blocks 0..7, inputs t = 0..7) first, q = 2, 3 second -- no extra store there.  Lets all 16
// waves of a workgroup run the 1024-point Tx-mask transforms at once (16 x 4 KB instead of 8 x 8 KB).
template <int DIR>
__device__ __forceinline__ void fft_1024_half(v2f (&v)[1][4][4], v2f *fb, const v2f *tw, int lane)
{
    constexpr int T2 = 48;
    dft_lane<1024, DIR>(v[0]);
    v2f u4[4][4];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        if ((lane >> 5) == hf) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) fb[swz<16>(16 * (lane & 31) + r + 4 * q)] = v[0][q][r];
        }
        wave_sync();
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) u4[q][2 * hf + rr] = fb[swz<16>(lane + 64 * q + 256 * rr)];
        wave_sync();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = (lane + 64 * q) & 15;
#pragma unroll
        for (int r = 1; r < 4; ++r) u4[q][r] = twid<DIR>(u4[q][r], tw[3 * k + r - 1]);
        radix4<DIR>(u4[q]);
    }
    v2f x[4][4];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int k = lane & 15;                      // (lane + 64 q) & 15
#pragma unroll
            for (int r = 0; r < 4; ++r) fb[((lane - k) << 2) + 256 * qq + k + 16 * r] = u4[2 * hf + qq][r];
        }
        wave_sync();
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int r = 2 * hf + rr, t = q + 4 * r;
                x[q][r] = fb[lane + 64 * (t - 8 * hf)];
                if (t > 0) x[q][r] = twid<DIR>(x[q][r], tw[T2 + 15 * lane + t - 1]);
            }
        wave_sync();
    }
    dft_lane<1024, DIR>(x);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int u = q + 4 * r;
            v[0][q][r] = x[u >> 2][u & 3];
        }
}

// ---------------------------------------------------------------------------------------------
// N = 256, four symbols per wave ("quarter-wave" layout): 16 lanes own one symbol, lane l of the
// quarter holds elements l + 16 t, t = q + 4 r.  256 = 16.16: both stages are in-register 16-point
// DFTs with ONE LDS round trip between them.  Twiddles exp(-2 pi i t l/256) at tw[15 l + t-1].
template <int DIR>
__device__ __forceinline__ void fft_qw(v2f (&v)[1][4][4], v2f *sc, const v2f *tw, int ll)
{
    dft16<DIR>(v[0]);                                   // X_u at [q'][r'], u = r' + 4 q'
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[16 * ll + ((r + 4 * q) ^ ll)] = v[0][q][r];
    wave_sync();
    v2f x[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = q + 4 * r;
            x[q][r] = sc[16 * t + (ll ^ t)];            // in[ll + 16 t], un-swizzled
            if (t > 0) x[q][r] = twid<DIR>(x[q][r], tw[15 * ll + t - 1]);
        }
    dft16<DIR>(x);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[0][q][r] = x[r][q];   // the lane owns u = q + 4 r
    wave_sync();
}
__device__ __forceinline__ void fill_twiddles_qw(v2f *tw, int tid, int nthreads)
{
    for (int i = tid; i < 240; i += nthreads) {
        float sv, cv;
        sincospif(-2.0f * (float)((i / 15) * (1 + i % 15)) / 256.0f, &sv, &cv);
        tw[i] = mk(cv, sv);
    }
}

// ---------------------------------------------------------------------------------------------
// N = 256 on the matrix pipe (layouts 10, 11): 256 = 16 . 16, both stages as 16-point DFT matrix products in split f16
// with fp32 accumulation, NO LDS exchange and no cross-lane traffic.  Lane (a = lane % 16, g = lane / 16) holds elements
// lane + 64 j, j < 4, of each of the wave's four symbols, as input and as output.  With n = 16 n1 + n2, k = k1 + 16 k2,
// th = exp(-2 pi i / 16), om = exp(-2 pi i / 256):
//   stage 1   T[n2][k1] = sum_n1 x[16 n1 + n2] th^(n1 k1):  the DATA is the A operand (row m = n2 = a; K slot (g, j) <-> n1 =
//             g + 4 j, i.e. the lane's own four words), the DFT matrix the B operand (column k1); the result comes back as
//             lane (k1 = a, g), element j' = row n2 = 4 g + j' -- which is where the B operand of stage 2 wants it;
//   twiddle   T' = T om^(n2 k1) = T om^((4 g + j') a): four constants per lane;
//   stage 2   X[k1 + 16 k2] = sum_n2 th^(n2 k2) T'[n2][k1]:  the DFT matrix is the A operand with its rows permuted (row m <->
//             k2 = m / 4 + 4 (m % 4)), so that element j'' of lane (a, g) is row 4 g + j'' <-> k2 = g + 4 j'': X[lane + 64 j''].
// Complex products ride in the real matrices {Fr, -Fi; Fi, Fr}: K = (index, re | im) = the packed (re | im << 16) words the
// frame's f16 planes are made of; real and imaginary outputs are two accumulators.  Every operand is an f16 pair hi + lo
// (split_h), three terms per product (hi hi, hi lo, lo hi) as in the FIR -- the QAM symbols in front of the inverse
// transform are small integers, exact in f16: two terms.  The inverse transform is conj(DFT(conj X)): the constellation
// table holds the conjugates and the Tx window multiply takes the second conjugate along in a sign modifier.
// The six (four) MFMAs of a stage are two interleaved in-place chains of compiler builtins, scheduled BY THE COMPILER among
// the vector instructions around them (it knows the wait states an MFMA result needs; the matrix pipe works under the
// twiddles, splits and -- in phase B -- the noise draw of the same wave instead of stalling it: C2 13.2 -> 12.65 ms).  That
// freedom is what the other layouts must not have: an MFMA that follows a burst of MFMAs after a gap -- 7 to 1 000
// cycles, whatever fills it -- makes v_pk_*_f32 instructions with an op_sel source swizzle return wrong values in lanes
// 48..63 of the SIMD's OTHER waves (WOFDM_MMA_ALIGN above: a gap inside the FIR's chain; tools/ubench/mfma_block_train.hip,
// profiles/r03_mfma_block_train.txt: trains of blocks of three, four, six, eight MFMAs at every spacing).  Every other
// instruction form these kernels use stayed exact in 10^10 checks each beside the worst train (v_pk_* with op_sel_hi
// broadcasts, with neg, without modifiers; v_cvt_pk_f16_f32 + v_fma_mixlo/mixhi_f16).  So the guard here is the ABSENCE OF
// VICTIMS: the kernels of layouts 10 / 11 contain no v_pk_* instruction with an op_sel swizzle (no cmul, add_mi, add_pi:
// the VALU transforms were their only users), which tests/test_code_layout.py checks in the built library.  (Kernels WITH
// such instructions -- every other layout -- must not share a SIMD with these: one device runs one plan's launches at a
// time, include/wofdm.h.)
// Two things the compiler does not know about v_mfma_f32_16x16x32_f16 (new in gfx950) have to be done by hand: the
// instruction reads its A / B operands well after it has issued, so (1) a destination must not sit on top of an operand --
// it carries no early-clobber constraint -- and (2) nothing may WRITE an operand register for about a dozen cycles behind
// it: there is no interlock for that either (seen as a transform whose last set of inputs was wrong in 16 lanes, in one
// to three symbols per frame, when the register allocator reused the words right behind the last MFMA).  Both by one asm
// statement behind every chain: it takes the chain's results AND all its operands (so they stay allocated, and apart, up
// to there) and spends the 12 wait states the hand-written chain has behind it.  tests/test_code_layout.py checks both
// in the built library: no destination on an operand, no VALU write to an operand within 12 cycles of its MFMA.
#ifndef WOFDM_MMA_TAIL
#define WOFDM_MMA_TAIL "s_nop 7\n\ts_nop 3"
#endif
__device__ __forceinline__ void mma33(f4 &re, f4 &im, h8 a0, h8 b0, h8 a1, h8 b1, h8 a2, h8 b2,
                                      h8 a3, h8 b3, h8 a4, h8 b4, h8 a5, h8 b5)
{
    const f4 z = {0.f, 0.f, 0.f, 0.f};
    f4 r = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, z, 0, 0, 0);
    f4 i = __builtin_amdgcn_mfma_f32_16x16x32_f16(a3, b3, z, 0, 0, 0);
    r = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, r, 0, 0, 0);
    i = __builtin_amdgcn_mfma_f32_16x16x32_f16(a4, b4, i, 0, 0, 0);
    r = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b2, r, 0, 0, 0);
    i = __builtin_amdgcn_mfma_f32_16x16x32_f16(a5, b5, i, 0, 0, 0);
    asm volatile(WOFDM_MMA_TAIL : "+v"(r), "+v"(i) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4),
                 "v"(b4), "v"(a5), "v"(b5));
    re = r; im = i;
}
__device__ __forceinline__ void mma22(f4 &re, f4 &im, h8 a, h8 b0, h8 b1, h8 b2, h8 b3)
{
    const f4 z = {0.f, 0.f, 0.f, 0.f};
    f4 r = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b0, z, 0, 0, 0);
    f4 i = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b2, z, 0, 0, 0);
    r = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1, r, 0, 0, 0);
    i = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b3, i, 0, 0, 0);
    asm volatile(WOFDM_MMA_TAIL : "+v"(r), "+v"(i) : "v"(a), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
    re = r; im = i;
}
// The same MFMAs WITHOUT the statement behind them (round 4): for callers that place the next symbol's vector work behind the chain
// themselves and then guard it (WOFDM_TIE / WOFDM_GUARD below) -- the wait states are spent on instructions that had to be issued anyway.
__device__ __forceinline__ void mma33_issue(f4 &re, f4 &im, h8 a0, h8 b0, h8 a1, h8 b1, h8 a2, h8 b2,
                                            h8 a3, h8 b3, h8 a4, h8 b4, h8 a5, h8 b5)
{
    const f4 z = {0.f, 0.f, 0.f, 0.f};
    f4 r = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, z, 0, 0, 0);
    f4 i = __builtin_amdgcn_mfma_f32_16x16x32_f16(a3, b3, z, 0, 0, 0);
    r = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, r, 0, 0, 0);
    i = __builtin_amdgcn_mfma_f32_16x16x32_f16(a4, b4, i, 0, 0, 0);
    r = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b2, r, 0, 0, 0);
    i = __builtin_amdgcn_mfma_f32_16x16x32_f16(a5, b5, i, 0, 0, 0);
    re = r; im = i;
}
__device__ __forceinline__ void mma22_issue(f4 &re, f4 &im, h8 a, h8 b0, h8 b1, h8 b2, h8 b3)
{
    const f4 z = {0.f, 0.f, 0.f, 0.f};
    f4 r = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b0, z, 0, 0, 0);
    f4 i = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b2, z, 0, 0, 0);
    r = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1, r, 0, 0, 0);
    i = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b3, i, 0, 0, 0);
    re = r; im = i;
}
// Order by dependencies (MFMA builtins and vector arithmetic are pure: scheduling hints do not bind the IR passes; see the FIR tile in
// phase B).  WOFDM_TIE2: an empty volatile statement that a chain's two accumulators and the inputs of the vector work that is to run
// BEHIND the chain pass through.  The guard that closes such a group is written out at its place: it takes the results of that vector
// work as in-out operands and the chain's accumulators and ALL its operands as inputs, so that the operands stay allocated, and apart
// from the accumulators, up to there -- two dozen vector instructions behind the last MFMA instead of twelve idle wait states.
#define WOFDM_TIE2(r, i, a, b) asm volatile("" : "+v"(r), "+v"(i), "+v"(a), "+v"(b))
// sample x window value, as two plain multiplies the compiler cannot re-pack: where window values arrive as pairs (the
// consecutive elements of layout 12) it multiplies the second sample by the pair's HIGH half -- v_pk_mul_f32 with an
// op_sel swizzle, the one instruction form these kernels must not contain (mma33)
__device__ __forceinline__ v2f wmul(v2f x, float w)
{
    float a, b;
    asm("v_mul_f32 %0, %1, %2" : "=v"(a) : "v"(x.x), "v"(w));
    asm("v_mul_f32 %0, %1, %2" : "=v"(b) : "v"(x.y), "v"(w));
    return mk(a, b);
}
__device__ __forceinline__ v2f wfma(v2f x, float w, v2f acc)
{
    float a, b;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(x.x), "v"(w), "v"(acc.x));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(b) : "v"(x.y), "v"(w), "v"(acc.y));
    return mk(a, b);
}
// the wave's constants of the two stages: operand rows of th^(..) for real / imaginary outputs, hi / lo halves, and the
// inter-stage twiddles (table: wofdm_abi.hip; rows 0..3 and 8, 9 sit in LDS, 4..7 come from L2)
struct mdft_consts { h8 brh, brl, bih, bil, arh, arl, aih, ail; f4 twr, twi; };
// (t_re + i t_im) *= (twr + i twi), element-wise on the lane's four values
__device__ __forceinline__ void mdft_twiddle(f4 &tr, f4 &ti, f4 wr, f4 wi)
{
    const f4 r = tr * wr - ti * wi, i = tr * wi + ti * wr;
    tr = r; ti = i;
}
__device__ __forceinline__ void mdft_split4(f4 re, f4 im, h8 &hi, h8 &lo)
{
    uint32_t h[4], l[4];
    split_h(mk(re.x, im.x), h[0], l[0]);
    split_h(mk(re.y, im.y), h[1], l[1]);
    split_h(mk(re.z, im.z), h[2], l[2]);
    split_h(mk(re.w, im.w), h[3], l[3]);
    hi = __builtin_bit_cast(h8, (u4){h[0], h[1], h[2], h[3]});
    lo = __builtin_bit_cast(h8, (u4){l[0], l[1], l[2], l[3]});
    mma_operand_fence(hi, lo);
}
// N = 256 NC (layout 12): n = N/16 a + NC b + c, k = ka + 16 kb + 256 kc.  Per c the two stages above on the 256 elements
// x[N/16 a + NC b + c] (stage 1 over a, twiddle om256^(b ka), stage 2 over b: the same operand and twiddle rows as N = 256),
// then the twiddle om_N^(c (ka + 16 kb)) = om_N^(c (lane + 64 j)) and a radix-NC stage over c -- in registers, real and
// imaginary parts apart: no swizzle.  Set c, element j of lane (b, g) in: element N/16 (g + 4 j) + NC b + c; set kc,
// element j out: element lane + 64 j + 256 kc.
template <int NC, bool EXACT>
__device__ __forceinline__ void mdft_big(const h8 (&xh)[NC], const h8 (&xl)[NC], const mdft_consts &c, const f4 (&t2r)[NC],
                                         const f4 (&t2i)[NC], f4 (&yr)[NC], f4 (&yi)[NC])
{
    // (round 4: the sets as a pipeline like the four symbols of layouts 10 / 11 -- second-stage chain of set s, behind it the twiddle
    // and split of set s + 1, no wait states -- was built and measured: N = 512 -1.5 %, N = 1024 -6 %.  These kernels sit at their
    // 128-register limit, and the operands of two sets alive at once cost more than the tails; profiles/r04_other_configs.txt)
    f4 tr[NC], ti[NC], dr[NC], di[NC];
#pragma unroll
    for (int s = 0; s < NC; ++s) {
        if constexpr (EXACT) mma22(tr[s], ti[s], xh[s], c.brl, c.brh, c.bil, c.bih);
        else mma33(tr[s], ti[s], xl[s], c.brh, xh[s], c.brl, xh[s], c.brh, xl[s], c.bih, xh[s], c.bil, xh[s], c.bih);
    }
#pragma unroll
    for (int s = 0; s < NC; ++s) {
        mdft_twiddle(tr[s], ti[s], c.twr, c.twi);
        h8 th, tl;
        mdft_split4(tr[s], ti[s], th, tl);
        mma33(dr[s], di[s], c.arh, tl, c.arl, th, c.arh, th, c.aih, tl, c.ail, th, c.aih, th);
        if (s > 0) mdft_twiddle(dr[s], di[s], t2r[s], t2i[s]);
    }
    if constexpr (NC == 1) {
        yr[0] = dr[0]; yi[0] = di[0];
    } else if constexpr (NC == 2) {
        yr[0] = dr[0] + dr[1]; yi[0] = di[0] + di[1];
        yr[1] = dr[0] - dr[1]; yi[1] = di[0] - di[1];
    } else {
        static_assert(NC == 4, "N = 512 or 1024");
        const f4 a0r = dr[0] + dr[2], a0i = di[0] + di[2], a1r = dr[0] - dr[2], a1i = di[0] - di[2];
        const f4 a2r = dr[1] + dr[3], a2i = di[1] + di[3], a3r = dr[1] - dr[3], a3i = di[1] - di[3];
        yr[0] = a0r + a2r; yi[0] = a0i + a2i;
        yr[2] = a0r - a2r; yi[2] = a0i - a2i;
        yr[1] = a1r + a3i; yi[1] = a1i - a3r;                  // a1 - i a3
        yr[3] = a1r - a3i; yi[3] = a1i + a3r;                  // a1 + i a3
    }
}

// The transposed flow of mdft_big<4> (decimation in frequency): the radix-4 stage over the sets FIRST -- input: set kc, element j =
// element lane + 64 j + 256 kc, the OUTPUT order of mdft_big<4> --, then the twiddle om_1024^(c (lane + 64 j)) and per c the two matrix
// stages: z[c + 4 n'] = sum_k om_256^(k n') [ om_1024^(k c) sum_kc (-i)^(kc c) Z[k + 256 kc] ].  Output: set c, element j = element
// 4 (lane + 64 j) + c, the INPUT order of mdft_big<4>: a transform and its inverse back to back need no exchange (layout 15).
__device__ __forceinline__ void mdft_big_dif4(const f4 (&zr)[4], const f4 (&zi)[4], const mdft_consts &c, const f4 (&t2r)[4],
                                              const f4 (&t2i)[4], f4 (&yr)[4], f4 (&yi)[4])
{
    f4 ur[4], ui[4], tr[4], ti[4];
    {
        const f4 a0r = zr[0] + zr[2], a0i = zi[0] + zi[2], a1r = zr[0] - zr[2], a1i = zi[0] - zi[2];
        const f4 a2r = zr[1] + zr[3], a2i = zi[1] + zi[3], a3r = zr[1] - zr[3], a3i = zi[1] - zi[3];
        ur[0] = a0r + a2r; ui[0] = a0i + a2i;
        ur[2] = a0r - a2r; ui[2] = a0i - a2i;
        ur[1] = a1r + a3i; ui[1] = a1i - a3r;                  // a1 - i a3
        ur[3] = a1r - a3i; ui[3] = a1i + a3r;                  // a1 + i a3
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (s > 0) mdft_twiddle(ur[s], ui[s], t2r[s], t2i[s]);
        h8 uh, ul;
        mdft_split4(ur[s], ui[s], uh, ul);
        mma33(tr[s], ti[s], ul, c.brh, uh, c.brl, uh, c.brh, ul, c.bih, uh, c.bil, uh, c.bih);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        mdft_twiddle(tr[s], ti[s], c.twr, c.twi);
        h8 th, tl;
        mdft_split4(tr[s], ti[s], th, tl);
        mma33(yr[s], yi[s], c.arh, tl, c.arl, th, c.arh, th, c.aih, tl, c.ail, th, c.aih, th);
    }
}

// N = 16 C, C = 4 or 8 (layouts 13, 14): n = C a + c, k = ka + 16 kc.  ONE matrix stage over a (the same operand rows as stage 1 of
// N = 256) with 16 (symbol, c) pairs as the rows of a set: lane (ka, g') receives, as the four elements of its accumulator, c = 4 p + j'
// of symbol g' of the set's group (p = part: one set per group at C = 4, two at C = 8) -- the twiddle om_N^(c ka) and the radix-C
// stage over c then run on the elements of ONE lane's registers, real and imaginary parts apart.
// radix-4 over the four elements of (r, i): out element kc = sum_c in[c] (-i)^(c kc)
__device__ __forceinline__ void radix4_elems(f4 &r, f4 &i)
{
    const float a0r = r.x + r.z, a0i = i.x + i.z, a1r = r.x - r.z, a1i = i.x - i.z;
    const float a2r = r.y + r.w, a2i = i.y + i.w, a3r = r.y - r.w, a3i = i.y - i.w;
    r = (f4){a0r + a2r, a1r + a3i, a0r - a2r, a1r - a3i};
    i = (f4){a0i + a2i, a1i - a3r, a0i - a2i, a1i + a3r};
}
// radix-8 over (t0 = elements c = 0..3, t1 = c = 4..7): even outputs kc = 2 e in (r0, i0), odd outputs kc = 2 e + 1 in (r1, i1)
__device__ __forceinline__ void radix8_elems(f4 &r0, f4 &i0, f4 &r1, f4 &i1)
{
    constexpr float h = 0.70710678118654752f;
    const f4 er = r0 + r1, ei = i0 + i1, orr = r0 - r1, oi = i0 - i1;
    // odd branch: times om_8^j = {1, (1 - i) h, -i, (-1 - i) h}
    const f4 wr = {1.f, h, 0.f, -h}, wi = {0.f, -h, -1.f, -h};
    f4 pr = orr * wr - oi * wi, pi = orr * wi + oi * wr;
    r0 = er; i0 = ei;
    radix4_elems(r0, i0);
    radix4_elems(pr, pi);
    r1 = pr; i1 = pi;
}

// registers -> (LDS stages) -> registers, natural order in and out, SPW symbols at once
template <int N, int DIR, int SPW>
__device__ __forceinline__ void fft_wave(v2f (&v)[SPW][geo<N>::BPL][4], v2f *fb, int sb, const v2f *tw,
                                         int lane)
{
    if constexpr ((N == 512 || N == 1024) && WOFDM_FFT_BIG_RADIX) {
        static_assert(SPW == 1, "one symbol per wave at N >= 512");
        fft_big<N, DIR>(v, fb, tw, lane);
        return;
    }
    fft_first<N, DIR, SPW>(v, fb, sb, lane);
    if constexpr (N == 64) {
        fft_mid4<N, 4, DIR, SPW>(fb, sb, tw, lane);
    } else if constexpr (N == 128) {
        fft_mid2<N, 4, DIR, SPW>(fb, sb, tw, lane);
        fft_mid4<N, 8, DIR, SPW>(fb, sb, tw, lane);
    } else if constexpr (N == 256) {
        fft_mid4<N, 4, DIR, SPW>(fb, sb, tw, lane);
        fft_mid4<N, 16, DIR, SPW>(fb, sb, tw, lane);
    } else if constexpr (N == 512) {
        fft_mid4<N, 4, DIR, SPW>(fb, sb, tw, lane);
        fft_mid2<N, 16, DIR, SPW>(fb, sb, tw, lane);
        fft_mid4<N, 32, DIR, SPW>(fb, sb, tw, lane);
    } else {
        static_assert(N == 1024, "unsupported DFT length");
        fft_mid4<N, 4, DIR, SPW>(fb, sb, tw, lane);
        fft_mid4<N, 16, DIR, SPW>(fb, sb, tw, lane);
        fft_mid4<N, 64, DIR, SPW>(fb, sb, tw, lane);
    }
    fft_last<N, DIR, SPW>(v, fb, sb, tw, lane);
}

// Fill the per-stage twiddle tables (once per workgroup).
template <int N> __device__ __forceinline__ void fill_twiddles(v2f *tw, int tid, int nthreads)
{
    if constexpr ((N == 512 || N == 1024) && WOFDM_FFT_BIG_RADIX) {
        // tables of fft_big
        constexpr int R = N / 64, T2 = N == 1024 ? 48 : 56;
        for (int i = tid; i < T2 + (R - 1) * 64; i += nthreads) {
            float num, den;
            if (i < T2) {
                const int per = N == 1024 ? 3 : 7;
                num = (float)((i / per) * (1 + i % per)); den = 64.0f;                   // r k / 64
            } else {
                const int e = i - T2;
                num = (float)((e / (R - 1)) * (1 + e % (R - 1))); den = (float)N;        // t j / N
            }
            float sv, cv;
            sincospif(-2.0f * num / den, &sv, &cv);
            tw[i] = mk(cv, sv);
        }
        return;
    }
    int off = 0, ns = 4;
    while (ns <= N / 4) {
        const bool r2 = (N == 128 && ns == 4) || (N == 512 && ns == 16);
        const int cnt = r2 ? ns : 3 * ns;
        for (int i = tid; i < cnt; i += nthreads) {
            const int k = r2 ? i : i / 3, r = r2 ? 1 : 1 + i % 3;
            float sv, cv;
            sincospif(-2.0f * (float)(r * k) / (float)((r2 ? 2 : 4) * ns), &sv, &cv);
            tw[off + i] = mk(cv, sv);
        }
        off += cnt;
        ns *= r2 ? 2 : 4;
    }
}

// CNT consecutive FIR outputs starting at window base w (w[i] = tx[j0 - (LT-1) + i]).
// The taps are wave-uniform and read through a noalias kernel argument, so they arrive by
// scalar loads as SGPR pairs and feed v_pk_fma_f32 directly: 2 instructions per complex MAC.
template <int CNT>
__device__ __forceinline__ void fir_chunk(const v2f *w, const v2f *__restrict__ taps, v2f *acc)
{
    constexpr int LT = WOFDM_LT;
    v2f win[CNT + LT - 1];
#pragma unroll
    for (int i = 0; i < CNT + LT - 1; ++i) win[i] = w[i];
#pragma unroll
    for (int r = 0; r < CNT; ++r) acc[r] = mk(0.f, 0.f);
#pragma unroll
    for (int l = 0; l < LT; ++l) {
        const v2f t = taps[l];
        const v2f tn = mk(-t.y, t.y);
#pragma unroll
        for (int r = 0; r < CNT; ++r) {
            const v2f x = win[r + LT - 1 - l];
            acc[r] = __builtin_elementwise_fma(t.xx, x, acc[r]);
            acc[r] = __builtin_elementwise_fma(tn, x.yx, acc[r]);
        }
    }
}

// one complex unit normal from two Philox words (philox.h).  The angle uses the top 23 bits of b as the
// mantissa of a float in [1, 2): v_sin / v_cos take revolutions, so the integer part drops out and no
// convert + scale is needed.  UNIT = false leaves out the factor sqrt(2 ln 2) of the radius
// (-2 ln u1 = 2 ln2 * -log2 u1): where the noise only meets its own measured power (g = sqrt(Ps nlin / Pn),
// r = c + g n) a common factor cancels exactly; WOFDM_NOISE_UNSCALE restores it for stage dumps.
#define WOFDM_NOISE_UNSCALE 1.1774100225154747f
template <bool UNIT = true> __device__ __forceinline__ v2f box_muller(uint32_t a, uint32_t b)
{
    const float u1 = fmaf((float)a, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
    const float u2 = __builtin_bit_cast(float, __builtin_amdgcn_alignbit(0x7Fu, b, 9));   // 1 + (b >> 9) 2^-23
    const float l2 = __builtin_amdgcn_logf(u1);
    const float rad = UNIT ? __builtin_amdgcn_sqrtf(-1.3862943611198906f * l2) : __builtin_amdgcn_sqrtf(-l2);
    return mk(__builtin_amdgcn_cosf(u2), __builtin_amdgcn_sinf(u2)) * rad;
}

// One block of stream `stream` of (seed, cell, frame), Philox4x32-10 as in philox.h with the
// three-input XORs fused (v_bitop3_b32) and the round keys advanced by scalar adds per call
// (opaque key: otherwise the 20 round keys sit in SGPRs for the whole frame loop).
template <bool LAUNDER = true>
__device__ __forceinline__ philox_out stream_block(uint32_t block, uint32_t f_lo, uint32_t f_hi,
                                                   uint32_t stream_cell, uint32_t k0, uint32_t k1)
{
    if constexpr (LAUNDER) asm volatile("" : "+s"(k0), "+s"(k1));
    uint32_t c0 = block, c1 = f_lo, c2 = f_hi, c3 = stream_cell;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    philox_out o;
    o.w[0] = c0; o.w[1] = c1; o.w[2] = c2; o.w[3] = c3;
    return o;
}

// Sum over the 64 lanes without LDS round trips: four DPP steps give every lane its 16-lane
// row sum, the four row sums are then read as scalars.  (The reduction sits on the critical path
// in front of barrier 2; ds_bpermute shuffles cost six dependent LDS latencies there.)
__device__ __forceinline__ float wave_sum(float x)
{
    auto dpp = [](float v, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
            0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    x += dpp(x, std::integral_constant<int, 0xB1>{});     // quad_perm [1,0,3,2]
    x += dpp(x, std::integral_constant<int, 0x4E>{});     // quad_perm [2,3,0,1]
    x += dpp(x, std::integral_constant<int, 0x141>{});    // row_half_mirror
    x += dpp(x, std::integral_constant<int, 0x140>{});    // row_mirror
    const int xi = __builtin_bit_cast(int, x);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 48));
    return (r0 + r1) + (r2 + r3);
}
// every lane its 16-lane row's sum (four DPP steps)
__device__ __forceinline__ float row_sum(float x)
{
    auto dpp = [](float v, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
            0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    x += dpp(x, std::integral_constant<int, 0xB1>{});
    x += dpp(x, std::integral_constant<int, 0x4E>{});
    x += dpp(x, std::integral_constant<int, 0x141>{});
    x += dpp(x, std::integral_constant<int, 0x140>{});
    return x;
}
// (same scheme for the error counters: no lane-index vectors, which the compiler would hoist out of
// the frame loop and spill)
__device__ __forceinline__ unsigned wave_sum_u(unsigned x)
{
    auto dpp = [](unsigned v, auto ctrl) {
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, decltype(ctrl)::value, 0xF, 0xF, true);
    };
    x += dpp(x, std::integral_constant<int, 0xB1>{});
    x += dpp(x, std::integral_constant<int, 0x4E>{});
    x += dpp(x, std::integral_constant<int, 0x141>{});
    x += dpp(x, std::integral_constant<int, 0x140>{});
    return (unsigned)__builtin_amdgcn_readlane((int)x, 0) + (unsigned)__builtin_amdgcn_readlane((int)x, 16)
           + (unsigned)__builtin_amdgcn_readlane((int)x, 32) + (unsigned)__builtin_amdgcn_readlane((int)x, 48);
}

// Spin until the LDS word reaches `target` (monotonic iteration counter written by another wave
// of the workgroup).  Bounded: a wave never hangs the GPU on a protocol error, it falls through.
// A wave that gives up leaves a mark in LDS (gave_up), turned into bit 0 of the plan's status word
// when the workgroup retires -- the only trace of this in the frame loop is the loop's own counter.
// The flag words are addressed as LDS (ds_read / ds_write), not through the generic address space (flat_*: the vector-memory
// path to the LDS and back).
typedef volatile __attribute__((address_space(3))) int lds_vint;
__device__ __forceinline__ void wait_flag(const lds_vint *flag, int target, lds_vint *gave_up)
{
    int budget = 1 << 22;
    while (__builtin_amdgcn_readfirstlane(*flag) < target) {
        if (__builtin_expect(--budget == 0, 0)) {
            if (WOFDM_CHECKED_SYNC) *gave_up = 1;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void post_flag(lds_vint *flag, int value, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) *flag = value;
}

__device__ __forceinline__ v2f ldg2(const float2 *p) { const float2 t = *p; return mk(t.x, t.y); }

// Tx mask stage (VAR 2): a lane owns NO consecutive outputs of the 2P-1 <= 2(N+128)-1 samples of
// the filtered symbol; the inputs are walked MB at a time.  rg[] is the periodic extension of the
// mask's impulse response (complex: the reference's mask is not even around bin 0 when P is odd),
// RG_OFF entries of history in front.
template <int N> struct mask_geo {
    static constexpr int LMAX = 2 * (N + wofdm_lds<N>::CPCS_MAX) - 1;
    static constexpr int NO = (LMAX + 63) / 64;
    static constexpr int MB = 4;
    static constexpr int RG_OFF = N + wofdm_lds<N>::CPCS_MAX + 2 * MB;
    static constexpr int RG_LEN = RG_OFF + 64 * NO + MB;
};

// Tx mask, FFT form (VAR 3, n_fft <= 256): MF-point transforms, SLOTS scratch rows
struct maskfft_geo {
    static constexpr int MF = WOFDM_TXFFT_LEN, SLOTS = WOFDM_TXFFT_SLOTS;
};

// FIR outputs per lane: one wave covers SPW symbols = SPW*B consecutive samples.  For SPW = 2
// the count is even and every lane starts on an even sample, so its unit noise is exactly
// RB/2 Philox blocks (2.5 per symbol instead of 3).
// (LAY = layout id = symbols per wave, except 5 = four symbols with 20 instead of 18 outputs per
// lane, for strides of up to 320 samples)
template <int N, int LAY> struct fir_geo {
    static constexpr int RB = (LAY == 8 || LAY == 9 || LAY == 12 || LAY == 15) ? 2 * wofdm_fir8_tiles(N) : LAY >= 6 ? (LAY == 14 ? 22 : ((LAY == 7 || LAY == 11 || LAY == 13 || LAY == 16) ? 20 : 18))
                              : (LAY == 1 ? N / 64 + 1 : (LAY == 5 ? 20 : LAY * (N / 64) + 2));
    static constexpr bool EVEN = (LAY != 1) && (RB % 2 == 0);
    static constexpr int NBK = EVEN ? RB / 2 : RB / 2 + 1;      // Philox blocks per lane
    static constexpr int CH = RB <= 6 ? RB : (RB % 5 == 0 ? 5 : 6);
};

template <int RB, int CH>
__device__ __forceinline__ void fir_lane(const v2f *w, const v2f *__restrict__ taps, v2f (&acc)[RB])
{
    constexpr int FULL = RB / CH, REM = RB % CH;
#pragma unroll
    for (int c = 0; c < FULL; ++c) fir_chunk<CH>(w + c * CH, taps, &acc[c * CH]);
    if constexpr (REM != 0) fir_chunk<REM>(w + FULL * CH, taps, &acc[FULL * CH]);
}

// VAR: 0 = every subcarrier loaded (main_BER_calculation.m); 1 = subcarrier allocation: only the
// bins flagged in g_amask carry data, the others transmit zero and are not counted
// (main_channel_mask.m:387-390, 367-371); 2 = allocation + the per-symbol spectral Tx mask
// dft_rc_filt (main_channel_mask.m:398-417), g_tmask = its length-(2P-1) circular impulse response
template <int N, int K, int LAY, bool INJECT, bool DUMP, int VAR>
__global__ void __launch_bounds__((LAY == 8 || LAY == 9 || LAY == 12 || LAY == 15) ? 1024 : ((LAY == 13 || LAY == 14) ? N : (LAY >= 4 ? 256 : 1024 / LAY)),
                                   (LAY >= 4 && LAY != 8 && LAY != 9 && LAY != 12 && LAY != 15) ? 3 : WOFDM_MIN_WAVES_PER_SIMD)
wofdm_frames_kernel(const wofdm_kparams p, const float *__restrict__ g_wtx,
                    const float *__restrict__ g_wrx, const float2 *__restrict__ g_h_,
                    const float *__restrict__ g_nlin, const int *__restrict__ gm,
                    const uint32_t *__restrict__ g_amask, const float2 *__restrict__ g_tmask,
                    const uint4 *__restrict__ g_fira)
{
    // layouts 6, 7 (quarter-wave, four symbols per wave) and 8 (one symbol per wave, N >= 512): the
    // 21-tap FIR on the matrix pipe as a block-Toeplitz product, NT tiles of 128 samples per wave (phase B)
    // layouts 10, 11: 6, 7 with both 256-point transforms on the matrix pipe too (mdft_fwd): four symbols per wave, lane l
    // holds elements l + 64 j of each
    constexpr bool MDFT = LAY == 10 || LAY == 11;
    // layout 12: 8 with both transforms on the matrix pipe (N = 512, 1024: 16 . 16 . NC, NC = N / 256, the last stage in
    // registers; mdft_big): lane (b = lane % 16, g = lane / 16) holds the INPUT elements N/16 (g + 4 j) + NC b + c and the
    // OUTPUT elements lane + 64 j + 256 c, j < 4, c < NC
    constexpr bool MD8 = LAY == 12;
    // layout 15: the Tx mask's fast-convolution form at N = 256 with EVERYTHING on the matrix pipe: layout 9's frame handling (rows as fp32
    // through the mask stage, f16 planes from phase B on), the symbol's own transforms as in layout 12 with NC = 1, and the mask's
    // two 1024-point transforms as mdft_big<4> with one exchange between them
    constexpr bool MDM = LAY == 15;
    constexpr bool MDX = MD8 || MDM;                       // one symbol per wave, transforms by mdft_big
    // layouts 13, 14: N = 64 / 128, sixteen / eight symbols per wave, ONE matrix stage + the radix-N/16 stage in registers (radix4_elems)
    // layout 16: 13 with a RUN-TIME number of symbols per wave, SPWR <= SPW (even; gm[WOFDM_G_SPWR], wofdm_small_spwr), and a partly
    // filled last wave: wave w takes the symbols w SPWR ..., its rows are SPWR B words per plane, and the symbol slots it does not
    // fill (slot u = 4 group + lane group >= nreal) compute along but store nothing -- what lies behind a wave's samples is the next
    // wave's, or the zeros behind the frame.  For the geometries layouts 13 / 14 do not take: S not a multiple of 16 / 8, strides
    // beyond their tiles (N = 64 at CP 32: two waves of eight symbols); they ran layout 2 before (round 4).
    constexpr bool PART = LAY == 16;
    constexpr bool MDS = LAY == 13 || LAY == 14 || PART;
    constexpr int SC = N / 16, SCS = MDS ? SC / 4 : 1, SGR = 4 / SCS;      // elements per lane and symbol, sets per group, groups
    constexpr bool MPIPE = MDFT || MD8 || MDS || MDM;                    // kernels without op_sel-swizzled packed arithmetic (see mma33)
    constexpr int NC = MD8 ? N / 256 : 1;                  // (layout 15: one set)
    // layout 9: the Tx-mask variants with the FIR on the matrix pipe -- layout 8's frame format; the windowed symbols and the
    // mask stage live in the rows as fp32, phase B converts each row to the two f16 planes in place
    constexpr bool FIR8M = LAY == 9 || LAY == 15;
    constexpr bool FIRQ = LAY == 6 || LAY == 7 || MDFT || MDS, FIR8 = LAY == 8 || MD8 || FIR8M, FIRM = FIRQ || FIR8;
    constexpr int SPW = MDS ? 1024 / N : (FIR8 ? 1 : (LAY >= 5 ? 4 : LAY));     // symbols per wave
    constexpr int NT = FIR8 ? wofdm_fir8_tiles(N) : (LAY == 14 ? 11 : ((LAY == 7 || LAY == 11 || LAY == 13 || PART) ? 10 : 9)), PRE = WOFDM_FIRM_PRE;
    constexpr int VT = WOFDM_FIR8_VT;
    static_assert(!FIR8 || FIR8M || (N >= 512 && VAR <= 1), "layout 8 is built for N >= 512 without Tx mask");
    static_assert(!FIR8M || VAR >= 2, "layout 9 is the Tx-mask variants' matrix-pipe FIR layout");
    static_assert(!MDM || (N == 256 && VAR == 3), "layout 15 is the fast-convolution Tx mask at N = 256");
    constexpr bool ALLOC = VAR >= 1, TXMASK = VAR == 2, TXFFT = VAR == 3;
    // flags instead of barriers 1 and 3 (not in the instrumented and masked variants, whose extra
    // stages have their own workgroup barriers)
    constexpr bool RELAX = WOFDM_RELAXED_SYNC && !DUMP && VAR < 2;
    // The Tx-mask variants (one symbol per wave) have one more hand-over: symbol s adds the second half of its
    // masked output onto the row of symbol s + 1 (dft_rc_filt's overlap, m:411-415).  Flags as well: [32 + s] = "my own
    // row holds my masked symbol", then [s] = "... and my spill is on my successor's row".  Phase B of wave w needs its
    // own row (complete once w - 1 has spilled: flag w - 1) and the end of its predecessor's (complete once w - 2
    // has: flag w - 2).
    constexpr bool RELAXM = WOFDM_RELAXED_SYNC && !DUMP && VAR >= 2;
    constexpr bool RELAXF = RELAX || RELAXM;         // flags instead of barriers 1 and 3
    static_assert(!(TXMASK || TXFFT) || SPW == 1, "the Tx mask stage runs one symbol per wave");
    constexpr int LT = WOFDM_LT;
    constexpr int BPL = geo<N>::BPL, NQ = geo<N>::NQ;
    constexpr bool FULL = geo<N>::FULL;
    // SPW = 4: quarter-wave layout (fft_qw): 16 lanes per symbol, 16 subcarriers per lane, 4-wave
    // workgroups, three of them per CU at up to 168 VGPRs
    constexpr bool QW = SPW == 4 && !MDFT && !MDS;
    static_assert(!(QW || MDFT) || (N == 256 && VAR <= 1), "the four-symbol layouts are built for N = 256 without Tx mask");
    static_assert(!MDS || ((N == 64 || N == 128) && VAR <= 1), "layouts 13 / 14 / 16 are built for N = 64, 128, without Tx mask");
    constexpr int VS = QW ? 1 : (MDS ? 1 : SPW), VB = QW ? 4 : BPL;      // register arrays [VS][VB][4]
    constexpr int RB = fir_geo<N, LAY>::RB, NBK = fir_geo<N, LAY>::NBK;
    constexpr bool EVEN = fir_geo<N, LAY>::EVEN;
    // Large DFTs would keep RB = N/64+1 noise samples AND FIR outputs per lane alive across
    // barrier 2 (68 VGPRs at N = 1024) and spill.  There the unit noise is parked in a per-workgroup
    // HBM scratch row ([wave][r][lane]: 512-byte coalesced rows, written and read back by the
    // same lane, L2-resident) instead of registers; with injected noise it is simply re-read.
    constexpr bool RENOISE = N >= WOFDM_NOISE_SCRATCH_MIN_N;
    // Kernels at their VGPR limit make the lane id opaque again at every phase: otherwise per-lane index
    // vectors of one phase's FFT are kept for the next phase's -- in scratch (N = 1024: 20 spilled VGPRs
    // -> 0).  Elsewhere the recomputation costs more than it saves (N = 512: -2.5 %; C2 would spill).
    constexpr bool RELAUNDER = N >= 1024 || (N >= 512 && VAR >= 2);
    // fewest samples a wave holds (layout 8: B >= N, wofdm_spw): tiles below that are full in every geometry and
    // carry no validity tests.  (Not used in the quarter-wave layouts: C2 runs the all-full instantiation anyway,
    // and the other one got 1 % slower with it.)
    constexpr int LW_MIN = FIR8 ? N : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane0 = tid & 63;
    // the wave index is wave-uniform: keep it (and everything derived from it) in SGPRs
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // first of this wave's symbols (layout 16: SPWR of them, a run-time count; SPWR2 = half of it, the rows per f16 plane)
    const int SPWR = PART ? __builtin_amdgcn_readfirstlane(gm[WOFDM_G_SPWR]) : SPW, SPWR2 = SPWR >> 1;
    (void)SPWR2;
    const int s0 = wv * SPWR;
    int lane = lane0;
    // The structure lengths live in a small device array (gm[WOFDM_G_*]) that every phase
    // re-reads by scalar loads through a laundered pointer (GEO_PHASE): held in registers for
    // the whole frame loop they overflow the SGPR file and come back as v_readlane traffic.
#define GEO_PHASE()                                                                            \
    int goff_ = 0;                    /* opaque OFFSET: the pointer keeps its noalias provenance, */ \
    asm volatile("" : "+s"(goff_));   /* so the reads stay scalar loads (a laundered pointer      */ \
    const int *__restrict__ gq = gm + goff_   /* turns them into flat vector loads)               */

    // LDS carve with compile-time offsets (wofdm_lds<N>): only the frame buffer, last, has a
    // run-time length.  Fewer live scalars = fewer SGPR spills in the frame loop.
    using L = wofdm_lds<N, (LAY == 13 || LAY == 14 || LAY == 16)>;
    v2f *tw = reinterpret_cast<v2f *>(smem + L::off_tw);
    v2f *G = reinterpret_cast<v2f *>(smem + L::off_g);
    float *sums = reinterpret_cast<float *>(smem + L::off_sums);
    lds_vint *flags = (lds_vint *)(smem + L::off_flags);
    float *wtx = reinterpret_cast<float *>(smem + L::off_wtx);
    float *wrx = reinterpret_cast<float *>(smem + L::off_wrx);
    // (fall tails: S rows of tail_tx samples right behind the frame buffer, at a run-time offset that
    // every phase re-reads: TAILS())
    v2f *qlut = reinterpret_cast<v2f *>(smem + L::off_lut);
    v2f *fbuf = reinterpret_cast<v2f *>(smem + L::off_fbuf);
    const v2f *g_h = reinterpret_cast<const v2f *>(g_h_);
    // Matrix-pipe layouts: the frame buffer's bytes are two planes of packed-f16 words, Hp[i] / Lp[i] =
    // hi / lo halves (re | im << 16) of sample i - PRE, Lp = Hp + plen (plen = gm[WOFDM_G_FBUF]); the fall
    // tails likewise (tH, tL).  Wave w's four symbol rows of B words in EACH plane double as its
    // private scratch: 2 x 4B words = four rows of B complex floats (row()).
    // Layout 8 interleaves the planes row by row instead: 8 zero words, then per symbol s B words of
    // plane H and B words of plane L (word 8 + 2 B s: the wave's private row of B complex floats is
    // one piece), then a short virtual row S (VT + VT words) for the last symbol's fall tail.
    uint32_t *Hp = reinterpret_cast<uint32_t *>(smem + L::off_fbuf);
#define TAILS()                                                                                \
    const int tail_off = gq[WOFDM_G_FBUF];            /* float2 units from fbuf */                \
    v2f *tailb = fbuf + tail_off;                                                                 \
    uint32_t *tH = reinterpret_cast<uint32_t *>(tailb), *tL = tH + gq[WOFDM_G_S] * gq[WOFDM_G_BETA]; \
    (void)tH; (void)tL; (void)tail_off

    DELAY_AT(11);
    for (int i = tid; i < gm[WOFDM_G_FBUF]; i += blockDim.x) fbuf[i] = mk(0.f, 0.f);
    if (tid < 64) flags[tid] = 0;
    if (tid < 64) sums[tid] = 0.f;                 // (waves a short frame does not have leave their partial sums at zero)
    int iter = 0;                                  // frames this workgroup has started
    if constexpr (MDS) {
        // (N = 64, 128: up to a dozen workgroups per CU -- the operand rows come from L2 in every phase)
    } else if constexpr (MPIPE) {
        // rows 0..3 (stage-1 operands) and 8, 9 (twiddles) of the operand table, [6][64] 16-byte rows; rows 4..7 (stage 2,
        // which a transform needs last) come from L2, requested at the start of the phase (all ten in LDS would cost the
        // structures with the longest strides their third workgroup per CU)
        u4 *dl = reinterpret_cast<u4 *>(smem + L::off_tw);
        const u4 *dg = reinterpret_cast<const u4 *>(p.dftc);
        for (int i = tid; i < 6 * 64; i += blockDim.x) dl[i] = dg[i < 256 ? i : i + 256];
    } else if constexpr (QW) fill_twiddles_qw(tw, tid, (int)blockDim.x);
    else fill_twiddles<N>(tw, tid, (int)blockDim.x);
    // constellation table: qammod(label) (Gray, unit average power; m:248-249)
    if (tid < (1 << K)) {
        constexpr int hb = K >> 1, mm = (1 << hb) - 1;
        constexpr float qs = K == 2 ? 0.70710678118654752f : (K == 4 ? 0.31622776601683794f
                                                                     : 0.15430334996209191f);
        const uint32_t gi = (uint32_t)tid >> hb, gq = (uint32_t)tid & (uint32_t)mm;
        const int li = (int)(gi ^ (gi >> 1) ^ (gi >> 2)), lq = (int)(gq ^ (gq >> 1) ^ (gq >> 2));
        if constexpr (MPIPE) {
            // matrix-pipe transforms: the point with real and imaginary part SWAPPED (IDFT(X) = swap(DFT(swap X)): no sign
            // anywhere) as a packed f16 word of small integers (exact); the scale qs rides in the Tx window table
            const h2 w = {(_Float16)(float)(mm - 2 * lq), (_Float16)(float)(2 * li - mm)};
            reinterpret_cast<uint32_t *>(qlut)[tid] = __builtin_bit_cast(uint32_t, w);
        } else {
            qlut[tid] = mk((float)(2 * li - mm), (float)(mm - 2 * lq)) * qs;
        }
    }
    // (constellation scale: unit average power)
    constexpr float qscale = K == 2 ? 0.70710678118654752f : (K == 4 ? 0.31622776601683794f : 0.15430334996209191f);
    const uint32_t *qlw = reinterpret_cast<const uint32_t *>(qlut);
    (void)qlw;
    // Tx mask: periodic extension of the mask's impulse response behind the frame buffer,
    // rg[t] = g[(t - RG_OFF) mod (2P-1)], so that the stage below indexes it without a modulo
    v2f *rg = fbuf + gm[WOFDM_G_FBUF] + gm[WOFDM_G_S] * gm[WOFDM_G_BETA];
    if constexpr (TXMASK) {
        const int Lm = 2 * gm[WOFDM_G_P] - 1;
        for (int t = tid; t < mask_geo<N>::RG_LEN; t += blockDim.x) {
            int kk = (t - mask_geo<N>::RG_OFF) % Lm;
            if (kk < 0) kk += Lm;
            rg[t] = ldg2(g_tmask + kk);
        }
    }
    // Tx mask, FFT form: twiddles of the MF-point transforms and MF_SLOTS scratch rows behind
    // the frame buffer
    v2f *mtw = fbuf + gm[WOFDM_G_FBUF] + gm[WOFDM_G_S] * gm[WOFDM_G_BETA];
    v2f *mscr = mtw + maskfft_geo::MF;
    if constexpr (TXFFT && !MDM) fill_twiddles<maskfft_geo::MF>(mtw, tid, (int)blockDim.x);
    // Layout 15 keeps behind the fall tails instead (wofdm_lds_bytes) per wave a row of MDM_PK samples where the mask stage parks the
    // part of its output that belongs to the NEXT symbol (the spill): that wave adds it when it turns its row into the f16 planes
    // (phase B) -- no hand-over inside phase A.  (The mask's spectrum and the twiddles of its 1024-point transforms come from L2: an
    // LDS copy gained under 1 %.)
    constexpr int MDM_PK = 344;
    v2f *mdm_park = reinterpret_cast<v2f *>(smem + ((L::off_fbuf + 8 * (gm[WOFDM_G_FBUF] + gm[WOFDM_G_S] * gm[WOFDM_G_BETA]) + 15) & ~15));
    (void)mdm_park;
    __syncthreads();
    DELAY_AT(12);

    // QAM constants (qammod/qamdemod Gray, unit average power; m:248-249, 269-270)
    constexpr int k = K, half = K >> 1, m1 = (1 << half) - 1;
    constexpr uint32_t lmask = (1u << K) - 1u;
    constexpr float qinv = K == 2 ? 1.4142135623730951f : (K == 4 ? 3.1622776601683795f
                                                                   : 6.4807406984078604f);
    constexpr int ks = K == 6 ? 8 : K;
    constexpr int bps = N * ks / 128;             // Philox blocks of data bits per symbol
    static_assert(SPW * bps <= 64, "data-bit blocks of a wave must fit one pass");

    // Work items (cell, frame), cell-major: every workgroup takes one CONTIGUOUS run of them
    // (items_q or items_q + 1 items), so that it stays on a cell for as many frames as possible --
    // with the reference's 100 frames per cell a grid-strided walk would change cell (counter
    // flush, channel taps, noise level) on every single frame.  One 64-bit divide per workgroup
    // here; the loop itself steps with scalar adds and compares only.
    const uint64_t F = p.frames_per_cell;
    const uint64_t bq = blockIdx.x;
    const uint64_t item0 = bq * p.items_q + (bq < p.items_r ? bq : p.items_r);
    uint64_t n_items = p.items_q + (bq < p.items_r ? 1u : 0u);
    const uint32_t cell_rel = (uint32_t)(item0 / F);
    uint32_t cell = __builtin_amdgcn_readfirstlane(p.first_cell + cell_rel);
    uint64_t fidx = item0 - (uint64_t)cell_rel * F;
    {
        const uint32_t flo = __builtin_amdgcn_readfirstlane((uint32_t)fidx);
        const uint32_t fhi = __builtin_amdgcn_readfirstlane((uint32_t)(fidx >> 32));
        fidx = ((uint64_t)fhi << 32) | flo;
    }
    const int n_ch = gm[WOFDM_G_NCH], n_snr = gm[WOFDM_G_NSNR];
    // cell = (pair*n_snr + sn)*n_ch + ch
    int ch = __builtin_amdgcn_readfirstlane((int)(cell % (uint32_t)n_ch));
    int sn = __builtin_amdgcn_readfirstlane((int)((cell / (uint32_t)n_ch) % (uint32_t)n_snr));
    int pair = __builtin_amdgcn_readfirstlane((int)(cell / ((uint32_t)n_ch * (uint32_t)n_snr)));
    auto next_cell = [&]() {
        ++cell;
        if (++ch == n_ch) { ch = 0; if (++sn == n_snr) { sn = 0; ++pair; } }
    };

    const uint32_t seed_lo = __builtin_amdgcn_readfirstlane(p.seed_lo);
    const uint32_t seed_hi = __builtin_amdgcn_readfirstlane(p.seed_hi);
    uint32_t cur_cell = 0xFFFFFFFFu;
    int cur_pair = -1;
    // Error counters of the cell in progress: per-lane partial sums, or -- at N >= 512, where the
    // kernels are at their VGPR limit and per-lane accumulators ended up in scratch, re-read and
    // re-written every frame -- wave totals in scalar registers (one DPP reduction per frame).
    constexpr bool SCALAR_ACC = N >= 512 || LAY == 15;
    uint32_t bit_err = 0, sym_err = 0, nfr = 0;
    float nlin = 0.f;

    auto flush = [&](uint32_t cabs) {
        const uint32_t c = cabs - p.inject_base_cell;
        const unsigned be = SCALAR_ACC ? bit_err : wave_sum_u(bit_err), se = SCALAR_ACC ? sym_err : wave_sum_u(sym_err);
        if (lane == 0) {
            atomicAdd(&p.counts[4 * (size_t)c + 0], (unsigned long long)be);
            atomicAdd(&p.counts[4 * (size_t)c + 2], (unsigned long long)se);
        }
        if (tid == 0) {
            const int S = gm[WOFDM_G_S];
            const int nact = ALLOC ? gm[WOFDM_G_NACT] : N;      // loaded subcarriers
            atomicAdd(&p.counts[4 * (size_t)c + 1], (unsigned long long)nfr * (S - 1) * nact * k);
            atomicAdd(&p.counts[4 * (size_t)c + 3], (unsigned long long)nfr * (S - 1) * nact);
        }
        bit_err = 0; sym_err = 0; nfr = 0;
    };

#ifdef WOFDM_STAMP
    unsigned long long stamp_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
#endif
    for (; n_items != 0; --n_items) {
        STAMP(7);                                   // loop control, cell changes
        WAVE_PRIO(WOFDM_PRIO_A);
        // the 32-bit per-lane and per-wave error sums are flushed every 2^14 frames at the latest (a wave
        // counts at most 6144 bit errors per frame)
        if (cell != cur_cell || nfr == (1u << 14)) {
            if (cur_cell != 0xFFFFFFFFu) flush(cur_cell);
            // (matrix-pipe transforms: the Rx window table carries the cell's power of two that centres the received
            // samples in the f16 range -- everything behind it is homogeneous in that scale -- so it is refilled per cell)
            const bool refill = pair != cur_pair || (MPIPE && cell != cur_cell);
            cur_cell = cell;
            if (refill) {
                __syncthreads();
                // 1/N of the IDFT (dftmtx(N)'/N, m:370) is folded into the Tx window copy
                const int P = gm[WOFDM_G_P], delta = gm[WOFDM_G_DELTA];
                int t0 = tid;                      // (opaque: keeps the fill loops' addresses out of
                asm volatile("" : "+v"(t0));       // the frame loop's live set)
                const float txs = MPIPE ? p.tx_scale * qscale : p.tx_scale;
                float rxs = 1.0f;
                if constexpr (MPIPE) rxs = p.rx_scale[sn * n_ch + ch];
                for (int i = t0; i < P; i += blockDim.x)
                    wtx[i] = g_wtx[(size_t)pair * P + i] * txs;
                for (int i = t0; i < N + delta; i += blockDim.x)
                    wrx[i] = g_wrx[(size_t)pair * (N + delta) + i] * rxs;
                __syncthreads();
                cur_pair = pair;
            }
            nlin = g_nlin[sn];
        }
        // Hide the lane id from loop-invariant code motion: otherwise every per-lane LDS address
        // of every stage is hoisted out of the frame loop and the kernel spills ~1500 VGPRs.
        lane = lane0;
        asm volatile("" : "+v"(lane));
        ++iter;
        float *sums_it = sums + 32 * (iter & 1);
        const uint64_t frame = p.frame_offset + fidx;
        const uint32_t f_lo = (uint32_t)frame, f_hi = (uint32_t)(frame >> 32);
        const size_t inj = ((size_t)(cell - p.inject_base_cell) * F + fidx);
        ++nfr;
        DELAY_AT(1);
#ifdef WOFDM_AUDIT
        float aud_g = 0.f, aud_ps = 0.f, aud_pn = 0.f;    // developer build: per-frame record (tools/audit_suite.py)
#endif

        v2f v[VS][VB][4];
        uint32_t lab[VS][VB];
        // matrix-pipe transforms: the spectra of the wave's four symbols, real and imaginary parts apart (element j of
        // symbol u = subcarrier lane + 64 j)
        f4 yr[4], yi[4];
        (void)yr; (void)yi;
        // layout 12: the transmitted labels in OUTPUT element order (byte j of labo[kc] = subcarrier lane + 64 j + 256 kc)
        uint32_t labo[4] = {0u, 0u, 0u, 0u};
        (void)labo;
        auto mdft_tw2 = [&](f4 (&t2r)[NC], f4 (&t2i)[NC]) {     // the twiddles in front of the radix-NC stage (L2)
            const u4 *dg = reinterpret_cast<const u4 *>(p.dftc) + lane;
#pragma unroll
            for (int c = 1; c < NC; ++c) {
                t2r[c] = __builtin_bit_cast(f4, dg[(10 + 2 * (c - 1)) * 64]);
                t2i[c] = __builtin_bit_cast(f4, dg[(11 + 2 * (c - 1)) * 64]);
            }
            t2r[0] = t2i[0] = (f4){0.f, 0.f, 0.f, 0.f};
        };
        (void)mdft_tw2;
        struct mdft_early { u4 arh, arl, aih, ail; };
        auto mdft_request = [&]() {                     // the four rows that come from L2: asked for early in the phase
            mdft_early e;
            const u4 *dg = reinterpret_cast<const u4 *>(p.dftc) + 256 + lane;
            e.arh = dg[0]; e.arl = dg[64]; e.aih = dg[128]; e.ail = dg[192];
            return e;
        };
        auto mdft_load = [&](const mdft_early &e) {
            mdft_consts c;
            const u4 *dl = reinterpret_cast<const u4 *>(smem + L::off_tw) + lane;
            c.brh = __builtin_bit_cast(h8, dl[0]); c.brl = __builtin_bit_cast(h8, dl[64]);
            c.bih = __builtin_bit_cast(h8, dl[128]); c.bil = __builtin_bit_cast(h8, dl[192]);
            c.twr = __builtin_bit_cast(f4, dl[256]); c.twi = __builtin_bit_cast(f4, dl[320]);
            c.arh = __builtin_bit_cast(h8, e.arh); c.arl = __builtin_bit_cast(h8, e.arl);
            c.aih = __builtin_bit_cast(h8, e.aih); c.ail = __builtin_bit_cast(h8, e.ail);
            return c;
        };
        (void)mdft_load; (void)mdft_request;
        // element ownership: symbol of register slot u, subcarrier of (q, r)
        const int usq = QW ? (lane >> 4) : 0, llq = lane & 15;
        auto sym_of = [&](int u) { return QW ? s0 + usq : s0 + u; };
        auto sub_of = [&](int q, int r) { return QW ? llq + 16 * (q + 4 * r) : lane + 64 * q + r * NQ; };
        auto owns = [&](int q) { return QW || FULL || lane + 64 * q < NQ; };
        v2f acc[RB], nz[RB];
        int j0 = 0, cnt = 0;
        bool is_main;
        // unit noise of the lane's RB samples j0.. (randn + 1j*randn, m:290)
        auto make_noise = [&](v2f (&out)[RB], int NLv) {
            if (INJECT) {
                const float2 *src = p.unit_noise + inj * NLv + j0;
                // two samples per 16-byte load where the lane's run starts on an even sample of an
                // even row (always so in the two- and four-symbol layouts with even lengths)
                if (EVEN && ((inj * NLv) & 1) == 0) {
                    const float4 *src4 = reinterpret_cast<const float4 *>(src);
#pragma unroll
                    for (int r = 0; r + 1 < RB; r += 2) {
                        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (r + 1 < cnt) t = src4[r >> 1];
                        else if (r < cnt) { const float2 a = src[r]; t.x = a.x; t.y = a.y; }
                        out[r] = mk(t.x, t.y);
                        out[r + 1] = mk(t.z, t.w);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < RB; ++r) out[r] = (r < cnt) ? ldg2(src + r) : mk(0.f, 0.f);
                }
            } else {
                // (lanes without outputs compute unused values: every later use is under r < cnt)
                v2f cand[2 * NBK];
                const uint32_t b0 = (uint32_t)j0 >> 1;
#pragma unroll
                for (int b = 0; b < NBK; ++b) {
                    const philox_out o = stream_block(b0 + b, f_lo, f_hi,
                                                      (WOFDM_STREAM_NOISE << 28) | cell, seed_lo, seed_hi);
                    cand[2 * b] = box_muller(o.w[0], o.w[1]);
                    cand[2 * b + 1] = box_muller(o.w[2], o.w[3]);
                }
                if constexpr (EVEN) {
#pragma unroll
                    for (int r = 0; r < RB; ++r) out[r] = cand[r];
                } else {
                    const bool odd = (j0 & 1) != 0;
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        const v2f a = cand[r], b = cand[r + 1 < 2 * NBK ? r + 1 : r];
                        out[r] = odd ? b : a;
                    }
                }
            }
        };
        // ------------------------------------------------------------ A: bits, QAM, IFFT, Tx
        {
        GEO_PHASE();
        const int S = gq[WOFDM_G_S], B = gq[WOFDM_G_B], mu = gq[WOFDM_G_MU], rho = gq[WOFDM_G_RHO];
        // one symbol per wave with the FIR on the matrix pipe (layouts 8, 12): the LDS rows are BR >= B words per plane apart, BR a
        // multiple of 4, so that every row and both of its planes start on 16 bytes whatever the stride (wofdm_row_stride)
        const int BR = (FIR8 && !FIR8M) ? ((B + 3) & ~3) : B;
        (void)BR;
        const int plen = FIRQ ? gq[WOFDM_G_FBUF] : 0;
        const int TS = gq[WOFDM_G_BETA];           // row length of the fall-tail buffer
        TAILS();
        // this wave's SPW symbol slices of the frame
        v2f *fbw = FIR8 ? reinterpret_cast<v2f *>(Hp + 8 + 2 * BR * s0) : fbuf + (LT - 1) + s0 * B;
        // private row of B complex floats of the wave's symbol slot u (scratch of the transforms,
        // later the received block)
        auto row = [&](int u) -> v2f * {
            if constexpr (PART)
                return reinterpret_cast<v2f *>(Hp + (u < SPWR2 ? 0 : plen) + PRE + s0 * B) + (u < SPWR2 ? u : u - SPWR2) * B;
            else if constexpr (FIRQ)
                return reinterpret_cast<v2f *>(Hp + (u < SPW / 2 ? 0 : plen) + PRE + s0 * B) + (u % (SPW / 2)) * B;
            else
                return fbw + u * B;
        };
        // (layout 16: the symbol slots this wave fills)
        const int nreal = PART ? min(SPWR, S - s0) : SPW;
        (void)nreal;
        mdft_early dce;
        if constexpr (MDFT || MDX) dce = mdft_request();
        if (!INJECT) {
            // Philox words of the wave's symbols, staged in the (still unused) frame slices
            if (lane < SPW * bps && (!PART || lane / bps < nreal)) {
                const int ub = lane / bps, blk = lane % bps;
                const philox_out o = stream_block((uint32_t)((s0 + ub) * bps + blk), f_lo, f_hi,
                                                  (WOFDM_STREAM_BITS << 28) | cell, seed_lo, seed_hi);
                uint32_t *bw = reinterpret_cast<uint32_t *>(row(ub));
#pragma unroll
                for (int i = 0; i < 4; ++i) bw[4 * blk + i] = o.w[i];
            }
            wave_sync();
        }
        STAMPF(8);
        uint32_t xw[4][4];                 // matrix-pipe transforms: the swapped QAM points as packed f16 words
        (void)xw;
        f4 xr[4], xi[4];                   // matrix-pipe transforms: DFT(swap X) = swap(N x[t]): xr = imaginary, xi = real parts
        (void)xr; (void)xi;
        if constexpr (MDS) {
            // Set t = SCS group + part.  IN: lane (m = lane % 16, g = lane / 16), element j = subcarrier SC (g + 4 j) + c of symbol
            // 4 group + m / 4, c = 4 part + m % 4.  OUT: lane (ka = lane % 16, g'), element e = subcarrier ka + 16 kc of symbol
            // 4 group + g', kc = e (SC = 4) or 2 e + part (SC = 8).
            const int lm = lane & 15, lg = lane >> 4;
            const u4 *dg = reinterpret_cast<const u4 *>(p.dftc) + lane;
            const h8 brh = __builtin_bit_cast(h8, dg[0]), brl = __builtin_bit_cast(h8, dg[64]);
            const h8 bih = __builtin_bit_cast(h8, dg[128]), bil = __builtin_bit_cast(h8, dg[192]);
            f4 twr[SCS], twi[SCS];
#pragma unroll
            for (int pp = 0; pp < SCS; ++pp) {
                twr[pp] = __builtin_bit_cast(f4, dg[(8 + 2 * pp) * 64]);
                twi[pp] = __builtin_bit_cast(f4, dg[(9 + 2 * pp) * 64]);
            }
            uint32_t li[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int sr = 4 * (t / SCS) + (lm >> 2), c = 4 * (t % SCS) + (lm & 3);
                // (layout 16: an unfilled slot reads slot 0's words and transmits nothing)
                const bool son = !PART || sr < nreal;
                const int srr = son ? sr : 0;
                const uint32_t *bw = reinterpret_cast<const uint32_t *>(row(srr));
                li[t] = 0;
                // byte j of the word: 0x80 when subcarrier SC (g + 4 j) + c is NOT loaded (second part of the table: this layout's
                // order, one word per set and lane); the flag rides in the label byte down to phase D
                uint32_t am = 0;
                if constexpr (ALLOC) am = g_amask[NQ + 64 * t + lane] & 0x80808080u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = SC * (lg + 4 * j) + c;
                    uint32_t Lb;
                    if (INJECT) {
                        Lb = p.labels[(inj * S + s0 + srr) * N + n] & lmask;
                    } else {
                        const uint32_t bit = (uint32_t)n * (uint32_t)ks;
                        Lb = (bw[bit >> 5] >> (bit & 31u)) & lmask;
                    }
                    li[t] |= Lb << (8 * j);
                    xw[t][j] = qlw[Lb];
                    if constexpr (ALLOC) {
                        if ((am >> (8 * j)) & 0x80u) xw[t][j] = 0u;
                    }
                    if constexpr (PART) {
                        if (!son) xw[t][j] = 0u;
                    }
                    if (DUMP && son) {
                        const hpair hw = __builtin_bit_cast(hpair, xw[t][j]);
                        if (p.dump.labels_tx) p.dump.labels_tx[(s0 + sr) * N + n] = (uint8_t)Lb;
                        if (p.dump.X) p.dump.X[(s0 + sr) * N + n] = make_float2((float)hw.y * qscale, (float)hw.x * qscale);
                    }
                }
                if constexpr (ALLOC) li[t] |= am;
            }
            // the labels once through the wave's (still unused) rows into OUTPUT order for phase D: one byte per subcarrier
            uint8_t *lbytes = reinterpret_cast<uint8_t *>(row(0));
            wave_sync();
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int sr = 4 * (t / SCS) + (lm >> 2), c = 4 * (t % SCS) + (lm & 3);
                if (!PART || sr < nreal) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) lbytes[sr * N + SC * (lg + 4 * j) + c] = (uint8_t)(li[t] >> (8 * j));
                }
            }
            wave_sync();
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                labo[t] = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kc = SC == 4 ? e : 2 * e + (t % SCS);
                    labo[t] |= (uint32_t)lbytes[(4 * (t / SCS) + lg) * N + lm + 16 * kc] << (8 * e);
                }
                asm volatile("" : "+v"(labo[t]));
            }
            wave_sync();
            if constexpr (PART) {
                // The partly filled last wave: its rows -- this phase's scratch, and the received samples of the frame before, kept as
                // complex floats across BOTH planes of the chunk -- reach into the plane positions behind its symbols, which the
                // FIR reads as the frame's end: the last symbol's fall tail (written below) and zeros.  Put the zeros back.
                if (nreal < SPWR) {
                    const int z1 = PRE + (s0 + SPWR) * B;
                    for (int i = PRE + (s0 + nreal) * B + TS + lane; i < z1; i += 64) { Hp[i] = 0u; Hp[plen + i] = 0u; }
                }
            }
            STAMPF(9);
            f4 tr[4], ti[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const h8 xa = __builtin_bit_cast(h8, (u4){xw[t][0], xw[t][1], xw[t][2], xw[t][3]});
                mma22(tr[t], ti[t], xa, brl, brh, bil, bih);
                mdft_twiddle(tr[t], ti[t], twr[t % SCS], twi[t % SCS]);
            }
#pragma unroll
            for (int gr = 0; gr < SGR; ++gr) {
                if constexpr (SC == 4) radix4_elems(tr[gr], ti[gr]);
                else radix8_elems(tr[2 * gr], ti[2 * gr], tr[2 * gr + 1], ti[2 * gr + 1]);
            }
            STAMPF(10);
            // Tx write: element e of set t = sample ka + 16 kc of symbol s0 + 4 group + g' (the transform returned swap(N x): real parts
            // in ti, imaginary parts in tr); sixteen lanes own one symbol, as in the quarter-wave layouts
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int s = s0 + 4 * (t / SCS) + lg;
                uint32_t *hrow = Hp + PRE + s * B;
                const bool lastsym = s == S - 1;
                const int DtH = lastsym ? 0 : 2 * tail_off + s * TS - (PRE + (s + 1) * B);
                const int DtL = lastsym ? 0 : DtH + S * TS - plen;
                if (PART && 4 * (t / SCS) + lg >= nreal) continue;       // (an unfilled slot stores nothing)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kc = SC == 4 ? e : 2 * e + (t % SCS);
                    const int tl = lm + 16 * kc;
                    const float re = ti[t][e], im = tr[t][e];
                    auto put = [&](int i) {
                        const float w = wtx[i];
                        uint32_t hi, lo;
                        split_h(mk(re * w, im * w), hi, lo);
                        const int tlm = i >= B ? -1 : 0;                    // (DtH, DtL are zero for the last symbol)
                        hrow[i + (DtH & tlm)] = hi;
                        hrow[i + plen + (DtL & tlm)] = lo;
                    };
                    put(tl + mu);
                    if (16 * kc + 15 >= N - mu) {
                        if (tl >= N - mu) put(tl + mu - N);
                    }
                    if (16 * kc < rho) {
                        if (tl < rho) put(tl + mu + N);
                    }
                }
            }
        } else if constexpr (MDX) {
            // labels and constellation words in INPUT element order: set c, element j = subcarrier N/16 (g + 4 j) + NC b + c;
            // the NC labels of one (j) sit in one staged word
            const int s = s0;
            const uint32_t *bw = reinterpret_cast<const uint32_t *>(row(0));
            const int lb = lane & 15, lg = lane >> 4;
            uint8_t *lbytes = reinterpret_cast<uint8_t *>(row(0));
            uint32_t labi[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n0 = (N / 16) * (lg + 4 * j) + NC * lb;
                uint32_t w = 0;
                if (!INJECT) {
                    const uint32_t bit = (uint32_t)n0 * (uint32_t)ks;
                    w = bw[bit >> 5] >> (bit & 31u);
                }
                labi[j] = 0;
                // byte c of the word: 0x80 when subcarrier n0 + c is NOT loaded (third part of the table: this layout's order);
                // the flag rides in the label byte down to phase D
                uint32_t am = 0;
                if constexpr (ALLOC) am = g_amask[(MDM ? 2 * NQ : NQ) + lane + 64 * j] & 0x80808080u;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    uint32_t Lb;
                    if (INJECT) Lb = p.labels[(inj * S + s) * N + n0 + c] & lmask;
                    else Lb = (w >> (c * ks)) & lmask;
                    labi[j] |= Lb << (8 * c);
                    xw[c][j] = qlw[Lb];
                    if constexpr (ALLOC) {
                        if ((am >> (8 * c)) & 0x80u) xw[c][j] = 0u;
                    }
                    if (DUMP) {
                        const hpair hw = __builtin_bit_cast(hpair, xw[c][j]);
                        if (p.dump.labels_tx) p.dump.labels_tx[s * N + n0 + c] = (uint8_t)Lb;
                        if (p.dump.X) p.dump.X[s * N + n0 + c] = make_float2((float)hw.y * qscale, (float)hw.x * qscale);
                    }
                }
                if constexpr (ALLOC) labi[j] |= am;
            }
            // ... and once through the (still unused) row into OUTPUT element order for phase D: NC label bytes per (j) at
            // byte n0, read back one byte per output element
            wave_sync();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n0 = (N / 16) * (lg + 4 * j) + NC * lb;
                if constexpr (NC == 4) *reinterpret_cast<uint32_t *>(lbytes + n0) = labi[j];
                else if constexpr (NC == 2) *reinterpret_cast<uint16_t *>(lbytes + n0) = (uint16_t)labi[j];
                else lbytes[n0] = (uint8_t)labi[j];
            }
            wave_sync();
#pragma unroll
            for (int kc = 0; kc < NC; ++kc) {
                labo[kc] = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) labo[kc] |= (uint32_t)lbytes[lane + 64 * j + 256 * kc] << (8 * j);
                asm volatile("" : "+v"(labo[kc]));
            }
            wave_sync();
            STAMPF(9);
            const mdft_consts dc = mdft_load(dce);
            f4 t2r[NC], t2i[NC];
            mdft_tw2(t2r, t2i);
            h8 xa[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) xa[c] = __builtin_bit_cast(h8, (u4){xw[c][0], xw[c][1], xw[c][2], xw[c][3]});
            f4 orr[NC], oi[NC];
            mdft_big<NC, true>(xa, xa, dc, t2r, t2i, orr, oi);
#pragma unroll
            for (int c = 0; c < NC; ++c) { xr[c] = orr[c]; xi[c] = oi[c]; }
            STAMPF(10);
            if constexpr (MDM) {
            // ---- layout 15: the symbol goes into its fp32 row (layout 9's frame: tx_write below, element j = sample lane + 64 j;
            // real parts in xi, imaginary parts in xr), the mask stage works on it there
            const v2f *xt;
            v2f *fb = fbw;
            const bool last = s == S - 1;
            {
                const int Bs = last ? 0x3fffffff : B;
                const int Dt = tail_off + s * TS - 4 - (s + 1) * B;
                const bool body_tail = rho < gq[WOFDM_G_BETA];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = lane + 64 * j;
                    const v2f x = mk(xi[0][j], xr[0][j]);
                    auto put_plain = [&](int i) { fb[i] = wmul(x, wtx[i]); };
                    auto put_tail = [&](int i) { fb[i + (i >= Bs ? Dt : 0)] = wmul(x, wtx[i]); };
                    if (body_tail) put_tail(t + mu);
                    else put_plain(t + mu);
                    if (63 + 64 * j >= N - L::CPCS_MAX)
                        if (t >= N - mu) put_plain(t + mu - N);
                    if (64 * j < L::CPCS_MAX)
                        if (t < rho) put_tail(t + mu + N);
                }
                xt = last ? fb + B : tailb + s * TS;
            }
            // dft_rc_filt as fast convolution (see the VALU form below: y = IDFT_MF(DFT_MF(x) . g_tmask), outputs at j = n + P - 1),
            // both 1024-point transforms on the matrix pipe and NO exchange between them: the forward transform (mdft_big<4>)
            // takes the symbol as element 4 (lane + 64 j) + c of set c and returns bin lane + 64 j + 256 kc in set kc; the
            // inverse one runs the transposed flow (mdft_big_dif4) from that order back to the first.  The samples enter scaled
            // by 2^-4 (so that the 16-term sums of stage 1 stay in the f16 range), which the spectrum table undoes; the inverse
            // transform is swap(DFT(swap(.))): the spectrum product is written swapped, real parts come back in the second array.
            const int P = gq[WOFDM_G_P];
            wave_sync();
            STAMPM(13);
            f4 m2r[4], m2i[4];
            {
                const u4 *dl = reinterpret_cast<const u4 *>(p.dftc) + 10 * 64 + lane;      // (rows 10..15: L2)
#pragma unroll
                for (int c = 1; c < 4; ++c) {
                    m2r[c] = __builtin_bit_cast(f4, dl[(2 * (c - 1)) * 64]);
                    m2i[c] = __builtin_bit_cast(f4, dl[(2 * (c - 1) + 1) * 64]);
                }
                m2r[0] = m2i[0] = (f4){0.f, 0.f, 0.f, 0.f};
            }
            // Which elements can fall where is known at compile time from N <= B <= P <= PMAX (this form of the mask runs when
            // 3 P - 2 <= 1024): element K = 256 j + c of lane l is sample 4 l + K.  Everything below tests only what can fail, and
            // instead of branching around a load or store sends the lanes it does not concern to a spare address (loads: the
            // zero words in front of the frame; stores: eight bytes at the end of the workgroup's LDS) -- per-lane bases once,
            // the element's K as the instruction's offset.
            constexpr int PMAX = (N + L::CPCS_MAX) < (maskfft_geo::MF + 2) / 3 ? (N + L::CPCS_MAX) : (maskfft_geo::MF + 2) / 3;
            const v2f *zeros = reinterpret_cast<const v2f *>(Hp);
            v2f *spare = reinterpret_cast<v2f *>(smem + (p.lds_bytes - 8));
            h8 mh[4], ml[4];
            {
                const v2f *pF = fb + 4 * lane, *pT = xt - B + 4 * lane;      // sample m = 4 l + K at pF[K] (row) or pT[K] (fall tail)
                const int m0 = 4 * lane;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    uint32_t wh[4], wl[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int KE = 256 * j + c;
                        wh[j] = 0u; wl[j] = 0u;
                        if (KE < PMAX) {                                    // (else: zeros at compile time)
                            const v2f *src = pF;
                            if (KE + 252 >= N) src = (m0 < B - KE) ? pF : pT;            // m < B: always so in the first quarter
                            if (KE + 252 >= N) src = (m0 < P - KE) ? src : zeros - KE;    // m < P
                            split_h(wmul(src[KE], 0.0625f), wh[j], wl[j]);
                        }
                    }
                    mh[c] = __builtin_bit_cast(h8, (u4){wh[0], wh[1], wh[2], wh[3]});
                    ml[c] = __builtin_bit_cast(h8, (u4){wl[0], wl[1], wl[2], wl[3]});
                    mma_operand_fence(mh[c], ml[c]);
                }
            }
            f4 Fr[4], Fi[4];
            mdft_big<4, false>(mh, ml, dc, m2r, m2i, Fr, Fi);
            STAMPM(14);
            f4 zr[4], zi[4];
            {
                // g_tmask here: the mask's spectrum as [kc][re | im][lane] rows of four (element j), times 2^4 (wofdm_abi.hip), from L2
                const f4 *G4 = reinterpret_cast<const f4 *>(g_tmask) + lane;
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    const f4 gr = G4[(2 * kc) * 64], gi = G4[(2 * kc + 1) * 64];
                    zr[kc] = Fr[kc] * gi + Fi[kc] * gr;
                    zi[kc] = Fr[kc] * gr - Fi[kc] * gi;
                }
            }
            f4 ymi[4], ymr[4];
            mdft_big_dif4(zr, zi, dc, m2r, m2i, ymi, ymr);
            STAMPM(15);
            // own row <- y[0..P): element 4 l + K is output n = 4 l + K - (P - 1)
            const int L4 = 4 * lane - (P - 1);
            {
                v2f *pF = fb + L4, *pT = last ? pF : tailb + s * TS - B + L4;
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int KE = 256 * j + c;
                        if (KE + 252 >= N - 1 && KE <= 2 * PMAX - 2) {                       // (else: never an output of the row)
                            v2f *dst = pF;
                            if (KE + 252 - (N - 1) >= N) dst = (L4 < B - KE) ? pF : pT;      // n < B, else the fall tail
                            if (KE < PMAX - 1) dst = (L4 >= -KE) ? dst : spare - KE;          // n >= 0
                            if (KE + 252 >= 2 * N - 1) dst = (L4 < P - KE) ? dst : spare - KE;   // n < P
                            dst[KE] = mk(ymr[c][j], ymi[c][j]);
                        }
                    }
            }
            // y[P..2P-1), which main_channel_mask.m adds to the first P - 1 samples of the NEXT symbol, is parked in the wave's own
            // row behind the frame: the next wave (and, for the part that reaches into that symbol's fall tail, the one after it)
            // adds it in phase B
            if (!last) {
                const int L5 = L4 - P;                                       // spill sample jx = 4 l + K + (L5 - 4 l)
                v2f *qF = mdm_park + wv * MDM_PK + L5;
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int KE = 256 * j + c;
                        if (KE + 252 >= 2 * N - 1) {                                        // (else: never part of the spill)
                            v2f *dst = qF;
                            if (KE < 2 * PMAX - 1) dst = (L5 >= -KE) ? dst : spare - KE;      // jx >= 0
                            if (KE + 252 >= 3 * N - 2) dst = (L5 < P - 1 - KE) ? dst : spare - KE;   // jx < P - 1
                            dst[KE] = mk(ymr[c][j], ymi[c][j]);
                        }
                    }
            }
            } else {
            // Tx write: output element (kc, j) = sample t = lane + 64 (j + 4 kc); real parts in xi, imaginary parts in xr
            const bool lastsym = s == S - 1;
            uint32_t *hrow = Hp + 8 + 2 * BR * s;
            const int DtH = lastsym ? 2 * BR - B : 2 * tail_off + s * TS - B - (8 + 2 * BR * s);
            const int DtL = lastsym ? 2 * BR - B + VT : DtH + S * TS;
            const bool body_tail = rho < gq[WOFDM_G_BETA];
            uint32_t *pH = hrow + (lane + mu), *pL = pH + BR;
            const float *pW = wtx + (lane + mu);
            auto tx12 = [&](auto body_tail_c) {
#pragma unroll
                for (int kc = 0; kc < NC; ++kc) {
                    const v2f w01 = mk(pW[256 * kc], pW[256 * kc + 64]), w23 = mk(pW[256 * kc + 128], pW[256 * kc + 192]);
                    const v2f re01 = mk(xi[kc].x, xi[kc].y) * w01, re23 = mk(xi[kc].z, xi[kc].w) * w23;
                    const v2f im01 = mk(xr[kc].x, xr[kc].y) * w01, im23 = mk(xr[kc].z, xr[kc].w) * w23;
                    const float re[4] = {re01.x, re01.y, re23.x, re23.y}, im[4] = {im01.x, im01.y, im23.x, im23.y};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 64 * j + 256 * kc;
                        uint32_t hi, lo;
                        split_h(mk(re[j], im[j]), hi, lo);
                        if constexpr (decltype(body_tail_c)::value) {
                            const int i = lane + e + mu;
                            const bool tl = i >= B;
                            hrow[i + (tl ? DtH : 0)] = hi;
                            hrow[i + (tl ? DtL : BR)] = lo;
                        } else {
                            pH[e] = hi;
                            pL[e] = lo;
                        }
                    }
                }
                // prefix / suffix copies (one exec-masked region per element that can have one)
#pragma unroll
                for (int kc = 0; kc < NC; ++kc)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 64 * j + 256 * kc;
                        if (63 + e >= N - L::CPCS_MAX)
                            if (e + 63 >= N - mu) {
                                if (lane + e >= N - mu) {
                                    const float w = pW[e - N];
                                    uint32_t hi, lo;
                                    split_h(mk(xi[kc][j] * w, xr[kc][j] * w), hi, lo);
                                    pH[e - N] = hi;
                                    pL[e - N] = lo;
                                }
                            }
                        if (e < L::CPCS_MAX)
                            if (e < rho) {
                                if (lane + e < rho) {
                                    const int i = lane + e + mu + N;
                                    uint32_t hi, lo;
                                    split_h(mk(xi[kc][j] * wtx[i], xr[kc][j] * wtx[i]), hi, lo);
                                    const bool tl = i >= B;
                                    hrow[i + (tl ? DtH : 0)] = hi;
                                    hrow[i + (tl ? DtL : BR)] = lo;
                                }
                            }
                    }
            };
            if (body_tail) tx12(std::true_type{});
            else tx12(std::false_type{});
            }
        } else {
#pragma unroll
        for (int u = 0; u < VS; ++u) {
            const int s = sym_of(u);
            const uint32_t *bw = reinterpret_cast<const uint32_t *>(row(s - s0));
#pragma unroll
            for (int q = 0; q < VB; ++q) {
                const int j = lane + 64 * q;
                lab[u][q] = 0;
                if (owns(q)) {
                    // byte r of the word: 0x80 when subcarrier j + r N/4 is NOT loaded; the flag
                    // rides in the label word (labels use 6 bits at most) down to phase D
                    uint32_t am = 0;
                    // (second half of the table: the same flags in the quarter-wave element order)
                    if constexpr (ALLOC) am = g_amask[QW ? NQ + 16 * q + llq : j] & 0x80808080u;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = sub_of(q, r);
                        uint32_t Lb;
                        if (INJECT) {
                            Lb = p.labels[(inj * S + s) * N + n] & lmask;
                        } else if constexpr (QW) {
                            const uint32_t bit = (uint32_t)n * (uint32_t)ks;
                            Lb = (bw[bit >> 5] >> (bit & 31u)) & lmask;
                        } else {
                            // n*ks = j*ks + r*(NQ*ks); the second term is a multiple of 32 bits
                            const uint32_t bit = (uint32_t)j * (uint32_t)ks;
                            Lb = (bw[(bit >> 5) + r * (NQ * ks / 32)] >> (bit & 31u)) & lmask;
                        }
                        lab[u][q] |= Lb << (8 * r);
                        if constexpr (MDFT) {
                            xw[u][r] = qlw[Lb];
                            if constexpr (ALLOC) {
                                if ((am >> (8 * r)) & 0x80u) xw[u][r] = 0u;
                            }
                            if (DUMP) {
                                const hpair hw = __builtin_bit_cast(hpair, xw[u][r]);
                                if (p.dump.labels_tx) p.dump.labels_tx[s * N + n] = (uint8_t)Lb;
                                if (p.dump.X) p.dump.X[s * N + n] = make_float2((float)hw.y * qscale, (float)hw.x * qscale);
                            }
                        } else {
                        v[u][q][r] = qlut[Lb];
                        if constexpr (ALLOC) {
                            if ((am >> (8 * r)) & 0x80u) v[u][q][r] = mk(0.f, 0.f);
                        }
                        if (DUMP) {
                            if (p.dump.labels_tx) p.dump.labels_tx[s * N + n] = (uint8_t)Lb;
                            if (p.dump.X) p.dump.X[s * N + n] = make_float2(v[u][q][r].x, v[u][q][r].y);
                        }
                        }
                    }
                    if constexpr (ALLOC) lab[u][q] |= am;
                    // opaque at the large sizes: otherwise the unpacked labels themselves stay alive
                    // (and spill) all the way to the pilot estimate and the demapper instead of
                    // this one word
                    if constexpr (N >= 512 || QW || MDFT) asm volatile("" : "+v"(lab[u][q]));
                }
            }
        }
        wave_sync();
        STAMPF(9);
        WAVE_PRIO(WOFDM_PRIO_X);
        if constexpr (MDFT && WOFDM_MDFT_PIPELINE) {
            // The transform as a pipeline over the wave's four symbols (round 4): the sixteen MFMAs of stage 1 are issued at once,
            // then every group is "the six MFMAs of symbol u's second stage, and BEHIND them the twiddle and the f16 split of symbol
            // u + 1" (26 vector instructions, 110 cycles, while the chain takes 96 on the matrix pipe) -- the guard behind that work
            // needs no wait states.  Only the last symbol's chain, which has nothing behind it, ends in the twelve.
            const mdft_consts dc = mdft_load(dce);
            f4 tr[4], ti[4];
            h8 xa[4], th[4], tl[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xa[u] = __builtin_bit_cast(h8, (u4){xw[u][0], xw[u][1], xw[u][2], xw[u][3]});
                mma22_issue(tr[u], ti[u], xa[u], dc.brl, dc.brh, dc.bil, dc.bih);
            }
            asm volatile("" : "+v"(tr[0]), "+v"(ti[0]), "+v"(tr[1]), "+v"(ti[1]), "+v"(tr[2]), "+v"(ti[2]), "+v"(tr[3]), "+v"(ti[3]));
            mdft_twiddle(tr[0], ti[0], dc.twr, dc.twi);
            mdft_split4(tr[0], ti[0], th[0], tl[0]);
            asm volatile("" : "+v"(th[0]), "+v"(tl[0]) : "v"(xa[0]), "v"(xa[1]), "v"(xa[2]), "v"(xa[3]), "v"(dc.brl), "v"(dc.brh), "v"(dc.bil),
                         "v"(dc.bih), "v"(tr[1]), "v"(ti[1]), "v"(tr[2]), "v"(ti[2]), "v"(tr[3]), "v"(ti[3]));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                mma33_issue(xr[u], xi[u], dc.arh, tl[u], dc.arl, th[u], dc.arh, th[u], dc.aih, tl[u], dc.ail, th[u], dc.aih, th[u]);
                if (u < 3) {
                    WOFDM_TIE2(xr[u], xi[u], tr[u + 1], ti[u + 1]);
                    mdft_twiddle(tr[u + 1], ti[u + 1], dc.twr, dc.twi);
                    mdft_split4(tr[u + 1], ti[u + 1], th[u + 1], tl[u + 1]);
                    asm volatile("" : "+v"(th[u + 1]), "+v"(tl[u + 1]) : "v"(th[u]), "v"(tl[u]), "v"(xr[u]), "v"(xi[u]), "v"(dc.arh), "v"(dc.arl),
                                 "v"(dc.aih), "v"(dc.ail));
                } else {
                    asm volatile(WOFDM_MMA_TAIL : "+v"(xr[u]), "+v"(xi[u]) : "v"(th[u]), "v"(tl[u]), "v"(dc.arh), "v"(dc.arl), "v"(dc.aih), "v"(dc.ail));
                }
            }
        } else if constexpr (MDFT) {
            const mdft_consts dc = mdft_load(dce);
            f4 tr[4], ti[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const h8 xa = __builtin_bit_cast(h8, (u4){xw[u][0], xw[u][1], xw[u][2], xw[u][3]});
                mma22(tr[u], ti[u], xa, dc.brl, dc.brh, dc.bil, dc.bih);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                mdft_twiddle(tr[u], ti[u], dc.twr, dc.twi);
                h8 th, tl;
                mdft_split4(tr[u], ti[u], th, tl);
                mma33(xr[u], xi[u], dc.arh, tl, dc.arl, th, dc.arh, th, dc.aih, tl, dc.ail, th, dc.aih, th);
            }
        } else if constexpr (QW) fft_qw<+1>(v, row(usq), tw, llq);
        else fft_wave<N, +1, SPW>(v, fbw, B, tw, lane);     // v = N x[t]
        STAMPF(10);

        // add_redundancy (m:419-439) x diag(windowTx) (m:375): x[t] lands at i = t+mu, and at
        // t+mu-N (prefix) / t+mu+N (suffix) when those exist.  i >= B is the fall tail that
        // overlaps the next symbol (m:253-256): parked in tailb until barrier 1.
        // Only suffix copies -- and body copies when rho < beta -- can land in the fall tail
        // (i >= B): those go to the tail buffer, Dt v2f's away from fb+i, branch-free; the last
        // symbol has no successor, its tail stays in the frame buffer.  Copies that cannot
        // exist for this (q, r) under cp, cs <= CPCS_MAX are dropped at compile time.
        auto tx_write = [&](auto body_may_tail) {
#pragma unroll
            for (int u = 0; u < SPW; ++u) {
                const int s = s0 + u;
                v2f *fb = fbw + u * B;
                const int Bs = (s == S - 1) ? 0x3fffffff : B;
                // tailb + s TS - (fb + B) in v2f units, from the LDS offsets (a pointer
                // difference would be taken on 64-bit generic addresses)
                // (layout 9: row s starts 4 v2f into the buffer -- 8 zero words -- instead of LT - 1)
                const int Dt = tail_off + s * TS - (FIR8M ? 4 : LT - 1) - (s + 1) * B;
#pragma unroll
                for (int q = 0; q < BPL; ++q) {
                    const int j = lane + 64 * q;
                    if (FULL || j < NQ) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int t = j + r * NQ;
                            const v2f x = v[u][q][r];
                            auto put_plain = [&](int i) { fb[i] = x * wtx[i]; };
                            auto put_tail = [&](int i) { fb[i + (i >= Bs ? Dt : 0)] = x * wtx[i]; };
                            if (decltype(body_may_tail)::value) put_tail(t + mu);
                            else put_plain(t + mu);
                            if (63 + 64 * q + r * NQ >= N - L::CPCS_MAX)
                                if (t >= N - mu) put_plain(t + mu - N);
                            if (64 * q + r * NQ < L::CPCS_MAX)
                                if (t < rho) put_tail(t + mu + N);
                        }
                    }
                }
            }
        };
        if constexpr (FIR8 && !FIR8M) {
            // one symbol per wave, rows of plane H and plane L side by side; the fall tail goes to the
            // tail planes, the last symbol's into the virtual row behind the frame
            const int s = s0;
            uint32_t *hrow = Hp + 8 + 2 * BR * s;
            const bool lastsym = s == S - 1;
            const int DtH = lastsym ? 2 * BR - B : 2 * tail_off + s * TS - B - (8 + 2 * BR * s);
            const int DtL = lastsym ? 2 * BR - B + VT : DtH + S * TS;
            const bool body_tail = rho < gq[WOFDM_G_BETA];
            // per-lane bases once, element offsets as instruction immediates (see the quarter-wave form)
            uint32_t *pH = hrow + (lane + mu), *pL = pH + BR;
            const float *pW = wtx + (lane + mu);
            uint32_t *pHp = pH - N, *pLp = pL - N;
            const float *pWp = pW - N;
            // (two instantiations: whether body copies can land in the fall tail is a property of the
            // structure, decided once per frame instead of once per element)
            auto tx8 = [&](auto body_tail_c) {
#pragma unroll
            for (int q = 0; q < BPL; ++q) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = 64 * q + r * NQ, t = lane + e;
                    const v2f x = v[0][q][r];
                    uint32_t hi, lo;
                    auto put_tail = [&](int i) {
                        split_h(x * wtx[i], hi, lo);
                        const bool tl = i >= B;
                        hrow[i + (tl ? DtH : 0)] = hi;
                        hrow[i + (tl ? DtL : BR)] = lo;
                    };
                    if constexpr (decltype(body_tail_c)::value) {
                        put_tail(t + mu);
                    } else {
                        split_h(x * pW[e], hi, lo);
                        pH[e] = hi;
                        pL[e] = lo;
                    }
                    if (63 + e >= N - L::CPCS_MAX)
                        if (e + 63 >= N - mu) {
                            if (t >= N - mu) {
                                split_h(x * pWp[e], hi, lo);
                                pHp[e] = hi;
                                pLp[e] = lo;
                            }
                        }
                    if (e < L::CPCS_MAX)
                        if (e < rho) {
                            if (t < rho) put_tail(t + mu + N);
                        }
                }
            }
            };
            if (body_tail) tx8(std::true_type{});
            else tx8(std::false_type{});
        } else if constexpr (MDFT) {
            // four symbols, element j of symbol u = sample t = lane + 64 j: 64 consecutive words per store.  The transform
            // returned swap(N x[t]): the real parts are in xi, the imaginary parts in xr.  Whether a lane's element has a
            // prefix / suffix copy does not depend on the symbol: ONE exec-masked region per element for all four symbols.
            const bool body_tail = rho < gq[WOFDM_G_BETA];
            const int i0 = lane + mu;                               // position of element 0 in its symbol's row
            uint32_t *pH0 = Hp + PRE + s0 * B + i0;
            const float *pW = wtx + i0;
            // word offsets from a row's position to the same position of its fall tail (the last symbol of the frame has
            // no successor: its tail stays in the frame buffer)
            int dtH[4], dtL[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = s0 + u;
                const bool last = s == S - 1;
                dtH[u] = last ? 0 : 2 * tail_off + s * TS - (PRE + (s + 1) * B);
                dtL[u] = last ? 0 : dtH[u] + S * TS - plen;
            }
            auto txm = [&](auto body_tail_c, auto small_c) {
            constexpr int CPB = decltype(small_c)::value ? 48 : L::CPCS_MAX;
            const v2f w01 = mk(pW[0], pW[64]), w23 = mk(pW[128], pW[192]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                uint32_t *pH = pH0 + u * B, *pL = pH + plen;
                const v2f re01 = mk(xi[u].x, xi[u].y) * w01, re23 = mk(xi[u].z, xi[u].w) * w23;
                const v2f im01 = mk(xr[u].x, xr[u].y) * w01, im23 = mk(xr[u].z, xr[u].w) * w23;
                const float re[4] = {re01.x, re01.y, re23.x, re23.y}, im[4] = {im01.x, im01.y, im23.x, im23.y};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t hi, lo;
                    split_h(mk(re[j], im[j]), hi, lo);
                    if constexpr (decltype(body_tail_c)::value) {
                        const int tlm = i0 + 64 * j >= B ? -1 : 0;      // (dtH, dtL are zero for the last symbol)
                        pH[64 * j + (dtH[u] & tlm)] = hi;
                        pL[64 * j + (dtL[u] & tlm)] = lo;
                    } else {
                        pH[64 * j] = hi;
                        pL[64 * j] = lo;
                    }
                }
            }
            STAMPX(9);
            // prefix: samples t >= N - mu once more, N words in front
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (64 * j + 63 >= N - CPB)
                    if (64 * j + 63 >= N - mu) {
                        if (lane + 64 * j >= N - mu) {
                            const float w = pW[64 * j - N];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                uint32_t *pH = pH0 + u * B, *pL = pH + plen;
                                uint32_t hi, lo;
                                split_h(mk(xi[u][j] * w, xr[u][j] * w), hi, lo);
                                pH[64 * j - N] = hi;
                                pL[64 * j - N] = lo;
                            }
                        }
                    }
            }
            STAMPX(10);
            // suffix: samples t < rho once more, N words behind -- in the fall tail from position B on
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (64 * j < CPB)
                    if (64 * j < rho) {
                        if (lane + 64 * j < rho) {
                            const float w = pW[64 * j + N];
                            const int tlm = i0 + 64 * j + N >= B ? -1 : 0;       // (a mask, not a branch per store)
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                uint32_t *pH = pH0 + u * B, *pL = pH + plen;
                                uint32_t hi, lo;
                                split_h(mk(xi[u][j] * w, xr[u][j] * w), hi, lo);
                                pH[64 * j + N + (dtH[u] & tlm)] = hi;
                                pL[64 * j + N + (dtL[u] & tlm)] = lo;
                            }
                        }
                    }
            }
            };
            const bool small = mu <= 48 && rho <= 48;
            STAMPX(8);
            if (body_tail) { if (small) txm(std::true_type{}, std::true_type{}); else txm(std::true_type{}, std::false_type{}); }
            else { if (small) txm(std::false_type{}, std::true_type{}); else txm(std::false_type{}, std::false_type{}); }
        } else if constexpr (FIRQ) {
            // the same copies, every sample split into its two packed-f16 words; the fall tail goes to
            // the tail planes, DtH / DtL words away from the row's word in plane H / L
            const int s = s0 + usq;
            uint32_t *hrow = Hp + PRE + s * B;
            const int Bs = (s == S - 1) ? 0x3fffffff : B;
            const int DtH = 2 * tail_off + s * TS - (PRE + (s + 1) * B);
            const int DtL = DtH + S * TS - plen;
            const bool body_tail = rho < gq[WOFDM_G_BETA];
            // per-lane bases once, element offsets as instruction immediates: sample t + mu of the row
            // with t = llq + 16 tt sits 16 tt words behind the base; prefix copies N words in front of it
            uint32_t *pH = hrow + (llq + mu), *pL = pH + plen;
            const float *pW = wtx + (llq + mu);
            uint32_t *pHp = pH - N, *pLp = pL - N;
            const float *pWp = pW - N;
            // (four instantiations: whether body copies can reach the fall tail, and whether prefix and suffix are
            // at most 48 samples long -- then only three elements each can have a copy, not nine and eight: every
            // such site is an exec-masked region with its own skip branch)
            auto tx4 = [&](auto body_tail_c, auto small_c) {
            constexpr int CPB = decltype(small_c)::value ? 48 : L::CPCS_MAX;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int tt = q + 4 * r, t = llq + 16 * tt;
                    const v2f x = v[0][q][r];
                    uint32_t hi, lo;
                    if constexpr (decltype(body_tail_c)::value) {
                        const int i = t + mu;
                        split_h(x * wtx[i], hi, lo);
                        const bool tl = i >= Bs;
                        hrow[i + (tl ? DtH : 0)] = hi;
                        hrow[i + plen + (tl ? DtL : 0)] = lo;
                    } else {
                        split_h(x * pW[16 * tt], hi, lo);
                        pH[16 * tt] = hi;
                        pL[16 * tt] = lo;
                    }
                    // (the wave-uniform test first: most elements have no lane in the prefix / suffix)
                    if (15 + 16 * tt >= N - CPB)
                        if (16 * tt + 15 >= N - mu) {
                            if (t >= N - mu) {
                                split_h(x * pWp[16 * tt], hi, lo);
                                pHp[16 * tt] = hi;
                                pLp[16 * tt] = lo;
                            }
                        }
                    if (16 * tt < CPB)
                        if (16 * tt < rho) {
                            if (t < rho) {
                                const int i = t + mu + N;
                                split_h(x * wtx[i], hi, lo);
                                const bool tl = i >= Bs;
                                hrow[i + (tl ? DtH : 0)] = hi;
                                hrow[i + plen + (tl ? DtL : 0)] = lo;
                            }
                        }
                }
            };
            const bool small = mu <= 48 && rho <= 48;
            if (body_tail) { if (small) tx4(std::true_type{}, std::true_type{}); else tx4(std::true_type{}, std::false_type{}); }
            else { if (small) tx4(std::false_type{}, std::true_type{}); else tx4(std::false_type{}, std::false_type{}); }
        } else if constexpr (QW) {
            // the same copies with per-lane symbol geometry (the four quarters of the wave sit in
            // four different symbols)
            const int s = s0 + usq;
            v2f *fb = fbw + usq * B;
            const int Bs = (s == S - 1) ? 0x3fffffff : B;
            const int Dt = tail_off + s * TS - (LT - 1) - (s + 1) * B;
            const bool body_tail = rho < gq[WOFDM_G_BETA];        // only then a body copy can reach the tail
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = llq + 16 * (q + 4 * r);
                    const v2f x = v[0][q][r];
                    auto put_plain = [&](int i) { fb[i] = x * wtx[i]; };
                    auto put_tail = [&](int i) { fb[i + (i >= Bs ? Dt : 0)] = x * wtx[i]; };
                    if (body_tail) put_tail(t + mu);
                    else put_plain(t + mu);
                    if (15 + 16 * (q + 4 * r) >= N - L::CPCS_MAX)
                        if (t >= N - mu) put_plain(t + mu - N);
                    if (16 * (q + 4 * r) < L::CPCS_MAX)
                        if (t < rho) put_tail(t + mu + N);
                }
        } else {
            if (rho < gq[WOFDM_G_BETA]) tx_write(std::true_type{});
            else tx_write(std::false_type{});
        }

        if constexpr (TXFFT) {
            // dft_rc_filt as fast convolution.  y[n] = sum_m g[(n - m) mod (2P-1)] x[m], n < 2P-1,
            // is the linear convolution of x with gt[t] = g[(t - (P-1)) mod (2P-1)], t < 3P-2, read
            // at j = n + P - 1; an MF-point circular convolution gives exactly those samples when
            // 3P - 2 <= MF (only j < P-1 alias).  g_tmask holds FFT_MF(gt) / MF.  All waves transform at
            // once, each through its own half-size exchange buffer (fft_1024_half).
            constexpr int MF = maskfft_geo::MF, MQ = MF / 4, MBPL = MQ / 64, SLOTS = maskfft_geo::SLOTS;
            const int P = gq[WOFDM_G_P];
            const int s = s0;
            v2f *fb = fbw;
            const bool last = s == S - 1;
            const v2f *xt = last ? fb + B : tailb + s * TS;
            static_assert(MF == 1024 && MBPL == 4 && SLOTS == 8, "fft_1024_half: 16 half-size exchange rows");
            v2f *scr = mscr + wv * (MF / 2);          // every wave its own 4 KB exchange buffer
            v2f y[1][MBPL][4];
            wave_sync();
            STAMPM(13);
#pragma unroll
            for (int q = 0; q < MBPL; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = lane + 64 * q + r * MQ;
                    v2f xm = mk(0.f, 0.f);
                    // (P <= N + CPCS_MAX: the elements behind that are zeros at compile time, which the first stage of the
                    // transform folds away)
                    if (64 * q + r * MQ < N + L::CPCS_MAX)
                        if (m < P) xm = (m < B) ? fb[m] : xt[m - B];
                    y[0][q][r] = xm;
                }
            fft_1024_half<-1>(y, scr, mtw, lane);
            STAMPM(14);
#pragma unroll
            for (int q = 0; q < MBPL; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    y[0][q][r] = cmul(y[0][q][r], ldg2(g_tmask + lane + 64 * q + r * MQ));
            fft_1024_half<+1>(y, scr, mtw, lane);
            STAMPM(15);
            // own row <- y[0..P): element j = n + P - 1
#pragma unroll
            for (int q = 0; q < MBPL; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = lane + 64 * q + r * MQ - (P - 1);
                    if (n >= 0 && n < P) {
                        if (n < B || last) fb[n] = y[0][q][r];
                        else tailb[s * TS + (n - B)] = y[0][q][r];
                    }
                }
            if constexpr (RELAXM) {
                wave_sync();
                post_flag(&flags[32 + wv], iter, lane);
                if (!last) wait_flag(&flags[32 + wv + 1], iter, &flags[20]);
            } else {
                __syncthreads();
            }
            if (!last) {
                const bool nlast = s + 1 == S - 1;
                v2f *fn = fb + B;
#pragma unroll
                for (int q = 0; q < MBPL; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = lane + 64 * q + r * MQ - (P - 1) - P;
                        if (j >= 0 && j < P - 1) {
                            v2f *dst = (j < B || nlast) ? fn + j : tailb + (s + 1) * TS + (j - B);
                            *dst = *dst + y[0][q][r];
                        }
                    }
            }
        }
        if constexpr (TXMASK) {
            // dft_rc_filt (main_channel_mask.m:398-417): every windowed symbol x (P samples,
            // zero-padded to 2P-1) is circularly convolved with the mask's impulse response g:
            // y[n] = sum_m g[(n - m) mod (2P-1)] x[m].  y[0..P) replaces the symbol, y[P..2P-1)
            // is added to the FIRST P-1 samples of the next symbol's row (m:411-415), the last
            // symbol's spill is dropped.  Lane l owns NO consecutive outputs; the table values
            // slide through registers (one LDS read per input sample and lane).
            constexpr int NO = mask_geo<N>::NO, MB = mask_geo<N>::MB, OFF = mask_geo<N>::RG_OFF;
            const int P = gq[WOFDM_G_P];
            const int s = s0;
            v2f *fb = fbw;
            const bool last = s == S - 1;
            // the symbol's row: samples [0, B) in the frame slice, [B, P) in the tail buffer
            // (the last symbol keeps them in the frame buffer)
            const v2f *xt = last ? fb + B : tailb + s * TS;
            wave_sync();
            v2f y[NO];
#pragma unroll
            for (int i = 0; i < NO; ++i) y[i] = mk(0.f, 0.f);
            for (int m0 = 0; m0 < P; m0 += MB) {
                v2f win[NO + MB - 1];
                const v2f *rb = rg + (OFF + lane * NO - m0 - (MB - 1));
#pragma unroll
                for (int j = 0; j < NO + MB - 1; ++j) win[j] = rb[j];
#pragma unroll
                for (int mm = 0; mm < MB; ++mm) {
                    const int m = m0 + mm;
                    v2f xm = mk(0.f, 0.f);
                    if (m < P) xm = (m < B) ? fb[m] : xt[m - B];
                    const v2f xn = mk(-xm.y, xm.y);
#pragma unroll
                    for (int i = 0; i < NO; ++i) {
                        const v2f gg = win[i - mm + MB - 1];        // y += x g (complex)
                        y[i] = __builtin_elementwise_fma(xm.xx, gg, y[i]);
                        y[i] = __builtin_elementwise_fma(xn, gg.yx, y[i]);
                    }
                }
            }
            wave_sync();
            // own row <- y[0..P)
#pragma unroll
            for (int i = 0; i < NO; ++i) {
                const int n = lane * NO + i;
                if (n < P) {
                    if (n < B || last) fb[n] = y[i];
                    else tailb[s * TS + (n - B)] = y[i];
                }
            }
            if constexpr (RELAXM) {
                wave_sync();
                post_flag(&flags[32 + wv], iter, lane);
                if (!last) wait_flag(&flags[32 + wv + 1], iter, &flags[20]);
            } else {
                __syncthreads();
            }
            // next symbol's row += y[P..2P-1)
            if (!last) {
                const bool nlast = s + 1 == S - 1;
                v2f *fn = fb + B;
#pragma unroll
                for (int i = 0; i < NO; ++i) {
                    const int j = lane * NO + i - P;
                    if (j >= 0 && j < P - 1) {
                        v2f *dst = (j < B || nlast) ? fn + j : tailb + (s + 1) * TS + (j - B);
                        *dst = *dst + y[i];
                    }
                }
            }
        }
        }       // (layouts other than 12)
        }
        DELAY_AT(2);
        STAMP(0);
        WAVE_PRIO(WOFDM_PRIO_B3);
        if constexpr (RELAXF) {
            // ---- "barrier" 1: publish "my symbols are written"; phase B waits for the
            // predecessor wave only (its last L-1 samples and its fall tail)
            wave_sync();
            if (!(WOFDM_FAULT_SKIP_FLAG && wv == 1 && iter == 3)) post_flag(&flags[wv], iter, lane);
            if (wv > 0) wait_flag(&flags[wv - 1], iter, &flags[20]);
            if constexpr (RELAXM) {
                if (wv > 1) wait_flag(&flags[wv - 2], iter, &flags[20]);
            }
        } else {
            __syncthreads();                                                 // ---- barrier 1
        }
        STAMP(1);
        DELAY_AT(3);

        // ------------------------------------------------------------ B: overlap-add, noise, FIR
        // (large DFTs: the lane id is made opaque again per phase, or per-lane index vectors of one
        // phase's FFT are kept for the next phase's FFT -- in scratch, at the 128-VGPR limit)
        if constexpr (RELAUNDER) asm volatile("" : "+v"(lane));
        if constexpr (FIRM) {
        // The 21-tap complex FIR as a block-Toeplitz product on the matrix pipe.  One
        // v_mfma_f32_16x16x32_f16 tile = 8 consecutive outputs (re and im rows interleaved: 16 rows)
        // of 16 blocks (columns), K = 32 window samples x (re, im) in two instructions of K = 32:
        //   y[8b + i] = sum_l h[l] x[8b + i - l]:   A[2i + o][2jj + c] = { hr, -hi ; hi, hr }[o][c] of
        //   tap i + 24 - jj,   B[2jj + c][b] = (re, im)[c] of x[8b - 24 + jj],   jj = 0..31.
        // Operands are f16 pairs (hi + lo, split_h): h_hi x_hi + h_hi x_lo + h_lo x_hi in fp32
        // accumulation = 6 MFMAs per tile of 128 samples; lane (column n, row group g) receives the
        // complex samples 8n + 2g and 8n + 2g + 1 of the tile as (re, im, re, im) -- the unit noise of
        // exactly these two samples is one Philox block.  The A operands (per channel, built by the
        // host in MFMA layout) come from L2; the B operands are 16-byte rows of the f16 planes.
        GEO_PHASE();
        const int S = gq[WOFDM_G_S], B = gq[WOFDM_G_B], beta = gq[WOFDM_G_BETA], NL = gq[WOFDM_G_NL];
        const int BR = (FIR8 && !FIR8M) ? ((B + 3) & ~3) : B;          // (LDS row stride per plane, see phase A)
        (void)BR;
        const int plen = FIRQ ? gq[WOFDM_G_FBUF] : 0;
        const int W = PART ? (S + SPWR - 1) / SPWR : S / SPW;
        const int nreal = PART ? min(SPWR, S - s0) : SPW;        // (layout 16: the symbol slots this wave fills)
        uint32_t *Lp = Hp + plen;
        TAILS();
        const int ln = lane & 15, lg = lane >> 4;       // MFMA column / row group of the lane
        int ch_now = __builtin_amdgcn_readfirstlane(ch);
        asm volatile("" : "+s"(ch_now));
        const u4 *fa = reinterpret_cast<const u4 *>(g_fira) + (size_t)ch_now * 256 + lane;
        h8 A[4];                                   // [2 half + part]: window half 0/1, h_hi / h_lo
#pragma unroll
        for (int a = 0; a < 4; ++a) A[a] = __builtin_bit_cast(h8, fa[64 * a]);
        if constexpr (FIR8M) {
            // Tx-mask variants: the row still holds the masked symbol as fp32 (own part and the predecessor's spill: both
            // complete, barrier 1).  Overlap-add of the previous symbol's fall tail (m:253-259) in fp32, then the row --
            // the same bytes -- becomes its two f16 planes in place; the last symbol's own fall tail, which sits behind its
            // row, becomes the virtual row the trailing tile reads.  One more hand-over: the successor's first tile reaches
            // back into this row's last samples (flags [48 + w], or a barrier in the instrumented kernels).
            // (layout 15: the mask stage of the wave in front parked its spill -- P - 1 samples from this row's first on; the part
            // beyond B reaches into THIS symbol's fall tail, which the next wave, or the virtual row, takes along)
            v2f *fb = reinterpret_cast<v2f *>(Hp + 8 + 2 * B * s0);
            if (s0 > 0 && lane < beta) {
                v2f t = tailb[(s0 - 1) * beta + lane];
                if constexpr (MDM) {
                    if (s0 > 1 && lane < beta - 1) t = t + mdm_park[(s0 - 2) * MDM_PK + B + lane];
                }
                fb[lane] = fb[lane] + t;
            }
            wave_sync();
            constexpr int RQ = (N + L::CPCS_MAX + 63) / 64;
            v2f xs[RQ];
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                const int i = lane + 64 * q;
                xs[q] = i < B ? fb[i] : mk(0.f, 0.f);
            }
            if constexpr (MDM) {
                if (s0 > 0) {
                    const int lim = min(B, gq[WOFDM_G_P] - 1);
                    const v2f *pk = mdm_park + (s0 - 1) * MDM_PK, *zeros = reinterpret_cast<const v2f *>(Hp);
#pragma unroll
                    for (int q = 0; q < RQ; ++q) {
                        const int i = lane + 64 * q;
                        if (64 * q < MDM_PK) xs[q] = xs[q] + *((i < lim) ? pk + i : zeros);
                    }
                }
            }
            const bool lastsym = s0 == S - 1;
            v2f xtl = mk(0.f, 0.f);
            if (lastsym && lane < beta) {
                xtl = fb[B + lane];
                if constexpr (MDM) {
                    if (s0 > 0 && lane < beta - 1) xtl = xtl + mdm_park[(s0 - 1) * MDM_PK + B + lane];
                }
            }
            wave_sync();
            uint32_t *hrow = Hp + 8 + 2 * B * s0;
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                const int i = lane + 64 * q;
                if (i < B) {
                    uint32_t hi, lo;
                    split_h(xs[q], hi, lo);
                    hrow[i] = hi;
                    hrow[B + i] = lo;
                }
            }
            if (lastsym && lane < VT) {
                uint32_t hi = 0, lo = 0;
                if (lane < beta) split_h(xtl, hi, lo);
                hrow[2 * B + lane] = hi;
                hrow[2 * B + VT + lane] = lo;
            }
            wave_sync();
            if constexpr (RELAXM) {
                post_flag(&flags[48 + wv], iter, lane);
                if (wv > 0) wait_flag(&flags[48 + wv - 1], iter, &flags[20]);
            } else {
                __syncthreads();
            }
        } else if constexpr (FIR8) {
            // overlap-add of the previous symbol's fall tail (m:253-259), in fp32, re-split
            if (s0 > 0 && lane < beta) {
                uint32_t *hw = Hp + 8 + 2 * BR * s0 + lane;
                const int it = (s0 - 1) * beta + lane;
                uint32_t hi, lo;
                split_h(join_h(hw[0], hw[BR]) + join_h(tH[it], tL[it]), hi, lo);
                hw[0] = hi;
                hw[BR] = lo;
            }
        } else {
#pragma unroll
            for (int sb = 0; sb < SPW; sb += 4) {              // (sixteen lanes per symbol, four symbols at a time)
                const int s = s0 + sb + lg;
                if (s > 0 && ln < beta && (!PART || sb + lg < nreal)) {
                    const int idx = PRE + s * B + ln, it = (s - 1) * beta + ln;
                    uint32_t hi, lo;
                    split_h(join_h(Hp[idx], Lp[idx]) + join_h(tH[it], tL[it]), hi, lo);
                    Hp[idx] = hi;
                    Lp[idx] = lo;
                }
            }
        }
        wave_sync();
        if (DUMP) {
            __syncthreads();
            if (p.dump.tx)
                for (int i = tid; i < gq[WOFDM_G_T]; i += blockDim.x) {
                    v2f t;
                    if constexpr (FIR8) {
                        const int sy = min(i / B, S), off = i - sy * B;
                        const uint32_t *hw = Hp + 8 + 2 * BR * sy + off;
                        t = join_h(hw[0], hw[sy == S ? VT : BR]);
                    } else {
                        t = join_h(Hp[PRE + i], Lp[PRE + i]);
                    }
                    t = t * p.dump_unscale_tx;
                    p.dump.tx[i] = make_float2(t.x, t.y);
                }
            __syncthreads();
        }
        const int jl = 8 * ln + 2 * lg;                 // the lane's samples of a tile: jl, jl + 1
        const int jw = s0 * B, LW = nreal * B;
        const bool all_full = !PART && !DUMP && LW == 128 * NT;  // every lane of every tile owns two samples
        // The beta + L - 1 trailing samples of the frame only feed the power sums.  One symbol per wave (layouts 8, 12): the last
        // symbol's last tile is a quarter full (544 = 4.25 x 128, 1056 = 8.25 x 128), and the samples behind the row's end are
        // exactly the ones the tile's next columns would compute -- their operand words sit in the virtual row behind the last
        // symbol, their noise pairs continue the row's Philox blocks.  So the LAST wave takes them along in its last tile (round
        // 4): before, wave 0 ran a tile of its own for them, 31 % / 39 % of its time at N = 512 / 1024, while the other fifteen
        // waves waited 22 % of theirs at barrier 2 (profiles/r03_stamp_report.txt).  Needs rows of whole 16-byte operand words and an
        // even number of trailing samples (a lane's two samples of a tile count or do not count together).
        const int tail_total = NL - S * B;                  // beta+L-1 (MATLAB order) or 0
        const bool fold_tail = FIR8 && !FIR8M && WOFDM_FOLD_TAIL && tail_total > 0 && (tail_total & 1) == 0 && (B & 3) == 0 && LW - 128 * (NT - 1) + tail_total <= 128
                               && tail_total <= VT - 12;
        const bool fold_here = fold_tail && wv == W - 1;
        const int LWS = LW + (fold_here ? tail_total : 0);      // samples of this wave that count in the power sums
        const v2f zero2 = mk(0.f, 0.f);
        v2f pn2 = zero2, ps2 = zero2;
        // (the round keys are made opaque once per phase here: they may sit in SGPRs for the tile loop,
        // not for the whole frame loop)
        uint32_t key0 = seed_lo, key1 = seed_hi;
        asm volatile("" : "+s"(key0), "+s"(key1));
        auto noise_pair = [&](int j, bool v0, bool v1, v2f &n0, v2f &n1) {   // samples j (even), j + 1
            if (INJECT) {
                const float2 *src = p.unit_noise + inj * NL + j;
                n0 = zero2; n1 = zero2;
                if (v0 && v1 && ((inj * NL + j) & 1) == 0) {
                    const float4 t = *reinterpret_cast<const float4 *>(src);
                    n0 = mk(t.x, t.y); n1 = mk(t.z, t.w);
                } else {
                    if (v0) n0 = ldg2(src);
                    if (v1) n1 = ldg2(src + 1);
                }
            } else if (j & 1) {
                // odd strides with one symbol per wave: the rows of the odd symbols start on an odd sample of the frame, and a
                // lane's two samples of a tile are the SECOND half of one Philox block and the FIRST half of the next (the
                // streams are keyed by the frame's sample index, not by the layout): two blocks per lane in those rows
                const philox_out oa = stream_block<false>((uint32_t)(j - 1) >> 1, f_lo, f_hi,
                                                          (WOFDM_STREAM_NOISE << 28) | cell, key0, key1);
                const philox_out ob = stream_block<false>((uint32_t)(j + 1) >> 1, f_lo, f_hi,
                                                          (WOFDM_STREAM_NOISE << 28) | cell, key0, key1);
                n0 = box_muller<false>(oa.w[2], oa.w[3]);
                n1 = box_muller<false>(ob.w[0], ob.w[1]);
            } else {
                const philox_out o = stream_block<false>((uint32_t)j >> 1, f_lo, f_hi,
                                                         (WOFDM_STREAM_NOISE << 28) | cell, key0, key1);
                n0 = box_muller<false>(o.w[0], o.w[1]);
                n1 = box_muller<false>(o.w[2], o.w[3]);
            }
        };
        // (generated noise is short of sqrt(2 ln 2), see box_muller: only the dumps care)
        const float nuns = INJECT ? 1.0f : WOFDM_NOISE_UNSCALE;
        struct bops { h8 h0, h1, l0, l1; };
        auto ld16 = [](const uint32_t *q) { return __builtin_bit_cast(h8, *reinterpret_cast<const u4 *>(q)); };
        // operand rows of the tile whose first output is sample `first` of the frame (wave-uniform):
        // four 16-byte rows = samples first - 24 + 4 lg + 16 half + (0..3) of column ln, both planes
        auto fir_load = [&](int first, int G, bool trailing) -> bops {
            bops o;
            if constexpr (FIR8) {
                // the wave's own symbol row, or the virtual row S for the trailing tile.  Window samples
                // in front of the row (tile 0 only) are the last ones of the previous symbol's rows, or
                // the zero words in front of symbol 0.
                const int sy = trailing ? S : s0;
                const uint32_t *rh = Hp + 8 + 2 * BR * sy;
                const int loff = sy == S ? VT : BR;
                // (columns without a sample read the last column that has one: the trailing tile stays
                // inside the short virtual row, the wave's last tile inside the frame)
                int lc = ln;
                if (trailing) lc = min(ln, 5);
                else if (G == NT - 1) lc = min(ln, max(0, (LWS - 128 * (NT - 1) + 7) / 8 - 1));
#ifdef WOFDM_PROBE_FIRREAD      // timing probe (results wrong): the operand rows of a quarter 16 bytes apart -- no bank conflicts
                const int q0 = 128 * G + 4 * lc + 64 * (lg & 1) - 24;
#else
                const int q0 = 128 * G + 8 * lc + 4 * lg - 24;
#endif
                const uint32_t *ph0 = rh + q0, *pl0 = rh + loff + q0, *ph1 = ph0 + 16, *pl1 = pl0 + 16;
                if (G == NT - 1 && !trailing && fold_here) {
                    // words behind the row's end come from the virtual row S: plane H sits B words behind this row's position
                    // of the same index (this row's plane L lies between), plane L another VT
                    if (q0 >= B) { ph0 += 2 * BR - B; pl0 += BR - B + VT; }
                    if (q0 + 16 >= B) { ph1 += 2 * BR - B; pl1 += BR - B + VT; }
                }
                if (G == 0) {
                    const bool z = sy == 0;
                    // (the previous row's last samples: its plane H ends B words behind its start, 2 BR words in front of this row)
                    if (q0 < 0) { ph0 = z ? Hp : rh + q0 + B - 2 * BR; pl0 = z ? Hp : rh + q0 + B - BR; }
                    if (q0 + 16 < 0) { ph1 = z ? Hp : rh + q0 + 16 + B - 2 * BR; pl1 = z ? Hp : rh + q0 + 16 + B - BR; }
                }
                o.h0 = ld16(ph0); o.h1 = ld16(ph1); o.l0 = ld16(pl0); o.l1 = ld16(pl1);
                // (layout 9: a short row -- N < 512 -- can end in ANY tile, and the rows behind it may still hold fp32 data)
                if ((FIR8M || G == NT - 1) && !trailing && !fold_here) {
                    // The block that holds the symbol's last samples may reach up to 7 samples past
                    // them (B is even, not a multiple of 8).  The Toeplitz entries that meet those are
                    // zero, but the words there are another row's (possibly the next wave's scratch:
                    // any bit pattern, NaN included) -- zero them, word by word where a 16-byte row straddles the end
                    // (B = 2 mod 4).
                    auto clip = [&](h8 &hh, h8 &ll, int q) {
                        const int r = B - q;                              // words of this row that are the symbol's
                        if (r < 4) {
                            u4 a = __builtin_bit_cast(u4, hh), c = __builtin_bit_cast(u4, ll);
                            if (r < 1) { a.x = 0u; c.x = 0u; }
                            if (r < 2) { a.y = 0u; c.y = 0u; }
                            if (r < 3) { a.z = 0u; c.z = 0u; }
                            a.w = 0u; c.w = 0u;
                            hh = __builtin_bit_cast(h8, a); ll = __builtin_bit_cast(h8, c);
                        }
                    };
                    clip(o.h0, o.l0, q0);
                    clip(o.h1, o.l1, q0 + 16);
                }
            } else {
                // plane-H word of x[j - 24] is Hp[PRE + j - 24] = Hp[j]
                const uint32_t *b = Hp + first + 128 * G + 8 * ln + 4 * lg;
                o.h0 = ld16(b); o.h1 = ld16(b + 16); o.l0 = ld16(b + plen); o.l1 = ld16(b + plen + 16);
                if ((PART || G == NT - 1) && !trailing && !all_full) {
                    // odd strides: the block with the wave's last samples reaches 4 samples into the next
                    // wave's rows, which may still hold that wave's scratch (0 x NaN): zero those words
                    // (layout 16: a wave's samples can end in any tile, at any stride)
                    const int q0 = 128 * G + 8 * ln + 4 * lg - 24;
                    const h8 zr = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (q0 >= LW) { o.h0 = zr; o.l0 = zr; }
                    if (q0 + 16 >= LW) { o.h1 = zr; o.l1 = zr; }
                }
            }
            return o;
        };
        // The six MFMAs of a tile as ONE in-place accumulation chain (vDst = SrcC), written as a single asm block:
        //  * back to back: a gap of 7 or more wait states in front of one of the later MFMAs of the run corrupts packed
        //    op_sel arithmetic of the SIMD's other waves (see WOFDM_MMA_ALIGN above) -- left to the compiler (builtins) the
        //    chain was interleaved with the noise draw's VALU work, and round 2's "sporadically wrong frames" were that;
        //  * the destination early-clobber by hand ("=&v"): this MFMA carries no such constraint in the compiler, which
        //    then puts a destination on top of the instruction's own B operand;
        //  * the leading s_nop covers a VALU write of an operand just in front, the trailing ones the 12 wait states a
        //    VALU read of the result needs (not interlocked: tools/ubench/mfma_gap.hip).
        auto fir_mma = [&](const bops &o) -> f4 {
            f4 d;
            if constexpr (MPIPE) {
                // layouts 10 / 11 / 12 hold nothing a gap in this chain could corrupt (see mma33): compiler builtins, spread
                // over the tile's noise draw (below)
                const f4 z = {0.f, 0.f, 0.f, 0.f};
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[1], o.h0, z, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[3], o.h1, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[0], o.l0, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[2], o.l1, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[0], o.h0, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[2], o.h1, d, 0, 0, 0);
                asm volatile(WOFDM_MMA_TAIL : "+v"(d) : "v"(o.h0), "v"(o.h1), "v"(o.l0), "v"(o.l1), "v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]));
                return d;
            }
            asm volatile(".p2align " WOFDM_MMA_ALIGN "\n\t"
                         "s_nop 1\n\t"
                         "v_mfma_f32_16x16x32_f16 %0, %1, %5, 0\n\t"        // h_lo x_hi
                         "v_mfma_f32_16x16x32_f16 %0, %2, %6, %0\n\t"
                         "v_mfma_f32_16x16x32_f16 %0, %3, %7, %0\n\t"       // h_hi x_lo
                         "v_mfma_f32_16x16x32_f16 %0, %4, %8, %0\n\t"
                         "v_mfma_f32_16x16x32_f16 %0, %3, %5, %0\n\t"       // h_hi x_hi
                         "v_mfma_f32_16x16x32_f16 %0, %4, %6, %0\n\t"
                         "s_nop 7\n\ts_nop 3"
                         : "=&v"(d)
                         : "v"(A[1]), "v"(A[3]), "v"(A[0]), "v"(A[2]), "v"(o.h0), "v"(o.h1), "v"(o.l0), "v"(o.l1));
            return d;
        };
        // (large DFTs: the unit noise of the tiles from NKEEP on is parked in HBM scratch; the first NKEEP tiles' stays in
        // registers -- as many as the 128-VGPR budget allows: what is parked has to FIT the XCD's 4 MB of L2 together with
        // the other 31 workgroups' rows, or every access of the cyclic write / read-back pattern misses; with all nine tiles
        // parked (4.7 MB per XCD) the N = 1024 kernel drew 250 W more, ran at the 1 400 W cap and 5 % slower)
        constexpr int NKEEP = WOFDM_NOISE_KEEP_TILES;
        f4 *nscr = nullptr;
        if constexpr (RENOISE)
            nscr = reinterpret_cast<f4 *>(p.noise_scratch) + ((size_t)blockIdx.x * 16 + wv) * (NT * 64) + lane;
        // (two instantiations of the tile loop: with every lane of every tile in use -- C2 -- the
        // validity selects are not even emitted)
        auto tiles = [&](auto full_c, auto odd_c) {
        constexpr bool FULLT = decltype(full_c)::value;
        // (ODDB: an odd stride with one symbol per wave -- its own instantiation, so that the even strides carry none of it)
        constexpr bool ODDB = decltype(odd_c)::value;
        // (N = 1024 is at its 128-VGPR limit: no operand prefetch there, the rows are requested per tile)
        constexpr bool PREFETCH = true;
        bops bq;
        if constexpr (PREFETCH) bq = fir_load(jw, 0, false);
#pragma unroll
        for (int G = 0; G < NT; ++G) {
            const int jr = 128 * G + jl;
            if constexpr (PART) {
                // (tiles behind the wave's samples are skipped: two waves of eight symbols at N = 64, CP 32 fill six of ten)
                if (128 * G >= LWS) {
                    acc[2 * G] = acc[2 * G + 1] = zero2;
                    nz[2 * G] = nz[2 * G + 1] = zero2;
                    continue;
                }
            }
            // (a wave holds at least LW_MIN = SPW N samples: the tiles below that are full in every geometry)
            bool valid = FULLT || jr < LWS;
            if constexpr (LW_MIN > 0) valid = valid || 128 * (G + 1) <= LW_MIN;
            // (odd strides: the wave's last pair holds ONE sample of its row -- the second one is the next row's first)
            bool valid1 = valid;
            if constexpr (ODDB) valid1 = valid && (jr + 1 < LWS || ((LW_MIN > 0) && 128 * (G + 1) <= LW_MIN));
            // (layout 16: an odd number of symbols at an odd stride ends the wave on an odd sample -- the pair's second one is the first
            // of the frame's trailing samples, which wave 0's trailing tile counts)
            if constexpr (PART) valid1 = valid && jr + 1 < LWS;
            v2f n0, n1;
            f4 d;
            // (not at N = 1024: that kernel sits at its 128-register limit, two tiles' operand rows alive at once spill, and the
            // compiler's own order is 1.5 % faster there -- interleaved A/B, profiles/r04_other_configs.txt)
            // (odd strides take the compiler's order: the pipeline with both Philox blocks of an odd row between the MFMAs was built
            // and measured -- CPW N = 512 3.55 against 3.63e8, nothing gained)
            if constexpr (MPIPE && !INJECT && WOFDM_TILE_PIPELINE && N < 1024 && !ODDB) {
                // The tile as a hand-placed pipeline (round 4).  A wave issues a DEPENDENT vector instruction every 8.3 cycles
                // at best and an independent one every 4.3 (tools/ubench/valu_dep.hip); the six MFMAs of the chain are dependent
                // (16 cycles apart), their operand rows take an LDS round trip, and a vector instruction must neither read the
                // chain's result nor WRITE one of its operand registers for 12 wait states behind the last MFMA (mma33).  Left
                // to the compiler the tile was: six MFMAs back to back, twelve idle wait states, THEN the noise draw, with the
                // next tile's rows requested ten instructions ahead of their use -- 499 cycles per tile for a wave on its own,
                // 274 of them vector issue (profiles/r04_stamp_occ.txt).  Here the Philox rounds of the tile's noise pair sit
                // BETWEEN the MFMAs (two rounds = eight instructions = 34 cycles per gap: the chain never waits), the Box-Muller
                // transform (18 instructions) behind the last one takes the place of the wait states, the guard behind it keeps
                // the operands allocated up to there without spending a cycle, and the next tile's rows are requested a whole
                // tile ahead.  tests/test_code_layout.py checks the distances in the built code.
                bops o = bq;
                if (G + 1 < NT) bq = fir_load(jw, G + 1, false);
                uint32_t c0 = (uint32_t)(jw + jr) >> 1, c1 = f_lo, c2 = f_hi, c3 = (WOFDM_STREAM_NOISE << 28) | cell;
                uint32_t k0 = key0, k1 = key1;
                auto rounds2 = [&]() {
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
                        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
                        const uint32_t m0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
                        const uint32_t m2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
                        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = m0; c2 = m2;
                        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
                    }
                };
                // The order is pinned by DEPENDENCIES, not by scheduling hints: an MFMA builtin and the Philox arithmetic are pure, so
                // every IR pass may move them across a sched_barrier (an instrumented instantiation came out as six MFMAs back
                // to back with the noise draw behind the guard -- and an operand register overwritten two instructions behind
                // its MFMA: tests/test_code_layout.py caught it).  Each TIE is an empty volatile asm statement through which
                // the chain's accumulator AND the Philox words pass: what produces its inputs comes before it, what consumes
                // its outputs after it, and volatile statements keep their order.
                asm volatile("" : "+v"(o.h0), "+v"(c0));                                   // (the tile starts here)
#define WOFDM_TIE() asm volatile("" : "+v"(d), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3))
                const f4 z = {0.f, 0.f, 0.f, 0.f};
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[1], o.h0, z, 0, 0, 0);
                rounds2();
                WOFDM_TIE();
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[3], o.h1, d, 0, 0, 0);
                rounds2();
                WOFDM_TIE();
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[0], o.l0, d, 0, 0, 0);
                rounds2();
                WOFDM_TIE();
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[2], o.l1, d, 0, 0, 0);
                rounds2();
                WOFDM_TIE();
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[0], o.h0, d, 0, 0, 0);
                rounds2();
                WOFDM_TIE();
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[2], o.h1, d, 0, 0, 0);
                WOFDM_TIE();                      // the Box-Muller transform starts BEHIND the last MFMA ...
#undef WOFDM_TIE
                n0 = box_muller<false>(c0, c1);
                n1 = box_muller<false>(c2, c3);
                // ... and ends in front of the guard: 18 vector instructions (two of them per 16 cycles at best) between the last
                // MFMA and the first instruction that may read its result or write one of its operand registers
                asm volatile("" : "+v"(d), "+v"(n0), "+v"(n1) : "v"(o.h0), "v"(o.h1), "v"(o.l0), "v"(o.l1), "v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]));
            } else {
            // the six MFMAs go first and run on the matrix pipe under the noise draw of the same tile,
            // the next tile's operand rows are requested in between
            if constexpr (!PREFETCH) bq = fir_load(jw, G, false);
            const bops bcur = bq;
            d = fir_mma(bcur);
            if constexpr (PREFETCH) {
                if (G + 1 < NT) bq = fir_load(jw, G + 1, false);
            }
            noise_pair(jw + jr, valid, valid1, n0, n1);
            if constexpr (MPIPE && !INJECT) {
                // the tile's six MFMAs spread over its noise draw: one MFMA, then four vector instructions (an MFMA holds
                // the vector issue for half of its 16 cycles; interleaved A/B of 3, 4, 5, 6, 8: all within 1 %, -2.3 %
                // against the compiler's own placement)
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
            }
            }
            const v2f c0 = mk(d.x, d.y), c1 = mk(d.z, d.w);
            if (RENOISE && (INJECT || G >= NKEEP)) {
                if (!INJECT) nscr[64 * G] = (f4){n0.x, n0.y, n1.x, n1.y};
            } else {
                nz[2 * G] = n0; nz[2 * G + 1] = n1;
            }
            acc[2 * G] = c0; acc[2 * G + 1] = c1;
            // (selects, not branches: columns behind the wave's samples may hold anything)
            const v2f a0 = valid ? c0 : zero2, a1 = valid1 ? c1 : zero2;
            const v2f m0 = valid ? n0 : zero2, m1 = valid1 ? n1 : zero2;
            ps2 = __builtin_elementwise_fma(a0, a0, ps2);
            ps2 = __builtin_elementwise_fma(a1, a1, ps2);
            pn2 = __builtin_elementwise_fma(m0, m0, pn2);
            pn2 = __builtin_elementwise_fma(m1, m1, pn2);
            if (DUMP) {
                const v2f e0 = c0 * p.dump_unscale_rx, e1 = c1 * p.dump_unscale_rx;
                if (p.dump.conv) {
                    float2 *dc = valid ? p.dump.conv + jw + jr : p.dump.sink;
                    dc[0] = make_float2(e0.x, e0.y);
                    *(valid1 ? dc + 1 : p.dump.sink) = make_float2(e1.x, e1.y);
                }
                if (p.dump.unit_noise) {
                    float2 *dn = valid ? p.dump.unit_noise + jw + jr : p.dump.sink;
                    dn[0] = make_float2(n0.x * nuns, n0.y * nuns);
                    *(valid1 ? dn + 1 : p.dump.sink) = make_float2(n1.x * nuns, n1.y * nuns);
                }
            }
        }
        };
        DELAY_AT(4);
        STAMPF(11);
        WAVE_PRIO(WOFDM_PRIO_TILES);
        if constexpr (FIR8) {
            if (B & 1) tiles(std::false_type{}, std::true_type{});
            else if (all_full) tiles(std::true_type{}, std::false_type{});
            else tiles(std::false_type{}, std::false_type{});
        } else {
            if (all_full) tiles(std::true_type{}, std::false_type{});
            else tiles(std::false_type{}, std::false_type{});
        }
        WAVE_PRIO(WOFDM_PRIO_B3);
        STAMPF(12);
        DELAY_AT(5);
        if (tail_total > 0 && wv == 0 && !fold_tail) {
            // the trailing samples of the frame (they only feed the power sums): one more tile, by
            // wave 0, the wave that reaches barrier 2 first; it reads behind the last wave's symbols
            if constexpr (RELAX) {
                if (W > 1) wait_flag(&flags[W - 1], iter, &flags[20]);
            }
            if constexpr (FIR8M && RELAXM) {
                if (W > 1) wait_flag(&flags[48 + W - 1], iter, &flags[20]);       // (the last row and its virtual row are planes)
            }
            const int jt = S * B;
            const bool v0 = jl < tail_total, v1 = jl + 1 < tail_total;
            v2f n0, n1;
            noise_pair(jt + jl, v0, v1, n0, n1);
            const f4 d = fir_mma(fir_load(jt, 0, true));
            const v2f c0 = mk(d.x, d.y), c1 = mk(d.z, d.w);
            const v2f a0 = v0 ? c0 : zero2, a1 = v1 ? c1 : zero2;
            const v2f m0 = v0 ? n0 : zero2, m1 = v1 ? n1 : zero2;
            ps2 = __builtin_elementwise_fma(a0, a0, ps2);
            ps2 = __builtin_elementwise_fma(a1, a1, ps2);
            pn2 = __builtin_elementwise_fma(m0, m0, pn2);
            pn2 = __builtin_elementwise_fma(m1, m1, pn2);
            if (DUMP) {
                const v2f e0 = c0 * p.dump_unscale_rx, e1 = c1 * p.dump_unscale_rx;
                if (p.dump.conv) {
                    *(v0 ? p.dump.conv + jt + jl : p.dump.sink) = make_float2(e0.x, e0.y);
                    *(v1 ? p.dump.conv + jt + jl + 1 : p.dump.sink) = make_float2(e1.x, e1.y);
                }
                if (p.dump.unit_noise) {
                    *(v0 ? p.dump.unit_noise + jt + jl : p.dump.sink) = make_float2(n0.x * nuns, n0.y * nuns);
                    *(v1 ? p.dump.unit_noise + jt + jl + 1 : p.dump.sink) = make_float2(n1.x * nuns, n1.y * nuns);
                }
            }
        }
        float ps = ps2.x + ps2.y, pn = pn2.x + pn2.y;
        ps = wave_sum(ps); pn = wave_sum(pn);
        if (lane == 0) { sums_it[wv] = ps; sums_it[16 + wv] = pn; }
        if constexpr (RENOISE) {
            // the parked noise comes back HERE, after the FIR's registers have died: the HBM/L2
            // latency of the reload runs under the wait at barrier 2 instead of opening phase C
#pragma unroll
            for (int G = 0; G < NT; ++G) {
                if (INJECT) {
                    bool valid = 128 * G + jl < LW;
                    if constexpr (LW_MIN > 0) valid = valid || 128 * (G + 1) <= LW_MIN;
                    noise_pair(jw + 128 * G + jl, valid, valid, nz[2 * G], nz[2 * G + 1]);
                } else if (G >= NKEEP) {
                    const f4 t = nscr[64 * G];
                    nz[2 * G] = mk(t.x, t.y); nz[2 * G + 1] = mk(t.z, t.w);
                }
            }
        }
        } else {
        GEO_PHASE();
        const int S = gq[WOFDM_G_S], B = gq[WOFDM_G_B], beta = gq[WOFDM_G_BETA], NL = gq[WOFDM_G_NL];
        const int W = S / SPW;                         // waves per workgroup
        v2f *fbw = fbuf + (LT - 1) + s0 * B;
        TAILS();
#pragma unroll
        for (int u = 0; u < SPW; ++u) {
            const int s = s0 + u;
            v2f *fb = fbw + u * B;
            if (s > 0 && lane < beta) fb[lane] = fb[lane] + tailb[(s - 1) * beta + lane];
        }
        wave_sync();
        if (DUMP) {
            __syncthreads();
            if (p.dump.tx)
                for (int i = tid; i < gq[WOFDM_G_T]; i += blockDim.x)
                    p.dump.tx[i] = make_float2(fbuf[(LT - 1) + i].x, fbuf[(LT - 1) + i].y);
            __syncthreads();
        }

        // lane -> RB consecutive outputs of the wave's SPW*B samples; the beta+L-1 trailing
        // samples of the frame (power sums only) ride in the idle lanes of the last waves
        const int LW = SPW * B;
        const int nmain = (LW + RB - 1) / RB;
        const int tail_total = NL - S * B;                  // beta+L-1 (MATLAB order) or 0
        const int idle = 64 - nmain;
        const int ntc = (tail_total + RB - 1) / RB;
        const bool tail_in_idle = idle * W >= ntc;
        if constexpr (RELAXF) {
            // waves that carry trailing samples of the frame read behind the last symbol (whose row, in the Tx-mask
            // variants, is complete once its predecessor has spilled onto it)
            if (tail_in_idle && wv != W - 1 && (W - 1 - wv) * idle < ntc) {
                wait_flag(&flags[W - 1], iter, &flags[20]);
                if constexpr (RELAXM) {
                    if (W > 1) wait_flag(&flags[W - 2], iter, &flags[20]);
                }
            }
        }
        is_main = lane < nmain;
        if (is_main) {
            j0 = s0 * B + lane * RB;
            cnt = min(RB, LW - lane * RB);
        } else if (tail_in_idle) {
            const int c = (W - 1 - wv) * idle + (lane - nmain);
            if (c < ntc) { j0 = S * B + c * RB; cnt = min(RB, tail_total - c * RB); }
        }

        // unit noise of the same samples first (its Philox keys and the FIR's 42 tap scalars
        // would otherwise fight over the SGPR file)
        v2f nzB[RB];
        make_noise(nzB, NL);
        v2f pn2 = mk(0.f, 0.f);                          // (sum re^2, sum im^2) of the noise
        // (instrumented builds: the per-lane dumps are unconditional stores with the address of
        // lanes without a sample redirected to a sink word -- no divergent regions inside these
        // register-hungry loops)
        // every lane owns exactly RB samples (e.g. four symbols of 288 samples on 64 x 18): the
        // r < cnt predicates -- RB SGPR pairs kept alive across the FIR -- are not needed at all
        const bool all_full = QW && !DUMP && LW == 64 * RB;
        if (all_full) {
#pragma unroll
            for (int r = 0; r < RB; ++r) pn2 = __builtin_elementwise_fma(nzB[r], nzB[r], pn2);
        } else {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                if (r < cnt) pn2 = __builtin_elementwise_fma(nzB[r], nzB[r], pn2);
                if (DUMP && p.dump.unit_noise)
                    *(r < cnt ? p.dump.unit_noise + j0 + r : p.dump.sink) = make_float2(nzB[r].x, nzB[r].y);
            }
        }
        if constexpr (!RENOISE) {
#pragma unroll
            for (int r = 0; r < RB; ++r) nz[r] = nzB[r];
        } else if (!INJECT) {
            v2f *ns = reinterpret_cast<v2f *>(p.noise_scratch)
                      + ((size_t)blockIdx.x * 16 + wv) * (RB * 64) + lane;
#pragma unroll
            for (int r = 0; r < RB; ++r) ns[r * 64] = nzB[r];
        }
        // The tap pointer is made opaque HERE so that the 42 scalar tap loads are issued after
        // barrier 1 and die with the FIR.
        int ch_now = __builtin_amdgcn_readfirstlane(ch);
        asm volatile("" : "+s"(ch_now));
        const v2f *__restrict__ taps = g_h + ch_now * LT;
        fir_lane<RB, fir_geo<N, LAY>::CH>(fbuf + j0, taps, acc);   // fbuf + (LT-1) + j0 - (LT-1)

        v2f ps2 = mk(0.f, 0.f);
        if (all_full) {
#pragma unroll
            for (int r = 0; r < RB; ++r) ps2 = __builtin_elementwise_fma(acc[r], acc[r], ps2);
        } else {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                if (r < cnt) ps2 = __builtin_elementwise_fma(acc[r], acc[r], ps2);
                if (DUMP && p.dump.conv)
                    *(r < cnt ? p.dump.conv + j0 + r : p.dump.sink) = make_float2(acc[r].x, acc[r].y);
            }
        }
        float ps = ps2.x + ps2.y, pn = pn2.x + pn2.y;
        if (!tail_in_idle && tail_total > 0 && wv == 0) {
            // no idle lanes: the trailing samples (they only feed the power sums) go to wave 0, the
            // wave that reaches barrier 2 first; it needs the last wave's symbols for them
            if constexpr (RELAXF) {
                if (W > 1) wait_flag(&flags[W - 1], iter, &flags[20]);
                if constexpr (RELAXM) {
                    if (W > 2) wait_flag(&flags[W - 2], iter, &flags[20]);
                }
            }
            for (int t = lane; t < tail_total; t += 64) {
                const int j = S * B + t;
                v2f c;
                fir_chunk<1>(fbuf + j, taps, &c);      // all 21 window loads in flight at once
                v2f nn;
                if (INJECT) {
                    nn = ldg2(p.unit_noise + inj * NL + j);
                } else {
                    const philox_out o = stream_block((uint32_t)j >> 1, f_lo, f_hi,
                                                      (WOFDM_STREAM_NOISE << 28) | cell, seed_lo, seed_hi);
                    nn = (j & 1) ? box_muller(o.w[2], o.w[3]) : box_muller(o.w[0], o.w[1]);
                }
                ps += c.x * c.x + c.y * c.y;
                pn += nn.x * nn.x + nn.y * nn.y;
                if (DUMP) {
                    if (p.dump.conv) p.dump.conv[j] = make_float2(c.x, c.y);
                    if (p.dump.unit_noise) p.dump.unit_noise[j] = make_float2(nn.x, nn.y);
                }
            }
        }
        ps = wave_sum(ps); pn = wave_sum(pn);
        if (lane == 0) { sums_it[wv] = ps; sums_it[16 + wv] = pn; }
        if constexpr (RENOISE) {
            // the parked noise comes back HERE, after the FIR's registers have died: the HBM/L2
            // latency of the reload runs under the wait at barrier 2 instead of opening phase C
            if (INJECT) {
                make_noise(nz, NL);
            } else {
                const v2f *ns = reinterpret_cast<const v2f *>(p.noise_scratch)
                                + ((size_t)blockIdx.x * 16 + wv) * (RB * 64) + lane;
#pragma unroll
                for (int r = 0; r < RB; ++r) nz[r] = ns[r * 64];
            }
        }
        }
        DELAY_AT(6);
        // Phase C's structure lengths are requested HERE, in front of the barrier: the scalar loads' latency runs under
        // the wait instead of opening the phase, where the wave has nothing else to issue (a stamped build showed the
        // start of phase C -- these loads, then a loop of dependent LDS reads over the waves' partial sums -- taking
        // 8.6 % of a wave's time for two dozen instructions).
        int goffc_ = 0;
        asm volatile("" : "+s"(goffc_));
        const int *__restrict__ gqc = gm + goffc_;
        const int cS = gqc[WOFDM_G_S], cB = gqc[WOFDM_G_B], cDelta = gqc[WOFDM_G_DELTA], cGam = gqc[WOFDM_G_GAMMA];
        const int cPlen = FIRQ ? gqc[WOFDM_G_FBUF] : 0;
        STAMP(2);
        __syncthreads();                                                     // ---- barrier 2
        STAMP(3);
        WAVE_PRIO(WOFDM_PRIO_C);
        DELAY_AT(7);

        // ------------------------------------------------------------ C: noise scale, Rx, FFT
        if constexpr (RELAUNDER) asm volatile("" : "+v"(lane));
        {
        const int S = cS, B = cB, delta = cDelta, gam = cGam, kap = 0;
        const int BR = (FIR8 && !FIR8M) ? ((B + 3) & ~3) : B;          // (LDS row stride per plane, see phase A)
        (void)BR;
        const int plen = cPlen;
        (void)S;
        v2f *fbw = FIR8 ? reinterpret_cast<v2f *>(Hp + 8 + 2 * BR * s0) : fbuf + (LT - 1) + s0 * B;
        auto row = [&](int u) -> v2f * {
            if constexpr (PART)
                return reinterpret_cast<v2f *>(Hp + (u < SPWR2 ? 0 : plen) + PRE + s0 * B) + (u < SPWR2 ? u : u - SPWR2) * B;
            else if constexpr (FIRQ)
                return reinterpret_cast<v2f *>(Hp + (u < SPW / 2 ? 0 : plen) + PRE + s0 * B) + (u % (SPW / 2)) * B;
            else
                return fbw + u * B;
        };
        const int nreal = PART ? min(SPWR, S - s0) : SPW;        // (layout 16: the symbol slots this wave fills)
        (void)nreal;
        // Swizzled rows of received samples (layouts 10 / 11 / 12; see the noise-scaling stores below).  The swizzle permutes the
        // four 16-byte units of every aligned group of eight samples of a row (one symbol per wave: B samples) or of a plane's chunk
        // (four symbols per wave: 2B samples); where the LAST group is incomplete -- B not a multiple of 8 resp. 4: half of the
        // reference's even cyclic prefixes, and every odd stride -- its samples keep their order (rx_part: writer and readers test
        // for it in instantiations / branches of their own, the strides of C2, C4 and C5 carry none of it).
        // (Layout 15 -- the Tx-mask kernel, whose time goes into the mask stage -- measured nothing with it, -1 % at strides with an
        // incomplete group: it keeps its rows in order.  profiles/r04_rx_swizzle_ab.txt)
        constexpr bool RXSWZ = (MDFT || MD8) && WOFDM_RX_SWIZZLE;
        constexpr int rx_sw = RXSWZ ? 6 : 0;
        const int rx_len = FIR8 ? B : 2 * B, rx_xb = rx_len & ~7;
        const bool rx_part = RXSWZ && (rx_len & 7) != 0;
        (void)rx_xb; (void)rx_part;
        auto rx_swz = [&](int x) { return x ^ ((((x >> 4) & 2) | ((x >> 2) & 4)) & rx_sw); };      // sample index -> position in the row
        (void)rx_swz;
        // total powers: ONE LDS round trip -- lane l reads partial sum l & 31 (signal powers of the waves in 0..15, noise
        // powers in 16..31; entries of waves the frame does not have stay zero), rows 0 and 1 of the wave add up
        float Ps, Pn;
        {
            const float part = row_sum(sums_it[lane & 31]);
            Ps = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part), 0));
            Pn = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part), 16));
        }
        const float g = __builtin_amdgcn_sqrtf(Ps * nlin * __builtin_amdgcn_rcpf(Pn));   // lengths cancel (m:289-292)
        mdft_early dce;
        if constexpr (MDFT || MDX) dce = mdft_request();
#ifdef WOFDM_AUDIT
        aud_g = g; aud_ps = Ps; aud_pn = Pn;
#endif
        if constexpr (FIRM) {
            // r = c + g n (m:292-293) as two-sample rows into the wave's private rows.  Quarter-wave
            // layouts: samples [0, 2B) of the wave sit in its chunk of plane H, [2B, 4B) in that of plane L
            const int jl = 8 * (lane & 15) + 2 * (lane >> 4), LW = nreal * B;
            const bool all_full = !PART && !DUMP && LW == 128 * NT;
            // The received samples go into the rows SWIZZLED (round 4, layouts 10 / 11 / 12): lane (ln, lg) holds the 16-byte unit 4 ln + lg
            // of a tile, and a ds_write_b128 is served in groups of eight consecutive lanes against 32 banks (MI355X_MICROARCH.md, LDS):
            // eight units 64 bytes apart sit on two 16-byte slots of the 128-byte bank row -- four-way conflicts, 32 LDS cycles per
            // store where 8 do, all of this phase's conflict cycles, and ON the critical path (a timing probe with conflict-free
            // store addresses: C2 -8 %, N = 512 -5 %, N = 1024 -7 %, profiles/r04_lds_conflict_probe.txt).  Sample x of a row (or of a
            // plane's chunk) is kept at rx_swz(x): bit 1 of the index XORed with its bit 5, bit 2 with its bit 4 -- the unit's two
            // low index bits with bits 4 and 3 of the unit index.  For the writer that is lg ^ (bits 2 and 1 of ln, swapped), a
            // per-lane constant folded into its two bases (in the plane-L chunk, which starts 2B samples in, the pair's index there
            // modulo a tile takes the place of jl): the eight lanes of a group hit the eight slots of a bank row.  The readers undo it in their per-lane bases
            // (below): their elements lie multiples of 64 samples apart, which leaves bits 4..5 of the index alone.  (Which bits
            // go where is chosen for the READERS: with this pairing layout 12's 32-byte-strided ds_read_b128 at N = 1024 -- 2-way
            // conflicts before -- are conflict-free as well, by the bank rule and lane groups of MI355X_MICROARCH.md; the pairing
            // bit 1 <-> 4, 2 <-> 5 serves the writer alike and leaves those at 2-way.)
            int jlH = jl, jlL = jl;
            if constexpr (RXSWZ) {
                jlH = rx_swz(jl);
                const int m = (jl - 2 * B) & 127;
                jlL = jl + (rx_swz(m) - m);
            }
#ifdef WOFDM_PROBE_RXWRITE      // timing probe (results wrong): the stores of a quarter 16 bytes apart -- no bank conflicts
            jlH = jlL = 2 * (lane & 15) + 32 * (lane >> 4);
#endif
            v2f *rxb0_ = FIR8 ? fbw : reinterpret_cast<v2f *>(Hp + PRE + s0 * B);
            v2f *rxb = rxb0_ + jlH;
            const int dlt = FIR8 ? 0 : (plen - SPWR * B) / 2;
            v2f *rxb1 = rxb0_ + dlt + jlL;                          // the lane's base in the plane-L chunk (samples from 2B on)
            v2f *sink = reinterpret_cast<v2f *>(smem + L::off_flags + 4 * 24);   // 16 idle bytes
            // (two instantiations, as for the tile loop: with every lane of every tile in use -- C2 -- a store's address is
            // one select between the two per-lane bases plus an instruction immediate; a third and fourth where the row's or the
            // chunk's last group of eight samples is incomplete and keeps its order)
            auto noise_scale = [&](auto full_c, auto odd_c, auto part_c) {
            constexpr bool FULLC = decltype(full_c)::value;
            constexpr bool ODDB = decltype(odd_c)::value;        // (odd stride, one symbol per wave: see the tile loop)
            constexpr bool PARTG = decltype(part_c)::value && RXSWZ;
#pragma unroll
            for (int G = 0; G < NT; ++G) {
                const int jr = 128 * G + jl;
                if constexpr (PART) {
                    if (128 * G >= LW) continue;
                }
                bool valid = FULLC || jr < LW;
                if constexpr (LW_MIN > 0) valid = valid || 128 * (G + 1) <= LW_MIN;
                // (odd strides, one symbol per wave: the row's last pair holds one sample of the row; the 16 bytes would reach
                // into the next wave's row)
                const bool half = ODDB && jr == LW - 1;
                const v2f r0 = __builtin_elementwise_fma(mk(g, g), nz[2 * G], acc[2 * G]);
                const v2f r1 = __builtin_elementwise_fma(mk(g, g), nz[2 * G + 1], acc[2 * G + 1]);
                v2f *dst = (FIR8 || jl < SPWR2 * B - 128 * G ? rxb : rxb1) + 128 * G;
                if constexpr (PARTG) {
                    const bool idn = FIR8 ? jr >= rx_xb : ((jr >= rx_xb && jr < rx_len) || jr >= rx_len + rx_xb);
                    if (idn) dst = rxb0_ + (FIR8 || jr < rx_len ? 0 : dlt) + jr;
                }
                v2f *dst0 = dst;
                if (!FULLC) dst = (valid && !half) ? dst : sink;
                *reinterpret_cast<f4 *>(dst) = (f4){r0.x, r0.y, r1.x, r1.y};
                if constexpr (ODDB) {
                    if (128 * G < LW && LW <= 128 * (G + 1)) {                      // (wave-uniform: the tile with the row's end)
                        if (half) *dst0 = r0;
                    }
                }
                if (DUMP && p.dump.rx) {
                    float2 *dr = valid ? p.dump.rx + s0 * B + jr : p.dump.sink;
                    const v2f e0 = r0 * p.dump_unscale_rx, e1 = r1 * p.dump_unscale_rx;
                    dr[0] = make_float2(e0.x, e0.y);
                    *((valid && !half) ? dr + 1 : p.dump.sink) = make_float2(e1.x, e1.y);
                }
            }
            };
            if constexpr (FIR8) {
                if (B & 1) noise_scale(std::false_type{}, std::true_type{}, std::true_type{});
                else if (rx_part) noise_scale(std::false_type{}, std::false_type{}, std::true_type{});
                else if (all_full) noise_scale(std::true_type{}, std::false_type{}, std::false_type{});
                else noise_scale(std::false_type{}, std::false_type{}, std::false_type{});
            } else {
                if (rx_part) noise_scale(std::false_type{}, std::false_type{}, std::true_type{});
                else if (all_full) noise_scale(std::true_type{}, std::false_type{}, std::false_type{});
                else noise_scale(std::false_type{}, std::false_type{}, std::false_type{});
            }
        } else if constexpr (DUMP) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const bool live = is_main && r < cnt;
                const v2f y = __builtin_elementwise_fma(mk(g, g), nz[r], acc[r]);
                if (live) fbw[lane * RB + r] = y;
                if (p.dump.rx) *(live ? p.dump.rx + s0 * B + lane * RB + r : p.dump.sink) = make_float2(y.x, y.y);
            }
        } else if (QW && SPW * B == 64 * RB) {           // every lane owns exactly RB samples
#pragma unroll
            for (int r = 0; r < RB; ++r) fbw[lane * RB + r] = __builtin_elementwise_fma(mk(g, g), nz[r], acc[r]);
        } else if (is_main) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                if (r < cnt) fbw[lane * RB + r] = __builtin_elementwise_fma(mk(g, g), nz[r], acc[r]);
            }
        }
        // (undo the signal's scale and, where the generated noise is short of sqrt(2 ln 2), that factor)
        if (DUMP && p.dump.gain && tid == 0)
            p.dump.gain[0] = g * p.dump_unscale_rx * ((FIRM && !INJECT) ? 1.0f / WOFDM_NOISE_UNSCALE : 1.0f);
        wave_sync();
        STAMPC(13);

        // remove_redundancy, windowRx, overlap_and_add, circular_shift (m:302-308) collapse to
        // z[t] = sum_{m = t+kappa+delta/2 (mod N), m < N+delta} w_rx[m] y[gamma+m]
        const int h2 = delta >> 1;
        // The circular shift is left out, in the instrumented kernels as well: z'[t] = z[t - kappa - delta/2] turns into
        // the factor e^{-2 pi i (kappa + delta/2) n / N} on subcarrier n of EVERY symbol of the frame, the pilot included,
        // and the one-tap equaliser X^[s] = Y[s] X0 / Y0 (m:266-268) divides it out again.  Without it the lane's
        // elements sit at fixed offsets from one per-lane base (addresses as instruction immediates), and only the
        // first tail_rx <= 64 samples of a symbol can have a folded partner.  (The dumped Y is therefore the
        // reference's Y times that ramp -- the tests put it back on the host; Xhat and everything behind it is the
        // reference's.)  Two passes: all main-tap loads in flight together, free of branches; the folded samples only
        // where there is an Rx tail at all.
        if constexpr (MDS) {
            (void)kap; (void)h2;
            const int lm = lane & 15, lg = lane >> 4;
            const u4 *dg = reinterpret_cast<const u4 *>(p.dftc) + lane;
            const h8 brh = __builtin_bit_cast(h8, dg[0]), brl = __builtin_bit_cast(h8, dg[64]);
            const h8 bih = __builtin_bit_cast(h8, dg[128]), bil = __builtin_bit_cast(h8, dg[192]);
            f4 twr[SCS], twi[SCS];
#pragma unroll
            for (int pp = 0; pp < SCS; ++pp) {
                twr[pp] = __builtin_bit_cast(f4, dg[(8 + 2 * pp) * 64]);
                twi[pp] = __builtin_bit_cast(f4, dg[(9 + 2 * pp) * 64]);
            }
            h8 xh[4], xl[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int sr = 4 * (t / SCS) + (lm >> 2), c = 4 * (t % SCS) + (lm & 3);
                const v2f *fy = row((PART && sr >= nreal) ? 0 : sr) + gam;      // (layout 16: an unfilled slot reads slot 0's row)
                uint32_t h[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = SC * (lg + 4 * j) + c;
                    v2f x = wmul(fy[n], wrx[n]);
                    if (delta > 0) {
                        if (n < delta) x = wfma(fy[n + N], wrx[n + N], x);
                    }
                    split_h(x, h[j], l[j]);
                }
                xh[t] = __builtin_bit_cast(h8, (u4){h[0], h[1], h[2], h[3]});
                xl[t] = __builtin_bit_cast(h8, (u4){l[0], l[1], l[2], l[3]});
                mma_operand_fence(xh[t], xl[t]);
            }
            wave_sync();
            STAMPC(14);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                mma33(yr[t], yi[t], xl[t], brh, xh[t], brl, xh[t], brh, xl[t], bih, xh[t], bil, xh[t], bih);
                mdft_twiddle(yr[t], yi[t], twr[t % SCS], twi[t % SCS]);
            }
#pragma unroll
            for (int gr = 0; gr < SGR; ++gr) {
                if constexpr (SC == 4) radix4_elems(yr[gr], yi[gr]);
                else radix8_elems(yr[2 * gr], yi[2 * gr], yr[2 * gr + 1], yi[2 * gr + 1]);
            }
            STAMPC(15);
            if (DUMP && p.dump.Y) {
                const float us = p.dump_unscale_rx / p.rx_scale[sn * n_ch + ch];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int kc = SC == 4 ? e : 2 * e + (t % SCS);
                        if (PART && 4 * (t / SCS) + lg >= nreal) continue;
                        p.dump.Y[(s0 + 4 * (t / SCS) + lg) * N + lm + 16 * kc] = make_float2(yr[t][e] * us, yi[t][e] * us);
                    }
            }
            if (wv == 0) {
                // pilot = symbol 0 = group 0 of lanes g' = 0: G = X0 / Y0, one 16-byte row of real and one of imaginary parts per part
                f4 *G4 = reinterpret_cast<f4 *>(G);
                if (lg == 0) {
#pragma unroll
                    for (int pp = 0; pp < SCS; ++pp) {
                        f4 gr, gi;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const hpair hw = __builtin_bit_cast(hpair, qlw[(labo[pp] >> (8 * e)) & lmask]);
                            float x0r = (float)hw.y, x0i = (float)hw.x;
                            if constexpr (ALLOC) {
                                if ((labo[pp] >> (8 * e)) & 0x80u) { x0r = 0.f; x0i = 0.f; }      // G = 0 there
                            }
                            const float y0r = yr[pp][e], y0i = yi[pp][e];
                            const float inv = __builtin_amdgcn_rcpf(y0r * y0r + y0i * y0i);
                            gr[e] = (x0r * y0r + x0i * y0i) * inv;
                            gi[e] = (x0i * y0r - x0r * y0i) * inv;
                        }
                        G4[pp * 16 + lm] = gr;
                        G4[(SCS + pp) * 16 + lm] = gi;
                    }
                }
                wave_sync();
                if constexpr (RELAXF) post_flag(&flags[16], iter, lane);
            }
        } else if constexpr (MDX) {
            (void)kap; (void)h2;
            // Rx window / fold into INPUT element order (NC consecutive samples per element j), split, transform
            const int lb = lane & 15, lg = lane >> 4;
            const v2f *fy = row(0) + gam;
            v2f vin[NC][4];
            // (swizzled rows: the lane's NC consecutive samples are one or two 16-byte units, each at its own per-lane base -- with an
            // odd gamma, where they straddle units, single samples at NC bases; the elements j lie N / 4 = 128 or 256 samples apart,
            // which leaves the XOR term as it is: instruction offsets as before)
            auto rx_load = [&](auto single_c) {
                constexpr int UM = decltype(single_c)::value ? ~0 : ~1;             // a base per sample, or per pair of samples
                const v2f *fyc[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    fyc[c] = fy;
                    if constexpr (RXSWZ) {
                        const int o = (N / 16) * lg + NC * lb + (c & UM);
                        fyc[c] = row(0) + rx_swz(gam + o) - o;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n0 = (N / 16) * (lg + 4 * j) + NC * lb;
#ifdef WOFDM_PROBE_RXREAD       // timing probe (results wrong): 16-byte reads of a quarter 16 bytes apart -- no bank conflicts
#pragma unroll
                    for (int c = 0; c < NC; ++c) vin[c][j] = wmul(fy[(N / 16) * (lg + 4 * j) + 2 * lb + (c & 1) + 32 * (c >> 1)], wrx[n0 + c]);
#else
                    if constexpr (!decltype(single_c)::value && NC >= 2) {
                        // an even gamma: the pairs of samples are whole 16-byte words of the row, asked for as such -- left to the
                        // compiler, which sees 8-byte alignment only, they were ds_read2_b64 (half the rate of ds_read_b128, served
                        // against 32 banks in groups of sixteen consecutive lanes: four-way conflicts at this 32-byte lane stride)
                        f4 wq = {0.f, 0.f, 0.f, 0.f};
                        if constexpr (NC == 4) wq = *reinterpret_cast<const f4 *>(wrx + n0);
                        else { const v2f w2 = *reinterpret_cast<const v2f *>(wrx + n0); wq.x = w2.x; wq.y = w2.y; }
#pragma unroll
                        for (int c = 0; c < NC; c += 2) {
                            const f4 t = *reinterpret_cast<const f4 *>(fyc[c] + n0 + c);
                            vin[c][j] = wmul(mk(t.x, t.y), wq[c]);
                            vin[c + 1][j] = wmul(mk(t.z, t.w), wq[c + 1]);
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < NC; ++c) vin[c][j] = wmul(fyc[c][n0 + c], wrx[n0 + c]);
                    }
#endif
                }
                if (delta > 0) {
                    // (only element 0 of lane group 0 can have a folded partner: tail_rx <= 64 <= N/16)
                    const int n0 = (N / 16) * lg + NC * lb;
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        if (n0 + c < delta) {
                            vin[c][0] = wfma(fyc[c][n0 + c + N], wrx[n0 + c + N], vin[c][0]);
                        }
                }
                if constexpr (RXSWZ) {
                    if (rx_part) {
                        // the row's incomplete last group keeps its order: the elements in it -- element 3 of the last lanes, folded
                        // partners -- are read once more, where they are
                        const int n3 = (N / 16) * (lg + 12) + NC * lb;
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                            if (gam + n3 + c >= rx_xb) vin[c][3] = wmul(fy[n3 + c], wrx[n3 + c]);
                        if (delta > 0) {
                            const int n0 = (N / 16) * lg + NC * lb;
#pragma unroll
                            for (int c = 0; c < NC; ++c)
                                if (n0 + c < delta && gam + n0 + c + N >= rx_xb)
                                    vin[c][0] = wfma(fy[n0 + c + N], wrx[n0 + c + N], wmul(fyc[c][n0 + c], wrx[n0 + c]));
                        }
                    }
                }
            };
            if (NC == 1 || (gam & 1) != 0) rx_load(std::true_type{});
            else rx_load(std::false_type{});
            wave_sync();
            STAMPC(14);
            const mdft_consts dc = mdft_load(dce);
            f4 t2r[NC], t2i[NC];
            mdft_tw2(t2r, t2i);
            h8 xh[NC], xl[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                uint32_t h[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) split_h(vin[c][j], h[j], l[j]);
                xh[c] = __builtin_bit_cast(h8, (u4){h[0], h[1], h[2], h[3]});
                xl[c] = __builtin_bit_cast(h8, (u4){l[0], l[1], l[2], l[3]});
                mma_operand_fence(xh[c], xl[c]);
            }
            f4 orr[NC], oi[NC];
            mdft_big<NC, false>(xh, xl, dc, t2r, t2i, orr, oi);
#pragma unroll
            for (int c = 0; c < NC; ++c) { yr[c] = orr[c]; yi[c] = oi[c]; }
            STAMPC(15);
            if (DUMP && p.dump.Y) {
                const float us = p.dump_unscale_rx / p.rx_scale[sn * n_ch + ch];
#pragma unroll
                for (int kc = 0; kc < NC; ++kc)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        p.dump.Y[s0 * N + lane + 64 * j + 256 * kc] = make_float2(yr[kc][j] * us, yi[kc][j] * us);
            }
            if (wv == 0) {
                // pilot: G = X0 / Y0 in OUTPUT element order, one 16-byte row of real and one of imaginary parts per kc
                f4 *G4 = reinterpret_cast<f4 *>(G);
#pragma unroll
                for (int kc = 0; kc < NC; ++kc) {
                    f4 gr, gi;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const hpair hw = __builtin_bit_cast(hpair, qlw[(labo[kc] >> (8 * j)) & lmask]);
                        float x0r = (float)hw.y, x0i = (float)hw.x;
                        if constexpr (ALLOC) {
                            if ((labo[kc] >> (8 * j)) & 0x80u) { x0r = 0.f; x0i = 0.f; }      // G = 0 there
                        }
                        const float y0r = yr[kc][j], y0i = yi[kc][j];
                        const float inv = __builtin_amdgcn_rcpf(y0r * y0r + y0i * y0i);
                        gr[j] = (x0r * y0r + x0i * y0i) * inv;
                        gi[j] = (x0i * y0r - x0r * y0i) * inv;
                    }
                    G4[(2 * kc) * 64 + lane] = gr;
                    G4[(2 * kc + 1) * 64 + lane] = gi;
                }
                if constexpr (RELAXF) {
                    wave_sync();
                    post_flag(&flags[16], iter, lane);
                }
            }
        } else {
        {
            (void)kap; (void)h2;
            const int l0 = QW ? llq : lane;                      // the lane's first element of a symbol
            if constexpr (MDFT && RXSWZ) {
                // swizzled rows (see the noise-scaling stores above): element r of symbol u is sample x + 64 r of the chunk of its
                // plane, x = (u % 2) B + gamma + lane; 64 samples on leave the XOR term alone -- one per-lane base per symbol, the
                // elements (and the folded partner, N samples on) at instruction offsets as before
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const v2f *cb = reinterpret_cast<const v2f *>(Hp + (u < 2 ? 0 : plen) + PRE + s0 * B);
                    const v2f *fe = cb + rx_swz((u & 1) * B + gam + lane);
                    const float *wr = wrx + lane;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[u][0][r] = fe[64 * r] * wr[64 * r];
                    if (delta > 0) {
                        if (lane < delta) {
                            const float w2 = wr[N];
                            v[u][0][0] = __builtin_elementwise_fma(mk(w2, w2), fe[N], v[u][0][0]);
                        }
                    }
                }
                if (rx_part) {
                    // the chunk's incomplete last group keeps its order: what the SECOND symbol of a plane's chunk has in it -- element 3
                    // of its last lanes, folded partners -- is read once more, where it is
#pragma unroll
                    for (int u = 1; u < 4; u += 2) {
                        const v2f *cb = reinterpret_cast<const v2f *>(Hp + (u < 2 ? 0 : plen) + PRE + s0 * B);
                        const int x0 = B + gam + lane;
                        const float *wr = wrx + lane;
                        if (x0 + 192 >= rx_xb) v[u][0][3] = cb[x0 + 192] * wr[192];
                        if (delta > 0) {
                            if (lane < delta && x0 + N >= rx_xb) {
                                const float w2 = wr[N];
                                v[u][0][0] = __builtin_elementwise_fma(mk(w2, w2), cb[x0 + N], cb[rx_swz(x0)] * wr[0]);
                            }
                        }
                    }
                }
            } else {
#pragma unroll
            for (int u = 0; u < VS; ++u) {
                const v2f *fy = row(sym_of(u) - s0) + (gam + l0);
                const float *wr = wrx + l0;
#pragma unroll
                for (int q = 0; q < VB; ++q) {
                    if (owns(q)) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int e = QW ? 16 * (q + 4 * r) : 64 * q + r * NQ;      // sub_of(q, r) - l0
                            v[u][q][r] = fy[e] * wr[e];
                        }
                    }
                }
            }
            if (delta > 0) {
#pragma unroll
                for (int u = 0; u < VS; ++u) {
                    const v2f *fy = row(sym_of(u) - s0) + (gam + l0);
                    const float *wr = wrx + l0;
#pragma unroll
                    for (int q = 0; q < VB; ++q) {
                        if (owns(q)) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int e = QW ? 16 * (q + 4 * r) : 64 * q + r * NQ;
                                if (e < L::TAILRX_MAX) {                 // compile time: sample e + l0 < tail_rx?
                                    if (e + l0 < delta) {
                                        const float w2 = wr[e + N];
                                        v[u][q][r] = __builtin_elementwise_fma(mk(w2, w2), fy[e + N], v[u][q][r]);
                                    }
                                }
                            }
                        }
                    }
                }
            }
            }       // (not swizzled)
        }
        wave_sync();
        STAMPC(14);
        WAVE_PRIO(WOFDM_PRIO_X);
        // the pilot is symbol slot 0 of wave 0, four subcarriers per lane; G = X0 / Y0 (X0 as the table's small integers:
        // the demapper's levels are scaled to match) goes out as one 16-byte row of real and one of imaginary parts
        auto pilot_mdft = [&]() {
            f4 gr, gi;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t Lb = (lab[0][0] >> (8 * j)) & 0xFFu;
                const hpair hw = __builtin_bit_cast(hpair, qlw[Lb & lmask]);
                float x0r = (float)hw.y, x0i = (float)hw.x;
                if constexpr (ALLOC) {
                    if (Lb & 0x80u) { x0r = 0.f; x0i = 0.f; }
                }
                const float y0r = yr[0][j], y0i = yi[0][j];
                const float inv = __builtin_amdgcn_rcpf(y0r * y0r + y0i * y0i);
                gr[j] = (x0r * y0r + x0i * y0i) * inv;                 // X0 conj(Y0) / |Y0|^2
                gi[j] = (x0i * y0r - x0r * y0i) * inv;
            }
            f4 *G4 = reinterpret_cast<f4 *>(G);
            G4[lane] = gr;
            G4[64 + lane] = gi;
            if constexpr (RELAXF) {
                wave_sync();
                post_flag(&flags[16], iter, lane);
            }
        };
        (void)pilot_mdft;
        if constexpr (MDFT && WOFDM_MDFT_PIPELINE) {
            // the same pipeline as in phase A, one stage longer: behind the first-stage chain of symbol u the split of symbol u + 1's
            // received samples, behind that of symbol 3 the twiddle and split of symbol 0, then the second stage as there
            const mdft_consts dc = mdft_load(dce);
            f4 tr[4], ti[4];
            h8 xh[4], xl[4], th[4], tl[4];
            auto split_in = [&](int u) {
                uint32_t h[4], l[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) split_h(v[u][0][r], h[r], l[r]);
                xh[u] = __builtin_bit_cast(h8, (u4){h[0], h[1], h[2], h[3]});
                xl[u] = __builtin_bit_cast(h8, (u4){l[0], l[1], l[2], l[3]});
                mma_operand_fence(xh[u], xl[u]);
            };
            split_in(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                mma33_issue(tr[u], ti[u], xl[u], dc.brh, xh[u], dc.brl, xh[u], dc.brh, xl[u], dc.bih, xh[u], dc.bil, xh[u], dc.bih);
                if (u < 3) {
                    asm volatile("" : "+v"(tr[u]), "+v"(ti[u]), "+v"(v[u + 1][0][0]), "+v"(v[u + 1][0][1]), "+v"(v[u + 1][0][2]), "+v"(v[u + 1][0][3]));
                    split_in(u + 1);
                    asm volatile("" : "+v"(xh[u + 1]), "+v"(xl[u + 1]) : "v"(xh[u]), "v"(xl[u]), "v"(tr[u]), "v"(ti[u]), "v"(dc.brh), "v"(dc.brl),
                                 "v"(dc.bih), "v"(dc.bil));
                } else {
                    WOFDM_TIE2(tr[3], ti[3], tr[0], ti[0]);
                    mdft_twiddle(tr[0], ti[0], dc.twr, dc.twi);
                    mdft_split4(tr[0], ti[0], th[0], tl[0]);
                    asm volatile("" : "+v"(th[0]), "+v"(tl[0]) : "v"(xh[3]), "v"(xl[3]), "v"(tr[3]), "v"(ti[3]), "v"(dc.brh), "v"(dc.brl), "v"(dc.bih),
                                 "v"(dc.bil));
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                mma33_issue(yr[u], yi[u], dc.arh, tl[u], dc.arl, th[u], dc.arh, th[u], dc.aih, tl[u], dc.ail, th[u], dc.aih, th[u]);
                // the pilot wave publishes the equaliser as soon as ITS symbol is through (behind the second symbol's chain), not
                // after all four: the other waves waited for it 2.8 % of their time (profiles/r04_stamp_occ.txt)
                if (u == 1 && wv == 0) pilot_mdft();
                if (u < 3) {
                    WOFDM_TIE2(yr[u], yi[u], tr[u + 1], ti[u + 1]);
                    mdft_twiddle(tr[u + 1], ti[u + 1], dc.twr, dc.twi);
                    mdft_split4(tr[u + 1], ti[u + 1], th[u + 1], tl[u + 1]);
                    asm volatile("" : "+v"(th[u + 1]), "+v"(tl[u + 1]) : "v"(th[u]), "v"(tl[u]), "v"(yr[u]), "v"(yi[u]), "v"(dc.arh), "v"(dc.arl),
                                 "v"(dc.aih), "v"(dc.ail));
                } else {
                    asm volatile(WOFDM_MMA_TAIL : "+v"(yr[u]), "+v"(yi[u]) : "v"(th[u]), "v"(tl[u]), "v"(dc.arh), "v"(dc.arl), "v"(dc.aih), "v"(dc.ail));
                }
            }
        } else if constexpr (MDFT) {
            const mdft_consts dc = mdft_load(dce);
            f4 tr[4], ti[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                uint32_t h[4], l[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) split_h(v[u][0][r], h[r], l[r]);
                h8 xh = __builtin_bit_cast(h8, (u4){h[0], h[1], h[2], h[3]});
                h8 xl = __builtin_bit_cast(h8, (u4){l[0], l[1], l[2], l[3]});
                mma_operand_fence(xh, xl);
                mma33(tr[u], ti[u], xl, dc.brh, xh, dc.brl, xh, dc.brh, xl, dc.bih, xh, dc.bil, xh, dc.bih);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                mdft_twiddle(tr[u], ti[u], dc.twr, dc.twi);
                h8 th, tl;
                mdft_split4(tr[u], ti[u], th, tl);
                mma33(yr[u], yi[u], dc.arh, tl, dc.arl, th, dc.arh, th, dc.aih, tl, dc.ail, th, dc.aih, th);
            }
        } else if constexpr (QW) fft_qw<-1>(v, row(usq), tw, llq);
        else fft_wave<N, -1, SPW>(v, fbw, B, tw, lane);     // v = Y[n]
        STAMPC(15);
        WAVE_PRIO(WOFDM_PRIO_C);

        if constexpr (MDFT) {
            if (DUMP && p.dump.Y) {
                const float us = p.dump_unscale_rx / p.rx_scale[sn * n_ch + ch];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        p.dump.Y[(s0 + u) * N + lane + 64 * j] = make_float2(yr[u][j] * us, yi[u][j] * us);
            }
        } else
        if (DUMP && p.dump.Y) {
#pragma unroll
            for (int u = 0; u < VS; ++u)
#pragma unroll
                for (int q = 0; q < VB; ++q) {
                    if (owns(q))
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            p.dump.Y[sym_of(u) * N + sub_of(q, r)] = make_float2(v[u][q][r].x * (FIRM ? p.dump_unscale_rx : 1.f),
                                                                                  v[u][q][r].y * (FIRM ? p.dump_unscale_rx : 1.f));
                }
        }
        if constexpr (MDFT) {
            // (with the transforms as a pipeline the pilot wave has published its equaliser already: see the second stage above)
            if (wv == 0 && !WOFDM_MDFT_PIPELINE) pilot_mdft();
        } else if (QW && wv == 0) {
            // quarter-wave layout: the pilot symbol sits in lanes 0..15 of wave 0, and every other
            // wave waits for its equaliser.  Those 16 lanes only hand their Y0 and packed labels
            // over (through G itself and the pilot's now idle frame slice); all 64 lanes then
            // form G, four subcarriers each.
            uint32_t *plab = reinterpret_cast<uint32_t *>(row(0));
            if (usq == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    plab[16 * q + llq] = lab[0][q];
#pragma unroll
                    for (int r = 0; r < 4; ++r) G[sub_of(q, r)] = v[0][q][r];
                }
            }
            wave_sync();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = lane + 64 * i, t = n >> 4;            // n = l + 16 t, t = q + 4 r
                const uint32_t Lb = (plab[16 * (t & 3) + (n & 15)] >> (8 * (t >> 2))) & 0xFFu;
                v2f x0 = qlut[Lb & lmask];
                if constexpr (ALLOC) {
                    if (Lb & 0x80u) x0 = mk(0.f, 0.f);
                }
                const v2f y0 = G[n];
                const float inv = __builtin_amdgcn_rcpf(y0.x * y0.x + y0.y * y0.y);
                G[n] = cmul_conj(x0, y0) * inv;
            }
            if constexpr (RELAXF) {
                wave_sync();
                post_flag(&flags[16], iter, lane);
            }
        } else if (wv == 0) {
            // estimatedChannel = Y0 ./ X0 (m:266); we publish its reciprocal X0 ./ Y0
#pragma unroll
            for (int q = 0; q < VB; ++q) {
                if (owns(q) && (!QW || usq == 0)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v2f x0 = qlut[(lab[0][q] >> (8 * r)) & lmask];
                        if constexpr (ALLOC) {
                            if ((lab[0][q] >> (8 * r)) & 0x80u) x0 = mk(0.f, 0.f);   // G = 0 there
                        }
                        const v2f y0 = v[0][q][r];
                        const float inv = __builtin_amdgcn_rcpf(y0.x * y0.x + y0.y * y0.y);
                        G[sub_of(q, r)] = cmul_conj(x0, y0) * inv;        // X0 conj(Y0) / |Y0|^2
                    }
                }
            }
            if constexpr (RELAXF) {
                wave_sync();
                post_flag(&flags[16], iter, lane);
            }
        }
        }       // (layouts other than 12)
        }
        DELAY_AT(8);
        STAMP(4);
        WAVE_PRIO(WOFDM_PRIO_C);
        if constexpr (RELAXF) {
            if (wv != 0) wait_flag(&flags[16], iter, &flags[20]);                        // ---- "barrier" 3
        } else {
            __syncthreads();                                                 // ---- barrier 3
        }
        STAMP(5);
        WAVE_PRIO(WOFDM_PRIO_D);
        DELAY_AT(9);

        // ------------------------------------------------------------ D: equalise, demap, count
        if constexpr (RELAUNDER) asm volatile("" : "+v"(lane));
        uint32_t be_f = 0, se_f = 0;                   // this frame's errors of the lane (SCALAR_ACC)
        if constexpr (MDS) {
            const int lm = lane & 15, lg = lane >> 4;
            const f4 *G4 = reinterpret_cast<const f4 *>(G);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int pp = t % SCS, s = s0 + 4 * (t / SCS) + lg;
                const bool son = !PART || 4 * (t / SCS) + lg < min(SPWR, gm[WOFDM_G_S] - s0);      // (layout 16: a symbol slot this wave fills)
                const f4 gr = G4[pp * 16 + lm], gi = G4[(SCS + pp) * 16 + lm];
                const f4 ehr = yr[t] * gr - yi[t] * gi, ehi = yr[t] * gi + yi[t] * gr;
                const f4 lvi = ehr * 0.5f + 0.5f * (float)m1, lvq = ehi * -0.5f + 0.5f * (float)m1;
                uint32_t iw = 0, qw = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    iw = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fminf(lvi[e], (float)m1), e, iw);
                    qw = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fminf(lvq[e], (float)m1), e, qw);
                    if (DUMP && p.dump.Xhat && s > 0 && son)
                        p.dump.Xhat[(s - 1) * N + lm + 16 * (SC == 4 ? e : 2 * e + pp)] = make_float2(ehr[e] * qscale, ehi[e] * qscale);
                }
                constexpr uint32_t GM = K == 2 ? 0u : (K == 4 ? 0x05050505u : 0x1B1B1B1Bu);
                constexpr uint32_t LM = lmask * 0x01010101u;
                const uint32_t cw = (iw << half) | qw;
                const uint32_t gw = cw ^ ((cw >> 1) & GM);
                uint32_t diff = (gw ^ labo[t]) & LM;
                if constexpr (ALLOC) diff &= ~(((labo[t] >> 7) & 0x01010101u) * 0xFFu);   // bit 7 = not loaded: not counted
                if (s == 0 || !son) diff = 0;                       // the pilot symbol is not counted (nor an unfilled slot)
                bit_err += __popc(diff);
                sym_err += __popc((diff + 0x7F7F7F7Fu) & 0x80808080u);
                if (DUMP && p.dump.labels_rx && s > 0 && son) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        p.dump.labels_rx[(s - 1) * N + lm + 16 * (SC == 4 ? e : 2 * e + pp)] = (uint8_t)((gw >> (8 * e)) & lmask);
                }
            }
        }
        if constexpr (MDX) {
            if (s0 > 0) {
                const f4 *G4 = reinterpret_cast<const f4 *>(G);
#pragma unroll
                for (int kc = 0; kc < NC; ++kc) {
                    const f4 gr = G4[(2 * kc) * 64 + lane], gi = G4[(2 * kc + 1) * 64 + lane];
                    const f4 ehr = yr[kc] * gr - yi[kc] * gi, ehi = yr[kc] * gi + yi[kc] * gr;
                    const f4 lvi = ehr * 0.5f + 0.5f * (float)m1, lvq = ehi * -0.5f + 0.5f * (float)m1;
                    uint32_t iw = 0, qw = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        iw = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fminf(lvi[j], (float)m1), j, iw);
                        qw = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fminf(lvq[j], (float)m1), j, qw);
                        if (DUMP && p.dump.Xhat)
                            p.dump.Xhat[(s0 - 1) * N + lane + 64 * j + 256 * kc] = make_float2(ehr[j] * qscale, ehi[j] * qscale);
                    }
                    constexpr uint32_t GM = K == 2 ? 0u : (K == 4 ? 0x05050505u : 0x1B1B1B1Bu);
                    constexpr uint32_t LM = lmask * 0x01010101u;
                    const uint32_t cw = (iw << half) | qw;
                    const uint32_t gw = cw ^ ((cw >> 1) & GM);
                    uint32_t diff = (gw ^ labo[kc]) & LM;
                    if constexpr (ALLOC) diff &= ~(((labo[kc] >> 7) & 0x01010101u) * 0xFFu);   // bit 7 = not loaded: not counted
                    be_f += __popc(diff);
                    se_f += __popc((diff + 0x7F7F7F7Fu) & 0x80808080u);
                    if (DUMP && p.dump.labels_rx) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            p.dump.labels_rx[(s0 - 1) * N + lane + 64 * j + 256 * kc] = (uint8_t)((gw >> (8 * j)) & lmask);
                    }
                }
            }
        }
        f4 g4r = {0.f, 0.f, 0.f, 0.f}, g4i = g4r;
        if constexpr (MDFT) {
            g4r = reinterpret_cast<const f4 *>(G)[lane];
            g4i = reinterpret_cast<const f4 *>(G)[64 + lane];
        }
#pragma unroll
        for (int u = 0; u < ((MDX || MDS) ? 0 : VS); ++u) {
            const int s = sym_of(u);
            if (s > 0) {
#pragma unroll
                for (int q = 0; q < VB; ++q) {
                    if (owns(q)) {
                        // (matrix-pipe transforms: the equalised symbols in the table's integer scale, four at once)
                        f4 ehr = g4r, ehi = g4r, lvi = g4r, lvq = g4r;
                        if constexpr (MDFT) {
                            ehr = yr[u] * g4r - yi[u] * g4i;
                            ehi = yr[u] * g4i + yi[u] * g4r;
                            lvi = ehr * 0.5f + 0.5f * (float)m1;
                            lvq = ehi * -0.5f + 0.5f * (float)m1;
                        }
                        // The four subcarriers of a butterfly are demapped together, one byte each of the
                        // packed words: per-axis level index = rne(clamp(+-x qinv/2 + m1/2)) converted,
                        // clamped below and packed by v_cvt_pk_u8_f32; binary reflected Gray code on all
                        // eight fields at once; popcount of the XOR with the packed transmitted labels =
                        // bit errors, its non-zero bytes = symbol errors.
                        uint32_t iw = 0, qw = 0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int n = sub_of(q, r);
                            v2f xh, lev;
                            if constexpr (MDFT) {
                                xh = mk(ehr[r], ehi[r]) * qscale;
                                lev = mk(lvi[r], lvq[r]);
                            } else {
                                xh = cmul(v[u][q][r], G[n]);
                                lev = __builtin_elementwise_fma(
                                    xh, mk(0.5f * qinv, -0.5f * qinv), mk(0.5f * (float)m1, 0.5f * (float)m1));
                            }
                            iw = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fminf(lev.x, (float)m1), r, iw);
                            qw = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fminf(lev.y, (float)m1), r, qw);
                            if (DUMP && p.dump.Xhat) p.dump.Xhat[(s - 1) * N + n] = make_float2(xh.x, xh.y);
                        }
                        constexpr uint32_t GM = K == 2 ? 0u : (K == 4 ? 0x05050505u : 0x1B1B1B1Bu);
                        constexpr uint32_t LM = lmask * 0x01010101u;
                        const uint32_t cw = (iw << half) | qw;
                        const uint32_t gw = cw ^ ((cw >> 1) & GM);            // Gray labels, one per byte
                        uint32_t diff = (gw ^ lab[u][q]) & LM;
                        if constexpr (ALLOC) {
                            // bit 7 of a byte = subcarrier not loaded: not counted
                            diff &= ~(((lab[u][q] >> 7) & 0x01010101u) * 0xFFu);
                        }
                        // (bytes < 64: bit 7 of byte + 0x7F <=> byte non-zero)
                        if constexpr (SCALAR_ACC) {
                            be_f += __popc(diff);
                            se_f += __popc((diff + 0x7F7F7F7Fu) & 0x80808080u);
                        } else {
                            bit_err += __popc(diff);
                            sym_err += __popc((diff + 0x7F7F7F7Fu) & 0x80808080u);
                        }
                        if (DUMP && p.dump.labels_rx) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                p.dump.labels_rx[(s - 1) * N + sub_of(q, r)] = (uint8_t)((gw >> (8 * r)) & lmask);
                        }
                    }
                }
            }
        }

        if constexpr (SCALAR_ACC) {
            bit_err += wave_sum_u(be_f);
            sym_err += wave_sum_u(se_f);
#ifdef WOFDM_AUDIT
            if (p.audit && (uint32_t)(iter - 1) < p.audit_items) {
                const unsigned ab = wave_sum_u(be_f), as = wave_sum_u(se_f);
                if (lane == 0) {
                    uint32_t *a = p.audit + (((size_t)blockIdx.x * p.audit_items + (uint32_t)(iter - 1)) * 16 + wv) * 8;
                    a[0] = ab; a[1] = as; a[2] = __float_as_uint(aud_g); a[3] = __float_as_uint(aud_ps);
                    a[4] = __float_as_uint(aud_pn); a[5] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
                    a[6] = __builtin_amdgcn_s_getreg(20 | (31 << 11)); a[7] = (uint32_t)__builtin_amdgcn_s_memtime();
                }
            }
#endif
        }
        STAMP(6);
        if (++fidx == F) { fidx = 0; next_cell(); }
    }
    if (cur_cell != 0xFFFFFFFFu) flush(cur_cell);
    if constexpr (RELAXF && WOFDM_CHECKED_SYNC) {
        __syncthreads();
        if (tid == 0 && flags[20] != 0) atomicOr(p.status, 1u);   // host: results not to be used
    }
#ifdef WOFDM_STAMP
    if (lane0 == 0) {
        unsigned long long *dst = p.counts + 4 * (size_t)(p.first_cell + p.n_cells)
                                  + ((size_t)blockIdx.x * 16 + wv) * 16;
        for (int i = 0; i < 16; ++i) dst[i] = stamp_acc[i];
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Closed-form ICI + ISI power of a structure:
//   calculate_interference  matlab/main_interference_calculation.m:177-225
//   interf_power            python/ofdm_utils/interf_calc.py:20-113
//     A_m = W K P V_rx R  H_m  V_tx Gamma W^-1,   H_m[b, c] = h[m B + b - c],   m = 0, 1
//     P[n] = sum_{n' != n} |A_0[n, n']|^2 + sum_{n'} |A_1[n, n']|^2
// Column n' of A_m is the frame pipeline's own answer to a unit symbol on subcarrier n': the windowed,
// CP/CS-extended complex exponential x (the Tx matrix applied to e_n'), the 21-tap convolution over two
// symbol periods, and per period the Rx window / fold / shift + DFT of the kernels above -- no dense
// matrices, no RNG.  One workgroup per (window pair, channel) job; a wave takes the columns
// n' = wave, wave + W, ... and keeps |A|^2 row sums of its subcarriers (FFT output layout) in registers;
// one LDS reduction over the waves at the end.  (M = 1 + ceil((L - 1 + beta) / B) = 2 for every
// supported structure: B >= 64 > 36.)
struct wofdm_iparams {
    int P, B, mu, delta, gam, kap, n_ch, rowlen;   // rowlen: float2 per wave row (24 + 2B + 24 rounded up)
    float *power;                                  // [pairs][n_ch][N]
};
template <int N> struct interf_geo {
    static constexpr int WAVES = N <= 256 ? 16 : (N == 512 ? 8 : 4);
    static constexpr int RB2 = 2 * (N / 64 + 1);                      // FIR outputs per lane over 2B samples
    static constexpr int CH = RB2 % 6 == 0 ? 6 : (RB2 % 5 == 0 ? 5 : (RB2 % 4 == 0 ? 4 : 2));
};

template <int N>
__global__ void __launch_bounds__(interf_geo<N>::WAVES * 64)
wofdm_interf_kernel(const wofdm_iparams p, const float *__restrict__ g_wtx, const float *__restrict__ g_wrx,
                    const float2 *__restrict__ g_h_)
{
    constexpr int WAVES = interf_geo<N>::WAVES, RB2 = interf_geo<N>::RB2, LT = WOFDM_LT;
    constexpr int BPL = geo<N>::BPL, NQ = geo<N>::NQ;
    constexpr bool FULL = geo<N>::FULL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: FFT stage twiddles [N] | e^{+2 pi i k / N} [N] | w_rx [N + 64] | per wave: row [rowlen] + scratch [N]
    v2f *tw = reinterpret_cast<v2f *>(smem);
    v2f *wn = tw + N;
    float *wrx = reinterpret_cast<float *>(wn + N);
    v2f *rows = reinterpret_cast<v2f *>(wrx + N + 64);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int job = blockIdx.x, pair = job / p.n_ch, ch = job - pair * p.n_ch;
    const int P = p.P, B = p.B;
    fill_twiddles<N>(tw, tid, WAVES * 64);
    for (int i = tid; i < N; i += WAVES * 64) {
        float sv, cv;
        sincospif(2.0f * (float)i / (float)N, &sv, &cv);
        wn[i] = mk(cv, sv);
    }
    for (int i = tid; i < N + p.delta; i += WAVES * 64) wrx[i] = g_wrx[(size_t)pair * (N + p.delta) + i];
    v2f *row = rows + (size_t)wv * (p.rowlen + N);
    v2f *scr = row + p.rowlen;
    for (int i = lane; i < p.rowlen; i += 64) row[i] = mk(0.f, 0.f);
    __syncthreads();
    const v2f *__restrict__ taps = reinterpret_cast<const v2f *>(g_h_) + (size_t)ch * LT;
    const float *__restrict__ wtx = g_wtx + (size_t)pair * P;
    float pw[BPL][4];
#pragma unroll
    for (int q = 0; q < BPL; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) pw[q][r] = 0.f;
    const int h2 = p.delta >> 1;
    for (int np = wv; np < N; np += WAVES) {
        // x[c] = w_tx[c] e^{2 pi i ((c - mu) mod N) n' / N} / N at row[24 + c]  (tx matrix column, m:358-376)
        for (int c = lane; c < P; c += 64) {
            const int t = (c - p.mu) & (N - 1);
            row[24 + c] = wn[(t * np) & (N - 1)] * (wtx[c] * (1.0f / (float)N));
        }
        for (int c = P + lane; c < 2 * B + 24; c += 64) row[24 + c] = mk(0.f, 0.f);
        wave_sync();
        // z = conv(h, x) over two symbol periods (m:260): lane -> RB2 consecutive outputs from j0
        v2f acc[RB2];
        const int j0 = lane * RB2;
        fir_lane<RB2, interf_geo<N>::CH>(row + 24 - (LT - 1) + j0, taps, acc);
        wave_sync();
#pragma unroll
        for (int r = 0; r < RB2; ++r)
            if (j0 + r < 2 * B) row[24 + j0 + r] = acc[r];
        wave_sync();
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            // Rx window, fold, circular shift (m:297-355) of period m, then the DFT
            const v2f *fb = row + 24 + m * B;
            v2f v[1][BPL][4];
#pragma unroll
            for (int q = 0; q < BPL; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[0][q][r] = mk(0.f, 0.f);
                    if (!(FULL || lane + 64 * q < NQ)) continue;
                    const int m0 = (lane + 64 * q + r * NQ + p.kap + h2) & (N - 1);
                    v2f z = fb[p.gam + m0] * wrx[m0];
                    if (m0 < p.delta) {
                        const float w2 = wrx[m0 + N];
                        z = __builtin_elementwise_fma(mk(w2, w2), fb[p.gam + m0 + N], z);
                    }
                    v[0][q][r] = z;
                }
            fft_wave<N, -1, 1>(v, scr, 0, tw, lane);
#pragma unroll
            for (int q = 0; q < BPL; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = lane + 64 * q + r * NQ;
                    const float e = v[0][q][r].x * v[0][q][r].x + v[0][q][r].y * v[0][q][r].y;
                    if (FULL || lane + 64 * q < NQ)
                        pw[q][r] += (m == 0 && n == np) ? 0.f : e;    // the wanted term A_0[n, n] is no interference
                }
        }
        wave_sync();
    }
    // sum over the waves (each wave's row is free now): float [WAVES][N] in the rows area
    __syncthreads();
    float *red = reinterpret_cast<float *>(rows);
#pragma unroll
    for (int q = 0; q < BPL; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (FULL || lane + 64 * q < NQ) red[wv * N + lane + 64 * q + r * NQ] = pw[q][r];
    __syncthreads();
    for (int n = tid; n < N; n += WAVES * 64) {
        float t = 0.f;
        for (int w = 0; w < WAVES; ++w) t += red[w * N + n];
        p.power[(size_t)job * N + n] = t;
    }
}

// ---------------------------------------------------------------------------------------------
// Tx-side spectrum estimate: the transmitted waveform of a long run of symbols
// and its averaged periodogram,
//   wOFDMSystem.estimate_obr   python/ofdm_utils/timefreq_simulation.py:216-296 (Tx chain 242-258)
//   psd_estimate               timefreq_simulation.py:101-123
// The waveform kernel is phase A of the frame kernel fed with given symbols X[s][n] (any complex values,
// zeros on unloaded bins): IDFT, CP/CS copy, Tx window, overlap-add of the `overlap` tail samples --
// one wave per symbol, the overlapping samples by float atomics (two addends: order-independent).
struct wofdm_wparams {
    int P, mu, rho, overlap, no_symbols;
    float2 *x;                         // [overlap + no_symbols * (P - overlap)], zeroed by the host
};
template <int N>
__global__ void __launch_bounds__(1024) wofdm_txwave_kernel(const wofdm_wparams p, const float *__restrict__ g_wtx,
                                                            const float2 *__restrict__ X)
{
    constexpr int BPL = geo<N>::BPL, NQ = geo<N>::NQ;
    constexpr bool FULL = geo<N>::FULL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v2f *tw = reinterpret_cast<v2f *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    v2f *scr = tw + N + (size_t)wv * N;
    fill_twiddles<N>(tw, tid, 1024);
    __syncthreads();
    const int s = blockIdx.x * 16 + wv;
    if (s >= p.no_symbols) return;
    v2f v[1][BPL][4];
#pragma unroll
    for (int q = 0; q < BPL; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[0][q][r] = mk(0.f, 0.f);
            if (FULL || lane + 64 * q < NQ) v[0][q][r] = ldg2(X + (size_t)s * N + lane + 64 * q + r * NQ);
        }
    fft_wave<N, +1, 1>(v, scr, 0, tw, lane);                    // N x[t]
    const int Bo = p.P - p.overlap;
    float2 *out = p.x + (size_t)s * Bo;
    auto put = [&](int i, v2f val) {
        val = val * (g_wtx[i] * (1.0f / (float)N));
        if (i < p.overlap || i >= Bo) {                          // shared with a neighbour symbol
            atomicAdd(&out[i].x, val.x);
            atomicAdd(&out[i].y, val.y);
        } else {
            out[i] = make_float2(val.x, val.y);
        }
    };
#pragma unroll
    for (int q = 0; q < BPL; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (!(FULL || lane + 64 * q < NQ)) continue;
            const int t = lane + 64 * q + r * NQ;
            put(t + p.mu, v[0][q][r]);
            if (t >= N - p.mu) put(t + p.mu - N, v[0][q][r]);
            if (t < p.rho) put(t + p.mu + N, v[0][q][r]);
        }
}

// Sum over consecutive FL-sample slices of x (the zero-padded remainder included) of |FFT_FL|^2, written
// fftshift-ed; the caller divides by the reference's slice count.  FL = 2048 runs as two 1024-point
// transforms of the even and odd samples and one radix-2 combination in registers.
template <int FL>
__global__ void __launch_bounds__(512) wofdm_psd_kernel(const float2 *__restrict__ x, int len, int n_slices,
                                                        float *__restrict__ psd)
{
    constexpr int M = FL == 2048 ? 1024 : FL, H = FL / M;        // transform length, transforms per slice
    constexpr int BPL = geo<M>::BPL, NQ = geo<M>::NQ, WAVES = 8;
    static_assert(geo<M>::FULL, "at least 256 points");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v2f *tw = reinterpret_cast<v2f *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    v2f *scr = tw + M + (size_t)wv * M;
    fill_twiddles<M>(tw, tid, WAVES * 64);
    __syncthreads();
    float acc[H][BPL][4];
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
        for (int q = 0; q < BPL; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[h][q][r] = 0.f;
    for (int sl = wv; sl < n_slices; sl += WAVES) {
        v2f e[1][BPL][4], o[1][BPL][4];
#pragma unroll
        for (int h = 0; h < H; ++h) {
            v2f (&dst)[1][BPL][4] = h == 0 ? e : o;
#pragma unroll
            for (int q = 0; q < BPL; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = sl * FL + H * (lane + 64 * q + r * NQ) + h;       // even / odd samples
                    dst[0][q][r] = idx < len ? ldg2(x + idx) : mk(0.f, 0.f);
                }
            fft_wave<M, -1, 1>(dst, scr, 0, tw, lane);
        }
#pragma unroll
        for (int q = 0; q < BPL; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (H == 2) {
                    const int k = lane + 64 * q + r * NQ;
                    float sv, cv;
                    sincospif(-2.0f * (float)k / (float)FL, &sv, &cv);
                    const v2f wo = cmul(o[0][q][r], mk(cv, sv));
                    const v2f a = e[0][q][r] + wo, b = e[0][q][r] - wo;
                    acc[0][q][r] += a.x * a.x + a.y * a.y;
                    acc[1][q][r] += b.x * b.x + b.y * b.y;
                } else {
                    acc[0][q][r] += e[0][q][r].x * e[0][q][r].x + e[0][q][r].y * e[0][q][r].y;
                }
            }
    }
    __syncthreads();
    float *red = reinterpret_cast<float *>(tw + M);                 // [WAVES][FL] over the scratch rows
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
        for (int q = 0; q < BPL; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wv * FL + h * M + lane + 64 * q + r * NQ] = acc[h][q][r];
    __syncthreads();
    for (int k = tid; k < FL; k += WAVES * 64) {
        float t = 0.f;
        for (int w = 0; w < WAVES; ++w) t += red[w * FL + k];
        psd[(k + FL / 2) & (FL - 1)] = t;
    }
}

__global__ void philox_kat_kernel(const uint32_t *ck, uint32_t *out)
{
    if (threadIdx.x == 0) {
        const philox_out o = philox4x32_10(ck[0], ck[1], ck[2], ck[3], ck[4], ck[5]);
        for (int i = 0; i < 4; ++i) out[i] = o.w[i];
    }
}

template <int N, int K, int SPW, int VAR> wofdm_kernel_fn pick_mode(int mode)
{
    switch (mode) {
    case WOFDM_MODE_GEN: return wofdm_frames_kernel<N, K, SPW, false, false, VAR>;
    case WOFDM_MODE_INJECT: return wofdm_frames_kernel<N, K, SPW, true, false, VAR>;
    case WOFDM_MODE_DUMP_GEN: return wofdm_frames_kernel<N, K, SPW, false, true, VAR>;
    case WOFDM_MODE_DUMP_INJECT: return wofdm_frames_kernel<N, K, SPW, true, true, VAR>;
    }
    return nullptr;
}

template <int N, int K, int SPW> wofdm_kernel_fn pick_var(int mode, int var)
{
    switch (var) {
    case WOFDM_VAR_PLAIN: return pick_mode<N, K, SPW, WOFDM_VAR_PLAIN>(mode);
    case WOFDM_VAR_ALLOC: return pick_mode<N, K, SPW, WOFDM_VAR_ALLOC>(mode);
    case WOFDM_VAR_TXMASK:
        if constexpr (SPW == 1 && N <= WOFDM_TXMASK_MAX_N) return pick_mode<N, K, SPW, WOFDM_VAR_TXMASK>(mode);
        break;
    case WOFDM_VAR_TXFFT:
        if constexpr (SPW == 1 && N <= WOFDM_TXFFT_MAX_N) return pick_mode<N, K, SPW, WOFDM_VAR_TXFFT>(mode);
        break;
    }
    return nullptr;
}

template <int N, int K> wofdm_kernel_fn pick_spw(int spw, int mode, int var)
{
    if (spw == 1) return pick_var<N, K, 1>(mode, var);
    if constexpr (N == 64 || N == 128) {
        if (spw == 13 && var <= WOFDM_VAR_ALLOC)
            return var ? pick_mode<N, K, 13, WOFDM_VAR_ALLOC>(mode) : pick_mode<N, K, 13, WOFDM_VAR_PLAIN>(mode);
        if (spw == 14 && var <= WOFDM_VAR_ALLOC)
            return var ? pick_mode<N, K, 14, WOFDM_VAR_ALLOC>(mode) : pick_mode<N, K, 14, WOFDM_VAR_PLAIN>(mode);
        if (spw == 16 && var <= WOFDM_VAR_ALLOC)
            return var ? pick_mode<N, K, 16, WOFDM_VAR_ALLOC>(mode) : pick_mode<N, K, 16, WOFDM_VAR_PLAIN>(mode);
    }
    if (spw == 9) {
        if constexpr (N <= WOFDM_TXMASK_MAX_N) {
            if (var == WOFDM_VAR_TXMASK) return pick_mode<N, K, 9, WOFDM_VAR_TXMASK>(mode);
        }
        if constexpr (N <= WOFDM_TXFFT_MAX_N) {
            if (var == WOFDM_VAR_TXFFT) return pick_mode<N, K, 9, WOFDM_VAR_TXFFT>(mode);
        }
        return nullptr;
    }
    if constexpr (N <= 256) {
        if (spw == 2) return pick_var<N, K, 2>(mode, var);
    }
    if constexpr (N >= 512) {
        if (spw == 12 && var <= WOFDM_VAR_ALLOC)
            return var ? pick_mode<N, K, 12, WOFDM_VAR_ALLOC>(mode) : pick_mode<N, K, 12, WOFDM_VAR_PLAIN>(mode);
        if (spw == 8 && var <= WOFDM_VAR_ALLOC)
            return var ? pick_mode<N, K, 8, WOFDM_VAR_ALLOC>(mode) : pick_mode<N, K, 8, WOFDM_VAR_PLAIN>(mode);
    }
    if constexpr (N == 256) {
        if (spw == 15) return var == WOFDM_VAR_TXFFT ? pick_mode<N, K, 15, WOFDM_VAR_TXFFT>(mode) : nullptr;
        if ((spw == 10 || spw == 11) && var <= WOFDM_VAR_ALLOC) {
            if (spw == 10) return var ? pick_mode<N, K, 10, WOFDM_VAR_ALLOC>(mode) : pick_mode<N, K, 10, WOFDM_VAR_PLAIN>(mode);
            return var ? pick_mode<N, K, 11, WOFDM_VAR_ALLOC>(mode) : pick_mode<N, K, 11, WOFDM_VAR_PLAIN>(mode);
        }
        if (spw >= 4 && spw <= 7 && var <= WOFDM_VAR_ALLOC) {
            if (spw == 4) return var ? pick_mode<N, K, 4, WOFDM_VAR_ALLOC>(mode) : pick_mode<N, K, 4, WOFDM_VAR_PLAIN>(mode);
            if (spw == 5) return var ? pick_mode<N, K, 5, WOFDM_VAR_ALLOC>(mode) : pick_mode<N, K, 5, WOFDM_VAR_PLAIN>(mode);
            if (spw == 6) return var ? pick_mode<N, K, 6, WOFDM_VAR_ALLOC>(mode) : pick_mode<N, K, 6, WOFDM_VAR_PLAIN>(mode);
            return var ? pick_mode<N, K, 7, WOFDM_VAR_ALLOC>(mode) : pick_mode<N, K, 7, WOFDM_VAR_PLAIN>(mode);
        }
    }
    return nullptr;
}

template <int N> wofdm_kernel_fn pick(int k, int spw, int mode, int var)
{
    // (one constellation size per translation unit, see below)
    return k == WOFDM_TU_K ? pick_spw<N, WOFDM_TU_K>(spw, mode, var) : nullptr;
}

}  // namespace

// This file is compiled once per (DFT length, bits per subcarrier) (-DWOFDM_TU_N=<N>
// -DWOFDM_TU_K=<k>, see the Makefile) so that the kernel family builds in parallel;
// wofdm_kernel.h dispatches on n_fft and bits_per_sc.
#if !defined(WOFDM_TU_N) || !defined(WOFDM_TU_K)
#error "compile with -DWOFDM_TU_N=<64|128|256|512|1024> -DWOFDM_TU_K=<2|4|6>"
#endif
#define WOFDM_CAT2(a, b) a##b
#define WOFDM_CAT(a, b) WOFDM_CAT2(a, b)

wofdm_kernel_fn WOFDM_CAT(WOFDM_CAT(WOFDM_CAT(wofdm_select_kernel_n, WOFDM_TU_N), _k), WOFDM_TU_K)(int spw, int mode, int var)
{
    return pick<WOFDM_TU_N>(WOFDM_TU_K, spw, mode, var);
}

#if WOFDM_TU_K == 2
// one interference kernel per DFT length, compiled in the k = 2 translation units
hipError_t WOFDM_CAT(wofdm_interf_launch_n, WOFDM_TU_N)(int jobs, int P, int B, int mu, int delta, int gam, int kap,
                                                        int n_ch, const float *wtx, const float *wrx, const float2 *h,
                                                        float *power, hipStream_t s)
{
    constexpr int N = WOFDM_TU_N, W = interf_geo<N>::WAVES;
    wofdm_iparams ip;
    ip.P = P; ip.B = B; ip.mu = mu; ip.delta = delta; ip.gam = gam; ip.kap = kap; ip.n_ch = n_ch;
    ip.rowlen = (24 + 2 * B + 24 + 64 * interf_geo<N>::RB2 - 2 * B + 1) / 2 * 2;   // covers every lane's FIR window
    ip.power = power;
    const size_t lds = 8 * (size_t)N * 2 + 4 * (size_t)(N + 64) + (size_t)W * 8 * (ip.rowlen + N);
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(wofdm_interf_kernel<N>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(wofdm_interf_kernel<N>, dim3(jobs), dim3(W * 64), lds, s, ip, wtx, wrx, h);
    return hipGetLastError();
}
#endif

#if WOFDM_TU_K == 2 && WOFDM_TU_N <= 256
// Tx waveform + periodogram (row f4), per DFT length, in the k = 2 translation units
hipError_t WOFDM_CAT(wofdm_psd_launch_n, WOFDM_TU_N)(int P, int mu, int rho, int overlap, int no_symbols, const float *wtx,
                                                     const float2 *X, float2 *x, int len, float *psd, hipStream_t s)
{
    constexpr int N = WOFDM_TU_N, FL = 8 * N, M = FL == 2048 ? 1024 : FL;
    wofdm_wparams wp;
    wp.P = P; wp.mu = mu; wp.rho = rho; wp.overlap = overlap; wp.no_symbols = no_symbols; wp.x = x;
    const size_t lds_a = 8 * (size_t)N * 17, lds_b = 8 * (size_t)M * 9;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(wofdm_psd_kernel<FL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(wofdm_txwave_kernel<N>, dim3((no_symbols + 15) / 16), dim3(1024), lds_a, s, wp, wtx, X);
    hipLaunchKernelGGL(wofdm_psd_kernel<FL>, dim3(1), dim3(512), lds_b, s, (const float2 *)x, len, (len + FL - 1) / FL, psd);
    return hipGetLastError();
}
#endif

#if WOFDM_TU_N == 64 && WOFDM_TU_K == 2
hipError_t wofdm_philox_kat_launch(const uint32_t *ctr_key_dev, uint32_t *out_dev, hipStream_t s)
{
    hipLaunchKernelGGL(philox_kat_kernel, dim3(1), dim3(64), 0, s, ctr_key_dev, out_dev);
    return hipGetLastError();
}
#endif
