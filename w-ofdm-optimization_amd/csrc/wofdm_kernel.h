// Host <-> kernel contract of the frame kernel (internal to libwofdm_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WOFDM_LT 21   // taps the FIR is unrolled for (channels are zero-padded to it)

struct wofdm_kdump {          // device pointers, all may be null
    uint8_t *labels_tx;
    float2  *X, *tx, *conv, *rx, *Y, *Xhat;
    uint8_t *labels_rx;
    float   *gain;
    float2  *unit_noise;
    float2  *sink;            // where the per-lane stage dumps of lanes without a sample go: the
                              // instrumented kernels store unconditionally instead of branching
};

// LDS carve: fixed-size regions first (compile-time offsets), the frame buffer last.
//   tw    float2[N]          twiddles exp(-2 pi i m / N)    (N = 256, 512: 6 KB -- the matrix-pipe layouts 10 / 11 / 12 keep
//                            six of the ten operand rows of the transforms there, [6][64] 16-byte rows)
//   g     float2[N]          pilot equaliser X0/Y0
//   sums  float [2][32]      per-wave signal / noise power partials, double-buffered by frame parity
//   flags int   [64]         [w] = last loop iteration whose phase A wave w has finished,
//                            [16] = last iteration whose pilot equaliser G is published, [20] = a wave gave up waiting,
//                            [32 + w] = (Tx-mask variants) last iteration whose masked symbol wave w has written to its row,
//                            [48 + w] = (Tx-mask variants on the matrix pipe, layouts 9 / 15) last iteration whose row wave w has
//                            turned into its two f16 planes (the successor's first tile reads the row's last samples)
//   wtx   float [N + CPCS]   Tx window / N      (needs cp + cs <= CPCS_MAX = 128; 64 at N = 1024,
//                            where the frame buffer leaves no room for more anyway)
//   wrx   float [N + 64]     Rx window          (needs tail_rx <= 64)
//   lut   float2[64]         QAM constellation by label
//   fbuf  float2[fbuf_len]   WOFDM_LT-1 zeros | frame (T) | zeros
//   tail  float2[S][tail_tx] fall tails, behind the frame buffer (then the Tx-mask tables, if any)
// (NOTW: layouts 13 / 14 / 16 -- N = 64, 128 with the transforms on the matrix pipe -- keep no twiddle table: their operand rows come
// from L2.  Those 512 bytes decide at N = 64 whether a workgroup takes twelve or thirteen of the LDS's 128 allocation units --
// ten or nine workgroups per CU: wofdm_lds_workgroups_per_cu)
template <int N, bool NOTW = false> struct wofdm_lds {
    static constexpr int TAIL_MAX = 16, CPCS_MAX = N >= 1024 ? 64 : 128, TAILRX_MAX = 64;
    static constexpr int off_tw = 0;
    static constexpr int TW_BYTES = NOTW ? 0 : (N == 256 || N == 512 ? 6 * 64 * 16 : 8 * N);
    static constexpr int off_g = off_tw + TW_BYTES;
    static constexpr int off_sums = off_g + 8 * N;
    static constexpr int off_flags = off_sums + 4 * 64;
    static constexpr int off_wtx = off_flags + 4 * 64;
    static constexpr int off_wrx = off_wtx + 4 * (N + CPCS_MAX);
    static constexpr int off_lut = off_wrx + 4 * (N + TAILRX_MAX);
    static constexpr int off_fbuf = off_lut + 8 * 64;
};

// geometry array read by the kernel through a laundered pointer (see GEO_PHASE)
enum { WOFDM_G_S, WOFDM_G_MU, WOFDM_G_RHO, WOFDM_G_BETA, WOFDM_G_DELTA, WOFDM_G_GAMMA, WOFDM_G_KAPPA,
       WOFDM_G_L, WOFDM_G_P, WOFDM_G_B, WOFDM_G_T, WOFDM_G_NL, WOFDM_G_NSNR, WOFDM_G_NCH, WOFDM_G_FBUF,
       WOFDM_G_NACT,      // loaded subcarriers (WOFDM_VAR_ALLOC)
       WOFDM_G_SPWR,      // layout 16: symbols a wave takes (wofdm_small_spwr; the other layouts hold theirs at compile time)
       WOFDM_G_COUNT };

struct wofdm_kparams {
    uint32_t n_cells;       // cells covered by this launch, starting at first_cell
    uint32_t first_cell;
    uint32_t inject_base_cell;   // injected arrays AND the counter array are indexed from this cell
    unsigned lds_bytes;
    uint64_t frames_per_cell, frame_offset;
    uint64_t items_q, items_r;     // (cell, frame) items per workgroup: q, and one more for the first r
    uint32_t seed_lo, seed_hi;
    unsigned long long *counts;   // [cells][4], entry 0 = cell inject_base_cell
    // layouts 10 / 11 / 12 (both transforms on the matrix pipe): operand table [10 + 2 (N/256 - 1)][64] x 16 bytes
    // (wofdm_abi.hip) and the power of two per (snr, channel) that centres the received samples in the f16 range
    const uint4 *dftc;
    const float *rx_scale;
    const uint8_t *labels;  // inject: [cells][frames][S][N]
    const float2  *unit_noise;    // inject: [cells][frames][NL]
    float2 *noise_scratch;        // generate, N >= WOFDM_NOISE_SCRATCH_MIN_N: [grid][16][RB][64]
    unsigned *status;             // device word, bit 0 set if a wave gave up waiting on a flag
    // Scaling of the on-air signal inside the kernel.  The Tx window table holds w_tx * tx_scale
    // (1/N of the IDFT, and in the matrix-pipe FIR layouts a power of two that centres the samples
    // in the f16 range); the channel's Toeplitz operands carry their own power of two.  Everything
    // behind the FIR is homogeneous in that scale (the noise gain is derived from the measured
    // powers, the equaliser divides by the pilot), so only the stage dumps undo it.
    float tx_scale, dump_unscale_tx, dump_unscale_rx;
    wofdm_kdump dump;
#ifdef WOFDM_DELAY
    // developer build (tools/delay_probe.py): the waves of delay_waves sleep delay_len x 4 us at point delay_point
    uint32_t delay_point, delay_waves, delay_len;
#endif
#ifdef WOFDM_AUDIT
    // developer build (tools/audit_suite.py): per (workgroup, frame, wave) record of 8 words -- bit errors,
    // symbol errors, noise gain, Ps, Pn, HW_ID, XCC_ID, clock -- for the first audit_items frames of a workgroup
    uint32_t *audit;
    uint32_t audit_items;
#endif
};

// DFT lengths from which the generated unit noise is parked in HBM scratch between the FIR and
// the noise-scaling phase instead of registers
#ifndef WOFDM_NOISE_SCRATCH_MIN_N
#define WOFDM_NOISE_SCRATCH_MIN_N 1024
#endif
// largest cp + cs the Tx window table of the kernel holds (wofdm_lds<N>::CPCS_MAX)
static inline int wofdm_cpcs_max(int n_fft) { return n_fft >= 1024 ? 64 : 128; }
static inline int wofdm_kslot(int k) { return k == 6 ? 8 : k; }
// FIR outputs per lane for `spw` symbols per wave (fir_geo in wofdm_kernel.hip)
// (spw is the kernel's layout id: symbols per wave, or 5 = four symbols with 20 outputs per lane)
// 6 / 7 = four symbols per wave with the FIR on the matrix pipe (wofdm_firm_tiles tiles of 128
// samples per wave, two samples per lane and tile)
// 8 = one symbol per wave with the FIR on the matrix pipe (N >= 512; wofdm_fir8_tiles tiles per wave)
// 10 / 11 = 6 / 7 with both 256-point transforms on the matrix pipe as well (lane l holds elements l + 64 j of each of the
// wave's four symbols)
// 12 = 8 with both transforms on the matrix pipe as well (N = 512, 1024: 16 . 16 . N/256, the last stage in registers)
// 13 / 14 = N = 64, 128 with the FIR and both transforms on the matrix pipe: 16 / 8 symbols per wave (one MFMA stage over the stride-N/16
// index, the radix-N/16 stage in registers), 10 / 11 FIR tiles per wave
// 16 = 13 with a RUN-TIME number of symbols per wave (even, <= 1024 / N) and a partly filled last wave: the N = 64, 128 geometries
// layouts 13 / 14 do not take -- S not a multiple of 16 / 8, strides beyond their tiles (N = 64 at CP 32: two waves of eight
// symbols) -- which ran layout 2 before (round 4)
static inline bool wofdm_is_mdft(int spw) { return spw == 10 || spw == 11 || spw == 13 || spw == 14 || spw == 16; }
static inline bool wofdm_is_small(int spw) { return spw == 13 || spw == 14 || spw == 16; }
// 9 = one symbol per wave, FIR on the matrix pipe, for the Tx-mask variants (any N <= 512): layout 8's frame format; the mask
// stage works on the rows as fp32, phase B turns them into the f16 planes in place
// 15 = 9 at N = 256 for the fast-convolution Tx mask with every transform on the matrix pipe: the symbol's own two as in layout 12
// (one set), the mask's two 1024-point ones as four sets each with no exchange in between (no mask scratch in LDS)
static inline bool wofdm_is_fir8(int spw) { return spw == 8 || spw == 9 || spw == 12 || spw == 15; }
static inline bool wofdm_is_firm(int spw) { return (spw >= 6 && spw <= 9) || wofdm_is_mdft(spw) || spw == 12 || spw == 15; }
static inline int wofdm_firm_tiles(int spw) { return spw == 14 ? 11 : ((spw == 7 || spw == 11 || spw == 13 || spw == 16) ? 10 : 9); }
static constexpr int wofdm_fir8_tiles(int n_fft) { return n_fft >= 1024 ? 9 : (n_fft >= 512 ? 5 : 3); }
#ifndef WOFDM_SMALL_PARTIAL
#define WOFDM_SMALL_PARTIAL 1 // layout 16 (N = 64, 128 with a run-time number of symbols per wave); 0: layout 2 there, as before round 4
#endif
#ifndef WOFDM_ODD_STRIDES
#define WOFDM_ODD_STRIDES 1   // odd strides on the matrix pipe with one symbol per wave (layout 12); 0: layout 1 as before round 4
#endif
#define WOFDM_FIR8_VT 48      // words per plane of layout 8's virtual row behind the last symbol
#define WOFDM_FIRM_PRE 24     // zero samples in front of the frame in the f16 planes (taps - 1 <= 24, 16-byte rows)
static inline int wofdm_rb(int n_fft, int spw = 1)
{
    if (wofdm_is_fir8(spw)) return 2 * wofdm_fir8_tiles(n_fft);
    if (wofdm_is_firm(spw)) return 2 * wofdm_firm_tiles(spw);
    return spw == 1 ? n_fft / 64 + 1 : (spw == 5 ? 20 : spw * (n_fft / 64) + 2);
}
static inline int wofdm_nsym(int spw, int n_fft = 256)
{
    if (wofdm_is_small(spw)) return 1024 / n_fft;           // 16 symbols per wave at N = 64, 8 at N = 128 (layout 16: at most)
    return wofdm_is_fir8(spw) ? 1 : ((spw == 5 || wofdm_is_firm(spw)) ? 4 : spw);
}
// Layout 16: the symbols a wave takes -- the frame spread evenly over the fewest waves (at most four) whose share, rounded up to
// an even count (a wave's rows are split between the two f16 planes), fits the 1024 / N symbol slots and the ten tiles of a
// wave; 0: none fits.  The last wave takes what is left (S - (W - 1) spwr symbols, any count >= 1).
static inline int wofdm_small_spwr(int n_fft, int S, int B)
{
    const int slots = 1024 / n_fft;
    for (int W = 1; W <= 4; ++W) {
        const int spwr = ((S + W - 1) / W + 1) & ~1;
        if (spwr <= slots && spwr * B <= 128 * 10 && (W - 1) * spwr < S) return spwr;
    }
    return 0;
}
// symbols per wave as the kernel of layout `spw` runs this geometry, and the waves of its workgroup
static inline int wofdm_spwr(int spw, int n_fft, int S, int B)
{
    return spw == 16 ? wofdm_small_spwr(n_fft, S, B) : wofdm_nsym(spw, n_fft);
}
static inline int wofdm_waves(int spw, int n_fft, int S, int B)
{
    const int r = wofdm_spwr(spw, n_fft, S, B);
    return r > 0 ? (S + r - 1) / r : 0;
}
// symbols per wave: four at N = 256 without Tx mask (quarter-wave layout, S a multiple of 4,
// four symbols within the 64 x 18 FIR outputs of a wave), else two where the register budget allows
// it (N <= 256) and S is even, else one.  WOFDM_MAX_SPW (developer switch) caps it.
#ifndef WOFDM_MAX_SPW
#define WOFDM_MAX_SPW 4
#endif
static inline int wofdm_spw(int n_fft, int S, int B, bool plain = false, bool firm = true, bool mdft = true)
{
    // (the matrix-pipe kernels take a stride of at least n_fft for granted: their tiles below SPW n_fft
    // samples carry no validity tests)
    firm = firm && B >= n_fft;
    if (WOFDM_MAX_SPW >= 4 && plain && mdft && firm && n_fft <= 128 && S % (1024 / n_fft) == 0) {
        if ((1024 / n_fft) * B <= 128 * wofdm_firm_tiles(13)) return 13;
        if ((1024 / n_fft) * B <= 128 * wofdm_firm_tiles(14)) return 14;
    }
    if (WOFDM_MAX_SPW >= 4 && WOFDM_SMALL_PARTIAL && plain && mdft && firm && n_fft <= 128 && wofdm_small_spwr(n_fft, S, B) > 0) return 16;
    if (WOFDM_MAX_SPW >= 4 && plain && n_fft == 256 && S % 4 == 0) {
        if (firm && 4 * B <= 128 * wofdm_firm_tiles(6)) return mdft ? 10 : 6;
        if (firm && 4 * B <= 128 * wofdm_firm_tiles(7)) return mdft ? 11 : 7;
        if (4 * B <= 64 * wofdm_rb(n_fft, 4)) return 4;
        if (4 * B <= 64 * wofdm_rb(n_fft, 5)) return 5;       // 288 < B <= 320: 20 outputs per lane
    }
    // one symbol per wave, matrix-pipe FIR: every stride (a 16-byte operand row that straddles the end of a symbol is cut word by
    // word, fir_load; with an odd stride the rows of the odd symbols start on an odd sample of the frame: their noise pairs take
    // two Philox blocks, their last pair holds one sample -- round 4; the LDS takes 16-byte accesses at any 4-byte alignment)
    if (firm && plain && n_fft >= 512 && (B % 2 == 0 || (mdft && WOFDM_ODD_STRIDES)) && B <= 128 * wofdm_fir8_tiles(n_fft)) return mdft ? 12 : 8;
    return (n_fft <= 256 && S % 2 == 0 && 2 * B <= 64 * wofdm_rb(n_fft, 2)) ? 2 : 1;
}

// layout of the Tx-mask variants: 9 where the matrix-pipe FIR fits (B >= n_fft, B within its tiles; every stride since round 4 --
// the rows of these layouts keep the stride itself as their pitch, so odd strides put planes on 4- and 8-byte boundaries, which
// the LDS takes at a price: CPW N = 256 masked 2.63e8 symbols/s against 2.85e8 at an even stride, 1.99e8 in layout 1), else 1
static inline int wofdm_spw_masked(int n_fft, int B, bool firm)
{
    return (firm && B >= n_fft && (B % 2 == 0 || WOFDM_ODD_STRIDES) && B <= 128 * wofdm_fir8_tiles(n_fft)) ? 9 : 1;
}

// float2 elements of noise scratch per workgroup (0: not used for this DFT length)
static inline size_t wofdm_noise_scratch_len(int n_fft, int spw)
{
    return n_fft >= WOFDM_NOISE_SCRATCH_MIN_N ? (size_t)16 * 64 * wofdm_rb(n_fft, spw) : 0;
}

// words per plane between the LDS rows of two symbols, one symbol per wave: the stride itself in the Tx-mask layouts (9, 15: the row
// holds the masked symbol as fp32 first), rounded up to whole 16-byte operand rows in layouts 8 / 12 -- every row and both of its
// planes then start on 16 bytes whatever the stride (round 4: strides of 2 mod 4 ran on 8-byte-aligned planes before, odd ones not
// at all)
static inline int wofdm_row_stride(int spw, int B) { return (spw == 8 || spw == 12) ? ((B + 3) & ~3) : B; }
// float2 elements of the frame buffer.  Matrix-pipe layouts: the same bytes hold two planes of
// packed-f16 words (hi and lo halves of every sample), each `len` words long: 24 zeros, the frame,
// and zeros up to the end of the tile that covers the trailing samples behind the last wave.
static inline int wofdm_fbuf_len(int N, int T, int spw, int S = 0, int B = 0)
{
    // (behind the last wave's first sample: its tiles and one more of zeros; layout 16: the last wave starts at (W - 1) spwr B)
    if (wofdm_is_small(spw))
        return (WOFDM_FIRM_PRE + (wofdm_waves(spw, N, S, B) - 1) * wofdm_spwr(spw, N, S, B) * B + 128 * (wofdm_firm_tiles(spw) + 1) + 3) / 4 * 4;
    if (wofdm_is_fir8(spw)) return (8 + 2 * S * wofdm_row_stride(spw, B) + 2 * WOFDM_FIR8_VT) / 2;
    if (wofdm_is_firm(spw))
        return (WOFDM_FIRM_PRE + (S - 4) * B + 128 * (wofdm_firm_tiles(spw) + 1) + 3) / 4 * 4;
    return ((WOFDM_LT - 1) + T + (WOFDM_LT - 1) + wofdm_rb(N, spw) + 8 + 1) / 2 * 2;
}
// The LDS of a CU is handed out in units of 1 280 bytes (160 KiB / 128) on this GPU -- measured, round 4: with 15 808 bytes per
// one-wave workgroup (thirteen units) a CU holds NINE workgroups, not the ten the occupancy API reports (158 080 bytes do fit
// 160 KiB), and a grid of ten per CU ran as two rounds -- nine, then one -- at the rate of five (profiles/r04_occ_small.txt);
// every other kernel's residency agrees with the same unit (three workgroups of 52 640 bytes = 3 x 42 units do fit).  The plan's
// grid counts workgroups by it.
#define WOFDM_LDS_GRANULE 1280
static inline int wofdm_lds_workgroups_per_cu(unsigned lds_bytes)
{
    const unsigned units = (lds_bytes + WOFDM_LDS_GRANULE - 1) / WOFDM_LDS_GRANULE;
    return units ? (int)(160u * 1024u / WOFDM_LDS_GRANULE / units) : 32;
}
static inline unsigned wofdm_lds_bytes(int N, int T, int spw, int S, int B)
{
    const int fixed = (wofdm_is_small(spw) ? 0 : (N == 256 || N == 512 ? 6 * 64 * 16 : 8 * N)) + 8 * N + 4 * 64 + 4 * 64 + 4 * (N + wofdm_cpcs_max(N)) + 4 * (N + 64) + 8 * 64;
    const int beta = T - S * B;
    // (layout 15, behind the fall tails: 16 bytes of alignment, a row of 344 samples per wave for the mask stage's spill, and 64 spare
    // bytes at the very end -- the target of the mask stage's stores that have no output)
    return (unsigned)(fixed + 8 * wofdm_fbuf_len(N, T, spw, S, B) + 8 * S * beta + (spw == 15 ? 16 + S * 344 * 8 + 64 : 0));
}

// kernel registry (wofdm_kernel.hip)
// constants travel as separate noalias arguments so that uniform reads become scalar loads:
// w_tx[pairs][P], w_rx[pairs][N+delta], h[n_ch][WOFDM_LT] zero padded, noise_lin[n_snr]
// fira[n_ch][4][64] (uint4): the channel's Toeplitz operands of the matrix-pipe FIR, in MFMA A layout
typedef void (*wofdm_kernel_fn)(wofdm_kparams, const float *, const float *, const float2 *,
                                const float *, const int *, const uint32_t *, const float2 *,
                                const uint4 *);
enum { WOFDM_MODE_GEN = 0, WOFDM_MODE_INJECT = 1, WOFDM_MODE_DUMP_GEN = 2, WOFDM_MODE_DUMP_INJECT = 3 };
// kernel variants: every subcarrier loaded / a subcarrier allocation mask / allocation + per-symbol
// spectral Tx mask (one symbol per wave, n_fft <= WOFDM_TXMASK_MAX_N: the mask table needs LDS)
// (direct form, g_tmask = impulse response) / the same as fast convolution (g_tmask = spectrum; n_fft
// <= WOFDM_TXFFT_MAX_N and 3P-2 <= WOFDM_TXFFT_LEN)
enum { WOFDM_VAR_PLAIN = 0, WOFDM_VAR_ALLOC = 1, WOFDM_VAR_TXMASK = 2, WOFDM_VAR_TXFFT = 3, WOFDM_VAR_COUNT };
#define WOFDM_TXMASK_MAX_N 512
#define WOFDM_TXFFT_MAX_N 256
#define WOFDM_TXFFT_LEN 1024
#define WOFDM_TXFFT_SLOTS 8
// LDS behind the frame buffer in the FFT form: twiddles + scratch rows
static inline unsigned wofdm_txfft_lds_bytes(void)
{
    return 8u * (unsigned)(WOFDM_TXFFT_LEN * (1 + WOFDM_TXFFT_SLOTS));
}
// bytes of LDS the (complex) Tx mask table takes behind the frame buffer (mask_geo in
// wofdm_kernel.hip)
static inline unsigned wofdm_txmask_lds_bytes(int n_fft)
{
    const int cpcs = wofdm_cpcs_max(n_fft);
    const int lmax = 2 * (n_fft + cpcs) - 1, no = (lmax + 63) / 64, mb = 4;
    return 8u * (unsigned)((n_fft + cpcs + 2 * mb) + 64 * no + mb);
}
// one translation unit per (DFT length, bits per subcarrier): wofdm_kernel.hip with
// -DWOFDM_TU_N=<N> -DWOFDM_TU_K=<k>
wofdm_kernel_fn wofdm_select_kernel_n64_k2(int spw, int mode, int var);
wofdm_kernel_fn wofdm_select_kernel_n64_k4(int spw, int mode, int var);
wofdm_kernel_fn wofdm_select_kernel_n64_k6(int spw, int mode, int var);
wofdm_kernel_fn wofdm_select_kernel_n128_k2(int spw, int mode, int var);
wofdm_kernel_fn wofdm_select_kernel_n128_k4(int spw, int mode, int var);
wofdm_kernel_fn wofdm_select_kernel_n128_k6(int spw, int mode, int var);
wofdm_kernel_fn wofdm_select_kernel_n256_k2(int spw, int mode, int var);
wofdm_kernel_fn wofdm_select_kernel_n256_k4(int spw, int mode, int var);
wofdm_kernel_fn wofdm_select_kernel_n256_k6(int spw, int mode, int var);
wofdm_kernel_fn wofdm_select_kernel_n512_k2(int spw, int mode, int var);
wofdm_kernel_fn wofdm_select_kernel_n512_k4(int spw, int mode, int var);
wofdm_kernel_fn wofdm_select_kernel_n512_k6(int spw, int mode, int var);
wofdm_kernel_fn wofdm_select_kernel_n1024_k2(int spw, int mode, int var);
wofdm_kernel_fn wofdm_select_kernel_n1024_k4(int spw, int mode, int var);
wofdm_kernel_fn wofdm_select_kernel_n1024_k6(int spw, int mode, int var);
static inline wofdm_kernel_fn wofdm_select_kernel(int n_fft, int bits_per_sc, int spw, int mode, int var)
{
    if (n_fft == 64 && bits_per_sc == 2) return wofdm_select_kernel_n64_k2(spw, mode, var);
    if (n_fft == 64 && bits_per_sc == 4) return wofdm_select_kernel_n64_k4(spw, mode, var);
    if (n_fft == 64 && bits_per_sc == 6) return wofdm_select_kernel_n64_k6(spw, mode, var);
    if (n_fft == 128 && bits_per_sc == 2) return wofdm_select_kernel_n128_k2(spw, mode, var);
    if (n_fft == 128 && bits_per_sc == 4) return wofdm_select_kernel_n128_k4(spw, mode, var);
    if (n_fft == 128 && bits_per_sc == 6) return wofdm_select_kernel_n128_k6(spw, mode, var);
    if (n_fft == 256 && bits_per_sc == 2) return wofdm_select_kernel_n256_k2(spw, mode, var);
    if (n_fft == 256 && bits_per_sc == 4) return wofdm_select_kernel_n256_k4(spw, mode, var);
    if (n_fft == 256 && bits_per_sc == 6) return wofdm_select_kernel_n256_k6(spw, mode, var);
    if (n_fft == 512 && bits_per_sc == 2) return wofdm_select_kernel_n512_k2(spw, mode, var);
    if (n_fft == 512 && bits_per_sc == 4) return wofdm_select_kernel_n512_k4(spw, mode, var);
    if (n_fft == 512 && bits_per_sc == 6) return wofdm_select_kernel_n512_k6(spw, mode, var);
    if (n_fft == 1024 && bits_per_sc == 2) return wofdm_select_kernel_n1024_k2(spw, mode, var);
    if (n_fft == 1024 && bits_per_sc == 4) return wofdm_select_kernel_n1024_k4(spw, mode, var);
    if (n_fft == 1024 && bits_per_sc == 6) return wofdm_select_kernel_n1024_k6(spw, mode, var);
    return nullptr;
}
hipError_t wofdm_philox_kat_launch(const uint32_t *ctr_key_dev, uint32_t *out_dev, hipStream_t s);
// closed-form ICI/ISI power kernels (wofdm_kernel.hip, k = 2 translation units)
#define WOFDM_INTERF_DECL(n)                                                                                  \
    hipError_t wofdm_interf_launch_n##n(int jobs, int P, int B, int mu, int delta, int gam, int kap, int n_ch, \
                                        const float *wtx, const float *wrx, const float2 *h, float *power,     \
                                        hipStream_t s)
WOFDM_INTERF_DECL(64);
WOFDM_INTERF_DECL(128);
WOFDM_INTERF_DECL(256);
WOFDM_INTERF_DECL(512);
WOFDM_INTERF_DECL(1024);
// Tx waveform + averaged periodogram (row f4), N <= 256 (transform length 8 N <= 2048)
#define WOFDM_PSD_DECL(n)                                                                                         \
    hipError_t wofdm_psd_launch_n##n(int P, int mu, int rho, int overlap, int no_symbols, const float *wtx,        \
                                     const float2 *X, float2 *x, int len, float *psd, hipStream_t s)
WOFDM_PSD_DECL(64);
WOFDM_PSD_DECL(128);
WOFDM_PSD_DECL(256);
