/*
 * wofdm_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, IEEE-double restatement of the reference's w-OFDM Monte-Carlo
 * link simulator:
 *     matlab/main_BER_calculation.m:230-493   (run_simulation + local functions)
 *     python/ofdm_utils/wofdm_simulation.py:85-242, 244-366, 391-418
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (libwofdm_hip.so) never links, loads or
 * calls it.
 *
 * Parity pin: the reference MATLAB path cannot be executed (no MATLAB/Octave);
 * its Python twin is imported in the build container by
 * tests/golden/make_golden.py, which emits the fixtures under tests/golden/
 * that tests/test_oracle_golden.py checks this file against.  The MATLAB-only
 * conventions (Gray labelling / bit order of Communications Toolbox qammod)
 * are "parity unpinned" by the reference -- see DESIGN.md.
 */
#ifndef WOFDM_ORACLE_H
#define WOFDM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Structure parameters of one w-OFDM system. */
typedef struct {
    int32_t n_fft;          /* N  : DFT length (power of two)                          */
    int32_t bits_per_sc;    /* k  : bits per subcarrier (2, 4, 6)                       */
    int32_t syms_per_frame; /* S  : symbolsPerTx, symbol 0 is the pilot                 */
    int32_t cp;             /* mu : cyclic prefix                                       */
    int32_t cs;             /* rho: cyclic suffix                                       */
    int32_t tail_tx;        /* beta : Tx window tail / inter-symbol overlap             */
    int32_t tail_rx;        /* delta: Rx window tail (even)                             */
    int32_t prefix_rm;      /* gamma: samples dropped before the Rx window              */
    int32_t circ_shift;     /* kappa: circular shift after the Rx fold                  */
    int32_t n_taps;         /* L  : channel taps                                        */
    int32_t noise_before_truncate; /* 1 = MATLAB order (main_BER_calculation.m:260-261),
                                      0 = Python order (wofdm_simulation.py:208-215)    */
    /* main_channel_mask.m variant (both NULL = main_BER_calculation.m):                */
    const uint8_t *active;  /* [N] non-zero = bin carries data (m:387-390 zero padding +
                               ifftshift; Rx selection m:367-369); NULL = all bins      */
    const double *tx_mask;  /* [2P-1] DFT-domain gains, natural bin order, of the
                               per-symbol spectral mask dft_rc_filt (m:398-417);
                               NULL = no mask                                           */
} wofdm_oracle_sys;

/* Optional per-frame stage dumps (any pointer may be NULL). Complex arrays are
 * interleaved (re, im) doubles. */
typedef struct {
    double  *X;        /* [S][N]      mapped QAM symbols                                  */
    double  *tx;       /* [beta+S*B]  serialised on-air signal                            */
    double  *conv;     /* [T+L-1]     channel output (before noise)                       */
    double  *rx;       /* [S*B]       noisy, truncated received samples                   */
    double  *Y;        /* [S][N]      Rx DFT output                                       */
    double  *Xhat;     /* [S-1][N]    equalised symbols 1..S-1                            */
    uint8_t *labels_rx;/* [S-1][N]    decided labels                                      */
    double  *gain;     /* [1]         noise scale sqrt(Ps*10^(-snr/10)/Pn)                */
} wofdm_oracle_dump;

/* Sizes: P = N+cp+cs, B = P-tail_tx, T = tail_tx+S*B,
 * noise_len = noise_before_truncate ? T+L-1 : S*B. */
int wofdm_oracle_noise_len(const wofdm_oracle_sys *sys);

/* MATLAB qammod(...,'InputType','bit','UnitAveragePower',true) Gray table for
 * M = 2^k (k = 2, 4, 6): table[label] = (re, im).  label bit k-1 (MSB) is the
 * first bit of the subcarrier (main_BER_calculation.m:246-249). */
int wofdm_oracle_qam_table(int bits_per_sc, double *table /* [M][2] */);

/* One frame with explicit randomness.
 *   labels      [S][N]  constellation indices
 *   unit_noise  [noise_len][2]  N(0,1) draws (re, im) before scaling
 *   qam_table   [M][2] or NULL (= MATLAB Gray table)
 *   nearest     0: per-axis slicer on the MATLAB table (qamdemod);
 *               1: argmin |table - x| with first-minimum tie break
 *                  (wofdm_simulation.py:142-166 `decision`)
 * counts[4] += {bit errors, bits, symbol errors, symbols} over symbols 1..S-1. */
int wofdm_oracle_frame(const wofdm_oracle_sys *sys,
                       const double *w_tx /* [P] */, const double *w_rx /* [N+delta] */,
                       const double *h /* [L][2] */, double snr_db,
                       const double *qam_table, int nearest,
                       const uint8_t *labels, const double *unit_noise,
                       uint64_t counts[4], wofdm_oracle_dump *dump);

/* Philox4x32-10 (Salmon et al., SC'11).  out[4] = philox(ctr[4], key[2]). */
void wofdm_oracle_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* Draws of the build's on-device RNG stream definition (DESIGN.md "RNG"):
 * labels[S][N] of (seed, cell, frame) and unit_noise[noise_len][2]. */
void wofdm_oracle_gen_labels(const wofdm_oracle_sys *sys, uint64_t seed, uint32_t cell,
                             uint64_t frame, uint8_t *labels);
void wofdm_oracle_gen_noise(const wofdm_oracle_sys *sys, uint64_t seed, uint32_t cell,
                            uint64_t frame, double *unit_noise);

/* Generate-mode sweep: cells = pairs x snr x channels, frames
 * [frame_offset, frame_offset+frames_per_cell) of every cell.
 * counts[pairs][n_snr][n_channels][4] accumulated into.  n_threads <= 0 -> all
 * cores (OpenMP).  Returns 0 or a negative error. */
int wofdm_oracle_run(const wofdm_oracle_sys *sys, int n_pairs, int n_snr, int n_channels,
                     const double *w_tx /* [pairs][P] */, const double *w_rx /* [pairs][N+delta] */,
                     const double *h /* [n_channels][L][2] */, const double *snr_db /* [n_snr] */,
                     uint64_t seed, uint64_t frame_offset, uint64_t frames_per_cell,
                     int n_threads, uint64_t *counts);

/* The same sweep with the reference's own cost structure ("faithful" CPU leg of bench.py): Tx and Rx as
 * dense matrix products with the operators hoisted out of the frame loop, as the Python reference does
 * (wofdm_simulation.py:464-471, 187-189, 219-222), instead of FFTs.  Results equal wofdm_oracle_run's to
 * fp64 rounding (tests/test_oracle_golden.py). */
int wofdm_oracle_run_dense(const wofdm_oracle_sys *sys, int n_pairs, int n_snr, int n_channels,
                           const double *w_tx, const double *w_rx, const double *h, const double *snr_db,
                           uint64_t seed, uint64_t frame_offset, uint64_t frames_per_cell,
                           int n_threads, uint64_t *counts);

/* Plain DFT helpers exported so tests can check the oracle's FFT against
 * numpy: dir = -1 forward (no scale), +1 inverse (scaled by 1/N). */
void wofdm_oracle_fft(int n, int dir, double *x /* [n][2] in place */);

int wofdm_oracle_threads(void);

#ifdef __cplusplus
}
#endif
#endif
