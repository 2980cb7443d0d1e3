/*
 * wofdm.h -- C ABI of libwofdm_hip.so, the MI355X (gfx950) implementation of
 * the w-OFDM Monte-Carlo BER hot path.
 *
 * The reference has no FFI: the hot path is a MATLAB local function and a
 * Python method.  The entry points below are what a binding for that path
 * replaces:
 *
 *   wofdm_run              <- ber = run_simulation(ensemble, symbolsPerTx, bitsPerSubcarrier,
 *                                numSubcar, cpLength, csLength, windowTx, channel, snr, tailTx,
 *                                tailRx, windowRx, prefixRemovalLength, circularShiftLength)
 *                             matlab/main_BER_calculation.m:230-274, batched over the
 *                             (window pair x SNR x channel) loop nest of lines 64-201;
 *                          <- wOFDMSystem.run_simulation(channel_models, window_tx, window_rx,
 *                                ensemble, snr_arr, no_symbols)
 *                             python/ofdm_utils/wofdm_simulation.py:432-481 (loop nest 168-242)
 *   wofdm_plan_*           <- the same, split into "upload constants once" + "launch a frame
 *                             range", so a driver can shard frames over GPUs and keep the
 *                             constants resident in HBM
 *   wofdm_*_injected       <- the same frame pipeline with the random draws
 *                             (main_BER_calculation.m:246,290 / wofdm_simulation.py:136,183)
 *                             supplied by the caller -- parity / HBM-streaming mode
 *
 * Conventions: plain pointers and sizes only.  `*_dev` pointers are device (HBM) addresses
 * on the plan's GPU, all others are host addresses.  All pointers are caller-owned and not
 * retained after the call returns (plans copy what they need).  Every function returns
 * WOFDM_OK (0) or a negative WOFDM_E_* code; wofdm_last_error() gives the message of the
 * last failure on the calling thread.  There is no CPU fallback: without a usable gfx950
 * device every compute entry point fails with WOFDM_E_HIP.
 *
 * A *cell* is one (window pair, SNR, channel) triple, cell = (pair*n_snr + snr)*n_channels
 * + channel.  counts[cell][4] = {bit errors, bits, symbol errors, symbols} over the data
 * symbols 1..S-1 of every frame (symbol 0 is the pilot, main_BER_calculation.m:250,266-268);
 * counters are ACCUMULATED into, never reset by a launch.
 */
#ifndef WOFDM_H
#define WOFDM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WOFDM_ABI_VERSION 1

#define WOFDM_OK            0
#define WOFDM_E_INVALID    -1   /* NULL pointer / inconsistent lengths                  */
#define WOFDM_E_UNSUPPORTED -2  /* N, k, S, taps or tails outside the built kernels     */
#define WOFDM_E_HIP        -3   /* HIP runtime error (no device, launch failure, ...)   */
#define WOFDM_E_NOMEM      -4

#define WOFDM_MAX_TAPS     21   /* taps of the reference's 200 ns Veh-A lines           */
#define WOFDM_MAX_SYMS     16   /* symbolsPerTx (matlab/window_optimization.m:39-47)    */

typedef struct wofdm_cfg {
    int32_t  n_fft;            /* N: 64, 128, 256, 512, 1024                                */
    int32_t  bits_per_sc;      /* k: 2, 4, 6 (QPSK, 16-QAM, 64-QAM, MATLAB Gray labels)     */
    int32_t  syms_per_frame;   /* S: 2..16, symbol 0 = pilot                                */
    int32_t  cp, cs;           /* mu, rho                                                   */
    int32_t  tail_tx, tail_rx; /* beta, delta (delta even)                                  */
    int32_t  prefix_rm;        /* gamma; N + delta + gamma == N + mu + rho - beta required  */
    int32_t  circ_shift;       /* kappa                                                     */
    int32_t  n_taps;           /* L <= WOFDM_MAX_TAPS                                       */
    int32_t  n_channels, n_snr, n_window_pairs;
    int32_t  noise_before_truncate; /* 1: MATLAB order (m:260-261); 0: Python (py:208-215)  */
    uint64_t frames_per_cell;  /* frames [frame_offset, frame_offset+frames_per_cell)       */
    uint64_t frame_offset;     /*   of every cell (global frame index keys the RNG)         */
    uint64_t seed;
} wofdm_cfg;

/* Optional stage dump of ONE frame (host float buffers, complex = interleaved re,im;
 * any pointer may be NULL).  Same stages as the oracle's dump. */
typedef struct wofdm_dump {
    uint8_t *labels_tx;  /* [S][N]                                      */
    float   *X;          /* [S][N][2]                                   */
    float   *tx;         /* [beta + S*B][2]                             */
    float   *conv;       /* [T + L - 1][2]                              */
    float   *rx;         /* [S*B][2]                                    */
    float   *Y;          /* [S][N][2]  (the kernels leave circular_shift, m:313-333, to the equaliser, which
                          * divides the resulting per-subcarrier phase out; the dump puts that phase,
                          * e^{+2 pi i (kappa + delta/2) n / N}, back on the host, so this IS the reference's Y) */
    float   *Xhat;       /* [S-1][N][2]                                 */
    uint8_t *labels_rx;  /* [S-1][N]                                    */
    float   *gain;       /* [1]                                         */
    float   *unit_noise; /* [noise_len][2] the unit normals the kernel used */
} wofdm_dump;

typedef struct wofdm_plan wofdm_plan;

int         wofdm_version(void);
int         wofdm_device_count(void);            /* < 0 on HIP failure                    */
const char *wofdm_last_error(void);
int         wofdm_noise_len(const wofdm_cfg *cfg);/* unit-noise samples per frame         */

/* Upload windows [pairs][P] / [pairs][N+delta], channels [n_channels][L][2] and SNR points
 * [n_snr] (dB) to `device` and select the kernel.  frames_per_cell / frame_offset of cfg are
 * ignored here (given per launch). */
int wofdm_plan_create(wofdm_plan **plan, const wofdm_cfg *cfg, int device,
                      const float *w_tx, const float *w_rx, const float *h,
                      const float *snr_db);
int wofdm_plan_destroy(wofdm_plan *plan);

/* Asynchronous launch on `stream` (a hipStream_t, NULL = default stream): simulate frames
 * [frame_offset, frame_offset+frames_per_cell) of every cell with the on-device Philox4x32-10
 * streams and add into counts_dev[cells][4] (uint64, device memory of the plan's GPU).
 * EXCLUSIVE USE OF THE DEVICE while a launch is in flight.  The default kernels (transforms on the matrix
 * pipe, every N) issue MFMAs in a rhythm that, on MI355X, was MEASURED to corrupt op_sel-swizzled packed fp32
 * arithmetic (v_pk_*_f32 with an op_sel source swizzle) of OTHER waves on the same SIMD (DESIGN.md section 4,
 * hazards 1 and 4: microbenchmarks of this repository, asserted in its GPU tests; no vendor erratum is known to
 * us -- unconfirmed outside these measurements).  They contain no such instruction themselves.  What the library
 * does: every launch waits (on the device, through an event) for the previous launch this process made on that
 * GPU, whatever plan or stream; the synchronous entry points that run other kernels (wofdm_interference,
 * wofdm_tx_psd) hold the same gate.  What the CALLER must ensure: no kernel of its own (other libraries, other
 * streams) and no other process runs on the device while a frame launch is in flight -- one process per GPU,
 * synchronise before handing the GPU to other work (bench.py, distributed.py do).  Where that cannot be
 * guaranteed (a shared GPU), select the VALU transforms, wofdm_plan_set_option(plan, WOFDM_OPT_DFT_VALU, 1): those
 * kernels issue only single, cache-line-aligned six-MFMA chains, the shape measured to be harmless. */
int wofdm_plan_launch(wofdm_plan *plan, uint64_t frame_offset, uint64_t frames_per_cell,
                      uint64_t *counts_dev, void *stream);

/* Same, bracketed by HIP events on `stream`; blocks until the kernel finished and returns
 * its duration in milliseconds. */
int wofdm_plan_launch_timed(wofdm_plan *plan, uint64_t frame_offset, uint64_t frames_per_cell,
                            uint64_t *counts_dev, void *stream, float *kernel_ms);

/* Injected randomness, device buffers: labels_dev[cells][frames][S][N] (uint8),
 * unit_noise_dev[cells][frames][noise_len][2] (float, N(0,1) per component). */
int wofdm_plan_launch_injected(wofdm_plan *plan, uint64_t frames_per_cell,
                               const uint8_t *labels_dev, const float *unit_noise_dev,
                               uint64_t *counts_dev, void *stream);

/* Run ONE frame of `cell` on the GPU and copy its intermediate stages to host buffers.
 * labels / unit_noise: host arrays to inject, or NULL to use the Philox streams of
 * (seed, cell, frame).  counts[4] (host) accumulated into. */
int wofdm_plan_dump_frame(wofdm_plan *plan, uint32_t cell, uint64_t frame,
                          const uint8_t *labels, const float *unit_noise,
                          uint64_t *counts, wofdm_dump *out);

/* Subcarrier allocation: active[n_fft] (host), non-zero = bin n carries data.  Unloaded bins
 * transmit zero and are left out of the channel estimate and of all four counters, which then
 * count (S-1) * n_active subcarriers per frame.  NULL restores "every bin loaded".  Replaces the
 * zero-padding `symbolsInOFDM = [zeros(S,offset) transmittedSymbols zeros(S,offset)]` + ifftshift of
 * matlab/main_channel_mask.m:387-390 and the `offset+1:end-offset` selections of 367-369 (there:
 * bins [0,N/4) and [3N/4,N)); also the guard-band `subcar_alloc_mat` of
 * python/ofdm_utils/timefreq_simulation.py:223-233.  The data-bit stream keeps one slot per bin.
 * Synchronises the device; do not call while launches of this plan are in flight elsewhere. */
int wofdm_plan_set_allocation(wofdm_plan *plan, const uint8_t *active);

/* Per-symbol spectral Tx mask: mask[2P-1] (host), DFT-domain gains in natural bin order, P =
 * n_fft+cp+cs.  Every windowed symbol is zero-padded to 2P-1 samples, its DFT multiplied by the
 * mask and transformed back; the first P samples replace the symbol, the remaining P-1 are added
 * to the first P-1 samples of the NEXT symbol's row before the overlap-add (the last symbol's
 * spill is dropped).  Replaces `dft_rc_filt` (matlab/main_channel_mask.m:398-417; its mask is
 * `ifftshift(gen_raised_cosine(floor((2P-1)/2), rollOff, 2P-1))`, 402-405, 443-458).  Needs
 * n_fft <= 512 (the mask's impulse response is staged in LDS), else WOFDM_E_UNSUPPORTED.  NULL
 * removes the mask.  Synchronises the device. */
int wofdm_plan_set_tx_mask(wofdm_plan *plan, const float *mask);

/* Health of the plan's finished launches (call after synchronising the stream; copies one
 * word back): WOFDM_OK, or WOFDM_E_HIP if a kernel reported that a wave gave up waiting for its
 * workgroup -- the counters of that plan are then not to be used.  The synchronous entry points
 * (wofdm_run, wofdm_run_injected, wofdm_plan_launch_timed) check it themselves. */
int wofdm_plan_status(wofdm_plan *plan);

/* Kernel resource facts of the plan: {waves per workgroup, LDS bytes per workgroup,
 * workgroups launched, workgroups resident per CU (occupancy API, capped by the LDS allocation units: DESIGN.md section 3), CUs}. */
int wofdm_plan_info(wofdm_plan *plan, int32_t info[5]);

/* Which instantiation of the frame kernel the plan launches: {layout id, variant}.  Layout: 1, 2 =
 * one / two symbols per wave with the FIR on the VALU; 4, 5 = four symbols per wave (N = 256), FIR on
 * the VALU; 6, 7 = the same with the FIR on the matrix pipe; 9 = one symbol per wave with the FIR on the matrix pipe (Tx-mask
 * variants); 10, 11 = 6, 7 with both 256-point transforms on
 * the matrix pipe as well; 8 = one symbol per wave (N >= 512), FIR on the matrix pipe; 12 = 8 with both transforms on the
 * matrix pipe; 13, 14 = N = 64 / 128, sixteen / eight symbols per wave, FIR and transforms on the matrix pipe; 16 = 13 with a run-time
 * number of symbols per wave and a partly filled last wave (no_symbols not a multiple of 16 / 8, long strides); 15 = 9 at N = 256 with
 * the fast-convolution Tx mask: the symbol's transforms and the mask's two 1024-point transforms on the matrix pipe as well.  Variant: 0 plain, 1 subcarrier allocation, 2 / 3 = Tx mask in direct / fast-
 * convolution form.  (Test and profiling aid; the results do not depend on it beyond fp32 rounding.) */
int wofdm_plan_kernel_id(wofdm_plan *plan, int32_t id[2]);

/* Diagnostic choice among the kernels of the family (A/B measurements and the tests; the results do not depend
 * on it beyond fp32 rounding, the defaults are the fastest kernels).  Takes effect for the launches that follow;
 * returns WOFDM_E_UNSUPPORTED -- and leaves the plan as it was -- when no kernel fits the geometry under the option.
 *   WOFDM_OPT_FIR_VALU       1 = the 21-tap FIR (conv, main_BER_calculation.m:260) on the VALU in every layout
 *                            (round-1 kernels) instead of the matrix pipe; 0 = default
 *   WOFDM_OPT_MAX_SPW        at most 1, 2 or 4 OFDM symbols per wavefront; 0 = default (the most that fits)
 *   WOFDM_OPT_TXMASK_DIRECT  1 = the Tx mask (wofdm_plan_set_tx_mask) always as a direct-form convolution instead
 *                            of fast convolution where that fits; 0 = default
 *   WOFDM_OPT_DFT_VALU       1 = the 256-point IDFT / DFT (dftmtx, main_BER_calculation.m:306, 370) as in-register radix-16
 *                            stages on the VALU (layouts 6, 7) instead of split-f16 products on the matrix pipe (layouts
 *                            10, 11); likewise n_fft = 512, 1024 (8 instead of 12), 64, 128 (2 instead of 13, 14, 16) and the Tx-mask
 *                            kernel at n_fft = 256 (9 instead of 15); 0 = default */
#define WOFDM_OPT_FIR_VALU       0
#define WOFDM_OPT_MAX_SPW        1
#define WOFDM_OPT_TXMASK_DIRECT  2
#define WOFDM_OPT_DFT_VALU       3
int wofdm_plan_set_option(wofdm_plan *plan, int32_t option, int32_t value);

/* One-shot, host pointers in / host counters out (synchronous):
 * counts[pairs][n_snr][n_channels][4] accumulated into. */
int wofdm_run(const wofdm_cfg *cfg, int device, const float *w_tx, const float *w_rx,
              const float *h, const float *snr_db, uint64_t *counts);
int wofdm_run_injected(const wofdm_cfg *cfg, int device, const float *w_tx, const float *w_rx,
                       const float *h, const float *snr_db, const uint8_t *labels,
                       const float *unit_noise, uint64_t *counts);

/* Closed-form ICI + ISI power per subcarrier of the structure in cfg, for every (window pair, channel):
 * power[pairs][n_channels][n_fft] (host, float) = sum_{n' != n} |A_0[n,n']|^2 + sum_{n'} |A_1[n,n']|^2 with
 * A_m = W K P V_rx R H_m V_tx Gamma W^-1.  Replaces calculate_interference
 * (matlab/main_interference_calculation.m:177-225; its scalar is the sum over n) and interf_power
 * (python/ofdm_utils/interf_calc.py:20-113; its np.diag(PISI + PICI1) is power[0][0][:]).  Uses n_fft, cp,
 * cs, tail_tx, tail_rx, prefix_rm, circ_shift, n_taps, n_channels, n_window_pairs of cfg.
 * Synchronous; host pointers. */
int wofdm_interference(const wofdm_cfg *cfg, int device, const float *w_tx, const float *w_rx,
                       const float *h, float *power);

/* Tx-side spectrum estimate: the waveform of no_symbols consecutive symbols X[no_symbols][n_fft][2] (host,
 * complex values on the bins, zeros on unloaded ones) through IDFT, CP/CS copy, Tx window w_tx[P] and the
 * overlap-add of `overlap` tail samples (tail_tx for the Tx-windowed structures, 0 otherwise), then the sum
 * over consecutive slices of 8 n_fft samples (zero-padded remainder included) of |FFT|^2, fftshift-ed:
 * psd[8 n_fft] (host).  Replaces the Tx chain and psd_estimate of wOFDMSystem.estimate_obr
 * (python/ofdm_utils/timefreq_simulation.py:242-258, 101-123); the caller divides by the reference's slice
 * count (full slices + 1).  Uses n_fft, cp, cs of cfg; n_fft in {64, 128, 256}.  Synchronous. */
int wofdm_tx_psd(const wofdm_cfg *cfg, int device, const float *w_tx, const float *X, int no_symbols,
                 int overlap, float *psd);

/* Philox4x32-10 known-answer hook (runs one block on the GPU). */
int wofdm_philox_kat(int device, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif
