/*
 * wofdm_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See wofdm_oracle.h for scope and the parity pin.  IEEE double throughout.
 *
 * Every function cites the reference lines it restates
 * (paths relative to /root/reference).
 */
#include "wofdm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ sizes */

static int sys_P(const wofdm_oracle_sys *s) { return s->n_fft + s->cp + s->cs; }
static int sys_B(const wofdm_oracle_sys *s) { return sys_P(s) - s->tail_tx; }
static int sys_T(const wofdm_oracle_sys *s) { return s->tail_tx + s->syms_per_frame * sys_B(s); }

int wofdm_oracle_noise_len(const wofdm_oracle_sys *s)
{
    /* MATLAB: noise over the full convolution, main_BER_calculation.m:260,290.
     * Python: noise over the truncated signal, wofdm_simulation.py:208-215,136. */
    return s->noise_before_truncate ? sys_T(s) + s->n_taps - 1
                                    : s->syms_per_frame * sys_B(s);
}

static int sys_check(const wofdm_oracle_sys *s)
{
    int n = s->n_fft;
    if (n < 2 || (n & (n - 1))) return -1;
    if (s->bits_per_sc != 2 && s->bits_per_sc != 4 && s->bits_per_sc != 6) return -2;
    if (s->syms_per_frame < 2) return -3;
    if (s->cp < 0 || s->cs < 0 || s->tail_tx < 0 || s->tail_rx < 0 || (s->tail_rx & 1)) return -4;
    if (s->prefix_rm < 0 || s->circ_shift < 0 || s->circ_shift >= n || s->n_taps < 1) return -5;
    /* B = N + delta + gamma must hold (main_BER_calculation.m:262-263 reshape) */
    if (sys_B(s) != n + s->tail_rx + s->prefix_rm) return -6;
    if (2 * s->tail_tx > sys_P(s) || s->tail_rx > n) return -7;
    return 0;
}

/* -------------------------------------------------------------------- FFT */

/* Iterative radix-2, in place.  dir=-1: Y[n] = sum_t z[t] e^{-j2pi nt/N}
 * (dftmtx(N), main_BER_calculation.m:306; receiver.py:127-131).
 * dir=+1: x[t] = (1/N) sum_n X[n] e^{+j2pi nt/N}
 * (dftmtx(N)'/N, main_BER_calculation.m:370; transmitter.py:53-58). */
void wofdm_oracle_fft(int n, int dir, double *x)
{
    /* per-thread twiddle table exp(-2 pi i k / n), k < n/2 (rebuilt when n changes) */
    static __thread double *tw = NULL;
    static __thread int tw_n = 0;
    int i, j, len;
    if (tw_n != n) {
        free(tw);
        tw = (double *)malloc(sizeof(double) * (size_t)(n > 1 ? n : 2));
        for (i = 0; i < n / 2; i++) {
            tw[2 * i] = cos(-2.0 * M_PI * (double)i / (double)n);
            tw[2 * i + 1] = sin(-2.0 * M_PI * (double)i / (double)n);
        }
        tw_n = n;
    }
    for (i = 1, j = 0; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double tr = x[2 * i], ti = x[2 * i + 1];
            x[2 * i] = x[2 * j]; x[2 * i + 1] = x[2 * j + 1];
            x[2 * j] = tr; x[2 * j + 1] = ti;
        }
    }
    for (len = 2; len <= n; len <<= 1) {
        int half = len >> 1, k, step = n / len;
        for (i = 0; i < n; i += len) {
            for (k = 0; k < half; k++) {
                double wr = tw[2 * k * step], wi = dir > 0 ? -tw[2 * k * step + 1] : tw[2 * k * step + 1];
                double *a = x + 2 * (i + k), *b = x + 2 * (i + k + half);
                double tr = b[0] * wr - b[1] * wi, ti = b[0] * wi + b[1] * wr;
                b[0] = a[0] - tr; b[1] = a[1] - ti;
                a[0] += tr; a[1] += ti;
            }
        }
    }
    if (dir > 0) {
        double inv = 1.0 / (double)n;
        for (i = 0; i < 2 * n; i++) x[i] *= inv;
    }
}

/* -------------------------------------------------------------------- QAM */

static unsigned gray_dec(unsigned g) { unsigned b = g; while (g >>= 1) b ^= g; return b; }
static unsigned gray_enc(unsigned b) { return b ^ (b >> 1); }

/* Communications Toolbox qammod, 'gray' (default) symbol order, bit input,
 * unit average power -- call site main_BER_calculation.m:248-249.  Source is
 * not in the reference; convention restated from the toolbox documentation
 * (SURVEY.md 3.4-2): upper k/2 bits Gray-select I ascending from -(m-1),
 * lower k/2 bits Gray-select Q descending from +(m-1). */
int wofdm_oracle_qam_table(int k, double *table)
{
    int M, m, half, l;
    double scale;
    if (k != 2 && k != 4 && k != 6) return -2;
    M = 1 << k; half = k / 2; m = 1 << half;
    scale = 1.0 / sqrt(2.0 * (M - 1) / 3.0);
    for (l = 0; l < M; l++) {
        unsigned hi = (unsigned)l >> half, lo = (unsigned)l & (unsigned)(m - 1);
        table[2 * l]     = scale * (-(m - 1) + 2.0 * gray_dec(hi));
        table[2 * l + 1] = scale * ( (m - 1) - 2.0 * gray_dec(lo));
    }
    return 0;
}

/* qamdemod(...,'OutputType','bit','UnitAveragePower',true), hard decision --
 * call site main_BER_calculation.m:269-270. */
static unsigned slice_label(int k, double re, double im)
{
    int half = k / 2, m = 1 << half, M = 1 << k, ii, qi;
    double a = sqrt(2.0 * (M - 1) / 3.0);
    ii = (int)floor((re * a + (m - 1)) * 0.5 + 0.5);
    qi = (int)floor(((m - 1) - im * a) * 0.5 + 0.5);
    if (ii < 0) ii = 0;
    if (ii > m - 1) ii = m - 1;
    if (qi < 0) qi = 0;
    if (qi > m - 1) qi = m - 1;
    return (gray_enc((unsigned)ii) << half) | gray_enc((unsigned)qi);
}

/* wofdm_simulation.py:142-166: argmin(abs(symbols - x)), first minimum. */
static unsigned nearest_label(int M, const double *table, double re, double im)
{
    int l, best = 0;
    double bd = 0.0;
    for (l = 0; l < M; l++) {
        double d = hypot(table[2 * l] - re, table[2 * l + 1] - im);
        if (l == 0 || d < bd) { bd = d; best = l; }
    }
    return (unsigned)best;
}

/* ------------------------------------------------------------------ frame */

/* tx_mat [P][N] and rx_mat [N][B] (complex, row major) non-NULL: the "faithful" cost structure of
 * the Python reference, which hoists the two dense products W_tx R IDFT and DFT K P V_rx R out of
 * the frame loop (wofdm_simulation.py:464-471) and applies them as matrix products in every frame
 * (187-189, 219-222) instead of FFTs.  Same results to fp64 rounding. */
static int frame_impl(const wofdm_oracle_sys *sys,
                      const double *w_tx, const double *w_rx,
                      const double *h, double snr_db,
                      const double *qam_table, int nearest,
                      const uint8_t *labels, const double *unit_noise,
                      uint64_t counts[4], wofdm_oracle_dump *dump,
                      const double *tx_mat, const double *rx_mat)
{
    const int N = sys->n_fft, k = sys->bits_per_sc, S = sys->syms_per_frame;
    const int mu = sys->cp, delta = sys->tail_rx;
    const int gam = sys->prefix_rm, kap = sys->circ_shift, L = sys->n_taps;
    const int M = 1 << k;
    int P, B, T, CL, NL, s, i, j, l, t, n, rc;
    double table_buf[2 * 64];
    double *X, *tx, *c, *r, *z, *Y, *xs;
    double Ps = 0.0, Pn = 0.0, g;
    uint64_t bit_err = 0, sym_err = 0;

    if ((rc = sys_check(sys)) != 0) return rc;
    P = sys_P(sys); B = sys_B(sys); T = sys_T(sys);
    CL = T + L - 1; NL = wofdm_oracle_noise_len(sys);
    if (!qam_table) { wofdm_oracle_qam_table(k, table_buf); qam_table = table_buf; }

    {   /* one per-thread workspace, grown on demand (no malloc traffic in the frame loop) */
        static __thread double *ws = NULL;
        static __thread size_t ws_len = 0;
        size_t need = 2 * ((size_t)2 * S * N + T + CL + (size_t)S * B + 2 * (size_t)N);
        if (ws_len < need) {
            free(ws);
            ws = (double *)malloc(sizeof(double) * need);
            ws_len = ws ? need : 0;
        }
        if (!ws) return -100;
        X = ws; Y = X + 2 * (size_t)S * N; tx = Y + 2 * (size_t)S * N; c = tx + 2 * (size_t)T;
        r = c + 2 * (size_t)CL; xs = r + 2 * (size_t)S * B; z = xs + 2 * (size_t)N;
        memset(tx, 0, sizeof(double) * 2 * (size_t)T);
    }

    /* qammod, main_BER_calculation.m:248-249 (Python: np.random.choice of the
     * alphabet, wofdm_simulation.py:179-186) */
    for (s = 0; s < S; s++)
        for (n = 0; n < N; n++) {
            unsigned lab = labels[(size_t)s * N + n];
            if (lab >= (unsigned)M) { rc = -8; goto done; }
            X[2 * ((size_t)s * N + n)]     = qam_table[2 * lab];
            X[2 * ((size_t)s * N + n) + 1] = qam_table[2 * lab + 1];
            if (sys->active && !sys->active[n]) {      /* main_channel_mask.m:387-390 */
                X[2 * ((size_t)s * N + n)] = 0.0;
                X[2 * ((size_t)s * N + n) + 1] = 0.0;
            }
        }

    /* wofdm_tx, main_BER_calculation.m:358-376: IDFT (1/N), add_redundancy
     * (419-439: CP = last mu samples in front, CS = first rho samples behind),
     * diagonal Tx window; then the overlap-add + serialise of 253-259
     * (wofdm_simulation.py:187-203): symbol s starts at s*B. */
    if (!sys->tx_mask && tx_mat) {
        /* tx_mat @ X, one symbol (column) at a time (wofdm_simulation.py:187-189) */
        for (s = 0; s < S; s++) {
            const double *x = X + 2 * (size_t)s * N;
            for (i = 0; i < P; i++) {
                const double *row = tx_mat + 2 * (size_t)i * N;
                double ar = 0.0, ai = 0.0;
                for (n = 0; n < N; n++) {
                    ar += row[2 * n] * x[2 * n] - row[2 * n + 1] * x[2 * n + 1];
                    ai += row[2 * n] * x[2 * n + 1] + row[2 * n + 1] * x[2 * n];
                }
                tx[2 * (s * B + i)] += ar;
                tx[2 * (s * B + i) + 1] += ai;
            }
        }
    } else if (!sys->tx_mask) {
        for (s = 0; s < S; s++) {
            memcpy(xs, X + 2 * (size_t)s * N, sizeof(double) * 2 * N);
            wofdm_oracle_fft(N, +1, xs);
            for (i = 0; i < P; i++) {
                int src = ((i - mu) % N + N) % N;
                tx[2 * (s * B + i)]     += w_tx[i] * xs[2 * src];
                tx[2 * (s * B + i) + 1] += w_tx[i] * xs[2 * src + 1];
            }
        }
    } else {
        /* main_channel_mask.m:381-417: the windowed symbols (rows of P samples) go through
         * dft_rc_filt before the overlap-add: zero-pad to Lm = 2P-1, DFT, multiply by the mask,
         * inverse DFT; samples [0,P) stay in the row, samples [P,2P-1) are added to the first
         * P-1 samples of the NEXT row (filterTail, 411-415); the last row's spill is dropped. */
        const int Lm = 2 * P - 1;
        double *cs = (double *)malloc(sizeof(double) * 2 * (size_t)Lm);      /* e^{-2 pi i q / Lm} */
        double *row = (double *)malloc(sizeof(double) * 2 * (size_t)P);
        double *spec = (double *)malloc(sizeof(double) * 2 * (size_t)Lm);
        double *rows = (double *)calloc((size_t)(S + 1) * P * 2, sizeof(double));
        int q, kk;
        if (!cs || !row || !spec || !rows) { free(cs); free(row); free(spec); free(rows); rc = -100; goto done; }
        for (q = 0; q < Lm; q++) {
            cs[2 * q] = cos(2.0 * M_PI * q / Lm); cs[2 * q + 1] = -sin(2.0 * M_PI * q / Lm);
        }
        for (s = 0; s < S; s++) {
            memcpy(xs, X + 2 * (size_t)s * N, sizeof(double) * 2 * N);
            wofdm_oracle_fft(N, +1, xs);
            for (i = 0; i < P; i++) {
                int src = ((i - mu) % N + N) % N;
                row[2 * i] = w_tx[i] * xs[2 * src]; row[2 * i + 1] = w_tx[i] * xs[2 * src + 1];
            }
            for (kk = 0; kk < Lm; kk++) {               /* DFT of the zero-padded row x mask */
                double ar = 0.0, ai = 0.0;
                for (i = 0; i < P; i++) {
                    const double *w = cs + 2 * (int)(((long)kk * i) % Lm);
                    ar += row[2 * i] * w[0] - row[2 * i + 1] * w[1];
                    ai += row[2 * i] * w[1] + row[2 * i + 1] * w[0];
                }
                spec[2 * kk] = ar * sys->tx_mask[kk]; spec[2 * kk + 1] = ai * sys->tx_mask[kk];
            }
            for (i = 0; i < Lm; i++) {                  /* inverse DFT, 1/Lm */
                double ar = 0.0, ai = 0.0;
                double *dst;
                for (kk = 0; kk < Lm; kk++) {
                    const double *w = cs + 2 * (int)(((long)kk * i) % Lm);
                    ar += spec[2 * kk] * w[0] + spec[2 * kk + 1] * w[1];    /* conj twiddle */
                    ai += spec[2 * kk + 1] * w[0] - spec[2 * kk] * w[1];
                }
                dst = (i < P) ? rows + 2 * ((size_t)s * P + i) : rows + 2 * ((size_t)(s + 1) * P + (i - P));
                dst[0] += ar / Lm; dst[1] += ai / Lm;
            }
        }
        for (s = 0; s < S; s++)                          /* overlap-add + serialise, m:426-432 */
            for (i = 0; i < P; i++) {
                tx[2 * (s * B + i)]     += rows[2 * ((size_t)s * P + i)];
                tx[2 * (s * B + i) + 1] += rows[2 * ((size_t)s * P + i) + 1];
            }
        free(cs); free(row); free(spec); free(rows);
    }

    /* conv(channel, transmittedSignal), main_BER_calculation.m:260
     * (np.convolve, wofdm_simulation.py:205-207) */
    for (j = 0; j < CL; j++) {
        double ar = 0.0, ai = 0.0;
        for (l = 0; l < L; l++) {
            int q = j - l;
            if (q < 0 || q >= T) continue;
            ar += h[2 * l] * tx[2 * q] - h[2 * l + 1] * tx[2 * q + 1];
            ai += h[2 * l] * tx[2 * q + 1] + h[2 * l + 1] * tx[2 * q];
        }
        c[2 * j] = ar; c[2 * j + 1] = ai;
    }

    /* add_wgn, main_BER_calculation.m:277-294 (awgn, wofdm_simulation.py:
     * 119-140): both powers measured over the same NL samples. */
    for (j = 0; j < NL; j++) {
        Ps += c[2 * j] * c[2 * j] + c[2 * j + 1] * c[2 * j + 1];
        Pn += unit_noise[2 * j] * unit_noise[2 * j] + unit_noise[2 * j + 1] * unit_noise[2 * j + 1];
    }
    Ps /= (double)NL; Pn /= (double)NL;
    g = sqrt(Ps * pow(10.0, -0.1 * snr_db) / Pn);
    /* truncate to S*B samples, main_BER_calculation.m:261 */
    for (j = 0; j < S * B; j++) {
        r[2 * j]     = c[2 * j]     + g * unit_noise[2 * j];
        r[2 * j + 1] = c[2 * j + 1] + g * unit_noise[2 * j + 1];
    }

    /* wofdm_rx, main_BER_calculation.m:297-310: remove gamma samples (442-454),
     * Rx window, overlap-and-add fold (336-355), circular shift (313-333), DFT.
     * K*P*V*R collapses to z[t] = sum_{m = t+kappa+delta/2 (mod N)} w[m] y[gamma+m]. */
    for (s = 0; s < S && rx_mat; s++) {
        /* rx_mat @ received block (wofdm_simulation.py:219-222) */
        const double *y = r + 2 * (size_t)s * B;
        for (n = 0; n < N; n++) {
            const double *row = rx_mat + 2 * (size_t)n * B;
            double ar = 0.0, ai = 0.0;
            for (j = 0; j < B; j++) {
                ar += row[2 * j] * y[2 * j] - row[2 * j + 1] * y[2 * j + 1];
                ai += row[2 * j] * y[2 * j + 1] + row[2 * j + 1] * y[2 * j];
            }
            Y[2 * ((size_t)s * N + n)] = ar;
            Y[2 * ((size_t)s * N + n) + 1] = ai;
        }
    }
    for (s = 0; s < S && !rx_mat; s++) {
        const double *y = r + 2 * (size_t)s * B;
        for (t = 0; t < N; t++) {
            int m0 = (t + kap + delta / 2) % N;
            double zr = w_rx[m0] * y[2 * (gam + m0)], zi = w_rx[m0] * y[2 * (gam + m0) + 1];
            if (m0 + N < N + delta) {
                zr += w_rx[m0 + N] * y[2 * (gam + m0 + N)];
                zi += w_rx[m0 + N] * y[2 * (gam + m0 + N) + 1];
            }
            z[2 * t] = zr; z[2 * t + 1] = zi;
        }
        wofdm_oracle_fft(N, -1, z);
        memcpy(Y + 2 * (size_t)s * N, z, sizeof(double) * 2 * N);
    }

    /* pilot LS estimate + one-tap equaliser, main_BER_calculation.m:266-268
     * (wofdm_simulation.py:223-232); hard decision 269-270 (233-234);
     * biterr 272 (np.mean(!=), 235). */
    for (s = 1; s < S; s++)
        for (n = 0; n < N; n++) {
            double y0r, y0i;
            if (sys->active && !sys->active[n]) continue;   /* main_channel_mask.m:367-369 */
            y0r = Y[2 * n]; y0i = Y[2 * n + 1];
            double x0r = X[2 * n], x0i = X[2 * n + 1];
            double xd = x0r * x0r + x0i * x0i;
            double hr = (y0r * x0r + y0i * x0i) / xd, hi = (y0i * x0r - y0r * x0i) / xd;
            double yr = Y[2 * ((size_t)s * N + n)], yi = Y[2 * ((size_t)s * N + n) + 1];
            double hd = hr * hr + hi * hi;
            double er = (yr * hr + yi * hi) / hd, ei = (yi * hr - yr * hi) / hd;
            unsigned lab_tx = labels[(size_t)s * N + n];
            unsigned lab_rx = nearest ? nearest_label(M, qam_table, er, ei)
                                      : slice_label(k, er, ei);
            bit_err += (uint64_t)__builtin_popcount(lab_tx ^ lab_rx);
            sym_err += (lab_tx != lab_rx);
            if (dump && dump->Xhat) {
                dump->Xhat[2 * ((size_t)(s - 1) * N + n)] = er;
                dump->Xhat[2 * ((size_t)(s - 1) * N + n) + 1] = ei;
            }
            if (dump && dump->labels_rx) dump->labels_rx[(size_t)(s - 1) * N + n] = (uint8_t)lab_rx;
        }

    {
        int nact = N;
        if (sys->active) for (nact = 0, n = 0; n < N; n++) nact += sys->active[n] != 0;
        counts[0] += bit_err;
        counts[1] += (uint64_t)(S - 1) * nact * k;
        counts[2] += sym_err;
        counts[3] += (uint64_t)(S - 1) * nact;
    }

    if (dump) {
        if (dump->X)    memcpy(dump->X, X, sizeof(double) * 2 * (size_t)S * N);
        if (dump->tx)   memcpy(dump->tx, tx, sizeof(double) * 2 * (size_t)T);
        if (dump->conv) memcpy(dump->conv, c, sizeof(double) * 2 * (size_t)CL);
        if (dump->rx)   memcpy(dump->rx, r, sizeof(double) * 2 * (size_t)S * B);
        if (dump->Y)    memcpy(dump->Y, Y, sizeof(double) * 2 * (size_t)S * N);
        if (dump->gain) dump->gain[0] = g;
    }
    rc = 0;
done:
    return rc;
}

int wofdm_oracle_frame(const wofdm_oracle_sys *sys,
                       const double *w_tx, const double *w_rx,
                       const double *h, double snr_db,
                       const double *qam_table, int nearest,
                       const uint8_t *labels, const double *unit_noise,
                       uint64_t counts[4], wofdm_oracle_dump *dump)
{
    return frame_impl(sys, w_tx, w_rx, h, snr_db, qam_table, nearest, labels, unit_noise, counts, dump,
                      NULL, NULL);
}

/* The hoisted dense operators of the Python reference (wofdm_simulation.py:464-471):
 *   tx_mat = W_tx R IDFT          [P][N]: tx_mat[i][n] = w_tx[i] e^{+2 pi i src(i) n / N} / N
 *   rx_mat = DFT K P V_rx R       [N][B]: rx_mat[q][gamma + m] = w_rx[m] e^{-2 pi i q t(m) / N},
 *                                          t(m) = m - kappa - delta/2 (mod N), m < N + delta */
static void build_dense(const wofdm_oracle_sys *sys, const double *w_tx, const double *w_rx,
                        double *tx_mat, double *rx_mat)
{
    const int N = sys->n_fft, P = sys_P(sys), B = sys_B(sys);
    int i, n, m, q;
    for (i = 0; i < P; i++) {
        int src = ((i - sys->cp) % N + N) % N;
        for (n = 0; n < N; n++) {
            double a = 2.0 * M_PI * (double)(((long)src * n) % N) / N;
            tx_mat[2 * ((size_t)i * N + n)] = w_tx[i] * cos(a) / N;
            tx_mat[2 * ((size_t)i * N + n) + 1] = w_tx[i] * sin(a) / N;
        }
    }
    memset(rx_mat, 0, sizeof(double) * 2 * (size_t)N * B);
    for (m = 0; m < N + sys->tail_rx; m++) {
        int t = (((m - sys->circ_shift - sys->tail_rx / 2) % N) + N) % N;
        for (q = 0; q < N; q++) {
            double a = -2.0 * M_PI * (double)(((long)q * t) % N) / N;
            rx_mat[2 * ((size_t)q * B + sys->prefix_rm + m)] += w_rx[m] * cos(a);
            rx_mat[2 * ((size_t)q * B + sys->prefix_rm + m) + 1] += w_rx[m] * sin(a);
        }
    }
}

/* ----------------------------------------------------------------- Philox */

void wofdm_oracle_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    int r;
    for (r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Stream definition shared with the HIP kernels (DESIGN.md "RNG"):
 *   key     = (seed lo, seed hi)
 *   counter = (block, frame lo, frame hi, stream<<28 | cell)
 *   stream 0 = data bits, stream 1 = unit noise. */
static void stream_block(uint64_t seed, uint32_t stream, uint32_t cell, uint64_t frame,
                         uint32_t block, uint32_t out[4])
{
    uint32_t ctr[4], key[2];
    ctr[0] = block; ctr[1] = (uint32_t)frame; ctr[2] = (uint32_t)(frame >> 32);
    ctr[3] = (stream << 28) | (cell & 0x0FFFFFFFu);
    key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32);
    wofdm_oracle_philox(ctr, key, out);
}

static int kslot_of(int k) { return k == 6 ? 8 : k; }

/* randi([0 1], N*k, S) + qammod bit grouping, main_BER_calculation.m:246-249:
 * subcarrier n of symbol s owns a kslot-bit field (kslot = 2,4,8 for k = 2,4,6)
 * at bit offset n*kslot of the symbol's bit stream; label = low k bits. */
void wofdm_oracle_gen_labels(const wofdm_oracle_sys *sys, uint64_t seed, uint32_t cell,
                             uint64_t frame, uint8_t *labels)
{
    const int N = sys->n_fft, S = sys->syms_per_frame, k = sys->bits_per_sc;
    const int ks = kslot_of(k), bps = N * ks / 128;
    int s, n;
    uint32_t w[4], cur = 0xFFFFFFFFu;
    for (s = 0; s < S; s++)
        for (n = 0; n < N; n++) {
            uint32_t bit = (uint32_t)n * (uint32_t)ks;
            uint32_t blk = (uint32_t)s * (uint32_t)bps + (bit >> 7);
            if (blk != cur) { stream_block(seed, 0u, cell, frame, blk, w); cur = blk; }
            labels[(size_t)s * N + n] = (uint8_t)((w[(bit >> 5) & 3] >> (bit & 31)) & ((1u << k) - 1u));
        }
}

/* randn + 1j*randn, main_BER_calculation.m:290: Box-Muller on the stream-1
 * blocks; block p gives samples 2p (words 0,1) and 2p+1 (words 2,3).
 * The uniforms are formed in float exactly as the kernels do; the
 * transcendental part is evaluated in double here. */
void wofdm_oracle_gen_noise(const wofdm_oracle_sys *sys, uint64_t seed, uint32_t cell,
                            uint64_t frame, double *unit_noise)
{
    const int NL = wofdm_oracle_noise_len(sys);
    int j;
    uint32_t w[4];
    for (j = 0; j < NL; j++) {
        float u1, u2;
        double rad, ang;
        if ((j & 1) == 0) stream_block(seed, 1u, cell, frame, (uint32_t)(j >> 1), w);
        u1 = fmaf((float)w[2 * (j & 1)], 2.3283064365386963e-10f, 1.1641532182693481e-10f);
        u2 = (float)(w[2 * (j & 1) + 1] >> 9) * 1.1920928955078125e-07f;   /* top 23 bits (philox.h) */
        rad = sqrt(-2.0 * log((double)u1));
        ang = 2.0 * M_PI * (double)u2;
        unit_noise[2 * j]     = rad * cos(ang);
        unit_noise[2 * j + 1] = rad * sin(ang);
    }
}

/* -------------------------------------------------------------- sweep loop */

int wofdm_oracle_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Driver loop of main_BER_calculation.m:64-201 restricted to the counting
 * part: for every (window pair, snr, channel) cell simulate the frames and
 * accumulate integer counters (quirk Q1: the reference keeps only the last
 * frame's BER; we accumulate all of them, like main_channel_mask.m:341-359). */
static int run_impl(const wofdm_oracle_sys *sys, int n_pairs, int n_snr, int n_channels,
                    const double *w_tx, const double *w_rx, const double *h,
                    const double *snr_db, uint64_t seed, uint64_t frame_offset,
                    uint64_t frames_per_cell, int n_threads, uint64_t *counts, int dense)
{
    int rc = sys_check(sys);
    double *tx_mats = NULL, *rx_mats = NULL;
    const int N = sys->n_fft, S = sys->syms_per_frame, L = sys->n_taps;
    const int P = sys_P(sys), NW = N + sys->tail_rx, NL = wofdm_oracle_noise_len(sys);
    const long n_cells = (long)n_pairs * n_snr * n_channels;
    const long total = n_cells * (long)frames_per_cell;
    int err = 0;
    if (rc) return rc;
    if (n_cells > 0x0FFFFFFFL) return -9;
    if (dense && !sys->tx_mask) {
        const size_t tsz = 2 * (size_t)P * N, rsz = 2 * (size_t)N * sys_B(sys);
        int pr;
        tx_mats = (double *)malloc(sizeof(double) * tsz * n_pairs);
        rx_mats = (double *)malloc(sizeof(double) * rsz * n_pairs);
        if (!tx_mats || !rx_mats) { free(tx_mats); free(rx_mats); return -100; }
        for (pr = 0; pr < n_pairs; pr++)
            build_dense(sys, w_tx + (size_t)pr * P, w_rx + (size_t)pr * NW, tx_mats + tsz * pr, rx_mats + rsz * pr);
    }
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
#pragma omp parallel num_threads(n_threads)
    {
        uint8_t *labels = (uint8_t *)malloc((size_t)S * N);
        double *noise = (double *)malloc(sizeof(double) * 2 * (size_t)NL);
        uint64_t *local = (uint64_t *)calloc((size_t)n_cells * 4, sizeof(uint64_t));
        long w;
#pragma omp for schedule(static)
        for (w = 0; w < total; w++) {
            long cell = w / (long)frames_per_cell;
            uint64_t frame = frame_offset + (uint64_t)(w % (long)frames_per_cell);
            int ch = (int)(cell % n_channels), sn = (int)((cell / n_channels) % n_snr);
            int pr = (int)(cell / ((long)n_channels * n_snr));
            int e;
            if (!labels || !noise || !local) { err = -100; continue; }
            wofdm_oracle_gen_labels(sys, seed, (uint32_t)cell, frame, labels);
            wofdm_oracle_gen_noise(sys, seed, (uint32_t)cell, frame, noise);
            e = frame_impl(sys, w_tx + (size_t)pr * P, w_rx + (size_t)pr * NW,
                           h + (size_t)ch * L * 2, snr_db[sn], NULL, 0,
                           labels, noise, local + 4 * cell, NULL,
                           tx_mats ? tx_mats + 2 * (size_t)P * N * pr : NULL,
                           rx_mats ? rx_mats + 2 * (size_t)N * sys_B(sys) * pr : NULL);
            if (e) err = e;
        }
#pragma omp critical
        {
            long q;
            if (local) for (q = 0; q < n_cells * 4; q++) counts[q] += local[q];
        }
        free(labels); free(noise); free(local);
    }
    free(tx_mats); free(rx_mats);
    return err;
}

int wofdm_oracle_run(const wofdm_oracle_sys *sys, int n_pairs, int n_snr, int n_channels,
                     const double *w_tx, const double *w_rx, const double *h,
                     const double *snr_db, uint64_t seed, uint64_t frame_offset,
                     uint64_t frames_per_cell, int n_threads, uint64_t *counts)
{
    return run_impl(sys, n_pairs, n_snr, n_channels, w_tx, w_rx, h, snr_db, seed, frame_offset,
                    frames_per_cell, n_threads, counts, 0);
}

int wofdm_oracle_run_dense(const wofdm_oracle_sys *sys, int n_pairs, int n_snr, int n_channels,
                           const double *w_tx, const double *w_rx, const double *h,
                           const double *snr_db, uint64_t seed, uint64_t frame_offset,
                           uint64_t frames_per_cell, int n_threads, uint64_t *counts)
{
    return run_impl(sys, n_pairs, n_snr, n_channels, w_tx, w_rx, h, snr_db, seed, frame_offset,
                    frames_per_cell, n_threads, counts, 1);
}
