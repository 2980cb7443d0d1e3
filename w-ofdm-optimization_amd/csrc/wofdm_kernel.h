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
};

// LDS carve (float2 units first, then floats); must match the kernel.
struct wofdm_lds_layout {
    int fbuf_len;   // float2: WOFDM_LT-1 | T + L - 1 ... | pad
    int tail_len;   // float2: S * beta
    int tw_len;     // float2: N
    int g_len;      // float2: N
    int wtx_len;    // float : P
    int wrx_len;    // float : N + delta
    int sums_len;   // float : 2 * S
    size_t bytes;
};

struct wofdm_kparams {
    // structure: S k mu rho beta delta gamma kappa L, P = N+mu+rho, B = P-beta,
    // T = beta+S*B
    int S, k, mu, rho, beta, delta, gamma, kappa, L, P, B, T;
    int NL;                 // unit-noise samples per frame (T+L-1 or S*B)
    int n_snr, n_ch;        // cell = (pair*n_snr + snr)*n_ch + ch
    uint32_t n_cells;       // cells covered by this launch, starting at first_cell
    uint32_t first_cell;
    uint32_t inject_base_cell;   // injected arrays are indexed from this cell
    float qam_scale, qam_inv;    // 1/sqrt(2(M-1)/3) and its reciprocal
    wofdm_lds_layout lds;
    uint64_t frames_per_cell, frame_offset;
    uint32_t seed_lo, seed_hi;
    unsigned long long *counts;   // [cells][4]
    const uint8_t *labels;  // inject: [cells][frames][S][N]
    const float2  *unit_noise;    // inject: [cells][frames][NL]
    wofdm_kdump dump;
};

static inline int wofdm_kslot(int k) { return k == 6 ? 8 : k; }
static inline int wofdm_rb(int n_fft) { return n_fft / 64 + 1; }   // outputs per lane in the FIR

static inline wofdm_lds_layout wofdm_make_layout(int N, int S, int P, int B, int beta, int delta)
{
    wofdm_lds_layout l;
    const int T = beta + S * B;
    auto up = [](int v, int a) { return (v + a - 1) / a * a; };
    l.fbuf_len = up((WOFDM_LT - 1) + T + (WOFDM_LT - 1) + wofdm_rb(N) + 8, 2);
    l.tail_len = up(S * (beta > 0 ? beta : 1), 2);
    l.tw_len = N;
    l.g_len = N;
    l.wtx_len = up(P, 4);
    l.wrx_len = up(N + delta, 4);
    l.sums_len = up(2 * S, 4);
    l.bytes = (size_t)8 * (l.fbuf_len + l.tail_len + l.tw_len + l.g_len)
            + (size_t)4 * (l.wtx_len + l.wrx_len + l.sums_len);
    return l;
}

// kernel registry (wofdm_kernel.hip)
// constants travel as separate noalias arguments so that uniform reads become scalar loads:
// w_tx[pairs][P], w_rx[pairs][N+delta], h[n_ch][WOFDM_LT] zero padded, noise_lin[n_snr]
typedef void (*wofdm_kernel_fn)(wofdm_kparams, const float *, const float *, const float2 *,
                                const float *);
enum { WOFDM_MODE_GEN = 0, WOFDM_MODE_INJECT = 1, WOFDM_MODE_DUMP_GEN = 2, WOFDM_MODE_DUMP_INJECT = 3 };
wofdm_kernel_fn wofdm_select_kernel(int n_fft, int mode);
hipError_t wofdm_philox_kat_launch(const uint32_t *ctr_key_dev, uint32_t *out_dev, hipStream_t s);
