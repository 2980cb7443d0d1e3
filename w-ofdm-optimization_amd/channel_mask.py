"""``matlab/main_channel_mask.m`` as functions -- SURVEY.md 8f row f1.

The BER experiment of main_BER_calculation.m run twice on the same data: as is, and with every
windowed symbol passed through a DFT-domain raised-cosine "channel mask" before the overlap-add;
in both only the centre half of the spectrum carries data.  The frame loop runs on the GPU:
``Plan.set_allocation`` (zero padding + ifftshift of lines 387-390, bin selection of 367-369) and
``Plan.set_tx_mask`` (``dft_rc_filt``, 398-417).

BER is accumulated over the whole ensemble here as there (lines 341-359).  The reference sends
the same data bits through both runs with independent noise (two ``add_wgn`` calls); here the
plain and the masked run simply use disjoint frame ranges (independent bits and noise), which
leaves every BER estimate unbiased.
"""
import os

import numpy as np

from . import simulation as S
from . import variants as V

ROLL_OFF = 10             # main_channel_mask.m:54


def half_band_allocation(n_fft):
    """Loaded bins of lines 387-390: ``[zeros(offset) data zeros(offset)]`` with offset = N/4
    (line 53), centred spectrum -> after ifftshift the data sit on bins [0, N/4) and [3N/4, N)."""
    a = np.zeros(n_fft, dtype=bool)
    a[:n_fft // 4] = True
    a[3 * n_fft // 4:] = True
    return a


def gen_raised_cosine(window_length, roll_off, total_length):
    """Centred raised-cosine mask (lines 443-458): zeros | rising sin^2 edge of ``roll_off``
    samples | ``window_length`` ones | falling edge | zeros."""
    axis = np.arange(-(roll_off + 1) / 2 + 1, (roll_off + 1) / 2 - 1 + 0.5, 1.0)
    rc = np.sin(np.pi / 2 * (0.5 + axis / roll_off)) ** 2
    rest = total_length - window_length - 2 * roll_off
    return np.concatenate([np.zeros(rest // 2), rc, np.ones(window_length), rc[::-1],
                           np.zeros(rest - rest // 2)])


def tx_mask(sym_len, roll_off=ROLL_OFF):
    """DFT-domain gains of ``dft_rc_filt`` in natural bin order (lines 402-405):
    ``ifftshift(gen_raised_cosine(floor((2P-1)/2), rollOff, 2P-1))``."""
    L = 2 * sym_len - 1
    return np.fft.ifftshift(gen_raised_cosine(L // 2, roll_off, L))


def dft_rc_filt(rows, roll_off=ROLL_OFF):
    """Host restatement of lines 398-417 on [S, P] rows (numpy FFT of length 2P-1): used by the
    tests as an independent check of the mask stage."""
    rows = np.asarray(rows, dtype=np.complex128)
    n_sym, P = rows.shape
    L = 2 * P - 1
    y = np.fft.ifft(np.fft.fft(rows, L, axis=1) * tx_mask(P, roll_off)[None, :], axis=1)
    out = y[:, :P].copy()
    out[1:, :P - 1] += y[:-1, P:]
    return out


def run_sim_mc(system, n_fft, cp, w_tx, w_rx, channels, snr_db, ensemble, bits_per_subcar=4,
               symbols_per_tx=16, tail_tx=None, tail_rx=None, roll_off=ROLL_OFF, seed=0, device=0,
               frame_range=None):
    """``run_sim_mc`` (lines 334-360) for every (window pair, SNR, channel) at once.

    w_tx [pairs][P], w_rx [pairs][N+delta].  Returns (counts_masked, counts_plain), uint64
    [pairs][n_snr][n_channels][4]; BER = counts[..., 0] / counts[..., 1]."""
    st = V.make_structure(system, n_fft, cp, tail_tx, tail_rx)
    w_tx, w_rx = np.atleast_2d(w_tx), np.atleast_2d(w_rx)
    h = np.atleast_2d(np.asarray(channels))
    snr = np.atleast_1d(np.asarray(snr_db, dtype=np.float64))
    cfg = S.make_cfg(st, bits_per_subcar, symbols_per_tx, h.shape[1], h.shape[0], snr.size,
                     w_tx.shape[0], noise_before_truncate=True, seed=seed)
    lo, n = (0, ensemble) if frame_range is None else frame_range
    with S.Plan(cfg, w_tx, w_rx, h.astype(np.complex64), snr.astype(np.float32), device=device) as plan:
        plan.set_allocation(half_band_allocation(n_fft))
        plain = plan.run(lo, n)
        plan.set_tx_mask(tx_mask(st.sym_len, roll_off))
        # fresh frames: the reference draws new noise for the masked run (line 352)
        masked = plan.run(ensemble + lo, n)
    return masked, plain


def ber_for_window_file(type_ofdm, cp, windows, channels, snr_db, num_subcar=256, bits_per_subcar=4,
                        symbols_per_tx=16, ensemble=100, tail_tx=8, tail_rx=10, seed=0, device=0,
                        frame_range=None):
    """Loop nest of lines 56-263 for one window file: every window pair of the file + the RC
    pair, plain and masked.  Returns ({variable name: BER vs SNR}, (counts_masked, counts_plain))
    with the saved variable names of lines 287-330 (``berSNR``, ``berMaskedSNR``, ``berRCSNR``,
    ``berMaskedRCSNR``, ``ber[Masked]SNRStep{1,2,3}{A,B}``)."""
    st = V.make_structure(type_ofdm, num_subcar, cp, tail_tx if type_ofdm in V.TX_WINDOWED else 0,
                          tail_rx if type_ofdm in V.RX_WINDOWED else 0)
    rc = {"tx": V.tx_rc_window(st), "rx": V.rx_rc_window(st)}
    plan = V.matlab_pair_plan(type_ofdm)

    def pick(key, side):
        return rc[side] if key == "rc" else np.asarray(windows[key], dtype=np.float64)

    names = [n for n, _ in plan]
    w_tx = np.stack([pick(k[0], "tx") for _, k in plan])
    w_rx = np.stack([pick(k[1], "rx") for _, k in plan])
    masked, plain = run_sim_mc(type_ofdm, num_subcar, cp, w_tx, w_rx, channels, snr_db, ensemble,
                               bits_per_subcar, symbols_per_tx, st.tail_tx, st.tail_rx, seed=seed,
                               device=device, frame_range=frame_range)
    return results_from_counts(names, masked, plain), (masked, plain)


def results_from_counts(names, masked, plain):
    def ber(c):      # mean over channels of per-channel BER (lines 84-117)
        return (c[..., 0] / np.maximum(c[..., 1], 1)).mean(axis=-1)
    out = {}
    bm, bp = ber(masked), ber(plain)
    for i, name in enumerate(names):
        if name == "opt":
            out["berSNR"], out["berMaskedSNR"] = bp[i], bm[i]
        elif name == "rc":
            out["berRCSNR"], out["berMaskedRCSNR"] = bp[i], bm[i]
        else:
            out["berSNRStep" + name], out["berMaskedSNRStep" + name] = bp[i], bm[i]
    return out


def save_results(results_path, type_ofdm, cp, results):
    """Files of lines 118-125, 261-263 under ``ber_results/simulation_with_channel_mask``."""
    from scipy.io import savemat
    os.makedirs(results_path, exist_ok=True)
    base = "ber_%s_%dCP" % (type_ofdm, cp)
    groups = {"optimized_": [k for k in results if k == "berSNR" or k.startswith("berSNRStep")],
              "masked_optimized_": [k for k in results if k == "berMaskedSNR"
                                    or k.startswith("berMaskedSNRStep")],
              "rc_": ["berRCSNR"], "masked_rc_": ["berMaskedRCSNR"]}
    paths = []
    for prefix, keys in groups.items():
        keys = [k for k in keys if k in results]
        if keys:
            paths.append(os.path.join(results_path, prefix + base + ".mat"))
            savemat(paths[-1], {k: np.asarray(results[k]).reshape(-1, 1) for k in keys})
    return paths
