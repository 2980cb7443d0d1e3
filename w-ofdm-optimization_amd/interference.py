"""Closed-form ICI + ISI power of a w-OFDM structure (host, numpy) -- SURVEY.md 8f row f2.

Mirrors ``calculate_interference`` (matlab/main_interference_calculation.m:177-225) and
``interf_power`` (python/ofdm_utils/interf_calc.py:20-113):

    A_m = W K P V_rx R  H_m  V_tx Gamma W^-1,      H_m[b, c] = h[m (N+mu+rho-beta) + b - c]
    P[n] = sum_{n' != n} |A_0[n, n']|^2 + sum_{m >= 1} sum_{n'} |A_m[n, n']|^2

built from the same index formulas as the kernels (no dense DFT loops): the Tx map is an IDFT
matrix with rows picked by the CP/CS copy and scaled by the Tx window, the Rx map is the DFT of
the windowed fold of SURVEY.md 3.4-10.  It is the deterministic companion of every BER curve
and an analytic, RNG-free check of the frame pipeline.
"""
import numpy as np


def tx_matrix(st, w_tx):
    """[P, N]: V_tx Gamma W^-1 (transmitter.py:13-58, wofdm_simulation.py:464)."""
    n = st.n_fft
    i = np.arange(st.sym_len)
    t = (i - st.cp) % n
    return np.asarray(w_tx)[:, None] * np.exp(2j * np.pi * np.outer(t, np.arange(n)) / n) / n


def rx_matrix(st, w_rx):
    """[N, B]: W K P V_rx R (receiver.py:13-133, wofdm_simulation.py:468-469)."""
    n, delta = st.n_fft, st.tail_rx
    fold = np.zeros((n, st.stride))
    for t in range(n):
        m0 = (t + st.circ_shift + delta // 2) % n
        fold[t, st.prefix_rm + m0] += w_rx[m0]
        if m0 + n < n + delta:
            fold[t, st.prefix_rm + m0 + n] += w_rx[m0 + n]
    return np.fft.fft(fold, axis=0)


def channel_tensor(st, h):
    """[M, B, P] with H_m[b, c] = h[m*B + b - c] (channel.py:14-53, channel_array.m:20-36)."""
    h = np.asarray(h).reshape(-1)
    B, P = st.stride, st.sym_len
    M = 1 + int(np.ceil((h.size - 1 + st.tail_tx) / B))
    idx = (np.arange(M)[:, None, None] * B + np.arange(B)[None, :, None] - np.arange(P)[None, None, :])
    ok = (idx >= 0) & (idx < h.size)
    return np.where(ok, h[np.clip(idx, 0, h.size - 1)], 0)


def interference_matrices(st, w_tx, w_rx, h):
    """(A_0 [N, N], A_m [M-1, N, N])."""
    T, R, H = tx_matrix(st, w_tx), rx_matrix(st, w_rx), channel_tensor(st, h)
    A = R @ H @ T
    return A[0], A[1:]


def interf_power(st, w_tx, w_rx, h):
    """Per-subcarrier ICI + ISI power, ``np.diag(PISI + PICI1)`` of interf_calc.py:91-100."""
    a0, am = interference_matrices(st, w_tx, w_rx, h)
    off = a0 - np.diag(np.diag(a0))
    return (np.abs(off) ** 2).sum(axis=1) + (np.abs(am) ** 2).sum(axis=(0, 2))


def total_interference(st, w_tx, w_rx, h):
    """Scalar of the MATLAB function: tr(offdiag(A0)^H offdiag(A0)) + sum_m tr(A_m^H A_m)."""
    return float(interf_power(st, w_tx, w_rx, h).sum())


def interf_power_gpu(st, w_tx, w_rx, h, device=0):
    """The same per-subcarrier power on the GPU (``wofdm_interference``: the frame kernels' own Tx /
    FIR / Rx chain applied to the N unit symbols, one workgroup per (window pair, channel)), batched:
    w_tx [pairs][P], w_rx [pairs][N + tail_rx], h [n_channels][taps]  ->  float32 [pairs][n_channels][N].
    No CPU fallback."""
    import ctypes as C
    from . import _lib
    from .simulation import make_cfg
    w_tx = _lib.f32(np.atleast_2d(w_tx))
    w_rx = _lib.f32(np.atleast_2d(w_rx))
    hh = np.atleast_2d(np.asarray(h))
    if w_tx.shape != (w_rx.shape[0], st.sym_len) or w_rx.shape[1] != st.rx_win_len:
        raise ValueError("window shapes %s / %s do not fit the structure" % (w_tx.shape, w_rx.shape))
    cfg = make_cfg(st, 4, 16, hh.shape[1], hh.shape[0], 1, w_tx.shape[0])
    hf = _lib.c64_as_f32(hh, (cfg.n_channels, cfg.n_taps))
    out = np.zeros((w_tx.shape[0], hh.shape[0], st.n_fft), dtype=np.float32)
    lib = _lib.load()
    _lib.check(lib.wofdm_interference(C.byref(cfg), int(device), w_tx.ctypes.data, w_rx.ctypes.data,
                                      hf.ctypes.data, out.ctypes.data))
    return out
