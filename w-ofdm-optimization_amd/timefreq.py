"""Tx-side spectrum analysis (host, numpy) -- SURVEY.md 8f row f4.

Mirrors ``python/ofdm_utils/timefreq_simulation.py``: the averaged-periodogram PSD estimate of
the transmitted w-OFDM waveform (lines 101-123), the out-of-band-radiation figure and main-band
samples (216-296), the closed-form PSD (155-214) and the per-(system, CP) work item with its
``timefreq/{opt,rc}_<sys>_<cp>.npz`` / ``CP_<cp>.npz`` outputs (17-76).

The Tx chain is the same index arithmetic as the BER kernels' phase A (IDFT, CP/CS copy, window,
overlap-add of the tails) written with FFTs; no dense matrices.  Reference behaviours kept:
16-QAM symbols are *not* normalised (line 239), the last (partial, zero-padded) periodogram slice
counts as a full one in the average (118-121), the cosine term of the closed form takes
``f/delta_f`` without the 2 pi (207-213).
"""
import os

import numpy as np

from . import variants as V

#: timefreq_simulation.py:235-238
SYMBOLS_16QAM = np.array([a + 1j * b for a in (-3, -1, 1, 3) for b in (-3, -1, 1, 3)])
NO_SYMBOLS = 256          # line 219
GUARD_BAND = 48           # line 220


def allocation_index(n_fft, guard_band=GUARD_BAND):
    """Bins that carry data (subcar_alloc_mat, lines 223-233): 1..N/2-gb and N/2+gb..N-1."""
    half = n_fft // 2 - guard_band
    return np.r_[1:1 + half, n_fft // 2 + guard_band:n_fft]


def draw_symbols(n_fft, rng=None, no_symbols=NO_SYMBOLS, guard_band=GUARD_BAND):
    """[N-2gb, no_symbols] 16-QAM draw of line 240; ``rng`` a ``RandomState`` (the reference
    uses numpy's global legacy generator) or None for a fresh one."""
    rng = np.random.RandomState() if rng is None else rng
    return rng.choice(SYMBOLS_16QAM, size=(n_fft - 2 * guard_band, no_symbols), replace=True)


def tx_symbols(st, X, w_tx, guard_band=GUARD_BAND):
    """[no_symbols, P]: window x (CP/CS copy of the IDFT of the allocated bins), lines 242-246."""
    n = st.n_fft
    grid = np.zeros((n, X.shape[1]), dtype=np.complex128)
    grid[allocation_index(n, guard_band)] = X
    t = np.fft.ifft(grid, axis=0)                     # idft_mat = conj(DFT)/N
    idx = (np.arange(st.sym_len) - st.cp) % n
    return (np.asarray(w_tx)[:, None] * t[idx]).T


def overlap_and_add(x, beta):
    """[S, P] -> serialised frame (lines 84-99)."""
    if beta == 0:
        return x.reshape(-1)
    body = x[:, beta:].copy()
    body[:-1, -beta:] += x[1:, :beta]
    return np.concatenate([x[0, :beta], body.reshape(-1)])


def psd_estimate(x, fft_len):
    """Mean of |fftshift(FFT)|^2 over consecutive length-``fft_len`` slices, the zero-padded
    remainder included as one more slice (lines 101-123)."""
    n_full = len(x) // fft_len
    acc = (np.abs(np.fft.fft(x[:n_full * fft_len].reshape(n_full, fft_len), axis=1)) ** 2).sum(axis=0)
    rest = x[n_full * fft_len:]
    if rest.size:
        acc = acc + np.abs(np.fft.fft(rest, fft_len)) ** 2
    return np.fft.fftshift(acc) / (n_full + 1)


def psd_estimate_gpu(st, X, w_tx, overlap, guard_band=GUARD_BAND, device=0):
    """``psd_estimate(overlap_and_add(tx_symbols(st, X, w_tx), overlap), 8 N)`` on the GPU
    (``wofdm_tx_psd``: the frame kernel's Tx half on the given symbols + the averaged periodogram).
    X: [N - 2 gb, no_symbols] like ``draw_symbols``.  No CPU fallback; N in {64, 128, 256}."""
    import ctypes as C
    from . import _lib
    from .simulation import make_cfg
    n = st.n_fft
    X = np.asarray(X)
    grid = np.zeros((X.shape[1], n), dtype=np.complex64)
    grid[:, allocation_index(n, guard_band)] = X.T
    w = _lib.f32(np.asarray(w_tx).reshape(-1), (st.sym_len,))
    cfg = make_cfg(st, 4, 16, 1, 1, 1, 1)
    out = np.zeros(8 * n, dtype=np.float32)
    gf = _lib.c64_as_f32(grid)
    _lib.check(_lib.load().wofdm_tx_psd(C.byref(cfg), int(device), w.ctypes.data, gf.ctypes.data,
                                        int(X.shape[1]), int(overlap), out.ctypes.data))
    length = overlap + X.shape[1] * (st.sym_len - overlap)
    return out.astype(np.float64) / (length // (8 * n) + 1)


def analytical_psd(st, w_tx, sampling_period, guard_band=GUARD_BAND, fft_len=None):
    """(S_opt, S_rc, S_cp) of lines 155-214 for the Tx window ``w_tx`` (vector, length P)."""
    from scipy.signal import firwin
    n, cp, cs = st.n_fft, st.cp, st.cs
    fft_len = 8 * n if fft_len is None else fft_len
    up = fft_len / n
    f_axis = np.linspace(-.5, .5 - 1 / fft_len, fft_len) / sampling_period
    delta_f = 1 / (n * sampling_period)
    sigma2 = (n / (n - guard_band)) ** 2
    g_i = firwin(st.sym_len, [f_axis[int(fft_len / 2 + up)], f_axis[-int(guard_band * up)]],
                 window="boxcar", fs=1 / sampling_period, pass_zero=False)

    def corr(win):
        return ((win ** 2).sum(), (win[:cp] * win[n:n + cp]).sum(),
                (win[n + cp:n + cp + cs] * win[cp:cp + cs]).sum())

    def spectrum(win, denom, c):
        G = np.abs(np.fft.fftshift(np.fft.fft(g_i * win, fft_len))) ** 2
        return G * (n * sigma2 / denom) / up * (c[0] + 2 * (c[1] + c[2]) * np.cos(f_axis / delta_f))

    w_tx = np.asarray(w_tx, dtype=np.float64)
    w_rc = V.tx_rc_window(st)
    return (spectrum(w_tx, st.sym_len, corr(w_tx)), spectrum(w_rc, st.sym_len, corr(w_rc)),
            spectrum(np.ones(st.sym_len), n + cp, (n + cp, cp, 0.0)))


def estimate_obr(st, w_tx, samp_period=200e-9, X=None, rng=None, gpu=False):
    """``wOFDMSystem.estimate_obr`` (lines 216-296): three dicts (optimised window, RC window,
    plain CP-OFDM) with the reference's keys.  gpu=True: waveform and periodogram by ``wofdm_tx_psd``."""
    n = st.n_fft
    fft_len = 8 * n
    X = draw_symbols(n, rng) if X is None else np.asarray(X)
    w_tx = np.asarray(w_tx, dtype=np.float64)
    overlap = st.tail_tx if st.system in ("wtx", "CPwtx", "WOLA", "CPW") else 0
    x_opt = overlap_and_add(tx_symbols(st, X, w_tx), overlap)
    x_rc = overlap_and_add(tx_symbols(st, X, V.tx_rc_window(st)), overlap)
    x_cp = tx_symbols(st, X, np.ones(st.sym_len)).reshape(-1)
    f_axis = np.linspace(-.5, .5 - 1 / fft_len, fft_len) / 200e-9
    interp = fft_len / n
    gb = int(interp * GUARD_BAND)
    S = analytical_psd(st, w_tx, samp_period, GUARD_BAND, fft_len)
    out = []
    wins = {"opt": (w_tx, overlap), "rc": (V.tx_rc_window(st), overlap), "cp": (np.ones(st.sym_len), 0)}
    for tag, x, s in (("opt", x_opt, S[0]), ("rc", x_rc, S[1]), ("cp", x_cp, S[2])):
        est = psd_estimate_gpu(st, X, *wins[tag]) if gpu else psd_estimate(x, fft_len)
        out.append({"X_est_" + tag: est, "S_" + tag: s, "f_axis": f_axis,
                    "obr_" + tag: np.mean(np.hstack((est[:gb], est[-gb:]))),
                    "mf_band_" + tag: np.hstack((est[gb:fft_len // 2],
                                                 est[-int(fft_len / 2 - interp):-gb]))})
    return tuple(out)


def timefreq_fun(data, rng=None):
    """Work item of ``-m run_timefreq`` (lines 17-76): data = (system, dft_len, cp_len, tail_tx,
    tail_rx, window_path, folder_path)."""
    system, n_fft, cp, tail_tx, tail_rx, window_path, folder_path = data
    st = V.make_structure(system, n_fft, cp, tail_tx if system in V.TX_WINDOWED else 0,
                          tail_rx if system in V.RX_WINDOWED else 0)
    if system in V.TX_WINDOWED:
        vec = np.load(os.path.join(window_path, "%s_%d.npy" % (system, cp)))
        xt, _ = V.split_tail_file(st, vec)
        w_tx = V.expand_tx_window(st, xt)
    else:
        w_tx = np.ones(st.sym_len)
    opt, rc, cpd = estimate_obr(st, w_tx, 200e-9, rng=rng)
    path = os.path.join(folder_path, "timefreq")
    os.makedirs(path, exist_ok=True)
    np.savez(os.path.join(path, "opt_%s_%d.npz" % (system, cp)), **opt)
    np.savez(os.path.join(path, "rc_%s_%d.npz" % (system, cp)), **rc)
    np.savez(os.path.join(path, "CP_%d.npz" % cp), **cpd)
    return opt, rc, cpd
