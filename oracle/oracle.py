"""ctypes front end of the CPU ORACLE (oracle/wofdm_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package never imports this module
(tests/test_layout.py enforces it).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OracleSys(C.Structure):
    """Mirror of wofdm_oracle_sys (one w-OFDM structure)."""
    _fields_ = [(n, C.c_int32) for n in (
        "n_fft", "bits_per_sc", "syms_per_frame", "cp", "cs", "tail_tx", "tail_rx",
        "prefix_rm", "circ_shift", "n_taps", "noise_before_truncate")] + [
        ("active", C.c_void_p), ("tx_mask", C.c_void_p)]

    @property
    def P(self):
        return self.n_fft + self.cp + self.cs

    @property
    def B(self):
        return self.P - self.tail_tx

    @property
    def T(self):
        return self.tail_tx + self.syms_per_frame * self.B


class _Dump(C.Structure):
    _fields_ = [("X", C.c_void_p), ("tx", C.c_void_p), ("conv", C.c_void_p),
                ("rx", C.c_void_p), ("Y", C.c_void_p), ("Xhat", C.c_void_p),
                ("labels_rx", C.c_void_p), ("gain", C.c_void_p)]


def build(force=False, asan=False):
    """Compile the oracle with gcc (make)."""
    name = "libwofdm_oracle_asan.so" if asan else "libwofdm_oracle.so"
    path = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "wofdm_oracle.c")
    if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "asan" if asan else "all"], check=True,
                       stdout=subprocess.DEVNULL)
    return path


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libwofdm_oracle.so")
        if not os.path.exists(path):
            path = build()
        L = C.CDLL(path)
        dp, u8p, u64p = C.c_void_p, C.c_void_p, C.c_void_p
        L.wofdm_oracle_noise_len.argtypes = [C.POINTER(OracleSys)]
        L.wofdm_oracle_qam_table.argtypes = [C.c_int, dp]
        L.wofdm_oracle_frame.argtypes = [C.POINTER(OracleSys), dp, dp, dp, C.c_double, dp,
                                         C.c_int, u8p, dp, u64p, C.POINTER(_Dump)]
        L.wofdm_oracle_philox.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.wofdm_oracle_philox.restype = None
        L.wofdm_oracle_gen_labels.argtypes = [C.POINTER(OracleSys), C.c_uint64, C.c_uint32,
                                              C.c_uint64, u8p]
        L.wofdm_oracle_gen_labels.restype = None
        L.wofdm_oracle_gen_noise.argtypes = [C.POINTER(OracleSys), C.c_uint64, C.c_uint32,
                                             C.c_uint64, dp]
        L.wofdm_oracle_gen_noise.restype = None
        L.wofdm_oracle_run.argtypes = [C.POINTER(OracleSys), C.c_int, C.c_int, C.c_int, dp, dp,
                                       dp, dp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, u64p]
        L.wofdm_oracle_run_dense.argtypes = L.wofdm_oracle_run.argtypes
        L.wofdm_oracle_fft.argtypes = [C.c_int, C.c_int, dp]
        L.wofdm_oracle_fft.restype = None
        L.wofdm_oracle_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _c(a):
    """complex array -> contiguous interleaved float64 view"""
    a = np.ascontiguousarray(a, dtype=np.complex128)
    return a.view(np.float64)


def make_sys(n_fft, bits_per_sc, syms_per_frame, cp, cs, tail_tx, tail_rx, prefix_rm,
             circ_shift, n_taps, noise_before_truncate=1, active=None, tx_mask=None):
    """active: [N] bool subcarrier allocation; tx_mask: [2P-1] DFT-domain mask gains
    (main_channel_mask.m variant); the arrays are kept alive on the returned object."""
    sys = OracleSys(n_fft, bits_per_sc, syms_per_frame, cp, cs, tail_tx, tail_rx, prefix_rm,
                    circ_shift, n_taps, noise_before_truncate)
    if active is not None:
        sys._active = np.ascontiguousarray(np.asarray(active) != 0, dtype=np.uint8)
        assert sys._active.shape == (n_fft,)
        sys.active = sys._active.ctypes.data
    if tx_mask is not None:
        sys._tx_mask = np.ascontiguousarray(tx_mask, dtype=np.float64)
        assert sys._tx_mask.shape == (2 * sys.P - 1,)
        sys.tx_mask = sys._tx_mask.ctypes.data
    return sys


def noise_len(sys):
    return lib().wofdm_oracle_noise_len(C.byref(sys))


def qam_table(k):
    t = np.zeros(1 << k, dtype=np.complex128)
    rc = lib().wofdm_oracle_qam_table(k, t.ctypes.data)
    assert rc == 0
    return t


def philox(ctr, key):
    ctr = np.ascontiguousarray(ctr, dtype=np.uint32)
    key = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().wofdm_oracle_philox(ctr.ctypes.data, key.ctypes.data, out.ctypes.data)
    return out


def fft(x, direction):
    buf = np.array(x, dtype=np.complex128)
    lib().wofdm_oracle_fft(buf.size, direction, buf.ctypes.data)
    return buf


def gen_labels(sys, seed, cell, frame):
    out = np.zeros((sys.syms_per_frame, sys.n_fft), dtype=np.uint8)
    lib().wofdm_oracle_gen_labels(C.byref(sys), seed, cell, frame, out.ctypes.data)
    return out


def gen_noise(sys, seed, cell, frame):
    out = np.zeros(noise_len(sys), dtype=np.complex128)
    lib().wofdm_oracle_gen_noise(C.byref(sys), seed, cell, frame, out.ctypes.data)
    return out


def frame(sys, w_tx, w_rx, h, snr_db, labels, unit_noise, table=None, nearest=False,
          dump=False):
    """One frame with explicit randomness -> (counts[4], stages dict or None)."""
    S, N = sys.syms_per_frame, sys.n_fft
    w_tx = np.ascontiguousarray(w_tx, dtype=np.float64)
    w_rx = np.ascontiguousarray(w_rx, dtype=np.float64)
    assert w_tx.shape == (sys.P,) and w_rx.shape == (N + sys.tail_rx,)
    h = np.ascontiguousarray(h, dtype=np.complex128)
    assert h.shape == (sys.n_taps,)
    labels = np.ascontiguousarray(labels, dtype=np.uint8)
    assert labels.shape == (S, N)
    unit_noise = np.ascontiguousarray(unit_noise, dtype=np.complex128)
    assert unit_noise.shape == (noise_len(sys),)
    tab = None
    if table is not None:
        tab = np.ascontiguousarray(table, dtype=np.complex128)
        assert tab.shape == (1 << sys.bits_per_sc,)
    counts = np.zeros(4, dtype=np.uint64)
    d, st = None, None
    if dump:
        st = dict(X=np.zeros((S, N), np.complex128), tx=np.zeros(sys.T, np.complex128),
                  conv=np.zeros(sys.T + sys.n_taps - 1, np.complex128),
                  rx=np.zeros(S * sys.B, np.complex128), Y=np.zeros((S, N), np.complex128),
                  Xhat=np.zeros((S - 1, N), np.complex128),
                  labels_rx=np.zeros((S - 1, N), np.uint8), gain=np.zeros(1, np.float64))
        d = _Dump(*[st[f].ctypes.data for f, _ in _Dump._fields_])
    rc = lib().wofdm_oracle_frame(C.byref(sys), w_tx.ctypes.data, w_rx.ctypes.data,
                                  h.ctypes.data, float(snr_db),
                                  tab.ctypes.data if tab is not None else None,
                                  1 if nearest else 0, labels.ctypes.data,
                                  unit_noise.ctypes.data, counts.ctypes.data,
                                  C.byref(d) if d is not None else None)
    if rc != 0:
        raise ValueError("wofdm_oracle_frame failed: %d" % rc)
    return counts, st


def run(sys, w_tx, w_rx, h, snr_db, seed, frame_offset, frames_per_cell, n_threads=0, dense=False):
    """Generate-mode sweep -> counts[pairs][n_snr][n_channels][4] (uint64).  dense=True: Tx / Rx as
    the reference's hoisted dense matrix products instead of FFTs (``wofdm_oracle_run_dense``)."""
    w_tx = np.ascontiguousarray(np.atleast_2d(w_tx), dtype=np.float64)
    w_rx = np.ascontiguousarray(np.atleast_2d(w_rx), dtype=np.float64)
    h = np.ascontiguousarray(np.atleast_2d(h), dtype=np.complex128)
    snr_db = np.ascontiguousarray(np.atleast_1d(snr_db), dtype=np.float64)
    n_pairs, n_ch, n_snr = w_tx.shape[0], h.shape[0], snr_db.shape[0]
    assert w_tx.shape[1] == sys.P and w_rx.shape == (n_pairs, sys.n_fft + sys.tail_rx)
    assert h.shape[1] == sys.n_taps
    counts = np.zeros((n_pairs, n_snr, n_ch, 4), dtype=np.uint64)
    fn = lib().wofdm_oracle_run_dense if dense else lib().wofdm_oracle_run
    rc = fn(C.byref(sys), n_pairs, n_snr, n_ch, w_tx.ctypes.data,
                                w_rx.ctypes.data, h.ctypes.data, snr_db.ctypes.data,
                                int(seed), int(frame_offset), int(frames_per_cell),
                                int(n_threads), counts.ctypes.data)
    if rc != 0:
        raise ValueError("wofdm_oracle_run failed: %d" % rc)
    return counts


def threads():
    return lib().wofdm_oracle_threads()
