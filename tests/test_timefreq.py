"""Tx-side PSD / OBR mirror against the reference's own outputs
(tests/golden/timefreq.npz, made by tests/golden/make_golden.py from
python/ofdm_utils/timefreq_simulation.py with its global generator seeded)."""
import os

import numpy as np
import pytest

import wofdm_amd as W  # noqa: F401
from wofdm_amd import timefreq as T
from wofdm_amd import variants as V

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "timefreq.npz"))
N_FFT, CP = (int(v) for v in GOLD["cfg"])


@pytest.mark.parametrize("system", ["wtx", "CPW", "wrx", "CPwtx"])
def test_estimate_obr_replays_the_reference(system):
    st = V.make_structure(system, N_FFT, CP)
    w_tx = V.expand_tx_window(st, GOLD[system + "_xt"])
    rng = np.random.RandomState(int(GOLD[system + "_seed"]))
    dicts = T.estimate_obr(st, w_tx, 200e-9, rng=rng)
    for tag, d in zip(("opt", "rc", "cp"), dicts):
        assert set(d) == {"X_est_" + tag, "S_" + tag, "f_axis", "obr_" + tag, "mf_band_" + tag}
        for k, v in d.items():
            ref = GOLD["f_axis"] if k == "f_axis" else GOLD[system + "_" + k]
            assert np.shape(v) == ref.shape, k
            assert np.allclose(v, ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max()), k


def test_psd_estimate_counts_the_padded_remainder_as_a_slice():
    x = np.arange(10, dtype=complex)
    ref = (np.abs(np.fft.fftshift(np.fft.fft(x[:4]))) ** 2 + np.abs(np.fft.fftshift(np.fft.fft(x[4:8]))) ** 2
           + np.abs(np.fft.fftshift(np.fft.fft(x[8:], 4))) ** 2) / 3
    assert np.allclose(T.psd_estimate(x, 4), ref)
    # exact multiple: the empty remainder still counts in the divisor (lines 118-121)
    assert np.allclose(T.psd_estimate(x[:8], 4), ref * 3 / 3 - np.abs(np.fft.fftshift(np.fft.fft(x[8:], 4))) ** 2 / 3)


def test_overlap_and_add_is_the_kernel_frame_layout():
    st = V.make_structure("wtx", 64, 12)
    rs = np.random.RandomState(0)
    x = rs.randn(5, st.sym_len) + 1j * rs.randn(5, st.sym_len)
    y = T.overlap_and_add(x, st.tail_tx)
    assert y.size == st.frame_len(5)
    ref = np.zeros(st.frame_len(5), complex)
    for s in range(5):
        ref[s * st.stride:s * st.stride + st.sym_len] += x[s]
    assert np.allclose(y, ref)


def test_windowing_lowers_the_out_of_band_radiation():
    st = V.make_structure("wtx", 256, 16)
    opt, rc, cp = T.estimate_obr(st, V.tx_rc_window(st), rng=np.random.RandomState(4))
    assert rc["obr_rc"] < cp["obr_cp"]
    assert np.isclose(opt["obr_opt"], rc["obr_rc"])


def test_timefreq_fun_writes_the_reference_files(tmp_path):
    st = V.make_structure("WOLA", 128, 12)
    vec = np.r_[1.0, np.linspace(0.9, 0.1, 8), 1.0, np.linspace(0.45, 0.05, 5)]
    os.makedirs(tmp_path / "win")
    np.save(tmp_path / "win" / "WOLA_12.npy", vec.reshape(-1, 1))
    T.timefreq_fun(("WOLA", 128, 12, 8, 10, str(tmp_path / "win"), str(tmp_path / "sim")),
                   rng=np.random.RandomState(1))
    for name, key in (("opt_WOLA_12.npz", "X_est_opt"), ("rc_WOLA_12.npz", "X_est_rc"), ("CP_12.npz", "X_est_cp")):
        d = np.load(tmp_path / "sim" / "timefreq" / name)
        assert d[key].shape == (8 * st.n_fft,)
