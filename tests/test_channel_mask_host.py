"""Host side of the main_channel_mask.m variant (row f1), no GPU: the mask generator, the
allocation, and the oracle's mask / allocation stages against an independent numpy restatement."""
import os

import numpy as np
import pytest

import wofdm_amd as W
from oracle import oracle as O
from wofdm_amd import channel_mask as CM

CH = np.load(os.path.join(os.path.dirname(__file__), "golden", "channels_vehA.npz"))["h"]


def test_raised_cosine_mask_shape():
    # main_channel_mask.m:443-458 with P = 296: 591 bins, 295 ones, 10-sample edges, 138 zeros a side
    m = CM.gen_raised_cosine(295, 10, 591)
    assert m.size == 591 and (m[:138] == 0).all() and (m[-138:] == 0).all()
    assert (m[148:148 + 295] == 1).all()
    assert np.allclose(m[138:148], np.sin(np.pi / 2 * (0.5 + np.arange(-4.5, 5.0) / 10)) ** 2)
    assert np.allclose(m, m[::-1])                              # P even: even mask, real response
    nat = CM.tx_mask(296)
    assert nat[0] == 1 and np.allclose(nat[1:], nat[1:][::-1])
    assert np.abs(np.fft.ifft(nat).imag).max() < 1e-15
    # P odd (wrx: cs = 5, CPW: cs = 13): the centred mask is lopsided, its response complex
    assert np.abs(np.fft.ifft(CM.tx_mask(293)).imag).max() > 1e-4
    # odd remainder: the extra zero goes to the right (ceil), line 452-453
    m2 = CM.gen_raised_cosine(100, 10, 225)
    assert m2.size == 225 and np.flatnonzero(m2)[0] == 52 and np.flatnonzero(m2)[-1] == 52 + 119


def test_half_band_allocation_is_the_ifftshifted_centre():
    n = 64
    centred = np.r_[np.zeros(16), np.ones(32), np.zeros(16)]    # [zeros(offset) data zeros(offset)]
    assert np.array_equal(CM.half_band_allocation(n), np.fft.ifftshift(centred) != 0)


@pytest.mark.parametrize("system,n_fft,cp", [("wtx", 64, 16), ("WOLA", 128, 20), ("CP", 128, 32)])
def test_oracle_mask_and_allocation_stages(system, n_fft, cp):
    S, k = 6, 4
    st = W.make_structure(system, n_fft, cp)
    active = CM.half_band_allocation(n_fft)
    mask = CM.tx_mask(st.sym_len)
    osys = O.make_sys(n_fft, k, S, st.cp, st.cs, st.tail_tx, st.tail_rx, st.prefix_rm, st.circ_shift,
                      21, 1, active=active, tx_mask=mask)
    lab, noise = O.gen_labels(osys, 3, 0, 5), O.gen_noise(osys, 3, 0, 5)
    w_tx, w_rx = W.tx_rc_window(st), W.rx_rc_window(st)
    c, d = O.frame(osys, w_tx, w_rx, CH[0], 200.0, lab, noise, dump=True)
    assert (d["X"][:, ~active] == 0).all() and (np.abs(d["X"][:, active]) > 0).all()
    rows = np.fft.ifft(d["X"], axis=1)[:, (np.arange(st.sym_len) - st.cp) % n_fft] * w_tx[None, :]
    filt = CM.dft_rc_filt(rows)
    tx = np.zeros(st.frame_len(S), complex)
    for s in range(S):
        tx[s * st.stride:s * st.stride + st.sym_len] += filt[s]
    assert np.abs(d["tx"] - tx).max() < 1e-13 * np.abs(tx).max() + 1e-16
    assert int(c[1]) == (S - 1) * int(active.sum()) * k and int(c[3]) == (S - 1) * int(active.sum())
    # noise-free: the long impulse response of the mask (and its tail landing at the start of the
    # next row rather than where it belongs in time) leaves a little ISI, nothing more
    assert int(c[0]) < 0.05 * int(c[1])
    # the mask only removes out-of-band leakage: in-band energy is (almost) untouched
    osys2 = O.make_sys(n_fft, k, S, st.cp, st.cs, st.tail_tx, st.tail_rx, st.prefix_rm,
                       st.circ_shift, 21, 1, active=active)
    c2, d2 = O.frame(osys2, w_tx, w_rx, CH[0], 200.0, lab, noise, dump=True)
    assert int(c2[0]) < 0.05 * int(c2[1])                       # (21 taps against a short CP)
    e_m, e_p = np.sum(np.abs(d["tx"]) ** 2), np.sum(np.abs(d2["tx"]) ** 2)
    assert 0.9 < e_m / e_p <= 1.0 + 1e-9


def test_results_and_files(tmp_path):
    rs = np.random.RandomState(0)
    masked = rs.randint(1, 50, size=(7, 3, 2, 4)).astype(np.uint64)
    plain = rs.randint(1, 50, size=(7, 3, 2, 4)).astype(np.uint64)
    masked[..., 1] = plain[..., 1] = 1000
    names = [n for n, _ in W.variants.matlab_pair_plan("WOLA")]
    res = CM.results_from_counts(names, masked, plain)
    assert set(res) == {"berRCSNR", "berMaskedRCSNR"} | {p + s for p in ("berSNRStep", "berMaskedSNRStep")
                                                        for s in ("1A", "2A", "3A", "1B", "2B", "3B")}
    assert np.allclose(res["berMaskedSNRStep2B"], (masked[5, :, :, 0] / 1000).mean(axis=-1))
    paths = CM.save_results(str(tmp_path), "WOLA", 16, res)
    assert sorted(os.path.basename(p) for p in paths) == [
        "masked_optimized_ber_WOLA_16CP.mat", "masked_rc_ber_WOLA_16CP.mat",
        "optimized_ber_WOLA_16CP.mat", "rc_ber_WOLA_16CP.mat"]
    from scipy.io import loadmat
    m = loadmat(paths[0])
    assert any(k.startswith("ber") for k in m)
    res1 = CM.results_from_counts(["opt", "rc"], masked[:2], plain[:2])
    assert set(res1) == {"berSNR", "berMaskedSNR", "berRCSNR", "berMaskedRCSNR"}
