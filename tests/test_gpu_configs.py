"""BASELINE.json configs as parity cases.  bench.py measures C2; the
others are checked here: against the oracle at sizes it finishes in seconds, and through
size-independent properties at full size."""
import os

import numpy as np
import pytest

import wofdm_amd as W
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _osys(st, k, S, n_taps, matlab=True):
    return O.make_sys(st.n_fft, k, S, st.cp, st.cs, st.tail_tx, st.tail_rx, st.prefix_rm,
                      st.circ_shift, n_taps, 1 if matlab else 0)


def _close(got, want, tol_frac=1e-4):
    assert np.array_equal(got[..., 1], want[..., 1]) and np.array_equal(got[..., 3], want[..., 3])
    tol = max(2.0, tol_frac * float(want[..., 1].max()))
    assert (np.abs(got[..., 0].astype(np.int64) - want[..., 0].astype(np.int64)) <= tol).all()
    assert (np.abs(got[..., 2].astype(np.int64) - want[..., 2].astype(np.int64)) <= tol).all()


def test_c1_plumbing_n64_qpsk(channels):
    """C1: wtx, N=64, QPSK, CP=16, RC Tx window, 1 channel, SNR {0,15,30} dB, 63 frames."""
    st = W.make_structure("wtx", 64, 16)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), np.ones(st.rx_win_len, np.float32)
    snr = np.array([0.0, 15.0, 30.0], np.float32)
    cfg = W.make_cfg(st, 2, 16, 21, 1, 3, 1, seed=1, frames_per_cell=63)
    got = W.run_counts(cfg, w_tx, w_rx, channels[:1].astype(np.complex64), snr)
    want = O.run(_osys(st, 2, 16, 21), w_tx.astype(np.float64), w_rx.astype(np.float64),
                 channels[:1].astype(np.complex64).astype(np.complex128), snr.astype(np.float64), 1, 0, 63)
    assert got[0, 0, 0, 3] == 63 * 15 * 64          # 1008 symbols incl. pilots = 945 data symbols
    _close(got, want)


def _tail_vectors(st, rs):
    xt = np.concatenate(([1.0], np.sort(rs.uniform(0.05, 0.95, st.tail_tx))[::-1])) if st.tail_tx else np.ones(1)
    xr = np.concatenate(([1.0], np.sort(rs.uniform(0.05, 0.45, st.tail_rx // 2))[::-1])) if st.tail_rx else np.ones(1)
    return xt, xr


def test_c3_all_variants_with_loaded_windows(channels, tmp_path):
    """C3: the six structures with 'optimised' windows loaded from tail-vector files in the
    reference's on-disk format (+ plain CP), N=256, 16-QAM, CP 32, 20 dB, through the
    simulation_fun mirror; counters vs the oracle on a frame subset, SER files written."""
    rs = np.random.RandomState(3)
    chan_path = tmp_path / "vehicularA.npy"
    np.save(chan_path, channels[:1].T)              # [taps, n_ch] like the reference
    win_dir = tmp_path / "windows"
    win_dir.mkdir()
    for system in W.SYSTEMS:
        st = W.make_structure(system, 256, 32)
        xt, xr = _tail_vectors(st, rs)
        if system in ("WOLA", "CPW"):
            vec = np.concatenate((xt, xr))
        elif system in ("wtx", "CPwtx"):
            vec = xt
        else:
            vec = xr
        if system != "CP":
            np.save(win_dir / ("%s_32.npy" % system), vec)
        tails = W.variants.default_tails(system)
        out = W.simulation_fun((system, 256, 32, tails[0], tails[1], str(chan_path), str(win_dir),
                                200, np.array([20.0]), 16, str(tmp_path / "out")))
        ser = out if system == "CP" else out[0]
        assert 0.05 < float(ser[0]) < 0.2           # BASELINE.md anchor: ~0.11 at 20 dB
        # same windows straight through the ABI vs the oracle (Python noise order, like the mirror)
        w_tx = (W.expand_tx_window(st, xt) if st.tail_tx else np.ones(st.sym_len)).astype(np.float32)
        w_rx = (W.expand_rx_window(st, xr) if st.tail_rx else np.ones(st.rx_win_len)).astype(np.float32)
        cfg = W.make_cfg(st, 4, 16, 21, 1, 1, 1, noise_before_truncate=False, seed=0, frames_per_cell=24)
        got = W.run_counts(cfg, w_tx, w_rx, channels[:1].astype(np.complex64), [20.0])
        want = O.run(_osys(st, 4, 16, 21, matlab=False), w_tx.astype(np.float64), w_rx.astype(np.float64),
                     channels[:1].astype(np.complex64).astype(np.complex128), [20.0], 0, 0, 24)
        _close(got, want)
    assert (tmp_path / "out" / "ser" / "opt_WOLA_32.npy").exists()
    assert (tmp_path / "out" / "ser" / "CP_32.npy").exists()


def test_c3_full_size_counters_and_matlab_driver(channels):
    """C3 at full size (1e6 symbols per variant) through the MATLAB driver mirror: counter totals
    are exact, BER of the RC pair sits where the other variants do, 7 pairs for WOLA."""
    st = W.make_structure("WOLA", 256, 32)
    rs = np.random.RandomState(4)
    xt, xr = _tail_vectors(st, rs)
    wins = {"optimizedWindowCaseAStep1": np.diag(W.expand_tx_window(st, xt)),
            "optimizedWindowCaseAStep2": np.diag(W.expand_rx_window(st, xr)),
            "optimizedWindowCaseAStep3": W.expand_tx_window(st, xt),
            "optimizedWindowCaseBStep1": W.expand_rx_window(st, xr),
            "optimizedWindowCaseBStep2": W.expand_tx_window(st, xt),
            "optimizedWindowCaseBStep3": W.expand_rx_window(st, xr)}
    res, counts = W.ber_for_window_file("WOLA", 32, wins, channels[:1], [20.0], ensemble=62500, seed=3)
    assert counts.shape == (7, 1, 1, 4)
    assert (counts[..., 1] == 62500 * 15 * 256 * 4).all() and (counts[..., 3] == 62500 * 15 * 256).all()
    assert set(res) == {"berRCSNR", "berSNRStep1A", "berSNRStep2A", "berSNRStep3A", "berSNRStep1B",
                        "berSNRStep2B", "berSNRStep3B"}
    ber = np.array([float(v[0]) for v in res.values()])
    assert (ber > 0.01).all() and (ber < 0.06).all()
    assert ber.std() / ber.mean() < 0.25


def test_c4_wola_n1024_64qam_cells(channels):
    """C4 (reduced frames): WOLA, N=1024, 64-QAM, 100 channels x 20 SNR points = 2000 cells."""
    st = W.make_structure("WOLA", 1024, 32)
    assert (st.cs, st.prefix_rm, st.circ_shift) == (8, 22, 5)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    snr = (-20.0 + 3.0 * np.arange(20)).astype(np.float32)
    h = channels.astype(np.complex64)
    cfg = W.make_cfg(st, 6, 16, 21, 100, 20, 1, seed=4, frames_per_cell=2, frame_offset=7)
    got = W.run_counts(cfg, w_tx, w_rx, h, snr)
    assert got.shape == (1, 20, 100, 4) and (got[..., 1] == 2 * 15 * 1024 * 6).all()
    # oracle on a slice of the cell grid (SNR points 3, 11, 19 x channels 0..4) -- cell numbering
    # must agree, so run the oracle on the full grid shape but compare the slice only
    sub_snr, sub_ch = [3, 11, 19], list(range(5))
    osys = _osys(st, 6, 16, 21)
    for si in sub_snr:
        for ci in sub_ch:
            cell = si * 100 + ci
            tot = np.zeros(4, np.uint64)
            for f in (7, 8):
                lab = O.gen_labels(osys, 4, cell, f)
                nz = O.gen_noise(osys, 4, cell, f)
                c, _ = O.frame(osys, w_tx.astype(np.float64), w_rx.astype(np.float64),
                               h[ci].astype(np.complex128), float(snr[si]), lab, nz)
                tot += c
            assert abs(int(got[0, si, ci, 0]) - int(tot[0])) <= 20, (si, ci)
            assert int(got[0, si, ci, 1]) == int(tot[1])
    ber = got[0, :, :, 0].sum(axis=1) / got[0, :, :, 1].sum(axis=1)
    assert (np.diff(ber) < 0).all() and ber[0] > 0.3


def test_c4_at_the_references_size_adds_up(channels):
    """C4 as the reference runs it (2 000 cells x 100 frames of 16 symbols, 3.2e6 OFDM symbols): the counters of
    one launch equal the sum over three unequal frame ranges bit for bit, the bit totals are the closed form, and
    the BER curve averaged over the 100 channels falls with the SNR."""
    st = W.make_structure("WOLA", 1024, 32)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    snr = (-20.0 + 3.0 * np.arange(20)).astype(np.float32)
    h = channels.astype(np.complex64)
    cfg = W.make_cfg(st, 6, 16, 21, 100, 20, 1, seed=4)
    with W.Plan(cfg, w_tx, w_rx, h, snr) as plan:
        whole = plan.run(0, 100)
        parts = plan.run(0, 37) + plan.run(37, 1) + plan.run(38, 62)
    assert np.array_equal(whole, parts)
    assert (whole[..., 1] == 100 * 15 * 1024 * 6).all() and (whole[..., 3] == 100 * 15 * 1024).all()
    ber = whole[0, :, :, 0].sum(axis=1) / whole[0, :, :, 1].sum(axis=1)
    assert (np.diff(ber) < 0).all() and ber[0] > 0.3 and ber[-1] < 0.05


@pytest.mark.parametrize("n_fft,k", [(256, 2), (512, 4), (1024, 6), (512, 2), (256, 6)])
def test_c5_sweep_sample(channels, n_fft, k):
    """C5 sample: a few (N, QAM) points of the full sweep, all seven structures as window
    pairs are not mixable across structures, so per structure: WOLA and CPwtx here."""
    for system in ("WOLA", "CPwtx"):
        st = W.make_structure(system, n_fft, 32)
        w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
        snr = np.array([4.0, 16.0, 28.0], np.float32)
        h = channels[40:42].astype(np.complex64)
        cfg = W.make_cfg(st, k, 16, 21, 2, 3, 1, seed=5, frames_per_cell=6)
        got = W.run_counts(cfg, w_tx, w_rx, h, snr)
        want = O.run(_osys(st, k, 16, 21), w_tx.astype(np.float64), w_rx.astype(np.float64),
                     h.astype(np.complex128), snr.astype(np.float64), 5, 0, 6)
        _close(got, want, tol_frac=2e-4)


@pytest.mark.parametrize("system", list(W.SYSTEMS))
@pytest.mark.parametrize("n_fft", [256, 512, 1024])
@pytest.mark.parametrize("k", [2, 4, 6])
def test_c5_full_grid(channels, system, n_fft, k):
    """BASELINE config C5 as a grid: all 7 structures x {256, 512, 1024} x {QPSK, 16-QAM, 64-QAM} = 63
    points (wofdm_simulation.py:391-418 / main_BER_calculation.m:467-492 give the structure table),
    each 2 channels x 2 SNR points x 3 frames against the oracle on the same Philox streams.  At these
    sizes a single wrong sample in a frame shows: the tolerance is 6 bit errors per cell."""
    st = W.make_structure(system, n_fft, 32)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    snr = np.array([6.0, 22.0], np.float32) + (k - 4) * 3.0
    h = channels[50:52].astype(np.complex64)
    cfg = W.make_cfg(st, k, 16, 21, 2, 2, 1, seed=5, frames_per_cell=3)
    got = W.run_counts(cfg, w_tx, w_rx, h, snr)
    want = O.run(_osys(st, k, 16, 21), w_tx.astype(np.float64), w_rx.astype(np.float64),
                 h.astype(np.complex128), snr.astype(np.float64), 5, 0, 3)
    assert np.array_equal(got[..., 1], want[..., 1]) and np.array_equal(got[..., 3], want[..., 3])
    d = np.abs(got[..., 0].astype(np.int64) - want[..., 0].astype(np.int64))
    assert d.max() <= 6 and want[0, 0, :, 0].min() > 100, (d, want[..., 0])


def test_matlab_script_mirror_end_to_end(channels, tmp_path):
    """main_BER_calculation.m as a function: window .mat files in, ber_results/*.mat out with the
    reference's variable names; curves vs the oracle through the same cell numbering."""
    from scipy.io import loadmat, savemat
    D = W.driver
    settings = {k: dict(v) for k, v in D.DEFAULT_SETTINGS.items()}
    settings["generalSettings"].update(ensemble=12, snrValues=np.array([5.0, 20.0, 35.0]))
    D.save_settings(str(tmp_path / "settingsData.mat"), settings)
    (tmp_path / "optimized_windows").mkdir()
    (tmp_path / "channels").mkdir()
    savemat(str(tmp_path / "channels" / "vehA200channel2.mat"), {"vehA200channel2": channels[:3]})
    rs = np.random.RandomState(8)
    st_w = W.make_structure("wtx", 256, 32)
    xt = np.concatenate(([1.0], np.sort(rs.uniform(.05, .95, 8))[::-1]))
    savemat(str(tmp_path / "optimized_windows" / "optimal_win_wtx_VehA200_32CP.mat"),
            {"optimizedWindow": np.diag(W.expand_tx_window(st_w, xt))})
    st_c = W.make_structure("CPwrx", 256, 14)
    xr = np.concatenate(([1.0], np.sort(rs.uniform(.05, .45, 5))[::-1]))
    savemat(str(tmp_path / "optimized_windows" / "optimal_win_CPwrx_VehA200_14CP.mat"),
            {"optimizedWindow": np.diag(W.expand_rx_window(st_c, xr))})
    (tmp_path / "optimized_windows" / "run.log").write_text("not a window file")
    out = D.run_ber_calculation(str(tmp_path / "settingsData.mat"), str(tmp_path / "optimized_windows"),
                                str(tmp_path / "channels" / "vehA200channel2.mat"),
                                str(tmp_path / "ber_results"), seed=6, log=None)
    assert set(out) == {"optimal_win_wtx_VehA200_32CP.mat", "optimal_win_CPwrx_VehA200_14CP.mat"}
    a = loadmat(str(tmp_path / "ber_results" / "optimized_ber_wtx_32CP.mat"))["berSNR"].ravel()
    b = loadmat(str(tmp_path / "ber_results" / "rc_ber_wtx_32CP.mat"))["berRCSNR"].ravel()
    assert a.shape == (3,) and (np.diff(a) < 0).all() and (np.diff(b) < 0).all()
    # oracle: pair 0 = (optimised Tx, RC Rx), pair 1 = (RC, RC); mean over the 3 channels
    w_tx = np.stack([W.expand_tx_window(st_w, xt), W.tx_rc_window(st_w)]).astype(np.float32)
    w_rx = np.stack([W.rx_rc_window(st_w)] * 2).astype(np.float32)
    want = O.run(_osys(st_w, 4, 16, 21), w_tx.astype(np.float64), w_rx.astype(np.float64),
                 channels[:3].astype(np.complex64).astype(np.complex128), [5.0, 20.0, 35.0], 6, 0, 12)
    ber = want[..., 0].sum(axis=2) / want[..., 1].sum(axis=2)
    assert np.abs(a - ber[0]).max() < 2e-5 and np.abs(b - ber[1]).max() < 2e-5
    assert (tmp_path / "ber_results" / "optimized_ber_CPwrx_14CP.mat").exists()


def test_reference_workflow_windows_then_ber_then_mask(channels, tmp_path):
    """The reference's scripts in sequence, every hand-over through its own file formats:
    window_optimization.m (window_design, MATLAB flavour) -> optimal_win_*.mat ->
    main_BER_calculation.m (driver, GPU) -> ber_results/*.mat, and main_channel_mask.m
    (channel_mask, GPU) on the same window files; plus the Python route
    optimization_fun -> <sys>_<cp>.npy -> simulation_fun -> ser/*.npy."""
    from scipy.io import loadmat, savemat
    D, WD, CM = W.driver, W.window_design, W.channel_mask
    settings = {k: dict(v) for k, v in D.DEFAULT_SETTINGS.items()}
    snr = np.array([0.0, 15.0, 30.0])
    settings["generalSettings"].update(numberSubcarriers=64, cyclicPrefix=np.array([12, 16]), ensemble=40,
                                       snrValues=snr)
    D.save_settings(str(tmp_path / "settingsData.mat"), settings)
    (tmp_path / "channels").mkdir()
    savemat(str(tmp_path / "channels" / "vehA200channel2.mat"), {"vehA200channel2": channels[:4]})
    files = WD.run_window_optimization(channels[:4], settings, str(tmp_path / "optimized_windows"),
                                       systems=["wtx", "WOLA"])
    assert sorted(os.path.basename(f) for f in files) == [
        "optimal_win_WOLA_VehA200_12CP.mat", "optimal_win_WOLA_VehA200_16CP.mat",
        "optimal_win_wtx_VehA200_12CP.mat", "optimal_win_wtx_VehA200_16CP.mat"]
    out = D.run_ber_calculation(str(tmp_path / "settingsData.mat"), str(tmp_path / "optimized_windows"),
                                str(tmp_path / "channels" / "vehA200channel2.mat"),
                                str(tmp_path / "ber_results"), seed=3, log=None)
    assert len(out) == 4
    m = loadmat(str(tmp_path / "ber_results" / "optimized_ber_WOLA_16CP.mat"))
    assert {"berSNRStep1A", "berSNRStep2A", "berSNRStep3A", "berSNRStep1B", "berSNRStep2B",
            "berSNRStep3B"} <= set(m)
    rc = loadmat(str(tmp_path / "ber_results" / "rc_ber_wtx_12CP.mat"))["berRCSNR"].ravel()
    opt = loadmat(str(tmp_path / "ber_results" / "optimized_ber_wtx_12CP.mat"))["berSNR"].ravel()
    assert rc.shape == opt.shape == (3,) and rc[0] > rc[2] and opt[0] > opt[2]
    # CP = 12 < channel length: at 30 dB the interference floor dominates and the optimised Tx
    # window (less ICI+ISI by construction) must not be worse than the raised cosine
    assert opt[2] <= rc[2] * 1.05 + 1e-4
    # main_channel_mask.m on one of the files
    win = D.load_window_file(str(tmp_path / "optimized_windows" / "optimal_win_wtx_VehA200_16CP.mat"))
    res, _ = CM.ber_for_window_file("wtx", 16, win, channels[:4], snr, num_subcar=64, ensemble=40,
                                    tail_tx=8, tail_rx=0, seed=5)
    assert set(res) == {"berSNR", "berMaskedSNR", "berRCSNR", "berMaskedRCSNR"}
    paths = CM.save_results(str(tmp_path / "ber_results" / "simulation_with_channel_mask"), "wtx", 16, res)
    assert len(paths) == 4 and all(os.path.exists(p) for p in paths)
    assert res["berSNR"][0] > res["berSNR"][2] and res["berMaskedSNR"][0] > res["berMaskedSNR"][2]
    # the Python route
    cpath = tmp_path / "vehicularA.npy"
    np.save(cpath, channels[:4].T)
    x, _ = WD.optimization_fun(("CPwtx", 64, 16, str(cpath), str(tmp_path / "pywin")))
    ser = W.simulation_fun(("CPwtx", 64, 16, 8, 0, str(cpath), str(tmp_path / "pywin"), 20, snr, 16,
                            str(tmp_path / "pysim")))
    assert os.path.exists(tmp_path / "pysim" / "ser" / "opt_CPwtx_16.npy")
    assert os.path.exists(tmp_path / "pysim" / "ser" / "rc_CPwtx_16.npy")


def test_interference_closed_form_on_gpu(golden, channels):
    """Row f2 on the GPU (wofdm_interference): against the reference's own interf_power output
    (tests/golden/interference.npz, N = 64, all seven structures), and at N = 256 / 512 / 1024, batched over
    window pairs and channels, against the host mirror of the same formula (fp64)."""
    from wofdm_amd import variants as V
    g = golden("interference.npz")
    for system in W.SYSTEMS:
        n_fft, cp, cs, ttx, trx, rm, shift = [int(v) for v in g[system + "_cfg"]]
        st = V.Structure(system, n_fft, cp, ttx, trx, cs, rm, shift)
        got = W.interference.interf_power_gpu(st, V.tx_rc_window(st), V.rx_rc_window(st), g["h"])[0, 0]
        want = g[system + "_P_rc"]
        assert np.abs(got - want).max() < 2e-5 * np.abs(want).max() + 1e-9, system
    rs = np.random.RandomState(2)
    for system, n_fft, cp in (("WOLA", 256, 32), ("CPW", 256, 10), ("wtx", 512, 24), ("wrx", 1024, 32), ("CP", 256, 16)):
        st = W.make_structure(system, n_fft, cp)
        xt, xr = _tail_vectors(st, rs)
        w_tx = np.stack([W.tx_rc_window(st), W.expand_tx_window(st, xt) if st.tail_tx else np.ones(st.sym_len)])
        w_rx = np.stack([W.rx_rc_window(st), W.expand_rx_window(st, xr) if st.tail_rx else np.ones(st.rx_win_len)])
        h = channels[3:6]
        got = W.interference.interf_power_gpu(st, w_tx, w_rx, h)
        assert got.shape == (2, 3, n_fft)
        for pi in range(2):
            for ci in range(3):
                want = W.interference.interf_power(st, w_tx[pi], w_rx[pi], h[ci])
                assert np.abs(got[pi, ci] - want).max() < 5e-5 * np.abs(want).max() + 1e-9, (system, pi, ci)


@pytest.mark.parametrize("system", ["wtx", "CPW", "wrx", "CPwtx"])
def test_tx_psd_on_gpu_replays_the_reference(golden, system):
    """Row f4 on the GPU (wofdm_tx_psd): the reference's own seeded estimate_obr outputs
    (tests/golden/timefreq.npz, N = 128: PSD estimate, OBR, main-band samples of the optimised, RC and plain
    CP waveforms) reproduced with the waveform and the averaged periodogram computed by the HIP kernels."""
    from wofdm_amd import timefreq as T
    from wofdm_amd import variants as V
    g = golden("timefreq.npz")
    n_fft, cp = (int(v) for v in g["cfg"])
    st = V.make_structure(system, n_fft, cp)
    w_tx = V.expand_tx_window(st, g[system + "_xt"])
    dicts = T.estimate_obr(st, w_tx, 200e-9, rng=np.random.RandomState(int(g[system + "_seed"])), gpu=True)
    for tag, d in zip(("opt", "rc", "cp"), dicts):
        for key in ("X_est_" + tag, "obr_" + tag, "mf_band_" + tag):
            ref = g[system + "_" + key]
            assert np.allclose(d[key], ref, rtol=2e-4, atol=2e-5 * np.abs(g[system + "_X_est_" + tag]).max()), key


@pytest.mark.parametrize("system,n_fft,cp", [("WOLA", 256, 32), ("CP", 64, 16), ("wtx", 256, 10)])
def test_tx_psd_on_gpu_matches_the_host_mirror(system, n_fft, cp):
    from wofdm_amd import timefreq as T
    st = W.make_structure(system, n_fft, cp)
    rs = np.random.RandomState(n_fft + cp)
    gb = 48 if n_fft >= 128 else 8
    X = T.draw_symbols(n_fft, rs, no_symbols=100, guard_band=gb)
    w_tx = W.tx_rc_window(st)
    ov = st.tail_tx
    want = T.psd_estimate(T.overlap_and_add(T.tx_symbols(st, X, w_tx, gb), ov), 8 * n_fft)
    got = T.psd_estimate_gpu(st, X, w_tx, ov, guard_band=gb)
    assert np.abs(got - want).max() < 2e-5 * np.abs(want).max()
