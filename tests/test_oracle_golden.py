"""The CPU oracle (oracle/wofdm_oracle.c) against fixtures produced by the reference itself
(tests/golden/make_golden.py).  This is what pins the oracle; the GPU parity tests then
compare the HIP path with the oracle."""
import numpy as np
import pytest

from oracle import oracle as O

SYSTEMS = ["wtx", "wrx", "WOLA", "CPW", "CPwtx", "CPwrx", "CP"]
# wofdm_simulation.py:179-182
SYM16 = np.array((-3-3j, -3-1j, -3+1j, -3+3j, -1-3j, -1-1j, -1+1j, -1+3j, 1-3j, 1-1j,
                  1+1j, 1+3j, 3-3j, 3-1j, 3+1j, 3+3j))


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        assert tuple(int(x) for x in O.philox(ctr, key)) == want


@pytest.mark.parametrize("n", [2, 8, 64, 256, 1024])
def test_fft_matches_numpy(n):
    rs = np.random.RandomState(n)
    x = rs.randn(n) + 1j * rs.randn(n)
    assert np.allclose(O.fft(x, -1), np.fft.fft(x), rtol=0, atol=1e-11)
    assert np.allclose(O.fft(x, +1), np.fft.ifft(x), rtol=0, atol=1e-13)


def test_qam_tables_matlab_gray():
    # qammod(0:M-1, M) Gray order as listed in SURVEY.md 3.4-2
    t4 = O.qam_table(2) * np.sqrt(2)
    assert np.allclose(t4, [-1+1j, -1-1j, 1+1j, 1-1j])
    t16 = O.qam_table(4) * np.sqrt(10)
    want = [-3+3j, -3+1j, -3-3j, -3-1j, -1+3j, -1+1j, -1-3j, -1-1j,
            3+3j, 3+1j, 3-3j, 3-1j, 1+3j, 1+1j, 1-3j, 1-1j]
    assert np.allclose(t16, want)
    for k in (2, 4, 6):
        t = O.qam_table(k)
        assert np.isclose(np.mean(np.abs(t) ** 2), 1.0)
        # Gray: nearest neighbours differ in exactly one bit
        d = np.abs(t[:, None] - t[None, :])
        dmin = d[d > 0].min()
        for a in range(t.size):
            for b in range(t.size):
                if a != b and np.isclose(d[a, b], dmin):
                    assert bin(a ^ b).count("1") == 1


def _sys_from_cfg(cfg, n_taps, matlab_order, S=16, k=4):
    n_fft, cp, cs, ttx, trx, rm, shift = [int(v) for v in cfg[:7]]
    return O.make_sys(n_fft, k, S, cp, cs, ttx, trx, rm, shift, n_taps, 1 if matlab_order else 0)


def test_stage_chain_against_reference_matrices(golden):
    """tx / conv / Rx-DFT outputs of the reference's dense-matrix chain (noise-free)."""
    g = golden("stages.npz")
    keys = sorted(k[:-4] for k in g.files if k.endswith("_cfg"))
    assert len(keys) == 9
    for key in keys:
        cfg = g[key + "_cfg"]
        h = g[key + "_h"]
        sys = _sys_from_cfg(cfg, h.size, matlab_order=False)
        labels = g[key + "_labels"].T.copy()        # [S, N]
        noise = np.ones(O.noise_len(sys), dtype=np.complex128)
        counts, st = O.frame(sys, g[key + "_wtx"], g[key + "_wrx"], h, 400.0, labels, noise,
                             table=g[key + "_table"], nearest=True, dump=True)
        scale = np.abs(g[key + "_tx"]).max()
        assert np.abs(st["tx"] - g[key + "_tx"]).max() < 1e-12 * max(scale, 1), key
        assert np.abs(st["conv"] - g[key + "_conv"]).max() < 1e-12 * max(scale, 1), key
        Y = g[key + "_Y"].T                          # [S, N]
        assert np.abs(st["Y"] - Y).max() < 1e-9 * np.abs(Y).max(), key
        assert counts[3] == 15 * sys.n_fft


def _replay_case(g, idx):
    key = "case%d" % idx
    system = str(g[key + "_system"])
    n_fft, cp, cs, ttx, trx, rm, shift, ens, seed = [int(v) for v in g[key + "_cfg"]]
    snr, h = g[key + "_snr"], g[key + "_h"]         # h [n_ch][taps]
    S = 16
    sys = O.make_sys(n_fft, 4, S, cp, cs, ttx, trx, rm, shift, h.shape[1], 0)
    B = sys.B
    if system == "CP":
        pairs = [(np.ones(sys.P), np.ones(n_fft + trx))]
    else:
        pairs = [(g[key + "_wtx"], g[key + "_wrx"]), (g[key + "_wtx_rc"], g[key + "_wrx_rc"])]
    rs = np.random.RandomState(seed)                # legacy stream == np.random.seed(seed)
    ser = np.zeros((len(pairs), snr.size))
    for si, s_db in enumerate(snr):
        for ci in range(h.shape[0]):
            acc = np.zeros(len(pairs))
            for _ in range(ens):
                lab = rs.choice(16, size=(n_fft, S)).T.astype(np.uint8)   # wofdm_simulation.py:183
                for pi, (wt, wr) in enumerate(pairs):
                    nre = rs.randn(S * B)                                 # :136, real part first
                    nim = rs.randn(S * B)
                    c, _ = O.frame(sys, wt, wr, h[ci], s_db, lab, nre + 1j * nim,
                                   table=SYM16, nearest=True)
                    acc[pi] += c[2] / c[3]
            ser[:, si] += acc / ens
    ser /= h.shape[0]
    return system, ser


def test_seeded_ser_replay_matches_reference(golden):
    """End-to-end: same numpy draws -> the reference simulator's SER, exactly."""
    g = golden("ser_replay.npz")
    n = int(g["n_cases"])
    assert n == 10
    for idx in range(n):
        system, ser = _replay_case(g, idx)
        key = "case%d" % idx
        if system == "CP":
            assert np.array_equal(ser[0], g[key + "_ser_cp"]), (idx, system)
        else:
            assert np.abs(ser[0] - g[key + "_ser_opt"]).max() < 1e-15, (idx, system)
            assert np.abs(ser[1] - g[key + "_ser_rc"]).max() < 1e-15, (idx, system)


def test_slicer_equals_nearest_point_on_gray_tables(channels):
    """qamdemod restated as a per-axis slicer must agree with exhaustive nearest-point."""
    sys = O.make_sys(64, 6, 8, 16, 8, 8, 0, 16, 0, 21, 1)
    rs = np.random.RandomState(5)
    labels = rs.randint(0, 64, size=(8, 64)).astype(np.uint8)
    noise = rs.randn(O.noise_len(sys)) + 1j * rs.randn(O.noise_len(sys))
    from wofdm_amd import variants as V
    st = V.make_structure("wtx", 64, 16)
    a, sa = O.frame(sys, V.tx_rc_window(st), V.rx_rc_window(st), channels[0], 18.0, labels, noise,
                    dump=True)
    b, sb = O.frame(sys, V.tx_rc_window(st), V.rx_rc_window(st), channels[0], 18.0, labels, noise,
                    table=O.qam_table(6), nearest=True, dump=True)
    assert np.array_equal(sa["labels_rx"], sb["labels_rx"]) and np.array_equal(a, b)
    assert 0 < a[0] < a[1]


def test_interference_power_matches_closed_form(golden):
    """Noise-free pre-equaliser residual power of symbols s >= 1 equals the reference's
    closed-form ICI+ISI power diag(PISI + PICI1) (interf_calc.py:91-100), per subcarrier."""
    g = golden("interference.npz")
    h = g["h"]
    rs = np.random.RandomState(3)
    from wofdm_amd import variants as V
    for system in SYSTEMS:
        n_fft, cp, cs, ttx, trx, rm, shift = [int(v) for v in g[system + "_cfg"]]
        S, frames = 16, 400
        sys = O.make_sys(n_fft, 2, S, cp, cs, ttx, trx, rm, shift, h.size, 0)
        st = V.Structure(system, n_fft, cp, ttx, trx, cs, rm, shift)
        wt, wr = V.tx_rc_window(st), V.rx_rc_window(st)
        noise = np.ones(O.noise_len(sys), dtype=np.complex128)
        # A0 diagonal from an impulse-free estimate: least squares of Y on X over many frames
        num = np.zeros(n_fft, complex); den = np.zeros(n_fft)
        Ys, Xs = [], []
        for _ in range(frames):
            lab = rs.randint(0, 4, size=(S, n_fft)).astype(np.uint8)
            _, d = O.frame(sys, wt, wr, h, 400.0, lab, noise, dump=True)
            Ys.append(d["Y"][1:]); Xs.append(d["X"][1:])
        Y = np.concatenate(Ys); X = np.concatenate(Xs)
        a0 = (Y * np.conj(X)).sum(0) / (np.abs(X) ** 2).sum(0)
        resid = (np.abs(Y - a0 * X) ** 2).mean(0)
        ratio = resid / g[system + "_P_rc"]
        assert abs(ratio.mean() - 1.0) < 0.05, (system, ratio.mean())


def test_generate_mode_streams_are_frame_and_cell_keyed():
    sys = O.make_sys(64, 2, 16, 16, 8, 8, 0, 16, 0, 21, 1)
    a = O.gen_labels(sys, 7, 3, 11)
    assert a.shape == (16, 64) and a.max() <= 3
    assert not np.array_equal(a, O.gen_labels(sys, 7, 4, 11))
    assert not np.array_equal(a, O.gen_labels(sys, 7, 3, 12))
    assert np.array_equal(a, O.gen_labels(sys, 7, 3, 11))
    n = np.concatenate([O.gen_noise(sys, 1, 0, f) for f in range(40)])
    assert abs(n.real.var() - 1) < 0.03 and abs(n.imag.var() - 1) < 0.03
    assert abs(np.mean(n.real * n.imag)) < 0.02


def test_run_accumulates_and_shards_exactly(channels):
    from wofdm_amd import variants as V
    st = V.make_structure("WOLA", 64, 16)
    sys = O.make_sys(64, 4, 16, st.cp, st.cs, st.tail_tx, st.tail_rx, st.prefix_rm, st.circ_shift,
                     21, 1)
    args = (sys, V.tx_rc_window(st), V.rx_rc_window(st), channels[:2], [5.0, 25.0], 9)
    whole = O.run(*args, 0, 12)
    parts = O.run(*args, 0, 5) + O.run(*args, 5, 7)
    assert np.array_equal(whole, parts)
    assert whole.shape == (1, 2, 2, 4)
    assert (whole[..., 1] == 12 * 15 * 64 * 4).all() and (whole[..., 3] == 12 * 15 * 64).all()
    ber = whole[..., 0] / whole[..., 1]
    assert (ber[0, 0] > ber[0, 1]).all()


def test_oracle_under_address_and_ub_sanitizers():
    """gcc -fsanitize=address,undefined build of the oracle runs every structure family clean."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "oracle"), "selftest"], check=True,
                   stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", OMP_NUM_THREADS="2")
    r = subprocess.run([os.path.join(root, "oracle", "selftest_asan")], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "selftest ok" in r.stdout, r.stdout + r.stderr


def test_dense_cost_structure_gives_the_same_counters():
    """wofdm_oracle_run_dense (bench.py's "faithful" CPU leg: the reference's hoisted tx_mat / rx_mat
    products, wofdm_simulation.py:464-471) must count exactly what the FFT form counts."""
    import os
    import wofdm_amd as W
    ch = np.load(os.path.join(os.path.dirname(__file__), "golden", "channels_vehA.npz"))["h"][:2]
    for system, n, cp in (("wtx", 256, 32), ("WOLA", 64, 16), ("CPW", 128, 20), ("wrx", 64, 12)):
        st = W.make_structure(system, n, cp)
        osys = O.make_sys(n, 4, 16, st.cp, st.cs, st.tail_tx, st.tail_rx, st.prefix_rm, st.circ_shift, 21, 1)
        args = (osys, W.tx_rc_window(st), W.rx_rc_window(st), ch, [5.0, 15.0], 3)
        a, b = O.run(*args, 2, 6), O.run(*args, 2, 6, dense=True)
        assert np.array_equal(a, b) and a[..., 0].min() > 0
