#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE itself.

Build-container-only tool: it imports the reference's Python packages from
/root/reference/python (never copied, never shipped) and writes small .npz
data files -- inputs and the reference's outputs -- next to this script.
Nothing under tests/ or the library imports this module; the GPU box never
runs it (there is no /root/reference there).

The reference imports `numba` (absent from this image) only for its
`@njit`/`@jit` decorators; they are replaced by identity decorators through a
throw-away module created in a temp dir, so the reference's *own* Python code
runs unmodified, just un-jitted.

Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference/python"
HERE = os.path.dirname(os.path.abspath(__file__))

if not os.path.isdir(REF):
    sys.exit("make_golden.py: %s not found - this tool only runs in the build container" % REF)

sys.dont_write_bytecode = True
_tmp = tempfile.mkdtemp(prefix="wofdm_golden_")
os.makedirs(os.path.join(_tmp, "numba"))
with open(os.path.join(_tmp, "numba", "__init__.py"), "w") as f:
    f.write("def _ident(*a, **k):\n"
            "    if len(a) == 1 and callable(a[0]) and not k:\n"
            "        return a[0]\n"
            "    return lambda fn: fn\n"
            "njit = jit = _ident\n")
sys.path[:0] = [_tmp, REF]

import channel_model  # noqa: E402  (reference)
import ofdm_utils  # noqa: E402  (reference)
from ofdm_utils import wofdm_simulation as ref_sim  # noqa: E402
from ofdm_utils.interf_calc import interf_power  # noqa: E402
from optimization_tools.utils import reduce_variable_rx, reduce_variable_tx  # noqa: E402

SYSTEMS = ["wtx", "wrx", "WOLA", "CPW", "CPwtx", "CPwrx", "CP"]
TAILS = {"wtx": (8, 0), "wrx": (0, 10), "WOLA": (8, 10), "CPW": (8, 10),
         "CPwtx": (8, 0), "CPwrx": (0, 10), "CP": (0, 0)}
# wofdm_simulation.py:179-182
SYM16 = np.array((-3-3j, -3-1j, -3+1j, -3+3j, -1-3j, -1-1j, -1+1j, -1+3j, 1-3j, 1-1j,
                  1+1j, 1+3j, 3-3j, 3-1j, 3+1j, 3+3j))


def gen_channels(n_ch, seed):
    """wofdm_optimization.py:66-86 (mode gen_chan), seeded."""
    from scipy.constants import speed_of_light
    np.random.seed(seed)
    fc, ts, v, taps, nsym, nfft = 2e9, 200e-9, 100 / 3.6, 21, 16, 256
    fd = (v / speed_of_light) * fc
    out = np.zeros((taps, n_ch), dtype=np.complex128)
    for i in range(n_ch):
        out[:, i] = channel_model.gen_chan("vehicularA", taps, fd, 1 / ts, nsym * nfft * ts, 1)[:, 0]
    return out  # [taps, n_ch] (Python layout, wofdm_simulation.py:49,205)


def tail_vectors(system, rs):
    """Deliberately non-RC 'optimised' tail vectors in the on-disk format of
    optimization_tools/utils.py:13-73 (x0 = flat level, then the tail)."""
    btx, brx = TAILS[system]
    xt = np.array([1.0]) if btx == 0 else np.concatenate(
        ([1.0 + 0.05 * rs.randn()], np.sort(rs.uniform(0.02, 0.98, btx))[::-1]))
    xr = np.array([1.0]) if brx == 0 else np.concatenate(
        ([1.0 + 0.05 * rs.randn()], np.sort(rs.uniform(0.02, 0.48, brx // 2))[::-1]))
    return xt, xr


def build_system(system, n_fft, cp, out_dir):
    btx, brx = TAILS[system]
    return ref_sim.wOFDMSystem(system, n_fft, cp, btx, brx, out_dir)


def windows_for(model, xt, xr):
    wtx = np.diagflat(reduce_variable_tx(model.dft_len, model.cp_len, model.cs_len,
                                         model.tail_tx) @ xt.reshape(-1, 1))
    wrx = np.diagflat(reduce_variable_rx(model.dft_len, model.tail_rx) @ xr.reshape(-1, 1))
    return wtx, wrx


def fixture_channels():
    ch = gen_channels(100, 2024)
    np.savez(os.path.join(HERE, "channels_vehA.npz"), h=ch.T.copy(), seed=2024,
             note="h[n_ch=100][taps=21] complex128, row = realisation; gen_chan('vehicularA',"
                  " 21, fd(2GHz,100km/h), 5e6, 16*256*200e-9, 1) under np.random.seed(2024)")
    return ch


def fixture_params():
    """(rho, gamma, kappa) and RC windows of every structure from the reference."""
    out = {}
    for system in SYSTEMS:
        for n_fft, cp in ((64, 16), (256, 32), (256, 10)):
            m = build_system(system, n_fft, cp, _tmp)
            key = "%s_N%d_cp%d" % (system, n_fft, cp)
            out[key + "_params"] = np.array([m.cs_len, m.rm_len, m.shift_len, m.tail_tx, m.tail_rx])
            if m.tail_tx:
                out[key + "_rc_tx"] = np.diag(ofdm_utils.gen_rc_window_tx(n_fft, cp, m.cs_len, m.tail_tx))
            if m.tail_rx:
                out[key + "_rc_rx"] = np.diag(ofdm_utils.gen_rc_window_rx(n_fft, m.tail_rx))
    rs = np.random.RandomState(7)
    for system in SYSTEMS:
        m = build_system(system, 64, 16, _tmp)
        xt, xr = tail_vectors(system, rs)
        wtx, wrx = windows_for(m, xt, xr)
        out[system + "_tailvec_tx"] = xt
        out[system + "_tailvec_rx"] = xr
        out[system + "_expanded_tx"] = np.diag(wtx)
        out[system + "_expanded_rx"] = np.diag(wrx)
    np.savez(os.path.join(HERE, "structure_params.npz"), **out)


def fixture_stages(ch):
    """Deterministic (noise-free) stage outputs of the reference's dense-matrix
    chain, composed exactly as wofdm_simulation.py:464-471,187-222 does."""
    out = {}
    rs = np.random.RandomState(11)
    S = 16
    for system, n_fft, cp in [(s, 64, 16) for s in SYSTEMS] + [("wtx", 256, 32), ("WOLA", 256, 32)]:
        m = build_system(system, n_fft, cp, _tmp)
        xt, xr = tail_vectors(system, rs)
        wtx, wrx = windows_for(m, xt, xr)
        tx_mat = wtx @ m.add_red_mat @ m.idft_mat
        rx_mat = m.dft_mat @ m.circ_shift_mat @ m.overlap_add_mat @ wrx @ m.rm_red_mat
        # symbols from an arbitrary 16-point complex alphabet (not a QAM grid), so that the
        # check does not depend on any constellation convention
        table = rs.randn(16) + 1j * rs.randn(16)
        labels = rs.randint(0, 16, size=(n_fft, S))
        X = table[labels]
        h = ch[:, 3]
        tail = m.tail_tx
        frame_tx = (tx_mat @ X).T
        sig_ov = frame_tx[:, tail:].copy()
        if tail:
            sig_ov[:-1, -tail:] += frame_tx[1:, :tail]
        signal_tx = np.hstack((frame_tx[0, :tail], sig_ov.flatten()))
        conv = np.convolve(h, signal_tx)
        trunc = conv[:-(len(h) + tail - 1)]
        frame_rx = trunc.reshape((S, rx_mat.shape[1]))
        pre = rx_mat @ frame_rx.T
        key = "%s_N%d_cp%d" % (system, n_fft, cp)
        out[key + "_table"] = table
        out[key + "_labels"] = labels.astype(np.uint8)   # [N, S]
        out[key + "_cfg"] = np.array([n_fft, cp, m.cs_len, m.tail_tx, m.tail_rx, m.rm_len, m.shift_len])
        out[key + "_wtx"] = np.diag(wtx)
        out[key + "_wrx"] = np.diag(wrx)
        out[key + "_h"] = h
        out[key + "_tx"] = signal_tx
        out[key + "_conv"] = conv
        out[key + "_Y"] = pre          # [N, S]
    np.savez(os.path.join(HERE, "stages.npz"), **out)


def fixture_ser_replay(ch):
    """Seeded end-to-end runs of the reference simulator.  The test replays the
    same legacy-RandomState draws (np.random.seed(seed); per ensemble iteration:
    choice(16,(N,S)) -> randn(S*B) re, randn(S*B) im for the optimised window ->
    the same two draws for the RC window; wofdm_simulation.py:171-215) into the
    oracle and must reproduce these SER arrays."""
    out = {}
    rs = np.random.RandomState(23)
    cases = [(s, 64, 16, 2, np.array([0., 12., 24., 36.]), 3) for s in SYSTEMS]
    cases += [("wtx", 256, 32, 1, np.array([5., 20., 35.]), 2),
              ("WOLA", 256, 32, 1, np.array([5., 20., 35.]), 2),
              ("CPW", 256, 10, 1, np.array([20., 40.]), 2)]
    for idx, (system, n_fft, cp, n_ch, snr, ens) in enumerate(cases):
        seed = 1000 + idx
        m = build_system(system, n_fft, cp, os.path.join(_tmp, "ser_%d" % idx))
        xt, xr = tail_vectors(system, rs)
        wtx, wrx = windows_for(m, xt, xr)
        chans = ch[:, 5:5 + n_ch]
        # check the draw-order assumption the test relies on
        np.random.seed(seed)
        a = np.random.choice(SYM16, size=(n_fft, 16), replace=True)
        b = SYM16[np.random.RandomState(seed).choice(16, size=(n_fft, 16))]
        assert np.array_equal(a, b)
        np.random.seed(seed)
        m.run_simulation(chans, wtx, wrx, ens, snr, 16)
        d = os.path.join(m.folder_path, "ser")
        key = "case%d" % idx
        out[key + "_system"] = system
        out[key + "_cfg"] = np.array([n_fft, cp, m.cs_len, m.tail_tx, m.tail_rx, m.rm_len,
                                      m.shift_len, ens, seed])
        out[key + "_snr"] = snr
        out[key + "_h"] = chans.T.copy()
        if system == "CP":
            out[key + "_ser_cp"] = np.load(os.path.join(d, "CP_%d.npy" % cp))
        else:
            out[key + "_wtx"] = np.diag(wtx)
            out[key + "_wrx"] = np.diag(wrx)
            out[key + "_wtx_rc"] = np.diag(ofdm_utils.gen_rc_window_tx(n_fft, cp, m.cs_len, m.tail_tx))
            out[key + "_wrx_rc"] = np.diag(ofdm_utils.gen_rc_window_rx(n_fft, m.tail_rx))
            out[key + "_ser_opt"] = np.load(os.path.join(d, "opt_%s_%d.npy" % (system, cp)))
            out[key + "_ser_rc"] = np.load(os.path.join(d, "rc_%s_%d.npy" % (system, cp)))
    out["n_cases"] = len(cases)
    np.savez(os.path.join(HERE, "ser_replay.npz"), **out)


def fixture_interference(ch):
    """Closed-form ICI+ISI power (interf_calc.py:20-113) for RC windows and the
    mean channel; the analytic, RNG-free cross-check of SURVEY.md section 4."""
    out = {}
    cwd = os.getcwd()
    work = os.path.join(_tmp, "interf")
    os.makedirs(os.path.join(work, "channels"))
    np.save(os.path.join(work, "channels", "vehicularA.npy"), ch[:, 9:10])
    os.chdir(work)
    try:
        for system in SYSTEMS:
            btx, brx = TAILS[system]
            n_fft, cp = 64, 12
            m = build_system(system, n_fft, cp, work)
            if system == "CP":
                p = interf_power(system, [], n_fft, cp, btx, brx)
                out[system + "_P_rc"] = p
            else:
                vtx = ofdm_utils.gen_rc_window_tx(n_fft, cp, m.cs_len, btx) if btx else np.eye(n_fft + cp + m.cs_len)
                vrx = ofdm_utils.gen_rc_window_rx(n_fft, brx) if brx else np.eye(n_fft + brx)
                p_opt, p_rc = interf_power(system, [vtx, vrx], n_fft, cp, btx, brx)
                out[system + "_P_rc"] = p_rc
            out[system + "_cfg"] = np.array([n_fft, cp, m.cs_len, btx, brx, m.rm_len, m.shift_len])
        out["h"] = ch[:, 9].copy()
    finally:
        os.chdir(cwd)
    np.savez(os.path.join(HERE, "interference.npz"), **out)


def fixture_window_design(ch):
    """Window design (optimizers.py:25-873): the reference's Hessians (its own O(P^2 N^2) loops,
    un-jitted) and the minimisers of its own solvers, at N=32, CP=12, tails 8/10, for the mean
    of the first 20 channel realisations.  (Re-running this reproduces the Hessians to 1e-15;
    the WOLA/CPW minimisers of scipy's trust-constr move by ~3e-8 from run to run -- threaded
    BLAS -- far inside the 2e-5 the test allows.)"""
    from optimization_tools.optimizers import OptimizerRx, OptimizerTx, OptimizerTxRx
    n_fft, cp = 32, 12
    h_avg = ch[:, :20].mean(axis=1)
    out = {"h_avg": h_avg, "cfg": np.array([n_fft, cp, 8, 10])}
    for system in ("wtx", "CPwtx", "wrx", "CPwrx", "WOLA", "CPW"):
        if system in ("wtx", "CPwtx"):
            m, reg = OptimizerTx(system, n_fft, cp, 8), 1e-12
        elif system in ("wrx", "CPwrx"):
            m, reg = OptimizerRx(system, n_fft, cp, 10), 1e-12
        else:
            m, reg = OptimizerTxRx(system, n_fft, cp, 8, 10), 1e-16
        H = m.gen_hessian(m.calculate_chann_matrices(h_avg))
        Hs = .5 * (H + H.T)                                    # optimizers.py:61-64
        x, _ = m.optimize(Hs + reg * np.eye(H.shape[0]), 1e-20)
        out[system + "_H"] = H
        out[system + "_x"] = np.asarray(x, dtype=np.float64).reshape(-1)
    np.savez(os.path.join(HERE, "window_design.npz"), **out)


def fixture_timefreq():
    """PSD / OBR estimate (timefreq_simulation.py:216-296) with the legacy global generator
    seeded (the replay draws the same 16-QAM symbols from RandomState(seed)): N=128, CP=12,
    non-RC tail vectors."""
    from ofdm_utils import timefreq_simulation as tf
    rs = np.random.RandomState(77)
    out = {"cfg": np.array([128, 12])}
    for i, system in enumerate(("wtx", "CPW", "wrx", "CPwtx")):
        btx, brx = TAILS[system]
        n_fft, cp = 128, 12                      # guard band 48 needs N > 96
        xt, _ = tail_vectors(system, rs)
        m = tf.wOFDMSystem(system, n_fft, cp, btx, brx, _tmp)
        win = np.diagflat(reduce_variable_tx(n_fft, cp, m.cs_len, btx) @ xt.reshape(-1, 1))
        np.random.seed(1000 + i)
        opt, rc, cpd = m.estimate_obr(win, 200e-9)
        out[system + "_seed"] = np.array(1000 + i)
        out[system + "_xt"] = xt
        for d in (opt, rc, cpd):
            for k, v in d.items():
                if k == "f_axis" and i:
                    continue
                out[("" if k == "f_axis" else system + "_") + k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "timefreq.npz"), **out)


if __name__ == "__main__":
    if sys.argv[1:] == ["timefreq"]:
        fixture_timefreq()
        sys.exit(0)
    if sys.argv[1:] == ["window_design"]:
        fixture_window_design(np.load(os.path.join(HERE, "channels_vehA.npz"))["h"].T)
        sys.exit(0)
    ch = fixture_channels()
    fixture_params()
    fixture_stages(ch)
    fixture_ser_replay(ch)
    fixture_interference(ch)
    fixture_window_design(ch)
    fixture_timefreq()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("%-28s %8d bytes" % (f, os.path.getsize(os.path.join(HERE, f))))
