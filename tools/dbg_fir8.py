import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wofdm_amd as W
from oracle import oracle as O
ch = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "channels_vehA.npz"))["h"]
def case(system, n_fft, k, S, btx, brx, cp, taps, matlab, F=3, seed=7, off=1000, n_ch=2):
    st = W.make_structure(system, n_fft, cp, btx, brx)
    w_tx = W.tx_rc_window(st).astype(np.float32); w_rx = W.rx_rc_window(st).astype(np.float32)
    h = ch[:n_ch, :taps].astype(np.complex64); h[:, 0] += 0.5
    snrs = np.array([6.0, 25.0], dtype=np.float32)
    cfg = W.make_cfg(st, k, S, taps, n_ch, 2, 1, noise_before_truncate=matlab, seed=seed)
    osys = O.make_sys(n_fft, k, S, st.cp, st.cs, st.tail_tx, st.tail_rx, st.prefix_rm, st.circ_shift, taps, 1 if matlab else 0)
    with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
        got = [plan.run(off, F)[..., 0].ravel() for _ in range(3)]
        info = plan.info()
    want = O.run(osys, w_tx.astype(np.float64), w_rx.astype(np.float64), h.astype(np.complex128), snrs.astype(np.float64), seed, off, F)[..., 0].ravel()
    ok = all(np.abs(g.astype(np.int64) - want.astype(np.int64)).max() <= 30 for g in got)
    print("OK " if ok else "BAD", system, n_fft, k, "S", S, "btx", btx, "brx", brx, "cp", cp, "taps", taps, "matlab", matlab, "B", st.stride, [g.tolist() for g in got], want.tolist(), info["lds_bytes"])
case("WOLA", 1024, 4, 13, 4, 2, 60, 1, True)
case("WOLA", 1024, 4, 13, 4, 2, 60, 1, False)
case("WOLA", 1024, 4, 16, 4, 2, 28, 1, True)
case("WOLA", 1024, 4, 13, 4, 2, 60, 21, True)
case("WOLA", 1024, 4, 13, 8, 10, 32, 21, True)
case("WOLA", 1024, 4, 16, 8, 10, 32, 21, True)
case("WOLA", 1024, 4, 12, 4, 2, 60, 1, True)
case("WOLA", 1024, 4, 13, 4, 2, 32, 1, True)
case("WOLA", 512, 4, 13, 4, 2, 60, 1, True)
case("WOLA", 1024, 4, 13, 4, 2, 60, 1, True, F=40)
