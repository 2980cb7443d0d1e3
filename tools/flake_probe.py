import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wofdm_amd as W
ch = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "channels_vehA.npz"))["h"]
def sharp(system, n_fft, cp, k, S, reps=1):
    st = W.make_structure(system, n_fft, cp)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    snrs = np.array([5.0, 15.0, 25.0], np.float32)
    F = max(4, int(1e7 / ((S - 1) * n_fft * k)))
    cfg = W.make_cfg(st, k, S, 21, 2, 3, 1, seed=8)
    out = []
    with W.Plan(cfg, w_tx, w_rx, ch[11:13].astype(np.complex64), snrs) as plan:
        for _ in range(reps):
            out.append(plan.run(3, F)[..., 0].ravel().tolist())
    return out
def repeated(system, n_fft, k, frames):
    st = W.make_structure(system, n_fft, 32)
    cfg = W.make_cfg(st, k, 16, 21, 2, 3, 1, seed=99)
    with W.Plan(cfg, W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32), ch[60:62].astype(np.complex64), np.array([4., 12., 20.], np.float32)) as plan:
        first = plan.run(10, frames)
        for _ in range(4):
            assert np.array_equal(plan.run(10, frames), first)
if "--pre" in sys.argv:
    for a in (("wtx", 256, 4, 6000), ("WOLA", 64, 2, 20000), ("CPW", 512, 4, 1500), ("WOLA", 1024, 6, 700)):
        repeated(*a)
    for a in (("WOLA", 512, 32, 4, 16), ("CPwtx", 512, 20, 2, 16), ("wtx", 256, 32, 4, 16), ("CPW", 64, 16, 2, 16)):
        sharp(*a)
for i in range(3):
    for r in sharp("WOLA", 1024, 32, 6, 16, reps=3):
        print(i, r, "" if r[-1] == 409847 else "  <-- differs from oracle 409847 by %d" % (r[-1] - 409847))
