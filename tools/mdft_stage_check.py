import sys, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import wofdm_amd as W
from oracle import oracle as O
ch = np.load(os.path.join(R, "tests", "golden", "channels_vehA.npz"))["h"]
def rel(a,b): return float(np.abs(a-b).max()/np.abs(b).max())
for system, n_fft, cp, k in [("wtx",64,16,2),("WOLA",64,16,4),("CPW",64,16,6),("wrx",64,16,2),("WOLA",128,32,4),("wtx",128,16,6),("CPwrx",128,20,2),("wtx",64,64,2)]:
    S, seed, frame = 16, 11, 123456789012
    st = W.make_structure(system, n_fft, cp)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    h = ch[4:7].astype(np.complex64)
    snrs = np.array([8.0, 22.0], dtype=np.float32)
    cfg = W.make_cfg(st, k, S, 21, 3, 2, 1, seed=seed)
    osys = O.make_sys(st.n_fft, k, S, st.cp, st.cs, st.tail_tx, st.tail_rx, st.prefix_rm, st.circ_shift, 21, 1)
    cell = 4
    lab = O.gen_labels(osys, seed, cell, frame); noise = O.gen_noise(osys, seed, cell, frame)
    oc, od = O.frame(osys, w_tx.astype(np.float64), w_rx.astype(np.float64), h[1].astype(np.complex128), float(snrs[1]), lab, noise, dump=True)
    for dv in (0, 1):
        with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
            plan.set_option("dft_valu", dv)
            kid = plan.kernel_id()
            gc, gd = plan.dump_frame(cell, frame)
        print(system, cp, k, "layout", kid, "labels", np.array_equal(gd["labels_tx"], lab),
              " ".join("%s %.2e" % (n, rel(gd[n], od[n])) for n in ("X","tx","conv","rx","Y","Xhat")), "counts", gc, oc)
