import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import wofdm_amd as W
from oracle import oracle as O
ch = np.load(os.path.join(R, "tests", "golden", "channels_vehA.npz"))["h"]
system, n_fft, cp, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
S, seed, frame = 16, 11, 123456789012
st = W.make_structure(system, n_fft, cp)
w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
h = ch[4:7].astype(np.complex64); snrs = np.array([8.0, 22.0], dtype=np.float32)
cfg = W.make_cfg(st, k, S, 21, 3, 2, 1, seed=seed)
osys = O.make_sys(st.n_fft, k, S, st.cp, st.cs, st.tail_tx, st.tail_rx, st.prefix_rm, st.circ_shift, 21, 1)
cell = 4
lab = O.gen_labels(osys, seed, cell, frame); noise = O.gen_noise(osys, seed, cell, frame)
oc, od = O.frame(osys, w_tx.astype(np.float64), w_rx.astype(np.float64), h[1].astype(np.complex128), float(snrs[1]), lab, noise, dump=True)
with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
    print(plan.kernel_id(), st)
    gc, gd = plan.dump_frame(cell, frame)
Y, Yo = gd["Y"], od["Y"]
e = np.abs(Y - Yo) / np.abs(Yo).max()
print("Y err per symbol", e.max(axis=1))
print("worst subcarriers of symbol 1:", np.argsort(e[1])[-12:], e[1][np.argsort(e[1])[-12:]])
ratio = Y[1] / Yo[1]
print("ratio first 8", ratio[:8]); print("ratio |.| min/max", np.abs(ratio).min(), np.abs(ratio).max())
print("phase step", np.angle(ratio[1:9] / ratio[0:8]))
for sy in (5, 6):
    es = e[sy]
    bad = np.where(es > 1e-4)[0]
    print("symbol", sy, "bad subcarriers:", len(bad), bad[:40])
    # reconstruct the FFT input the kernel must have used: ifft of the (un-ramped) Y vs of the oracle's
    zi = np.fft.ifft(Y[sy]); zo = np.fft.ifft(Yo[sy])
    d = np.abs(zi - zo) / np.abs(zo).max()
    print("  time-domain differences at:", np.where(d > 1e-4)[0][:40], d.max())
print("=== repeat runs: wrong symbols and wrong input positions")
for rep in range(4):
    with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
        gc, gd = plan.dump_frame(cell, frame)
    Y = gd["Y"]; out = []
    for sy in range(S):
        zi = np.fft.ifft(Y[sy]); zo = np.fft.ifft(Yo[sy]); d = np.abs(zi - zo) / np.abs(zo).max()
        w = np.where(d > 1e-4)[0]
        if len(w): out.append((sy, int(w.min()), int(w.max()), len(w)))
    print(rep, out)
