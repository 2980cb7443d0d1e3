#!/usr/bin/env python3
"""Developer tool (GPU box), hazard 3: which STAGE of a frame goes wrong in a probe build (WOFDM_LIB=...)?
The instrumented kernel (barriers between the phases, one workgroup) on injected frames against the fp64 oracle:
first stage whose dump is off, and the sample indices there.

    WOFDM_LIB=ab/lib_mid.so python tools/hazard3_stages.py [n_frames] [n_fft] [k]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wofdm_amd as W
from oracle import oracle as O
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n_fft = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
k = int(sys.argv[3]) if len(sys.argv) > 3 else 6
S, seed = 16, 5
ch = np.load(os.path.join(ROOT, "tests", "golden", "channels_vehA.npz"))["h"]
st = W.make_structure("WOLA", n_fft, 32)
h = ch[:2].astype(np.complex64)
w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
snrs = np.array([20.0, 30.0], dtype=np.float32)
cfg = W.make_cfg(st, k, S, 21, 2, 2, 1, seed=seed)
osys = O.make_sys(n_fft, k, S, st.cp, st.cs, st.tail_tx, st.tail_rx, st.prefix_rm, st.circ_shift, 21, 1)
def runs(idx):
    out, start, prev = [], int(idx[0]), int(idx[0])
    for v in idx[1:]:
        v = int(v)
        if v - prev > 2: out.append((start, prev)); start = v
        prev = v
    out.append((start, prev))
    return out
with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
    print("kernel", plan.kernel_id(), plan.info())
    for frame in range(n_frames):
        for inject in (1, 0):
            cell = frame % 4
            lab, noise = O.gen_labels(osys, seed, cell, frame), O.gen_noise(osys, seed, cell, frame)
            oc, od = O.frame(osys, w_tx.astype(np.float64), w_rx.astype(np.float64), h[cell % 2].astype(np.complex128),
                             float(snrs[(cell // 2) % 2]), lab, noise, dump=True)
            gc, gd = plan.dump_frame(cell, frame, lab, noise.astype(np.complex64)) if inject else plan.dump_frame(cell, frame)
            line = "frame %2d inject %d counts gpu %s oracle %s:" % (frame, inject, [int(x) for x in gc[:3:2]], [int(x) for x in oc[:3:2]])
            for key in ("X", "tx", "conv", "rx", "Y"):
                a, b = gd[key].reshape(-1), od[key].reshape(-1)
                n = min(len(a), len(b))
                err = np.abs(a[:n] - b[:n]) / np.abs(b).max()
                bad = np.nonzero(err > 1e-4)[0]
                line += "  %s %.1e" % (key, err.max())
                if len(bad):
                    line += " BAD n=%d at %s" % (len(bad), runs(bad)[:8])
                    break
            print(line)
