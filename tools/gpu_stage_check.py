#!/usr/bin/env python3
"""Developer tool (GPU box): per-stage comparison of one frame, HIP kernel vs CPU oracle.

    python tools/gpu_stage_check.py [system] [n_fft] [cp] [k] [snr_db]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wofdm_amd as W  # noqa: E402
from oracle import oracle as O  # noqa: E402


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def check(system="wtx", n_fft=256, cp=32, k=4, snr=20.0, matlab=1, inject=False, S=16, seed=5,
          cell=0, frame=3, verbose=True):
    ch = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden",
                              "channels_vehA.npz"))["h"]
    st = W.make_structure(system, n_fft, cp)
    h = ch[:2].astype(np.complex64)
    w_tx = W.tx_rc_window(st).astype(np.float32)
    w_rx = W.rx_rc_window(st).astype(np.float32)
    snrs = np.array([snr, snr + 10], dtype=np.float32)
    cfg = W.make_cfg(st, k, S, 21, 2, 2, 1, noise_before_truncate=bool(matlab), seed=seed)
    osys = O.make_sys(n_fft, k, S, st.cp, st.cs, st.tail_tx, st.tail_rx, st.prefix_rm,
                      st.circ_shift, 21, matlab)
    lab = O.gen_labels(osys, seed, cell, frame)
    noise = O.gen_noise(osys, seed, cell, frame)
    chi, sni = cell % 2, (cell // 2) % 2
    oc, od = O.frame(osys, w_tx.astype(np.float64), w_rx.astype(np.float64),
                     h[chi].astype(np.complex128), float(snrs[sni]), lab, noise, dump=True)
    with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
        if verbose:
            print(plan.info())
        if inject:
            gc, gd = plan.dump_frame(cell, frame, lab, noise.astype(np.complex64))
        else:
            gc, gd = plan.dump_frame(cell, frame)
    res = {"labels_tx_equal": bool(np.array_equal(gd["labels_tx"], lab))}
    nconv = len(od["conv"]) if matlab else S * st.stride   # Python order never needs the tail
    for key in ("X", "tx", "rx", "Y", "Xhat"):
        res[key] = rel(gd[key], od[key])
    res["conv"] = rel(gd["conv"][:nconv], od["conv"][:nconv])
    res["unit_noise"] = rel(gd["unit_noise"], noise)
    res["gain"] = abs(float(gd["gain"][0]) - float(od["gain"][0])) / float(od["gain"][0])
    res["labels_rx_mismatch"] = int((gd["labels_rx"] != od["labels_rx"]).sum())
    res["counts_gpu"] = [int(x) for x in gc]
    res["counts_oracle"] = [int(x) for x in oc]
    if verbose:
        print("%s N=%d cp=%d k=%d snr=%g matlab=%d inject=%d" % (system, n_fft, cp, k, snr, matlab, inject))
        for kk, vv in res.items():
            print("   %-20s %s" % (kk, vv))
    return res


if __name__ == "__main__":
    a = sys.argv[1:]
    if a:
        check(a[0], int(a[1]) if len(a) > 1 else 256, int(a[2]) if len(a) > 2 else 32,
              int(a[3]) if len(a) > 3 else 4, float(a[4]) if len(a) > 4 else 20.0)
    else:
        for system in W.SYSTEMS:
            check(system, 64, 16, 2)
        check("wtx", 256, 32, 4)
        check("WOLA", 256, 32, 4, inject=True)
        check("CPW", 256, 32, 6, matlab=0)
        check("WOLA", 128, 32, 4)
        check("WOLA", 512, 32, 4)
        check("WOLA", 1024, 32, 6)
