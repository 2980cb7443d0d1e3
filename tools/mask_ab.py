#!/usr/bin/env python3
"""Developer tool (GPU box): the Tx-mask kernels against each other -- layout 15 (everything on the matrix pipe), layout 9
(plan option dft_valu) and layout 1 (fir_valu) must count the same errors on the same frames -- and the instrumented kernel's
'tx' stage against the oracle for a number of frames.

    python tools/mask_ab.py [frames_per_cell] [n_dumps]
"""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import wofdm_amd as W  # noqa: E402
from wofdm_amd import channel_mask as CM  # noqa: E402
from oracle import oracle as O  # noqa: E402

ch = np.load(os.path.join(R, "tests", "golden", "channels_vehA.npz"))["h"]
F = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ND = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for system, cp in (("wtx", 32), ("WOLA", 44)):
    st = W.make_structure(system, 256, cp) if system == "wtx" else W.make_structure(system, 256, cp, 8, 10)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    snr = np.arange(0.0, 36.0, 6.0).astype(np.float32)
    h = ch[3:5].astype(np.complex64)
    cfg = W.make_cfg(st, 4, 16, 21, 2, snr.size, 1, seed=21)
    active, mask = CM.half_band_allocation(256), CM.tx_mask(st.sym_len).astype(np.float32)
    res = {}
    for name, opts in (("15", {}), ("9", {"dft_valu": 1}), ("1", {"fir_valu": 1})):
        with W.Plan(cfg, w_tx, w_rx, h, snr) as plan:
            for k, v in opts.items():
                plan.set_option(k, v)
            plan.set_allocation(active); plan.set_tx_mask(mask)
            kid = plan.kernel_id()
            res[name] = (kid, plan.run(7, F)[..., 0].ravel().copy(), plan.run(7, F)[..., 0].ravel().copy())
    ref = res["1"][1]
    for name, (kid, a, b) in res.items():
        print("%-5s kernel %s  repeat identical %s  bit errors %s  vs layout 1: max |diff| %d  sum |diff| %d of %d"
              % (system, kid, np.array_equal(a, b), a[:4], np.abs(a.astype(np.int64) - ref.astype(np.int64)).max(),
                 np.abs(a.astype(np.int64) - ref.astype(np.int64)).sum(), ref.sum()))
    osys = O.make_sys(256, 4, 16, st.cp, st.cs, st.tail_tx, st.tail_rx, st.prefix_rm, st.circ_shift, 21, 1, active=active,
                      tx_mask=mask.astype(np.float64))
    worst = []
    with W.Plan(cfg, w_tx, w_rx, h, snr) as plan:
        plan.set_allocation(active); plan.set_tx_mask(mask)
        kid = plan.kernel_id()
        for d in range(ND):
            cell, frame = d % (2 * snr.size), 100 + 17 * d
            lab, noise = O.gen_labels(osys, 21, cell, frame), O.gen_noise(osys, 21, cell, frame)
            oc, od = O.frame(osys, w_tx.astype(np.float64), w_rx.astype(np.float64), h[cell % 2].astype(np.complex128),
                             float(snr[cell // 2]), lab, noise, dump=True)
            gc, gd = plan.dump_frame(cell, frame)
            worst.append(max(float(np.abs(gd[s_][:od[s_].size] - od[s_]).max() / np.abs(od[s_]).max()) for s_ in ("tx", "conv", "Y")))
    print("%-5s instrumented kernel %s, %d frames: worst stage error %.2e (all: %s)" % (system, kid, ND, max(worst), " ".join("%.1e" % w for w in worst)))
