#!/usr/bin/env python3
"""Developer tool (GPU box): bit-error counts of the production (non-instrumented) kernels vs the
oracle on the same Philox streams, ~1e7 bits per cell -- differences beyond a handful mean a
wrong sample somewhere, long before any BER curve moves."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wofdm_amd as W  # noqa: E402
from oracle import oracle as O  # noqa: E402

ch = np.load(os.path.join(ROOT, "tests/golden/channels_vehA.npz"))["h"]
for system, n, cp, k, S in [("WOLA", 512, 32, 4, 16), ("CPwtx", 512, 20, 2, 16), ("wtx", 256, 32, 4, 16),
                            ("CPW", 64, 16, 2, 16), ("WOLA", 1024, 32, 6, 16), ("wrx", 128, 20, 6, 16),
                            ("WOLA", 512, 32, 6, 9), ("CPW", 256, 24, 4, 7)]:
    st = W.make_structure(system, n, cp)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    snrs = np.array([5.0, 15.0, 25.0], np.float32)
    F = max(4, int(1e7 / ((S - 1) * n * k)))
    cfg = W.make_cfg(st, k, S, 21, 2, 3, 1, seed=8)
    h = ch[11:13].astype(np.complex64)
    with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
        if os.environ.get("WOFDM_FIR_VALU") == "1":      # (tool switch: the round-1 kernels, FIR on the VALU)
            plan.set_option("fir_valu", 1)
        got = plan.run(3, F)
    osys = O.make_sys(n, k, S, st.cp, st.cs, st.tail_tx, st.tail_rx, st.prefix_rm, st.circ_shift, 21, 1)
    want = O.run(osys, w_tx.astype(np.float64), w_rx.astype(np.float64), h.astype(np.complex128),
                 snrs.astype(np.float64), 8, 3, F)
    d = got[..., 0].astype(np.int64) - want[..., 0].astype(np.int64)
    print("%-6s N=%-4d k=%d S=%-2d bits/cell %.2e errors %s diff %s" % (
        system, n, k, S, float(want[0, 0, 0, 1]), want[0, :, 0, 0].tolist(), d.reshape(-1).tolist()))
