#!/usr/bin/env python3
"""Developer tool (GPU box): strides of 2 mod 4, and odd strides, in the one-symbol-per-wave matrix-pipe layouts (12, 15) against layout 1
(plan option fir_valu), which ran them before round 3 / round 4.   python tools/bench_even_strides.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import wofdm_amd as W  # noqa: E402
from wofdm_amd import channel_mask as CM  # noqa: E402

ch = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "channels_vehA.npz"))["h"]
snr = np.arange(-5.0, 51.0, 5.0).astype(np.float32)
for system, n, cp, k, F, masked in (("wtx", 256, 30, 4, 20000, True), ("WOLA", 512, 30, 4, 10000, False), ("WOLA", 1024, 30, 6, 5000, False),
                                    # odd strides (wrx / CPW / CPwrx: delta = 10): layout 12 since round 4
                                    ("CPW", 512, 32, 4, 10000, False), ("CPW", 1024, 32, 6, 5000, False), ("CPW", 256, 32, 4, 20000, True)):
    st = W.make_structure(system, n, cp)
    cfg = W.make_cfg(st, k, 16, 21, 1, snr.size, 1, seed=3)
    for opts in ({}, {"fir_valu": 1}):
        with W.Plan(cfg, W.tx_rc_window(st), W.rx_rc_window(st), ch[:1].astype(np.complex64), snr) as plan:
            for key, val in opts.items():
                plan.set_option(key, val)
            if masked:
                plan.set_allocation(CM.half_band_allocation(n))
                plan.set_tx_mask(CM.tx_mask(st.sym_len))
            c = plan.new_counts()
            plan.launch(0, F // 10, c)
            ms = min(plan.launch_timed((i + 1) * F, F, c) for i in range(3))
            print("%-5s N=%-4d cp=%d stride %-4d %-6s kernel %-7s %7.2f ms  %.3e sym/s"
                  % (system, n, cp, st.stride, "masked" if masked else "plain", plan.kernel_id(), ms, F * 16 * snr.size / ms * 1e3))
