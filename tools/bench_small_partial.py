#!/usr/bin/env python3
"""Developer tool (GPU box): N = 64 / 128 geometries that layouts 13 / 14 do not take (symbolsPerTx not a multiple of 16 / 8, strides
beyond their tiles) in layout 16 -- a run-time number of symbols per wave, partly filled last wave -- against layout 2 (plan option
dft_valu), which ran them before round 4; the geometries of layouts 13 / 14 beside them.   python tools/bench_small_partial.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import wofdm_amd as W  # noqa: E402

ch = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "channels_vehA.npz"))["h"]
snr = np.arange(-5.0, 51.0, 5.0).astype(np.float32)
for system, n, cp, k, S, F in (("wtx", 64, 16, 2, 16, 160000), ("wtx", 64, 32, 2, 16, 160000), ("WOLA", 64, 16, 2, 12, 160000), ("CPW", 64, 32, 4, 16, 160000),
                               ("WOLA", 128, 32, 4, 16, 80000), ("WOLA", 128, 32, 4, 12, 80000), ("wrx", 128, 56, 4, 16, 80000)):
    st = W.make_structure(system, n, cp)
    cfg = W.make_cfg(st, k, S, 21, 1, snr.size, 1, seed=3)
    for opts in ({}, {"dft_valu": 1}):
        with W.Plan(cfg, W.tx_rc_window(st), W.rx_rc_window(st), ch[:1].astype(np.complex64), snr) as plan:
            for key, val in opts.items():
                plan.set_option(key, val)
            c = plan.new_counts()
            plan.launch(0, F // 10, c)
            ms = min(plan.launch_timed((i + 1) * F, F, c) for i in range(3))
            info = plan.info()
            print("%-5s N=%-4d cp=%d S=%-2d stride %-4d kernel %-7s waves/WG %d WG/CU %-2d %7.2f ms  %.3e sym/s"
                  % (system, n, cp, S, st.stride, plan.kernel_id(), info["waves_per_workgroup"], info["workgroups_per_cu"], ms,
                     F * S * snr.size / ms * 1e3))
