#!/usr/bin/env python3
"""Developer tool: one timed launch of one config (for rocprofv3 --pmc runs).
    python tools/run_one.py SYSTEM N K N_CH N_SNR FRAMES        (RUN_MASK=1: half-band allocation + Tx mask, row f1)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wofdm_amd as W
system, n, k, nch, nsnr, frames = sys.argv[1], *[int(x) for x in sys.argv[2:7]]
ch = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "channels_vehA.npz"))["h"]
st = W.make_structure(system, n, int(os.environ.get("RUN_CP", "32")))
cfg = W.make_cfg(st, k, 16, 21, nch, nsnr, 1, seed=4)
snr = (-20 + 3.0 * np.arange(nsnr)).astype(np.float32)
with W.Plan(cfg, W.tx_rc_window(st), W.rx_rc_window(st), ch[:nch].astype(np.complex64), snr) as plan:
    if os.environ.get("RUN_MASK") == "1":
        from wofdm_amd import channel_mask as CM
        plan.set_allocation(CM.half_band_allocation(n))
        plan.set_tx_mask(CM.tx_mask(st.sym_len))
    counts = plan.new_counts()
    ms = [plan.launch_timed(i * frames, frames, counts) for i in range(3)]
    print(system, n, k, "ms", ms, "sym/s %.3e" % (frames * 16 * nch * nsnr / min(ms) * 1e3), plan.info())
