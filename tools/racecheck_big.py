#!/usr/bin/env python3
"""Developer tool (GPU box): run-to-run repeatability of the one-symbol-per-wave kernels (layout 8 / 1).
    python tools/racecheck_big.py [n_fft=1024] [k=6] [frames=108] [runs=12]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wofdm_amd as W
n, k, F, R = [int(v) for v in (sys.argv[1:] + ["1024", "6", "108", "12"][len(sys.argv) - 1:])]
ch = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "channels_vehA.npz"))["h"]
st = W.make_structure("WOLA", n, 32)
snrs = np.array([5.0, 15.0, 25.0], np.float32) + (k - 4) * 3.0
cfg = W.make_cfg(st, k, 16, 21, 2, 3, 1, seed=8)
with W.Plan(cfg, W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32), ch[11:13].astype(np.complex64), snrs) as plan:
    if os.environ.get("WOFDM_FIR_VALU") == "1":      # (tool switch: the round-1 kernels, FIR on the VALU)
        plan.set_option("fir_valu", 1)
    print("kernel", plan.kernel_id())
    runs = [plan.run(3, F)[..., 0].ravel() for _ in range(R)]
ref = runs[0]
for i, r in enumerate(runs):
    print(i, r.tolist(), "" if np.array_equal(r, ref) else "DIFFERENT")
