#!/usr/bin/env python3
"""Developer tool (GPU box): run-to-run repeatability of the production kernel.

    python tools/racecheck.py [frames_per_cell=4000] [runs=4]

C2 geometry, 12 SNR cells.  Repeated launches of the same frame range must give identical counters,
a split of the range must add up to the whole, and the cells at 45/50 dB must stay error-free (one
corrupted frame there shows up as hundreds of bit errors).  With WOFDM_FIR_VALU=1 the same for the
VALU FIR layouts."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wofdm_amd as W  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ch = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "channels_vehA.npz"))["h"]
st = W.make_structure("wtx", 256, 32)
w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
snrs = np.linspace(-5, 50, 12).astype(np.float32)
cfg = W.make_cfg(st, 4, 16, 21, 1, 12, 1, seed=2)
with W.Plan(cfg, w_tx, w_rx, ch[:1].astype(np.complex64), snrs) as plan:
    if os.environ.get("WOFDM_FIR_VALU") == "1":      # (tool switch: the round-1 kernels, FIR on the VALU)
        plan.set_option("fir_valu", 1)
    runs = [plan.run(0, F) for _ in range(R)]
    parts = plan.run(0, 1) + plan.run(1, F // 3) + plan.run(1 + F // 3, F - 1 - F // 3)
same = all(np.array_equal(runs[0], r) for r in runs[1:]) and np.array_equal(runs[0], parts)
print(os.environ.get("WOFDM_LIB", "default"), "frames", F, "identical" if same else "DIFFERENT",
      "| bit errors at 50 dB:", [int(r[0, -1, 0, 0]) for r in runs], int(parts[0, -1, 0, 0]),
      "| at -5 dB:", [int(r[0, 0, 0, 0]) for r in runs], int(parts[0, 0, 0, 0]))
sys.exit(0 if same else 1)
