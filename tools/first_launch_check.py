#!/usr/bin/env python3
"""Developer tool (GPU box): the first-launch anomaly hunt.  Prints the host / GPU identity, then runs the
N = 1024 sharp-parity launch in fresh plans and reports launches that differ from the majority."""
import os, socket, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wofdm_amd as W
try:
    uid = subprocess.run(["rocm-smi", "--showuniqueid", "--showbus"], capture_output=True, text=True, timeout=30).stdout
    uid = " ".join(l.strip() for l in uid.splitlines() if "Unique" in l or "PCI" in l)
except Exception as e:
    uid = repr(e)
print("host", socket.gethostname(), "|", uid[:300])
ch = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "channels_vehA.npz"))["h"]
def runs(n, k, reps, plans=3):
    st = W.make_structure("WOLA", n, 32)
    cfg = W.make_cfg(st, k, 16, 21, 2, 3, 1, seed=8)
    F = max(4, int(1e7 / (15 * n * k)))
    out = []
    for _ in range(plans):
        with W.Plan(cfg, W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32),
                    ch[11:13].astype(np.complex64), np.array([5.0, 15.0, 25.0], np.float32)) as plan:
            out += [tuple(plan.run(3, F)[..., [0, 2]].ravel().tolist()) for _ in range(reps)]
    return out
for n, k in ((256, 4), (512, 4), (1024, 6), (1024, 2), (1024, 6)):
    r = runs(n, k, 4)
    vals, cnt = np.unique(np.array(r), axis=0, return_counts=True)
    major = vals[np.argmax(cnt)]
    odd = [(i, (np.array(x) - major).tolist()) for i, x in enumerate(r) if not np.array_equal(x, major)]
    print("N=%d k=%d: %d launches, %d differ from the majority %s" % (n, k, len(r), len(odd), odd))
