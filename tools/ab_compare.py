#!/usr/bin/env python3
"""Developer tool (GPU box): two builds of libwofdm_hip.so must give bit-identical counters and
stage dumps.   python tools/ab_compare.py ab/lib_old.so ab/lib_new.so   (each run in a child
process: one HIP library per process)"""
import os
import pickle
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("WOLA", 512, 32, 4), ("wtx", 256, 32, 4), ("CPW", 64, 16, 2), ("WOLA", 1024, 32, 6), ("wrx", 128, 20, 6),
         ("CPwtx", 512, 20, 2)]


def child():
    sys.path.insert(0, ROOT)
    import wofdm_amd as W
    ch = np.load(os.path.join(ROOT, "tests/golden/channels_vehA.npz"))["h"]
    out = {}
    for system, n, cp, k in CASES:
        st = W.make_structure(system, n, cp)
        w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
        snrs = np.array([6.0, 21.0], np.float32)
        cfg = W.make_cfg(st, k, 16, 21, 3, 2, 1, seed=5)
        with W.Plan(cfg, w_tx, w_rx, ch[4:7].astype(np.complex64), snrs) as plan:
            out[(system, n, "counts")] = plan.run(7, 300)
            c, d = plan.dump_frame(4, 99)
            for key in ("tx", "conv", "rx", "Y", "Xhat", "labels_rx", "unit_noise"):
                out[(system, n, key)] = d[key]
    pickle.dump(out, sys.stdout.buffer)


if __name__ == "__main__":
    if len(sys.argv) == 2 and sys.argv[1] == "--child":
        child()
        sys.exit(0)
    res = []
    for lib in sys.argv[1:3]:
        env = dict(os.environ, WOFDM_LIB=os.path.abspath(lib))
        res.append(pickle.loads(subprocess.run([sys.executable, __file__, "--child"], env=env, check=True,
                                               stdout=subprocess.PIPE).stdout))
    a, b = res
    for key in a:
        same = np.array_equal(a[key], b[key])
        extra = ""
        if not same:
            diff = np.flatnonzero(np.asarray(a[key]).reshape(-1) != np.asarray(b[key]).reshape(-1))
            extra = " first diffs at %s (%d)" % (diff[:8], diff.size)
        print("%-28s %s%s" % (key, "identical" if same else "DIFFERENT", extra))
