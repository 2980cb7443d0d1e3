#!/usr/bin/env python3
"""Developer tool (GPU box): throughput of the main_channel_mask.m variant (row f1) -- half-band
allocation alone and with the spectral Tx mask -- next to the plain kernel, wtx N=256 16-QAM.

    python tools/bench_channel_mask.py [frames_per_cell]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import wofdm_amd as W  # noqa: E402
from wofdm_amd import channel_mask as CM  # noqa: E402

ch = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "channels_vehA.npz"))["h"]


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    for system, cp in (("wtx", 32), ("CPW", 22)):
        st = W.make_structure(system, 256, cp)
        snr = np.arange(-5.0, 51.0, 5.0).astype(np.float32)
        cfg = W.make_cfg(st, 4, 16, 21, 1, snr.size, 1, seed=3)
        with W.Plan(cfg, W.tx_rc_window(st), W.rx_rc_window(st), ch[:1].astype(np.complex64), snr) as plan:
            counts = plan.new_counts()
            for name, alloc, mask in (("plain", None, None), ("half-band", CM.half_band_allocation(256), None),
                                      ("half-band + mask", CM.half_band_allocation(256), CM.tx_mask(st.sym_len))):
                plan.set_allocation(alloc)
                plan.set_tx_mask(mask)
                plan.launch(0, max(1, F // 10), counts)
                ms = min(plan.launch_timed((i + 1) * F, F, counts) for i in range(3))
                info = plan.info()
                syms = F * 16 * snr.size
                print("%-5s %-18s %8.2f ms  %.3e sym/s  waves/WG=%d WG/CU=%d LDS=%d"
                      % (system, name, ms, syms / ms * 1e3, info["waves_per_workgroup"],
                         info["workgroups_per_cu"], info["lds_bytes"]))


if __name__ == "__main__":
    main()
