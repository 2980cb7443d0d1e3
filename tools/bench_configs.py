#!/usr/bin/env python3
"""Developer tool (GPU box): throughput of other BASELINE configs (not the bench.py line).

    python tools/bench_configs.py [frames_per_cell]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import wofdm_amd as W  # noqa: E402

ch = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "channels_vehA.npz"))["h"]


def run(system, n_fft, k, n_ch, n_snr, frames, cp=32, S=16, reps=3):
    st = W.make_structure(system, n_fft, cp)
    cfg = W.make_cfg(st, k, S, 21, n_ch, n_snr, 1, seed=4)
    snr = (-20 + 3.0 * np.arange(n_snr)).astype(np.float32)
    with W.Plan(cfg, W.tx_rc_window(st), W.rx_rc_window(st), ch[:n_ch].astype(np.complex64), snr) as plan:
        if os.environ.get("WOFDM_FIR_VALU") == "1":      # (tool switch: the round-1 kernels, FIR on the VALU)
            plan.set_option("fir_valu", 1)
        for kv in filter(None, os.environ.get("WOFDM_OPTS", "").split(",")):   # (tool switch: plan options, "dft_valu=1,...")
            plan.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        counts = plan.new_counts()
        plan.launch(0, max(1, frames // 4), counts)
        torch.cuda.synchronize()
        ms = [plan.launch_timed((i + 1) * frames, frames, counts) for i in range(reps)]
        info = plan.info()
        kid = plan.kernel_id()
    syms = frames * S * n_ch * n_snr
    best = min(ms)
    print("%-6s N=%-4d k=%d cells=%-5d frames/cell=%-6d %8.2f ms  %.3e sym/s  waves/WG=%d WG/CU=%d LDS=%d layout=%d"
          % (system, n_fft, k, n_ch * n_snr, frames, best, syms / best * 1e3,
             info["waves_per_workgroup"], info["workgroups_per_cu"], info["lds_bytes"], kid[0]))


if __name__ == "__main__":
    f = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2000
    run("wtx", 256, 4, 1, 12, 62500)
    if "--c2-only" in sys.argv:
        sys.exit(0)
    for system in W.SYSTEMS:
        run(system, 256, 4, 1, 1, 62500)
    run("WOLA", 64, 2, 4, 5, f * 8, cp=16)            # (C1 has CP 16)
    run("WOLA", 128, 4, 4, 5, f * 4)
    run("WOLA", 512, 4, 10, 20, f // 2)
    run("WOLA", 1024, 6, 100, 20, max(1, f // 20))
    run("WOLA", 1024, 2, 10, 20, f // 2)
    if "--c4-full" in sys.argv:
        # BASELINE config 4 at full size on ONE GPU: 100 channels x 20 SNR points x 62 500 frames = 2e9 symbols
        run("WOLA", 1024, 6, 100, 20, 62500, reps=1)
