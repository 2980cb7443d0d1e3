#!/usr/bin/env python3
"""Developer tool (GPU box): throughput of the injected-randomness variant, the one that streams
its inputs from HBM (labels u8 + unit noise f32 pairs), with the achieved HBM GB/s.

    python tools/bench_inject.py [frames_per_cell] [cells]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import wofdm_amd as W  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 62500
cells = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ch = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "channels_vehA.npz"))["h"]
st = W.make_structure("wtx", 256, 32)
snr = np.linspace(0, 30, cells).astype(np.float32)
cfg = W.make_cfg(st, 4, 16, 21, 1, cells, 1, seed=2)
with W.Plan(cfg, W.tx_rc_window(st), W.rx_rc_window(st), ch[:1].astype(np.complex64), snr) as plan:
    nl = plan.noise_len
    labels = torch.randint(0, 16, (cells, frames, 16, 256), dtype=torch.uint8, device="cuda")
    noise = torch.randn((cells, frames, nl, 2), dtype=torch.float32, device="cuda")
    counts = plan.new_counts()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    plan.launch_injected(frames, labels, noise, counts)
    torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        ev0.record(); plan.launch_injected(frames, labels, noise, counts); ev1.record()
        torch.cuda.synchronize(); ms.append(ev0.elapsed_time(ev1))
    host = counts.cpu().numpy().view(np.uint64)
best = min(ms)
syms = frames * 16 * cells
nbytes = labels.numel() + noise.numel() * 4
print("inject: %d cells x %d frames, %.2f ms: %.3e symbols/s, %.1f B/symbol, %.2f TB/s from HBM (%.1f %% of 8 TB/s)"
      % (cells, frames, best, syms / best * 1e3, nbytes / syms, nbytes / best * 1e-9, nbytes / best * 1e-9 / 8 * 100))
print("BER", (host[0, :, 0, 0] / host[0, :, 0, 1]).round(5))
