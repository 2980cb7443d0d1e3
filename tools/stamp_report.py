#!/usr/bin/env python3
"""Developer tool (GPU box): where a wave's cycles go, from the WOFDM_STAMP diagnostic build.

    WOFDM_LIB=$PWD/ab/lib_stamp.so python tools/stamp_report.py [N] [K]

Reads the per-wave cycle totals the stamped kernel writes behind the counters (phases A-D, the
three barrier waits, loop control).  Shares, not absolute times: the stamps fence the schedule.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import wofdm_amd as W  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ch = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "channels_vehA.npz"))["h"]
st = W.make_structure("wtx" if n <= 256 else "WOLA", n, 32)
snr = np.arange(-5.0, 51.0, 5.0).astype(np.float32)
cfg = W.make_cfg(st, k, 16, 21, 1, 12, 1, seed=2)
frames = 62500 * 256 // n
MASK = os.environ.get("STAMP_MASK") == "1"       # row f1: half-band allocation + Tx mask (build with -DWOFDM_STAMP_MASK)
if MASK:
    frames //= 4
with W.Plan(cfg, W.tx_rc_window(st), W.rx_rc_window(st), ch[:1].astype(np.complex64), snr) as plan:
    if MASK:
        from wofdm_amd import channel_mask as CM
        plan.set_allocation(CM.half_band_allocation(n))
        plan.set_tx_mask(CM.tx_mask(st.sym_len))
    info = plan.info()
    grid, waves = info["workgroups"], info["waves_per_workgroup"]
    buf = torch.zeros(4 * 12 + grid * 16 * 16, dtype=torch.int64, device="cuda")
    ms = plan.launch_timed(0, frames, buf)
    torch.cuda.synchronize()
    st_ = buf[48:].cpu().numpy().reshape(grid, 16, 16)[:, :waves, :].astype(np.float64)
names = ["A4 Tx write (+ mask)", "wait barrier 1", "B3 trailing tile, power sums", "wait barrier 2",
         "C4 pilot estimate", "wait barrier 3", "D equalise/demap", "loop control",
         "A1 Philox data bits", "A2 labels, QAM", "A3 IFFT", "B1 overlap-add, operands", "B2 tiles: FIR + noise",
         "C1 gain, r = c + g n", "C2 Rx window loads", "C3 FFT"]
if MASK:
    names[0] = "A7 mask: write-back, hand-over, spill"
    names[4] = "C  (whole phase)"
    names[13:16] = ["A4 Tx write", "A5 mask: load, forward transform", "A6 mask: spectrum, inverse transform"]
    print("kernel", info.get("kernel_id"), "LDS", info["lds_bytes"])
order = [7, 8, 9, 10, 13, 14, 15, 0, 1, 11, 12, 2, 3, 4, 5, 6] if MASK else [7, 8, 9, 10, 0, 1, 11, 12, 2, 3, 13, 14, 15, 4, 5, 6]
tot = st_.sum(axis=2)
print("N=%d k=%d: %.2f ms, %d workgroups x %d waves; mean cycles per wave %.3e" % (n, k, ms, grid, waves, tot.mean()))
for i in order:
    nm = names[i]
    sh = st_[:, :, i] / tot
    per_frame = st_[:, :, i].mean() / (frames * 12.0 / grid)          # cycles per frame of one wave
    print("  %-22s all waves %5.1f %%   wave 0 %5.1f %%   last wave %5.1f %%   %7.0f cycles/frame"
          % (nm, 100 * sh.mean(), 100 * sh[:, 0].mean(), 100 * sh[:, -1].mean(), per_frame))
