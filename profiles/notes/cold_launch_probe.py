#!/usr/bin/env python3
"""Developer tool (GPU box): is the FIRST launch of a frame kernel in a fresh process (cold instruction cache,
cold TLB) bit-identical to its later launches?  One kernel per process:

    python tools/cold_launch_probe.py n_fft k inject var [system]

prints one line: frames (one per cell, one cell per workgroup) whose counters differ between launch 0 and
launch 3.  (How the split-MFMA-chain fault was found and how its fix is checked: DESIGN.md section 4.)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wofdm_amd as W
ch = np.load(os.path.join(ROOT, "tests", "golden", "channels_vehA.npz"))["h"]
n_fft, k, inject, var = (int(a) for a in sys.argv[1:5])
system = sys.argv[5] if len(sys.argv) > 5 else "WOLA"
S, n_ch = 16, 8
st = W.make_structure(system, n_fft, 32 if n_fft >= 256 else 16)
w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
cfg0 = W.make_cfg(st, k, S, 21, n_ch, 4, 1, seed=8)
with W.Plan(cfg0, w_tx, w_rx, ch[:n_ch].astype(np.complex64), np.linspace(8, 36, 4).astype(np.float32)) as pl0:
    grid = pl0.info()["workgroups"]       # (no launch: plan creation only)
n_snr = max(1, min(64, grid // n_ch))
cells = n_snr * n_ch
snrs = np.linspace(8, 36, n_snr).astype(np.float32)
cfg = W.make_cfg(st, k, S, 21, n_ch, n_snr, 1, seed=8)
rs = np.random.RandomState(5)
if inject:
    nl = cfg.noise_len if hasattr(cfg, "noise_len") else None
with W.Plan(cfg, w_tx, w_rx, ch[11:11 + n_ch].astype(np.complex64), snrs) as plan:
    if var >= 1:
        a = rs.rand(n_fft) < 0.6
        a[0] = True
        plan.set_allocation(a)
    if var >= 2:
        from wofdm_amd import channel_mask as CM
        plan.set_tx_mask(CM.tx_mask(st.sym_len, roll_off=10))
    if inject:
        nl = plan.noise_len
        dl = torch.from_numpy(rs.randint(0, 1 << k, (cells, 1, S, n_fft)).astype(np.uint8)).cuda()
        dn = torch.from_numpy((rs.randn(cells, 1, nl, 2) * np.sqrt(0.5)).astype(np.float32)).cuda()
    out = []
    for rep in range(4):
        if inject:
            counts = plan.new_counts()
            plan.launch_injected(1, dl, dn, counts)
            torch.cuda.synchronize()
            plan.status()
            out.append(counts.cpu().numpy().view(np.uint64).astype(np.int64).reshape(cells, 4))
        else:
            out.append(plan.run(3, 1).astype(np.int64).reshape(cells, 4))
    d0 = int((out[0] != out[3]).any(axis=1).sum())
    d1 = int((out[1] != out[3]).any(axis=1).sum())
    d2 = int((out[2] != out[3]).any(axis=1).sum())
    print("N=%4d k=%d inject=%d var=%d %-5s kernel %s cells %3d: frames differing from launch 3: launch 0 %3d, launch 1 %3d, launch 2 %3d%s" % (
        n_fft, k, inject, var, system, plan.kernel_id(), cells, d0, d1, d2, "   <--" if d0 or d1 or d2 else ""))
