#!/usr/bin/env python3
"""Developer tool and GPU test helper (GPU box): in a FRESH process, launch every production instantiation of ONE (n_fft, k)
translation unit of the frame kernel for the first time -- cold instruction cache and translation for that kernel's code --
and compare that launch with the kernel's third one, frame by frame (one frame per cell, one cell per workgroup).

    python tools/first_launch_unit.py n_fft k

Prints one line per instantiation and a last line "TOTAL <kernels> <differing frames>".  (Hazard 3 of DESIGN.md section 4: an
instruction fetch that falls between two MFMAs of the FIR's chain -- then a page boundary inside a chain, on a first launch --
corrupted packed op_sel arithmetic of the other waves of the SIMD; tests/test_code_layout.py is the static guard.)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wofdm_amd as W
from wofdm_amd import channel_mask as CM
ch = np.load(os.path.join(ROOT, "tests", "golden", "channels_vehA.npz"))["h"]
n_fft, k = int(sys.argv[1]), int(sys.argv[2])
# (system, cp, S, plan options, variant): every layout the plan can pick at this DFT length
if n_fft == 256:
    geos = [("wtx", 32, 16, {}), ("wtx", 48, 16, {}), ("wtx", 32, 16, {"fir_valu": 1}), ("CPW", 32, 16, {"fir_valu": 1}),
            ("wtx", 32, 16, {"fir_valu": 1, "max_spw": 2}), ("wtx", 32, 9, {})]
elif n_fft >= 512:
    geos = [("WOLA", 32, 16, {}), ("WOLA", 32, 16, {"fir_valu": 1})]
else:
    # (layouts 13 / 14, 16 -- a partly filled wave, and two or three waves at a long stride --, and the VALU layouts 2 and 1)
    geos = [("wtx", 16, 16, {}), ("wtx", 16, 9, {}), ("wrx", 56, 16, {}), ("wtx", 16, 16, {"dft_valu": 1}), ("wtx", 16, 9, {"dft_valu": 1})]
total, bad, seen = 0, 0, set()
for system, cp, S, opts in geos:
    for var in (0, 1, 2, 3):
        if var >= 2 and (n_fft > 512 or (var == 3 and n_fft > 256)):
            continue
        for inject in (0, 1):
            st = W.make_structure(system, n_fft, cp)
            w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
            n_ch = 8
            cfg0 = W.make_cfg(st, k, S, 21, n_ch, 4, 1, seed=8)
            with W.Plan(cfg0, w_tx, w_rx, ch[:n_ch].astype(np.complex64), np.linspace(8, 36, 4).astype(np.float32)) as pl0:
                for key, val in opts.items():
                    pl0.set_option(key, val)
                grid = pl0.info()["workgroups"]          # (no launch: plan creation only)
            n_snr = max(1, min(64, grid // n_ch))
            cells = n_snr * n_ch
            cfg = W.make_cfg(st, k, S, 21, n_ch, n_snr, 1, seed=8)
            rs = np.random.RandomState(5)
            with W.Plan(cfg, w_tx, w_rx, ch[11:11 + n_ch].astype(np.complex64), np.linspace(8, 36, n_snr).astype(np.float32)) as plan:
                for key, val in opts.items():
                    plan.set_option(key, val)
                if var == 2:
                    plan.set_option("txmask_direct", 1)
                if var >= 1:
                    a = rs.rand(n_fft) < 0.6
                    a[0] = True
                    plan.set_allocation(a)
                if var >= 2:
                    plan.set_tx_mask(CM.tx_mask(st.sym_len, roll_off=10))
                kid = plan.kernel_id() + (inject,)
                if kid in seen or kid[1] != var:
                    continue
                seen.add(kid)
                if inject:
                    dl = torch.from_numpy(rs.randint(0, 1 << k, (cells, 1, S, n_fft)).astype(np.uint8)).cuda()
                    dn = torch.from_numpy((rs.randn(cells, 1, plan.noise_len, 2) * np.sqrt(0.5)).astype(np.float32)).cuda()
                out = []
                for rep in range(3):
                    if inject:
                        counts = plan.new_counts()
                        plan.launch_injected(1, dl, dn, counts)
                        torch.cuda.synchronize()
                        plan.status()
                        out.append(counts.cpu().numpy().reshape(cells, 4).copy())
                    else:
                        out.append(plan.run(3, 1).astype(np.int64).reshape(cells, 4))
                d0 = int((out[0] != out[2]).any(axis=1).sum())
                d1 = int((out[1] != out[2]).any(axis=1).sum())
                total += 1
                bad += d0 + d1
                print("N=%4d k=%d layout %d variant %d inject %d (%s cp %d S %d) cells %3d: frames differing from launch 2: launch 0 %3d, launch 1 %3d%s"
                      % (n_fft, k, kid[0], kid[1], inject, system, cp, S, cells, d0, d1, "   <--" if d0 or d1 else ""))
print("TOTAL %d %d" % (total, bad))
