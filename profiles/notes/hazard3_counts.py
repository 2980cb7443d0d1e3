#!/usr/bin/env python3
"""Developer tool (GPU box), hazard 3 of DESIGN.md section 4: the counters of four launches of one frame kernel in a
fresh process, one frame per workgroup, written to a file -- so that builds of the same arithmetic (WOFDM_LIB=...)
can be compared with each other frame by frame (injected inputs: bit-identical counters expected).

    WOFDM_LIB=ab/lib_x.so python tools/hazard3_counts.py out.npy [n_fft k inject]
    python tools/hazard3_counts.py --compare ref.npy a.npy b.npy ..."""
import os, sys
import numpy as np
if sys.argv[1] == "--compare":
    ref = np.load(sys.argv[2])
    print("%-28s launch 0..3 vs each other: %s" % (os.path.basename(sys.argv[2]), [int((ref[i] != ref[3]).any(axis=1).sum()) for i in range(3)]))
    for f in sys.argv[3:]:
        a = np.load(f)
        print("%-28s frames differing from the reference's launch 3: %s   (own launch 0..2 vs own 3: %s)" % (
            os.path.basename(f), [int((a[i] != ref[3]).any(axis=1).sum()) for i in range(4)],
            [int((a[i] != a[3]).any(axis=1).sum()) for i in range(3)]))
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wofdm_amd as W
out = sys.argv[1]
n_fft, k, inject = (int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (1024, 6, 1)
ch = np.load(os.path.join(ROOT, "tests", "golden", "channels_vehA.npz"))["h"]
S, n_ch = 16, 8
st = W.make_structure("WOLA", n_fft, 32 if n_fft >= 256 else 16)
w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
cfg0 = W.make_cfg(st, k, S, 21, n_ch, 4, 1, seed=8)
with W.Plan(cfg0, w_tx, w_rx, ch[:n_ch].astype(np.complex64), np.linspace(8, 36, 4).astype(np.float32)) as pl0:
    grid = pl0.info()["workgroups"]
n_snr = max(1, min(64, grid // n_ch))
cells = n_snr * n_ch
snrs = np.linspace(8, 36, n_snr).astype(np.float32)
cfg = W.make_cfg(st, k, S, 21, n_ch, n_snr, 1, seed=8)
rs = np.random.RandomState(5)
with W.Plan(cfg, w_tx, w_rx, ch[11:11 + n_ch].astype(np.complex64), snrs) as plan:
    res = []
    if inject:
        dl = torch.from_numpy(rs.randint(0, 1 << k, (cells, 1, S, n_fft)).astype(np.uint8)).cuda()
        dn = torch.from_numpy((rs.randn(cells, 1, plan.noise_len, 2) * np.sqrt(0.5)).astype(np.float32)).cuda()
    for rep in range(4):
        if inject:
            counts = plan.new_counts()
            plan.launch_injected(1, dl, dn, counts)
            torch.cuda.synchronize()
            plan.status()
            res.append(counts.cpu().numpy().reshape(cells, 4))
        else:
            res.append(plan.run(3, 1).astype(np.int64).reshape(cells, 4))
    res = np.stack(res)
    np.save(out, res)
    print("N=%d k=%d inject=%d kernel %s cells %d: launches 0..2 differing from launch 3 in %s frames" % (
        n_fft, k, inject, plan.kernel_id(), cells, [int((res[i] != res[3]).any(axis=1).sum()) for i in range(3)]))
