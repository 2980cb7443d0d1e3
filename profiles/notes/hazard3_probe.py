#!/usr/bin/env python3
"""Developer tool (GPU box), hazard 3: the PRODUCTION kernel's Tx rows (end of phase A) and FIR outputs (end of the tile
loop) of one frame per workgroup, from two -DWOFDM_PROBE builds of the same arithmetic (ab/build_probe.sh), compared
word by word: which phase is hit when an MFMA chain is not issued back to back?

    python tools/hazard3_probe.py ab/lib_p_head.so ab/lib_p_gap.so          (injected N = 1024 / 64-QAM frames)"""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORDS = 16 * 2 * 1088 + 16 * 9 * 64 * 4
if sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import torch
    import wofdm_amd as W
    out, inject = sys.argv[2], int(sys.argv[3])
    ch = np.load(os.path.join(ROOT, "tests", "golden", "channels_vehA.npz"))["h"]
    n_fft, k, S, n_ch, n_snr = 1024, 6, 16, 8, 32
    cells = n_ch * n_snr
    st = W.make_structure("WOLA", n_fft, 32)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    snrs = np.linspace(8, 36, n_snr).astype(np.float32)
    cfg = W.make_cfg(st, k, S, 21, n_ch, n_snr, 1, seed=8)
    rs = np.random.RandomState(5)
    with W.Plan(cfg, w_tx, w_rx, ch[11:11 + n_ch].astype(np.complex64), snrs) as plan:
        assert plan.info()["workgroups"] == cells
        dl = torch.from_numpy(rs.randint(0, 1 << k, (cells, 1, S, n_fft)).astype(np.uint8)).cuda()
        dn = torch.from_numpy((rs.randn(cells, 1, plan.noise_len, 2) * np.sqrt(0.5)).astype(np.float32)).cuda()
        res = []
        for rep in range(3):
            buf = torch.zeros(cells * 4 + cells * WORDS // 2, dtype=torch.int64, device="cuda:0")
            if inject:
                plan.launch_injected(1, dl, dn, buf)
            else:
                plan.launch(3, 1, buf)
            torch.cuda.synchronize()
            plan.status()
            h = buf.cpu().numpy()
            res.append((h[:cells * 4].reshape(cells, 4).copy(), h[cells * 4:].view(np.uint32).reshape(cells, WORDS).copy()))
        np.save(out + ".counts.npy", np.stack([r[0] for r in res]))
        np.save(out + ".probe.npy", np.stack([r[1] for r in res]))
    sys.exit(0)
libs = sys.argv[1:3]
inject = int(sys.argv[3]) if len(sys.argv) > 3 else 1
data = []
for i, lib in enumerate(libs):
    out = "/tmp/h3probe_%d" % i
    subprocess.run([sys.executable, os.path.abspath(__file__), "--one", out, str(inject)], check=True,
                   env=dict(os.environ, WOFDM_LIB=os.path.join(ROOT, lib)))
    data.append((np.load(out + ".counts.npy"), np.load(out + ".probe.npy")))
B, S, NT = 1056, 16, 9
(c0, p0), (c1, p1) = data
print("reference %s: launches 0, 1 vs 2: counters differ in %s frames, probe words in %s" % (
    libs[0], [int((c0[i] != c0[2]).any(axis=1).sum()) for i in range(2)], [int((p0[i] != p0[2]).sum()) for i in range(2)]))
ref_c, ref_p = c0[2], p0[2]
for rep in range(3):
    c, p = c1[rep], p1[rep]
    badf = np.nonzero((c != ref_c).any(axis=1))[0]
    tx = p[:, :S * 2 * B].reshape(-1, S, 2 * B); rtx = ref_p[:, :S * 2 * B].reshape(-1, S, 2 * B)
    cv = p[:, S * 2 * 1088 - S * 2 * (1088 - B):][:, :0]   # (unused)
    conv = p[:, S * 2 * B:S * 2 * B + S * NT * 256].reshape(-1, S, NT, 64, 4); rconv = ref_p[:, S * 2 * B:S * 2 * B + S * NT * 256].reshape(-1, S, NT, 64, 4)
    txbad = (tx != rtx); cvbad = (conv != rconv)
    print("%s launch %d: frames with wrong counters %d; frames with wrong Tx words %d (words %d); frames with wrong FIR outputs %d (values %d)" % (
        libs[1], rep, len(badf), int(txbad.any(axis=(1, 2)).sum()), int(txbad.sum()), int(cvbad.any(axis=(1, 2, 3, 4)).sum()), int(cvbad.sum())))
    both = set(np.nonzero(txbad.any(axis=(1, 2)))[0].tolist()) | set(np.nonzero(cvbad.any(axis=(1, 2, 3, 4)))[0].tolist())
    print("   frames with wrong counters but clean probe: %d; frames with a wrong probe but right counters: %d" % (
        len(set(badf.tolist()) - both), len(both - set(badf.tolist()))))
    if rep == 2:
        # where in the Tx rows: symbol, plane, sample index; lanes of the store instruction = sample % 64
        w, s, i = np.nonzero(txbad)
        sym_hist = np.bincount(s, minlength=S)
        print("   wrong Tx words by symbol:", sym_hist.tolist())
        samp = i % B
        print("   wrong Tx words by plane (H, L):", [int((i < B).sum()), int((i >= B).sum())])
        print("   wrong Tx words by (sample - 32) %% 64 // 16 (lane row of the store):", np.bincount(((samp - 32) % 64) // 16, minlength=4).tolist())
        shown = 0
        for f in sorted(set(w.tolist()))[:6]:
            m = w == f
            runs = sorted(set((int(a), int(b) // B, (int(b) % B) // 16 * 16) for a, b in zip(s[m], i[m])))
            print("   frame %3d: (symbol, plane, first sample of the 16-sample chunk): %s" % (f, runs[:24]))
        w2, s2, g2, l2, e2 = np.nonzero(cvbad)
        print("   wrong FIR outputs by symbol:", np.bincount(s2, minlength=S).tolist(), " by tile:", np.bincount(g2, minlength=NT).tolist(),
              " by lane row:", np.bincount(l2 // 16, minlength=4).tolist())
