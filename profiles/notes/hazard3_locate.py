#!/usr/bin/env python3
"""Developer tool (GPU box), hazard 3 of DESIGN.md section 4: WHERE is a wrong first launch wrong?

    WOFDM_LIB=ab/lib_ffu_dump.so python tools/hazard3_locate.py [out.npz]

Needs a -DWOFDM_LDSDUMP build of the injected N = 1024 / 64-QAM kernel (ab/build_probe.sh): every workgroup
copies its whole LDS behind the counters after its one frame.  Launch 0 (fresh process) is compared with launch 3,
workgroup by workgroup, region by region; for the symbol rows (which hold the forward transform's second exchange at
that point) the transform is finished on the host and the difference taken back to the time domain, so that the
received samples that differ are named."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wofdm_amd as W
ch = np.load(os.path.join(ROOT, "tests", "golden", "channels_vehA.npz"))["h"]
n_fft, k, S, n_ch = 1024, 6, 16, 8
st = W.make_structure("WOLA", n_fft, 32)
w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
cfg0 = W.make_cfg(st, k, S, 21, n_ch, 4, 1, seed=8)
with W.Plan(cfg0, w_tx, w_rx, ch[:n_ch].astype(np.complex64), np.linspace(8, 36, 4).astype(np.float32)) as pl0:
    info = pl0.info()
grid, lds = info["workgroups"], info["lds_bytes"]
n_snr = max(1, min(64, grid // n_ch))
cells = n_snr * n_ch
snrs = np.linspace(8, 36, n_snr).astype(np.float32)
cfg = W.make_cfg(st, k, S, 21, n_ch, n_snr, 1, seed=8)
rs = np.random.RandomState(5)
B, beta, gam, delta = st.stride, st.tail_tx, st.prefix_rm, st.tail_rx
OFF_G, OFF_SUMS, OFF_FLAGS, OFF_FBUF = 8 * n_fft, 16 * n_fft, 16 * n_fft + 256, 16 * n_fft + 256 + 128 + 4 * (n_fft + 64) + 4 * (n_fft + 64) + 512
with W.Plan(cfg, w_tx, w_rx, ch[11:11 + n_ch].astype(np.complex64), snrs) as plan:
    nl = plan.noise_len
    dl = torch.from_numpy(rs.randint(0, 1 << k, (cells, 1, S, n_fft)).astype(np.uint8)).cuda()
    dn = torch.from_numpy((rs.randn(cells, 1, nl, 2) * np.sqrt(0.5)).astype(np.float32)).cuda()
    words = lds // 4
    outs, dumps = [], []
    for rep in range(4):
        buf = torch.zeros(cells * 4 + (cells * words + 1) // 2, dtype=torch.int64, device="cuda:0")
        plan.launch_injected(1, dl, dn, buf)
        torch.cuda.synchronize()
        plan.status()
        h = buf.cpu().numpy()
        outs.append(h[:cells * 4].reshape(cells, 4).copy())
        if rep in (0, 3):
            dumps.append(h[cells * 4:].view(np.uint32)[:cells * words].reshape(cells, words).copy())
    bad = np.nonzero((outs[0] != outs[3]).any(axis=1))[0]
    print("lds bytes %d, cells %d; frames whose counters differ, launch 0 vs 3: %d %s" % (lds, cells, len(bad), list(bad)))
    print("launch 1 vs 3: %d, launch 2 vs 3: %d" % ((outs[1] != outs[3]).any(axis=1).sum(), (outs[2] != outs[3]).any(axis=1).sum()))
    d0, d3 = dumps
    lds_bad = np.nonzero((d0 != d3).any(axis=1))[0]
    print("workgroups whose LDS differs: %d" % len(lds_bad))
    good = [w for w in lds_bad if w not in set(bad.tolist())]
    if good:
        offs = sorted(set(int(o) for w in good for o in (np.nonzero(d0[w] != d3[w])[0] * 4)))
        print("   byte offsets that differ in workgroups with EQUAL counters (n=%d): %d distinct, %s ... %s" % (len(good), len(offs), offs[:12], offs[-12:]))
    # stage 3 of fft_big on the host: Y[l + 64 u] = sum_t fb[l + 64 t] e^{-2 pi i t l / 1024} e^{-2 pi i t u / 16}
    l = np.arange(64)[None, :, None]; t = np.arange(16)[:, None, None]; u = np.arange(16)[None, None, :]
    def finish(row):                        # row: complex64[1024] = in[lane + 64 t]
        x = row.reshape(16, 64)[:, :, None] * np.exp(-2j * np.pi * t * l / 1024.0)
        return (x * np.exp(-2j * np.pi * t * u / 16.0)).sum(axis=0).T.reshape(-1)     # index l + 64 u
    for wg in bad:
        a, b = d0[wg], d3[wg]
        dw = np.nonzero(a != b)[0] * 4
        regs = [("tw", 0, OFF_G), ("G", OFF_G, OFF_SUMS), ("sums", OFF_SUMS, OFF_FLAGS), ("flags", OFF_FLAGS, OFF_FLAGS + 128),
                ("wtx/wrx/lut", OFF_FLAGS + 128, OFF_FBUF), ("frame", OFF_FBUF, OFF_FBUF + 32 + 8 * B * S), ("virtual row + tails", OFF_FBUF + 32 + 8 * B * S, lds)]
        print("== workgroup %d (counters %s vs %s): %d differing words" % (wg, outs[0][wg].tolist(), outs[3][wg].tolist(), len(dw)))
        for name, lo, hi in regs:
            n = int(((dw >= lo) & (dw < hi)).sum())
            if n: print("   %-20s %6d words differ" % (name, n))
        s0 = a[OFF_SUMS // 4:OFF_SUMS // 4 + 64].view(np.float32); s3 = b[OFF_SUMS // 4:OFF_SUMS // 4 + 64].view(np.float32)
        for par in range(2):
            for nm, o in (("Ps", 0), ("Pn", 16)):
                x0, x3 = s0[32 * par + o:32 * par + o + 16], s3[32 * par + o:32 * par + o + 16]
                w = np.nonzero(x0 != x3)[0]
                if len(w): print("   sums[%d] %s differs for waves %s: rel %s" % (par, nm, list(w), ["%.2e" % ((x0[i] - x3[i]) / x3[i]) for i in w]))
        print("   flags launch0 %s" % a[OFF_FLAGS // 4:OFF_FLAGS // 4 + 24].tolist())
        for s in range(S):
            lo = (OFF_FBUF + 32 + 8 * B * s) // 4
            r0 = a[lo:lo + 2 * B].view(np.complex64); r3 = b[lo:lo + 2 * B].view(np.complex64)
            if (a[lo:lo + 2 * B] != b[lo:lo + 2 * B]).any():
                y0, y3 = finish(r0[:1024]), finish(r3[:1024])
                zd = np.fft.ifft(y0 - y3)
                ref = np.abs(np.fft.ifft(y3)).max()
                big = np.nonzero(np.abs(zd) > 1e-4 * ref)[0]
                tailw = np.nonzero(a[lo + 2048:lo + 2 * B] != b[lo + 2048:lo + 2 * B])[0] // 2 + 1024
                print("   symbol %2d: windowed rx samples t (block sample = %d + t) that differ: n=%d first..last %s  max rel %.3g; raw rx samples >= 1024 that differ: %s"
                      % (s, gam, len(big), (list(big[:40]), int(big[-1])) if len(big) else None, float(np.abs(zd).max() / ref), sorted(set(tailw.tolist()))))
    if len(sys.argv) > 1 and len(bad):
        np.savez_compressed(sys.argv[1], wgs=bad, launch0=d0[bad], launch3=d3[bad], counts0=outs[0], counts3=outs[3])
