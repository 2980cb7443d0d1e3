"""Developer tool (GPU box): localise the first-launch deviation.  After a few other kernels, an N = 1024 plan
with one frame per cell (4200 cells): first launch vs second launch, cell by cell."""
import os, sys, subprocess, numpy as np
os.environ["WOFDM_NO_WARMUP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wofdm_amd as W
ch = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "channels_vehA.npz"))["h"]
print(subprocess.run("rocm-smi --showuniqueid | grep 'Unique ID:'", shell=True, capture_output=True, text=True).stdout.strip())
def other(system, n, k, frames):
    st = W.make_structure(system, n, 32 if n >= 256 else 16)
    cfg = W.make_cfg(st, k, 16, 21, 2, 3, 1, seed=99)
    with W.Plan(cfg, W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32), ch[60:62].astype(np.complex64), np.array([4., 12., 20.], np.float32)) as plan:
        plan.run(10, frames)
for rep in range(4):
    for a in (("wtx", 256, 4, 6000), ("WOLA", 64, 2, 20000), ("CPW", 512, 4, 1500)):
        other(*a)
    st = W.make_structure("WOLA", 1024, 32)
    n_snr, n_ch = 42, 100
    cfg = W.make_cfg(st, 6, 16, 21, n_ch, n_snr, 1, seed=8)
    snrs = np.linspace(4, 30, n_snr).astype(np.float32)
    with W.Plan(cfg, W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32), ch[:n_ch].astype(np.complex64), snrs) as plan:
        info = plan.info()
        a = plan.run(3, 1).reshape(-1, 4).astype(np.int64)
        b = plan.run(3, 1).reshape(-1, 4).astype(np.int64)
        c = plan.run(3, 1).reshape(-1, 4).astype(np.int64)
    bad = np.nonzero((a != b).any(axis=1))[0]
    grid = info["workgroups"]
    q, r = divmod(a.shape[0], grid)
    def wg_of(item):
        return item // (q + 1) if item < r * (q + 1) else r + (item - r * (q + 1)) // q
    print("rep %d: %d cells, grid %d, items/WG %d(+1 for %d); b==c %s; first launch differs in %d cells" % (rep, a.shape[0], grid, q, r, np.array_equal(b, c), bad.size))
    for i in bad[:40]:
        w = wg_of(i)
        start = w * (q + 1) if w < r else r * (q + 1) + (w - r) * q
        print("   cell %5d  WG %3d item-in-WG %2d  d(bit,sym) = %d %d   (second launch %d %d)" % (i, w, i - start, a[i, 0] - b[i, 0], a[i, 2] - b[i, 2], b[i, 0], b[i, 2]))
