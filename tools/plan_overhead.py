import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import wofdm_amd as W
ch = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/channels_vehA.npz"))["h"]
st = W.make_structure("WOLA", 256, 22)
snr = np.linspace(-20, 50, 30).astype(np.float32)
cfg = W.make_cfg(st, 4, 16, 21, 100, 30, 7, seed=1)
wtx = np.tile(W.tx_rc_window(st), (7, 1)).astype(np.float32); wrx = np.tile(W.rx_rc_window(st), (7, 1)).astype(np.float32)
h = ch[:100].astype(np.complex64)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for i in range(20):
        p = W.Plan(cfg, wtx, wrx, h, snr); p.close()
    t1 = time.perf_counter()
    print("create+destroy: %.3f ms" % ((t1 - t0) / 20 * 1e3))
p = W.Plan(cfg, wtx, wrx, h, snr)
c = p.new_counts()
t0 = time.perf_counter()
for i in range(20):
    p.launch(i, 1, c)
torch.cuda.synchronize()
print("launch of 1 frame/cell (21000 cells): %.3f ms each" % ((time.perf_counter() - t0) / 20 * 1e3))
t0 = time.perf_counter()
for i in range(20):
    x = c.cpu()
print("counts D2H: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
