import os, sys
import numpy as np
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0,R)
import wofdm_amd as W
from oracle import oracle as O
ch=np.load(os.path.join(R,"tests/golden/channels_vehA.npz"))["h"]
system,n_fft,cp,k,matlab=("WOLA",512,32,4,1)
S, seed, frame = 16, 11, 123456789012
st = W.make_structure(system, n_fft, cp)
rs = np.random.RandomState(n_fft + cp + k)
xt = np.concatenate(([1.03], np.sort(rs.uniform(.05, .95, st.tail_tx))[::-1]))
xr = np.concatenate(([0.97], np.sort(rs.uniform(.05, .45, st.tail_rx // 2))[::-1]))
w_tx = W.expand_tx_window(st, xt).astype(np.float32); w_rx = W.expand_rx_window(st, xr).astype(np.float32)
h = ch[4:7].astype(np.complex64); snrs = np.array([8.0, 22.0], dtype=np.float32)
cfg = W.make_cfg(st, k, S, 21, 3, 2, 1, noise_before_truncate=True, seed=seed)
osys = O.make_sys(st.n_fft, k, S, st.cp, st.cs, st.tail_tx, st.tail_rx, st.prefix_rm, st.circ_shift, 21, 1)
cell=4
lab = O.gen_labels(osys, seed, cell, frame); noise = O.gen_noise(osys, seed, cell, frame)
oc, od = O.frame(osys, w_tx.astype(np.float64), w_rx.astype(np.float64), h[1].astype(np.complex128), float(snrs[1]), lab, noise, dump=True)
with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
    for rep in range(3):
        gc, gd = plan.dump_frame(cell, frame)
        d = np.abs(gd["rx"]-od["rx"]); bad=np.flatnonzero(d > 1e-4*np.abs(od["rx"]).max())
        print("rep",rep,"bad rx idx", bad[:20], len(bad), "B",st.stride, "bad%B", (bad%st.stride)[:20])
        d2 = np.abs(gd["conv"][:od["conv"].size]-od["conv"]); print(" conv bad", np.flatnonzero(d2 > 1e-4*np.abs(od["conv"]).max())[:10])
        d3 = np.abs(gd["unit_noise"]-noise); print(" noise bad", np.flatnonzero(d3 > 1e-4)[:10])
