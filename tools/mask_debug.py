"""Developer tool (GPU box): one fuzz geometry of the Tx-mask variants, layout 9 against layout 1 and the oracle, stage by stage."""
import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import wofdm_amd as W
from wofdm_amd import channel_mask as CM
from oracle import oracle as O
channels = np.load(os.path.join(R, "tests", "golden", "channels_vehA.npz"))["h"]
system, n_fft, k, S, btx, brx, cp, taps, opts, seed = "WOLA", 64, 2, 8, 2, 10, 44, 5, 2, 730075073
if len(sys.argv) > 1:
    system, n_fft, k, S, btx, brx, cp, taps, opts, seed = sys.argv[1], *[int(x) for x in sys.argv[2:11]]
rs = np.random.RandomState(seed)
st = W.make_structure(system, n_fft, cp, btx, brx)
print(st, "stride", st.stride, "sym_len", st.sym_len)
xt = np.r_[1.0, np.sort(rs.uniform(.05, .95, btx))[::-1]] if btx else np.ones(1)
xr = np.r_[1.0, np.sort(rs.uniform(.05, .45, brx // 2))[::-1]] if brx else np.ones(1)
w_tx = (W.expand_tx_window(st, xt) if btx else np.ones(st.sym_len)).astype(np.float32)
w_rx = (W.expand_rx_window(st, xr) if brx else np.ones(st.rx_win_len)).astype(np.float32)
h = channels[rs.randint(90):][:2, :taps].astype(np.complex64); h[:, 0] += 0.5
snrs = np.array([rs.uniform(0, 12), rs.uniform(18, 35)], dtype=np.float32)
matlab = bool(rs.randint(2)); F, off = 3, int(rs.randint(1 << 20))
active = (rs.rand(n_fft) < 0.55) if opts else None
if active is not None: active[rs.randint(n_fft)] = True
mask = CM.tx_mask(st.sym_len, roll_off=int(rs.choice([4, 10, 20]))) if opts == 2 else None
cfg = W.make_cfg(st, k, S, taps, 2, 2, 1, noise_before_truncate=matlab, seed=seed)
osys = O.make_sys(n_fft, k, S, st.cp, st.cs, st.tail_tx, st.tail_rx, st.prefix_rm, st.circ_shift, taps, 1 if matlab else 0, active=active,
                  tx_mask=None if mask is None else mask.astype(np.float32).astype(np.float64))
want = O.run(osys, w_tx.astype(np.float64), w_rx.astype(np.float64), h.astype(np.complex128), snrs.astype(np.float64), seed, off, F)
cell, frame = 1, off
lab = O.gen_labels(osys, seed, cell, frame); noise = O.gen_noise(osys, seed, cell, frame)
oc, od = O.frame(osys, w_tx.astype(np.float64), w_rx.astype(np.float64), h[cell % 2].astype(np.complex128), float(snrs[cell // 2]), lab, noise, dump=True)
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
for fv in (0, 1):
    with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
        plan.set_option("fir_valu", fv)
        plan.set_allocation(active); plan.set_tx_mask(mask)
        got = plan.run(off, F)
        gc, gd = plan.dump_frame(cell, frame)
        print("fir_valu", fv, "kernel", plan.kernel_id(), "counts", got[..., 0].ravel(), "want", want[..., 0].ravel())
    print("   stages:", " ".join("%s %.2e" % (n, rel(gd[n][:od[n].size] if gd[n].ndim == 1 else gd[n], od[n])) for n in ("X", "tx", "conv", "rx", "Y")), gc, oc)
    d = np.abs(gd["tx"] - od["tx"]); bad = np.where(d > 1e-4 * np.abs(od["tx"]).max())[0]
    print("   tx bad positions:", bad[:20], len(bad))
    d = np.abs(gd["conv"][:od["conv"].size] - od["conv"]); bad = np.where(d > 1e-4 * np.abs(od["conv"]).max())[0]
    print("   conv bad positions:", bad[:20], len(bad))
print("=== repeated production runs (layout 9), per-cell bit errors; then single frames")
with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
    plan.set_allocation(active); plan.set_tx_mask(mask)
    for rep in range(4):
        print(rep, plan.run(off, F)[..., 0].ravel())
    for f in range(3):
        print("frame", f, plan.run(off + f, 1)[..., 0].ravel(), "oracle", O.run(osys, w_tx.astype(np.float64), w_rx.astype(np.float64), h.astype(np.complex128), snrs.astype(np.float64), seed, off + f, 1)[..., 0].ravel())
if os.environ.get("MASK_DEBUG_TX") == "1":
    # the instrumented kernel's on-air frame against the oracle's, symbol by symbol (real and imaginary parts apart)
    with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
        plan.set_allocation(active); plan.set_tx_mask(mask)
        gc, gd = plan.dump_frame(cell, frame)
        d = gd["tx"] - od["tx"]; sc = np.abs(od["tx"]).max()
        Bst = st.stride
        for sy in range(S):
            seg = d[sy * Bst:(sy + 1) * Bst]
            print("symbol %2d: max |re err| %.2e  max |im err| %.2e   first bad %s" % (sy, np.abs(seg.real).max() / sc, np.abs(seg.imag).max() / sc,
                  np.where(np.abs(seg) > 1e-4 * sc)[0][:6]))
        print("tail:", np.abs(d[S * Bst:]).max() / sc)
