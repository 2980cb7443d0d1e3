#!/usr/bin/env python3
"""Developer tool (GPU box): hunt for holes in the wave-to-wave synchronisation of the frame kernel.

Needs the delay build (``bash tools/build_variant.sh delay "64 128 256 512 1024" "2 4 6" -- -DWOFDM_DELAY=1``).
For every delay point of the frame (DELAY_AT in wofdm_kernel.hip) and several sets of waves, the chosen
waves sleep there for delay_len x 4 us; the counters must not change.

    python tools/delay_probe.py [n_fft k [inject [system]]]        (PROBE_MASK=1: with allocation + Tx mask)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["WOFDM_LIB"] = os.path.join(ROOT, "ab", "lib_delay.so")
sys.path.insert(0, ROOT)
import torch
import wofdm_amd as W
from oracle import oracle as O
ch = np.load(os.path.join(ROOT, "tests", "golden", "channels_vehA.npz"))["h"]
n_fft = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
k = int(sys.argv[2]) if len(sys.argv) > 2 else 6
inject = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
system = sys.argv[4] if len(sys.argv) > 4 else "WOLA"
S = 16
st = W.make_structure(system, n_fft, 32 if n_fft >= 256 else 16)
w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
snrs = np.array([5.0, 15.0, 25.0], np.float32) + (k - 4) * 3.0
F = max(4, int(4e6 / ((S - 1) * n_fft * k)))
seed, off = 8, 3
cfg = W.make_cfg(st, k, S, 21, 2, 3, 1, seed=seed)
h = ch[11:13].astype(np.complex64)
if inject:
    osys = O.make_sys(n_fft, k, S, st.cp, st.cs, st.tail_tx, st.tail_rx, st.prefix_rm, st.circ_shift, 21, 1)
    nl = O.noise_len(osys)
    labels = np.zeros((6, F, S, n_fft), np.uint8)
    noise = np.zeros((6, F, nl), np.complex64)
    for cell in range(6):
        for f in range(F):
            labels[cell, f] = O.gen_labels(osys, seed, cell, off + f)
            noise[cell, f] = O.gen_noise(osys, seed, cell, off + f)
    dl = torch.from_numpy(labels).cuda()
    dn = torch.from_numpy(noise.view(np.float32).reshape(6, F, nl, 2)).cuda()
with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
    if os.environ.get("PROBE_MASK") == "1":              # the Tx-mask variants (layout 9 / 1): half-band allocation + spectral mask
        from wofdm_amd import channel_mask as CM
        plan.set_allocation(CM.half_band_allocation(n_fft))
        plan.set_tx_mask(CM.tx_mask(st.sym_len))
    nw = plan.info()["waves_per_workgroup"]
    print("N=%d k=%d %s inject=%d kernel %s F=%d waves %d" % (n_fft, k, system, inject, plan.kernel_id(), F, nw))
    def launch():
        if inject:
            counts = plan.new_counts()
            plan.launch_injected(F, dl, dn, counts)
            torch.cuda.synchronize()
            plan.status()
            return counts.cpu().numpy().view(np.uint64).astype(np.int64)
        return plan.run(off, F).astype(np.int64)
    os.environ["WOFDM_DELAY_POINT"] = "0"
    for _ in range(3):
        ref = launch()
    assert np.array_equal(ref, launch())
    full = (1 << nw) - 1
    sets = [("wave 0", 1), ("wave 1", 2), ("last wave", 1 << (nw - 1)), ("all but 0", full & ~1), ("all but last", full >> 1),
            ("odd", 0xAAAA & full), ("even", 0x5555 & full), ("lower half", (1 << (nw // 2)) - 1), ("upper half", full & ~((1 << (nw // 2)) - 1)),
            ("wave 5", 1 << min(5, nw - 1))]
    bad = 0
    for point in (11, 12, 1, 2, 3, 4, 5, 6, 7, 8, 9):
        res = []
        for name, mask in sets:
            os.environ["WOFDM_DELAY_POINT"] = str(point)
            os.environ["WOFDM_DELAY_WAVES"] = hex(mask)
            os.environ["WOFDM_DELAY_LEN"] = os.environ.get("PROBE_LEN", "8")
            d = launch() - ref
            if d.any():
                bad += 1
                res.append("%s: bit %s sym %s" % (name, d[..., 0].ravel().tolist(), d[..., 2].ravel().tolist()))
        print("point %2d: %s" % (point, "; ".join(res) if res else "no change under any of %d wave sets" % len(sets)))
    print("delay settings that changed the counters:", bad)
