#!/usr/bin/env python3
"""Developer tool (GPU box): does any production kernel read an LDS word, a register or a scratch slot before
writing it?  For each configuration: fill every CU's LDS, registers and scratch with a pattern
(libwofdm_poison.so: lds_poison, reg_poison, scratch_poison), launch, and
compare the counters with the same plan's launch after its own earlier launch (whose leftovers are what a
read-before-write would see on every launch but the first)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wofdm_amd as W
torch.cuda.init()        # (torch's HIP runtime first: the helper library brings /opt/rocm's)
torch.zeros(1, device="cuda")
P = ctypes.CDLL(os.path.join(ROOT, "tests", "native", "libwofdm_poison.so"))
P.lds_poison.argtypes = [ctypes.c_uint32]
P.lds_peek.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int]
P.reg_poison.argtypes = [ctypes.c_uint32]
P.scratch_poison.argtypes = [ctypes.c_uint32]
ch = np.load(os.path.join(ROOT, "tests", "golden", "channels_vehA.npz"))["h"]

# self-check of the tool
assert P.lds_poison(0xC0FFEE11) == 0
hits = np.zeros(512, np.uint32)
assert P.lds_peek(0xC0FFEE11, hits.ctypes.data, 512) == 0
print("lds_peek: %d of 512 workgroups saw a fully poisoned LDS (min share %.3f)" % ((hits == 40960).sum(), hits.min() / 40960))

PATS = (0xFFFFFFFF, 0x7BFF7BFF, 0x7F800000, 0x3C003C00, 0x00010001)
cases = []
for n, k in ((1024, 6), (1024, 2), (1024, 4), (512, 4), (512, 6), (256, 4), (256, 6), (128, 4), (64, 2)):
    for system in ("WOLA", "CPW", "wtx", "CPwtx", "wrx", "CPwrx", "CP"):
        cases.append((system, n, k, 16))
cases += [("WOLA", 1024, 6, 13), ("WOLA", 512, 4, 7), ("WOLA", 256, 4, 12), ("WOLA", 256, 4, 6)]
bad = 0
for system, n, k, S in cases:
    cp = 32 if n >= 256 else 16
    try:
        st = W.make_structure(system, n, cp)
    except Exception as e:
        print("skip", system, n, e); continue
    cfg = W.make_cfg(st, k, S, 21, 2, 3, 1, seed=8)
    F = max(4, int(2e6 / ((S - 1) * n * k)))
    wt, wr = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    with W.Plan(cfg, wt, wr, ch[11:13].astype(np.complex64), np.array([5.0, 15.0, 25.0], np.float32)) as plan:
        ref = None
        out = []
        for pat in PATS:
            assert P.lds_poison(pat) == 0 and P.scratch_poison(pat) == 0 and P.reg_poison(pat) == 0
            r = plan.run(3, F)[..., [0, 2]].astype(np.int64)
            r2 = plan.run(3, F)[..., [0, 2]].astype(np.int64)
            if ref is None:
                ref = r2
            out.append((pat, int(np.abs(r - ref).sum()), int(np.abs(r2 - ref).sum())))
        flag = any(a or b for _, a, b in out)
        bad += flag
        print("%-6s N=%4d k=%d S=%2d kernel %s F=%d: %s%s" % (
            system, n, k, S, plan.kernel_id(), F,
            " ".join("%08x:%d/%d" % o for o in out), "   <-- LDS-DEPENDENT" if flag else ""))
print("configurations whose result depends on what the LDS held:", bad)
