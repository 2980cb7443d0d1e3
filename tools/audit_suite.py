#!/usr/bin/env python3
"""Developer tool (GPU box): the GPU test suite with every N >= 512 ``Plan.run`` audited frame by frame.

Needs the audit build (``bash tools/build_variant.sh audit "64 128 256 512 1024" "2 4 6" -- -DWOFDM_AUDIT=1``):
its frame kernels write, per (workgroup, frame, wave), the frame's bit / symbol errors, the noise gain, the
measured powers, HW_ID / XCC_ID and a clock.  Each audited run is launched twice; where the first launch's
records differ from the second's, the differing frames go to gpurun_out/audit.txt.  (Hunt for the
first-launch deviation)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["WOFDM_LIB"] = os.path.join(ROOT, "ab", "lib_audit.so")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import pytest
import wofdm_amd as W

OUT = os.path.join(ROOT, "gpurun_out", "audit.txt")
os.makedirs(os.path.dirname(OUT), exist_ok=True)
log = open(OUT, "a")
stats = dict(audited=0, differing=0)
plain_run = W.Plan.run


def audited_run(self, frame_offset, frames_per_cell):
    c = self.cfg
    if c.n_fft < 512 or frames_per_cell == 0:
        return plain_run(self, frame_offset, frames_per_cell)
    lib = self.lib
    lib.wofdm_plan_set_audit.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    cells = int(np.prod(self.counts_shape[:-1]))
    total = cells * int(frames_per_cell)
    grid = min(self.info()["workgroups"], total)
    items = -(-total // grid)
    buf = torch.zeros(grid * items * 16 * 8, dtype=torch.int32, device="cuda:%d" % self.device)
    assert lib.wofdm_plan_set_audit(self._h_plan, buf.data_ptr(), items) == 0
    recs, outs = [], []
    for rep in range(2):
        buf.zero_()
        outs.append(plain_run(self, frame_offset, frames_per_cell))
        recs.append(buf.cpu().numpy().view(np.uint32).reshape(grid, items, 16, 8).copy())
    assert lib.wofdm_plan_set_audit(self._h_plan, None, 0) == 0
    stats["audited"] += 1
    a, b = recs
    d = (a[..., :5] != b[..., :5]).any(axis=-1)
    if d.any() or not np.array_equal(outs[0], outs[1]):
        stats["differing"] += 1
        test = os.environ.get("PYTEST_CURRENT_TEST", "?")
        print("=== %s\n    N=%d k=%d S=%d kernel %s cells %d frames/cell %d offset %d grid %d items/WG %d" % (
            test, c.n_fft, c.bits_per_sc, c.syms_per_frame, self.kernel_id(), cells, frames_per_cell,
            frame_offset, grid, items), file=log)
        print("    counters first - second launch (bit, sym per cell): %s" % (
            (outs[0].astype(np.int64) - outs[1].astype(np.int64))[..., [0, 2]].reshape(-1, 2).tolist()), file=log)
        q, r = divmod(total, grid)
        for wg, it in sorted(set(zip(*np.nonzero(d)[:2]))):
            start = wg * (q + 1) if wg < r else r * (q + 1) + (wg - r) * q
            item = start + it
            waves = np.nonzero(d[wg, it])[0]
            fa, fb = a[wg, it].view(np.float32), b[wg, it].view(np.float32)
            print("    WG %3d item %2d (cell %d frame %d) hw_id %08x xcc %x  gain %.9g / %.9g  Ps %.9g / %.9g  Pn %.9g / %.9g" % (
                wg, it, item // frames_per_cell, frame_offset + item % frames_per_cell, a[wg, it, 0, 5], a[wg, it, 0, 6] & 0xF,
                fa[0, 2], fb[0, 2], fa[0, 3], fb[0, 3], fa[0, 4], fb[0, 4]), file=log)
            print("        waves %s: bit errors %s / %s   symbol errors %s / %s   per-wave gain equal %s" % (
                waves.tolist(), a[wg, it, waves, 0].tolist(), b[wg, it, waves, 0].tolist(),
                a[wg, it, waves, 1].tolist(), b[wg, it, waves, 1].tolist(),
                bool((a[wg, it, :, 2] == b[wg, it, :, 2]).all())), file=log)
        log.flush()
    return outs[0]


plain_inj = W.Plan.launch_injected


def audited_inj(self, frames_per_cell, labels, unit_noise, counts, stream=None):
    """Injected launch: audited into a throw-away counter twice, then the real launch."""
    c = self.cfg
    if c.n_fft < 512:
        return plain_inj(self, frames_per_cell, labels, unit_noise, counts, stream)
    lib = self.lib
    lib.wofdm_plan_set_audit.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    cells = int(labels.shape[0])
    total = cells * int(frames_per_cell)
    grid = min(self.info()["workgroups"], total)
    items = -(-total // grid)
    buf = torch.zeros(grid * items * 16 * 8, dtype=torch.int32, device="cuda:%d" % self.device)
    assert lib.wofdm_plan_set_audit(self._h_plan, buf.data_ptr(), items) == 0
    recs, outs = [], []
    for rep in range(3):
        buf.zero_()
        tmp = torch.zeros_like(counts)
        plain_inj(self, frames_per_cell, labels, unit_noise, tmp, stream)
        torch.cuda.synchronize()
        outs.append(tmp.cpu().numpy())
        recs.append(buf.cpu().numpy().view(np.uint32).reshape(grid, items, 16, 8).copy())
    assert lib.wofdm_plan_set_audit(self._h_plan, None, 0) == 0
    stats["audited"] += 1
    test = os.environ.get("PYTEST_CURRENT_TEST", "?")
    q, r = divmod(total, grid)
    for x, y, nm in ((0, 1, "1st vs 2nd"), (1, 2, "2nd vs 3rd")):
        a, b = recs[x], recs[y]
        d = (a[..., :5] != b[..., :5]).any(axis=-1)
        if not d.any():
            continue
        stats["differing"] += 1
        print("=== INJECTED %s  %s\n    N=%d k=%d S=%d kernel %s cells %d frames/cell %d grid %d items/WG %d" % (
            nm, test, c.n_fft, c.bits_per_sc, c.syms_per_frame, self.kernel_id(), cells, frames_per_cell, grid, items), file=log)
        print("    counters (bit, sym per cell): %s" % ((outs[x] - outs[y])[..., [0, 2]].reshape(-1, 2).tolist()), file=log)
        for wg, it in sorted(set(zip(*np.nonzero(d)[:2]))):
            start = wg * (q + 1) if wg < r else r * (q + 1) + (wg - r) * q
            item = start + it
            waves = np.nonzero(d[wg, it])[0]
            fa, fb = a[wg, it].view(np.float32), b[wg, it].view(np.float32)
            print("    WG %3d item %2d (cell %d frame %d) hw_id %08x xcc %x  gain %.9g / %.9g  Ps %.9g / %.9g  Pn %.9g / %.9g" % (
                wg, it, item // frames_per_cell, item % frames_per_cell, a[wg, it, 0, 5], a[wg, it, 0, 6] & 0xF,
                fa[0, 2], fb[0, 2], fa[0, 3], fb[0, 3], fa[0, 4], fb[0, 4]), file=log)
            print("        waves %s: bit errors %s / %s   symbol errors %s / %s   per-wave gain equal %s" % (
                waves.tolist(), a[wg, it, waves, 0].tolist(), b[wg, it, waves, 0].tolist(),
                a[wg, it, waves, 1].tolist(), b[wg, it, waves, 1].tolist(),
                bool((a[wg, it, :, 2] == b[wg, it, :, 2]).all())), file=log)
    log.flush()
    return plain_inj(self, frames_per_cell, labels, unit_noise, counts, stream)


W.Plan.run = audited_run
W.Plan.launch_injected = audited_inj
rc = pytest.main(["tests", "-m", "gpu", "-q", "-x", "--no-header", "-p", "no:cacheprovider"] + sys.argv[1:])
print("audited runs %d, with a differing first launch %d, pytest rc %d" % (stats["audited"], stats["differing"], rc), file=log)
log.close()
print("audited runs %d, with a differing first launch %d" % (stats["audited"], stats["differing"]))
sys.exit(int(rc))
