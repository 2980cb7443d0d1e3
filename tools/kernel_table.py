#!/usr/bin/env python3
"""Developer tool (build container): registers / spills / scratch of EVERY instantiation of the frame
kernel, from `hipcc -S` of all 15 translation units.

    python tools/kernel_table.py > profiles/r02_kernel_table.json

tests/test_gpu_parity.py reads the committed table and runs the sharp-parity case for every production
(non-instrumented) instantiation whose ScratchSize is not zero."""
import json
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "w-ofdm-optimization_amd", "csrc", "wofdm_kernel.hip")
FIELDS = ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")


def one(nk):
    n, k = nk
    out = os.path.join(tempfile.mkdtemp(), "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                    "-fno-slp-vectorize", "-DWOFDM_TU_N=%d" % n, "-DWOFDM_TU_K=%d" % k, "-S",
                    "--cuda-device-only", "-o", out, SRC], check=True, stderr=subprocess.DEVNULL)
    rows, cur = [], None
    for l in open(out):
        m = re.match(r"\s+\.name:\s+(\S+)", l)
        if m:
            t = re.search(r"wofdm_frames_kernelILi(\d+)ELi(\d)ELi(\d)ELb(\d)ELb(\d)ELi(\d)", m.group(1))
            cur = dict(zip(("n_fft", "k", "layout", "inject", "dump", "var"), map(int, t.groups()))) if t else None
            if cur:
                rows.append(cur)
        m = re.match(r"\s+\.(\w+):\s+(\d+)", l)
        if m and cur is not None and m.group(1) in FIELDS:
            cur[m.group(1)] = int(m.group(2))
    return rows


if __name__ == "__main__":
    nks = [(n, k) for n in (64, 128, 256, 512, 1024) for k in (2, 4, 6)]
    with ThreadPoolExecutor(6) as ex:
        rows = [r for part in ex.map(one, nks) for r in part]
    rows.sort(key=lambda r: (r["n_fft"], r["k"], r["layout"], r["var"], r["inject"], r["dump"]))
    json.dump({"source": "hipcc -S, ROCm 7.2, gfx950; tools/kernel_table.py", "kernels": rows}, sys.stdout, indent=0)
