#!/usr/bin/env python3
"""Developer tool (build container): registers / spills / scratch of EVERY instantiation of the frame kernel, read
from the metadata notes of the BUILT library (what ships, per-translation-unit compiler flags included).

    python tools/kernel_table.py > profiles/kernel_table.json

tests/test_gpu_parity.py reads the committed table and runs the sharp-parity case for every production
(non-instrumented) instantiation whose ScratchSize is not zero; tests/test_code_layout.py checks that the table still
describes the built library."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
FIELDS = ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")


def table(lib):
    import test_code_layout as T0
    T = T0.V                                        # (the build's verifier: w-ofdm-optimization_amd/csrc/verify_code_layout.py)
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for co in T.code_objects(lib, tmp):
            notes = subprocess.run([os.path.join(T.LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True,
                                   check=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk)
                t = name and re.search(r"wofdm_frames_kernelILi(\d+)ELi(\d)ELi(\d+)ELb(\d)ELb(\d)ELi(\d)", name.group(1))
                if not t:
                    continue
                row = dict(zip(("n_fft", "k", "layout", "inject", "dump", "var"), map(int, t.groups())))
                for f in FIELDS:
                    row[f] = int(re.search(r"\." + f + r":\s+(\d+)", blk).group(1))
                rows.append(row)
    rows.sort(key=lambda r: (r["n_fft"], r["k"], r["layout"], r["var"], r["inject"], r["dump"]))
    return rows


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "w-ofdm-optimization_amd", "libwofdm_hip.so")
    json.dump({"source": "metadata notes of libwofdm_hip.so (hipcc, ROCm 7.2, gfx950); tools/kernel_table.py",
               "kernels": table(lib)}, sys.stdout, indent=0)
