#!/usr/bin/env python3
"""Developer tool: profiles/hbm_traffic.json from a PMC summary (tools/pmc_summary.py output), stamped
with the hash of the kernel sources it was measured on.  bench.py reports `roofline.traffic` only while
that stamp matches the sources it runs (wofdm_amd.kernel_source_hash()); otherwise null.

    python tools/hbm_traffic.py gpurun_out/r02_pmc_summary.txt "wofdm_frames_kernel<256,4,6,false,false,0>" > profiles/hbm_traffic.json
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wofdm_amd as W  # noqa: E402

vals = {}
for line in open(sys.argv[1]):
    m = re.match(r"(\S+)\s+n=\d+ mean=(\S+)", line)
    if m:
        vals[m.group(1)] = float(m.group(2))
fetch, write = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
json.dump({
    "kernel": sys.argv[2] if len(sys.argv) > 2 else "wofdm_frames_kernel",
    "kernel_source_hash": W.kernel_source_hash(),
    "workload": "bench.py default (C2), one launch = 12e6 OFDM symbols",
    "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
    "bytes_per_launch": int(round((2 * fetch + write) * 1024)),
    # per-pipe counters of the same launch, for bench.py's roofline.pipes (issued work, not algorithmic work)
    "valu_insts_per_launch": vals.get("SQ_INSTS_VALU"), "mfma_insts_per_launch": vals.get("SQ_INSTS_MFMA"),
    "valu_active_quadcycles_per_launch": vals.get("SQ_ACTIVE_INST_VALU"),
    "gui_active_cycles_per_launch": vals.get("GRBM_GUI_ACTIVE"),
    "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_round.sh); "
           "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 with the gfx950 x2 read correction of "
           "MI355X_MICROARCH.md.  The accesses here are scalar/dword constants and 8-byte atomics, which "
           "the guide calls uncalibrated: an order of magnitude, not a byte count.",
}, sys.stdout, indent=1)
