#!/usr/bin/env python3
"""Developer tool (build container): static instruction mix of the frame kernels, from `hipcc -S`.

    python tools/isa_mix.py 256 4 [-D...]            # every instantiation of that translation unit
    python tools/isa_mix.py 256 4 --kernel 6,0,0,0   # one (LAY, INJECT, DUMP, VAR), with the frame
                                                     # loop split at the `; wofdm_mark` comments

Per kernel: registers, scratch bytes, LDS, occupancy, and instruction counts by class for the whole
kernel and for the frame loop (the outermost back edge).  Counts are static: an instruction inside an
inner loop counts once.  Writes plain text to stdout (profiles/rNN_isa_mix.txt is made with it).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "w-ofdm-optimization_amd", "csrc", "wofdm_kernel.hip")

CLASSES = [
    ("pk_f32", r"v_pk_(fma|mul|add)_f32"),
    ("mfma", r"v_mfma_"),
    ("mad_u64", r"v_mad_u64_u32"),
    ("trans", r"v_(sin|cos|log|exp|sqrt|rcp|rsq)_f32"),
    ("lane_x", r"v_(readlane|writelane|readfirstlane)_b32"),
    ("dpp/perm", r"(_dpp|v_permlane|ds_bpermute|ds_swizzle|v_mov_b32_dpp)"),
    ("cvt", r"v_cvt_"),
    ("lds_ld", r"ds_read"),
    ("lds_st", r"ds_write"),
    ("vmem", r"(global|flat|buffer)_(load|store|atomic)"),
    ("scratch", r"scratch_(load|store)"),
    ("smem", r"s_(load|buffer_load)_"),
    ("salu", r"s_(?!waitcnt|nop|barrier|sleep|endpgm|branch|cbranch|load|buffer_load|setprio|sethalt|memtime)"),
    ("branch", r"s_(branch|cbranch)"),
    ("wait", r"s_(waitcnt|nop|barrier|sleep)"),
]


def classify(op):
    for name, pat in CLASSES:
        if re.match(pat, op) or (name == "dpp/perm" and re.search(pat, op)):
            return name
    if op.startswith("v_"):
        return "valu_other"
    return "other"


def compile_tu(n, k, extra):
    out = os.path.join(tempfile.mkdtemp(), "k.s")
    sched = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"] if n in ("64", "512", "1024") else []      # (the Makefile's SCHED_64 / SCHED_512 / SCHED_1024)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                    "-fno-slp-vectorize", *sched, "-DWOFDM_TU_N=" + n, "-DWOFDM_TU_K=" + k, *extra, "-S",
                    "--cuda-device-only", "-o", out, SRC], check=True, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def kernels(lines):
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines)
              if re.match(r"_ZN12_GLOBAL__N_119wofdm_frames_kernel\S+:", l)]
    meta = {}
    cur = None
    for l in lines:
        m = re.match(r"\s+\.name:\s+(\S+)", l)
        if m:
            cur = m.group(1)
            meta.setdefault(cur, {})
        m = re.match(r"\s+\.(vgpr_count|sgpr_count|agpr_count|private_segment_fixed_size|vgpr_spill_count|"
                     r"sgpr_spill_count|group_segment_fixed_size):\s+(\d+)", l)
        if m and cur:
            meta[cur][m.group(1)] = int(m.group(2))
    for st, name in starts:
        end = next(i for i in range(st, len(lines)) if lines[i].strip().startswith("s_endpgm"))
        tag = re.search(r"ILi(\d+)ELi(\d)ELi(\d+)ELb(\d)ELb(\d)ELi(\d)", name).groups()
        yield tag, name, lines[st:end + 1], meta.get(name, {})


def body_ops(body):
    """[(line index, opcode, text)] of the instructions, and {label: line index}"""
    ops, labels = [], {}
    for i, l in enumerate(body):
        t = l.strip()
        m = re.match(r"(\.LBB\S+|_Z\S+):", t)
        if m:
            labels[m.group(1)] = i
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        ops.append((i, t.split()[0], t))
    return ops, labels


def frame_loop(ops, labels):
    """(first, last) line index of the widest backward branch = the frame loop"""
    best = None
    for i, op, t in ops:
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] < i:
                span = i - labels[tgt]
                if best is None or span > best[1] - best[0]:
                    best = (labels[tgt], i)
    return best


def count(ops, lo=None, hi=None):
    c = {}
    for i, op, _ in ops:
        if lo is not None and not (lo <= i <= hi):
            continue
        cl = classify(op)
        c[cl] = c.get(cl, 0) + 1
    return c


VALU = ["pk_f32", "mfma", "mad_u64", "trans", "lane_x", "dpp/perm", "cvt", "valu_other"]
COLS = VALU + ["lds_ld", "lds_st", "vmem", "scratch", "smem", "salu", "branch", "wait"]


def row(label, c):
    valu = sum(c.get(x, 0) for x in VALU)
    return "%-22s %5d | " % (label, valu) + " ".join("%5d" % c.get(x, 0) for x in COLS)


def main():
    args = sys.argv[1:]
    n, k = args[0], args[1]
    only = None
    extra = []
    i = 2
    while i < len(args):
        if args[i] == "--kernel":
            only = tuple(args[i + 1].split(","))
            i += 2
        else:
            extra.append(args[i])
            i += 1
    lines = compile_tu(n, k, extra)
    hdr = "%-22s %5s | " % ("", "VALU") + " ".join("%5s" % x[:5] for x in COLS)
    print("# translation unit N=%s k=%s %s" % (n, k, " ".join(extra)))
    print("# kernel = (LAY, INJECT, DUMP, VAR); regs = VGPR/SGPR, spills = VGPR/SGPR, scratch bytes per lane")
    for tag, name, body, meta in kernels(lines):
        key = tag[2:]
        if only and key != only:
            continue
        ops, labels = body_ops(body)
        fl = frame_loop(ops, labels)
        print("\nkernel %s  vgpr %d sgpr %d  spills %d/%d  scratch %d B  instr %d" % (
            ",".join(key), meta.get("vgpr_count", -1), meta.get("sgpr_count", -1),
            meta.get("vgpr_spill_count", -1), meta.get("sgpr_spill_count", -1),
            meta.get("private_segment_fixed_size", -1), len(ops)))
        print(hdr)
        print(row("whole kernel", count(ops)))
        if fl:
            print(row("frame loop", count(ops, fl[0], fl[1])))
            if only:
                marks = [(i, l.strip()) for i, l in enumerate(body) if "wofdm_mark" in l and fl[0] <= i <= fl[1]]
                prev, pname = fl[0], "loop head"
                for i, t in marks:
                    print(row("  .. " + pname, count(ops, prev, i)))
                    prev, pname = i, t.split("wofdm_mark")[1].strip()
                print(row("  .. " + pname, count(ops, prev, fl[1])))


if __name__ == "__main__":
    main()
