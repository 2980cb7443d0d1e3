"""Static checks of the built library's device code (no GPU).

On gfx950 a gap of 7 or more wait states in front of one of the later MFMAs of the FIR's chain of six dependent
in-place MFMAs -- an instruction fetch, instructions scheduled in between -- corrupts packed op_sel arithmetic of
the other waves of the SIMD (hazard 1; tools/ubench/mfma_stall_victim.hip): that was behind
round 2's wrong first launches (a chain across a 4 KB page) and its sporadically wrong frames (compiler-scheduled
chains).  The guard is the shape of the code, and this file checks it in the disassembly of the BUILT library:
every run of MFMAs is six long, back to back, and sits inside ONE 64-byte instruction-cache line.  The kernels of
layouts 10 / 11 / 12 (both transforms on the matrix pipe: MFMAs as compiler builtins, scheduled among the vector
instructions -- trains of MFMAs at every spacing, which no alignment could make harmless, tools/ubench/
mfma_block_train.hip) are guarded the other way round: they contain NO victim, i.e. no v_pk_* instruction with an op_sel
source swizzle.  And no MFMA of any kernel has its destination on top of one of its own A / B operands.  Also: no kernel contains a flat instruction
(the LDS flag words are ds_read / ds_write), and the committed kernel table -- which selects the spilling kernels the
GPU tests visit -- describes this build."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "w-ofdm-optimization_amd", "libwofdm_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def _code_objects(lib, tmp):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
    for i, p in enumerate(starts):
        end = starts[i + 1] if i + 1 < len(starts) else len(data)
        b = os.path.join(tmp, "b%d.bin" % i)
        open(b, "wb").write(data[p:end])
        co = os.path.join(tmp, "b%d.co" % i)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=" + TARGET,
                        "--input=" + b, "--output=" + co], check=True)
        yield co


def scan(co):
    """([(function, address of the first MFMA, address behind the last, length)] of every run of consecutive
    MFMAs, {function: number of flat_* instructions}, {function: number of v_pk_* instructions with op_sel:[..]})."""
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
    fn, run, out, flat, swz, pending = None, [], [], {}, {}, []
    last_wr, last_ins = set(), ""          # vector registers the previous instruction wrote (a vector-ALU instruction only)
    rng = lambda t: (lambda m: range(int(m.group(1)), int(m.group(2)) + 1))(re.match(r"v\[(\d+):(\d+)\]", t))  # noqa: E731
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            fn, run = m.group(1), []
            flat[fn] = 0
            swz[fn] = 0
            continue
        m = re.search(r"//\s*([0-9A-Fa-f]+):", line)
        if not m:
            continue
        if re.match(r"\s*flat_", line):
            flat[fn] += 1
        if re.match(r"\s*v_pk_\w+ .*op_sel:\[", line):
            swz[fn] += 1
        ins = line.split("//")[0].strip()
        # cycles between an MFMA and the first vector-ALU write to one of its A / B operand registers (the instruction
        # reads them late and nothing interlocks; loads land later than that anyway)
        for pend in list(pending):
            wr = set()
            if ins.startswith("v_") and not ins.startswith("v_mfma") and not ins.startswith("v_cmp"):
                d = ins.split(None, 1)[1].split(",")[0].strip() if " " in ins else ""
                mm = re.match(r"v\[(\d+):(\d+)\]", d) or re.match(r"v(\d+)$", d)
                if mm:
                    wr = set(range(int(mm.group(1)), int(mm.group(mm.lastindex)) + 1))
            if wr & pend["src"]:
                swz.setdefault("__war__", []).append((fn[:70], pend["text"][:60], ins[:50], pend["cyc"]))
                pending.remove(pend)
                continue
            mn = re.match(r"s_nop (\d+)", ins)
            pend["cyc"] += (int(mn.group(1)) + 1) if mn else (8 if ins.startswith("v_mfma") else (4 if ins.startswith("v_") else 1))
            if pend["cyc"] >= 12 or ins.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier")):
                pending.remove(pend)
        if "v_mfma" in line:
            # the MFMA does not see what the vector instruction DIRECTLY in front of it wrote (tools/ubench/mfma_after_mix.hip: one
            # instruction in between is enough); the compiler spaces its own instructions, not the inline-asm ones of split_h
            srcs = set()
            for t in re.findall(r"v\[\d+:\d+\]", ins)[1:]:
                srcs |= set(rng(t))
            if srcs & last_wr:
                swz.setdefault("__raw__", []).append((fn[:70], last_ins[:60], ins[:60]))
            run.append(int(m.group(1), 16))
            ops0 = re.findall(r"v\[\d+:\d+\]", ins)
            if len(ops0) >= 3:
                pending.append({"src": set(rng(ops0[1])) | set(rng(ops0[2])), "cyc": 0, "text": ins})
            ops = re.findall(r"v\[\d+:\d+\]", line.split("//")[0])
            if len(ops) >= 3 and (set(rng(ops[0])) & (set(rng(ops[1])) | set(rng(ops[2])))):
                swz.setdefault("__overlap__", []).append((fn[:80], line.strip()[:90]))
        else:
            if len(run) > 1:
                out.append((fn, run[0], run[-1] + 8, len(run)))
            run = []
        last_wr, last_ins = set(), ins
        if ins.startswith("v_") and not ins.startswith("v_mfma") and not ins.startswith(("v_cmp_", "v_cmpx_")) and " " in ins:
            d = ins.split(None, 1)[1].split(",")[0].strip()
            mm = re.match(r"v\[(\d+):(\d+)\]", d) or re.match(r"v(\d+)$", d)
            if mm:
                last_wr = set(range(int(mm.group(1)), int(mm.group(mm.lastindex)) + 1))
    return out, flat, swz


def _is_mdft(fn):
    """kernel of a layout with transforms on the matrix pipe (third template argument of wofdm_frames_kernel)"""
    m = re.search(r"wofdm_frames_kernelILi\d+ELi\dELi(\d+)E", fn)
    return bool(m) and int(m.group(1)) in (10, 11, 12, 13, 14, 15)


def mfma_chains(co):
    return scan(co)[0]


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "llvm-objdump")) and shutil.which("objcopy")),
                    reason="needs the built library and the ROCm LLVM tools")
def test_mfma_chains_sit_in_one_line_and_no_kernel_uses_flat_instructions():
    n_chains, n_kernels, n_mdft, bad, flat_in, victims, no_victims = 0, 0, 0, [], [], [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for co in _code_objects(LIB, tmp):
            chains, flat, swz = scan(co)
            assert not swz.pop("__overlap__", []), "an MFMA's destination overlaps its own operand"
            raw = swz.pop("__raw__", [])
            assert not raw, ("a vector instruction writes an MFMA's source directly in front of it", raw[:5], len(raw))
            war = swz.pop("__war__", [])
            assert not war, ("an MFMA operand is overwritten within 12 cycles of the MFMA", war[:5])
            for fn, a, b, n in chains:
                if _is_mdft(fn):                                 # compiler-scheduled: runs of any length, anywhere
                    continue
                n_chains += 1
                assert n == 6, (fn, hex(a), n)                   # the chain is one block of six
                if a // 64 != (b - 1) // 64:
                    bad.append((fn[:80], hex(a)))
            for fn, n in flat.items():
                n_kernels += "wofdm_frames_kernel" in fn
                if n:                                            # (any kernel of the library)
                    flat_in.append((fn[:80], n))
            for fn, n in swz.items():
                if _is_mdft(fn):
                    n_mdft += 1
                    if n:
                        victims.append((fn[:80], n))
                elif "wofdm_frames_kernel" in fn and n:
                    no_victims += 1
    assert n_chains > 1000 and n_kernels == 708      # every kernel of the library was looked at
    assert not bad, bad[:5]
    assert not flat_in, flat_in[:5]
    # the kernels that issue MFMA trains hold nothing a train can corrupt (and the scan does see such instructions elsewhere)
    assert n_mdft == 204 and not victims, victims[:5]
    assert no_victims > 100


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "llvm-readelf")) and shutil.which("objcopy")),
                    reason="needs the built library and the ROCm LLVM tools")
def test_committed_kernel_table_describes_the_built_library():
    """profiles/kernel_table.json (tools/kernel_table.py) lists registers / spills / ScratchSize per instantiation;
    tests/test_gpu_parity.py::test_every_spilling_production_kernel takes its cases from it."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_table
    built = kernel_table.table(LIB)
    committed = json.load(open(os.path.join(ROOT, "profiles", "kernel_table.json")))["kernels"]
    key = lambda r: (r["n_fft"], r["k"], r["layout"], r["inject"], r["dump"], r["var"])   # noqa: E731
    assert len(built) == len(committed) == 708
    spill = lambda rows: sorted(key(r) for r in rows if r["private_segment_fixed_size"] > 0)   # noqa: E731
    assert spill(built) == spill(committed), "rebuild the table: python tools/kernel_table.py > profiles/kernel_table.json"
