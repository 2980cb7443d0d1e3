"""Static check of the built library's device code (no GPU): every chain of dependent in-place MFMAs of the
FIR (fir_mma in csrc/wofdm_kernel.hip) must sit inside ONE 64-byte instruction-cache line -- and so inside one
page.  A chain that straddled a 4 KB page produced wrong sums in a few frames of a kernel's first launch in a
process ("Compiler and hardware hazards")."""
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


def mfma_chains(co):
    """[(function, address of the first MFMA, address behind the last)] of every run of consecutive MFMAs."""
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
    fn, run, out = None, [], []
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            fn, run = m.group(1), []
            continue
        m = re.search(r"//\s*([0-9A-Fa-f]+):", line)
        if not m:
            continue
        if "v_mfma" in line:
            run.append(int(m.group(1), 16))
        else:
            if len(run) > 1:
                out.append((fn, run[0], run[-1] + 8, len(run)))
            run = []
    return out


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "llvm-objdump")) and shutil.which("objcopy")),
                    reason="needs the built library and the ROCm LLVM tools")
def test_no_mfma_chain_straddles_an_instruction_cache_line():
    n_chains, bad = 0, []
    with tempfile.TemporaryDirectory() as tmp:
        for co in _code_objects(LIB, tmp):
            for fn, a, b, n in mfma_chains(co):
                n_chains += 1
                assert n == 6, (fn, hex(a), n)                   # the chain is one block of six
                if a // 64 != (b - 1) // 64:
                    bad.append((fn[:80], hex(a)))
    assert n_chains > 1000            # every matrix-pipe kernel of the library was looked at
    assert not bad, bad[:5]
