"""Static checks of the built library's device code (no GPU).

The guards of DESIGN.md section 4 are properties of the GENERATED code: every run of MFMAs of a kernel that holds op_sel-swizzled
packed arithmetic is six long, back to back, inside one 64-byte instruction-cache line; the kernels whose MFMAs are compiler builtins
scheduled among the vector instructions (layouts 10 ... 15) hold no such instruction; no MFMA has its destination on top of one of
its own operands; no vector instruction writes an MFMA operand within the 12 wait states behind the MFMA, or fewer than two wait
states in front of it; no kernel contains a flat instruction.  The scan lives in w-ofdm-optimization_amd/csrc/verify_code_layout.py and
is a step of the BUILD (`make` runs it behind the link and does not put a failing library in place); this file runs it once more on
the library the tests use, checks that a deliberately mis-scheduled object is rejected, and that the committed kernel table -- which
selects the spilling kernels the GPU tests visit -- describes this build."""
import importlib.util
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "w-ofdm-optimization_amd", "csrc")
LIB = os.path.join(ROOT, "w-ofdm-optimization_amd", "libwofdm_hip.so")

_spec = importlib.util.spec_from_file_location("verify_code_layout", os.path.join(CSRC, "verify_code_layout.py"))
V = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(V)
LLVM = V.LLVM

needs_tools = pytest.mark.skipif(not (os.path.exists(LIB) and V.tools_present()), reason="needs the built library and the ROCm LLVM tools")


@needs_tools
def test_mfma_chains_sit_in_one_line_and_no_kernel_uses_flat_instructions():
    bad, summ = V.verify(LIB)
    assert not bad, bad[:5]
    assert summ["chains"] > 1000 and summ["kernels"] == 756      # every kernel of the library was looked at
    # the kernels that issue MFMA trains hold nothing a train can corrupt (and the scan does see such instructions elsewhere)
    assert summ["mdft_kernels"] == 252 and summ["kernels_with_swizzles"] > 100
    # every vector write of an MFMA source sits at least two wait states in front of the MFMA
    assert min(summ["write_to_mfma_states"]) >= V.MIN_WRITE_TO_MFMA, summ["write_to_mfma_states"]


@needs_tools
@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_a_mis_scheduled_build_is_rejected():
    """The same sources with the wait states behind the MFMA chains taken out (-DWOFDM_MMA_TAIL="s_nop 0": the register allocator is
    then free to reuse a chain's operand registers right behind it) must not pass the build's verification step."""
    with tempfile.TemporaryDirectory() as tmp:
        obj, so = os.path.join(tmp, "k.o"), os.path.join(tmp, "k.so")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize",
                        "-DWOFDM_TU_N=128", "-DWOFDM_TU_K=2", '-DWOFDM_MMA_TAIL="s_nop 0"', "-c", os.path.join(CSRC, "wofdm_kernel.hip"),
                        "-o", obj], check=True, stderr=subprocess.DEVNULL)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj], check=True,
                       stderr=subprocess.DEVNULL)
        bad, _ = V.verify(so)
        assert any("behind its MFMA" in b for b in bad), bad[:3]
        rc = subprocess.run(["python3", os.path.join(CSRC, "verify_code_layout.py"), so], capture_output=True).returncode
        assert rc != 0


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
    assert len(built) == len(committed) == 756
    spill = lambda rows: sorted(key(r) for r in rows if r["private_segment_fixed_size"] > 0)   # noqa: E731
    assert spill(built) == spill(committed), "rebuild the table: python tools/kernel_table.py > profiles/kernel_table.json"
