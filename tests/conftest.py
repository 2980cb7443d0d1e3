import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import wofdm_amd
        return wofdm_amd._lib.load().wofdm_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # A GPU test on a box without a usable device is an ERROR on the GPU box (no silent
    # fallback), but when somebody runs the whole suite on a CPU-only container we skip.
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no ROCm device visible")
    for item in items:
        if "gpu" in item.keywords and "gpu" not in (config.getoption("-m") or ""):
            item.add_marker(skip)


def pytest_report_header(config):
    # which GPU the -m gpu run was on (a transient first-launch deviation seen in round 2 is tracked per device)
    if "gpu" not in (config.getoption("-m") or "") or "not gpu" in (config.getoption("-m") or ""):
        return None
    try:
        import subprocess
        out = subprocess.run(["rocm-smi", "--showuniqueid"], capture_output=True, text=True, timeout=30).stdout
        ids = [l.split(":")[-1].strip() for l in out.splitlines() if "Unique ID" in l and "GPU[" in l]
        return "GPU unique id: %s" % ", ".join(ids)
    except Exception as e:                                   # noqa: BLE001
        return "GPU unique id: unavailable (%r)" % (e,)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def channels(golden):
    return golden("channels_vehA.npz")["h"]      # [100][21] complex128
