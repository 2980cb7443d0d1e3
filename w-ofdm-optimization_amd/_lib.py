"""ctypes binding of libwofdm_hip.so (include/wofdm.h).

There is no CPU fallback: if the library is missing or no gfx950 device is usable,
the calls raise.  torch is not needed here -- device pointers are plain integers
(e.g. ``tensor.data_ptr()``), streams are raw ``hipStream_t`` values.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# WOFDM_LIB: developer override for A/B builds of the same ABI (never a different backend)
LIB_PATH = os.environ.get("WOFDM_LIB", os.path.join(_HERE, "libwofdm_hip.so"))
_LIB = None

WOFDM_OK = 0
ERRORS = {-1: "WOFDM_E_INVALID", -2: "WOFDM_E_UNSUPPORTED", -3: "WOFDM_E_HIP", -4: "WOFDM_E_NOMEM"}
MAX_TAPS = 21
#: wofdm_plan_set_option: diagnostic kernel choice (include/wofdm.h)
OPTIONS = {"fir_valu": 0, "max_spw": 1, "txmask_direct": 2, "dft_valu": 3}
MAX_SYMS = 16

#: every symbol include/wofdm.h declares (tests check the .so exports them all)
EXPORTS = (
    "wofdm_version", "wofdm_device_count", "wofdm_last_error", "wofdm_noise_len",
    "wofdm_plan_create", "wofdm_plan_destroy", "wofdm_plan_launch", "wofdm_plan_launch_timed",
    "wofdm_plan_launch_injected", "wofdm_plan_dump_frame", "wofdm_plan_info", "wofdm_run",
    "wofdm_run_injected", "wofdm_philox_kat", "wofdm_plan_set_allocation",
    "wofdm_plan_set_tx_mask", "wofdm_plan_status", "wofdm_plan_kernel_id", "wofdm_plan_set_option",
    "wofdm_interference", "wofdm_tx_psd",
)


class WofdmError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "WOFDM_E_?"), code, message))
        self.code = code


class Cfg(C.Structure):
    """Mirror of ``wofdm_cfg``."""
    _fields_ = [(n, C.c_int32) for n in (
        "n_fft", "bits_per_sc", "syms_per_frame", "cp", "cs", "tail_tx", "tail_rx", "prefix_rm",
        "circ_shift", "n_taps", "n_channels", "n_snr", "n_window_pairs",
        "noise_before_truncate")] + [
        ("frames_per_cell", C.c_uint64), ("frame_offset", C.c_uint64), ("seed", C.c_uint64)]

    @property
    def sym_len(self):
        return self.n_fft + self.cp + self.cs

    @property
    def stride(self):
        return self.sym_len - self.tail_tx

    @property
    def frame_len(self):
        return self.tail_tx + self.syms_per_frame * self.stride

    @property
    def n_cells(self):
        return self.n_window_pairs * self.n_snr * self.n_channels


class Dump(C.Structure):
    """Mirror of ``wofdm_dump`` (host buffers)."""
    _fields_ = [(n, C.c_void_p) for n in ("labels_tx", "X", "tx", "conv", "rx", "Y", "Xhat",
                                          "labels_rx", "gain", "unit_noise")]


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the kernel sources and their build flags: what a committed
    measurement of the kernel (profiles/hbm_traffic.json) is stamped with."""
    import hashlib
    h = hashlib.sha256()
    for name in ("wofdm_kernel.hip", "wofdm_kernel.h", "philox.h", "Makefile"):
        with open(os.path.join(_HERE, "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build(verbose=False):
    """Compile libwofdm_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j8"], check=True, stdout=out)
    return LIB_PATH


def load():
    """Load the HIP library; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64/libhsa.  If
    # this library pulled in /opt/rocm's copy first, a later `import torch` would find "No HIP
    # GPUs".  Importing torch first (when it is installed) makes both share torch's runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `make -C %s` (hipcc, gfx950). The w-OFDM hot path has "
            "no CPU fallback." % (LIB_PATH, os.path.join(_HERE, "csrc")))
    L = C.CDLL(LIB_PATH)
    vp, u64 = C.c_void_p, C.c_uint64
    L.wofdm_version.restype = C.c_int
    L.wofdm_device_count.restype = C.c_int
    L.wofdm_last_error.restype = C.c_char_p
    L.wofdm_noise_len.argtypes = [C.POINTER(Cfg)]
    L.wofdm_plan_create.argtypes = [C.POINTER(vp), C.POINTER(Cfg), C.c_int, vp, vp, vp, vp]
    L.wofdm_plan_destroy.argtypes = [vp]
    L.wofdm_plan_launch.argtypes = [vp, u64, u64, vp, vp]
    L.wofdm_plan_launch_timed.argtypes = [vp, u64, u64, vp, vp, C.POINTER(C.c_float)]
    L.wofdm_plan_launch_injected.argtypes = [vp, u64, vp, vp, vp, vp]
    L.wofdm_plan_dump_frame.argtypes = [vp, C.c_uint32, u64, vp, vp, vp, C.POINTER(Dump)]
    L.wofdm_plan_info.argtypes = [vp, vp]
    L.wofdm_plan_kernel_id.argtypes = [vp, vp]
    L.wofdm_plan_set_allocation.argtypes = [vp, vp]
    L.wofdm_plan_set_tx_mask.argtypes = [vp, vp]
    L.wofdm_plan_status.argtypes = [vp]
    L.wofdm_plan_set_option.argtypes = [vp, C.c_int32, C.c_int32]
    L.wofdm_run.argtypes = [C.POINTER(Cfg), C.c_int, vp, vp, vp, vp, vp]
    L.wofdm_run_injected.argtypes = [C.POINTER(Cfg), C.c_int, vp, vp, vp, vp, vp, vp, vp]
    L.wofdm_philox_kat.argtypes = [C.c_int, vp, vp, vp]
    L.wofdm_interference.argtypes = [C.POINTER(Cfg), C.c_int, vp, vp, vp, vp]
    L.wofdm_tx_psd.argtypes = [C.POINTER(Cfg), C.c_int, vp, vp, C.c_int, C.c_int, vp]
    _LIB = L
    return L


def check(rc):
    if rc != WOFDM_OK:
        raise WofdmError(rc, load().wofdm_last_error().decode("utf-8", "replace"))
    return rc


def f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError("expected shape %s, got %s" % (tuple(shape), a.shape))
    return a


def c64_as_f32(a, shape=None):
    """complex array -> contiguous float32 array with a trailing (re, im) axis"""
    a = np.ascontiguousarray(a, dtype=np.complex64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError("expected shape %s, got %s" % (tuple(shape), a.shape))
    return a.view(np.float32).reshape(a.shape + (2,))
