"""Experiment driver: the script body of ``matlab/main_BER_calculation.m:21-202`` and the
``run_sim`` mode of ``python/wofdm_optimization.py:107-131`` as functions.

File formats are the reference's own:
  * ``settingsData.mat``      struct ``settingsData`` (matlab/window_optimization.m:38-49)
  * ``optimized_windows/optimal_win_<type>_VehA200_<cp>CP.mat`` with ``optimizedWindow`` or
    ``optimizedWindowCase{A,B}Step{1,2,3}`` (window_optimization.m:299-305, 351-357, 586-589)
  * ``channels/vehA200channel2.mat``  variable ``vehA200channel2`` [realisations x taps]
  * results ``ber_results/{optimized,rc}_ber_<type>_<cp>CP.mat`` (main_BER_calculation.m:205-227)

The frame loop itself runs on the GPU (simulation.py -> include/wofdm.h).  With several
processes (``torch.distributed`` initialised) every rank takes a contiguous share of the
``ensemble`` frames of every cell and the integer counters are all-reduced once per file.
"""
import os
import re

import numpy as np

from . import distributed as D
from . import simulation as S
from . import variants as V

#: matlab/window_optimization.m:39-47
DEFAULT_SETTINGS = {
    "generalSettings": {"numberSubcarriers": 256, "bitsPerSubcarrier": 4,
                        "cyclicPrefix": np.arange(10, 33, 2), "symbolsPerTx": 16, "ensemble": 100,
                        "snrValues": np.linspace(-20, 50, 30)},
    "wtx": {"tailTx": 8, "tailRx": 0}, "wrx": {"tailTx": 0, "tailRx": 10},
    "WOLA": {"tailTx": 8, "tailRx": 10}, "CPW": {"tailTx": 8, "tailRx": 10},
    "CPwtx": {"tailTx": 8, "tailRx": 0}, "CPwrx": {"tailTx": 0, "tailRx": 10},
}

_WINDOW_FILE = re.compile(r"^optimal_win_(?P<type>[A-Za-z]+)_.*?_(?P<cp>\d+)CP\.mat$")


def _struct_to_dict(obj):
    """scipy.io.loadmat(struct_as_record=False, squeeze_me=True) struct -> nested dict."""
    if hasattr(obj, "_fieldnames"):
        return {k: _struct_to_dict(getattr(obj, k)) for k in obj._fieldnames}
    return obj


def load_settings(path):
    """``settingsData.mat`` -> nested dict (main_BER_calculation.m:50-57)."""
    from scipy.io import loadmat
    m = loadmat(path, struct_as_record=False, squeeze_me=True)
    return _struct_to_dict(m["settingsData"])


def save_settings(path, settings=None):
    from scipy.io import savemat
    savemat(path, {"settingsData": settings or DEFAULT_SETTINGS})


def load_channels_mat(path, variable="vehA200channel2"):
    """[realisations x taps] complex (rows = realisations; SURVEY quirk Q2: the row count, not
    ``length()``, is the number of realisations)."""
    from scipy.io import loadmat
    h = np.atleast_2d(np.asarray(loadmat(path)[variable]))
    return h.astype(np.complex128)


def load_window_file(path):
    """Window ``.mat`` file -> {variable name: window vector} (diagonals of the stored matrices)."""
    from scipy.io import loadmat
    out = {}
    for k, v in loadmat(path).items():
        if k.startswith("optimizedWindow"):
            v = np.asarray(v)
            out[k] = np.real(np.diag(v) if v.ndim == 2 and v.shape[0] == v.shape[1] and v.shape[0] > 1
                             else v.reshape(-1)).astype(np.float64)
    return out


def parse_window_file_name(name):
    """``optimal_win_<type>_VehA200_<cp>CP.mat`` -> (type, cp) as main_BER_calculation.m:41-43
    does (token 3 and the number in front of ``CP.mat``); None for anything else."""
    m = _WINDOW_FILE.match(name)
    if not m or m.group("type") not in V.SYSTEMS:
        return None
    return m.group("type"), int(m.group("cp"))


def run_ber_calculation(settings_file="settingsData.mat", windows_folder="optimized_windows",
                        channels_file="./channels/vehA200channel2.mat", results_folder="ber_results",
                        device=0, seed=0, group=None, log=print):
    """The whole of main_BER_calculation.m: every window file of ``windows_folder`` -> BER curves
    for the optimised and the raised-cosine windows, written with the reference's file and
    variable names.  Returns {file name: results dict}."""
    settings = load_settings(settings_file) if isinstance(settings_file, str) else settings_file
    gen = settings["generalSettings"]
    channels = load_channels_mat(channels_file) if isinstance(channels_file, str) \
        else np.atleast_2d(np.asarray(channels_file))
    snr = np.atleast_1d(np.asarray(gen["snrValues"], dtype=np.float64))
    rank, world = 0, 1
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(group), dist.get_world_size(group)
    except ImportError:
        pass
    if rank == 0:
        os.makedirs(results_folder, exist_ok=True)
    done = {}
    for name in sorted(os.listdir(windows_folder)):
        parsed = parse_window_file_name(name)
        if parsed is None:                       # directories, *.log (m:32-39)
            continue
        type_ofdm, cp = parsed
        if rank == 0 and log:
            log("Working on file %s: %s-OFDM, CP %d." % (name, type_ofdm, cp))
        windows = load_window_file(os.path.join(windows_folder, name))
        tails = settings[type_ofdm]
        ensemble = int(gen["ensemble"])
        shard = D.frame_shard(ensemble, rank, world)
        _, counts = S.ber_for_window_file(
            type_ofdm, cp, windows, channels, snr, num_subcar=int(gen["numberSubcarriers"]),
            bits_per_subcar=int(gen["bitsPerSubcarrier"]), symbols_per_tx=int(gen["symbolsPerTx"]),
            ensemble=ensemble, tail_tx=int(tails["tailTx"]), tail_rx=int(tails["tailRx"]),
            seed=seed, device=device, frame_range=shard)
        if world > 1:
            if D._backend(group) == "nccl":           # RCCL reduces on this rank's GPU
                import torch
                torch.cuda.set_device(device)
            counts = D.reduce_counts_numpy(counts, group)
        results = S.results_from_counts(type_ofdm, counts)
        if rank == 0:
            S.save_ber_results(results_folder, type_ofdm, cp, results)
        done[name] = results
    return done


def run_sim(systems, cp_list, channel_path, window_path, simulation_path, monte_carlo=1,
            snr_arr=None, no_symbols=16, dft_len=256, device=0):
    """``python wofdm_optimization.py -m run_sim`` (lines 107-131): one work item per
    (system, CP); tails 8/10 as hard-coded there (118-123)."""
    snr_arr = np.arange(-21, 51, 3) if snr_arr is None else np.asarray(snr_arr)
    out = {}
    for system in systems:
        for cp in cp_list:
            ttx, trx = V.default_tails(system)
            out[(system, cp)] = S.simulation_fun((system, dft_len, cp, ttx, trx, channel_path,
                                                  window_path, monte_carlo, snr_arr, no_symbols,
                                                  simulation_path))
    return out
