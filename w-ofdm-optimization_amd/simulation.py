"""Host side of the w-OFDM Monte-Carlo hot path: the reference's operator interface on top of
the C ABI (include/wofdm.h).

Mirrors
  * ``run_simulation(...)``                 matlab/main_BER_calculation.m:230-274
  * ``wOFDMSystem(...).run_simulation(...)`` python/ofdm_utils/wofdm_simulation.py:368-481
  * ``simulation_fun(data)``                python/ofdm_utils/wofdm_simulation.py:20-73
  * the per-window-file driver loop         matlab/main_BER_calculation.m:64-201

All arithmetic of the frame pipeline runs in the HIP kernels; this module only prepares
constants, shards frame ranges and divides integer counters.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from . import variants as V


def _diag_or_vec(w, n):
    """The reference passes windows as dense diagonal matrices; accept both."""
    w = np.asarray(w)
    if w.ndim == 2 and w.shape[0] == w.shape[1]:
        w = np.diag(w)
    w = np.asarray(w, dtype=np.float64).reshape(-1)
    if w.size != n:
        raise ValueError("window has %d samples, expected %d" % (w.size, n))
    return w


def make_cfg(st, bits_per_sc, syms_per_frame, n_taps, n_channels, n_snr, n_window_pairs,
             noise_before_truncate=True, seed=0, frames_per_cell=0, frame_offset=0):
    """Build a ``wofdm_cfg`` from a :class:`variants.Structure`."""
    return _lib.Cfg(st.n_fft, bits_per_sc, syms_per_frame, st.cp, st.cs, st.tail_tx, st.tail_rx,
                    st.prefix_rm, st.circ_shift, n_taps, n_channels, n_snr, n_window_pairs,
                    1 if noise_before_truncate else 0, frames_per_cell, frame_offset, seed)


class Plan:
    """Constants resident in HBM + kernel selection (``wofdm_plan``).

    w_tx [pairs][P], w_rx [pairs][N+delta], h [n_channels][L] complex, snr_db [n_snr].
    """

    def __init__(self, cfg, w_tx, w_rx, h, snr_db, device=0):
        self.lib = _lib.load()
        self.cfg = cfg
        self.device = int(device)
        self._w_tx = _lib.f32(np.atleast_2d(w_tx), (cfg.n_window_pairs, cfg.sym_len))
        self._w_rx = _lib.f32(np.atleast_2d(w_rx), (cfg.n_window_pairs, cfg.n_fft + cfg.tail_rx))
        self._h = _lib.c64_as_f32(np.atleast_2d(h), (cfg.n_channels, cfg.n_taps))
        self._snr = _lib.f32(np.atleast_1d(snr_db), (cfg.n_snr,))
        self._h_plan = C.c_void_p()
        _lib.check(self.lib.wofdm_plan_create(
            C.byref(self._h_plan), C.byref(cfg), self.device, self._w_tx.ctypes.data,
            self._w_rx.ctypes.data, self._h.ctypes.data, self._snr.ctypes.data))
        self.noise_len = self.lib.wofdm_noise_len(C.byref(cfg))
        self.n_active = cfg.n_fft

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h_plan", None) and self._h_plan.value:
            self.lib.wofdm_plan_destroy(self._h_plan)
            self._h_plan = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- facts ------------------------------------------------------------------------------
    @property
    def counts_shape(self):
        c = self.cfg
        return (c.n_window_pairs, c.n_snr, c.n_channels, 4)

    def info(self):
        a = (C.c_int32 * 5)()
        _lib.check(self.lib.wofdm_plan_info(self._h_plan, a))
        return dict(waves_per_workgroup=a[0], lds_bytes=a[1], workgroups=a[2],
                    workgroups_per_cu=a[3], compute_units=a[4])

    def kernel_id(self):
        """(layout id, variant) of the kernel instantiation in use (``wofdm_plan_kernel_id``)."""
        a = (C.c_int32 * 2)()
        _lib.check(self.lib.wofdm_plan_kernel_id(self._h_plan, a))
        return int(a[0]), int(a[1])

    def set_allocation(self, active):
        """Subcarrier allocation (``wofdm_plan_set_allocation``): ``active`` [N] truthy = bin
        carries data; None = every bin (main_channel_mask.m:387-390, 367-369)."""
        if active is None:
            _lib.check(self.lib.wofdm_plan_set_allocation(self._h_plan, None))
            self.n_active = self.cfg.n_fft
            return
        a = np.ascontiguousarray(np.asarray(active).reshape(-1) != 0, dtype=np.uint8)
        if a.shape != (self.cfg.n_fft,):
            raise ValueError("allocation must have n_fft = %d entries" % self.cfg.n_fft)
        _lib.check(self.lib.wofdm_plan_set_allocation(self._h_plan, a.ctypes.data))
        self.n_active = int(a.sum())

    def set_tx_mask(self, mask):
        """Per-symbol spectral Tx mask (``wofdm_plan_set_tx_mask``): ``mask`` [2P-1] DFT-domain
        gains in natural bin order (main_channel_mask.m:398-417); None removes it."""
        if mask is None:
            _lib.check(self.lib.wofdm_plan_set_tx_mask(self._h_plan, None))
            return
        m = _lib.f32(np.asarray(mask).reshape(-1), (2 * self.cfg.sym_len - 1,))
        _lib.check(self.lib.wofdm_plan_set_tx_mask(self._h_plan, m.ctypes.data))

    def set_option(self, name, value):
        """Kernel choice (``wofdm_plan_set_option``): ``fir_valu`` 0/1, ``dft_valu`` 0/1 (the transforms on the vector pipe:
        the kernels for a GPU that is shared with other work, include/wofdm.h), ``max_spw`` 0/1/2/4, ``txmask_direct`` 0/1.
        Same results, other kernels of the family (A/B measurements, tests)."""
        _lib.check(self.lib.wofdm_plan_set_option(self._h_plan, _lib.OPTIONS[name], int(value)))

    def new_counts(self):
        """Zeroed device counter tensor [pairs][n_snr][n_channels][4] (int64 bit pattern of the
        kernel's uint64 counters; torch is only the allocator here)."""
        import torch
        return torch.zeros(self.counts_shape, dtype=torch.int64, device="cuda:%d" % self.device)

    # -- launches ---------------------------------------------------------------------------
    def launch(self, frame_offset, frames_per_cell, counts, stream=None):
        """Asynchronous generate-mode launch; ``counts`` is a device tensor (or a raw device
        address) accumulated into."""
        _lib.check(self.lib.wofdm_plan_launch(
            self._h_plan, int(frame_offset), int(frames_per_cell), _dev_ptr(counts),
            _stream_ptr(stream, self.device)))

    def launch_timed(self, frame_offset, frames_per_cell, counts, stream=None):
        """Launch, wait, return the kernel's duration in ms (HIP events on ``stream``)."""
        ms = C.c_float()
        _lib.check(self.lib.wofdm_plan_launch_timed(
            self._h_plan, int(frame_offset), int(frames_per_cell), _dev_ptr(counts),
            _stream_ptr(stream, self.device), C.byref(ms)))
        return float(ms.value)

    def launch_injected(self, frames_per_cell, labels, unit_noise, counts, stream=None):
        """labels: uint8 device tensor [cells][frames][S][N]; unit_noise: float32 device tensor
        [cells][frames][noise_len][2]."""
        _lib.check(self.lib.wofdm_plan_launch_injected(
            self._h_plan, int(frames_per_cell), _dev_ptr(labels), _dev_ptr(unit_noise),
            _dev_ptr(counts), _stream_ptr(stream, self.device)))

    def run(self, frame_offset, frames_per_cell):
        """Synchronous convenience: returns host counts (uint64 ndarray)."""
        import torch
        counts = self.new_counts()
        self.launch(frame_offset, frames_per_cell, counts)
        torch.cuda.synchronize(self.device)
        self.status()
        return counts.cpu().numpy().view(np.uint64)

    def status(self):
        """Raise if a finished kernel of this plan reported a problem (``wofdm_plan_status``);
        call after synchronising."""
        _lib.check(self.lib.wofdm_plan_status(self._h_plan))

    def dump_frame(self, cell, frame, labels=None, unit_noise=None):
        """Run one frame and return (counts[4], stages dict) -- parity instrumentation."""
        c = self.cfg
        S, N, B, T, L = c.syms_per_frame, c.n_fft, c.stride, c.frame_len, c.n_taps
        st = dict(labels_tx=np.zeros((S, N), np.uint8), X=np.zeros((S, N), np.complex64),
                  tx=np.zeros(T, np.complex64), conv=np.zeros(T + L - 1, np.complex64),
                  rx=np.zeros(S * B, np.complex64), Y=np.zeros((S, N), np.complex64),
                  Xhat=np.zeros((S - 1, N), np.complex64),
                  labels_rx=np.zeros((S - 1, N), np.uint8), gain=np.zeros(1, np.float32),
                  unit_noise=np.zeros(self.noise_len, np.complex64))
        d = _lib.Dump(*[st[f].ctypes.data for f, _ in _lib.Dump._fields_])
        counts = np.zeros(4, dtype=np.uint64)
        lab = noise = None
        if labels is not None:
            lab = np.ascontiguousarray(labels, dtype=np.uint8)
            noise = _lib.c64_as_f32(unit_noise, (self.noise_len,))
            assert lab.shape == (S, N)
        _lib.check(self.lib.wofdm_plan_dump_frame(
            self._h_plan, int(cell), int(frame), lab.ctypes.data if lab is not None else None,
            noise.ctypes.data if noise is not None else None, counts.ctypes.data, C.byref(d)))
        return counts, st


def _dev_ptr(t):
    if hasattr(t, "data_ptr"):
        if not t.is_cuda:
            raise ValueError("expected a device tensor")
        return t.data_ptr()
    return int(t)


def _stream_ptr(stream, device):
    if stream is None:
        try:
            import torch
            if torch.cuda.is_available():
                return torch.cuda.current_stream(device).cuda_stream
        except ImportError:
            pass
        return None
    return getattr(stream, "cuda_stream", stream)


def run_counts(cfg, w_tx, w_rx, h, snr_db, device=0):
    """One-shot ``wofdm_run``: host arrays in, host counters [pairs][n_snr][n_ch][4] out."""
    lib = _lib.load()
    w_tx = _lib.f32(np.atleast_2d(w_tx), (cfg.n_window_pairs, cfg.sym_len))
    w_rx = _lib.f32(np.atleast_2d(w_rx), (cfg.n_window_pairs, cfg.n_fft + cfg.tail_rx))
    h = _lib.c64_as_f32(np.atleast_2d(h), (cfg.n_channels, cfg.n_taps))
    snr = _lib.f32(np.atleast_1d(snr_db), (cfg.n_snr,))
    counts = np.zeros((cfg.n_window_pairs, cfg.n_snr, cfg.n_channels, 4), dtype=np.uint64)
    _lib.check(lib.wofdm_run(C.byref(cfg), int(device), w_tx.ctypes.data, w_rx.ctypes.data,
                             h.ctypes.data, snr.ctypes.data, counts.ctypes.data))
    return counts


def run_counts_injected(cfg, w_tx, w_rx, h, snr_db, labels, unit_noise, device=0):
    """One-shot ``wofdm_run_injected``.  labels uint8 [cells][frames][S][N], unit_noise
    complex [cells][frames][noise_len]."""
    lib = _lib.load()
    w_tx = _lib.f32(np.atleast_2d(w_tx), (cfg.n_window_pairs, cfg.sym_len))
    w_rx = _lib.f32(np.atleast_2d(w_rx), (cfg.n_window_pairs, cfg.n_fft + cfg.tail_rx))
    h = _lib.c64_as_f32(np.atleast_2d(h), (cfg.n_channels, cfg.n_taps))
    snr = _lib.f32(np.atleast_1d(snr_db), (cfg.n_snr,))
    nl = lib.wofdm_noise_len(C.byref(cfg))
    F = int(cfg.frames_per_cell)
    labels = np.ascontiguousarray(labels, dtype=np.uint8)
    if labels.shape != (cfg.n_cells, F, cfg.syms_per_frame, cfg.n_fft):
        raise ValueError("labels shape %s" % (labels.shape,))
    noise = _lib.c64_as_f32(unit_noise, (cfg.n_cells, F, nl))
    counts = np.zeros((cfg.n_window_pairs, cfg.n_snr, cfg.n_channels, 4), dtype=np.uint64)
    _lib.check(lib.wofdm_run_injected(C.byref(cfg), int(device), w_tx.ctypes.data,
                                      w_rx.ctypes.data, h.ctypes.data, snr.ctypes.data,
                                      labels.ctypes.data, noise.ctypes.data, counts.ctypes.data))
    return counts


def error_rates(counts):
    """counts[..., 4] -> (BER, SER) arrays."""
    c = np.asarray(counts).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        return c[..., 0] / c[..., 1], c[..., 2] / c[..., 3]


# ---------------------------------------------------------------------------------------------
# MATLAB operator: ber = run_simulation(...)            main_BER_calculation.m:230-232
# ---------------------------------------------------------------------------------------------
def run_simulation(ensemble, symbolsPerTx, bitsPerSubcarrier, numSubcar, cpLength, csLength,
                   windowTx, channel, snr, tailTx, tailRx, windowRx, prefixRemovalLength,
                   circularShiftLength, seed=0, device=0, frame_offset=0):
    """Same arguments and meaning as the reference's local function; returns the BER.

    Difference: the reference returns the BER of the *last* of the
    ``ensemble`` frames; this accumulates integer error counts over all of them.
    """
    st = V.Structure("custom", int(numSubcar), int(cpLength), int(tailTx), int(tailRx),
                     int(csLength), int(prefixRemovalLength), int(circularShiftLength))
    channel = np.asarray(channel).reshape(-1)
    cfg = make_cfg(st, int(bitsPerSubcarrier), int(symbolsPerTx), channel.size, 1, 1, 1,
                   noise_before_truncate=True, seed=seed, frames_per_cell=int(ensemble),
                   frame_offset=frame_offset)
    counts = run_counts(cfg, _diag_or_vec(windowTx, st.sym_len), _diag_or_vec(windowRx, st.rx_win_len),
                        channel, [snr], device=device)
    return float(counts[0, 0, 0, 0]) / float(counts[0, 0, 0, 1])


# ---------------------------------------------------------------------------------------------
# Python operator: wOFDMSystem                           wofdm_simulation.py:76-481
# ---------------------------------------------------------------------------------------------
class wOFDMSystem:
    """Drop-in for the reference class of the same name (same constructor and
    ``run_simulation`` signature, same output files ``<folder>/ser/{opt,rc}_<sys>_<cp>.npy`` or
    ``CP_<cp>.npy``), with the frame loop on the GPU.

    As in the reference the constellation is 16-QAM, the figure is the symbol error rate,
    noise is added after truncation, and the optimised and the raised-cosine windows are
    both simulated (with independent data here; the reference reuses the symbols, quirk Q7).
    """

    bits_per_sc = 4  # 16-QAM is hard-coded in the reference (wofdm_simulation.py:179-182)

    def __init__(self, system_design, dft_len, cp_len, tail_tx, tail_rx, folder_path,
                 device=0, seed=0):
        self.name = system_design
        self.dft_len, self.cp_len = int(dft_len), int(cp_len)
        self.tail_tx, self.tail_rx = int(tail_tx), int(tail_rx)
        self.folder_path = folder_path
        self.device, self.seed = device, seed
        self.structure = V.make_structure(system_design, self.dft_len, self.cp_len,
                                          self.tail_tx, self.tail_rx)
        self.cs_len = self.structure.cs
        self.rm_len = self.structure.prefix_rm
        self.shift_len = self.structure.circ_shift

    def simulate(self, channel_models, window_tx, window_rx, ensemble, snr_arr, no_symbols):
        """Returns counts[pairs][n_snr][n_ch][4]; pair 0 = given windows, pair 1 = RC windows
        (only pair 0 for plain CP-OFDM)."""
        st = self.structure
        ch = np.asarray(channel_models)
        if ch.ndim == 1:
            ch = ch.reshape(-1, 1)
        h = ch.T                                   # [taps, n_ch] -> [n_ch, taps]
        if self.name == "CP":
            w_tx = np.ones((1, st.sym_len))
            w_rx = np.ones((1, st.rx_win_len))
        else:
            w_tx = np.stack([_diag_or_vec(window_tx, st.sym_len), V.tx_rc_window(st)])
            w_rx = np.stack([_diag_or_vec(window_rx, st.rx_win_len), V.rx_rc_window(st)])
        cfg = make_cfg(st, self.bits_per_sc, int(no_symbols), h.shape[1], h.shape[0],
                       len(snr_arr), w_tx.shape[0], noise_before_truncate=False, seed=self.seed,
                       frames_per_cell=int(ensemble))
        return run_counts(cfg, w_tx, w_rx, h, np.asarray(snr_arr, dtype=np.float64),
                          device=self.device)

    def run_simulation(self, channel_models, window_tx, window_rx, ensemble, snr_arr,
                       no_symbols):
        counts = self.simulate(channel_models, window_tx, window_rx, ensemble, snr_arr, no_symbols)
        # mean over channels of per-channel SER (wofdm_simulation.py:237-240); every cell has
        # the same number of symbols, so this equals the ratio of sums
        ser = counts[..., 2].sum(axis=2) / counts[..., 3].sum(axis=2)
        path_to_ser = os.path.join(self.folder_path, "ser")
        os.makedirs(path_to_ser, exist_ok=True)
        if self.name == "CP":
            np.save(os.path.join(path_to_ser, "CP_%d.npy" % self.cp_len), ser[0])
            return ser[0]
        np.save(os.path.join(path_to_ser, "opt_%s_%d.npy" % (self.name, self.cp_len)), ser[0])
        np.save(os.path.join(path_to_ser, "rc_%s_%d.npy" % (self.name, self.cp_len)), ser[1])
        return ser[0], ser[1]


def simulation_fun(data):
    """Work-item entry point of the reference's pool (wofdm_simulation.py:20-73): ``data`` =
    (system_design, dft_len, cp_len, tail_tx, tail_rx, channel_path, window_path, ensemble,
    snr_arr, no_symbols, folder_path)."""
    system_design, dft_len, cp_len, tail_tx, tail_rx = data[0:5]
    channel_path, window_path, ensemble, snr_arr, no_symbols, folder_path = data[5:]
    channel_models = np.load(channel_path)
    model = wOFDMSystem(system_design, dft_len, cp_len, tail_tx, tail_rx, folder_path)
    st = model.structure
    if system_design == "CP":
        x_tx, x_rx = np.ones(1), np.ones(1)
    else:
        vec = np.load(os.path.join(window_path, "%s_%d.npy" % (system_design, cp_len)))
        x_tx, x_rx = V.split_tail_file(st, vec)
    win_tx = V.expand_tx_window(st, x_tx) if st.tail_tx else np.full(st.sym_len, float(x_tx[0]))
    win_rx = V.expand_rx_window(st, x_rx) if st.tail_rx else np.full(st.rx_win_len, float(x_rx[0]))
    return model.run_simulation(channel_models, win_tx, win_rx, ensemble, snr_arr, no_symbols)


# ---------------------------------------------------------------------------------------------
# MATLAB driver loop for one window file                  main_BER_calculation.m:64-201
# ---------------------------------------------------------------------------------------------
def ber_for_window_file(type_ofdm, cp_length, windows, channels, snr_values, num_subcar=256,
                        bits_per_subcar=4, symbols_per_tx=16, ensemble=100, tail_tx=None,
                        tail_rx=None, seed=0, device=0, frame_range=None):
    """BER curves of one ``optimal_win_<type>_VehA200_<cp>CP.mat`` work item.

    windows: dict with the MATLAB variable names of the window file (``optimizedWindow`` or
    ``optimizedWindowCase{A,B}Step{1,2,3}``) holding diagonal matrices or vectors.
    channels: [n_realisations][taps] (rows = realisations, quirk Q2).
    Returns (results, counts): results maps the reference's output variable names
    (``berSNR``, ``berRCSNR``, ``berSNRStep1A`` ...) to [n_snr] arrays -- the mean over the
    channel realisations, as lines 85-86 / 186-192 compute it.
    frame_range = (offset, frames) restricts the ensemble (multi-GPU sharding).
    """
    st = V.make_structure(type_ofdm, num_subcar, cp_length, tail_tx, tail_rx)
    rc = {"tx": V.tx_rc_window(st), "rx": V.rx_rc_window(st)}
    plan = V.matlab_pair_plan(type_ofdm)

    def pick(key, side):
        return rc[side] if key == "rc" else _diag_or_vec(
            windows[key], st.sym_len if side == "tx" else st.rx_win_len)

    w_tx = np.stack([pick(k[0], "tx") for _, k in plan])
    w_rx = np.stack([pick(k[1], "rx") for _, k in plan])
    channels = np.atleast_2d(np.asarray(channels))
    snr_values = np.asarray(snr_values, dtype=np.float64).reshape(-1)
    off, frames = (0, ensemble) if frame_range is None else frame_range
    cfg = make_cfg(st, bits_per_subcar, symbols_per_tx, channels.shape[1], channels.shape[0],
                   snr_values.size, len(plan), noise_before_truncate=True, seed=seed,
                   frames_per_cell=frames, frame_offset=off)
    counts = run_counts(cfg, w_tx, w_rx, channels, snr_values, device=device)
    return results_from_counts(type_ofdm, counts), counts


def results_from_counts(type_ofdm, counts):
    """counts[pairs][n_snr][n_ch][4] -> {MATLAB result variable: BER[n_snr]}."""
    plan = V.matlab_pair_plan(type_ofdm)
    ber = counts[..., 0].sum(axis=2) / np.maximum(counts[..., 1].sum(axis=2), 1)
    names = {"opt": "berSNR", "rc": "berRCSNR", "1A": "berSNRStep1A", "2A": "berSNRStep2A",
             "3A": "berSNRStep3A", "1B": "berSNRStep1B", "2B": "berSNRStep2B",
             "3B": "berSNRStep3B"}
    return {names[tag]: ber[i] for i, (tag, _) in enumerate(plan)}


def save_ber_results(folder, type_ofdm, cp_length, results):
    """Write ``optimized_ber_<type>_<cp>CP.mat`` / ``rc_ber_<type>_<cp>CP.mat`` with the
    reference's variable names (main_BER_calculation.m:88-90, 194-198, 205-227)."""
    from scipy.io import savemat
    os.makedirs(folder, exist_ok=True)
    base = "ber_%s_%dCP" % (type_ofdm, cp_length)
    opt = {k: np.asarray(v).reshape(-1, 1) for k, v in results.items() if k != "berRCSNR"}
    savemat(os.path.join(folder, "optimized_" + base + ".mat"), opt)
    savemat(os.path.join(folder, "rc_" + base + ".mat"),
            {"berRCSNR": np.asarray(results["berRCSNR"]).reshape(-1, 1)})
