"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, on the same seeded
inputs.  Integer work (Philox words, labels, bit totals) must be bit-exact; floating-point
stages are fp32 on the GPU vs fp64 in the oracle, tolerance written per check."""
import os

import numpy as np
import pytest

import wofdm_amd as W
from oracle import oracle as O

pytestmark = pytest.mark.gpu

SYSTEMS = ["wtx", "wrx", "WOLA", "CPW", "CPwtx", "CPwrx", "CP"]
STAGE_RTOL = 2e-5          # fp32 chain of an N<=1024 FFT + 21-tap FIR vs fp64, relative to max|x|
# wofdm_simulation.py:179-182
SYM16 = np.array((-3-3j, -3-1j, -3+1j, -3+3j, -1-3j, -1-1j, -1+1j, -1+3j, 1-3j, 1-1j,
                  1+1j, 1+3j, 3-3j, 3-1j, 3+1j, 3+3j))


def _osys(st, k, S, n_taps, matlab):
    return O.make_sys(st.n_fft, k, S, st.cp, st.cs, st.tail_tx, st.tail_rx, st.prefix_rm,
                      st.circ_shift, n_taps, 1 if matlab else 0)


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def test_device_present_and_philox_kat():
    lib = W._lib.load()
    assert lib.wofdm_device_count() >= 1, lib.wofdm_last_error()
    import ctypes as C
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        c = np.array(ctr, dtype=np.uint32); k = np.array(key, dtype=np.uint32)
        out = np.zeros(4, dtype=np.uint32)
        W._lib.check(lib.wofdm_philox_kat(0, c.ctypes.data, k.ctypes.data, out.ctypes.data))
        assert tuple(int(x) for x in out) == want


def _expected_layout(st, n_fft, S):
    """Layout id wofdm_plan_create picks for a plain plan (wofdm_spw in csrc/wofdm_kernel.h)."""
    B = st.stride
    if n_fft <= 128 and S % (1024 // n_fft) == 0 and B >= n_fft:
        if (1024 // n_fft) * B <= 128 * 10:
            return 13
        if (1024 // n_fft) * B <= 128 * 11:
            return 14
    if n_fft <= 128 and B >= n_fft:
        # layout 16: a run-time number of symbols per wave (wofdm_small_spwr): the frame spread evenly over at most four waves
        for waves in range(1, 5):
            spwr = (-(-S // waves) + 1) & ~1
            if spwr <= 1024 // n_fft and spwr * B <= 128 * 10 and (waves - 1) * spwr < S:
                return 16
    if n_fft == 256 and S % 4 == 0 and B >= n_fft:
        if 4 * B <= 128 * 9:
            return 10
        if 4 * B <= 128 * 10:
            return 11
    if n_fft == 256 and S % 4 == 0:
        if 4 * B <= 64 * 18:
            return 4
        if 4 * B <= 64 * 20:
            return 5
    if n_fft >= 512 and B >= n_fft and B <= 128 * (9 if n_fft == 1024 else 5):      # (every stride, odd ones since round 4)
        return 12
    return 2 if (n_fft <= 256 and S % 2 == 0 and 2 * B <= 64 * (2 * (n_fft // 64) + 2)) else 1


CASES = [(s, 64, 16, 2, 1) for s in SYSTEMS] + [
    ("wtx", 256, 32, 4, 1), ("WOLA", 256, 32, 4, 1), ("CPW", 256, 32, 6, 0), ("CPwtx", 256, 10, 4, 1),
    ("wrx", 256, 12, 2, 0), ("WOLA", 128, 32, 4, 1), ("WOLA", 512, 32, 4, 1), ("CPW", 512, 20, 6, 1),
    ("WOLA", 1024, 32, 6, 1), ("CPwrx", 1024, 32, 2, 0),
    # cp+cs-tail_tx = 64: no idle lanes, the trailing samples take the kernel's slow path
    ("wtx", 256, 64, 4, 1), ("wtx", 64, 64, 2, 1), ("wtx", 512, 64, 4, 1)]


@pytest.mark.parametrize("system,n_fft,cp,k,matlab", CASES)
@pytest.mark.parametrize("inject", [False, True])
def test_one_frame_stage_by_stage(channels, system, n_fft, cp, k, matlab, inject):
    S, seed, frame = 16, 11, 123456789012
    st = W.make_structure(system, n_fft, cp)
    rs = np.random.RandomState(n_fft + cp + k)
    # non-trivial windows: flat level != 1 and non-RC tails, rounded to float32 on both sides
    xt = np.concatenate(([1.03], np.sort(rs.uniform(.05, .95, st.tail_tx))[::-1])) if st.tail_tx else [1.0]
    xr = np.concatenate(([0.97], np.sort(rs.uniform(.05, .45, st.tail_rx // 2))[::-1])) if st.tail_rx else [1.0]
    w_tx = (W.expand_tx_window(st, xt) if st.tail_tx else np.ones(st.sym_len)).astype(np.float32)
    w_rx = (W.expand_rx_window(st, xr) if st.tail_rx else np.ones(st.rx_win_len)).astype(np.float32)
    h = channels[4:7].astype(np.complex64)
    snrs = np.array([8.0, 22.0], dtype=np.float32)
    cfg = W.make_cfg(st, k, S, 21, 3, 2, 1, noise_before_truncate=bool(matlab), seed=seed)
    osys = _osys(st, k, S, 21, matlab)
    cell = 4                                           # snr index 1, channel 1
    lab = O.gen_labels(osys, seed, cell, frame)
    noise = O.gen_noise(osys, seed, cell, frame)
    oc, od = O.frame(osys, w_tx.astype(np.float64), w_rx.astype(np.float64),
                     h[1].astype(np.complex128), float(snrs[1]), lab, noise, dump=True)
    with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
        kid = plan.kernel_id()
        if inject:
            gc, gd = plan.dump_frame(cell, frame, lab, noise.astype(np.complex64))
        else:
            gc, gd = plan.dump_frame(cell, frame)
    # the instrumented kernel is the instantiation of the SAME layout the plan launches for this geometry (same Tx
    # write, FIR, Rx window / fold and transforms as the kernel bench.py times; only the stage stores and the
    # barriers between the phases are added)
    assert kid[1] == 0 and kid[0] == _expected_layout(st, n_fft, S), kid
    # integer work: bit-exact
    assert np.array_equal(gd["labels_tx"], lab)
    assert int(gc[1]) == int(oc[1]) == (S - 1) * n_fft * k and int(gc[3]) == int(oc[3])
    # floating point stages
    nconv = od["conv"].size if matlab else S * st.stride
    assert _rel(gd["unit_noise"], noise) < 1e-5
    assert _rel(gd["X"], od["X"]) < 1e-6
    assert _rel(gd["tx"], od["tx"]) < STAGE_RTOL
    assert _rel(gd["conv"][:nconv], od["conv"][:nconv]) < STAGE_RTOL
    assert abs(float(gd["gain"][0]) / float(od["gain"][0]) - 1) < 1e-5
    assert _rel(gd["rx"], od["rx"]) < STAGE_RTOL
    assert _rel(gd["Y"], od["Y"]) < STAGE_RTOL           # (the circular shift's phase ramp put back by the ABI)
    # the equalised symbols, where the channel is not in a null (there the division amplifies fp32 rounding)
    Hest = od["Y"][0] / od["X"][0]
    ok = np.abs(Hest) > 0.1 * np.abs(Hest).max()
    assert ok.sum() > n_fft // 2
    assert np.abs(gd["Xhat"][:, ok] - od["Xhat"][:, ok]).max() < 20 * STAGE_RTOL * np.abs(od["Xhat"][:, ok]).max()
    # decisions: identical except where the equalised point sits on a decision boundary
    mism = gd["labels_rx"] != od["labels_rx"]
    if mism.any():
        a = np.sqrt(2 * (2 ** k - 1) / 3)
        z = od["Xhat"][mism] * a
        dist = np.minimum(np.abs(((z.real + 1) % 2) - 1), np.abs(((z.imag + 1) % 2) - 1))
        # boundaries of the slicer sit at even integers of the scaled grid
        edge = np.minimum(np.abs((z.real / 2) - np.round(z.real / 2)) * 2,
                          np.abs((z.imag / 2) - np.round(z.imag / 2)) * 2)
        assert (edge < 1e-3).all(), (int(mism.sum()), edge.max(), dist.min())
    assert abs(int(gc[0]) - int(oc[0])) <= 2 * int(mism.sum())
    assert abs(int(gc[2]) - int(oc[2])) <= int(mism.sum())


@pytest.mark.parametrize("system,n_fft,cp,k", [("wtx", 256, 32, 4), ("wtx", 256, 48, 6), ("WOLA", 512, 32, 4), ("WOLA", 1024, 32, 6)])
def test_fir_precision_matrix_pipe_vs_valu(channels, system, n_fft, cp, k):
    """conv(channel, tx) (main_BER_calculation.m:260) of one injected frame against the fp64 oracle: the matrix-pipe form
    (three f16 x f16 MFMA terms of hi/lo halves, fp32 accumulation: 22-bit operands, h_lo x_lo dropped) and the fp32
    VALU form of the same kernels (plan option fir_valu).  Error over the frame's rms; bounds 2e-6 (max) and 3e-7 (rms, FIR alone) for either -- measured 7e-7 / 1.3e-7 (matrix pipe) against
    7e-7 / 1.1e-7 (VALU): both sit at the fp32 rounding floor of the stage dumps."""
    S, seed, frame, cell = 16, 3, 77, 1
    st = W.make_structure(system, n_fft, cp)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    h = channels[20:22].astype(np.complex64)
    snrs = np.array([18.0], np.float32)
    cfg = W.make_cfg(st, k, S, 21, 2, 1, 1, seed=seed)
    osys = _osys(st, k, S, 21, True)
    lab, noise = O.gen_labels(osys, seed, cell, frame), O.gen_noise(osys, seed, cell, frame)
    oc, od = O.frame(osys, w_tx.astype(np.float64), w_rx.astype(np.float64), h[1].astype(np.complex128), float(snrs[0]),
                     lab, noise, dump=True)
    rms = np.sqrt(np.mean(np.abs(od["conv"]) ** 2))
    err = {}
    for name, valu in (("matrix pipe", 0), ("valu", 1)):
        with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
            plan.set_option("fir_valu", valu)
            layout = plan.kernel_id()[0]
            assert (layout in (6, 7, 8, 10, 11, 12, 13, 14, 16)) == (valu == 0), layout
            gc, gd = plan.dump_frame(cell, frame, lab, noise.astype(np.complex64))
        e = np.abs(gd["conv"] - od["conv"])
        # ... and the FIR alone: against the fp64 convolution of the kernel's OWN transmitted frame (the chain's error
        # above is mostly the fp32 IFFT in front of it)
        own = np.convolve(h[1].astype(np.complex128), gd["tx"].astype(np.complex128))
        f = np.abs(gd["conv"] - own)
        err[name] = (layout, float(e.max() / rms), float(np.sqrt(np.mean(e ** 2)) / rms), float(f.max() / rms),
                     float(np.sqrt(np.mean(f ** 2)) / rms))
    print("\nconv vs fp64, %s N=%d cp=%d: " % (system, n_fft, cp)
          + "; ".join("%s (layout %d): whole chain max %.2e rms %.2e, FIR alone max %.2e rms %.2e of the frame's rms"
                      % ((n,) + err[n]) for n in err))
    assert err["matrix pipe"][1] <= 2e-6 and err["valu"][1] <= 2e-6, err
    assert err["matrix pipe"][3] <= 2e-6 and err["valu"][3] <= 2e-6, err
    assert err["matrix pipe"][4] <= 3e-7 and err["valu"][4] <= 3e-7, err        # rms: both at the fp32 rounding floor of the dumps


@pytest.mark.parametrize("system,n_fft,cp,k", [("wtx", 256, 32, 4), ("CPW", 256, 32, 6), ("WOLA", 512, 32, 4), ("WOLA", 1024, 32, 6),
                                               ("wtx", 64, 16, 2), ("WOLA", 128, 32, 4)])
def test_dft_precision_matrix_pipe_vs_valu(channels, system, n_fft, cp, k):
    """The transforms (dftmtx(N)'/N and dftmtx(N), main_BER_calculation.m:370, 306) of one injected frame against the fp64 oracle:
    on the matrix pipe (layouts 10 ... 14: 16 x 16 DFT stages as three-term split-f16 MFMA products, fp32 accumulation
    and twiddles) and on the VALU (layouts 2 / 6 / 7 / 8, plan option dft_valu).  `tx` isolates the inverse transform (its input
    is exact), `Y` against the fp64 DFT of the kernel's OWN received blocks the forward one.  Error over the stage's rms:
    bounds 2e-6 (max) and 3e-7 (rms) for either form -- both sit at the fp32 rounding floor."""
    S, seed, frame, cell = 16, 5, 99, 1
    st = W.make_structure(system, n_fft, cp)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    h = channels[20:22].astype(np.complex64)
    snrs = np.array([18.0], np.float32)
    cfg = W.make_cfg(st, k, S, 21, 2, 1, 1, seed=seed)
    osys = _osys(st, k, S, 21, True)
    lab, noise = O.gen_labels(osys, seed, cell, frame), O.gen_noise(osys, seed, cell, frame)
    oc, od = O.frame(osys, w_tx.astype(np.float64), w_rx.astype(np.float64), h[1].astype(np.complex128), float(snrs[0]),
                     lab, noise, dump=True)
    err = {}
    for name, valu in (("matrix pipe", 0), ("valu", 1)):
        with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
            plan.set_option("dft_valu", valu)
            layout = plan.kernel_id()[0]
            assert (layout in (10, 11, 12, 13, 14, 16)) == (valu == 0), layout
            gc, gd = plan.dump_frame(cell, frame, lab, noise.astype(np.complex64))
        e_tx = np.abs(gd["tx"] - od["tx"]) / np.sqrt(np.mean(np.abs(od["tx"]) ** 2))
        # forward transform alone: the oracle's Rx stage (window, fold, shift, DFT) applied in fp64 to the kernel's own rx
        rx = gd["rx"][:S * st.stride].astype(np.complex128).reshape(S, st.stride)
        m = np.arange(n_fft + st.tail_rx)
        z = np.zeros((S, n_fft), np.complex128)
        np.add.at(z, (slice(None), m % n_fft), rx[:, st.prefix_rm + m] * w_rx.astype(np.float64)[m])
        z = np.roll(z, -(st.circ_shift + st.tail_rx // 2), axis=1)
        own = np.fft.fft(z, axis=1)
        e_y = np.abs(gd["Y"] - own) / np.sqrt(np.mean(np.abs(own) ** 2))
        err[name] = (layout, float(e_tx.max()), float(np.sqrt(np.mean(e_tx ** 2))), float(e_y.max()), float(np.sqrt(np.mean(e_y ** 2))))
    print("\ntransforms vs fp64, %s N=%d cp=%d: " % (system, n_fft, cp)
          + "; ".join("%s (layout %d): IDFT max %.2e rms %.2e, DFT alone max %.2e rms %.2e of the stage's rms" % ((n,) + err[n])
                      for n in err))
    for n in err:
        assert err[n][1] <= 2e-6 and err[n][3] <= 2e-6, err
        assert err[n][2] <= 3e-7 and err[n][4] <= 3e-7, err


@pytest.mark.parametrize("system,n_fft,cp,k,opts", [("wtx", 256, 32, 4, {}), ("CPW", 256, 32, 6, {}), ("WOLA", 512, 32, 4, {}),
                                                     ("WOLA", 1024, 32, 6, {}), ("WOLA", 128, 16, 4, {}),
                                                     ("wtx", 256, 32, 4, {"fir_valu": 1}), ("WOLA", 512, 32, 2, {"fir_valu": 1}),
                                                     ("wtx", 256, 32, 4, {"dft_valu": 1}), ("WOLA", 1024, 32, 6, {"dft_valu": 1}),
                                                     ("wtx", 64, 16, 2, {}), ("WOLA", 128, 32, 6, {}),
                                                     # layout 16: two waves of eight symbols (N = 64 at CP 32), three of six (N = 128, odd stride)
                                                     ("wtx", 64, 32, 2, {}), ("CPW", 64, 32, 4, {}), ("wrx", 128, 56, 6, {})])
def test_production_and_instrumented_kernels_count_the_same(channels, system, n_fft, cp, k, opts):
    """The stage-by-stage parity runs the instrumented instantiations (stage stores, barriers between the phases); the same
    injected frames through the PRODUCTION instantiation of the same layout must give the same error counters (the two
    differ in nothing that rounds: at most a decision on a slicer boundary per few frames)."""
    import torch
    S, seed, F = 16, 21, 24
    st = W.make_structure(system, n_fft, cp)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    h = channels[30:31].astype(np.complex64)
    snrs = np.array([6.0 + 3.0 * (k - 2), 16.0 + 3.0 * (k - 2)], np.float32)
    cfg = W.make_cfg(st, k, S, 21, 1, 2, 1, seed=seed)
    osys = _osys(st, k, S, 21, True)
    nl = O.noise_len(osys)
    labels = np.zeros((2, F, S, n_fft), np.uint8)
    noise = np.zeros((2, F, nl), np.complex64)
    for cell in range(2):
        for f in range(F):
            labels[cell, f] = O.gen_labels(osys, seed, cell, f)
            noise[cell, f] = O.gen_noise(osys, seed, cell, f)
    with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
        for key, val in opts.items():
            plan.set_option(key, val)
        counts = plan.new_counts()
        plan.launch_injected(F, torch.from_numpy(labels).cuda(), torch.from_numpy(noise.view(np.float32).reshape(2, F, nl, 2)).cuda(), counts)
        torch.cuda.synchronize()
        plan.status()
        prod = counts.cpu().numpy().view(np.uint64).reshape(2, 4).astype(np.int64)
        inst = np.zeros((2, 4), np.int64)
        for cell in range(2):
            for f in range(F):
                gc, _ = plan.dump_frame(cell, f, labels[cell, f], noise[cell, f])
                inst[cell] += gc.astype(np.int64)
    assert np.array_equal(prod[:, 1], inst[:, 1]) and np.array_equal(prod[:, 3], inst[:, 3])
    assert prod[:, 0].min() > 50, prod
    assert np.abs(prod - inst).max() <= 2, (prod, inst)


@pytest.mark.parametrize("system,n_fft,cp,k,opts,layout", [("WOLA", 512, 30, 4, {}, 12), ("CP", 512, 10, 2, {}, 12), ("WOLA", 1024, 30, 6, {}, 12),
                                                            ("WOLA", 512, 30, 4, {"dft_valu": 1}, 8), ("wtx", 512, 14, 6, {}, 12)])
@pytest.mark.parametrize("S", [16, 7])
def test_even_strides_that_are_not_multiples_of_four(channels, system, n_fft, cp, k, opts, layout, S):
    """One symbol per wave with the FIR on the matrix pipe (layouts 8, 12; 9 and 15 in test_gpu_channel_mask.py): a stride of 2 mod 4
    puts the end of a symbol inside a 16-byte operand row, which fir_load cuts word by word."""
    st = W.make_structure(system, n_fft, cp)
    assert st.stride % 4 == 2
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    snrs = np.array([8.0 + 3 * (k - 2), 20.0 + 3 * (k - 2)], np.float32)
    h = channels[5:7].astype(np.complex64)
    seed, off, F = 31, 3, 5
    cfg = W.make_cfg(st, k, S, 21, 2, 2, 1, seed=seed)
    osys = _osys(st, k, S, 21, True)
    want = O.run(osys, w_tx.astype(np.float64), w_rx.astype(np.float64), h.astype(np.complex128), snrs.astype(np.float64), seed, off, F)
    with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
        for key, val in opts.items():
            plan.set_option(key, val)
        assert plan.kernel_id() == (layout, 0)
        got = plan.run(off, F)
        gc, gd = plan.dump_frame(1, 4)
    assert np.array_equal(got[..., 1], want[..., 1]) and np.array_equal(got[..., 3], want[..., 3])
    assert np.abs(got[..., 0].astype(np.int64) - want[..., 0].astype(np.int64)).max() <= 2, (got[..., 0], want[..., 0])
    lab, noise = O.gen_labels(osys, seed, 1, 4), O.gen_noise(osys, seed, 1, 4)
    oc, od = O.frame(osys, w_tx.astype(np.float64), w_rx.astype(np.float64), h[1].astype(np.complex128), float(snrs[0]), lab, noise, dump=True)
    for stage in ("tx", "rx", "Y"):
        a, b = np.asarray(gd[stage]).ravel(), np.asarray(od[stage]).ravel()
        assert np.abs(a[:b.size] - b).max() / np.abs(b).max() < 2e-5, stage


@pytest.mark.parametrize("system,n_fft,cp,k", [("CPW", 512, 32, 4), ("wrx", 512, 22, 2), ("wrx", 1024, 22, 6), ("CPW", 1024, 10, 4)])
@pytest.mark.parametrize("S", [16, 7])
def test_odd_strides_with_one_symbol_per_wave(channels, system, n_fft, cp, k, S):
    """wrx / CPW / CPwrx have odd strides at every even CP the reference sweeps (delta = 10; matlab/window_optimization.m:39-47).  With
    one symbol per wave the rows of the odd symbols then start on an odd sample of the frame: their noise pairs take the second half
    of one Philox block and the first half of the next, the row's last pair holds ONE sample of the row, and the rows sit at 8- and
    4-byte alignments in LDS.  Since round 4 layout 12 takes them (layout 1 before): counters against the oracle on the same
    streams, and one frame stage by stage -- an odd symbol's noise, the FIR, the received rows, the transform."""
    st = W.make_structure(system, n_fft, cp)
    assert st.stride % 2 == 1
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    snrs = np.array([8.0 + 3 * (k - 2), 20.0 + 3 * (k - 2)], np.float32)
    h = channels[5:7].astype(np.complex64)
    seed, off, F = 37, 3, 6
    cfg = W.make_cfg(st, k, S, 21, 2, 2, 1, seed=seed)
    osys = _osys(st, k, S, 21, True)
    want = O.run(osys, w_tx.astype(np.float64), w_rx.astype(np.float64), h.astype(np.complex128), snrs.astype(np.float64), seed, off, F)
    with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
        assert plan.kernel_id() == (12, 0)
        got = plan.run(off, F)
        again = plan.run(off, F)
        gc, gd = plan.dump_frame(1, 4)
    assert np.array_equal(got, again)
    assert np.array_equal(got[..., 1], want[..., 1]) and np.array_equal(got[..., 3], want[..., 3])
    assert got[..., 0].max() > 20
    assert np.abs(got[..., 0].astype(np.int64) - want[..., 0].astype(np.int64)).max() <= 2, (got[..., 0], want[..., 0])
    lab, noise = O.gen_labels(osys, seed, 1, 4), O.gen_noise(osys, seed, 1, 4)
    oc, od = O.frame(osys, w_tx.astype(np.float64), w_rx.astype(np.float64), h[1].astype(np.complex128), float(snrs[0]), lab, noise, dump=True)
    assert _rel(gd["unit_noise"], noise) < 1e-5
    for stage in ("tx", "conv", "rx", "Y"):
        a, b = np.asarray(gd[stage]).ravel(), np.asarray(od[stage]).ravel()
        nn = min(a.size, b.size) if stage == "conv" else b.size
        assert np.abs(a[:nn] - b[:nn]).max() / np.abs(b).max() < 2e-5, stage
    assert np.array_equal(gc, oc) or np.abs(gc.astype(np.int64) - oc.astype(np.int64)).max() <= 1


@pytest.mark.parametrize("system,n_fft,cp,k,S,waves", [
    ("wtx", 64, 32, 2, 16, 2),       # VERDICT r3's case: 16 x 96 samples are 12 tiles -- two waves of eight symbols
    ("wtx", 64, 64, 2, 16, 2), ("WOLA", 64, 16, 4, 12, 1), ("wrx", 64, 16, 2, 11, 1), ("wtx", 64, 16, 6, 9, 1), ("CPW", 64, 32, 4, 16, 2),
    ("CPW", 128, 32, 4, 12, 2), ("WOLA", 128, 32, 6, 5, 1), ("CP", 128, 16, 2, 3, 1), ("WOLA", 128, 32, 4, 9, 2), ("wrx", 128, 56, 2, 16, 3)])
def test_small_dfts_with_partly_filled_waves(channels, system, n_fft, cp, k, S, waves):
    """N = 64, 128 where layouts 13 / 14 do not fit -- symbolsPerTx not a multiple of 16 / 8, strides beyond their ten / eleven tiles
    (N = 64 at CP 32) -- ran the round-1 layout 2 until round 4.  Layout 16 is layout 13 with a run-time number of symbols per wave and a
    partly filled last wave: counters against the oracle on the same streams, repeatability, and one frame stage by stage (the last
    symbol's fall tail and the zeros behind the frame sit INSIDE the last wave's rows there)."""
    st = W.make_structure(system, n_fft, cp)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    snrs = np.array([6.0 + 3 * (k - 2), 18.0 + 3 * (k - 2)], np.float32)
    h = channels[5:7].astype(np.complex64)
    seed, off, F = 41, 3, 9
    cfg = W.make_cfg(st, k, S, 21, 2, 2, 1, seed=seed)
    osys = _osys(st, k, S, 21, True)
    want = O.run(osys, w_tx.astype(np.float64), w_rx.astype(np.float64), h.astype(np.complex128), snrs.astype(np.float64), seed, off, F)
    with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
        assert plan.kernel_id() == (16, 0) and plan.info()["waves_per_workgroup"] == waves
        got = plan.run(off, F)
        again = plan.run(off, F)
        gc, gd = plan.dump_frame(1, 4)
        gc2, gd2 = plan.dump_frame(1, 5)                     # (a second frame through the same rows: nothing of the first is left behind)
    assert np.array_equal(got, again)
    assert np.array_equal(got[..., 1], want[..., 1]) and np.array_equal(got[..., 3], want[..., 3])
    assert got[0, 0, 0, 1] == F * (S - 1) * n_fft * k and got[..., 0].max() > 10
    assert np.abs(got[..., 0].astype(np.int64) - want[..., 0].astype(np.int64)).max() <= 2, (got[..., 0], want[..., 0])
    assert np.abs(got[..., 2].astype(np.int64) - want[..., 2].astype(np.int64)).max() <= 2, (got[..., 2], want[..., 2])
    for fr, c_, d_ in ((4, gc, gd), (5, gc2, gd2)):
        lab, noise = O.gen_labels(osys, seed, 1, fr), O.gen_noise(osys, seed, 1, fr)
        oc, od = O.frame(osys, w_tx.astype(np.float64), w_rx.astype(np.float64), h[1].astype(np.complex128), float(snrs[0]), lab, noise, dump=True)
        assert np.array_equal(d_["labels_tx"], lab)
        assert _rel(d_["unit_noise"], noise) < 1e-5
        for stage in ("tx", "conv", "rx", "Y"):
            a, b = np.asarray(d_[stage]).ravel(), np.asarray(od[stage]).ravel()
            nn = min(a.size, b.size) if stage == "conv" else b.size
            assert np.abs(a[:nn] - b[:nn]).max() / np.abs(b).max() < 2e-5, (stage, fr)
        assert np.abs(c_.astype(np.int64) - oc.astype(np.int64)).max() <= 1, (c_, oc)


@pytest.mark.parametrize("S", [2, 5, 7, 12])
def test_odd_and_short_frames(channels, S):
    """symbolsPerTx other than 16: odd S runs the one-symbol-per-wave kernel for every N."""
    st = W.make_structure("WOLA", 256, 32)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    h = channels[30:31].astype(np.complex64)
    cfg = W.make_cfg(st, 4, S, 21, 1, 2, 1, seed=S, frames_per_cell=40, frame_offset=3)
    snrs = np.array([6.0, 26.0], dtype=np.float32)
    got = W.run_counts(cfg, w_tx, w_rx, h, snrs)
    want = O.run(_osys(st, 4, S, 21, True), w_tx.astype(np.float64), w_rx.astype(np.float64),
                 h.astype(np.complex128), snrs.astype(np.float64), S, 3, 40)
    assert np.array_equal(got[..., 1], want[..., 1]) and got[0, 0, 0, 1] == 40 * (S - 1) * 256 * 4
    assert (np.abs(got[..., 0].astype(np.int64) - want[..., 0].astype(np.int64)) <= 2).all()
    assert (np.abs(got[..., 2].astype(np.int64) - want[..., 2].astype(np.int64)) <= 2).all()


@pytest.mark.parametrize("system,n_fft,cp,k,matlab", [("wtx", 64, 16, 2, 1), ("WOLA", 256, 32, 4, 1),
                                                      ("CPW", 256, 14, 6, 0), ("wrx", 512, 32, 4, 1)])
def test_generate_sweep_counts_match_oracle(channels, system, n_fft, cp, k, matlab):
    """Persistent multi-cell launch (2 window pairs x 3 SNR x 2 channels), frames from a
    non-zero offset: counters vs the oracle's generate-mode sweep on the same Philox streams."""
    S, seed, F, off = 16, 77, 24, 1000
    st = W.make_structure(system, n_fft, cp)
    w_tx = np.stack([W.tx_rc_window(st), np.full(st.sym_len, 1.0)]).astype(np.float32)
    w_rx = np.stack([W.rx_rc_window(st), W.rx_rc_window(st)]).astype(np.float32)
    h = channels[10:12].astype(np.complex64)
    snrs = np.array([0.0, 14.0, 28.0], dtype=np.float32)
    cfg = W.make_cfg(st, k, S, 21, 2, 3, 2, noise_before_truncate=bool(matlab), seed=seed,
                     frames_per_cell=F, frame_offset=off)
    got = W.run_counts(cfg, w_tx, w_rx, h, snrs)
    want = O.run(_osys(st, k, S, 21, matlab), w_tx.astype(np.float64), w_rx.astype(np.float64),
                 h.astype(np.complex128), snrs.astype(np.float64), seed, off, F)
    assert got.shape == want.shape == (2, 3, 2, 4)
    assert np.array_equal(got[..., 1], want[..., 1]) and np.array_equal(got[..., 3], want[..., 3])
    bits = float(want[0, 0, 0, 1])
    # fp32 vs fp64 marginal decisions only: <= 1e-4 of the bits per cell
    assert (np.abs(got[..., 0].astype(np.int64) - want[..., 0].astype(np.int64)) <= max(2, 1e-4 * bits)).all()
    assert (np.abs(got[..., 2].astype(np.int64) - want[..., 2].astype(np.int64)) <= max(2, 1e-4 * bits)).all()
    assert got[..., 0].sum() > 0


def test_injected_sweep_counts_match_oracle(channels):
    S, k, F = 16, 4, 3
    st = W.make_structure("WOLA", 256, 32)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    h = channels[20:22].astype(np.complex64)
    snrs = np.array([10.0, 30.0], dtype=np.float32)
    cfg = W.make_cfg(st, k, S, 21, 2, 2, 1, noise_before_truncate=True, seed=0, frames_per_cell=F)
    osys = _osys(st, k, S, 21, True)
    nl = O.noise_len(osys)
    rs = np.random.RandomState(1)
    labels = rs.randint(0, 16, size=(4, F, S, 256)).astype(np.uint8)
    noise = (rs.randn(4, F, nl) + 1j * rs.randn(4, F, nl)).astype(np.complex64)
    got = W.run_counts_injected(cfg, w_tx, w_rx, h, snrs, labels, noise)
    want = np.zeros((1, 2, 2, 4), dtype=np.uint64)
    for cell in range(4):
        sn, ch = divmod(cell, 2)
        for f in range(F):
            c, _ = O.frame(osys, w_tx.astype(np.float64), w_rx.astype(np.float64),
                           h[ch].astype(np.complex128), float(snrs[sn]), labels[cell, f],
                           noise[cell, f].astype(np.complex128))
            want[0, sn, ch] += c
    assert np.array_equal(got[..., 1], want[..., 1])
    assert (np.abs(got[..., 0].astype(np.int64) - want[..., 0].astype(np.int64)) <= 2).all()
    assert (np.abs(got[..., 2].astype(np.int64) - want[..., 2].astype(np.int64)) <= 2).all()


def test_reference_ser_replay_on_gpu(golden):
    """The reference simulator's own seeded SER (tests/golden/ser_replay.npz) reproduced by the
    HIP path in injected mode: the reference's 16-point alphabet is the Gray grid up to scale,
    so each drawn symbol is mapped to the label of the same constellation point."""
    g = golden("ser_replay.npz")
    table = O.qam_table(4) * np.sqrt(10)
    to_gray = np.array([int(np.argmin(np.abs(table - s))) for s in SYM16], dtype=np.uint8)
    assert np.allclose(table[to_gray], SYM16)
    S = 16
    for idx in (7, 8, 9):                              # the N=256 cases: wtx, WOLA, CPW(cp 10)
        key = "case%d" % idx
        n_fft, cp, cs, ttx, trx, rm, shift, ens, seed = [int(v) for v in g[key + "_cfg"]]
        snr, h = g[key + "_snr"], g[key + "_h"]
        st = W.variants.Structure(str(g[key + "_system"]), n_fft, cp, ttx, trx, cs, rm, shift)
        B = st.stride
        w_tx = np.stack([g[key + "_wtx"], g[key + "_wtx_rc"]]).astype(np.float32)
        w_rx = np.stack([g[key + "_wrx"], g[key + "_wrx_rc"]]).astype(np.float32)
        n_snr, n_ch = snr.size, h.shape[0]
        labels = np.zeros((2, n_snr, n_ch, ens, S, n_fft), dtype=np.uint8)
        noise = np.zeros((2, n_snr, n_ch, ens, S * B), dtype=np.complex64)
        rs = np.random.RandomState(seed)
        for si in range(n_snr):
            for ci in range(n_ch):
                for e in range(ens):
                    lab = to_gray[rs.choice(16, size=(n_fft, S)).T]
                    for pi in range(2):
                        labels[pi, si, ci, e] = lab
                        re = rs.randn(S * B); im = rs.randn(S * B)
                        noise[pi, si, ci, e] = re + 1j * im
        cfg = W.make_cfg(st, 4, S, h.shape[1], n_ch, n_snr, 2, noise_before_truncate=False,
                         frames_per_cell=ens)
        counts = W.run_counts_injected(cfg, w_tx, w_rx, h.astype(np.complex64),
                                       snr.astype(np.float32),
                                       labels.reshape(2 * n_snr * n_ch, ens, S, n_fft),
                                       noise.reshape(2 * n_snr * n_ch, ens, S * B))
        ser = counts[..., 2].sum(axis=2) / counts[..., 3].sum(axis=2)
        syms = ens * n_ch * (S - 1) * n_fft
        # exact up to fp32-vs-fp64 decisions on the boundary: at most 2 symbols per point
        assert np.abs(ser[0] - g[key + "_ser_opt"]).max() <= 2.0 / syms + 1e-12, key
        assert np.abs(ser[1] - g[key + "_ser_rc"]).max() <= 2.0 / syms + 1e-12, key


def test_frame_ranges_add_up_bit_exactly(channels):
    """Sharding property (what the multi-GPU path relies on): counters of [0,F) equal the sum of
    the counters of any split of the frame range, for every launch geometry."""
    st = W.make_structure("wtx", 256, 32)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    snrs = np.linspace(-5, 50, 12).astype(np.float32)
    cfg = W.make_cfg(st, 4, 16, 21, 1, 12, 1, seed=2)
    with W.Plan(cfg, w_tx, w_rx, channels[:1].astype(np.complex64), snrs) as plan:
        whole = plan.run(0, 4000)
        parts = plan.run(0, 1) + plan.run(1, 1499) + plan.run(1500, 2500)
        again = plan.run(0, 4000)
    assert np.array_equal(whole, parts) and np.array_equal(whole, again)
    ber = whole[0, :, 0, 0] / whole[0, :, 0, 1]
    assert (np.diff(ber) <= 0).all() and (np.diff(ber[:8]) < 0).all() and ber[0] > 0.2 and ber[-1] < 1e-3


def test_repeatable_at_baseline_scale(channels):
    """The C2 launch of bench.py, twice: identical counters, and no bit error at 45 / 50 dB -- one
    corrupted frame there would show as hundreds.  (Guards the matrix-pipe FIR's instruction-order
    hazard found in round 2, which only appeared as a few wrong frames in 10^4 under load.)"""
    st = W.make_structure("wtx", 256, 32)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    snrs = np.arange(-5.0, 51.0, 5.0).astype(np.float32)
    cfg = W.make_cfg(st, 4, 16, 21, 1, 12, 1, seed=2)
    with W.Plan(cfg, w_tx, w_rx, channels[:1].astype(np.complex64), snrs) as plan:
        a = plan.run(0, 62500)
        b = plan.run(0, 62500)
    assert np.array_equal(a, b)
    assert a[0, -1, 0, 0] == 0 and a[0, -2, 0, 0] == 0
    assert a[0, 0, 0, 1] == 62500 * 15 * 256 * 4


def test_awgn_textbook_ber_full_size():
    """Size-independent property at BASELINE scale: identity channel, rectangular windows,
    QPSK: BER must follow Q(sqrt(snr_eff)) with the one-symbol pilot doubling the noise."""
    from math import erfc, sqrt
    st = W.make_structure("CP", 256, 32)
    cfg = W.make_cfg(st, 2, 16, 1, 1, 3, 1, seed=9, frames_per_cell=20000)
    snrs = np.array([4.0, 8.0, 12.0], dtype=np.float32)
    counts = W.run_counts(cfg, np.ones(st.sym_len), np.ones(st.rx_win_len), np.ones((1, 1)), snrs)
    ber = counts[0, :, 0, 0] / counts[0, :, 0, 1]
    assert counts[0, 0, 0, 1] == 20000 * 15 * 256 * 2
    for b, s in zip(ber, snrs):
        # per-sample SNR s; after the N-point DFT signal and noise both scale by N; the LS pilot
        # estimate adds an equal, independent noise term -> effective SNR ~ s/2 (plus a small
        # ratio-distribution correction), so bracket between the two closed forms
        lo = 0.5 * erfc(sqrt(10 ** (s / 10) / 2))            # perfect channel knowledge
        hi = 0.5 * erfc(sqrt(10 ** (s / 10) / 2 / 4.0))
        assert lo < b < hi, (s, b, lo, hi)


def test_error_paths():
    st = W.make_structure("wtx", 256, 32)
    bad = W.make_cfg(st, 3, 16, 21, 1, 1, 1)
    with pytest.raises(W._lib.WofdmError) as e:
        W.run_counts(bad, np.ones(st.sym_len), np.ones(st.rx_win_len), np.ones((1, 21)), [10.0])
    assert e.value.code == -2
    cfg = W.make_cfg(st, 4, 16, 21, 1, 1, 1)
    cfg.prefix_rm = 5
    with pytest.raises(W._lib.WofdmError) as e:
        W.run_counts(cfg, np.ones(st.sym_len), np.ones(st.rx_win_len), np.ones((1, 21)), [10.0])
    assert e.value.code == -1
    with pytest.raises(W._lib.WofdmError) as e:
        W.Plan(W.make_cfg(st, 4, 16, 21, 1, 1, 1), np.ones(st.sym_len), np.ones(st.rx_win_len),
               np.ones((1, 21)), [10.0], device=99)
    assert e.value.code == -3


def test_reference_operator_mirrors(channels, tmp_path):
    """run_simulation (MATLAB signature) and wOFDMSystem.run_simulation (Python signature)."""
    st = W.make_structure("WOLA", 256, 32)
    ber = W.run_simulation(50, 16, 4, 256, 32, st.cs, np.diag(W.tx_rc_window(st)), channels[0], 20.0,
                           st.tail_tx, st.tail_rx, np.diag(W.rx_rc_window(st)), st.prefix_rm,
                           st.circ_shift, seed=3)
    want = O.run(_osys(st, 4, 16, 21, True), W.tx_rc_window(st).astype(np.float32).astype(np.float64),
                 W.rx_rc_window(st).astype(np.float32).astype(np.float64),
                 channels[:1].astype(np.complex64).astype(np.complex128), [20.0], 3, 0, 50)
    assert abs(ber - want[0, 0, 0, 0] / want[0, 0, 0, 1]) < 2e-4
    model = W.wOFDMSystem("wtx", 256, 32, 8, 0, str(tmp_path), seed=4)
    snr = np.arange(0, 40, 10)
    opt, rc = model.run_simulation(channels[:3].T, np.diag(W.tx_rc_window(model.structure)),
                                   np.diag(W.rx_rc_window(model.structure)), 20, snr, 16)
    assert (tmp_path / "ser" / "opt_wtx_32.npy").exists() and (tmp_path / "ser" / "rc_wtx_32.npy").exists()
    assert opt.shape == (4,) and (np.diff(opt) < 0).all()
    # same windows on both pairs, independent data: equal within Monte-Carlo noise
    assert np.abs(opt - rc).max() < 0.05
    # anchors from the reference run in BASELINE.md section 2 (wtx, N=256, RC windows)
    assert abs(opt[1] - 0.511) < 0.05 and abs(opt[2] - 0.112) < 0.03


def test_edge_sizes(channels):
    """Empty launch, one-tap channel, two-symbol frames, the largest CP+CS the kernel takes."""
    st = W.make_structure("wtx", 256, 32)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    cfg = W.make_cfg(st, 4, 16, 21, 1, 1, 1, seed=1, frames_per_cell=0)
    assert (W.run_counts(cfg, w_tx, w_rx, channels[:1].astype(np.complex64), [10.0]) == 0).all()
    # one tap (n_taps = 1): conv length T, MATLAB noise length T
    cfg = W.make_cfg(st, 4, 2, 1, 1, 1, 1, seed=3, frames_per_cell=50)
    got = W.run_counts(cfg, w_tx, w_rx, np.array([[0.8 - 0.3j]], np.complex64), [12.0])
    want = O.run(_osys(st, 4, 2, 1, True), w_tx.astype(np.float64), w_rx.astype(np.float64),
                 np.array([[0.8 - 0.3j]], np.complex64).astype(np.complex128), [12.0], 3, 0, 50)
    assert np.array_equal(got[..., 1], want[..., 1]) and abs(int(got[0, 0, 0, 0]) - int(want[0, 0, 0, 0])) <= 2
    # cp + cs = 128 (kernel maximum), CPW: cs = tail_tx + tail_rx/2
    st2 = W.make_structure("CPW", 256, 115)
    assert st2.cp + st2.cs == 128
    with pytest.raises(W._lib.WofdmError) as e:       # stride 256+115+13-8 = 376 > 320: FIR tiling limit
        W.run_counts(W.make_cfg(st2, 4, 16, 21, 1, 1, 1, frames_per_cell=1), W.tx_rc_window(st2),
                     W.rx_rc_window(st2), channels[:1], [10.0])
    assert e.value.code == -2
    st3 = W.make_structure("CPW", 256, 59)             # stride 256+59+13-8 = 320: the limit itself
    w3t, w3r = W.tx_rc_window(st3).astype(np.float32), W.rx_rc_window(st3).astype(np.float32)
    cfg = W.make_cfg(st3, 2, 16, 21, 1, 1, 1, seed=5, frames_per_cell=8)
    got = W.run_counts(cfg, w3t, w3r, channels[:1].astype(np.complex64), [15.0])
    want = O.run(_osys(st3, 2, 16, 21, True), w3t.astype(np.float64), w3r.astype(np.float64),
                 channels[:1].astype(np.complex64).astype(np.complex128), [15.0], 5, 0, 8)
    assert np.array_equal(got[..., 1], want[..., 1]) and abs(int(got[0, 0, 0, 0]) - int(want[0, 0, 0, 0])) <= 2


@pytest.mark.parametrize("system,n_fft,k,frames", [("wtx", 256, 4, 6000), ("WOLA", 64, 2, 20000),
                                                   ("CPW", 512, 4, 1500), ("WOLA", 1024, 6, 700)])
def test_repeated_launches_are_bit_identical(channels, system, n_fft, k, frames):
    """The waves of a workgroup synchronise through LDS flags and one barrier per frame; any
    ordering bug shows up as run-to-run differences.  Same frames five times, with grids of
    different sizes in between: the counters must not move by one bit."""
    st = W.make_structure(system, n_fft, 32)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    snrs = np.array([4.0, 12.0, 20.0], dtype=np.float32)
    cfg = W.make_cfg(st, k, 16, 21, 2, 3, 1, seed=99)
    with W.Plan(cfg, w_tx, w_rx, channels[60:62].astype(np.complex64), snrs) as plan:
        first = plan.run(10, frames)
        for _ in range(4):
            assert np.array_equal(plan.run(10, frames), first)
        # one ragged split (different workgroup-to-frame assignment), summed
        a, b = plan.run(10, frames // 3), plan.run(10 + frames // 3, frames - frames // 3)
        assert np.array_equal(a + b, first)
    assert first[..., 0].sum() > 0


SHARP = [("WOLA", 512, 32, 4, 16), ("CPwtx", 512, 20, 2, 16), ("wtx", 256, 32, 4, 16), ("CPW", 64, 16, 2, 16),
         ("WOLA", 1024, 32, 6, 16), ("wrx", 128, 20, 6, 16), ("WOLA", 512, 32, 6, 9), ("CPW", 256, 24, 4, 7),
         ("CP", 1024, 16, 2, 16), ("CPwrx", 256, 32, 6, 16),
         # strides of 293 / 291 samples: the 20-outputs-per-lane instantiation of the quarter-wave kernel
         ("CPW", 256, 32, 4, 16), ("wrx", 256, 30, 2, 8)]


@pytest.mark.parametrize("system,n_fft,cp,k,S", SHARP)
def test_production_kernels_sharp_parity(channels, system, n_fft, cp, k, S):
    """The stage-by-stage checks run the instrumented kernels; this one pins the production
    kernels: ~1e7 bits per cell at BER 0.3 ... 0.02 on the oracle's Philox streams, where one wrong
    noise or signal sample per symbol would move the error count by hundreds.  Observed: |diff| <= 3
    (fp32 vs fp64 decisions on the slicer boundaries)."""
    st = W.make_structure(system, n_fft, cp)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    snrs = np.array([5.0, 15.0, 25.0], np.float32)
    F = max(4, int(1e7 / ((S - 1) * n_fft * k)))
    cfg = W.make_cfg(st, k, S, 21, 2, 3, 1, seed=8)
    h = channels[11:13].astype(np.complex64)
    with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
        got = plan.run(3, F)
    want = O.run(_osys(st, k, S, 21, True), w_tx.astype(np.float64), w_rx.astype(np.float64),
                 h.astype(np.complex128), snrs.astype(np.float64), 8, 3, F)
    assert np.array_equal(got[..., 1], want[..., 1]) and np.array_equal(got[..., 3], want[..., 3])
    assert want[..., 0].min() > 5e3
    d = np.abs(got[..., 0].astype(np.int64) - want[..., 0].astype(np.int64))
    assert d.max() <= 12, d


# ---------------------------------------------------------------------------------------------
# Every production (non-instrumented) instantiation of the frame kernel that the compiler gave a
# non-zero ScratchSize -- the committed build table profiles/kernel_table.json
# (tools/kernel_table.py) -- is run once at sharp-parity size: register spills inside the divergent
# regions of these kernels are the hazard DESIGN.md section 4 records, and a reload of the wrong lanes
# would move the error count by hundreds.  Generate and injected mode, plain / allocation / both
# Tx-mask forms; the plan must really have picked the instantiation (wofdm_plan_kernel_id).
def _spilling_kernels():
    import json
    import os
    path = os.path.join(os.path.dirname(__file__), "..", "profiles", "kernel_table.json")
    rows = json.load(open(path))["kernels"]
    return [(r["n_fft"], r["k"], r["layout"], r["inject"], r["var"]) for r in rows
            if not r["dump"] and r["private_segment_fixed_size"] > 0]


def _poison():
    import ctypes
    import os
    W._lib.load()                                       # (one HIP runtime per process: the library first)
    path = os.path.join(os.path.dirname(__file__), "native", "libwofdm_poison.so")
    lib = ctypes.CDLL(path)
    lib.scratch_poison.argtypes = [ctypes.c_uint32]
    lib.lds_poison.argtypes = [ctypes.c_uint32]
    lib.reg_poison.argtypes = [ctypes.c_uint32]
    lib.lds_peek.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int]
    return lib


def test_scratch_poison_tool_works():
    """The helper's own check: a kernel that reads a scratch array it never wrote must see the pattern."""
    import ctypes
    lib = _poison()
    out = np.zeros(64 * 1024, np.uint32)
    assert lib.scratch_poison(0x7FC0DEAD) == 0
    assert lib.scratch_peek(out.ctypes.data_as(ctypes.c_void_p), 64) == 0
    assert (out == 0x7FC0DEAD).all()


def _geometry_for(n_fft, layout, var):
    """(system, cp, S, plan options) that make the plan pick `layout`."""
    env = {}
    if layout == 1:
        if n_fft >= 512:
            env["fir_valu"] = 1
            return "WOLA", 32, 16, env
        if var >= 2:
            env["fir_valu"] = 1                                                   # (else the masked variants run layout 9)
        return ("wtx", 32 if n_fft == 256 else 16, 16 if var >= 2 else 9, env)   # a mask forces one symbol per wave
    if layout in (9, 15):                                                         # Tx mask + matrix-pipe FIR: strides 4 | B, B >= N
        if layout == 9 and n_fft == 256 and var == 3:
            env["dft_valu"] = 1                                                   # (else the fast-convolution mask runs layout 15)
        return ("wtx" if n_fft >= 256 else "WOLA"), 32 if n_fft >= 256 else 16, 16, env
    if layout == 2:
        env.update(fir_valu=1, max_spw=2)
        return "wtx", 32 if n_fft == 256 else 16, 16, env
    if layout in (4, 5):
        env["fir_valu"] = 1
        return ("wtx" if layout == 4 else "CPW"), 32, 16, env        # strides 288 / 293
    if layout in (6, 7, 10, 11):
        if layout in (6, 7):
            env["dft_valu"] = 1
        return "wtx", (32 if layout in (6, 10) else 48), 16, env        # strides 288 / 304
    assert layout in (8, 12)
    if layout == 8:
        env["dft_valu"] = 1
    return "WOLA", 32, 16, env


@pytest.mark.parametrize("n_fft,k,layout,inject,var", _spilling_kernels())
def test_every_spilling_production_kernel(channels, n_fft, k, layout, inject, var):
    import torch
    from wofdm_amd import channel_mask as CM
    system, cp, S, env = _geometry_for(n_fft, layout, var)
    if var == 2 and n_fft <= 256:
        env["txmask_direct"] = 1
    st = W.make_structure(system, n_fft, cp)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    snrs = np.array([5.0, 15.0, 25.0], np.float32) + (k - 4) * 3.0     # BER 0.3 ... 0.01 for every k
    bits = 1e6 if var >= 2 else (2e6 if inject else 4e6)
    F = max(4, int(bits / ((S - 1) * n_fft * k)))
    seed, off = 8, 3
    cfg = W.make_cfg(st, k, S, 21, 2, 3, 1, seed=seed)
    h = channels[11:13].astype(np.complex64)
    rs = np.random.RandomState(n_fft + 7 * k + layout)
    active = None
    if var >= 1:
        active = rs.rand(n_fft) < 0.6
        active[0] = True
    mask = CM.tx_mask(st.sym_len, roll_off=10) if var >= 2 else None
    osys = O.make_sys(n_fft, k, S, st.cp, st.cs, st.tail_tx, st.tail_rx, st.prefix_rm, st.circ_shift, 21, 1,
                      active=active, tx_mask=None if mask is None else mask.astype(np.float32).astype(np.float64))
    with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
        for key, val in env.items():
            plan.set_option(key, val)
        if active is not None:
            plan.set_allocation(active)
        if mask is not None:
            plan.set_tx_mask(mask)
        assert plan.kernel_id() == (layout, var), (plan.kernel_id(), layout, var)
        # every scratch slot of the queue filled with a NaN pattern: a spill slot reloaded without having
        # been written in this launch cannot pass for the value an earlier launch left there
        assert _poison().scratch_poison(0x7FC0DEAD) == 0
        # ... and the same for every CU's LDS and every SIMD's vector registers
        assert _poison().lds_poison(0xFFFFFFFF) == 0 and _poison().reg_poison(0xFFFFFFFF) == 0
        if inject:
            # the oracle's own Philox draws, handed over as injected data
            nl = O.noise_len(osys)
            labels = np.zeros((6, F, S, n_fft), np.uint8)
            noise = np.zeros((6, F, nl), np.complex64)
            for cell in range(6):
                for f in range(F):
                    labels[cell, f] = O.gen_labels(osys, seed, cell, off + f)
                    noise[cell, f] = O.gen_noise(osys, seed, cell, off + f)
            counts = plan.new_counts()
            dl = torch.from_numpy(labels).cuda()
            dn = torch.from_numpy(noise.view(np.float32).reshape(6, F, nl, 2)).cuda()
            plan.launch_injected(F, dl, dn, counts)
            torch.cuda.synchronize()
            plan.status()
            got = counts.cpu().numpy().view(np.uint64)
        else:
            got = plan.run(off, F)
    want = O.run(osys, w_tx.astype(np.float64), w_rx.astype(np.float64), h.astype(np.complex128),
                 snrs.astype(np.float64), seed, off, F)
    assert np.array_equal(got[..., 1], want[..., 1]) and np.array_equal(got[..., 3], want[..., 3])
    assert want[..., 0].min() > 2e2
    d = np.abs(got[..., 0].astype(np.int64) - want[..., 0].astype(np.int64))
    assert d.max() <= 12, (d, got[..., 0], want[..., 0])


def test_lost_flag_is_reported(channels):
    """libwofdm_hip_fault.so is the library with one deliberate protocol error (wave 1 of every
    workgroup skips publishing its symbols in its third frame): the waiting wave must run out of its
    bounded wait, and the launch must come back as WOFDM_E_HIP -- not as plausible counters."""
    import os
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(__file__), "..")
    lib = os.path.join(root, "tests", "native", "libwofdm_hip_fault.so")
    assert os.path.exists(lib), "make -C tests/native builds it"
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import wofdm_amd as W
ch = np.load(%r)["h"]
st = W.make_structure("wtx", 256, 32)
cfg = W.make_cfg(st, 4, 16, 21, 1, 2, 1, seed=2)
with W.Plan(cfg, W.tx_rc_window(st), W.rx_rc_window(st), ch[:1].astype(np.complex64), [10.0, 20.0]) as plan:
    assert plan.kernel_id() == (10, 0)
    ok = plan.run(0, 2)                      # two frames per workgroup at most: the fault is not reached
    try:
        plan.run(0, 20000)
    except W._lib.WofdmError as e:
        print("CODE", e.code)
    else:
        print("NO ERROR")
""" % (os.path.abspath(root), os.path.abspath(os.path.join(root, "tests", "golden", "channels_vehA.npz")))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, WOFDM_LIB=lib),
                         capture_output=True, text=True, timeout=300)
    assert "CODE -3" in out.stdout, (out.stdout, out.stderr[-2000:])


@pytest.mark.parametrize("n_fft", [1024, 512, 256, 128, 64])
@pytest.mark.parametrize("k", [2, 4, 6])
def test_first_launch_in_a_fresh_process(n_fft, k):
    """Every production instantiation of one (n_fft, k) translation unit, each launched for the FIRST time in a fresh
    process (cold instruction cache and translation for its code), equals its own third launch frame by frame.
    Round 2's wrong first launches: an instruction fetch (a page boundary inside the FIR's MFMA chain) separated two
    MFMAs of a chain, and packed op_sel arithmetic of the SIMD's other waves came out wrong in lanes 48..63
    (tools/ubench/mfma_stall_victim.hip; tests/test_code_layout.py is the static guard)."""
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(__file__), "..")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "first_launch_unit.py"), str(n_fft), str(k)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    last = [l for l in r.stdout.split("\n") if l.startswith("TOTAL")][-1].split()
    n_kernels, n_bad = int(last[1]), int(last[2])
    assert n_kernels >= (8 if n_fft >= 512 else 6), r.stdout
    assert n_bad == 0, r.stdout


def test_back_to_back_mfma_chains_leave_the_neighbours_alone():
    """The assumption the FIR's chain rests on, checked on THIS GPU (tools/ubench/mfma_stall_victim.hip --quick): six
    in-place v_mfma_f32_16x16x32_f16 issued back to back -- up to 6 wait states between two of them -- never disturb packed
    op_sel arithmetic of the SIMD's other waves, even when those are the older (preferred) waves; 16 wait states between
    the fifth and the sixth do (asserted as well since round 4: the day this GPU family stops showing it, the test says so)."""
    import subprocess
    exe = os.path.join(os.path.dirname(__file__), "native", "mfma_stall_victim")
    assert os.path.exists(exe), "make -C tests/native"
    r = subprocess.run([exe, "--quick"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [dict(kv.split("=") for kv in l.split()[1:]) for l in r.stdout.split("\n") if l.startswith("RESULT")]
    assert len(rows) == 5, r.stdout
    for row in rows:
        assert int(row["chains_bad"]) == 0, row                       # the chain's own sums are never what breaks
        if int(row["gap"]) <= 6:
            assert int(row["bystanders_bad"]) == 0, row
        else:
            # the hazard side is asserted too (round 4): if a GPU / driver / firmware stops showing it, this fails, and the
            # contract that rests on it -- one frame kernel per GPU at a time, include/wofdm.h -- can be dropped.  Measured
            # behaviour of this silicon, not a documented rule (hazards 1 and 4).
            assert int(row["bystanders_bad"]) > 0, ("16 wait states inside an MFMA chain no longer disturb the SIMD's other waves: "
                                                    "revisit hazard 1 in DESIGN.md section 4", row)
            print("\n16 wait states inside the chain: %s wrong bystander values" % row["bystanders_bad"])


def test_mfma_trains_only_hit_op_sel_swizzles():
    """What layouts 10 / 11 rest on, checked on THIS GPU (tools/ubench/mfma_block_train.hip --quick): trains of MFMA blocks a few
    dozen cycles apart -- the matrix-pipe transforms -- leave every packed / mixed-precision instruction form those kernels
    contain exact in the SIMD's other waves (op_sel_hi broadcasts, neg, no modifier, v_cvt_pk_f16_f32 + v_fma_mix*), while
    v_pk_add_f32 with an op_sel swizzle -- which they do not contain, csrc/verify_code_layout.py -- is hit (asserted too)."""
    import subprocess
    exe = os.path.join(os.path.dirname(__file__), "native", "mfma_block_train")
    assert os.path.exists(exe), "make -C tests/native"
    r = subprocess.run([exe, "--quick"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [dict(kv.split("=") for kv in l.split()[1:]) for l in r.stdout.split("\n") if l.startswith("RESULT")]
    assert len(rows) == 9, r.stdout
    for row in rows:
        assert int(row["chains_bad"]) == 0, row                       # the blocks' own sums are never what breaks
        if int(row["bystander"]) != 3:
            assert int(row["bystanders_bad"]) == 0, row
        else:
            # (asserted as well: see test_back_to_back_mfma_chains_leave_the_neighbours_alone)
            assert int(row["bystanders_bad"]) > 0, ("MFMA trains no longer disturb op_sel-swizzled packed arithmetic of the SIMD's other "
                                                    "waves: revisit hazard 4 in DESIGN.md section 4", row)
            print("\nop_sel-swizzled v_pk_add_f32 beside trains of three-blocks: %s wrong values" % row["bystanders_bad"])


def test_mfma_sees_a_vector_write_one_instruction_earlier():
    """What mma_operand_fence (wofdm_kernel.hip) and the distance check of tests/test_code_layout.py rest on, on THIS GPU
    (tools/ubench/mfma_after_mix.hip --quick): an MFMA whose operand word was written by the vector instruction DIRECTLY in front of
    it computes with the old word (printed, not asserted); with one or more instructions in between it never does."""
    import subprocess
    exe = os.path.join(os.path.dirname(__file__), "native", "mfma_after_mix")
    assert os.path.exists(exe), "make -C tests/native"
    r = subprocess.run([exe, "--quick"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [dict(kv.split("=") for kv in l.split()[1:]) for l in r.stdout.split("\n") if l.startswith("RESULT")]
    assert len(rows) == 8, r.stdout
    for row in rows:
        assert all(int(row[g]) == 0 for g in ("gap1", "gap2", "gap4", "gap8")), row
    print("\nwrong MFMA results with the operand written directly in front: "
          + ", ".join("%s of %s" % (row["gap0"], row["total"]) for row in rows))


def test_lds_poison_tool_works():
    """The helper's own check: a fresh LDS allocation shows what the previous workgroup on the CU left there."""
    lib = _poison()
    hits = np.zeros(512, np.uint32)
    assert lib.lds_poison(0xC0FFEE11) == 0
    assert lib.lds_peek(0xC0FFEE11, hits.ctypes.data, 512) == 0
    assert (hits == 160 * 1024 // 4).all()


@pytest.mark.parametrize("system,n_fft,k,S", [("wtx", 256, 4, 16), ("CPW", 256, 6, 16), ("WOLA", 256, 2, 12), ("WOLA", 256, 4, 6),
                                               ("WOLA", 512, 4, 16), ("CPW", 512, 6, 7), ("WOLA", 1024, 6, 16), ("CPW", 1024, 2, 13),
                                               ("wtx", 128, 4, 16), ("WOLA", 64, 2, 16)])
def test_no_kernel_reads_what_an_earlier_kernel_left_on_chip(channels, system, n_fft, k, S):
    """LDS words, vector registers and scratch slots all survive from one kernel to the next: a frame kernel
    that read any of them before writing it would give the same answer launch after launch of ONE plan and a
    different one after a different kernel.  Every on-chip store is filled with NaN / zero / 1.0 patterns
    before a launch; the counters must not move."""
    st = W.make_structure(system, n_fft, 32 if n_fft >= 256 else 16)
    cfg = W.make_cfg(st, k, S, 21, 2, 3, 1, seed=8)
    F = max(4, int(1e6 / ((S - 1) * n_fft * k)))
    lib = _poison()
    with W.Plan(cfg, W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32),
                channels[11:13].astype(np.complex64), np.array([5.0, 15.0, 25.0], np.float32)) as plan:
        ref = plan.run(3, F)
        for pat in (0xFFFFFFFF, 0x00000000, 0x3C003C00, 0x7BFF7BFF):
            assert lib.lds_poison(pat) == 0 and lib.reg_poison(pat) == 0 and lib.scratch_poison(pat) == 0
            assert np.array_equal(plan.run(3, F), ref), hex(pat)
