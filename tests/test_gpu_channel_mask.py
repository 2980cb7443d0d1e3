"""GPU parity of the main_channel_mask.m variant: subcarrier allocation
(half-band loading) and the per-symbol spectral Tx mask, through the C ABI, against the oracle
on the same Philox streams."""
import numpy as np
import pytest

import wofdm_amd as W
from oracle import oracle as O

pytestmark = pytest.mark.gpu
STAGE_RTOL = 2e-5


def _osys(st, k, S, n_taps, matlab, active=None, tx_mask=None):
    return O.make_sys(st.n_fft, k, S, st.cp, st.cs, st.tail_tx, st.tail_rx, st.prefix_rm,
                      st.circ_shift, n_taps, 1 if matlab else 0, active=active, tx_mask=tx_mask)


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def _half_band(n):
    a = np.zeros(n, bool)
    a[:n // 4] = True
    a[3 * n // 4:] = True
    return a


def _check_decisions(gd, od, gc, oc, k, active):
    mism = (gd["labels_rx"] != od["labels_rx"]) & active[None, :]
    if mism.any():
        a = np.sqrt(2 * (2 ** k - 1) / 3)
        z = od["Xhat"][mism] * a
        edge = np.minimum(np.abs((z.real / 2) - np.round(z.real / 2)) * 2,
                          np.abs((z.imag / 2) - np.round(z.imag / 2)) * 2)
        assert (edge < 1e-3).all()
    assert abs(int(gc[0]) - int(oc[0])) <= 2 * int(mism.sum())
    assert abs(int(gc[2]) - int(oc[2])) <= int(mism.sum())


ALLOC_CASES = [("wtx", 256, 32, 4, "half"), ("WOLA", 256, 22, 4, "half"), ("CPW", 64, 16, 2, "half"),
               ("CPwrx", 128, 20, 6, "random"), ("wrx", 512, 32, 4, "guard"), ("WOLA", 1024, 32, 6, "half"),
               ("CP", 256, 16, 4, "single")]


def _allocation(kind, n):
    if kind == "half":
        return _half_band(n)
    if kind == "guard":                       # timefreq_simulation.py:223-233, guard band 48
        a = np.zeros(n, bool)
        a[1:n // 2 - 48 + 1] = True
        a[n // 2 + 48:] = True
        return a
    if kind == "single":
        a = np.zeros(n, bool)
        a[n // 4 + 3] = True
        return a
    return np.random.RandomState(n).rand(n) < 0.6


@pytest.mark.parametrize("system,n_fft,cp,k,kind", ALLOC_CASES)
@pytest.mark.parametrize("inject", [False, True])
def test_allocation_one_frame_stage_by_stage(channels, system, n_fft, cp, k, kind, inject):
    S, seed, frame, cell = 16, 5, 4242, 3
    st = W.make_structure(system, n_fft, cp)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    h = channels[20:22].astype(np.complex64)
    snrs = np.array([10.0, 24.0], dtype=np.float32)
    active = _allocation(kind, n_fft)
    cfg = W.make_cfg(st, k, S, 21, 2, 2, 1, seed=seed)
    osys = _osys(st, k, S, 21, True, active=active)
    lab, noise = O.gen_labels(osys, seed, cell, frame), O.gen_noise(osys, seed, cell, frame)
    oc, od = O.frame(osys, w_tx.astype(np.float64), w_rx.astype(np.float64),
                     h[1].astype(np.complex128), float(snrs[1]), lab, noise, dump=True)
    with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
        plan.set_allocation(active)
        gc, gd = plan.dump_frame(cell, frame, *((lab, noise.astype(np.complex64)) if inject else ()))
    nact = int(active.sum())
    assert np.array_equal(gd["labels_tx"], lab)
    assert int(gc[1]) == int(oc[1]) == (S - 1) * nact * k and int(gc[3]) == int(oc[3]) == (S - 1) * nact
    assert np.all(gd["X"][:, ~active] == 0)
    assert _rel(gd["X"], od["X"]) < 1e-6
    for stage in ("tx", "conv", "rx", "Y"):
        assert _rel(gd[stage], od[stage]) < STAGE_RTOL, stage
    _check_decisions(gd, od, gc, oc, k, active)


@pytest.mark.parametrize("system,n_fft,k", [("wtx", 256, 4), ("CPW", 256, 6), ("WOLA", 1024, 2)])
def test_allocation_sweep_counts_match_oracle(channels, system, n_fft, k):
    S, seed, F, off = 16, 31, 20, 77
    st = W.make_structure(system, n_fft, 24)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    h = channels[3:5].astype(np.complex64)
    snrs = np.array([2.0, 15.0, 30.0], dtype=np.float32)
    active = _half_band(n_fft)
    cfg = W.make_cfg(st, k, S, 21, 2, 3, 1, seed=seed, frames_per_cell=F, frame_offset=off)
    with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
        plan.set_allocation(active)
        got = plan.run(off, F)
        plan.set_allocation(None)                       # back to every bin loaded
        full = plan.run(off, F)
    want = O.run(_osys(st, k, S, 21, True, active=active), w_tx.astype(np.float64),
                 w_rx.astype(np.float64), h.astype(np.complex128), snrs.astype(np.float64), seed, off, F)
    assert np.array_equal(got[..., 1], want[..., 1]) and np.array_equal(got[..., 3], want[..., 3])
    assert got[0, 0, 0, 1] == F * (S - 1) * (n_fft // 2) * k
    bits = float(want[0, 0, 0, 1])
    assert (np.abs(got[..., 0].astype(np.int64) - want[..., 0].astype(np.int64)) <= max(2, 1e-4 * bits)).all()
    assert (np.abs(got[..., 2].astype(np.int64) - want[..., 2].astype(np.int64)) <= max(2, 1e-4 * bits)).all()
    assert full[0, 0, 0, 1] == F * (S - 1) * n_fft * k
    ref_full = O.run(_osys(st, k, S, 21, True), w_tx.astype(np.float64), w_rx.astype(np.float64),
                     h.astype(np.complex128), snrs.astype(np.float64), seed, off, F)
    assert (np.abs(full[..., 0].astype(np.int64) - ref_full[..., 0].astype(np.int64)) <= max(2, 1e-4 * 2 * bits)).all()


def test_allocation_rejects_bad_input(channels):
    st = W.make_structure("wtx", 64, 16)
    cfg = W.make_cfg(st, 2, 16, 21, 1, 1, 1)
    with W.Plan(cfg, W.tx_rc_window(st), W.rx_rc_window(st), channels[:1].astype(np.complex64),
                np.array([10.0], np.float32)) as plan:
        with pytest.raises(W._lib.WofdmError):
            plan.set_allocation(np.zeros(64, bool))
        with pytest.raises(ValueError):
            plan.set_allocation(np.ones(32, bool))


# ---------------------------------------------------------------------------- spectral Tx mask
from wofdm_amd import channel_mask as CM  # noqa: E402

MASK_CASES = [("wtx", 256, 32, 4), ("wtx", 256, 30, 4), ("WOLA", 256, 22, 4), ("CPW", 64, 16, 2), ("CPwrx", 128, 20, 6),
              ("wrx", 512, 32, 4), ("CP", 256, 16, 4), ("CPwtx", 256, 10, 6), ("CPW", 256, 32, 4), ("wrx", 256, 22, 6)]


@pytest.mark.parametrize("system,n_fft,cp,k", MASK_CASES)
@pytest.mark.parametrize("inject,direct", [(False, False), (True, False), (False, True)])
def test_tx_mask_one_frame_stage_by_stage(channels, system, n_fft, cp, k, inject, direct):
    # direct: force the direct-form convolution where the fast-convolution form would be used
    S, seed, frame, cell = 16, 9, 777, 1
    st = W.make_structure(system, n_fft, cp)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    h = channels[40:42].astype(np.complex64)
    snrs = np.array([26.0], dtype=np.float32)
    active = CM.half_band_allocation(n_fft)
    mask = CM.tx_mask(st.sym_len).astype(np.float32)
    cfg = W.make_cfg(st, k, S, 21, 2, 1, 1, seed=seed)
    osys = _osys(st, k, S, 21, True, active=active, tx_mask=mask.astype(np.float64))
    lab, noise = O.gen_labels(osys, seed, cell, frame), O.gen_noise(osys, seed, cell, frame)
    oc, od = O.frame(osys, w_tx.astype(np.float64), w_rx.astype(np.float64),
                     h[1].astype(np.complex128), float(snrs[0]), lab, noise, dump=True)
    with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
        plan.set_option("txmask_direct", 1 if direct else 0)
        plan.set_allocation(active)
        plan.set_tx_mask(mask)
        info = plan.info()
        assert info["waves_per_workgroup"] == S                 # one symbol per wave
        plan.set_tx_mask(None)
        base = plan.info()["lds_bytes"] if S % 2 else None       # (even S: plain runs 2 symbols/wave)
        plan.set_tx_mask(mask)
        fft_form = n_fft <= 256 and 3 * st.sym_len - 2 <= 1024 and not direct
        # fast-convolution form: 1024 twiddles + 8 scratch rows of 1024 points behind the frame
        if plan.kernel_id()[0] == 15:
            # (N = 256 with the FIR on the matrix pipe: the mask's transforms run there too, without exchange scratch; the
            # LDS behind the frame holds their tables and one spill row per wave instead)
            assert fft_form and info["lds_bytes"] < 9 * 1024 * 8 + 51136
        else:
            assert (info["lds_bytes"] >= 9 * 1024 * 8) == fft_form or n_fft == 512
        gc, gd = plan.dump_frame(cell, frame, *((lab, noise.astype(np.complex64)) if inject else ()))
    assert np.array_equal(gd["labels_tx"], lab)
    assert int(gc[1]) == int(oc[1]) and int(gc[3]) == int(oc[3])
    for stage in ("X", "tx", "conv", "rx", "Y"):
        assert _rel(gd[stage], od[stage]) < STAGE_RTOL, stage
    _check_decisions(gd, od, gc, oc, k, active)
    # independent restatement of the mask stage with numpy FFTs on the oracle's own symbols
    rows = (np.fft.ifft(od["X"], axis=1)[:, (np.arange(st.sym_len) - st.cp) % n_fft]
            * w_tx.astype(np.float64)[None, :])
    filt = CM.dft_rc_filt(rows)
    tx = np.zeros(st.frame_len(S), complex)
    for s in range(S):
        tx[s * st.stride:s * st.stride + st.sym_len] += filt[s]
    assert _rel(od["tx"], tx) < 1e-7            # the mask handed to both is float32-rounded
    assert _rel(gd["tx"], tx) < STAGE_RTOL


@pytest.mark.parametrize("S", [16, 5, 11])
def test_tx_mask_sweep_and_removal(channels, S):
    seed, F, off, n_fft, k = 13, 12, 5, 256, 4
    st = W.make_structure("WOLA", n_fft, 24)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    h = channels[8:10].astype(np.complex64)
    snrs = np.array([5.0, 25.0], dtype=np.float32)
    active = CM.half_band_allocation(n_fft)
    mask = CM.tx_mask(st.sym_len)
    cfg = W.make_cfg(st, k, S, 21, 2, 2, 1, seed=seed)
    args = (w_tx.astype(np.float64), w_rx.astype(np.float64), h.astype(np.complex128),
            snrs.astype(np.float64), seed, off, F)
    with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
        plain0 = plan.run(off, F)
        plan.set_allocation(active)
        plan.set_tx_mask(mask)
        masked = plan.run(off, F)
        plan.set_tx_mask(None)
        alloc = plan.run(off, F)
        plan.set_allocation(None)
        plain1 = plan.run(off, F)
    assert np.array_equal(plain0, plain1)                      # options leave no residue
    want_m = O.run(_osys(st, k, S, 21, True, active=active, tx_mask=mask), *args)
    want_a = O.run(_osys(st, k, S, 21, True, active=active), *args)
    for got, want in ((masked, want_m), (alloc, want_a)):
        assert np.array_equal(got[..., 1], want[..., 1])
        bits = float(want[0, 0, 0, 1])
        assert (np.abs(got[..., 0].astype(np.int64) - want[..., 0].astype(np.int64)) <= max(2, 1e-4 * bits)).all()


@pytest.mark.parametrize("system,cp", [("wtx", 32), ("WOLA", 24), ("CP", 0)])
def test_tx_mask_kernels_count_the_same(channels, system, cp):
    """The three masked kernels of N = 256 -- every transform on the matrix pipe (layout 15), FIR on the matrix pipe and the
    transforms on the VALU (layout 9, plan option dft_valu), everything on the VALU (layout 1, fir_valu) -- on the same frames:
    the same decisions but for a handful of symbols on a boundary; repeated launches bit-identical."""
    st = W.make_structure(system, 256, cp)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    snrs = np.arange(0.0, 36.0, 6.0).astype(np.float32)
    h = channels[3:5].astype(np.complex64)
    cfg = W.make_cfg(st, 4, 16, 21, 2, snrs.size, 1, seed=21)
    active, mask = CM.half_band_allocation(256), CM.tx_mask(st.sym_len).astype(np.float32)
    got = {}
    for layout, opts in ((15, {}), (9, {"dft_valu": 1}), (1, {"fir_valu": 1})):
        with W.Plan(cfg, w_tx, w_rx, h, snrs) as plan:
            for key, val in opts.items():
                plan.set_option(key, val)
            plan.set_allocation(active)
            plan.set_tx_mask(mask)
            assert plan.kernel_id() == (layout, 3)
            a, b = plan.run(7, 400), plan.run(7, 400)
            assert np.array_equal(a, b)
            got[layout] = a
    ref = got[1]
    for layout in (15, 9):
        assert np.array_equal(got[layout][..., 1], ref[..., 1]) and np.array_equal(got[layout][..., 3], ref[..., 3])
        d = np.abs(got[layout][..., 0].astype(np.int64) - ref[..., 0].astype(np.int64))
        assert d.max() <= 4 and d.sum() <= 12, (layout, d.ravel())


@pytest.mark.parametrize("system,n_fft,cp,k,opts", [("wtx", 256, 32, 4, {}), ("WOLA", 256, 24, 6, {}), ("wtx", 256, 32, 4, {"dft_valu": 1}),
                                                     ("WOLA", 128, 16, 4, {})])
def test_tx_mask_production_and_instrumented_kernels_count_the_same(channels, system, n_fft, cp, k, opts):
    """The stage-by-stage parity of the masked variant runs the instrumented kernels; the same injected frames through the
    PRODUCTION kernel of the same layout (15, 9) must give the same error counters."""
    import torch
    S, seed, F = 16, 23, 24
    st = W.make_structure(system, n_fft, cp)
    w_tx, w_rx = W.tx_rc_window(st).astype(np.float32), W.rx_rc_window(st).astype(np.float32)
    h = channels[30:31].astype(np.complex64)
    snrs = np.array([8.0 + 3.0 * (k - 2), 18.0 + 3.0 * (k - 2)], np.float32)
    active, mask = CM.half_band_allocation(n_fft), CM.tx_mask(st.sym_len).astype(np.float32)
    cfg = W.make_cfg(st, k, S, 21, 1, 2, 1, seed=seed)
    osys = _osys(st, k, S, 21, True, active=active, tx_mask=mask.astype(np.float64))
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
        plan.set_allocation(active)
        plan.set_tx_mask(mask)
        assert plan.kernel_id() == ((15 if n_fft == 256 and not opts else 9), 3)
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


def test_tx_mask_limits(channels):
    st = W.make_structure("WOLA", 1024, 32)
    cfg = W.make_cfg(st, 2, 16, 21, 1, 1, 1)
    with W.Plan(cfg, W.tx_rc_window(st), W.rx_rc_window(st), channels[:1].astype(np.complex64),
                np.array([10.0], np.float32)) as plan:
        with pytest.raises(W._lib.WofdmError) as e:
            plan.set_tx_mask(CM.tx_mask(st.sym_len))
        assert e.value.code == -2
        assert plan.run(0, 2)[0, 0, 0, 1] == 2 * 15 * 1024 * 2     # still usable, unmasked


def test_run_sim_mc_mirror(channels):
    st = W.make_structure("wtx", 256, 16)
    masked, plain = CM.run_sim_mc("wtx", 256, 16, W.tx_rc_window(st), W.rx_rc_window(st),
                                  channels[:2], [0.0, 30.0], ensemble=30)
    assert masked.shape == plain.shape == (1, 2, 2, 4)
    assert (plain[..., 1] == 30 * 15 * 128 * 4).all() and (masked[..., 1] == plain[..., 1]).all()
    ber_p, ber_m = plain[..., 0] / plain[..., 1], masked[..., 0] / masked[..., 1]
    assert (ber_p[0, 0] > ber_p[0, 1]).all() and (ber_m[0, 0] > ber_m[0, 1]).all()
    assert (ber_m[0, 1] < 0.1).all()
