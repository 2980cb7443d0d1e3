"""Window design against the reference's own Hessians and minimisers
(tests/golden/window_design.npz, made by tests/golden/make_golden.py from
python/optimization_tools/optimizers.py) and against the closed-form interference power."""
import os

import numpy as np
import pytest

import wofdm_amd as W
from wofdm_amd import interference as I
from wofdm_amd import variants as V
from wofdm_amd import window_design as D

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "window_design.npz"))
N_FFT, CP, BETA, DELTA = (int(v) for v in GOLD["cfg"])
SINGLE = {"wtx": "tx", "CPwtx": "tx", "wrx": "rx", "CPwrx": "rx"}


def _structure(system):
    btx = BETA if system in ("wtx", "CPwtx", "WOLA", "CPW") else 0
    brx = DELTA if system in ("wrx", "CPwrx", "WOLA", "CPW") else 0
    return V.make_structure(system, N_FFT, CP, btx, brx)


@pytest.mark.parametrize("system", ["wtx", "CPwtx", "wrx", "CPwrx", "WOLA", "CPW"])
def test_hessian_matches_reference_loops(system):
    st = _structure(system)
    fn = {"tx": D.hessian_tx, "rx": D.hessian_rx}.get(SINGLE.get(system), D.hessian_txrx)
    H = fn(st, GOLD["h_avg"])
    ref = GOLD[system + "_H"]
    assert H.shape == ref.shape
    assert np.abs(H - ref).max() <= 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("system", ["wtx", "CPwtx", "wrx", "CPwrx", "WOLA", "CPW"])
def test_minimiser_matches_reference_solver(system):
    st = _structure(system)
    x, info = D.optimize_tail_vector(system, N_FFT, CP, GOLD["h_avg"], st.tail_tx, st.tail_rx)
    ref = GOLD[system + "_x"]
    assert x.shape == ref.shape
    # the reference's interior-point / trust-constr iterates stop at their own tolerances
    assert np.abs(x - ref).max() <= 2e-5
    H = 0.5 * (GOLD[system + "_H"] + GOLD[system + "_H"].T)
    lift = (lambda v: v) if system in SINGLE else (lambda v: np.kron(v[BETA + 1:], v[:BETA + 1]))
    mine, theirs = 0.5 * lift(x) @ H @ lift(x), 0.5 * lift(ref) @ H @ lift(ref)
    assert mine <= theirs * (1 + 1e-9) + 1e-18           # at least as good a minimum


def test_ici_form_equals_the_explicit_double_sum():
    rs = np.random.RandomState(3)
    n, p = 6, 9
    B = rs.randn(n, p) + 1j * rs.randn(n, p)
    C = rs.randn(p, n) + 1j * rs.randn(p, n)
    ref = np.zeros((p, p), complex)
    for i in range(p):
        for j in range(p):
            for m in range(n):
                for k in range(n):
                    if m != k:
                        ref[i, j] += C[i, k] * B[m, i] * np.conj(C[j, k]) * np.conj(B[m, j])
    assert np.allclose(D._ici_form(B, C), ref, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("system", ["wtx", "CPwtx", "WOLA"])
def test_tx_quadratic_form_is_the_ici_power(system):
    """w^T Q_ici w equals the ICI power of interference.py for any Tx window (the ISI terms
    differ by the reference's modelling choices, so only alpha = 1 is compared)."""
    st = V.make_structure(system, 64, 14)
    h = GOLD["h_avg"]
    rs = np.random.RandomState(5)
    w_tx = V.tx_rc_window(st) * (1 + 0.1 * rs.rand(st.sym_len))
    w_rx = V.rx_rc_window(st)
    Q = D.quad_tx(st, h, w_rx, "matlab", alpha=1.0)
    a0, _ = I.interference_matrices(st, w_tx, w_rx, h)
    ici = (np.abs(a0 - np.diag(np.diag(a0))) ** 2).sum()
    assert np.isclose(0.5 * w_tx @ Q @ w_tx, ici, rtol=1e-10)


@pytest.mark.parametrize("system", ["wrx", "CPwrx", "CPW"])
def test_rx_quadratic_form_is_the_ici_power(system):
    st = V.make_structure(system, 64, 14)
    h = GOLD["h_avg"]
    rs = np.random.RandomState(6)
    w_tx = V.tx_rc_window(st)
    w_rx = V.rx_rc_window(st) * (1 + 0.1 * rs.rand(st.rx_win_len))
    Q = D.quad_rx(st, h, w_tx, "matlab", alpha=1.0)
    a0, _ = I.interference_matrices(st, w_tx, w_rx, h)
    ici = (np.abs(a0 - np.diag(np.diag(a0))) ** 2).sum()
    assert np.isclose(0.5 * w_rx @ Q @ w_rx, ici, rtol=1e-10)


def test_reduce_matrices_are_the_window_expansions():
    st = V.make_structure("WOLA", 32, 12)
    xt = np.r_[1.0, np.linspace(0.9, 0.1, st.tail_tx)]
    xr = np.r_[1.0, np.linspace(0.45, 0.05, st.tail_rx // 2)]
    assert np.allclose(D.reduce_matrix_tx(st) @ xt, V.expand_tx_window(st, xt))
    assert np.allclose(D.reduce_matrix_rx(st) @ xr, V.expand_rx_window(st, xr))


def test_solve_qp_small_known_answers():
    # unconstrained minimum inside the box
    H = np.array([[2.0, 0.5], [0.5, 1.0]])
    A = np.array([[1.0, 1.0]])
    x = D.solve_qp(H, A, np.array([1.0]), None, None)
    lam = np.linalg.solve(np.block([[H, A.T], [A, np.zeros((1, 1))]]), np.r_[0, 0, 1.0])[:2]
    assert np.allclose(x, lam)
    # active inequality: min x^2 + y^2 s.t. x + y = 1, x <= 0.2
    x = D.solve_qp(2 * np.eye(2), A, np.array([1.0]), np.array([[1.0, 0.0]]), np.array([0.2]))
    assert np.allclose(x, [0.2, 0.8])
    with pytest.raises(ValueError):
        D.solve_qp(np.eye(1), np.array([[1.0]]), np.array([2.0]), np.array([[1.0]]), np.array([1.0]))


def test_matlab_flavour_windows_are_feasible_and_no_worse_than_rc(tmp_path):
    h = GOLD["h_avg"]
    for system in ("wtx", "CPwrx", "WOLA"):
        st = V.make_structure(system, 64, 14)
        win = D.optimize_window_matlab(system, 64, 14, h)
        rc_tx, rc_rx = V.tx_rc_window(st), V.rx_rc_window(st)
        for name, w in win.items():
            assert w.min() >= -1e-9 and w.max() <= 1 + 1e-9
            if w.size == st.sym_len:                                   # Tx window
                assert np.allclose(w[st.tail_tx:st.sym_len - st.tail_tx], 1.0)
            else:                                                      # Rx window: folds to 1
                assert np.allclose(w[:st.tail_rx] + w[st.n_fft:], 1.0)
                assert np.allclose(w[st.tail_rx:st.n_fft], 1.0)
        if system == "wtx":
            Q = D.quad_tx(st, h, rc_rx, "matlab")
            w = win["optimizedWindow"]
            assert w @ Q @ w <= rc_tx @ Q @ rc_tx * (1 + 1e-9)
        path = D.save_window_mat(str(tmp_path), system, 14, win)
        from wofdm_amd import driver
        assert driver.parse_window_file_name(os.path.basename(path)) == (system, 14)
        back = driver.load_window_file(path)
        assert set(back) == set(win)
        for k in win:
            assert np.allclose(back[k], win[k])


def test_optimization_fun_writes_the_reference_file_layout(tmp_path):
    ch = np.load(os.path.join(os.path.dirname(__file__), "golden", "channels_vehA.npz"))["h"]
    cpath = tmp_path / "vehicularA.npy"
    np.save(cpath, ch[:8].T)                                           # [taps x realisations]
    x, info = D.optimization_fun(("CPwtx", 64, 16, str(cpath), str(tmp_path / "win")))
    saved = np.load(tmp_path / "win" / "CPwtx_16.npy")
    assert saved.shape == (9, 1) and np.allclose(saved[:, 0], x)
    assert np.load(tmp_path / "win" / "condition_number" / "CPwtx_16.npy") == info["condition_number"]
    # and the simulation side reads it back into a window (wofdm_simulation.py:50-66)
    st = V.make_structure("CPwtx", 64, 16)
    xt, xr = V.split_tail_file(st, saved)
    assert V.expand_tx_window(st, xt).shape == (st.sym_len,)
