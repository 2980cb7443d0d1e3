"""Window design (host, numpy) -- SURVEY.md 8f row f3.

Produces the "optimised windows" the BER hot path only *loads*: the quadratic form of the
ICI + ISI power in the window samples, and the small constrained QP that minimises it.

Two flavours, as the reference has two:

``python``  ``optimization_fun`` / ``OptimizerTx`` / ``OptimizerRx`` / ``OptimizerTxRx``
            (python/optimization_tools/optimizers.py:25-873): variables are the tail vectors
            of ``reduce_variable_{tx,rx}`` (optimization_tools/utils.py:13-73); constraints of
            utils.py:76-143; output ``<window_path>/<sys>_<cp>.npy``.
``matlab``  ``optimize_window`` (matlab/window_optimization.m:205-596) with
            ``quad_objective_tx/rx`` (599-680): variables are all window samples, flat part
            pinned to 1, tails boxed to [0, 1]; WOLA/CPW run the alternating cases A/B of three
            steps; output ``optimal_win_<type>_VehA200_<cp>CP.mat``.

The O(P^2 N^2) loops of the reference (optimizers.py:147-157, window_optimization.m:615-627)
are replaced by the closed form

    Q_ici[i, j] = Re{ (C C^H)[i, j] (B^T B^*)[i, j]  -  (D D^H)[i, j] },   D[i, n] = C[i, n] B[n, i]

(the full double sum minus its m == n terms).  The reference's modelling choices are kept
as they are: ISI matrices of all previous symbols are summed *before* squaring, the Python
ISI term uses ``(C C^H) o (B^H B)`` (conjugated the other way round than the ICI term,
optimizers.py:175), the MATLAB ISI term keeps only the diagonal of the matrix *product*
``B2^H B2 C C^H`` (window_optimization.m:632).
"""
import os

import numpy as np

from . import interference as I
from . import variants as V

#: optimizers.py:68-72
REG_PYTHON = {"wtx": 1e-12, "CPwtx": 1e-12, "wrx": 1e-12, "CPwrx": 1e-12, "WOLA": 1e-16, "CPW": 1e-16}


# --------------------------------------------------------------------------------------------
# operators of the chain, from the index formulas of interference.py
# --------------------------------------------------------------------------------------------
def fold_dft(st):
    """[N, N+delta] = W K P: DFT of the overlap-add + circular shift of the Rx block, *without*
    the window (optimizers.py:438, window_optimization.m:656)."""
    n = st.n_fft
    m = np.arange(st.rx_win_len)
    t = (m - st.circ_shift - st.tail_rx // 2) % n
    return np.exp(-2j * np.pi * np.outer(np.arange(n), t) / n)


def _channel_sums(st, h):
    """(H_0, sum_{m>=1} H_m), each [stride, P] (optimizers.py:264-267)."""
    ht = I.channel_tensor(st, h)
    return ht[0], ht[1:].sum(axis=0)


def reduce_matrix_tx(st):
    """[P, beta+1] (utils.py:13-43); column k is the window of the k-th unit tail vector."""
    eye = np.eye(st.tail_tx + 1)
    return np.stack([V.expand_tx_window(st, e) for e in eye], axis=1)


def reduce_matrix_rx(st):
    """[N+delta, delta/2+1] (utils.py:46-73)."""
    eye = np.eye(st.tail_rx // 2 + 1)
    return np.stack([V.expand_rx_window(st, e) for e in eye], axis=1)


def _ici_form(Bm, Cm):
    """sum_{m != n} C[i,n] B[m,i] conj(C[j,n]) conj(B[m,j]) (optimizers.py:147-157), complex."""
    G = Cm @ Cm.conj().T
    K = Bm.T @ Bm.conj()
    D = Cm * Bm.T
    return G * K - D @ D.conj().T


# --------------------------------------------------------------------------------------------
# quadratic forms in the *full* window vectors
# --------------------------------------------------------------------------------------------
def quad_tx(st, h, w_rx=None, flavour="python", alpha=0.5):
    """[P, P] Q with  w_tx^T Q w_tx / 2  = the reference's interference measure of a Tx window.

    python: 2 (Q1 + Q2) of OptimizerTx.gen_hessian (optimizers.py:258-272; Rx window = identity)
    matlab: HTx of quad_objective_tx (window_optimization.m:599-636) for the Rx window ``w_rx``.
    """
    w_rx = np.ones(st.rx_win_len) if w_rx is None else np.asarray(w_rx, dtype=np.float64)
    h0, h1 = _channel_sums(st, h)
    rx = I.rx_matrix(st, w_rx)                      # W K P V_rx R   [N, stride]
    Bm, B2 = rx @ h0, rx @ h1                       # [N, P]
    Cm = I.tx_matrix(st, np.ones(st.sym_len))       # Gamma W^-1     [P, N]
    q1 = _ici_form(Bm, Cm).real
    if flavour == "python":
        q2 = ((Cm @ Cm.conj().T) * (B2.conj().T @ B2)).real
        return 2.0 * (q1 + q2)
    if flavour == "matlab":
        q2 = np.diag(np.einsum("ij,ji->i", B2.conj().T @ B2, Cm @ Cm.conj().T).real)
        return 2.0 * (alpha * q1 + (1.0 - alpha) * q2)
    raise ValueError("flavour must be 'python' or 'matlab'")


def quad_rx(st, h, w_tx=None, flavour="python", alpha=0.5):
    """[N+delta, N+delta] Q for the Rx window (OptimizerRx.gen_hessian optimizers.py:425-446,
    quad_objective_rx window_optimization.m:639-680) given the Tx window ``w_tx``."""
    w_tx = np.ones(st.sym_len) if w_tx is None else np.asarray(w_tx, dtype=np.float64)
    h0, h1 = _channel_sums(st, h)
    T = I.tx_matrix(st, w_tx)                                           # [P, N]
    rows = st.prefix_rm + np.arange(st.rx_win_len)                      # R: keep N+delta samples
    Cm, C2 = (h0 @ T)[rows], (h1 @ T)[rows]                             # [N+delta, N]
    Bm = fold_dft(st)                                                   # [N, N+delta]
    q1 = _ici_form(Bm, Cm).real
    if flavour == "python":
        q2 = ((C2 @ C2.conj().T) * (Bm.conj().T @ Bm)).real
        return 2.0 * (q1 + q2)
    if flavour == "matlab":
        q2 = np.diag(np.einsum("ij,ji->i", Bm.conj().T @ Bm, C2 @ C2.conj().T).real)
        return 2.0 * (alpha * q1 + (1.0 - alpha) * q2)
    raise ValueError("flavour must be 'python' or 'matlab'")


def hessian_tx(st, h):
    """R^T 2(Q1+Q2) R in the beta+1 tail variables (optimizers.py:271-272)."""
    R = reduce_matrix_tx(st)
    return R.T @ quad_tx(st, h) @ R


def hessian_rx(st, h):
    """R^T 2(Q1+Q2) R in the delta/2+1 tail variables (optimizers.py:445-446)."""
    R = reduce_matrix_rx(st)
    return R.T @ quad_rx(st, h) @ R


def hessian_txrx(st, h):
    """[(delta/2+1)(beta+1)]^2 Hessian in the products x_rx[a] x_tx[b] (index a (beta+1) + b) of
    OptimizerTxRx.gen_hessian (optimizers.py:812-836): the (i, j) entry of A_0 / sum A_m is
    bilinear in the two tail vectors, M'[i,j] = R_rx^T diag(B[i,:]) C diag(D[:,j]) R_tx."""
    h0, h1 = _channel_sums(st, h)
    rows = st.prefix_rm + np.arange(st.rx_win_len)
    Bm = fold_dft(st)                                                    # [N, N+delta]
    Dm = I.tx_matrix(st, np.ones(st.sym_len))                            # [P, N]
    Rrx, Rtx = reduce_matrix_rx(st), reduce_matrix_tx(st)
    n = st.n_fft
    out = 0.0
    for Cm, skip_diag in ((h0[rows], True), (h1[rows], False)):
        U = np.einsum("ma,im,mn->ian", Rrx, Bm, Cm, optimize=True)      # [N, a, P]
        mm = np.einsum("ian,nb,nj->abij", U, Rtx, Dm, optimize=True)    # [a, b, N, N]
        if skip_diag:
            mm[:, :, np.arange(n), np.arange(n)] = 0.0
        mm = mm.reshape(Rrx.shape[1] * Rtx.shape[1], n * n)
        out = out + mm @ mm.conj().T
    return 2.0 * out.real


# --------------------------------------------------------------------------------------------
# constraints and the QP
# --------------------------------------------------------------------------------------------
def constraints_tx(tail_len):
    """A x = b, C x <= d of gen_constraints_tx (utils.py:76-109): x_0 = 1, x_i <= x_0."""
    A = np.zeros((1, tail_len + 1)); A[0, 0] = 1.0
    C = np.hstack([-np.ones((tail_len, 1)), np.eye(tail_len)])
    return A, np.ones(1), C, np.zeros(tail_len)


def constraints_rx(tail_len):
    """gen_constraints_rx (utils.py:112-143): x_0 = 1, x_i <= 1, x_0 - x_i <= 1/2."""
    half = tail_len // 2
    A = np.zeros((1, half + 1)); A[0, 0] = 1.0
    C = np.vstack([np.hstack([np.zeros((half, 1)), np.eye(half)]),
                   np.hstack([np.ones((half, 1)), -np.eye(half)])])
    return A, np.ones(1), C, np.concatenate([np.ones(half), 0.5 * np.ones(half)])


def solve_qp(H, A, b, C, d, x0=None, tol=1e-13, max_iter=500):
    """min 1/2 x^T H x  s.t.  A x = b,  C x <= d   (H symmetric positive definite, n <= ~300).

    Primal active-set method on dense KKT systems: exact on these tiny, badly scaled problems
    where a barrier method stalls at its centring tolerance.  The reference solves the same
    problem with its own interior-point code (quadratic_programming.py:18-70) or MATLAB's
    ``quadprog`` (window_optimization.m:267-270); the minimiser is unique, so they agree.
    """
    H = np.asarray(H, dtype=np.float64)
    n = H.shape[0]
    scale = np.abs(H).max() or 1.0
    Hs = H / scale
    A = np.zeros((0, n)) if A is None else np.atleast_2d(np.asarray(A, dtype=np.float64))
    b = np.zeros(0) if b is None else np.asarray(b, dtype=np.float64).reshape(-1)
    C = np.zeros((0, n)) if C is None else np.atleast_2d(np.asarray(C, dtype=np.float64))
    d = np.zeros(0) if d is None else np.asarray(d, dtype=np.float64).reshape(-1)
    x = _feasible_point(A, b, C, d, x0)
    active = list(np.flatnonzero(np.abs(C @ x - d) <= 1e-12)) if C.size else []
    active = _independent(A, C, active)
    for _ in range(max_iter):
        Aw = np.vstack([A, C[active]]) if active else A
        m = Aw.shape[0]
        kkt = np.block([[Hs, Aw.T], [Aw, np.zeros((m, m))]])
        rhs = np.concatenate([-Hs @ x, np.zeros(m)])
        sol = np.linalg.lstsq(kkt, rhs, rcond=None)[0]
        p, lam = sol[:n], sol[n:]
        if np.abs(p).max() <= tol * max(1.0, np.abs(x).max()):
            mu = lam[A.shape[0]:]
            if mu.size == 0 or mu.min() >= -tol:
                return x
            active.pop(int(np.argmin(mu)))
            continue
        step, block = 1.0, None
        if C.size:
            cp_ = C @ p
            slack = d - C @ x
            for i in np.flatnonzero(cp_ > 1e-15):
                if i in active:
                    continue
                s = slack[i] / cp_[i]
                if s < step:
                    step, block = max(s, 0.0), int(i)
        x = x + step * p
        if block is not None:
            active.append(block)
    raise RuntimeError("solve_qp: active-set iteration did not converge")


def _independent(A, C, active):
    keep, rows = [], [r for r in A]
    for i in active:
        trial = np.array(rows + [C[i]])
        if np.linalg.matrix_rank(trial, tol=1e-10) == len(trial):
            rows.append(C[i]); keep.append(int(i))
    return keep


def _feasible_point(A, b, C, d, x0):
    n = A.shape[1]
    if x0 is not None:
        x0 = np.asarray(x0, dtype=np.float64).reshape(-1)
        ok_eq = A.size == 0 or np.abs(A @ x0 - b).max() <= 1e-9
        ok_in = C.size == 0 or (C @ x0 - d).max() <= 1e-9
        if ok_eq and ok_in:
            return x0.copy()
    from scipy.optimize import linprog
    res = linprog(np.zeros(n), A_ub=C if C.size else None, b_ub=d if C.size else None,
                  A_eq=A if A.size else None, b_eq=b if A.size else None, bounds=(None, None))
    if not res.success:
        raise ValueError("solve_qp: constraints are infeasible")
    return res.x


# --------------------------------------------------------------------------------------------
# python flavour: tail-vector files
# --------------------------------------------------------------------------------------------
def optimize_tail_vector(system, n_fft, cp, h_avg, tail_tx=None, tail_rx=None, reg=None):
    """``optimization_fun`` (optimizers.py:25-105) for one (system, CP): returns
    (x, info) with x the vector saved as ``<sys>_<cp>.npy`` -- beta+1 Tx tail values, delta/2+1
    Rx tail values, or their concatenation for WOLA/CPW."""
    st = V.make_structure(system, n_fft, cp, tail_tx, tail_rx)
    reg = REG_PYTHON[system] if reg is None else reg
    if system in ("wtx", "CPwtx"):
        H = hessian_tx(st, h_avg)
        H = 0.5 * (H + H.T)
        x = solve_qp(H + reg * np.eye(H.shape[0]), *constraints_tx(st.tail_tx),
                     x0=np.r_[1.0, V.rc_tail(st.tail_tx)[::-1]])
        x_eval = x
    elif system in ("wrx", "CPwrx"):
        H = hessian_rx(st, h_avg)
        H = 0.5 * (H + H.T)
        x = solve_qp(H + reg * np.eye(H.shape[0]), *constraints_rx(st.tail_rx),
                     x0=_rc_tail_vector_rx(st))
        x_eval = x
    elif system in ("WOLA", "CPW"):
        H = hessian_txrx(st, h_avg)
        H = 0.5 * (H + H.T)
        x_tx, x_rx = _solve_bilinear(H + reg * np.eye(H.shape[0]), st)
        x = np.concatenate([x_tx, x_rx])
        x_eval = np.kron(x_rx, x_tx)
    else:
        raise ValueError("no window to optimise for %r" % (system,))
    info = {"hessian": H, "condition_number": float(np.linalg.cond(H)),
            "fval": float(0.5 * x_eval @ H @ x_eval), "reg": reg}
    return x, info


def _rc_tail_vector_rx(st):
    """RC Rx window as a tail vector: x_i = w[N + i - 1] (utils.py:63-72)."""
    w = V.rx_rc_window(st)
    return np.r_[1.0, w[st.n_fft:st.n_fft + st.tail_rx // 2]]


def _solve_bilinear(H, st, sweeps=200, tol=1e-14):
    """min 1/2 (x_rx (x) x_tx)^T H (x_rx (x) x_tx) over 0 <= x_tx[1:] <= 1, 1/2 <= x_rx[1:] <= 1,
    x_tx[0] = x_rx[0] = 1 -- the feasible set of OptimizerTxRx.optimize (optimizers.py:838-870,
    equality constraints 640-668 = "x is a Kronecker product").  The reference hands the
    54-variable lifted problem to scipy ``trust-constr``; here the two convex QPs in x_tx and
    x_rx are solved exactly in turn from the same raised-cosine start (block coordinate descent,
    monotone in the cost)."""
    nt, nr = st.tail_tx + 1, st.tail_rx // 2 + 1
    x_tx = np.r_[1.0, V.rc_tail(st.tail_tx)[::-1]] if st.tail_tx else np.ones(1)
    x_rx = _rc_tail_vector_rx(st)
    H4 = H.reshape(nr, nt, nr, nt)
    At = np.zeros((1, nt)); At[0, 0] = 1.0
    Ct = np.vstack([np.hstack([np.zeros((nt - 1, 1)), np.eye(nt - 1)]),
                    np.hstack([np.zeros((nt - 1, 1)), -np.eye(nt - 1)])])
    dt = np.concatenate([np.ones(nt - 1), np.zeros(nt - 1)])
    Ar = np.zeros((1, nr)); Ar[0, 0] = 1.0
    Cr = np.vstack([np.hstack([np.zeros((nr - 1, 1)), np.eye(nr - 1)]),
                    np.hstack([np.zeros((nr - 1, 1)), -np.eye(nr - 1)])])
    dr = np.concatenate([np.ones(nr - 1), -0.5 * np.ones(nr - 1)])
    eps = 1e-30
    prev = np.inf
    for _ in range(sweeps):
        Ht = np.einsum("a,abcd,c->bd", x_rx, H4, x_rx)
        x_tx = solve_qp(0.5 * (Ht + Ht.T) + eps * np.eye(nt), At, np.ones(1), Ct, dt, x0=x_tx)
        Hr = np.einsum("b,abcd,d->ac", x_tx, H4, x_tx)
        x_rx = solve_qp(0.5 * (Hr + Hr.T) + eps * np.eye(nr), Ar, np.ones(1), Cr, dr, x0=x_rx)
        k = np.kron(x_rx, x_tx)
        cost = 0.5 * k @ H @ k
        if prev - cost <= tol * max(abs(cost), 1e-300):
            break
        prev = cost
    return x_tx, x_rx


def optimization_fun(data):
    """Work item of ``python wofdm_optimization.py -m run_opt`` (optimizers.py:25-105):
    data = (system, dft_len, cp_len, channel_path, window_path); channel file [taps x realisations]."""
    system, n_fft, cp, channel_path, window_path = data
    h_avg = np.load(channel_path).mean(axis=1)
    x, info = optimize_tail_vector(system, n_fft, cp, h_avg)
    os.makedirs(os.path.join(window_path, "condition_number"), exist_ok=True)
    np.save(os.path.join(window_path, "%s_%d.npy" % (system, cp)), x.reshape(-1, 1))
    np.save(os.path.join(window_path, "condition_number", "%s_%d.npy" % (system, cp)),
            info["condition_number"])
    return x, info


# --------------------------------------------------------------------------------------------
# matlab flavour: full-window files
# --------------------------------------------------------------------------------------------
def _box_tx(st):
    """Aeq/beq/bounds of window_optimization.m:258-266 as (A, b, C, d): flat part = 1,
    0 <= tails <= 1."""
    P, beta = st.sym_len, st.tail_tx
    flat = np.arange(beta, P - beta)
    tails = np.r_[np.arange(beta), np.arange(P - beta, P)]
    A = np.zeros((flat.size, P)); A[np.arange(flat.size), flat] = 1.0
    E = np.zeros((tails.size, P)); E[np.arange(tails.size), tails] = 1.0
    return A, np.ones(flat.size), np.vstack([E, -E]), np.r_[np.ones(tails.size), np.zeros(tails.size)]


def _box_rx(st):
    """window_optimization.m:317-323: w[i] + w[N+i] = 1 on the tails, flat part = 1 (bounds),
    0 <= tails <= 1."""
    L, n, delta = st.rx_win_len, st.n_fft, st.tail_rx
    A = np.zeros((n, L))
    A[np.arange(delta), np.arange(delta)] = 1.0
    A[np.arange(delta), n + np.arange(delta)] = 1.0
    flat = np.arange(delta, n)
    A[flat, flat] = 1.0
    tails = np.r_[np.arange(delta), np.arange(n, L)]
    E = np.zeros((tails.size, L)); E[np.arange(tails.size), tails] = 1.0
    return A, np.ones(n), np.vstack([E, -E]), np.r_[np.ones(tails.size), np.zeros(tails.size)]


def _qp_full(Q, box, x0, reg=0.0):
    Q = 0.5 * (Q + Q.T)
    scale = np.abs(Q).max() or 1.0
    # the flat samples are pinned, so semi-definiteness there is harmless; a relative ridge keeps
    # the reduced KKT systems well posed where quadprog relies on its own presolve
    return solve_qp(Q + (reg + 1e-14 * scale) * np.eye(Q.shape[0]), *box, x0=x0)


def optimize_window_matlab(system, n_fft, cp, h_avg, tail_tx=None, tail_rx=None, alpha=0.5):
    """``optimize_window`` (window_optimization.m:205-596): dict of window *vectors* keyed by
    the variable names of the ``.mat`` files (``optimizedWindow`` or
    ``optimizedWindowCase{A,B}Step{1,2,3}``)."""
    st = V.make_structure(system, n_fft, cp, tail_tx, tail_rx)
    rc_tx, rc_rx = V.tx_rc_window(st), V.rx_rc_window(st)
    qtx = lambda w_rx: quad_tx(st, h_avg, w_rx, "matlab", alpha)
    qrx = lambda w_tx: quad_rx(st, h_avg, w_tx, "matlab", alpha)
    if system in ("wtx", "CPwtx"):
        return {"optimizedWindow": _qp_full(qtx(rc_rx), _box_tx(st), rc_tx)}
    if system in ("wrx", "CPwrx"):
        return {"optimizedWindow": _qp_full(qrx(rc_tx), _box_rx(st), rc_rx)}
    if system in ("WOLA", "CPW"):
        btx, brx = _box_tx(st), _box_rx(st)
        a1 = _qp_full(qtx(rc_rx), btx, rc_tx)            # m:374-407
        a2 = _qp_full(qrx(a1), brx, rc_rx)               # m:409-441
        a3 = _qp_full(qtx(a2), btx, a1)                  # m:443-475
        b1 = _qp_full(qrx(rc_tx), brx, rc_rx)            # m:477-509
        b2 = _qp_full(qtx(b1), btx, rc_tx)               # m:511-543
        b3 = _qp_full(qrx(b2), brx, b1)                  # m:545-577
        return {"optimizedWindowCaseAStep1": a1, "optimizedWindowCaseAStep2": a2,
                "optimizedWindowCaseAStep3": a3, "optimizedWindowCaseBStep1": b1,
                "optimizedWindowCaseBStep2": b2, "optimizedWindowCaseBStep3": b3}
    raise ValueError("no window to optimise for %r" % (system,))


def save_window_mat(folder, system, cp, windows):
    """``optimal_win_<type>_VehA200_<cp>CP.mat`` with diagonal matrices, as
    window_optimization.m:299-305, 351-357, 586-593 write and main_BER_calculation.m:46-47 reads."""
    from scipy.io import savemat
    os.makedirs(folder, exist_ok=True)
    path = os.path.join(folder, "optimal_win_%s_VehA200_%dCP.mat" % (system, cp))
    savemat(path, {k: np.diag(np.asarray(v, dtype=np.float64)) for k, v in windows.items()})
    return path


def run_window_optimization(channels, settings=None, folder="optimized_windows", systems=None,
                            alpha=0.5, log=None):
    """Script body of window_optimization.m:50-199: every system of the settings x every CP
    length -> one window file.  ``channels`` [realisations x taps] (mean over rows, m:222)."""
    from .driver import DEFAULT_SETTINGS
    settings = settings or DEFAULT_SETTINGS
    gen = settings["generalSettings"]
    h_avg = np.atleast_2d(np.asarray(channels)).mean(axis=0)
    done = []
    for system in systems or [s for s in V.SYSTEMS if s in settings]:
        tails = settings[system]
        for cp in np.atleast_1d(gen["cyclicPrefix"]).astype(int):
            if log:
                log("optimising %s-OFDM, CP %d" % (system, cp))
            w = optimize_window_matlab(system, int(gen["numberSubcarriers"]), int(cp), h_avg,
                                       int(tails["tailTx"]), int(tails["tailRx"]), alpha)
            done.append(save_window_mat(folder, system, int(cp), w))
    return done
