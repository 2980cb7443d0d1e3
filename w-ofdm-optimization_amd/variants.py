"""w-OFDM structure table and window vectors (host logic, no GPU needed).

Mirrors, as closed-form vectors instead of dense diagonal matrices:
  * calculate_parameters            matlab/main_BER_calculation.m:457-493
    (Python twin                    python/ofdm_utils/wofdm_simulation.py:391-418)
  * tx_rc_window / rx_rc_window     matlab/main_BER_calculation.m:379-416
    (gen_rc_window_tx/_rx           python/ofdm_utils/transmitter.py:61-87,
                                    python/ofdm_utils/receiver.py:36-56)
  * reduce_variable_tx/_rx          python/optimization_tools/utils.py:13-73
    (the on-disk "optimised window" format: a short tail vector)
"""
from dataclasses import dataclass

import numpy as np

#: the six structures of the reference plus plain CP-OFDM (Python only,
#: wofdm_simulation.py:415-418).  "RC" is not a structure: it is any structure
#: run with raised-cosine windows on both sides (main_BER_calculation.m:78-83).
SYSTEMS = ("wtx", "wrx", "WOLA", "CPW", "CPwtx", "CPwrx", "CP")

#: default tails of the reference (matlab/window_optimization.m:42-47,
#: python/wofdm_optimization.py:118-123)
DEFAULT_TAIL_TX = 8
DEFAULT_TAIL_RX = 10
TX_WINDOWED = ("wtx", "WOLA", "CPW", "CPwtx")
RX_WINDOWED = ("wrx", "WOLA", "CPW", "CPwrx")


def default_tails(system):
    """(tail_tx, tail_rx) the reference uses for `system`."""
    tx = DEFAULT_TAIL_TX if system in ("wtx", "WOLA", "CPW", "CPwtx") else 0
    rx = DEFAULT_TAIL_RX if system in ("wrx", "WOLA", "CPW", "CPwrx") else 0
    return tx, rx


@dataclass(frozen=True)
class Structure:
    """All integer lengths of one w-OFDM system (symbols as in SURVEY.md 3.4)."""
    system: str
    n_fft: int      # N
    cp: int         # mu
    tail_tx: int    # beta
    tail_rx: int    # delta
    cs: int         # rho
    prefix_rm: int  # gamma
    circ_shift: int  # kappa

    @property
    def sym_len(self):      # P: windowed Tx symbol length
        return self.n_fft + self.cp + self.cs

    @property
    def stride(self):       # B: symbol stride on air = Rx block length
        return self.sym_len - self.tail_tx

    @property
    def rx_win_len(self):   # N + delta
        return self.n_fft + self.tail_rx

    def frame_len(self, syms_per_frame):   # T
        return self.tail_tx + syms_per_frame * self.stride


def calculate_parameters(system, cp, tail_tx, tail_rx):
    """(cs, prefix_rm, circ_shift) of a structure.

    Same table as matlab/main_BER_calculation.m:467-492.
    """
    if tail_rx % 2:
        raise ValueError("tail_rx must be even")
    half = tail_rx // 2
    table = {
        "wtx":   (tail_tx,        cp,           0),
        "wrx":   (half,           cp - half,    0),
        "WOLA":  (tail_tx,        cp - tail_rx, half),
        "CPW":   (tail_tx + half, cp - half,    0),
        "CPwtx": (0,              cp - tail_tx, tail_tx),
        "CPwrx": (0,              cp - tail_rx, half),
        "CP":    (0,              cp,           0),
    }
    if system not in table:
        raise ValueError("unknown w-OFDM system %r (expected one of %s)" % (system, SYSTEMS))
    cs, rm, shift = table[system]
    if rm < 0:
        raise ValueError("cyclic prefix %d too short for %s with tails (%d, %d)"
                         % (cp, system, tail_tx, tail_rx))
    return cs, rm, shift


def make_structure(system, n_fft, cp, tail_tx=None, tail_rx=None):
    dtx, drx = default_tails(system)
    tail_tx = dtx if tail_tx is None else tail_tx
    tail_rx = drx if tail_rx is None else tail_rx
    cs, rm, shift = calculate_parameters(system, cp, tail_tx, tail_rx)
    st = Structure(system, n_fft, cp, tail_tx, tail_rx, cs, rm, shift)
    # identity the Rx reshape relies on (main_BER_calculation.m:262-263)
    assert st.stride == n_fft + tail_rx + rm
    return st


def rc_tail(tail_len):
    """Rising raised-cosine tail sin^2(pi (2i+1) / (4 tail)), i = 0..tail-1."""
    i = np.arange(tail_len, dtype=np.float64)
    return np.sin(np.pi * (2.0 * i + 1.0) / (4.0 * tail_len)) ** 2 if tail_len else i


def tx_rc_window(st):
    """Raised-cosine Tx window vector, length P (main_BER_calculation.m:379-400)."""
    w = np.ones(st.sym_len, dtype=np.float64)
    if st.tail_tx:
        t = rc_tail(st.tail_tx)
        w[:st.tail_tx] = t
        w[st.sym_len - st.tail_tx:] = t[::-1]
    return w


def rx_rc_window(st):
    """Raised-cosine Rx window vector, length N+delta (main_BER_calculation.m:403-416)."""
    w = np.ones(st.rx_win_len, dtype=np.float64)
    if st.tail_rx:
        t = rc_tail(st.tail_rx)
        w[:st.tail_rx] = t
        w[st.rx_win_len - st.tail_rx:] = t[::-1]
    return w


def expand_tx_window(st, tail_vec):
    """Tail vector x (beta+1 values) -> full Tx window, length P.

    x[0] is the flat level, x[1..beta] the falling tail from the inside out
    (optimization_tools/utils.py:13-43).
    """
    x = np.asarray(tail_vec, dtype=np.float64).reshape(-1)
    if x.size != st.tail_tx + 1:
        raise ValueError("Tx tail vector must have tail_tx+1 = %d values" % (st.tail_tx + 1))
    w = np.full(st.sym_len, x[0], dtype=np.float64)
    if st.tail_tx:
        w[:st.tail_tx] = x[:0:-1]
        w[st.sym_len - st.tail_tx:] = x[1:]
    return w


def expand_rx_window(st, tail_vec):
    """Tail vector x (delta/2+1 values) -> full Rx window, length N+delta, with
    w[i] + w[N+i] = x[0] on the folded samples (optimization_tools/utils.py:46-73)."""
    x = np.asarray(tail_vec, dtype=np.float64).reshape(-1)
    half = st.tail_rx // 2
    if x.size != half + 1:
        raise ValueError("Rx tail vector must have tail_rx/2+1 = %d values" % (half + 1))
    n = st.n_fft
    w = np.full(st.rx_win_len, x[0], dtype=np.float64)
    if half:
        w[:half] = x[0] - x[1:]
        w[half:2 * half] = x[:0:-1]
        w[n:n + half] = x[1:]
        w[n + half:] = x[0] - x[:0:-1]
    return w


def split_tail_file(st, vec):
    """Split a reference window file vector (python/ofdm_utils/wofdm_simulation.py:50-66):
    WOLA/CPW files hold the Tx tail vector followed by the Rx one."""
    v = np.asarray(vec, dtype=np.float64).reshape(-1)
    if st.system in ("WOLA", "CPW"):
        return v[:st.tail_tx + 1], v[st.tail_tx + 1:]
    if st.system in ("wtx", "CPwtx"):
        return v, np.ones(1)
    if st.system in ("wrx", "CPwrx"):
        return np.ones(1), v
    return np.ones(1), np.ones(1)


#: window pairs the MATLAB driver runs per window file
#: (main_BER_calculation.m:72-83, 98-110, 135-184): name -> (tx key, rx key)
MATLAB_WINDOW_PAIRS = {
    "tx_only": (("opt", ("optimizedWindow", "rc")), ("rc", ("rc", "rc"))),
    "rx_only": (("opt", ("rc", "optimizedWindow")), ("rc", ("rc", "rc"))),
    "both": (("rc", ("rc", "rc")),
             ("1A", ("optimizedWindowCaseAStep1", "rc")),
             ("2A", ("optimizedWindowCaseAStep1", "optimizedWindowCaseAStep2")),
             ("3A", ("optimizedWindowCaseAStep3", "optimizedWindowCaseAStep2")),
             ("1B", ("rc", "optimizedWindowCaseBStep1")),
             ("2B", ("optimizedWindowCaseBStep2", "optimizedWindowCaseBStep1")),
             ("3B", ("optimizedWindowCaseBStep2", "optimizedWindowCaseBStep3"))),
}


def matlab_pair_plan(system):
    if system in ("wtx", "CPwtx"):
        return MATLAB_WINDOW_PAIRS["tx_only"]
    if system in ("wrx", "CPwrx"):
        return MATLAB_WINDOW_PAIRS["rx_only"]
    if system in ("WOLA", "CPW"):
        return MATLAB_WINDOW_PAIRS["both"]
    raise ValueError("no MATLAB window-pair plan for %r" % (system,))
