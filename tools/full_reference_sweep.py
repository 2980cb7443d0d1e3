#!/usr/bin/env python3
"""The reference's whole BER experiment on one GPU, timed.

main_BER_calculation.m runs, for every window file (6 structures x 12 CP lengths), 30 SNR
points x all channel realisations x {2 | 7} window pairs x `ensemble` frames of 16 symbols.
With the Python CLI's 250 channels and ensemble 100 that is about 3.2e9 OFDM symbols.  The
optimised windows are not in the reference repository, so monotone random tails in the
reference's on-disk format stand in for them (throughput does not depend on the values).

    python tools/full_reference_sweep.py [--channels 250] [--ensemble 100]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wofdm_amd as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=250)
    ap.add_argument("--ensemble", type=int, default=100)
    a = ap.parse_args()
    np.random.seed(7)
    h = W.channels.gen_channel_file("vehicularA", no_channels=a.channels).T     # [n_ch, 21]
    snr = np.linspace(-20, 50, 30)
    rs = np.random.RandomState(1)
    total_syms, t_gpu, files = 0, 0.0, 0
    t0 = time.perf_counter()
    for system in ("wtx", "wrx", "WOLA", "CPW", "CPwtx", "CPwrx"):
        for cp in range(10, 33, 2):
            try:
                st = W.make_structure(system, 256, cp)
            except ValueError:
                continue
            xt = np.concatenate(([1.0], np.sort(rs.uniform(.05, .95, st.tail_tx))[::-1])) if st.tail_tx else None
            xr = np.concatenate(([1.0], np.sort(rs.uniform(.05, .45, st.tail_rx // 2))[::-1])) if st.tail_rx else None
            wt = W.expand_tx_window(st, xt) if st.tail_tx else None
            wr = W.expand_rx_window(st, xr) if st.tail_rx else None
            if system in ("wtx", "CPwtx"):
                windows = {"optimizedWindow": wt}
            elif system in ("wrx", "CPwrx"):
                windows = {"optimizedWindow": wr}
            else:
                windows = {"optimizedWindowCaseAStep1": wt, "optimizedWindowCaseAStep2": wr,
                           "optimizedWindowCaseAStep3": wt, "optimizedWindowCaseBStep1": wr,
                           "optimizedWindowCaseBStep2": wt, "optimizedWindowCaseBStep3": wr}
            t1 = time.perf_counter()
            res, counts = W.ber_for_window_file(system, cp, windows, h, snr, ensemble=a.ensemble, seed=files)
            t_gpu += time.perf_counter() - t1
            total_syms += int(counts[..., 3].sum()) // (15 * 256) * 16
            files += 1
    dt = time.perf_counter() - t0
    print("%d window files, %.3e OFDM symbols, %.2f s in wofdm_run (%.3e symbols/s incl. plan setup, "
          "H2D/D2H), %.2f s wall" % (files, total_syms, t_gpu, total_syms / t_gpu, dt))


if __name__ == "__main__":
    main()
