import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import wofdm_amd as W
np.random.seed(7)
h = W.channels.gen_channel_file("vehicularA", no_channels=250).T
snr = np.linspace(-20, 50, 30)
rs = np.random.RandomState(1)
for rep in range(2):
  for system in ("wtx", "wrx", "WOLA", "CPW", "CPwtx", "CPwrx"):
    line=[]
    for cp in range(10, 33, 2):
        st = W.make_structure(system, 256, cp)
        xt = np.concatenate(([1.0], np.sort(rs.uniform(.05, .95, st.tail_tx))[::-1])) if st.tail_tx else None
        xr = np.concatenate(([1.0], np.sort(rs.uniform(.05, .45, st.tail_rx // 2))[::-1])) if st.tail_rx else None
        wt = W.expand_tx_window(st, xt) if st.tail_tx else None
        wr = W.expand_rx_window(st, xr) if st.tail_rx else None
        if system in ("wtx", "CPwtx"): windows = {"optimizedWindow": wt}
        elif system in ("wrx", "CPwrx"): windows = {"optimizedWindow": wr}
        else: windows = {"optimizedWindowCaseAStep1": wt, "optimizedWindowCaseAStep2": wr, "optimizedWindowCaseAStep3": wt, "optimizedWindowCaseBStep1": wr, "optimizedWindowCaseBStep2": wt, "optimizedWindowCaseBStep3": wr}
        t1 = time.perf_counter()
        res, counts = W.ber_for_window_file(system, cp, windows, h, snr, ensemble=100, seed=cp)
        dt=(time.perf_counter() - t1)*1e3
        syms=int(counts[..., 3].sum()) // (15 * 256) * 16
        line.append("%d:%.0fms/%.2f" % (cp, dt, syms/dt/1e6))
    print(rep, system, " ".join(line))
