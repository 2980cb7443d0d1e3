"""wofdm_amd -- MI355X-native windowed-OFDM Monte-Carlo BER hot path.

The directory is named ``w-ofdm-optimization_amd`` (not an importable identifier); import it
as ``wofdm_amd`` through the shim module at the repository root.

  variants     structure table, raised-cosine / optimised window vectors  (host, numpy)
  simulation   run_simulation / wOFDMSystem / simulation_fun mirrors + Plan (ctypes -> HIP)
  distributed  frame-range sharding + counter all-reduce
  driver       main_BER_calculation.m / `-m run_sim` as functions, reference file formats
  channels     ITU-R tapped-delay-line channel realisations (gen_chan mirror)
  interference closed-form ICI+ISI power of a structure (interf_power mirror, host numpy)
  window_design  interference Hessians + QP -> optimised windows (optimizers.py / window_optimization.m)
  timefreq     Tx-side PSD / out-of-band-radiation estimate (timefreq_simulation.py)
  channel_mask main_channel_mask.m: half-band loading + spectral Tx mask (GPU: allocation / tx_mask)
  _lib         ctypes binding of libwofdm_hip.so (include/wofdm.h)
"""
from . import variants  # noqa: F401
from . import _lib  # noqa: F401
from . import distributed  # noqa: F401
from . import channels  # noqa: F401
from . import driver  # noqa: F401
from . import interference  # noqa: F401
from . import window_design  # noqa: F401
from . import timefreq  # noqa: F401
from . import channel_mask  # noqa: F401
from .simulation import (Plan, ber_for_window_file, error_rates, make_cfg,  # noqa: F401
                         results_from_counts, run_counts, run_counts_injected, run_simulation,
                         save_ber_results, simulation_fun, wOFDMSystem)
from ._lib import kernel_source_hash  # noqa: F401
from .variants import (SYSTEMS, Structure, calculate_parameters, expand_rx_window,  # noqa: F401
                       expand_tx_window, make_structure, rx_rc_window, tx_rc_window)

__version__ = "0.1.0"
