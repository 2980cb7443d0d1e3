for rep in 1 2; do
for lib in ab/lib_base.so w-ofdm-optimization_amd/libwofdm_hip.so; do
  echo "== $lib"
  WOFDM_LIB=$PWD/$lib timeout -k 10 200 python tools/bench_configs.py 2>/dev/null | grep "cells=1 "
  WOFDM_LIB=$PWD/$lib timeout -k 10 200 python tools/bench_even_strides.py 2>/dev/null | grep -v "(1, \|(9, " 
  WOFDM_LIB=$PWD/$lib timeout -k 10 200 python tools/bench_channel_mask.py 2>/dev/null | tail -4
done
done
