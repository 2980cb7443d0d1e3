for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    WOFDM_LIB=$PWD/$lib timeout -k 10 300 python tools/bench_configs.py 2>/dev/null | grep -v "cells=1 "
  done
done
