# Developer tool (GPU box): tools/bench_configs.py lines for N=256 (12 cells), 512 and 1024 of
# several builds of the library.   bash tools/ab_configs.sh ab/lib_a.so ab/lib_b.so ...
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    WOFDM_LIB=$PWD/$lib timeout -k 10 200 python tools/bench_configs.py 2>/dev/null | grep "cells=12 \|N=512\|N=1024 k=6"
  done
done
