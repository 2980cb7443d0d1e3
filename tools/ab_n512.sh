# Developer tool (GPU box): interleaved N = 512 / 1024 timing of several builds.   bash tools/ab_n512.sh lib_a.so lib_b.so ...
for i in 1 2 3; do
  for lib in "$@"; do
    WOFDM_LIB=$PWD/$lib python tools/run_one.py WOLA 512 4 10 20 1000 2>/dev/null | sed "s#^#$lib #" | cut -c1-150
  done
done
