for i in 1 2 3; do
  for lib in "$@"; do
    WOFDM_LIB=$PWD/$lib python tools/bench_configs.py --c2-only 2>/dev/null | sed "s#^#$lib #"
  done
done
