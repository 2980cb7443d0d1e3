for lib in "$@"; do echo "== $lib"; WOFDM_LIB=$PWD/$lib timeout -k 10 300 python tools/bench_configs.py --c4-full 2>/dev/null | grep "N=1024 k=6"; done
