for rep in 1 2 3; do for lib in "$@"; do echo -n "$lib "; WOFDM_LIB=$PWD/$lib timeout -k 10 200 python tools/run_one.py WOLA 1024 6 100 20 100 2>/dev/null | tail -1 | cut -c1-90; done; done
