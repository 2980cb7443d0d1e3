# Developer tool (GPU box): bench.py kernel time of several builds of the library, interleaved.
#   bash tools/ab_bench.sh ab/lib_a.so ab/lib_b.so ...
for rep in 1 2 3; do
  for lib in "$@"; do
    WOFDM_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['roofline']['kernel_ms_avg'],3), '%.4e' % d['value'])"
  done
done
