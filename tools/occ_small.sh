# Developer tool (GPU box): N = 64 / 128 rates against the number of workgroups launched per CU (-DWOFDM_DEV_OCC builds of wofdm_abi.hip)
for lib in "$@"; do
  echo "== $lib"
  WOFDM_LIB=$PWD/$lib timeout -k 10 200 python tools/bench_small_partial.py 2>/dev/null | grep -v "(2, 0)"
done
