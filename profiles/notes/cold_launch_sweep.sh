# Developer tool (GPU box): tools/cold_launch_probe.py over the production kernels, one fresh process each.
#   bash tools/cold_launch_sweep.sh [library]     (default: the in-tree build)
[ -n "$1" ] && export WOFDM_LIB=$1
cd "$(dirname "$0")/.."
for n in 1024 512 256 128 64; do for k in 6 4 2; do for inj in 1 0; do for var in 0 1 2; do [ $n = 1024 ] && [ $var = 2 ] && continue   # (no Tx-mask kernel at N = 1024)
  timeout -k 10 100 python tools/cold_launch_probe.py $n $k $inj $var 2>&1 | grep "frames differing" || echo "N=$n k=$k inject=$inj var=$var: FAILED TO RUN"
done; done; done; done
