# Developer tool (GPU box): rocprofv3 --kernel-trace --stats of the N = 1024 / N = 512 configs -> gpurun_out/<tag>_kernel_stats_N*.csv
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfgname in "WOLA 1024 6 100 20 100" "WOLA 512 4 10 20 1000"; do
  n=$(echo $cfgname | cut -d' ' -f2)
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_N$n -- python3 $R/tools/run_one.py $cfgname > $R/gpurun_out/stats_N$n.log 2>&1
  find $R/gpurun_out/stats_N$n -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/${TAG}_kernel_stats_N$n.csv
  head -2 $R/gpurun_out/${TAG}_kernel_stats_N$n.csv | cut -c1-200
done
