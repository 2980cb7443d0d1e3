# Developer tool (GPU box): the round's evidence for profiles/.   bash tools/profile_round.sh r02
# smoke, bench line, rocprofv3 kernel trace, then one --pmc pass per counter group.
set -e
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_stats.log 2>&1
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc$i -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc$i.log 2>&1 || echo "pmc pass $i failed"
  echo "pmc pass $i done"
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc*/*/*_counter_collection.csv > gpurun_out/${TAG}_pmc_summary.txt
find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_kernel_stats.csv
cat gpurun_out/smoke.log | tail -2; cat gpurun_out/${TAG}_kernel_stats.csv | head -5; cat gpurun_out/${TAG}_pmc_summary.txt
