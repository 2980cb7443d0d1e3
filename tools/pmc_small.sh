R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export RUN_CP=16
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc64_$i -- python3 $R/tools/run_one.py WOLA 64 2 4 5 128000 > $R/gpurun_out/pmc64_$i.log 2>&1
done
cd $R
tail -1 gpurun_out/pmc64_1.log
python tools/pmc_summary.py gpurun_out/pmc64_*/*/*_counter_collection.csv
