# Developer tool (GPU box): PMC counters of the Tx-mask kernel (row f1: wtx N=256 16-QAM, half-band allocation + mask)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export RUN_MASK=1
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmcM_$i -- python3 $R/tools/run_one.py wtx 256 4 1 12 20000 > $R/gpurun_out/pmcM_$i.log 2>&1
done
cd $R
tail -1 gpurun_out/pmcM_1.log
python tools/pmc_summary.py gpurun_out/pmcM_*/*/*_counter_collection.csv
