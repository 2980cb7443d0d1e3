set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfgname in "WOLA 1024 6 100 20 100" "WOLA 512 4 10 20 1000"; do
  tag=$(echo $cfgname | cut -d' ' -f2)
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmcN${tag}_$i -- python3 $R/tools/run_one.py $cfgname > $R/gpurun_out/pmcN${tag}_$i.log 2>&1
  done
  cd $R
  echo "== $cfgname"; tail -1 gpurun_out/pmcN${tag}_1.log
  python tools/pmc_summary.py gpurun_out/pmcN${tag}_*/*/*_counter_collection.csv
  cd /tmp
done
