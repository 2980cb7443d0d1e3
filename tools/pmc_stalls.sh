# Developer tool (GPU box): stall-side counters of the C2 kernel (latency levels, instruction fetch, LDS FIFOs, scalar / vector-memory
# issue cycles), one --pmc pass per group.   bash tools/pmc_stalls.sh r04 [library]
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
[ -n "$2" ] && export WOFDM_LIB=$R/$2
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmcs*
i=0
for grp in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAVE_CYCLES" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_MISC" "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmcs$i -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmcs$i.log 2>&1 || echo "pmc pass $i failed"
done
cd $R
python tools/pmc_summary.py gpurun_out/pmcs*/*/*_counter_collection.csv > gpurun_out/${TAG}_pmc_stalls.txt
cat gpurun_out/${TAG}_pmc_stalls.txt
