# Developer tool (GPU box): the round's profiles/<tag>_other_configs.txt.   bash tools/other_configs_round.sh r02
TAG=${1:-r02}
cd "$(dirname "$0")/.."
OUT=gpurun_out/${TAG}_other_configs.txt
{
echo "## tools/bench_configs.py --c4-full (HIP-event time of one launch per config; layouts 6/7/8 = matrix-pipe FIR)"
timeout -k 10 300 python tools/bench_configs.py --c4-full 2>&1 | grep -v amdgpu.ids
echo; echo "## the same with WOFDM_FIR_VALU=1 (round-1 layouts: FIR on the VALU)"
WOFDM_FIR_VALU=1 timeout -k 10 300 python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids
echo; echo "## tools/bench_inject.py (injected randomness streamed from HBM)"
timeout -k 10 300 python tools/bench_inject.py 2>&1 | grep -v amdgpu.ids
echo; echo "## tools/full_reference_sweep.py"
timeout -k 10 300 python tools/full_reference_sweep.py 2>&1 | grep -v amdgpu.ids
echo; echo "## tools/bench_channel_mask.py (row f1)"
timeout -k 10 300 python tools/bench_channel_mask.py 2>&1 | grep -v amdgpu.ids
echo; echo "## tools/bench_interference.py (row f2)"
timeout -k 10 300 python tools/bench_interference.py 2>&1 | grep -v amdgpu.ids
} > $OUT
tail -5 $OUT
