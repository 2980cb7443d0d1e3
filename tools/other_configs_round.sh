# Developer tool (GPU box): the round's profiles/<tag>_other_configs.txt.   bash tools/other_configs_round.sh r02
TAG=${1:-r02}
cd "$(dirname "$0")/.."
OUT=gpurun_out/${TAG}_other_configs.txt
{
echo "## tools/bench_configs.py --c4-full (HIP-event time of one launch per config; layouts 10/11/12 = FIR and both transforms on the matrix pipe)"
timeout -k 10 300 python tools/bench_configs.py --c4-full 2>&1 | grep -v amdgpu.ids
echo; echo "## the same with the plan option dft_valu = 1 (round-2 layouts 6/7/8: FIR on the matrix pipe, transforms on the VALU)"
WOFDM_OPTS=dft_valu=1 timeout -k 10 300 python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids
echo; echo "## the same with WOFDM_FIR_VALU=1 (round-1 layouts: FIR on the VALU)"
WOFDM_FIR_VALU=1 timeout -k 10 300 python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids
echo; echo "## tools/bench_inject.py (injected randomness streamed from HBM)"
timeout -k 10 300 python tools/bench_inject.py 2>&1 | grep -v amdgpu.ids
echo; echo "## tools/full_reference_sweep.py"
timeout -k 10 300 python tools/full_reference_sweep.py 2>&1 | grep -v amdgpu.ids
echo; echo "## tools/bench_channel_mask.py (row f1)"
timeout -k 10 300 python tools/bench_channel_mask.py 2>&1 | grep -v amdgpu.ids
echo; echo "## tools/bench_even_strides.py (strides of 2 mod 4 with one symbol per wave: matrix-pipe layouts against layout 1)"
timeout -k 10 300 python tools/bench_even_strides.py 2>&1 | grep -v amdgpu.ids
echo; echo "## tools/bench_interference.py (row f2)"
timeout -k 10 300 python tools/bench_interference.py 2>&1 | grep -v amdgpu.ids
} > $OUT
tail -5 $OUT
