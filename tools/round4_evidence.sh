set -x
bash tools/profile_round.sh r04 > gpurun_out/r04_profile_round.log 2>&1
tail -30 gpurun_out/r04_profile_round.log
bash tools/sustained_round.sh r04
