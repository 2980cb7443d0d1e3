# Developer tool (GPU box): everything profiles/r04_* is made of, in one call (about 8 minutes).   bash tools/round4_evidence.sh
set -x
bash tools/profile_round.sh r04 > gpurun_out/r04_profile_round.log 2>&1
tail -4 gpurun_out/r04_profile_round.log
bash tools/sustained_round.sh r04
bash tools/other_configs_round.sh r04 > /dev/null 2>&1
{ echo "# tools/bench_small_partial.py: N = 64 / 128 geometries in layout 16 (round 4: run-time symbols per wave, partly filled last wave) against layout 2 (plan option dft_valu), layouts 13 beside them; 12 SNR points, one channel"
  timeout -k 10 300 python tools/bench_small_partial.py 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r04_small_partial.txt
bash tools/stats_bign.sh r04 > gpurun_out/r04_stats_bign.log 2>&1
bash tools/pmc_bign.sh > gpurun_out/r04_pmc_bign.txt 2>&1
{ echo "# tools/stamp_report.py on the final kernels (-DWOFDM_STAMP build of the N = 256 / 512 / 1024 translation units; layouts 10 / 12 / 12): share of a wave's own"
  echo "# cycles per part of the frame, and cycles per frame.  The stamps fence the compiler's schedule: shares, not times."
  for cfg in "256 4" "512 4" "1024 6"; do WOFDM_LIB=$PWD/ab/lib_stamp.so python tools/stamp_report.py $cfg 2>/dev/null; done; } > gpurun_out/r04_stamp_report.txt
{ echo "# tools/first_launch_unit.py: every production instantiation of every (N, k) translation unit, FIRST launch in a fresh process against its third, frame by frame"
  for n in 64 128 256 512 1024; do for k in 2 4 6; do python tools/first_launch_unit.py $n $k 2>/dev/null | grep "TOTAL\|kernel"; done; done; } > gpurun_out/r04_first_launch.txt
tail -3 gpurun_out/r04_first_launch.txt
