# Developer tool (build container): build a variant of the library into ab/lib_<name>.so
#   bash tools/build_variant.sh <name> [N-list "256"] [K-list "4"] -- <extra hipcc flags>
# Only the listed (N, k) translation units are compiled with the extra flags; the rest come from csrc/*.o.
set -e
name=$1; shift
NS=${1:-256}; shift || true
KS=${1:-4}; shift || true
[ "$1" = "--" ] && shift
cd "$(dirname "$0")/../w-ofdm-optimization_amd/csrc"
mkdir -p ../../ab/obj_$name
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wall -Wno-unused-function"
objs=""
for o in wofdm_kernel_n*_k*.o; do
  n=$(echo $o | sed 's/.*_n\([0-9]*\)_k.*/\1/'); k=$(echo $o | sed 's/.*_k\([0-9]*\)\.o/\1/')
  if echo " $NS " | grep -q " $n " && echo " $KS " | grep -q " $k "; then
    sched=""; { [ "$n" = 64 ] || [ "$n" = 512 ] || [ "$n" = 1024 ]; } && sched="-mllvm -amdgpu-sched-strategy=max-ilp"      # (as the Makefile's SCHED_64 / SCHED_512 / SCHED_1024)
    /opt/rocm/bin/hipcc $FLAGS $sched "$@" -DWOFDM_TU_N=$n -DWOFDM_TU_K=$k -c wofdm_kernel.hip -o ../../ab/obj_$name/$o &
    objs="$objs ../../ab/obj_$name/$o"
  else
    objs="$objs $o"
  fi
done
/opt/rocm/bin/hipcc $FLAGS "$@" -c wofdm_abi.hip -o ../../ab/obj_$name/wofdm_abi.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab/lib_$name.so $objs ../../ab/obj_$name/wofdm_abi.o
echo built ab/lib_$name.so
