#!/bin/bash
# Experiment helper: builds swift-game-engine_amd/libsge_amd_<name>.so from the working tree with extra compiler flags
# (e.g. tools/build_variant.sh prio3 -DSGE_CCD_SETPRIO=3), for tools/variant_sweep.py. The variants are git-ignored.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/swift-game-engine_amd/csrc
obj=/tmp/sge_variant_$name
mkdir -p $obj
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-fast-math"
pids=()
for f in sge_api.hip sge_host.cpp sge_ccd.hip sge_pose.hip sge_blas.hip; do
  $HIPCC $COMMON -ffp-contract=off "$@" -x hip -c $src/$f -o $obj/${f%.*}.o & pids+=($!)
done
$HIPCC $COMMON "$@" -c $src/sge_skin.hip -o $obj/sge_skin.o & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -o $root/swift-game-engine_amd/libsge_amd_$name.so $obj/*.o
echo built $root/swift-game-engine_amd/libsge_amd_$name.so
