#!/bin/bash
# Builds A/B variants of the HIP library OUT OF TREE (here in the build container -- hipcc cross-compiles -- or on the GPU box): one copy of
# csrc/ per flag set under tools/scratch/ab_<tag>_<n>/ (not tracked, but it travels with gpurun).  The shipped .so is never touched.
# usage: tools/ab_build.sh <tag> "flags1" "flags2" ...        ("" = the shipped flags);  then on the GPU box: tools/ab_run.sh <tag> <config> bench|kstat
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
n=0
for flags in "$@"; do
  n=$((n+1))
  d=$root/tools/scratch/ab_${tag}_$n
  rm -rf $d; mkdir -p $d/based_on_point_cloud_partitions_amd/csrc $d/include
  cp $root/based_on_point_cloud_partitions_amd/csrc/*.hip $root/based_on_point_cloud_partitions_amd/csrc/*.h $root/based_on_point_cloud_partitions_amd/csrc/Makefile $d/based_on_point_cloud_partitions_amd/csrc/
  cp $root/include/*.h $d/include/
  echo "$flags" > $d/flags.txt
  make -C $d/based_on_point_cloud_partitions_amd/csrc -j8 EXTRA="$flags" > $d/build.log 2>&1 || { echo "build failed: [$flags]"; tail -3 $d/build.log; continue; }
  rm -f $d/based_on_point_cloud_partitions_amd/csrc/*.o
  echo "built $n: [$flags]"
done
