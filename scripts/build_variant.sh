#!/bin/bash
# Builds a variant of the library for scripts/gpu_ab_multi.sh:  scripts/build_variant.sh NAME [-DLV_FOO=1 ...]
# (or NAME --ref <git-ref>: the library of another commit).  Output: scripts/ab/NAME.so (git-ignored, travels with gpurun).
set -eu
ROOT=$(cd $(dirname $0)/.. && pwd)
NAME=$1; shift
B=/tmp/lv_variant_$NAME
rm -rf $B; mkdir -p $B
if [ "${1:-}" = "--ref" ]; then
  git -C $ROOT archive $2 limo-velo_amd/csrc include | tar -x -C $B
  shift 2
else
  mkdir -p $B/limo-velo_amd; cp -r $ROOT/limo-velo_amd/csrc $B/limo-velo_amd/; cp -r $ROOT/include $B/
  rm -f $B/limo-velo_amd/csrc/*.o
fi
make -s -j8 -C $B/limo-velo_amd/csrc EXTRA="$*" 2>&1 | grep -E "error" || true
mkdir -p $ROOT/scripts/ab
cp $B/limo-velo_amd/liblimovelo_hip.so $ROOT/scripts/ab/$NAME.so
ls -la $ROOT/scripts/ab/$NAME.so
