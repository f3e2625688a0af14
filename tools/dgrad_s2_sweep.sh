#!/bin/bash
# GPU box: the stride-2 input gradients of the step per forced variant (merged classes 0..9, per-class launches 10..19)
cd $GRAFT_REPO_ROOT
for layer in "512 1024 96" "128 256 192" "32 64 384"; do
  set -- $layer
  line="dgrad s2 $1<$2 @$3:"
  for v in 0 1 2 3 4 9 10 11 12 19; do
    t=$(LHG_AUTOTUNE=0 LHG_GGS_VARIANT=$v python3 tools/dbg/time_dgrad_s2.py $1 $2 $3 20 2>/dev/null | tail -1)
    line="$line | v$v $t"
  done
  echo "$line"
done
