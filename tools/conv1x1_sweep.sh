#!/bin/bash
# GPU box: the step's 1x1 convolutions (shortcut convs) per forced gather-GEMM variant
cd $GRAFT_REPO_ROOT
for layer in "64 128 384" "128 64 384" "128 256 192" "256 128 192" "256 512 96" "512 256 96" "512 1024 48" "1024 512 48" "1024 1024 24"; do
  set -- $layer
  line="1x1 $1>$2@$3:"
  for v in 0 1 2 3 4 9; do
    t=$(LHG_AUTOTUNE=0 LHG_GGS_VARIANT=$v python3 tools/time_layer.py $1 $2 $3 1 1 fp32_split_f16 20 2>/dev/null | tail -1)
    line="$line | v$v $t"
  done
  echo "$line"
done
