#!/bin/bash
# GPU box: forward time of the step's 3x3 stride-1 layers per forced gather-GEMM variant (LHG_GGS_VARIANT is read once per process)
# usage: tools/gg_variant_sweep.sh "0 3 8 10 11"
cd $GRAFT_REPO_ROOT
for layer in "64 64 384" "128 64 384" "64 128 192" "128 128 192" "256 128 192" "128 256 96" "256 256 96" "512 256 96" "256 512 96" "512 512 48" "1024 512 48" "1024 1024 24"; do
  set -- $layer
  line="$1>$2@$3:"
  for v in $VARIANTS; do
    t=$(LHG_AUTOTUNE=0 LHG_GGS_VARIANT=$v python3 tools/time_layer.py $1 $2 $3 3 1 fp32_split_f16 20 2>/dev/null | tail -1)
    line="$line  v$v $t"
  done
  echo "$line"
done
