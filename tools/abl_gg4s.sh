#!/bin/bash
# GPU box: timing ablations of the strip gather-GEMM (gg4s_kernel, LHG_GG_PRIO 20..23: WRONG results on purpose) on three layers.
# 1 = the kernel as shipped; 20 no scale / split in the producers; 21 weight tile not stored; 22 strip not stored; 23 no per-step barriers
cd $GRAFT_REPO_ROOT
for layer in "512 256 96" "128 128 192" "64 64 384"; do
  set -- $layer
  line="$1>$2@$3 (variant 8 = 128x128 strips):"
  for a in 1 20 21 22 23; do
    t=$(LHG_AUTOTUNE=0 LHG_GGS_VARIANT=8 LHG_GG_PRIO=$a python3 tools/time_layer.py $1 $2 $3 3 1 fp32_split_f16 20 2>/dev/null | tail -1)
    line="$line  [$a] $t"
  done
  echo "$line"
done
