#!/bin/bash
# GPU box: where the strip gather-GEMM's time goes on the short-K layers (LHG_GG_PRIO timing ablations, WRONG results on purpose):
# 1 as shipped; 20 no scale / split in the producers; 21 weight tile not stored; 22 strip not stored; 23 no per-step barriers; 24 no epilogue; 25 three K steps only
cd $GRAFT_REPO_ROOT
for layer in "64 64 384" "128 64 384" "128 128 192"; do
  set -- $layer
  for v in ${VARIANTS:-5 7 8 11}; do
    line="$1>$2@$3 variant $v:"
    for a in 1 20 21 22 23 24 25; do
      t=$(LHG_AUTOTUNE=0 LHG_GGS_VARIANT=$v LHG_GG_PRIO=$a python3 tools/time_layer.py $1 $2 $3 3 1 fp32_split_f16 20 2>/dev/null | tail -1)
      line="$line  [$a] $t"
    done
    echo "$line"
  done
done
