#!/bin/bash
# GPU box: forward time of 3x3 stride-1 layers per forced strip variant (round 5: 12 / 13 = two taps per barrier, 14 / 15 = swizzled unpadded LDS rows)
# usage: VARIANTS="5 7 8 12 14" tools/gg_tps_sweep.sh
cd $GRAFT_REPO_ROOT; export LHG_GG_EXPERIMENTAL=1
for layer in "64 64 384" "128 64 384" "64 128 192" "128 128 192" "256 128 192" "128 256 96" "256 256 96" "512 512 48"; do
  set -- $layer
  line="$1>$2@$3:"
  for v in ${VARIANTS:-5 7 8 12 13 14 15}; do
    t=$(LHG_AUTOTUNE=0 LHG_GGS_VARIANT=$v python3 tools/time_layer.py $1 $2 $3 3 1 fp32_split_f16 20 2>/dev/null | tail -1)
    line="$line  v$v $t"
  done
  echo "$line"
done
