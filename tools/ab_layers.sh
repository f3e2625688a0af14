#!/bin/bash
# GPU box: per-geometry GEMM tables (tools/layer_table.py) of the default bench under two environment settings, for the rows that match a filter.
# usage: tools/ab_layers.sh "VAR=a" "VAR=b" [awk-filter on column M, default: $2<=9216]
cd $GRAFT_REPO_ROOT
for cfg in "$1" "$2"; do
  f=gpurun_out/layers_$(echo $cfg | tr '= ' '__').csv
  rm -f $f
  env $cfg LHG_PROFILE_LOG=$f python3 bench.py --steps 4 --warmup 3 --cpu-baseline 0 --secondary 0 --other-modes 0 --cli-default 0 --profile-steps 4 > /dev/null 2>&1
  echo "== $cfg"
  python3 tools/layer_table.py $f 6 | awk 'NR==1 || ($1==0 && $2<=9216)'
done
