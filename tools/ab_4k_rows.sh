#!/bin/bash
# GPU box: interleaved A/B of the 4K frame (bench.py --mode infer: generator forward + 8-plane propagate) under environment switches.
# usage: tools/ab_4k_rows.sh "VAR=a" "VAR=b" ...
cd $GRAFT_REPO_ROOT
for r in 1 2; do for cfg in "$@"; do
 t=$(env $cfg python3 bench.py --mode infer --rows 2160 --cols 3840 --pad 72 --batch 1 --planes 8 --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
 echo "[$cfg] $t"
done; done
