#!/bin/bash
# GPU box: the four combinations of two environment switches (0 / 1) on the same box, interleaved.  usage: tools/ab_env2.sh VAR1 VAR2 [rounds]
n=${3:-2}
run() { env $1=$2 $3=$4 python bench.py --steps 20 --warmup 6 --cpu-baseline 0 --secondary 0 --other-modes 0 --cli-default 0 --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1=$2 $3=$4', d['ms_per_step'])"; }
for i in $(seq $n); do for a in 0 1; do for b in 0 1; do run $1 $a $2 $b || exit 1; done; done; done
