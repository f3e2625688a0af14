#!/bin/bash
# GPU box: A/B of one environment switch on the same box, interleaved.  usage: tools/ab_env.sh VAR=a VAR=b [rounds]
a=$1; b=$2; n=${3:-2}
run() { env "$1" python bench.py --steps 20 --warmup 6 --cpu-baseline 0 --secondary 0 --other-modes 0 --cli-default 0 --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for i in $(seq $n); do run $a && run $b || exit 1; done
