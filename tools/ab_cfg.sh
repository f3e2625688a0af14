#!/bin/bash
# GPU box: A/B/C... of whole environment settings on the same box, interleaved.  usage: tools/ab_cfg.sh rounds "VAR=a VAR2=b" "VAR=c" ...
# prints ms/step, the gather-GEMM / weight-gradient kernel ms per step (isolated pass) per setting
n=$1; shift
run() { env $1 python bench.py --steps 16 --warmup 6 --cpu-baseline 0 --secondary 0 --other-modes 0 --cli-default 0 --profile-steps 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; i=r.get('isolated',{})
print('$1 | ms/step', d['ms_per_step'], '| gg ms', r['kernel_ms_per_step'], 'iso', i.get('kernel_ms_per_step'), '| wg ms', r['wgrad_kernel']['kernel_ms_per_step'], 'iso', i.get('wgrad_kernel',{}).get('kernel_ms_per_step'), 'iso TF', i.get('wgrad_kernel',{}).get('achieved'))"; }
for i in $(seq $n); do for cfg in "$@"; do run "$cfg" || exit 1; done; done
