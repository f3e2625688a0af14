#!/bin/bash
# GPU box: the bf16 storage mode's step with / without the host-side measures (collector deferral, batched re-pack), interleaved.
run() { env LHG_DEFER_GC=$1 LHG_BATCHED_REPACK=$1 python bench.py --dtype bf16 --steps 20 --warmup 12 --cpu-baseline 0 --secondary 0 --other-modes 0 --cli-default 0 --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('host measures=$1', d['ms_per_step'])"; }
for i in 1 2; do run 0 && run 1 || exit 1; done
