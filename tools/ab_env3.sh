#!/bin/bash
# GPU box: interleaved A/B of the train step with environment switches.  usage: tools/ab_env3.sh "<VAR=val ...>" ["<VAR=val ...>" ...]   (each argument = one configuration; "" = defaults)
# Two rounds over all configurations on the same box; prints ms_per_step / host_ms_per_step of each run.
for round in 1 2; do
  for cfg in "$@"; do
    r=$(env $cfg python3 bench.py --steps 20 --warmup 5 --cpu-baseline 0 --secondary 0 --other-modes 0 --cli-default 0 --profile-steps 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['host_ms_per_step'])")
    echo "[$cfg] $r"
  done
done
