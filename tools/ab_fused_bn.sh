cd $GRAFT_REPO_ROOT
run() { env $1 python3 bench.py --steps 20 --warmup 5 --cpu-baseline 0 --secondary 0 --other-modes 0 --cli-default 0 --profile-steps 1 $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['host_ms_per_step'])"; }
for r in 1 2; do
  for cfg in "LHG_FUSED_BN=1" "LHG_FUSED_BN=0"; do
    echo "[bf16 $cfg] $(run $cfg '--dtype bf16')"
    echo "[192^2 $cfg] $(run $cfg '--rows 192 --cols 192 --pad 160')"
    echo "[96^2 $cfg] $(run $cfg '--rows 96 --cols 96 --pad 16')"
  done
done
