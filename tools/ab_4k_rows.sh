cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in 4 1 3; do
 t=$(LHG_ASM_ROWS_MIN_WG=$v python3 bench.py --mode infer --rows 2160 --cols 3840 --pad 72 --batch 1 --planes 8 --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
 echo "min_wg=$v $t"
done; done
