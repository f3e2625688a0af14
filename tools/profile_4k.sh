#!/bin/bash
# GPU box: steady-state kernel profile + PMC traffic of the 4K leg (BASELINE configs[3]: 3840x2160 bs=1 generator forward + 8-plane propagate).
# usage: tools/profile_4k.sh <tag e.g. r04> <git sha>      -> gpurun_out/<tag>_prof4k/{<tag>_4k_kernel_steady.txt,.csv, <tag>_4k_pmc_traffic.json}
tag=$1; sha=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${tag}_prof4k
rm -rf $out; mkdir -p $out
ARGS="--mode infer --rows 2160 --cols 3840 --pad 72 --batch 1 --planes 8 --steps 4 --warmup 3"
rocprofv3 --kernel-trace --output-format csv -d $out/kt -o trace -- python3 bench.py $ARGS > $out/kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -o t -- python3 bench.py $ARGS > $out/pmc_$c.log 2>&1
done
python3 tools/profile_4k_summary.py $out $tag $sha 4
rm -rf $out/kt $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
head -24 $out/${tag}_4k_kernel_steady.txt
