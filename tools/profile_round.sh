#!/bin/bash
# GPU box: rocprofv3 kernel stats of the default bench + the two PMC passes behind roofline.traffic.
# usage: tools/profile_round.sh <tag e.g. r02> <git sha>
tag=$1; sha=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${tag}_prof
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o trace -- python3 bench.py --steps 6 --warmup 3 --cpu-baseline 0 --secondary 0 --other-modes 0 --cli-default 0 --profile-steps 1 > $out/kt.log 2>&1
python3 tools/steady_profile.py $(find $out/kt -name "*kernel_trace.csv" | head -1) 4 $out/${tag}_kernel_steady.csv > $out/${tag}_kernel_steady.txt
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
rm -f $(find $out/kt -name "*kernel_trace.csv")
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -o t -- python3 bench.py --steps 3 --warmup 3 --cpu-baseline 0 --secondary 0 --other-modes 0 --cli-default 0 --profile-steps 1 > $out/pmc_$c.log 2>&1
done
python3 tools/pmc_traffic.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/${tag}_pmc_traffic.json 0.4 $sha
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/kt
head -30 $out/${tag}_kernel_steady.txt
