#!/bin/bash
# GPU box: matrix-pipe occupancy (tools/pmc_mfma.sh) of the layers the round's records quote -> gpurun_out/<tag>_pmc_mfma.jsonl
# usage: tools/pmc_mfma_round.sh <tag e.g. r04>
tag=$1
cd $GRAFT_REPO_ROOT
out=gpurun_out/${tag}_pmc_mfma.jsonl; rm -f $out
run() { bash tools/pmc_mfma.sh "$@" | tail -1 >> $out; }
run fwd_512_256_96 gg 512 256 96 3 1 fwd 12
run wgrad_256_512_96 wg6_kernel 256 512 96 3 1 wgrad 12
run wgrad_128_128_192 wg6_kernel 128 128 192 3 1 wgrad 12
run wgrad_64_64_384 wg6_kernel 64 64 384 3 1 wgrad 12
run wgrad_stride2_512_1024_96 wg6_kernel 512 1024 96 3 2 wgrad 12
run dgrad_stride2_1024_512 gg 512 1024 96 3 2 dgrad 12
run fwd_64_64_384 gg 64 64 384 3 1 fwd 12
run fwd_1x1_256_128_192 gg 256 128 192 1 1 fwd 12
run fwd_1024_1024_24 gg 1024 1024 24 3 1 fwd 12   # the split-K launch of the UNet bottleneck (three K ranges)
cat $out
