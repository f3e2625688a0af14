for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600+i)) tests/_dist_gpu_worker.py overlap 2>/dev/null | grep '"rank"' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['rank'], 'err %.2e' % r['err'], r['local_repeatable'], r['ranks_agree'], r['forward_repeats'], r['forward'] if not r['forward_repeats'] else '', r['recompute_notes'], r['diff_pass0_vs_pass2'][:1], r['diff_reduced_vs_mean'][:1])"
done
