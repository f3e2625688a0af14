import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_graph import _trainer, _state, DEV
from oracle import seeded
ratio = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 2
batches = [seeded.smooth_batch(B, 64, 64, seed=70 + k) for k in range(3)]
idxs = [torch.tensor([5, 2]), torch.tensor([0, 7]), torch.tensor([3, 3])]
alphas = [[torch.tensor([0.3 + 0.1 * k + 0.05 * r, 0.8 - 0.1 * k]).view(B, 1, 1, 1) for r in range(ratio)] for k in range(3)]
runs = []
for graph in (False, False, True, True):
    W = _trainer(ratio, graph)
    per = []
    for (rgbd, tamp, tphs), idx, al in zip(batches, idxs, alphas):
        out = W.train_step(rgbd.to(DEV), tamp.to(DEV), tphs.to(DEV), idx, [a.to(DEV) for a in al])
        torch.cuda.synchronize()
        st, cnt = _state(W)
        per.append(({k: v.detach().clone() for k, v in out.items()}, st, cnt))
    runs.append(per)
names = ["eager0", "eager1", "graph0", "graph1"]
for a in range(4):
    for b in range(a + 1, 4):
        diffs = []
        for k in range(3):
            for key in runs[a][k][0]:
                if not torch.equal(runs[a][k][0][key], runs[b][k][0][key]):
                    diffs.append((k, key))
            for key in runs[a][k][1]:
                if not torch.equal(runs[a][k][1][key], runs[b][k][1][key]):
                    diffs.append((k, "state." + key))
        print(names[a], "vs", names[b], "first diffs:", diffs[:8], "counts", runs[a][2][2], runs[b][2][2])
