"""GPU box: do the tiling variants of the fp16-split gather-GEMM give bit-identical results?  One child process per forced variant
(LHG_GGS_VARIANT is read once per process) runs a 3x3 forward, its input gradient, a stride-2 input gradient and a weight gradient (LHG_WG_VARIANT) on fixed data."""
import os, subprocess, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from learned_hologram_gan_amd import hip_ops as ops
    torch.manual_seed(3)
    x = torch.randn(2, 48, 40, 128, device="cuda"); w = torch.randn(128, 128, 3, 3, device="cuda") * 0.05
    gy = torch.randn(2, 48, 40, 128, device="cuda"); gy2 = torch.randn(2, 24, 20, 128, device="cuda")
    with torch.no_grad():
        y = ops.conv2d_forward_raw(x, w, None, 1)
        gx = ops.Conv2dInputGradFn.apply(gy, w, 1, 48, 40, 128)
        gx2 = ops.Conv2dInputGradFn.apply(gy2, w, 2, 48, 40, 128)
        slot = torch.zeros(128, 128, 3, 3, device="cuda")
        ops.conv2d_weight_grad_raw(x, gy, (128, 128, 3, 3), 1, slot)
    torch.cuda.synchronize()
    print(" ".join(hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:12] for t in (y, gx, gx2, slot)))
    sys.exit(0)
ref = None
for v, wv in ((2, 21), (0, 18), (1, 19), (3, 20), (4, 22), (9, 23), (5, 24), (6, 25), (7, 10), (8, 13)):
    env = dict(os.environ, LHG_AUTOTUNE="0", LHG_GGS_VARIANT=str(v), LHG_WG_VARIANT=str(wv))
    out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    ref = ref or out
    print("gather variant %2d, wgrad variant %2d: %s %s" % (v, wv, out, "" if out == ref else "  <-- differs from the first line"), flush=True)
