"""GPU box: does any kernel read LDS or registers it did not write?  Between all ABI calls of a small train step a debug kernel
(tools/micro/scribble.hip, compiled here with hipcc) fills every CU's LDS and VGPRs v30..v249 with a pattern; the step's outputs must
not depend on the pattern.  (Two processes time-slicing one GPU see each other's leftovers; one process sees only its own.)"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learned_hologram_gan_amd import hip_ops, native
from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/libscribble.so"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(here, "micro", "scribble.hip"), "-o", so], check=True)
scr = ctypes.CDLL(so)
scr.scribble.argtypes = [ctypes.c_uint, ctypes.c_void_p]

PATTERN = [None]
COUNT = [0]
real = native.load()


class Proxy:
    def __getattr__(self, name):
        fn = getattr(real, name)
        if not name.startswith("lhg_") or PATTERN[0] is None:
            return fn

        def wrapped(*a):
            if PATTERN[0] is not None and torch.cuda.is_available():
                rc = scr.scribble(PATTERN[0], torch.cuda.current_stream().cuda_stream)
                assert rc < 1000, rc
                COUNT[0] += 1
            return fn(*a)
        return wrapped


native._lib = Proxy()

dev = "cuda:0"
rows = cols = int(os.environ.get("PROBE_SIZE", "64"))
torch.manual_seed(5)
stack = torch.linspace(-4e-4, 0.0, 21)[:-1][:8]
W = watermelon(filter_radius_coefficient=0.45, pad_size=rows // 2, distance_stack=stack, input_shape=(1, 4, rows, cols))
W.generator.to(dev).train(); W.discriminator.to(dev).train()
W.configure(1, 0.0, 1, 1e-3, 0.1, 1e-3, 1e-3, 1, 10)
g = torch.Generator().manual_seed(91)
x = (torch.rand((2, 4, rows, cols), generator=g).to(dev), torch.rand((2, 3, rows, cols), generator=g).to(dev), torch.rand((2, 3, rows, cols), generator=g).to(dev),
     torch.tensor([5, 2]), [torch.tensor([0.3, 0.8]).view(2, 1, 1, 1).to(dev)])
grabbed = []
for opt in (W._opt_D, W._opt_G):
    def step(grad_scale=1.0, opt=opt):
        hip_ops.join_side_stream(); torch.cuda.synchronize(); grabbed.append(opt.flat.grad.detach().clone())
    opt.step = step


def run(pat):
    PATTERN[0] = pat
    out = W.train_step(*x)
    torch.cuda.synchronize()
    PATTERN[0] = None
    return {"POH": out["POH"].clone(), "hat_amps": out["hat_amps"].clone(), "target_amps": out["target_amps"].clone(), "G_loss": out["G_loss"].clone(),
            "D_loss": out["D_loss"].clone(), "gradD": grabbed[-2], "gradG": grabbed[-1]}


run(None); run(None)  # first sight of every geometry, then a reference without the debug kernel
base = run(None)
names = {0x0: "zeros", 0x7fc00000: "NaN", 0x3f800000: "1.0", 0x7f7fffff: "FLT_MAX", 0xdeadbeef: "deadbeef"}
bad = 0
for pat, nm in names.items():
    COUNT[0] = 0
    r = run(pat)
    diffs = [k for k in base if not torch.equal(base[k], r[k]) and not (torch.isnan(base[k]).all() and torch.isnan(r[k]).all())]
    print("pattern %-8s (%d debug launches): %s" % (nm, COUNT[0], "identical" if not diffs else "DIFFERENT: " + ", ".join(
        "%s (%d elements, max |d| %.3g)" % (k, int((base[k] != r[k]).sum()), float((base[k] - r[k]).abs().nan_to_num(nan=1e30).max())) for k in diffs)), flush=True)
    bad += bool(diffs)
again = run(None)
print("without the debug kernel again:", "identical" if all(torch.equal(base[k], again[k]) for k in base) else "DIFFERENT")
sys.exit(1 if bad else 0)
