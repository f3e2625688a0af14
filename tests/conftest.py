import os
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=False)
        return cache[name]

    return load


@pytest.fixture(scope="session")
def oracle_full_step():
    """BASELINE configs[1] at full size through the CPU oracle, computed once per session: 384x384, batch 4, pad 320 (1024^2 FFTs),
    20-plane stack, one critic update with the gradient penalty, generator loss / backward, both Adam steps (oracle/step.py).
    Returns (inputs, outputs); shared by the fp32 and the bf16-mode parity tests."""
    from oracle import seeded, step

    torch.set_num_threads(min(16, os.cpu_count() or 1))
    rows = cols = 384
    pad, coef, B = 320, 0.45, 4
    stack = torch.linspace(-4e-4, 0.0, 21)[:-1]
    rgbd, tamp, tphs = seeded.smooth_batch(B, rows, cols, seed=51)
    idx = torch.tensor([17, 3, 11, 6])
    alphas = [torch.tensor([0.2, 0.9, 0.55, 0.4]).view(B, 1, 1, 1)]
    st = step.make_state(rows, cols, pad, coef, stack, seeded.generator_state_dict(), seeded.critic_state_dict())
    ref = step.train_step(st, rgbd, tamp, tphs, step.LossWeights(d_ratio=1), idx, alphas, capture=True)
    cfg = dict(rows=rows, cols=cols, pad=pad, coef=coef, stack=stack, rgbd=rgbd, tamp=tamp, tphs=tphs, idx=idx, alphas=alphas)
    return cfg, ref


@pytest.fixture(scope="session")
def oracle_full_step_fp64(oracle_full_step):
    """The same full-size step evaluated in FLOAT64 (the fp32-defined constants w / H / mask are cast, not rebuilt): the truth against
    which both fp32 evaluations — the CPU oracle's (= the reference's arithmetic) and the GPU's — are measured."""
    from oracle import optics, seeded, step

    cfg, _ = oracle_full_step
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    st32 = step.make_state(cfg["rows"], cfg["cols"], cfg["pad"], cfg["coef"], cfg["stack"], seeded.generator_state_dict(), seeded.critic_state_dict())
    dbl = lambda sd: {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}  # noqa: E731
    o = st32.o
    o64 = optics.Optics(o.rows0, o.cols0, o.pad_r, o.pad_c, o.rows, o.cols, o.w.double(), o.mask.double())
    st = step.TrainState(o64, st32.H_fixed.to(torch.complex128), st32.H_stack.to(torch.complex128),
                         step.nets.as_parameters(dbl(seeded.generator_state_dict())), step.nets.as_parameters(dbl(seeded.critic_state_dict())))
    return step.train_step(st, cfg["rgbd"].double(), cfg["tamp"].double(), cfg["tphs"].double(), step.LossWeights(d_ratio=1), cfg["idx"],
                           [a.double() for a in cfg["alphas"]], capture=True)


RECORD_TAG = "r05"  # prefix of the measurement records the GPU tests append to gpurun_out/ (tools/collect_records.py moves them to profiles/)


def record_path(name: str) -> str:
    """gpurun_out/<RECORD_TAG>_<name> (LHG_RECORD_DIR overrides the directory)."""
    d = os.environ.get("LHG_RECORD_DIR") or os.path.join(REPO, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    return os.path.join(d, f"{RECORD_TAG}_{name}")


def record_stamp() -> dict:
    """What ties a record line to the build that produced it: the hash of the kernel sources (the GPU box has no git) and the ABI version.
    tools/collect_records.py refuses lines whose hash is not the committed sources'."""
    import bench
    from learned_hologram_gan_amd import native

    return {"kernel_src_sha16": bench.kernel_source_sha16(), "abi": native.ABI_VERSION}


def rel_err(a, b):
    """max|a-b| / max|b| — the fp32 parity measure used throughout (north_star: 1e-4).  A MAX-NORM ratio (the largest deviation over
    the largest reference magnitude), not an element-wise relative error: small elements are held to the tensor's scale."""
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    denom = b.abs().max().clamp_min(1e-30)
    return ((a - b).abs().max() / denom).item()


def phase_err(a, b):
    """max |e^{ia} - e^{ib}| — phases compared modulo 2*pi (SURVEY §7 hard part ii)."""
    return (torch.exp(1j * a) - torch.exp(1j * b)).abs().max().item()
