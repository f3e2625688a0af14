"""CPU-side checks of the product package: the C-ABI library loads and exports every declared
symbol, host-built constants equal the reference's bit for bit, modules carry the reference's
checkpoint schema, entry points keep their flags, and the ops refuse to run without the GPU
(no compute is executed here)."""

import ctypes
import os
import subprocess
import sys

import pytest
import torch

from conftest import REPO, rel_err

WL = torch.tensor([638e-9, 520e-9, 450e-9])


def test_library_exports_declared_abi():
    from learned_hologram_gan_amd import native

    declared = native.declared_symbols()
    assert len(declared) >= 25 and set(declared) == set(native._SIGNATURES)
    lib = native.load()
    raw = ctypes.CDLL(native.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert lib.lhg_abi_version() == native.ABI_VERSION == 10


def test_host_side_size_functions_and_supported_lengths():
    """Pure host functions of the ABI (no kernel launch): FFT table sizes follow the transform route (direct: 2n floats; Bluestein:
    m twiddles + n chirp values + m filter values, m the convolution length >= 2n - 1) and agree with the Python mirror of the supported
    lengths; packed-weight sizes follow the precision mode (the fp16-split panels carry max|w| in 16 bytes behind them)."""
    from learned_hologram_gan_amd import asm_ops, native

    lib = native.load()
    for n, direct in ((1024, True), (2304, True), (4096, True), (832, True), (2800, True), (198, True), (100, True),
                      (4976, False), (142, False), (87, False), (2362, False), (8000, False)):
        assert asm_ops.smooth_extent(n) == direct and asm_ops.supported_extent(n)
        floats = int(lib.lhg_fft_table_floats(n))
        if direct:
            assert floats == 2 * n
        else:
            m = 64
            while m < 2 * n - 1:
                m *= 2
            if m > 4096:  # above 4096: the shortest 2^a 3^b length (a multiple of 64) that fits 12 inputs per thread of the line's workgroup
                threads = lambda L: 256 if L <= 4096 else (512 if L <= 8192 else 1024)  # noqa: E731
                smooth = [c for c in (t << a for t in (3, 9, 27, 81, 243) for a in range(20)) if 2 * n - 1 <= c < m and c % 64 == 0 and c <= 12 * threads(c)]
                m = min(smooth + [m])
            assert floats == 2 * (2 * m + n), (n, m, floats)
    assert int(lib.lhg_fft_table_floats(4976)) == 2 * (2 * 10368 + 4976)  # the 4K frame with the CLI's pad: 2^7 3^4, not 16384
    assert not asm_ops.supported_extent(9350) and not asm_ops.supported_extent(8)
    assert lib.lhg_default_conv_precision() == 4 or "LHG_CONV_PRECISION" in __import__("os").environ
    mode = lib.lhg_get_conv_precision()
    try:
        elems = 9 * 128 * 64
        for prec, floats in ((0, elems), (1, elems), (2, elems * 3 // 2), (3, elems), (4, elems + 4)):
            assert lib.lhg_set_conv_precision(prec) == 0 and int(lib.lhg_packed_weight_floats(9, 128, 64)) == floats
    finally:
        lib.lhg_set_conv_precision(mode)


def test_split_k_rule_and_statistics_row_bound_are_functions_of_the_geometry():
    """Pure host functions of ABI 10 (no kernel launch).  lhg_gather_gemm_splitk_floats: the UNet bottleneck (2304 pixels x 1024 -> 1024
    channels, 3x3: 144 tiles of 128 x 128, 288 K steps) splits into three ranges of whole 32-channel chunks, its slabs cover the padded
    pixel axis of the strip kernels rounded to the largest tile; launches that fill the chip, short K axes and the other arithmetic
    modes do not split.  lhg_conv2d_stats_rows_bound covers the finest grouping (one row per 16 padded pixels: the split-K finish kernel)."""
    from learned_hologram_gan_amd import native

    lib = native.load()
    assert lib.lhg_get_conv_precision() == native_precision_f16_split()
    M, Mp = 4 * 24 * 24, 4 * 24 * 26
    assert lib.lhg_gather_gemm_splitk_floats(M, Mp, 1024, 1024, 9) == 3 * 2560 * 1024      # 144 tiles -> 3 ranges (11 + 11 + 10 chunks)
    assert lib.lhg_gather_gemm_splitk_floats(M, Mp, 512, 1024, 9) == 4 * 2560 * 512        # 72 tiles -> capped at 4 ranges
    assert lib.lhg_gather_gemm_splitk_floats(M, 0, 1024, 1024, 1) == 0                     # 1x1: 32 K steps, too short
    assert lib.lhg_gather_gemm_splitk_floats(4 * 48 * 48, 4 * 48 * 50, 512, 512, 9) == 0   # 288 tiles: the chip is full enough
    assert lib.lhg_gather_gemm_splitk_floats(4 * 384 * 384, 4 * 384 * 386, 64, 64, 9) == 0
    assert lib.lhg_gather_gemm_splitk_floats(192, 0, 64, 2080, 1) == 4 * 256 * 64          # a long 1x1 K axis on a small image: 65 chunks -> 4 ranges
    lib.lhg_set_conv_precision(0)  # exact fp32 kernels: no split
    try:
        assert lib.lhg_gather_gemm_splitk_floats(M, Mp, 1024, 1024, 9) == 0
    finally:
        lib.lhg_set_conv_precision(lib.lhg_default_conv_precision())
    for N, Ho, Wo in ((4, 384, 384), (4, 24, 24), (1, 17, 23), (1, 2160, 3840)):
        mp = N * Ho * (Wo + 2)
        bound = lib.lhg_conv2d_stats_rows_bound(N, Ho, Wo)
        assert bound >= (mp + 255) // 256 * 16 and bound >= (mp + 31) // 32 + 4 and bound <= mp // 16 + 64


def native_precision_f16_split():
    return 4  # LHG_PRECISION_F32_SPLIT_F16 (include/lhg_hip.h)


def test_library_is_gfx950_code_object():
    from learned_hologram_gan_amd import native

    blob = open(native.LIB_PATH, "rb").read()
    assert b"gfx950" in blob


def test_library_contains_no_packed_fp32_instructions():
    """packed-fp32 VALU instructions next to MFMA workgroups of another queue made the angular-spectrum operator non-
    repeatable on MI355X; the build switches the feature off for every kernel file (__graft_entry__.NO_PACKED_FP32).  Disassemble the
    seven gfx950 code objects of the built library and check that none came back (and that the MFMA kernels are there)."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import device_isa_scan

    got = device_isa_scan.scan(r"v_pk_(fma|mul|add)_f32|v_mfma_f32_32x32x16_f16")
    assert got["code_objects"] == 8 and got["instructions"] > 100000, got
    assert got["matches"].get("v_mfma_f32_32x32x16_f16", 0) > 0, got
    assert not [k for k in got["matches"] if k.startswith("v_pk_")], got


def test_ops_refuse_cpu_tensors():
    from learned_hologram_gan_amd import hip_ops, native
    from learned_hologram_gan_amd.angular_spectrum_method import bandLimitedAngularSpectrumMethod_for_single_fixed_distance as Fx

    with pytest.raises(native.NativeLibraryError):
        hip_ops.ToNHWC.apply(torch.rand(1, 3, 4, 4), 32)
    fx = Fx(48, 48, 8, 0.45, 3.74e-6, WL, False, False, torch.tensor([1e-3]))
    with pytest.raises(native.NativeLibraryError):
        fx.propagate_POH2AP_forward(torch.rand(1, 3, 48, 48))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from learned_hologram_gan_amd import native

    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(native.NativeLibraryError, match="no CPU"):
        native.load()


def test_wrong_result_switches_of_the_gather_gemm_are_gone_and_refused(monkeypatch):
    """LHG_GG_PRIO >= 10 used to select timing ablations inside the shipped gather-GEMM kernels (no stores, no
    barriers, no epilogue: wrong results on purpose).  The kernels no longer contain them; a stale value makes native.load() fail
    loudly (the C side refuses the launch as well), and the kernel sources carry no run-time ablation branch."""
    import os

    from learned_hologram_gan_amd import native

    for bad in ("22", "10", "25", "-1", "x"):
        monkeypatch.setattr(native, "_lib", None)
        monkeypatch.setenv("LHG_GG_PRIO", bad)
        with pytest.raises(native.NativeLibraryError, match="LHG_GG_PRIO"):
            native.load()
    for ok in ("0", "1", "2"):
        monkeypatch.setattr(native, "_lib", None)
        monkeypatch.setenv("LHG_GG_PRIO", ok)
        native.load()
    monkeypatch.delenv("LHG_GG_PRIO")
    monkeypatch.setattr(native, "_lib", None)
    native.load()
    csrc = os.path.join(os.path.dirname(native.__file__), "csrc")
    for name in ("gg3s_kernel.inc", "gg4s_kernel.inc", "gg_epilogue.inc"):
        src = open(os.path.join(csrc, name)).read()
        code = "\n".join(ln.split("//")[0] for ln in src.splitlines())  # comments may tell the history
        assert "abl !=" not in code and "abl ==" not in code and "int abl" not in code, name
        for n in range(10, 26):
            assert f"prio == {n}" not in src and f"prio != {n}" not in src and f"prio >= {n}" not in src, (name, n)
    engine = open(os.path.join(csrc, "conv_engine.hip")).read()
    assert "prio >= 0 && prio <= 2" in engine  # the launch itself refuses anything else


def test_constants_follow_reference_op_order(golden):
    from learned_hologram_gan_amd.angular_spectrum_method import (
        bandLimitedAngularSpectrumMethod_for_multiple_distances as Mu,
        bandLimitedAngularSpectrumMethod_for_single_fixed_distance as Fx,
    )

    for tag in ("sq48", "rect32x48"):
        g = golden("constants.pt")[tag]
        r0, c0, pad, coef = g["args"]
        fx = Fx(r0, c0, pad, coef, 3.74e-6, WL, False, False, torch.tensor([1e-3]))
        mu = Mu(r0, c0, g["distances"], pad, coef, 3.74e-6, WL, False, False)
        assert (fx.samplingRowNum, fx.samplingColNum) == tuple(g["shape"])
        # same op order as the reference => equal up to the host's last-bit sqrt behaviour (MKL VML differs between
        # CPU models; bit-identical on the build container where the fixture was made)
        assert ((fx.w_grid - g["w"]).abs() <= 0.25).all() and torch.equal(fx.diffraction_limited_mask, g["mask"])
        assert (fx.H - g["H_fixed"]).abs().max() < 2e-3 and (mu.H - g["H_stack"]).abs().max() < 2e-3
        fx.set_transfer_function(g["H_fixed"])
        assert torch.equal(fx.H, g["H_fixed"]) and torch.equal(fx._H_masked, g["H_fixed"] * g["mask"])
    with pytest.raises(ValueError):
        Fx(64, 64, 0, 0.6, 3.74e-6, WL, False, False, torch.tensor([1e-3]))


def test_pad_crop_api():
    from learned_hologram_gan_amd.angular_spectrum_method import bandLimitedAngularSpectrumMethod as Base

    b = Base(32, 48, 8, 0.35, 3.74e-6, WL)
    x = torch.rand(2, 3, 32, 48)
    assert b.padding(x).shape == (2, 3, 48, 72) and torch.equal(b.cropping(b.padding(x)), x)


def test_checkpoint_schema_matches_reference(golden):
    from learned_hologram_gan_amd.watermelon_hologram.discriminator import WGANGPDiscriminator192
    from learned_hologram_gan_amd.watermelon_hologram.generator import Generator
    from oracle import seeded

    g = golden("generator_small.pt")
    G = Generator(32, 32, 16, 0.45, 3, 3.74e-6, WL, torch.tensor([1e-3]))
    assert {k: tuple(v.shape) for k, v in G.state_dict().items()} == g["key_shapes"]
    assert sum(p.numel() for p in G.parameters()) == g["n_params"] == 32440274
    res = G.load_state_dict(seeded.generator_state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    c = golden("critic_small.pt")
    D = WGANGPDiscriminator192(None, 32, False)
    assert {k: tuple(v.shape) for k, v in D.state_dict().items()} == c["key_shapes"]
    assert sum(p.numel() for p in D.parameters()) == c["n_params"] == 6301377
    # the trainer reaches into these attributes (ref: watermelon.py:219)
    assert hasattr(G.part2.propagator, "propagate_POH2Freq_forward") and hasattr(G.part1, "part1")


def test_reference_initialisation_statistics():
    from learned_hologram_gan_amd.watermelon_hologram.generator import Generator

    torch.manual_seed(0)
    G = Generator(32, 32, 16)
    sd = G.state_dict()
    w = sd["part1.part1.encoder3.1.0.convolution_layer_1.weight"]  # xavier normal: std = sqrt(2/(fan_in+fan_out))
    assert abs(w.std().item() / (2.0 / (128 * 9 + 256 * 9)) ** 0.5 - 1) < 0.02
    assert sd["part1.part1.encoder3.1.0.convolution_layer_1.bias"].abs().max() == 0
    assert (sd["part1.part1.decoder2.0.0.batch_norm_layer_1.weight"] == 1).all()
    t = sd["part1.part1.decoder1.1.weight"]  # kaiming fan_out on (Cin, Cout, 2, 2): fan_out = Cin*4
    assert abs(t.std().item() / (2.0 / (512 * 4)) ** 0.5 - 1) < 0.02
    assert (sd["part2.part1.conv_r.params"] >= 0).all()


def test_symmetric_conv_matches_conv2d():
    from learned_hologram_gan_amd.neural_network_components import ChannelWiseSymmetricConv

    m = ChannelWiseSymmetricConv()
    x = torch.rand(2, 3, 9, 11)
    ref = []
    for c, conv in enumerate((m.conv_r, m.conv_g, m.conv_b)):
        w = conv.params[conv.distance_map].view(1, 1, 3, 3)
        ref.append(torch.nn.functional.conv2d(x[:, c:c + 1], w, conv.bias, padding=1))
    assert rel_err(m(x), torch.cat(ref, 1)) < 1e-6


def test_losses_match_oracle(golden):
    from learned_hologram_gan_amd.watermelon_hologram import loss_func

    g = golden("losses_small.pt")
    assert abs(loss_func.focal_sincos_phase_gradient_loss(g["hat_phs"], g["tgt_phs"]).item() - g["focal"]) < 1e-6
    assert abs(loss_func.total_variation_loss(g["hat_amp"], g["tgt_amp"]).item() - g["tv_loss"]) < 1e-6


def test_cli_flags_kept():
    for script, flags in (("generatePOH.py", ["--img_path", "--depth_path", "--index", "--model_path", "--poh_output_path", "--samplesNum",
                                              "--sample_row_num", "--sample_col_num", "--pad_size", "--pixel_pitch", "--wave_length",
                                              "--distance", "--filter_radius_coefficient", "--propagate", "--min_distance",
                                              "--max_distance", "--num_intervals", "--output_image_dir"]),
                          ("trainingModel.py", ["--train_img_path", "--train_depth_path", "--train_amp_path", "--train_phs_path",
                                                "--validate_img_path", "--validate_depth_path", "--validate_amp_path",
                                                "--validate_phs_path", "--samplesNum", "--channlesNum", "--height", "--width",
                                                "--batch_size", "--lr_G", "--lr_D", "--epoch_num", "--save_path_G", "--save_path_D",
                                                "--loss_metrics_file", "--save_path_img"])):
        out = subprocess.run([sys.executable, os.path.join(REPO, script), "--help"], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        for f in flags:
            assert f in out.stdout, (script, f)


def test_bin_datasets(tmp_path):
    import numpy as np

    from learned_hologram_gan_amd.watermelon_hologram.data_loader import dataloaderImgDepth, dataloaderImgDepthAmpPhs

    shape = (3, 3, 8, 8)
    paths = {}
    for i, k in enumerate(("img", "depth", "amp", "phs")):
        a = np.random.RandomState(i).rand(*shape).astype(np.float32)
        paths[k] = str(tmp_path / f"{k}.bin")
        a.tofile(paths[k])
        paths[k + "_a"] = a
    ds = dataloaderImgDepthAmpPhs(paths["img"], paths["depth"], paths["amp"], paths["phs"], 3, 3, 8, 8, cuda=False)
    rgbd, amp, phs = ds[1]
    assert rgbd.shape == (4, 8, 8) and torch.equal(rgbd[:3], torch.from_numpy(paths["img_a"][1]))
    assert torch.equal(rgbd[3], torch.from_numpy(paths["depth_a"][1][0])) and torch.equal(phs, torch.from_numpy(paths["phs_a"][1]))
    with pytest.raises(IndexError):
        ds[3]
    assert dataloaderImgDepth(paths["img"], paths["depth"], 3, 3, 8, 8)[2].shape == (4, 8, 8)


def test_plateau_schedule_for_fused_adam_matches_torch():
    """optim.ReduceOnPlateau (drives FusedAdam.lr in the pre-training loops) against torch's own scheduler."""
    from torch.optim.lr_scheduler import ReduceLROnPlateau

    from learned_hologram_gan_amd.optim import ReduceOnPlateau

    class _Opt:
        lr = 1e-3

    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1e-3)
    ref = ReduceLROnPlateau(opt, "min", factor=0.1, patience=4, threshold=1e-3, threshold_mode="rel", min_lr=1e-6)
    mine = ReduceOnPlateau(_Opt(), factor=0.1, patience=4, threshold=1e-3, min_lr=1e-6)
    g = torch.Generator().manual_seed(1)
    for v in [2.0, 1.5] + [1.5 + 0.1 * torch.rand((), generator=g).item() for _ in range(30)] + [1.0] + [1.2] * 11:
        ref.step(v)
        assert abs(mine.step(v) - opt.param_groups[0]["lr"]) < 1e-12


def test_metrics_two_restatements_agree():
    """The product's PSNR / SSIM (watermelon.py, device tensor ops) against the oracle's independent float64 restatement."""
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import psnr, ssim
    from oracle import losses

    g = torch.Generator().manual_seed(5)
    x = torch.rand((3, 3, 32, 48), generator=g)
    y = (x + 0.1 * torch.rand((3, 3, 32, 48), generator=g)).clamp(0, 1.3)
    assert abs(ssim(x, y).item() - losses.ssim(x, y).item()) < 1e-5
    assert abs(psnr(x, y).item() - losses.psnr(x, y).item()) < 1e-5


def _bin_dataset(tmp_path, N=11, C=3, H=8, W=8):
    import numpy as np

    from learned_hologram_gan_amd.watermelon_hologram.data_loader import dataloaderImgDepthAmpPhs

    rng = np.random.default_rng(0)
    paths = {}
    for k in ("img", "depth", "amp", "phs"):
        paths[k] = str(tmp_path / f"{k}.bin")
        rng.random((N, C, H, W), dtype=np.float32).tofile(paths[k])
    return dataloaderImgDepthAmpPhs(paths["img"], paths["depth"], paths["amp"], paths["phs"], N, C, H, W, cuda=False), paths


def _same(ref, got):
    return len(ref) == len(got) and all(torch.equal(a, b) for r, g in zip(ref, got) for a, b in zip(r, g))


def test_prefetch_loader_visits_samples_in_dataloader_order(tmp_path):
    """N3: batches of PrefetchLoader == batches of the reference's DataLoader (same global RNG state), incl. the ragged tail."""
    from torch.utils.data import DataLoader

    from learned_hologram_gan_amd.watermelon_hologram.data_loader import PrefetchLoader, dataloaderAmpPIPhs

    ds, paths = _bin_dataset(tmp_path)
    for shuffle, drop in ((False, False), (True, True), (True, False)):
        torch.manual_seed(5)
        ref = list(DataLoader(ds, batch_size=4, shuffle=shuffle, drop_last=drop))
        torch.manual_seed(5)
        loader = PrefetchLoader(ds, 4, shuffle=shuffle, drop_last=drop)
        got = list(loader)
        assert _same(ref, got) and len(loader) == len(ref)
    assert tuple(got[-1][0].shape) == (3, 4, 8, 8)  # RGB + depth channel 0
    ds2 = dataloaderAmpPIPhs(paths["amp"], paths["phs"], 11, 3, 8, 8, cuda=False)
    assert _same(list(DataLoader(ds2, batch_size=3)), list(PrefetchLoader(ds2, 3)))
    with pytest.raises(IndexError):
        ds[11]


def test_prefetch_loader_shards_like_distributed_sampler(tmp_path):
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler

    from learned_hologram_gan_amd.watermelon_hologram.data_loader import PrefetchLoader

    ds, _ = _bin_dataset(tmp_path)
    seen = []
    for rank in range(2):
        sampler = DistributedSampler(ds, 2, rank, shuffle=True, seed=7)
        sampler.set_epoch(3)
        ref = list(DataLoader(ds, batch_size=2, sampler=sampler, drop_last=True))
        loader = PrefetchLoader(ds, 2, shuffle=True, drop_last=True, rank=rank, world=2, seed=7)
        loader.set_epoch(3)
        got = list(loader)
        assert _same(ref, got)
        seen += [tuple(row.flatten()[:4].tolist()) for b in got for row in b[1]]
    assert len(seen) == 12 and len(set(seen)) == 11  # disjoint shards; 11 samples padded to 12 by wrapping around


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it."""
    import ast

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    offenders = []
    files = [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(root, "learned_hologram_gan_amd")) for f in fs if f.endswith(".py")]
    files += [os.path.join(root, f) for f in ("trainingModel.py", "generatePOH.py")] + \
             [os.path.join(root, "tools", f) for f in os.listdir(os.path.join(root, "tools")) if f.endswith(".py")]
    for path in files:
        for node in ast.walk(ast.parse(open(path).read())):
            names = [a.name for a in node.names] if isinstance(node, ast.Import) else [node.module or ""] if isinstance(node, ast.ImportFrom) else []
            if any(n == "oracle" or n.startswith("oracle.") for n in names):
                offenders.append(os.path.relpath(path, root))
    assert not offenders, offenders
    # bench.py: the oracle appears only inside cpu_baseline()
    tree = ast.parse(open(os.path.join(root, "bench.py")).read())
    for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef)]:
        uses = any(isinstance(n, ast.ImportFrom) and (n.module or "").startswith("oracle") for n in ast.walk(fn))
        assert uses == (fn.name == "cpu_baseline"), fn.name


def test_deferred_gc_keeps_the_collector_out_and_restores_it(monkeypatch):
    """hip_ops.deferred_gc (around every train step): the cyclic collector is off inside, back on afterwards — also when the step
    raises, when the caller had it off to begin with it stays off, and LHG_DEFER_GC=0 leaves it alone."""
    import gc

    from learned_hologram_gan_amd import hip_ops

    assert gc.isenabled()
    with hip_ops.deferred_gc():
        assert not gc.isenabled()
        with hip_ops.deferred_gc():  # nested (train -> train_step): the inner one finds it off and leaves it off
            assert not gc.isenabled()
        assert not gc.isenabled()
    assert gc.isenabled()
    with pytest.raises(RuntimeError):
        with hip_ops.deferred_gc():
            raise RuntimeError("step failed")
    assert gc.isenabled()
    gc.disable()
    try:
        with hip_ops.deferred_gc():
            assert not gc.isenabled()
        assert not gc.isenabled()  # the caller's choice survives
    finally:
        gc.enable()
    monkeypatch.setattr(hip_ops, "_DEFER_GC", False)
    with hip_ops.deferred_gc():
        assert gc.isenabled()


def test_pack_item_mirrors_the_header_struct():
    """native.PackItem is lhg_pack_item of include/lhg_hip.h field for field (two pointers, seven ints: 48 bytes with padding)."""
    from learned_hologram_gan_amd import native

    names = [n for n, _ in native.PackItem._fields_]
    assert names == ["w", "dst", "D0", "D1", "KH", "KW", "rows_from_d0", "rows_pad", "k_pad"]
    assert ctypes.sizeof(native.PackItem) == 48
    header = open(os.path.join(REPO, "include", "lhg_hip.h")).read()
    body = header[header.index("typedef struct lhg_pack_item {"):header.index("} lhg_pack_item;")]
    for n in names:
        assert n in body


def test_vgg19_features_schema_matches_torchvision_and_the_reference_taps():
    """N1 (perceptual loss): torchvision and its ImageNet weights cannot be downloaded here, so VALUES stay unpinned — but the layout the
    reference reads is pinned: ``torchvision.models.vgg19().features[:32]`` (configuration "E": conv3x3 + ReLU pairs, max-pools after
    2 / 4 / 8 / 12 convs) has exactly these ``<index>.weight`` / ``<index>.bias`` keys and shapes, so a torchvision ``vgg19`` or
    ``vgg19().features`` state_dict loads key for key (strict), and the tapped indices 3 / 8 / 13 / 22 / 31 (ref: loss_func.py:15, 30-47:
    ``if int(name) in self.feature_map_layers`` after ``x = layer(x)``) are the ReLU outputs relu1_2, relu2_2, relu3_2, relu4_2, relu5_2."""
    import warnings

    from torch import nn

    from learned_hologram_gan_amd.watermelon_hologram.perceptual import IMAGENET_MEAN, IMAGENET_STD, build_vgg19_features, perceptualLoss

    conv_at = {0: (64, 3), 2: (64, 64), 5: (128, 64), 7: (128, 128), 10: (256, 128), 12: (256, 256), 14: (256, 256), 16: (256, 256),
               19: (512, 256), 21: (512, 512), 23: (512, 512), 25: (512, 512), 28: (512, 512), 30: (512, 512)}
    pools = {4, 9, 18, 27}
    net = build_vgg19_features(31)
    assert len(net) == 32  # features[: max(taps) + 1]
    want = {}
    for i, layer in enumerate(net):
        if i in conv_at:
            co, ci = conv_at[i]
            assert isinstance(layer, nn.Conv2d) and layer.kernel_size == (3, 3) and layer.padding == (1, 1) and layer.stride == (1, 1)
            want[f"{i}.weight"], want[f"{i}.bias"] = (co, ci, 3, 3), (co,)
        elif i in pools:
            assert isinstance(layer, nn.MaxPool2d) and layer.kernel_size == 2 and layer.stride == 2
        else:
            assert isinstance(layer, nn.ReLU)
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == want
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pl = perceptualLoss(cuda=False)
    assert pl.feature_map_layers == [3, 8, 13, 22, 31] and pl.feature_map_layers_num == 5
    assert all(isinstance(pl.net[i], nn.ReLU) for i in pl.feature_map_layers)
    assert [sum(isinstance(pl.net[j], nn.Conv2d) for j in range(i)) for i in pl.feature_map_layers] == [2, 4, 6, 10, 14]  # relu{1..5}_2
    assert IMAGENET_MEAN == (0.485, 0.456, 0.406) and IMAGENET_STD == (0.229, 0.224, 0.225)  # transforms.Normalize of loss_func.py:40
    assert not any(p.requires_grad for p in pl.net.parameters())
    # a torchvision-style state_dict (with or without the "features." prefix, classifier keys ignored) loads strictly
    import os
    import tempfile

    sd = {"features." + k: torch.full_like(v, 0.5) for k, v in net.state_dict().items()}
    sd["classifier.0.weight"] = torch.zeros(4, 4)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "vgg19.pth")
        torch.save(sd, path)
        loaded = perceptualLoss(cuda=False, weights_path=path)
    assert loaded.pretrained and all(bool((v == 0.5).all()) for v in loaded.net.state_dict().values())


def test_precision_context_manager_restores_the_process_wide_modes():
    """hip_ops.precision(...): the GEMM mode / activation storage inside, whatever was set before restored on exit — also when the body
    raises (the switches are process-global: VERDICT r2 "what's weak" #12)."""
    from learned_hologram_gan_amd import hip_ops

    base = hip_ops.conv_precision()
    assert base == hip_ops.default_precision() and hip_ops.activation_storage() == "fp32"
    with hip_ops.precision("fp32"):
        assert hip_ops.conv_precision() == "fp32"
        with hip_ops.precision("fp32_split"):
            assert hip_ops.conv_precision() == "fp32_split"
        assert hip_ops.conv_precision() == "fp32"
    assert hip_ops.conv_precision() == base
    with pytest.raises(RuntimeError):
        with hip_ops.precision(storage="bf16"):
            assert hip_ops.activation_storage() == "bf16" and hip_ops.conv_precision() == "bf16"
            raise RuntimeError("boom")
    assert hip_ops.conv_precision() == base and hip_ops.activation_storage() == "fp32"
