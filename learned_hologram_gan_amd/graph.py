"""hipGraph capture of the inference path (eval-mode generator forward RGBD -> POH).

At batch 1 a 384x384 frame is ~60 kernel launches of 10-200 us each, so host launch overhead is a visible part of the
latency.  The whole forward is captured once into a HIP graph (through torch.cuda.CUDAGraph, which records every launch
made on the capture stream — including the ctypes launches of liblhg_hip.so, which neither allocate nor synchronise)
and replayed per frame with one host call.  Shapes are static: one GraphedGenerator per (batch, rows, cols).
"""

from __future__ import annotations

import torch


class GraphedGenerator:
    def __init__(self, generator: torch.nn.Module, example_rgbd: torch.Tensor, warmup: int = 3):
        if not example_rgbd.is_cuda:
            raise RuntimeError("GraphedGenerator needs a GPU input (no CPU fallback)")
        self.generator = generator.eval()
        self.static_in = example_rgbd.detach().clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):  # autotune every GEMM geometry, build twiddles / packed weights before capturing
                self.generator(self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        from . import hip_ops

        hip_ops.begin_graph_capture()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = self.generator(self.static_in)

    @torch.no_grad()
    def __call__(self, rgbd: torch.Tensor, clone: bool = True) -> torch.Tensor:
        if rgbd.shape != self.static_in.shape:
            raise ValueError(f"graph was captured for {tuple(self.static_in.shape)}, got {tuple(rgbd.shape)}")
        self.static_in.copy_(rgbd)
        self.graph.replay()
        return self.static_out.clone() if clone else self.static_out


class GraphedTrainStep:
    """One batch of the GAN loop (watermelon._train_step: generator forward, reconstruction, `d_ratio` critic updates with the gradient
    penalty, generator loss + backward, both Adam steps and the weight re-packs — ~900 launches on two streams) captured ONCE into a HIP
    graph and replayed with one host call per step.  ref: the per-batch body of watermelon.train, watermelon.py:207-277.

    What varies from step to step lives in static device buffers the captured kernels read: the batch (RGBD, target amplitude / phase),
    the plane index per sample (the reference's CPU ``randperm`` draw, watermelon.py:226 / angular_spectrum_method.py:536), the gradient
    penalty's alphas (CPU ``rand``, watermelon.py:460) and the Adam constants of every optimiser step of the batch (bias corrections
    for the step counts this replay stands for, learning rate, gradient scale: lhg_adam_step_scaled reads them from device memory).
    The host draws / computes them exactly as the eager step does, stages them through pinned memory and copies them in front of the
    replay on the same stream.  Same kernels, same order, same bits as the eager step (tests/test_gpu_graph.py).

    Shapes are static: one instance per (batch, rows, cols).  Every GEMM geometry must have been autotuned before the capture (the
    warm-up steps do that).  Single process only: collectives are not captured (watermelon falls back to the eager step when the
    process group has more than one rank)."""

    def __init__(self, trainer, RGBD: torch.Tensor, target_amp: torch.Tensor, target_phs: torch.Tensor, warmup: int = 2):
        if not RGBD.is_cuda:
            raise RuntimeError("GraphedTrainStep needs GPU inputs (no CPU fallback)")
        self.W = trainer
        dev = RGBD.device
        B = RGBD.shape[0]
        self.ratio = trainer.discriminator_train_ratio if trainer._opt_D is not None else 0
        self.rgbd, self.tamp, self.tphs = RGBD.detach().clone(), target_amp.detach().clone(), target_phs.detach().clone()
        self.idx = torch.zeros(B, dtype=torch.int64, device=dev)
        self.alphas = [torch.zeros((B, 1, 1, 1), dtype=torch.float32, device=dev) for _ in range(self.ratio)]
        opts = [(trainer._opt_G, 1)] + ([(trainer._opt_D, self.ratio)] if self.ratio else [])
        self._opts = opts
        # Building the graph must not train: the state every step touches is saved, the warm-up steps run eagerly on the statics
        # (autotune of every GEMM geometry, twiddles, packed weights, allocator pools), the step is captured, and the state is put
        # back — the first __call__ is then the first optimiser step of this batch, exactly as in the eager loop.
        saved = self._snapshot()
        for _ in range(max(warmup, 1)):
            self._stage(None, None)
            trainer._train_step(self.rgbd, self.tamp, self.tphs, self.idx, self.alphas)
        torch.cuda.synchronize()
        # this graph's own Adam constants (several graphs — one per batch shape — may share the optimisers): installed on the optimisers
        # for the capture only, so that every eager step before, between and after replays takes its constants from the host
        self._consts = [torch.zeros((calls, 4), dtype=torch.float32, device=dev) for _, calls in opts]
        self._stage(None, None)
        self._stage_consts()
        torch.cuda.synchronize()
        from . import hip_ops

        hip_ops.begin_graph_capture()
        self.graph = torch.cuda.CUDAGraph()
        try:
            for (opt, _), consts in zip(opts, self._consts):
                opt.device_consts, opt._consts_cursor = consts, 0
            with torch.cuda.graph(self.graph):
                self.out = trainer._train_step(self.rgbd, self.tamp, self.tphs, self.idx, self.alphas)
        finally:
            for opt, _ in opts:
                opt.device_consts = None
        self._restore(saved)

    def _snapshot(self):
        W = self.W
        state = {"rng": torch.get_rng_state(), "losses": W.train_losses_tensor.clone(), "metrics": W.train_metrics_tensor.clone(), "opts": [], "buffers": []}
        for opt, _ in self._opts:
            state["opts"].append((opt.flat.data.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.step_count))
        for mod in (W.generator, W.discriminator):
            state["buffers"].append([b.clone() for b in mod.buffers()])
        return state

    def _restore(self, state):
        from . import hip_ops

        W = self.W
        torch.cuda.synchronize()
        torch.set_rng_state(state["rng"])
        W.train_losses_tensor.copy_(state["losses"])
        W.train_metrics_tensor.copy_(state["metrics"])
        for (opt, _), (data, m, v, count) in zip(self._opts, state["opts"]):
            opt.flat.data.copy_(data)
            opt.exp_avg.copy_(m)
            opt.exp_avg_sq.copy_(v)
            opt.step_count = count
            for p_ in opt.flat.params:
                hip_ops.bump_version(p_)
            hip_ops.repack_weights(opt.flat.params)  # the packed forms the captured kernels read follow the restored weights
        for mod, saved in zip((W.generator, W.discriminator), state["buffers"]):
            for b, s_ in zip(mod.buffers(), saved):
                b.copy_(s_)
        torch.cuda.synchronize()

    def _stage(self, plane_indices, gp_alphas):
        """Host draws of one batch -> the static device buffers.  Every batch gets FRESH pinned staging tensors (the caching host
        allocator hands a block out again only after the copy that read it has run): the host may be several replays ahead of the GPU,
        so a staging buffer of its own would be overwritten before the copy of an earlier batch has executed."""
        W, B = self.W, self.idx.shape[0]
        idx = plane_indices if plane_indices is not None else W.propagator.draw_indices(B)
        self.idx.copy_(torch.as_tensor(idx).reshape(-1).to(torch.int64).cpu().pin_memory(), non_blocking=True)
        for k in range(self.ratio):
            a = gp_alphas[k] if gp_alphas is not None else torch.rand(B, 1, 1, 1)  # the reference's draw: CPU generator
            self.alphas[k].copy_(torch.as_tensor(a).reshape(B, 1, 1, 1).float().cpu().pin_memory(), non_blocking=True)

    def _stage_consts(self):
        for (opt, calls), consts in zip(self._opts, self._consts):
            scale = 1.0  # single process: the gradient buffer holds this rank's own gradient
            consts.copy_(opt.consts_rows(opt.step_count + 1, calls, scale).pin_memory(), non_blocking=True)

    def _replay(self):
        self.graph.replay()
        for opt, calls in self._opts:
            opt.step_count += calls  # the replay ran `calls` optimiser steps of this model

    def __call__(self, RGBD, target_amp, target_phs, plane_indices=None, gp_alphas=None):
        if RGBD.shape != self.rgbd.shape:
            raise ValueError(f"graph was captured for {tuple(self.rgbd.shape)}, got {tuple(RGBD.shape)}")
        for dst, src in ((self.rgbd, RGBD), (self.tamp, target_amp), (self.tphs, target_phs)):
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        self._stage(plane_indices, gp_alphas)
        self._stage_consts()
        self._replay()
        return self.out
