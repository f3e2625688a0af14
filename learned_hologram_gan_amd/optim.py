"""Flat-buffer parameters and the fused Adam step (C ABI: lhg_adam_step).

All parameters of a module are re-bound as views of ONE contiguous fp32 buffer and their
gradients as views of a second one, so that (a) Adam is a single kernel launch over the whole
model instead of ~200 per-tensor launches (ref: torch.optim.Adam call sites watermelon.py:137-138)
and (b) data-parallel gradient reduction works on a handful of large contiguous buckets
(distributed.GradSynchronizer).  ``state_dict()`` / ``load_state_dict()`` keep working because
they copy through the views.
"""

from __future__ import annotations

import os

import torch

from . import hip_ops


_BATCHED_REPACK = os.environ.get("LHG_BATCHED_REPACK", "1") != "0"  # 0: every conv re-packs its weight on its next use (for A/B timing)


class FlatParams:
    def __init__(self, module: torch.nn.Module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        # 64-float alignment keeps every view 256-byte aligned
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 63) // 64 * 64
        self.numel = off
        self.data = torch.zeros(off, dtype=dt, device=dev)
        self.grad = torch.zeros(off, dtype=dt, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                self.data[o:o + p.numel()].copy_(p.detach().reshape(-1))
                p.data = self.data[o:o + p.numel()].view_as(p)
                p.grad = self.grad[o:o + p.numel()].view_as(p)
                # the ops' backward kernels accumulate straight into these slots (conv / conv-transpose weights from the weight-gradient
                # stream; BatchNorm affine parameters and biases from the main stream) and hand autograd None
                hip_ops.register_grad_slot(p, p.grad)

    def zero_grad(self):
        """Keep .grad bound to the flat buffer (set_to_none would drop the views)."""
        if self.grad.is_cuda:
            hip_ops.join_side_stream(self.grad.device)
            hip_ops.reset_backward_state()
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view_as(p)

    def grad_view(self, i):
        o = self.offsets[i]
        return self.grad[o:o + self.params[i].numel()]


class FusedAdam:
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8) semantics on a FlatParams buffer."""

    def __init__(self, flat: FlatParams, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.flat, self.lr, self.betas, self.eps = flat, lr, betas, eps
        self.exp_avg = torch.zeros_like(flat.data)
        self.exp_avg_sq = torch.zeros_like(flat.data)
        self.step_count = 0
        # device-resident constants (graph.GraphedTrainStep): row k = {1 - b1^t, sqrt(1 - b2^t), grad scale, lr} of the k-th step() call of
        # a captured train step; the kernel reads them at execution time, so a replayed launch follows the step count and the schedule
        self.device_consts = None
        self._consts_cursor = 0

    def zero_grad(self, set_to_none: bool = False):
        self.flat.zero_grad()

    def step(self, grad_scale: float = 1.0):
        """``grad_scale``: the gradient buffer holds grad / grad_scale (a data-parallel SUM with grad_scale = 1 / world,
        GradSynchronizer(defer_scale=True)): the kernel multiplies on the fly."""
        self.step_count += 1
        if self.flat.grad.is_cuda:
            hip_ops.join_side_stream(self.flat.grad.device)  # weight gradients accumulated on the side stream
        consts = None
        # device constants are for launches being CAPTURED (their replays must follow the step count): an eager step — warm-up of a
        # capture, a step after ``use_graph`` was switched off, a ragged last batch — takes step_count / lr / grad_scale from the host
        if self.device_consts is not None and self.flat.grad.is_cuda and torch.cuda.is_current_stream_capturing():
            consts = self.device_consts[self._consts_cursor % self.device_consts.shape[0]]
            self._consts_cursor += 1
        hip_ops.adam_step_(self.flat.data, self.flat.grad, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0], self.betas[1],
                           self.eps, self.step_count, grad_scale, consts)
        for p in self.flat.params:
            hip_ops.bump_version(p)
        if self.flat.data.is_cuda and _BATCHED_REPACK:
            hip_ops.repack_weights(self.flat.params)  # every packed form the convs hold, in one batched call

    def consts_rows(self, first_step: int, calls: int, grad_scale: float = 1.0) -> torch.Tensor:
        """Host values of ``calls`` rows of device constants for steps first_step, first_step + 1, ..."""
        rows = torch.empty((calls, 4), dtype=torch.float32)
        # the library's arithmetic (lhg_adam_step_scaled): the betas as the floats they are passed as, powers and the root in double,
        # results rounded to float — bit-identical constants whether the host or the device supplies them
        b1, b2 = (float(torch.tensor(b, dtype=torch.float32)) for b in self.betas)
        for k in range(calls):
            t = first_step + k
            rows[k, 0] = 1.0 - b1 ** t
            rows[k, 1] = (1.0 - b2 ** t) ** 0.5
            rows[k, 2] = grad_scale
            rows[k, 3] = self.lr
        return rows

    def state_dict(self):
        return dict(step=self.step_count, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, lr=self.lr, betas=self.betas, eps=self.eps)

    def load_state_dict(self, sd):
        self.step_count = sd["step"]
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])


class ReduceOnPlateau:
    """``torch.optim.lr_scheduler.ReduceLROnPlateau(opt, "min", factor, patience, threshold, "rel", min_lr=...)`` for
    FusedAdam (which is not a ``torch.optim.Optimizer``).  ref: RGBD2AP.py:86-95, AP2POH.py:154-163."""

    def __init__(self, optimizer: FusedAdam, factor=0.1, patience=4, threshold=1e-3, min_lr=1e-6, eps=1e-8):
        if factor >= 1.0:
            raise ValueError("Factor should be < 1.0.")
        self.optimizer, self.factor, self.patience, self.threshold, self.min_lr, self.eps = optimizer, factor, patience, threshold, min_lr, eps
        self.best, self.num_bad_epochs = float("inf"), 0

    def step(self, metric):
        metric = float(metric)
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.num_bad_epochs = metric, 0
        else:
            self.num_bad_epochs += 1
        if self.num_bad_epochs > self.patience:
            new_lr = max(self.optimizer.lr * self.factor, self.min_lr)
            if self.optimizer.lr - new_lr > self.eps:
                self.optimizer.lr = new_lr
            self.num_bad_epochs = 0
        return self.optimizer.lr


def run_pretraining(model, batch_loss, train_loader, val_loader, epochs, lr, gamma, save_path, checkpoint_iterval):
    """The epoch loop shared by ``RGBD2AP.train_model`` and ``AP2POH.train_model`` (ref: RGBD2AP.py:97-137, AP2POH.py:165-218):
    Adam on every parameter, per-epoch train / validation loss = sum of batch losses / number of samples, plateau schedule on the
    validation loss, ``_epoch{n}`` checkpoints.  ``batch_loss(batch) -> (loss, samples)``."""
    if model.freeze:
        raise ValueError("The model is frozen, cannot be trained")
    if save_path is None:
        print("!!!!!!The save path is not specified, the model will not be saved!!!!!!")
    if model.pretrained_model_path is not None:
        print("The model is pretrained, will be fine-tuned or continued training")
    model.train_loss, model.test_loss = [], []
    model.optimizer = FusedAdam(FlatParams(model), lr=lr)
    model.scheduler = ReduceOnPlateau(model.optimizer, factor=gamma, patience=4, threshold=1e-3, min_lr=1e-6)
    for epoch in range(epochs):
        model.train()
        total, n = torch.zeros((), device=model.optimizer.flat.data.device), 0
        for batch in train_loader:
            with hip_ops.deferred_gc():  # no collector pauses while the host thread is feeding the GPU
                loss, samples = batch_loss(batch)
                model.optimizer.zero_grad()
                loss.backward()
                model.optimizer.step()
            total, n = total + loss.detach(), n + samples
        train_loss = total.item() / n
        model.eval()
        total, n = torch.zeros_like(total), 0
        for batch in val_loader:
            with torch.no_grad():
                loss, samples = batch_loss(batch)
            total, n = total + loss, n + samples
        test_loss = total.item() / n
        model.train_loss.append(train_loss)
        model.test_loss.append(test_loss)
        print(f"epoch {epoch + 1}, train loss {train_loss:.7f}, test loss {test_loss:.7f}")
        model.scheduler.step(test_loss)
        if epoch % checkpoint_iterval == 0 and epoch != 0 and save_path is not None:
            torch.save(model.state_dict(), save_path.replace(".pth", f"_epoch{epoch}.pth"))
    if save_path is not None:
        torch.save(model.state_dict(), save_path)
