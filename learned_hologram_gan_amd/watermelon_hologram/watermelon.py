"""GAN trainer with the reference's constructor / ``train`` keyword set, loss bookkeeping JSON schema
and checkpoint naming (ref: learnedMethodForHologram/watermelon_hologram/watermelon.py:33-631).

The per-batch body of the reference loop (:207-277) lives in ``train_step`` so that bench.py and
the parity tests drive exactly what ``train`` runs.  Differences that are deliberate:
  * hat / target planes come from one fused propagation each (no (2B,3,R,C) spectra in HBM);
  * both Adam updates are single fused launches over flat parameter buffers;
  * with torch.distributed initialised, gradients are averaged across ranks (RCCL) with
    bucketed all-reduces overlapped with backward; BN statistics / loss normalisers are per replica, or over
    the global batch with ``configure(sync_batch_stats=True)`` (W replicas of B == the reference at W x B).
The VGG19 perceptual term (loss_func.py:12-51) is ``perceptual.perceptualLoss``; it needs a local weights
file, so the trainer takes it as a constructor argument and ``perceptual_loss_weight`` must be 0 without one.
"""

from __future__ import annotations

import contextlib
import os
import json

import torch
import torch.autograd as autograd
import torch.nn.functional as F

from .. import hip_ops
from ..angular_spectrum_method import bandLimitedAngularSpectrumMethod_for_multiple_distances
from ..distributed import GradSynchronizer, broadcast_module_state
from ..optim import FlatParams, FusedAdam
from ..poh_ops import ReconLossFn, psnr_ssim
from ..utilities import save_planes_as_png, tensor_normalizor_2D, try_gpu
from .discriminator import WGANGPDiscriminator192, fakeDiscriminator
from .generator import Generator
from .loss_func import fakePerceptualLoss, focal_sincos_phase_gradient_loss, total_variation_loss

LOSS_NAMES = ("focal_phase_gradient_loss", "perceptual_loss", "pixel_loss", "TV_loss", "gan_loss", "G_loss", "D_loss")


class _frozen:
    """Context manager: parameters of `module` do not require grad while a forward pass is recorded."""

    def __init__(self, module, enabled=True):
        self.params = [p for p in module.parameters() if p.requires_grad] if enabled else []

    def __enter__(self):
        for p in self.params:
            p.requires_grad_(False)

    def __exit__(self, *exc):
        for p in self.params:
            p.requires_grad_(True)
        return False


def psnr(hat, target):
    """torchmetrics PeakSignalNoiseRatio() defaults as the reference calls it (one ``forward`` per batch, watermelon.py:447-456):
    the min/max target states start at 0.0, so data_range = max(max t, 0) - min(min t, 0)."""
    zero = torch.zeros((), dtype=target.dtype, device=target.device)
    rng = torch.maximum(target.max(), zero) - torch.minimum(target.min(), zero)
    return 10.0 * torch.log10(rng * rng / F.mse_loss(hat, target))


def ssim(hat, target, kernel_size=11, sigma=1.5, k1=0.01, k2=0.03):
    """Gaussian-window SSIM in the torchmetrics default configuration (parity unpinned: torchmetrics
    is not installed in the build image)."""
    C = hat.shape[1]
    rng = torch.maximum(hat.max() - hat.min(), target.max() - target.min())
    c1, c2 = (k1 * rng) ** 2, (k2 * rng) ** 2
    ax = torch.arange(kernel_size, dtype=torch.float32, device=hat.device) - (kernel_size - 1) / 2
    g = torch.exp(-(ax / sigma) ** 2 / 2)
    g = (g / g.sum()).view(1, 1, -1)
    win = (g.transpose(1, 2) * g).expand(C, 1, kernel_size, kernel_size).contiguous()
    pad = (kernel_size - 1) // 2
    stack = torch.cat((hat, target, hat * hat, target * target, hat * target), 0)
    mu = F.conv2d(F.pad(stack, (pad, pad, pad, pad), mode="reflect"), win, groups=C)
    B = hat.shape[0]
    mx, my, sxx, syy, sxy = mu[:B], mu[B:2 * B], mu[2 * B:3 * B], mu[3 * B:4 * B], mu[4 * B:]
    vx, vy, cxy = sxx - mx * mx, syy - my * my, sxy - mx * my
    s = ((2 * mx * my + c1) * (2 * cxy + c2)) / ((mx * mx + my * my + c1) * (vx + vy + c2))
    return s[..., pad:-pad, pad:-pad].mean()


class watermelon:
    def __init__(self, filter_radius_coefficient=0.5, pad_size=416, kernel_size=3,
                 distance_stack=torch.linspace(-1.5e-4, 0.0, 8)[:-1], pretrained_model_path_G=None, pretrained_model_path_D=None,
                 input_shape=(1, 4, 192, 192), cuda=True, perceptual_loss=None):
        self.device = try_gpu() if cuda else torch.device("cpu")
        self.distance_stack = distance_stack.to(self.device)
        self.distance_num = distance_stack.size(0)
        wl = torch.tensor([638e-9, 520e-9, 450e-9])
        self.generator = Generator(sample_row_num=input_shape[-2], sample_col_num=input_shape[-1], pad_size=pad_size,
                                   filter_radius_coefficient=filter_radius_coefficient, kernel_size=kernel_size, pixel_pitch=3.74e-6,
                                   wave_length=wl, distance=torch.tensor([1e-3]), pretrained_model_path=pretrained_model_path_G)
        self.discriminator = self._make_discriminator(pretrained_model_path_D)
        self.perceptual_loss = perceptual_loss if perceptual_loss is not None else fakePerceptualLoss()
        self._has_perceptual = perceptual_loss is not None
        self.propagator = bandLimitedAngularSpectrumMethod_for_multiple_distances(
            sample_row_num=input_shape[-2], sample_col_num=input_shape[-1], distances=distance_stack, pad_size=pad_size,
            filter_radius_coefficient=filter_radius_coefficient, pixel_pitch=3.74e-6, wave_length=wl, band_limit=False, cuda=True)
        if pretrained_model_path_G is not None:
            print(f"Generator loaded from {pretrained_model_path_G}")
        if pretrained_model_path_D is not None:
            print(f"Discriminator loaded from {pretrained_model_path_D}")
        self._opt_G = self._opt_D = self._sync_G = self._sync_D = None
        self.skip_unused_critic_grads = True  # see train_step: critic weight gradients of the generator pass are dead values
        # train_step as one hipGraph replay per batch (single process; static batch shape): LHG_TRAIN_GRAPH=1, or set the attribute
        self.pair_critic_passes = os.environ.get("LHG_PAIR_CRITIC", "1") != "0"  # A/B switch of forward_pair (round 4)
        self.use_graph = os.environ.get("LHG_TRAIN_GRAPH", "0") == "1"
        self._graphed = None

    def _make_discriminator(self, path):
        return WGANGPDiscriminator192(pretrained_model_path=path, cuda=True)

    # ------------------------------------------------------------------ configuration of one run
    def configure(self, phs_gradient_loss_weight=1, perceptual_loss_weight=1.0, pixel_loss_weight=1.0, TV_loss_weight=1e-3,
                  discriminator_loss_weight=1.0, lr_G=1e-3, lr_D=1e-3, discriminator_train_ratio=2, discriminator_lambda=10,
                  grad_buckets=4, sync_batch_stats=False, grad_payload=None):
        """``grad_payload`` (data-parallel runs): "fp32" or "bf16" wire format of the gradient buckets' all-reduce; default: bf16 when the
        activations are stored as bf16 (hip_ops.activation_storage(): the bf16 configs of BASELINE.json), fp32 otherwise.  The reduced
        sums land in the fp32 flat buffer either way and the division by the world size is done by the Adam kernel.
        ``sync_batch_stats`` (data-parallel runs): normalise every train-mode BatchNorm, the focal loss's max normalisers and the TV
        means over the GLOBAL batch (all-reduced statistics, hip_ops.set_sync_batch_stats) — W replicas of B samples then reproduce the
        reference's single-device step at batch W x B (ref: neural_network_components.py:23-24, loss_func.py:94-98, 152-157).  Off
        (default): those statistics are per replica, which at one sample per replica (BASELINE configs[4]: bs=8 over 8 GPUs) is a
        different model from the reference's."""
        if perceptual_loss_weight and not self._has_perceptual:
            raise ValueError("perceptual_loss_weight != 0 needs a perceptual module: construct the trainer with "
                             "perceptual_loss=perceptualLoss(weights_path=...) (watermelon_hologram/perceptual.py)")
        self.phs_gradient_loss_weight, self.perceptual_loss_weight = phs_gradient_loss_weight, perceptual_loss_weight
        self.pixel_loss_weight, self.TV_loss_weight = pixel_loss_weight, TV_loss_weight
        self.discriminator_loss_weight = discriminator_loss_weight
        self.discriminator_train_ratio, self.discriminator_lambda = discriminator_train_ratio, discriminator_lambda
        hip_ops.set_sync_batch_stats(sync_batch_stats)
        self._sync_batch_stats = bool(sync_batch_stats)
        self.generator.to(self.device)
        flat_G = FlatParams(self.generator)
        broadcast_module_state(self.generator, flat_G.data)  # data-parallel replicas start from rank 0's weights
        self._opt_G = FusedAdam(flat_G, lr=lr_G)
        if grad_payload is None:
            grad_payload = "bf16" if hip_ops.activation_storage() == "bf16" else "fp32"
        self._sync_G = GradSynchronizer(flat_G.params, flat_G.offsets, flat_G.grad, grad_buckets, payload=grad_payload, defer_scale=True)
        trainable_D = [p for p in self.discriminator.parameters() if p.requires_grad]
        if discriminator_train_ratio > 0 and trainable_D and isinstance(self.discriminator, WGANGPDiscriminator192):
            flat_D = FlatParams(self.discriminator)
            broadcast_module_state(self.discriminator, flat_D.data)
            self._opt_D = FusedAdam(flat_D, lr=lr_D)
            self._sync_D = GradSynchronizer(flat_D.params, flat_D.offsets, flat_D.grad, max(1, grad_buckets // 2), payload=grad_payload,
                                            defer_scale=True)
        else:
            self._opt_D = self._sync_D = None
        self.train_losses_tensor = torch.zeros(7, device=self.device)
        self.train_metrics_tensor = torch.zeros(2, device=self.device)
        self._graphed = None  # a captured step belongs to the optimisers / buffers it was recorded with

    # ------------------------------------------------------------------ pieces of the step
    def compute_gradient_penalty(self, real_samples, fake_samples, alpha=None):
        """ref: watermelon.py:458-477."""
        if alpha is None:
            alpha = torch.rand(real_samples.size(0), 1, 1, 1)  # the reference's draw: CPU generator
            # (pinned staging: a pageable copy would make the host wait for everything queued on the stream)
            alpha = alpha.pin_memory().to(self.device, non_blocking=True) if real_samples.is_cuda else alpha.to(self.device)
        interpolates = (alpha * real_samples + ((1 - alpha) * fake_samples)).requires_grad_(True)
        d_interpolates = self.discriminator(interpolates)
        # only d D / d x^ is asked for: the critic's parameter gradients of THIS pass would be computed and dropped by the engine
        with (hip_ops.only_input_gradients() if interpolates.is_cuda else contextlib.nullcontext()):
            gradients = autograd.grad(outputs=d_interpolates, inputs=interpolates, grad_outputs=torch.ones_like(d_interpolates),
                                      create_graph=True, retain_graph=True, only_inputs=True)[0]
        gradients = gradients.view(gradients.size(0), -1)
        return ((gradients.norm(2, dim=1) - 1) ** 2).mean()

    def G_loss(self, hat_amps, target_amps, hat_phs, target_phs, loss_from_discriminator, recorder=None):
        """ref: watermelon.py:418-445."""
        if hat_amps.is_cuda:  # fused HIP kernels (csrc/losses.hip)
            focal, mse, tv_diff = ReconLossFn.apply(hat_amps, target_amps, hat_phs, target_phs).unbind(0)
        else:  # host-side evaluation of the same definitions (loss_func.py)
            focal = focal_sincos_phase_gradient_loss(hat_phs, target_phs)
            mse, tv_diff = F.mse_loss(hat_amps, target_amps), total_variation_loss(hat_amps, target_amps)
        phs_loss = focal * self.phs_gradient_loss_weight
        perceptual = self.perceptual_loss(hat_amps, target_amps) * self.perceptual_loss_weight
        pixel = mse * self.pixel_loss_weight
        tv = tv_diff * self.TV_loss_weight
        gan = loss_from_discriminator * self.discriminator_loss_weight
        loss = phs_loss + perceptual + pixel + tv + gan
        if recorder is not None:
            with torch.no_grad():
                recorder += torch.stack([t.detach().reshape(()).float() for t in (phs_loss, perceptual, pixel, tv, gan, loss)]
                                        + [torch.zeros((), device=recorder.device)])
        return loss

    def record_metrics(self, hat_amps, target_amps, recorder=None):
        with torch.no_grad():
            if hat_amps.is_cuda:  # fused HIP kernels (csrc/metrics.hip)
                recorder += psnr_ssim(hat_amps, target_amps)
            else:                 # host-side evaluation of the same definitions (used by the CPU tests of the definitions)
                recorder += torch.stack([psnr(hat_amps, target_amps), ssim(hat_amps, target_amps)])

    def reconstruct(self, RGBD, target_amp, target_phs, plane_indices=None):
        """G forward + hat/target amplitude and phase at one plane per sample (watermelon.py:216-241)."""
        POH = self.generator(RGBD)
        if plane_indices is None:
            plane_indices = self.propagator.draw_indices(RGBD.size(0))
        hat_a, hat_p, tgt_a, tgt_p = self.propagator.reconstruct_planes(self.generator.part2.propagator, POH, target_amp, target_phs,
                                                                         plane_indices)
        return POH, hat_a, tgt_a, hat_p, tgt_p

    def train_step(self, RGBD, target_amp, target_phs, plane_indices=None, gp_alphas=None):
        """One batch of the reference loop (watermelon.py:207-277).  Returns detached tensors for logging."""
        if self._opt_G is None:
            raise RuntimeError("call configure(...) (or train(...)) before train_step")
        if getattr(self, "_sync_batch_stats", False):
            hip_ops.assert_equal_batches(RGBD.shape[0], RGBD.device)  # global-batch statistics weight the replicas equally
        with hip_ops.deferred_gc():  # no collector pauses while the host thread is feeding the GPU
            if self.use_graph and RGBD.is_cuda and self._sync_G.world == 1:
                # the whole batch as ONE hipGraph replay (graph.GraphedTrainStep): ~900 launches per step leave the host
                # one graph per batch shape (a ragged last batch alternates with the full one every epoch: no re-capture).  The returned
                # tensors are the graph's STATIC outputs — the next replay of the same shape overwrites them; clone what must outlive a step
                key = tuple(RGBD.shape)
                if self._graphed is None:
                    self._graphed = {}
                if key not in self._graphed:
                    from ..graph import GraphedTrainStep

                    self._graphed[key] = GraphedTrainStep(self, RGBD, target_amp, target_phs)
                return self._graphed[key](RGBD, target_amp, target_phs, plane_indices, gp_alphas)
            return self._train_step(RGBD, target_amp, target_phs, plane_indices, gp_alphas)

    def _train_step(self, RGBD, target_amp, target_phs, plane_indices, gp_alphas):
        ratio = self.discriminator_train_ratio if self._opt_D is not None else 0
        POH, hat_amps, target_amps, hat_phases, target_phases = self.reconstruct(RGBD, target_amp, target_phs, plane_indices)
        fake = hat_amps.detach()
        d_total = torch.zeros((), device=self.device)
        for it in range(ratio):
            if self.pair_critic_passes and hasattr(self.discriminator, "forward_pair"):
                # D(real) and D(fake) as one pass over the stacked batches (same values: WGANGPDiscriminator192.forward_pair)
                real_validity, fake_validity = self.discriminator.forward_pair(target_amps, fake)
            else:
                real_validity = self.discriminator(target_amps)
                fake_validity = self.discriminator(fake)
            gp = self.compute_gradient_penalty(target_amps, fake, None if gp_alphas is None else gp_alphas[it])
            d_loss = (-torch.mean(real_validity) + torch.mean(fake_validity)) + self.discriminator_lambda * gp
            self._opt_D.zero_grad()
            self._sync_D.start()
            d_loss.backward(retain_graph=True)
            self._sync_D.finish()
            self._opt_D.step(grad_scale=self._sync_D.grad_scale)
            d_total = d_total + d_loss.detach() / ratio
        # The critic's weight gradients of this pass are never read: the reference zeroes them (watermelon.py:252) before the
        # next critic backward and optimizer_G does not own them.  Recording the pass with frozen critic weights drops those
        # weight-gradient GEMMs; every tensor the step produces (losses, generator gradients, both Adam updates) is unchanged.
        with _frozen(self.discriminator, self.skip_unused_critic_grads):
            loss_from_discriminator = -torch.mean(self.discriminator(hat_amps))
        g_loss = self.G_loss(hat_amps, target_amps, hat_phases, target_phases, loss_from_discriminator, self.train_losses_tensor)
        self._opt_G.zero_grad()
        self._sync_G.start()
        g_loss.backward()
        self._sync_G.finish()
        self._opt_G.step(grad_scale=self._sync_G.grad_scale)
        self.train_losses_tensor[-1] += d_total
        return dict(POH=POH.detach(), hat_amps=hat_amps.detach(), target_amps=target_amps.detach(), G_loss=g_loss.detach(), D_loss=d_total)

    # ------------------------------------------------------------------ the reference's loop
    def train(self, data_loader_train, data_loader_val, phs_gradient_loss_weight=1, perceptual_loss_weight=1.0, pixel_loss_weight=1.0,
              TV_loss_weight=1e-3, discriminator_loss_weight=1.0, epoch_num=2, lr_G=1e-3, lr_D=1e-3, save_path_G=None, save_path_D=None,
              info_print_interval=100, info_plot_interval=600, loss_metrics_file=None, save_path_img=None, checkpoint_iterval=5,
              discriminator_train_ratio=2, discriminator_lambda=10, step_scheduler_G_gamma=0.1, step_scheduler_D_gamma=0.9999,
              visualization_RGBD_AP=None):
        if save_path_G is None:
            print("!!!!!!The save path of the generator is not specified, the model will not be saved!!!!!!")
        if save_path_D is None:
            print("!!!!!!The save path of the discriminator is not specified, the model will not be saved!!!!!!")
        self.configure(phs_gradient_loss_weight, perceptual_loss_weight, pixel_loss_weight, TV_loss_weight, discriminator_loss_weight,
                       lr_G, lr_D, discriminator_train_ratio, discriminator_lambda)
        self.dict_for_losses_metrics = {
            "epoch": [], "n_batch_in_epoch": [], "n_train": [], "n_batch": [],
            "train_losses_tensor": {k: [] for k in LOSS_NAMES}, "train_metrics_tensor": {"PSNR": [], "SSIM": []},
            "validate_losses_tensor": {k: [] for k in LOSS_NAMES}, "validate_metrics_tensor": {"PSNR": [], "SSIM": []},
        }
        losses_last = torch.zeros(7, device=self.device)
        metrics_last = torch.zeros(2, device=self.device)
        n_train = n_batch = n_batch_last = 0
        for epoch in range(epoch_num):
            if hasattr(data_loader_train, "set_epoch"):
                data_loader_train.set_epoch(epoch)  # per-rank shards reshuffle every epoch
            self.generator.train()
            self.discriminator.train()
            for n_batch_in_epoch, (RGBD, target_amp, target_phs) in enumerate(data_loader_train):
                n_batch += 1
                n_train += RGBD.size(0)
                out = self.train_step(RGBD, target_amp, target_phs)
                self.record_metrics(out["hat_amps"], out["target_amps"], self.train_metrics_tensor)
                if n_batch % info_print_interval == 0:
                    with torch.no_grad():
                        v_losses, v_metrics = self._validate_generator(data_loader_val)
                    t_losses = (self.train_losses_tensor - losses_last) / (n_batch - n_batch_last)
                    t_metrics = (self.train_metrics_tensor - metrics_last) / (n_batch - n_batch_last)
                    fmt = lambda t: ", ".join(f"{k} {v}" for k, v in zip(LOSS_NAMES, t.tolist()))  # noqa: E731
                    print(f"epoch {epoch}, batch {n_batch_in_epoch + 1} ({n_train} samples and {n_batch} batches have been trained):\n"
                          f"      train: {fmt(t_losses)};\n      train: PSNR {t_metrics[0]}, SSIM {t_metrics[1]};\n"
                          f"      validate: {fmt(v_losses)};\n      validate: PSNR {v_metrics[0]}, SSIM {v_metrics[1]};\n")
                    self._add_losses_metrics_to_dict(epoch, n_batch_in_epoch, n_train, n_batch, v_losses, v_metrics, t_losses, t_metrics,
                                                     self.dict_for_losses_metrics)
                    losses_last, metrics_last = self.train_losses_tensor.clone(), self.train_metrics_tensor.clone()
                    n_batch_last = n_batch
                if n_batch % info_plot_interval == 0 and visualization_RGBD_AP is not None:
                    self._visualize(visualization_RGBD_AP, save_path_img, f"in epoch {epoch}, batch {n_batch_in_epoch + 1}")
                    print(f"visualization saved at epoch {epoch}, batch {n_batch_in_epoch + 1}")
            if epoch % checkpoint_iterval == 0:
                self._checkpoint(save_path_G, save_path_D, loss_metrics_file, suffix=f"_epoch{epoch}")
                if visualization_RGBD_AP is not None:
                    self._visualize(visualization_RGBD_AP, save_path_img, f"in epoch {epoch}")
                    print(f"visualization saved at epoch {epoch}")
        self._checkpoint(save_path_G, save_path_D, loss_metrics_file, suffix="")

    def _visualize(self, sample, save_path_img, when):
        """``amp_hat <when>.png`` / ``phs_hat <when>.png``: the reconstruction of one validation sample at the generator's fixed
        distance, each plane normalised to [0, 1] (ref: watermelon.py:325-355 every ``info_plot_interval`` batches in the network's
        current mode, :376-404 at every checkpoint epoch under no_grad).  Written when ``save_path_img`` is given."""
        if save_path_img is None:
            return
        with torch.no_grad():
            POH = self.generator(sample[0].unsqueeze(0).to(self.device))
            amp_hat, phs_hat = self.generator.part2.propagator.propagate_POH2AP_forward(POH)
        save_planes_as_png(tensor_normalizor_2D(torch.cat((amp_hat, phs_hat), dim=0)), save_path_img, True,
                           titles=[f"amp_hat {when}", f"phs_hat {when}"])

    def _checkpoint(self, save_path_G, save_path_D, loss_metrics_file, suffix):
        """<save_path> and <save_path minus .pth>_epoch{n}.pth (watermelon.py:361-374, 406-412)."""
        for path, module, name in ((save_path_G, self.generator, "Generator"), (save_path_D, self.discriminator, "Discriminator")):
            if path is not None:
                p = path.replace(".pth", f"{suffix}.pth") if suffix else path
                torch.save(module.state_dict(), p)
                print(f"{name} saved to {p}")
        if loss_metrics_file is not None:
            self._save_losses_metrics_to_dict(loss_metrics_file)
            print(f"losses and metrics saved to {loss_metrics_file}")

    def _validate_generator(self, data_loader_val):
        """Eval-mode G/D over all planes (ref: watermelon.py:479-552)."""
        self.generator.eval()
        self.discriminator.eval()
        v_losses = torch.zeros(7, device=self.device)
        v_metrics = torch.zeros(2, device=self.device)
        n_batch = 0
        with torch.no_grad():
            fixed = self.generator.part2.propagator
            for RGBD, target_amp, target_phs in data_loader_val:
                n_batch += 1
                B = RGBD.size(0)
                POH = self.generator(RGBD)
                hat_target_freq = torch.cat((fixed.propagate_POH2Freq_forward(POH), self.propagator.filter_AP2filteredFreq(target_amp, target_phs)), 0)
                amps, phss = self.propagator.propagate_multiple_samples_with_all_fixed_multiple_distances_freq2amp(hat_target_freq)
                n = B * self.distance_num
                v_losses[-1] = 0.0
                adv = -torch.mean(self.discriminator(amps[:n]))
                self.G_loss(amps[:n], amps[n:], phss[:n], phss[n:], adv, v_losses)
                self.record_metrics(amps[:n], amps[n:], v_metrics)
        self.generator.train()
        self.discriminator.train()
        return v_losses / max(n_batch, 1), v_metrics / max(n_batch, 1)

    def _add_losses_metrics_to_dict(self, epoch, n_batch_in_epoch, n_train, n_batch, v_losses, v_metrics, t_losses, t_metrics, recorder=None):
        for k, v in (("epoch", epoch), ("n_batch_in_epoch", n_batch_in_epoch), ("n_train", n_train), ("n_batch", n_batch)):
            recorder[k].append(v)
        for i, name in enumerate(LOSS_NAMES):
            recorder["train_losses_tensor"][name].append(t_losses[i].item())
            recorder["validate_losses_tensor"][name].append(v_losses[i].item())
        for i, name in enumerate(("PSNR", "SSIM")):
            recorder["train_metrics_tensor"][name].append(t_metrics[i].item())
            recorder["validate_metrics_tensor"][name].append(v_metrics[i].item())

    def _save_losses_metrics_to_dict(self, loss_metrics_file):
        with open(loss_metrics_file, "w") as f:
            json.dump(self.dict_for_losses_metrics, f)


class watermelon_without_GAN(watermelon):
    """The variant the shipped CLI trains (trainingModel.py:4): no critic, ratio 0. ref: watermelon.py:637-715."""

    def __init__(self, filter_radius_coefficient=0.5, pad_size=416, distance_stack=torch.linspace(-1.5e-4, 0.0, 8)[:-1],
                 pretrained_model_path_G=None, pretrained_model_path_D=None, input_shape=(1, 4, 192, 192), cuda=True, perceptual_loss=None):
        super().__init__(filter_radius_coefficient=filter_radius_coefficient, pad_size=pad_size, distance_stack=distance_stack,
                         pretrained_model_path_G=pretrained_model_path_G, pretrained_model_path_D=None, input_shape=input_shape,
                         cuda=cuda, perceptual_loss=perceptual_loss)

    def _make_discriminator(self, path):
        return fakeDiscriminator(pretrained_model_path=None, feature_d=32, cuda=True)

    def train(self, data_loader_train, data_loader_val, **kw):
        """ref: watermelon.py:667-715 — forces the critic terms to zero and does NOT forward ``save_path_D``, ``lr_D`` or
        ``step_scheduler_D_gamma`` to the base loop: no critic checkpoint is written (the base loop prints its warning)."""
        for dropped in ("save_path_D", "lr_D", "step_scheduler_D_gamma"):
            kw.pop(dropped, None)
        kw.update(discriminator_loss_weight=0.0, discriminator_train_ratio=0, discriminator_lambda=0.0)
        super().train(data_loader_train, data_loader_val, **kw)
