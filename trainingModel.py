#!/usr/bin/env python3
"""Train the hologram generator.  Same flags, defaults, checkpoint names and metrics file as the
reference entry point (ref: trainingModel.py:13-137), running on the MI355X-native package.
Under ``python -m torch.distributed.run --nproc-per-node N`` each rank trains on its shard of
the data set and gradients are averaged over RCCL."""

from __future__ import annotations

import argparse
import os

import torch

REQUIRED_STR = ("train_img_path", "train_depth_path", "train_amp_path", "train_phs_path",
                "validate_img_path", "validate_depth_path", "validate_amp_path", "validate_phs_path")
REQUIRED_INT = ("samplesNum", "channlesNum", "height", "width")  # (sic) the reference spells it channlesNum
OPTIONAL = (("batch_size", int, 4), ("lr_G", float, 1e-3), ("lr_D", float, 1e-3), ("epoch_num", int, 50))
OUTPUTS = ("save_path_G", "save_path_D", "loss_metrics_file", "save_path_img")


def build_parser():
    parser = argparse.ArgumentParser(description="Train a GAN model for hologram generation.")
    for n in REQUIRED_STR:
        parser.add_argument(f"--{n}", type=str, required=True, help=f"Path: {n.replace('_', ' ')} (.bin, fp32 NCHW).")
    for n in REQUIRED_INT:
        parser.add_argument(f"--{n}", type=int, required=True, help=f"{n} of the dataset.")
    for n, t, d in OPTIONAL:
        parser.add_argument(f"--{n}", type=t, default=d, help=f"Default is {d}.")
    for n in OUTPUTS:
        parser.add_argument(f"--{n}", type=str, required=True, help=f"Output: {n.replace('_', ' ')}.")
    return parser


def check_and_create_folder(path):
    if path and not os.path.exists(path):
        print(f"Folder {path} does not exist, creating it...")
        os.makedirs(path, exist_ok=True)


def train_gan(a):
    from learned_hologram_gan_amd import distributed, utilities
    from learned_hologram_gan_amd.watermelon_hologram.data_loader import PrefetchLoader, dataloaderImgDepthAmpPhs
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon_without_GAN as watermelon  # as shipped (trainingModel.py:4)

    rank, world, _ = distributed.init_from_env()
    from learned_hologram_gan_amd import hip_ops

    hip_ops.apply_env_precision()  # LHG_CONV_PRECISION=bf16: bf16 operands in the conv GEMMs (the flags stay the reference's)
    utilities.set_seed(122731 + rank)

    def dataset(prefix, n):
        return dataloaderImgDepthAmpPhs(img_path=getattr(a, f"{prefix}_img_path"), depth_path=getattr(a, f"{prefix}_depth_path"),
                                        amp_path=getattr(a, f"{prefix}_amp_path"), phs_path=getattr(a, f"{prefix}_phs_path"),
                                        samplesNum=n, channlesNum=a.channlesNum, height=a.height, width=a.width, cuda=True)

    train_set, val_set = dataset("train", a.samplesNum), dataset("validate", 100)
    # same batches as DataLoader(shuffle=True, drop_last=True) [+ DistributedSampler per rank], gathered into pinned memory by a
    # background thread and copied asynchronously (ref: trainingModel.py:43-56 uses DataLoader(num_workers=0))
    train_loader = PrefetchLoader(train_set, batch_size=a.batch_size, shuffle=True, drop_last=True, rank=rank, world=world, seed=0)
    val_loader = PrefetchLoader(val_set, batch_size=max(1, a.batch_size // 2), shuffle=False)

    # The reference weighs a VGG19 perceptual term with 0.1 (trainingModel.py:78) using downloaded ImageNet weights; here the
    # weights must come from a local file ($LHG_VGG19_WEIGHTS), otherwise the term is switched off.
    perceptual, perceptual_weight = None, 0.0
    if os.environ.get("LHG_VGG19_WEIGHTS"):
        from learned_hologram_gan_amd.watermelon_hologram.perceptual import perceptualLoss

        perceptual, perceptual_weight = perceptualLoss(), 1e-1
    elif rank == 0:
        print("LHG_VGG19_WEIGHTS is not set: training without the VGG19 perceptual term (reference weight 0.1)")
    GAN = watermelon(filter_radius_coefficient=0.45, pad_size=320, distance_stack=torch.linspace(-4e-4, 0.0, 21)[:-1],
                     pretrained_model_path_G=None, pretrained_model_path_D=None, input_shape=(1, 4, a.height, a.width), cuda=True,
                     perceptual_loss=perceptual)
    if rank == 0:
        for p in (os.path.dirname(a.save_path_G), os.path.dirname(a.save_path_D), os.path.dirname(a.loss_metrics_file), a.save_path_img):
            check_and_create_folder(p)
    only0 = (lambda p: p if rank == 0 else None)
    GAN.train(data_loader_train=train_loader, data_loader_val=val_loader, phs_gradient_loss_weight=1,
              perceptual_loss_weight=perceptual_weight,
              pixel_loss_weight=1, TV_loss_weight=1e-3, discriminator_loss_weight=1e-1, epoch_num=a.epoch_num, lr_G=a.lr_G, lr_D=a.lr_D,
              save_path_G=only0(a.save_path_G), save_path_D=only0(a.save_path_D), info_print_interval=50, info_plot_interval=50,
              loss_metrics_file=only0(a.loss_metrics_file), save_path_img=a.save_path_img, checkpoint_iterval=1,
              discriminator_train_ratio=5, discriminator_lambda=10, step_scheduler_G_gamma=0.9999, step_scheduler_D_gamma=0.9999,
              visualization_RGBD_AP=(val_set[0] if rank == 0 else None))  # trainingModel.py:96: dataset_validate[0]


if __name__ == "__main__":
    train_gan(build_parser().parse_args())
