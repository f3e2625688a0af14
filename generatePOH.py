#!/usr/bin/env python3
"""RGBD frame -> phase-only hologram (+ optional multi-plane propagation to PNGs).
Same flags and artefacts as the reference entry point (ref: generatePOH.py:13-169), running on
the MI355X-native package."""

from __future__ import annotations

import argparse

import torch

FLAGS = (
    # name, kwargs
    ("--img_path", dict(type=str, required=True, help="Path to the input img.bin file")),
    ("--depth_path", dict(type=str, required=True, help="Path to the input depth.bin file")),
    ("--index", dict(type=int, required=True, help="Index of the sample to generate POH for")),
    ("--model_path", dict(type=str, required=True, help="Path to the pretrained model")),
    ("--poh_output_path", dict(type=str, required=True, help="Path to save the generated POH")),
    ("--samplesNum", dict(type=int, default=100, help="Number of samples")),
    ("--sample_row_num", dict(type=int, default=384, help="Number of sample rows")),
    ("--sample_col_num", dict(type=int, default=384, help="Number of sample columns")),
    ("--pad_size", dict(type=int, default=320, help="Padding size")),
    ("--pixel_pitch", dict(type=float, default=3.74e-6, help="Pixel pitch")),
    ("--wave_length", dict(nargs="+", type=float, default=[638e-9, 520e-9, 450e-9], help="Wavelengths for RGB channels")),
    ("--distance", dict(type=float, default=1e-3, help="Distance for propagation")),
    ("--filter_radius_coefficient", dict(type=float, default=0.35, help="Filter radius coefficient")),
    ("--propagate", dict(action="store_true", help="Flag to enable propagation")),
    ("--min_distance", dict(type=float, default=4e-4, help="Minimum distance for propagation")),
    ("--max_distance", dict(type=float, default=10e-4, help="Maximum distance for propagation")),
    ("--num_intervals", dict(type=int, default=1, help="Number of intervals for propagation distances")),
    ("--output_image_dir", dict(type=str, default=None, help="Directory to save propagated images")),
)


def build_parser():
    parser = argparse.ArgumentParser(description="Script for generating and propagating POH")
    for name, kw in FLAGS:
        parser.add_argument(name, **kw)
    return parser


def main(args):
    from learned_hologram_gan_amd import utilities
    from learned_hologram_gan_amd.angular_spectrum_method import bandLimitedAngularSpectrumMethod_for_multiple_distances as BLASM_v4
    from learned_hologram_gan_amd.watermelon_hologram.data_loader import dataloaderImgDepth
    from learned_hologram_gan_amd.watermelon_hologram.generator import Generator

    from learned_hologram_gan_amd import hip_ops

    hip_ops.apply_env_precision()  # LHG_CONV_PRECISION=bf16: bf16 operands in the conv GEMMs (the flags stay the reference's)
    dataset = dataloaderImgDepth(img_path=args.img_path, depth_path=args.depth_path, samplesNum=args.samplesNum, channlesNum=3,
                                 height=args.sample_row_num, width=args.sample_col_num, cuda=True)
    wave_length = torch.tensor(args.wave_length)
    model = Generator(sample_row_num=args.sample_row_num, sample_col_num=args.sample_col_num, pad_size=args.pad_size,
                      filter_radius_coefficient=0.45, pixel_pitch=args.pixel_pitch, wave_length=wave_length,
                      distance=torch.tensor([args.distance]), pretrained_model_path=args.model_path)
    model.to(utilities.try_gpu()).eval()
    with torch.no_grad():
        POH = model(dataset[args.index].unsqueeze(0))
    torch.save(POH.squeeze(0), args.poh_output_path)
    print(f"POH data saved at {args.poh_output_path}")

    if args.propagate:
        distances = torch.linspace(args.min_distance, args.max_distance, args.num_intervals)
        propagator = BLASM_v4(sample_row_num=args.sample_row_num, sample_col_num=args.sample_col_num, pad_size=args.pad_size,
                              distances=distances, filter_radius_coefficient=args.filter_radius_coefficient,
                              pixel_pitch=args.pixel_pitch, wave_length=wave_length, band_limit=False, cuda=True)
        with torch.no_grad():
            amp_hat = propagator(torch.ones_like(POH), POH, distances)
        if args.output_image_dir is not None:
            utilities.save_planes_as_png(utilities.tensor_normalizor_2D(amp_hat), args.output_image_dir, rgb_img=True)
        print(f"Propagated images saved at {args.output_image_dir}")


if __name__ == "__main__":
    main(build_parser().parse_args())
