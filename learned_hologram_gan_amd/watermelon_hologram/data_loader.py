"""Datasets over headerless fp32 (N,C,H,W) ``.bin`` memmaps.
ref: learnedMethodForHologram/watermelon_hologram/data_loader.py:8-123 (depth uses channel 0 only)."""

from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import Dataset

from ..utilities import try_gpu


class _BinDataset(Dataset):
    def __init__(self, paths, samplesNum, channlesNum, height, width, cuda):
        self.dataShape = (samplesNum, channlesNum, height, width)
        self._maps = {k: np.memmap(p, dtype=np.float32, mode="r", shape=self.dataShape) for k, p in paths.items()}
        self.device = try_gpu() if cuda else torch.device("cpu")

    def __len__(self):
        return self.dataShape[0]

    def _check(self, idx):
        if idx < 0 or idx >= len(self):
            raise IndexError("Index out of range")

    def _get(self, key, idx):
        return torch.from_numpy(np.array(self._maps[key][idx])).to(self.device)

    def _rgbd(self, idx):
        return torch.cat((self._get("img", idx), self._get("depth", idx)[0:1]), dim=0)


class dataloaderImgDepthAmpPhs(_BinDataset):
    def __init__(self, img_path, depth_path, amp_path, phs_path, samplesNum=3800, channlesNum=3, height=192, width=192, cuda=False):
        super().__init__(dict(img=img_path, depth=depth_path, amp=amp_path, phs=phs_path), samplesNum, channlesNum, height, width, cuda)
        self.img, self.depth, self.amp, self.phs = (self._maps[k] for k in ("img", "depth", "amp", "phs"))

    def __getitem__(self, idx):
        self._check(idx)
        return self._rgbd(idx), self._get("amp", idx), self._get("phs", idx)


class dataloaderAmpPIPhs(_BinDataset):
    def __init__(self, amp_path, phs_path, samplesNum=3800, channlesNum=3, height=192, width=192, cuda=False):
        super().__init__(dict(amp=amp_path, phs=phs_path), samplesNum, channlesNum, height, width, cuda)
        self.amp, self.phs = self._maps["amp"], self._maps["phs"]

    def __getitem__(self, idx):
        self._check(idx)
        return self._get("amp", idx), 2 * torch.pi * self._get("phs", idx)


class dataloaderImgDepth(_BinDataset):
    def __init__(self, img_path, depth_path, samplesNum=3800, channlesNum=3, height=192, width=192, cuda=False):
        super().__init__(dict(img=img_path, depth=depth_path), samplesNum, channlesNum, height, width, cuda)
        self.img, self.depth = self._maps["img"], self._maps["depth"]

    def __getitem__(self, idx):
        self._check(idx)
        return self._rgbd(idx)
