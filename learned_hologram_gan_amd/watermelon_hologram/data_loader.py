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


class PrefetchLoader:
    """Batch iterator over one of the ``.bin`` datasets above that keeps the input pipeline off the step's critical path
    (SURVEY §8f N3).  The reference wraps its datasets in ``DataLoader(num_workers=0)`` and copies every sample to the device
    separately and synchronously inside ``__getitem__`` (data_loader.py:43-49; trainingModel.py:43-56): at ~50 frames/s that is
    ~16 blocking pageable copies per step.  Here a background thread gathers the next batches from the memmaps into pinned
    staging buffers and issues ONE asynchronous copy per tensor on a side stream; the consumer stream waits on an event.

    Ordering is ``DataLoader``'s: with ``shuffle`` each epoch draws ``randperm`` from a generator seeded by the global RNG exactly
    as ``RandomSampler`` does, so a single-process run visits samples in the reference's order; with ``world > 1`` the epoch
    permutation is ``DistributedSampler``'s (seed + epoch, padded to a multiple of world, strided by rank).
    Yields the dataset's tuple with a leading batch dimension, on the dataset's device.
    """

    def __init__(self, dataset: _BinDataset, batch_size=1, shuffle=False, drop_last=False, rank=0, world=1, seed=0, depth=3):
        if batch_size < 1 or depth < 2:
            raise ValueError("batch_size must be >= 1 and depth >= 2")
        if not 0 <= rank < world:
            raise ValueError(f"rank {rank} outside world of size {world}")
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, batch_size, shuffle, drop_last
        self.rank, self.world, self.seed, self.depth, self.epoch = rank, world, seed, depth, 0
        self._fields = self._field_plan()

    def set_epoch(self, epoch):
        self.epoch = epoch

    def _field_plan(self):
        """Per output tensor: list of (memmap key, channel slice) concatenated on the channel axis, and a scale."""
        d = self.dataset
        if isinstance(d, dataloaderImgDepthAmpPhs):
            return [([("img", slice(None)), ("depth", slice(0, 1))], 1.0), ([("amp", slice(None))], 1.0), ([("phs", slice(None))], 1.0)]
        if isinstance(d, dataloaderAmpPIPhs):
            return [([("amp", slice(None))], 1.0), ([("phs", slice(None))], 2 * np.pi)]
        if isinstance(d, dataloaderImgDepth):
            return [([("img", slice(None)), ("depth", slice(0, 1))], 1.0)]
        raise TypeError(f"unsupported dataset {type(d).__name__}")

    def _indices(self):
        n = len(self.dataset)
        if self.world > 1:  # torch.utils.data.distributed.DistributedSampler
            if self.shuffle:
                g = torch.Generator()
                g.manual_seed(self.seed + self.epoch)
                idx = torch.randperm(n, generator=g).tolist()
            else:
                idx = list(range(n))
            total = -(-n // self.world) * self.world  # padded by wrapping around, as the sampler does by default
            idx += (idx * (-(-(total - n) // max(n, 1)) + 1))[: total - n] if total > n else []
            return idx[self.rank:total:self.world]
        if self.shuffle:  # torch.utils.data.RandomSampler with generator=None
            torch.empty((), dtype=torch.int64).random_()  # DataLoader's iterator draws its base seed first
            g = torch.Generator()
            g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
            return torch.randperm(n, generator=g).tolist()
        return list(range(n))

    def __len__(self):
        n = len(self.dataset)
        if self.world > 1:
            n = -(-n // self.world)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def _gather(self, batch_idx, staging):
        """memmap rows -> staging tensors (numpy views share their memory)."""
        d = self.dataset
        order = np.argsort(batch_idx)  # ascending file offsets
        for (parts, scale), dst in zip(self._fields, staging):
            view = dst.numpy()[: len(batch_idx)]
            c0 = 0
            for key, sl in parts:
                src = d._maps[key]
                c = len(range(*sl.indices(src.shape[1])))
                for j in order:
                    view[j, c0:c0 + c] = src[batch_idx[j], sl]
                c0 += c
            if scale != 1.0:
                view *= np.float32(scale)

    def __iter__(self):
        import queue
        import threading

        d = self.dataset
        idx = self._indices()
        B = self.batch_size
        batches = [idx[i:i + B] for i in range(0, len(idx), B)]
        if self.drop_last and batches and len(batches[-1]) < B:
            batches.pop()
        on_gpu = d.device.type == "cuda"
        chans = [sum(len(range(*sl.indices(d._maps[k].shape[1]))) for k, sl in parts) for parts, _ in self._fields]
        shape = lambda c: (B, c, d.dataShape[2], d.dataShape[3])  # noqa: E731
        slots = [[torch.empty(shape(c), dtype=torch.float32, pin_memory=on_gpu) for c in chans] for _ in range(self.depth)]
        free, ready = queue.Queue(), queue.Queue(maxsize=self.depth)
        for s in range(self.depth):
            free.put(s)
        copy_stream = torch.cuda.Stream(d.device) if on_gpu else None
        stop = threading.Event()

        def producer():
            try:
                for b in batches:
                    s = free.get()
                    if stop.is_set():
                        return
                    self._gather(b, slots[s])
                    if on_gpu:
                        with torch.cuda.stream(copy_stream):
                            out = [t[: len(b)].to(d.device, non_blocking=True) for t in slots[s]]
                            ev = torch.cuda.Event()
                            ev.record(copy_stream)
                    else:
                        out, ev = [t[: len(b)].clone() for t in slots[s]], None
                    ready.put((s, out, ev))
                ready.put(None)
            except BaseException as e:  # surfaced in the consumer
                ready.put(e)

        th = threading.Thread(target=producer, name="lhg-prefetch", daemon=True)
        th.start()
        try:
            while True:
                item = ready.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                s, out, ev = item
                if ev is not None:
                    torch.cuda.current_stream(d.device).wait_event(ev)
                    for t in out:
                        t.record_stream(torch.cuda.current_stream(d.device))
                    ev.synchronize()  # the staging slot may be refilled once its copy has completed
                free.put(s)
                yield tuple(out) if len(out) > 1 else out[0]
        finally:
            stop.set()
            free.put(0)
            th.join(timeout=10)
