"""Host-side input pipeline of reference utils.py / utils_gray.py without cv2 / torchvision / skimage
(none of which exist in the MI355X image): PIL + numpy only.

Semantics kept from the reference:
  * images are read in OpenCV channel order (BGR) and scaled to [0,1] by to_tensor, no mean/std,
    no resize (utils.py:151,90);
  * masks: colour datasets threshold  > 127 -> 1 (utils.py:156-157); grayscale datasets >= 127 -> 1 and the
    image is read as one channel (utils_gray.py:151,159-160);
  * JointTransform2D: optional random crop, horizontal flip with probability p_flip drawn from
    np.random (so train.py's np.random.seed(3000) governs it), long (class-index) masks.
Colour jitter / random affine are accepted and rejected loudly if requested: train.py never enables them
(color_jitter_params=None, p_random_affine default 0; train.py:85-86).
"""
from __future__ import annotations

import os
import threading
from collections import defaultdict
from numbers import Number

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset


def to_long_tensor(pic):
    return torch.from_numpy(np.array(pic, np.uint8)).long()


def correct_dims(*images):
    out = [np.expand_dims(img, axis=2) if img.ndim == 2 else img for img in images]
    return out[0] if len(out) == 1 else out


def imread(path, gray=False):
    """cv2.imread equivalent: uint8 HxWx3 in BGR order, or HxW when gray (cv2.IMREAD_GRAYSCALE)."""
    with Image.open(path) as im:
        if gray:
            return np.array(im.convert("L"), np.uint8)
        return np.array(im.convert("RGB"), np.uint8)[:, :, ::-1].copy()


def imwrite(path, arr):
    arr = np.asarray(arr)
    if arr.ndim == 3 and arr.shape[2] == 3:
        arr = arr[:, :, ::-1]
    Image.fromarray(arr.astype(np.uint8)).save(path)


def to_tensor(img_hwc_uint8):
    return torch.from_numpy(np.ascontiguousarray(img_hwc_uint8.transpose(2, 0, 1))).float().div_(255.0)


class JointTransform2D:
    def __init__(self, crop=(32, 32), p_flip=0.5, color_jitter_params=(0.1, 0.1, 0.1, 0.1), p_random_affine=0,
                 long_mask=False):
        if color_jitter_params or p_random_affine:
            raise NotImplementedError("colour jitter / random affine need torchvision; train.py never enables them")
        self.crop, self.p_flip, self.long_mask = crop, p_flip, long_mask
        self.color_jitter_params, self.p_random_affine = color_jitter_params, p_random_affine

    def __call__(self, image, mask):
        image, mask = np.asarray(image), np.asarray(mask)
        if self.crop:
            th, tw = self.crop
            h, w = image.shape[:2]
            i = 0 if h == th else int(torch.randint(0, h - th + 1, (1,)).item())
            j = 0 if w == tw else int(torch.randint(0, w - tw + 1, (1,)).item())
            image, mask = image[i:i + th, j:j + tw], mask[i:i + th, j:j + tw]
        if np.random.rand() < self.p_flip:
            image, mask = image[:, ::-1], mask[:, ::-1]
        image = to_tensor(correct_dims(image))
        if self.long_mask:
            mask = to_long_tensor(mask.reshape(mask.shape[0], mask.shape[1]))
        else:
            mask = to_tensor(correct_dims(mask))
        return image, mask


class ImageToImage2D(Dataset):
    """<root>/img/*.png + <root>/labelcol/<same stem>.png  ->  (image CHW float, mask HW long, filename)."""

    def __init__(self, dataset_path, joint_transform=None, one_hot_mask=False, gray=False):
        self.dataset_path = dataset_path
        self.input_path = os.path.join(dataset_path, "img")
        self.output_path = os.path.join(dataset_path, "labelcol")
        self.images_list = os.listdir(self.input_path)
        self.one_hot_mask = one_hot_mask
        self.gray = gray
        self.joint_transform = joint_transform or (lambda x, y: (to_tensor(correct_dims(x)), to_tensor(correct_dims(y))))

    def __len__(self):
        return len(self.images_list)

    def __getitem__(self, idx):
        name = self.images_list[idx]
        image = imread(os.path.join(self.input_path, name), self.gray)
        mask = imread(os.path.join(self.output_path, name[:-3] + "png"), gray=True).copy()
        if self.gray:
            mask = (mask >= 127).astype(np.uint8)
        else:
            mask = (mask > 127).astype(np.uint8)
        image, mask = correct_dims(image, mask)
        image, mask = self.joint_transform(image, mask)
        if self.one_hot_mask:
            assert self.one_hot_mask > 0, "one_hot_mask must be nonnegative"
            mask = torch.zeros((self.one_hot_mask, mask.shape[1], mask.shape[2])).scatter_(0, mask.long(), 1)
        return image, mask, name


class Image2D(Dataset):
    def __init__(self, dataset_path, transform=None, gray=False):
        self.dataset_path = dataset_path
        self.input_path = os.path.join(dataset_path, "img")
        self.images_list = os.listdir(self.input_path)
        self.gray = gray
        self.transform = transform or (lambda x: to_tensor(correct_dims(x)))

    def __len__(self):
        return len(self.images_list)

    def __getitem__(self, idx):
        name = self.images_list[idx]
        return self.transform(correct_dims(imread(os.path.join(self.input_path, name), self.gray))), name


def chk_mkdir(*paths):
    for p in paths:
        if not os.path.exists(p):
            os.makedirs(p)


class Logger:
    def __init__(self, verbose=False):
        self.logs, self.verbose = defaultdict(list), verbose

    def log(self, logs):
        for k, v in logs.items():
            self.logs[k].append(v)
        if self.verbose:
            print(logs)

    def get_logs(self):
        return self.logs

    def to_csv(self, path):
        import pandas as pd
        pd.DataFrame(self.logs).to_csv(path, index=None)


class MetricList:
    def __init__(self, metrics):
        assert isinstance(metrics, dict), "'metrics' must be a dictionary of callables"
        self.metrics = metrics
        self.results = {k: 0.0 for k in metrics}

    def __call__(self, y_out, y_batch):
        for k, fn in self.metrics.items():
            self.results[k] += fn(y_out, y_batch)

    def reset(self):
        self.results = {k: 0.0 for k in self.metrics}

    def get_results(self, normalize=False):
        assert isinstance(normalize, bool) or isinstance(normalize, Number), "'normalize' must be boolean or a number"
        if not normalize:
            return self.results
        return {k: v / normalize for k, v in self.results.items()}


def make_synthetic_dataset(root, n=16, size=128, seed=3000, gray=False):
    """BASELINE.json config 1 plumbing data: n PNG pairs img/NNNN.png (uniform uint8) + labelcol/NNNN.png (0/255)."""
    rng = np.random.RandomState(seed)
    chk_mkdir(os.path.join(root, "img"), os.path.join(root, "labelcol"))
    for k in range(n):
        img = rng.randint(0, 256, (size, size) if gray else (size, size, 3)).astype(np.uint8)
        lab = (rng.rand(size, size) < 0.5).astype(np.uint8) * 255
        Image.fromarray(img).save(os.path.join(root, "img", f"{k:04d}.png"))
        Image.fromarray(lab).save(os.path.join(root, "labelcol", f"{k:04d}.png"))
    return root


# Held by the prefetch thread around its device work (pinned allocations, event waits, H2D copies) and by
# medt_amd.trainer.TrainStep around hipGraph capture: runtime calls from another thread (hipHostMalloc, event queries,
# caching-allocator device allocations) must not land inside a capture -- the epoch-10 gate switch and a new last-batch
# shape both re-capture while the producer is running.
GPU_CAPTURE_LOCK = threading.RLock()


class DevicePrefetcher:
    """Iterate a DataLoader `depth` batches ahead of the training step.

    The reference loop decodes a batch with cv2/PIL on the host and copies it from pageable memory in the step's
    own stream (train.py:90,130-135: num_workers=0, blocking `.to(device)`); at ~4 ms per MI355X step that
    serialises host decode + H2D with the GPU work.  Here a background thread pulls batches from the loader (same
    order, same np.random stream for the flips), stages them in reusable PINNED buffers and issues the H2D copies
    on a dedicated copy stream; the consumer only waits on the copy's event.  Tensors past the first two entries of
    a batch (file names ...) pass through untouched.  On a CPU device it degenerates to plain iteration.
    """

    def __init__(self, loader, device, depth: int = 2):
        self.loader, self.device, self.depth = loader, torch.device(device), max(1, depth)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        if self.device.type != "cuda":
            for batch in self.loader:
                yield batch
            return
        import queue
        import threading
        copy_stream = torch.cuda.Stream(device=self.device)
        q = queue.Queue(maxsize=self.depth)
        ring = {}                      # (shape, dtype) -> list of [pinned buffer, event of its last copy]
        cursor = defaultdict(int)
        stop = threading.Event()

        def staged(t):
            key = (tuple(t.shape), t.dtype)
            bufs = ring.get(key)
            if bufs is None:           # the whole ring of a new batch shape at once (not one hipHostMalloc per step)
                bufs = ring[key] = [[torch.empty(t.shape, dtype=t.dtype).pin_memory(), None]
                                    for _ in range(self.depth + 2)]
            slot = bufs[cursor[key] % len(bufs)]
            cursor[key] += 1
            if slot[1] is not None:
                slot[1].synchronize()              # the copy that last read this staging buffer has finished
            slot[0].copy_(t)
            return slot

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def producer():
            try:
                torch.cuda.set_device(self.device)
                for batch in self.loader:
                    if stop.is_set():
                        return
                    out = list(batch)
                    with GPU_CAPTURE_LOCK, torch.cuda.stream(copy_stream):
                        slots = [staged(t) for t in out[:2]]
                        for i, slot in enumerate(slots):
                            out[i] = slot[0].to(self.device, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(copy_stream)
                        for slot in slots:
                            slot[1] = ev
                    if not put((out, ev)):
                        return
                put(None)
            except BaseException as e:              # surfaces in the consumer
                put(e)

        th = threading.Thread(target=producer, name="medt-prefetch", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                out, ev = item
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                for t in out[:2]:
                    t.record_stream(cur)
                yield tuple(out)
        finally:
            stop.set()
            th.join(timeout=10)
