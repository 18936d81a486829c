#!/usr/bin/env python3
"""Where do the milliseconds of DeviceCropper go?  Host packing into the pinned buffer (torch slice assignment vs numpy copyto), the
host-to-device copy, the crop kernel."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from infantposeestimation_gaussianbias_amd.datasets import transforms as T  # noqa: E402

dev = torch.device("cuda:0")
imgs = [np.random.default_rng(i).integers(0, 255, (480, 640, 3), dtype=np.uint8) for i in range(4)] * 16
mats = [np.array([[0.4, 0.0, -32.0], [0.0, 0.4, 32.0]], np.float64)] * len(imgs)
tot = sum(im.size for im in imgs)
pin = torch.empty(tot, dtype=torch.uint8).pin_memory()
pv = pin.numpy()
print("torch threads", torch.get_num_threads(), "cpus", len(os.sched_getaffinity(0)))


def t(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def pack_torch():
    o = 0
    for im in imgs:
        pin[o:o + im.size] = torch.from_numpy(im.reshape(-1))
        o += im.size


def pack_numpy():
    o = 0
    for im in imgs:
        np.copyto(pv[o:o + im.size], im.reshape(-1))
        o += im.size


print(f"pack (torch slice assignment) {t(pack_torch):.2f} ms; pack (numpy copyto) {t(pack_numpy):.2f} ms")
print(f"host-to-device copy of {tot / 1e6:.0f} MB: {t(lambda: pin.to(dev, non_blocking=True)):.2f} ms")
crop = T.DeviceCropper((192, 256), dev, nchw=False, nhwc8=True)
print(f"DeviceCropper call: {t(lambda: crop(imgs, mats), 6):.2f} ms")
for nt in (1, 4, 16):
    torch.set_num_threads(nt)
    print(f"  torch threads {nt}: pack (torch) {t(pack_torch):.2f} ms, DeviceCropper {t(lambda: crop(imgs, mats), 6):.2f} ms")
