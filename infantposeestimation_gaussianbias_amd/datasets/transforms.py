"""Data augmentation / crop transforms (drop-in names for the reference's datasets/transforms.py), split for the MI355X pipeline:

  * the per-sample DECISIONS and the 17-keypoint arithmetic (flip, half-body, scale / rotation draws, the affine matrix, keypoint
    transform and visibility) stay on the host -- a few dozen flops per sample, same numpy expressions and the same order of
    `np.random` draws as the reference, so a seeded run consumes the RNG stream identically;
  * the image work (affine crop, mirror, BGR->RGB, ToTensor, normalise) is NOT done per sample with OpenCV: the transforms only record
    `data['matrix']` / `data['flip']`, and `DeviceCropper` warps the whole batch in one kernel (pk_affine_crop_normalize) straight into
    the network's input layout.  At ~3 000 img/s per GPU the reference's 4 DataLoader workers doing cv2.warpAffine would be ~7x too slow.
"""
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .. import _lib
from .._lib import call, stream_ptr

MEAN = np.array([0.485, 0.456, 0.406], np.float32)      # datasets/coco_dataset.py:160-161
STD = np.array([0.229, 0.224, 0.225], np.float32)


class Compose:
    def __init__(self, transforms: List):
        self.transforms = transforms

    def __call__(self, data: Dict) -> Dict:
        for t in self.transforms:
            data = t(data)
        return data


def _get_dir(src_point, rot_rad):
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return np.array([src_point[0] * cs - src_point[1] * sn, src_point[0] * sn + src_point[1] * cs])


def _get_3rd_point(a, b):
    direct = a - b
    return b + np.array([-direct[1], direct[0]], dtype=np.float32)


def get_affine_matrix(center, scale, output_size, rot: float = 0.0) -> np.ndarray:
    """transforms.py:58-88: the 2x3 float64 matrix cv2.getAffineTransform returns for the three point pairs (6x6 solve)."""
    src_w, dst_w, dst_h = scale[0], output_size[0], output_size[1]
    src_dir = _get_dir([0, src_w * -0.5], np.pi * rot / 180)
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src, dst = np.zeros((3, 2), dtype=np.float32), np.zeros((3, 2), dtype=np.float32)
    src[0, :], src[1, :] = center, center + src_dir
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir
    src[2, :], dst[2, :] = _get_3rd_point(src[0, :], src[1, :]), _get_3rd_point(dst[0, :], dst[1, :])
    a, b = np.zeros((6, 6)), np.zeros(6)
    for i in range(3):
        a[i, 0:2], a[i, 2], a[i + 3, 3:5], a[i + 3, 5] = src[i], 1.0, src[i], 1.0
        b[i], b[i + 3] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(a, b).reshape(2, 3)


def invert_affine(m) -> np.ndarray:
    """Destination -> source matrix exactly as cv::warpAffine derives it (float64)."""
    m = np.asarray(m, np.float64).copy().reshape(6)
    d = m[0] * m[4] - m[1] * m[3]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[4] * d, m[0] * d
    m[0], m[1], m[3], m[4] = a11, m[1] * -d, m[3] * -d, a22
    b1, b2 = -m[0] * m[2] - m[1] * m[5], -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


class TopdownAffine:
    """Records the crop matrix and applies it to the visible keypoints (transforms.py:22-56); the image is warped later, on the device."""

    def __init__(self, input_size: Tuple[int, int]):
        self.input_size = np.array(input_size)

    def _matrix(self, data):
        return get_affine_matrix(data['center'], data['scale'], self.input_size, 0)

    def __call__(self, data: Dict) -> Dict:
        trans = self._matrix(data)
        kp = data['keypoints']
        for i in range(len(kp)):
            if data['keypoints_visible'][i] > 0:
                kp[i] = np.dot(trans, np.array([kp[i][0], kp[i][1], 1.]).T)[:2]
        data['matrix'], data['keypoints'] = trans, kp
        return data


class TopdownAffineWithRotation(TopdownAffine):
    """transforms.py:196-232: + rotation; a visible keypoint that leaves the crop becomes invisible."""

    def _matrix(self, data):
        return get_affine_matrix(data['center'], data['scale'], self.input_size, data.get('rotation', 0))

    def __call__(self, data: Dict) -> Dict:
        trans = self._matrix(data)
        kp, vis = data['keypoints'], data['keypoints_visible']
        for i in range(len(kp)):
            if vis[i] > 0:
                kp[i] = np.dot(trans, np.array([kp[i][0], kp[i][1], 1.]).T)[:2]
                if kp[i, 0] < 0 or kp[i, 0] >= self.input_size[0] or kp[i, 1] < 0 or kp[i, 1] >= self.input_size[1]:
                    vis[i] = 0
        data['matrix'], data['keypoints'] = trans, kp
        return data


class RandomFlip:
    """transforms.py:108-150; the image itself is mirrored by the crop kernel (`data['flip']`)."""

    def __init__(self, flip_prob: float = 0.5, rng=None):
        self.flip_prob, self.rng = flip_prob, rng or np.random

    def __call__(self, data: Dict) -> Dict:
        if self.rng.random() < self.flip_prob:
            w = data['img_width']
            data['center'][0] = w - data['center'][0] - 1
            kp, vis = data['keypoints'], data['keypoints_visible']
            kp[:, 0] = w - kp[:, 0] - 1
            for a, b in data.get('flip_pairs', []):
                kp[a], kp[b] = kp[b].copy(), kp[a].copy()
                vis[a], vis[b] = vis[b], vis[a]
            data['flip'] = not data.get('flip', False)
        return data


class RandomBBoxTransform:
    """transforms.py:153-193."""

    def __init__(self, rotation_factor: float = 40.0, scale_factor: Tuple[float, float] = (0.5, 1.5), rotation_prob: float = 0.6, rng=None):
        self.rotation_factor, self.scale_factor, self.rotation_prob, self.rng = rotation_factor, scale_factor, rotation_prob, rng or np.random

    def __call__(self, data: Dict) -> Dict:
        data['scale'] = data['scale'] * self.rng.uniform(self.scale_factor[0], self.scale_factor[1])
        if self.rng.random() < self.rotation_prob:
            data['rotation'] = np.clip(self.rng.randn() * self.rotation_factor, -self.rotation_factor * 2, self.rotation_factor * 2)
        else:
            data['rotation'] = 0
        return data


class RandomHalfBody:
    """transforms.py:235-290."""
    UPPER_BODY_IDS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10]
    LOWER_BODY_IDS = [11, 12, 13, 14, 15, 16]

    def __init__(self, prob: float = 0.3, min_keypoints: int = 3, rng=None):
        self.prob, self.min_keypoints, self.rng = prob, min_keypoints, rng or np.random

    def __call__(self, data: Dict) -> Dict:
        if self.rng.random() > self.prob:
            return data
        kp, vis = data['keypoints'], data['keypoints_visible']
        upper = [kp[i] for i in self.UPPER_BODY_IDS if i < len(vis) and vis[i] > 0]
        lower = [kp[i] for i in self.LOWER_BODY_IDS if i < len(vis) and vis[i] > 0]
        if len(upper) >= self.min_keypoints and len(lower) >= self.min_keypoints:
            sel = upper if self.rng.random() < 0.5 else lower
        elif len(upper) >= self.min_keypoints:
            sel = upper
        elif len(lower) >= self.min_keypoints:
            sel = lower
        else:
            return data
        sel = np.array(sel)
        lo, hi = sel.min(axis=0), sel.max(axis=0)
        data['center'] = sel.mean(axis=0)
        data['scale'] = np.maximum(np.array([hi[0] - lo[0], hi[1] - lo[1]]) * 1.5, data['scale'] * 0.5)
        return data


def get_train_transforms(input_size, flip_prob=0.5, rotation_factor=40.0, scale_factor=(0.5, 1.5), rng=None) -> Compose:
    return Compose([RandomFlip(flip_prob, rng), RandomHalfBody(0.3, rng=rng), RandomBBoxTransform(rotation_factor, scale_factor, rng=rng),
                    TopdownAffineWithRotation(input_size)])


def get_val_transforms(input_size) -> Compose:
    return Compose([TopdownAffine(input_size)])


_CROP_DTYPE = np.dtype([("off", "<i8"), ("H", "<i4"), ("W", "<i4"), ("flip", "<i4"), ("bgr", "<i4"), ("minv", "<f8", (6,))])


class DeviceCropper:
    """Batched affine crop + mirror + normalise on the GPU.  `images`: list of (H,W,3) uint8 arrays (numpy or torch, host);
    `matrices`: the forward 2x3 crop matrices; -> (fp32 NCHW batch | None, bf16 NHWC-8 batch | None)."""

    def __init__(self, input_size, device="cuda", nchw=True, nhwc8=True):
        self.w, self.h, self.device, self.nchw, self.nhwc8 = int(input_size[0]), int(input_size[1]), torch.device(device), nchw, nhwc8
        # two reusable pinned staging buffers (pin_memory() per batch is a hipHostMalloc of ~15 MB each time); a slot is rewritten only
        # after the host-to-device copy that last read it has completed (its event)
        self._pinned, self._pin_ev, self._slot = [None, None], [None, None], 0

    def _staging(self, nbytes):
        if not torch.cuda.is_available():
            return torch.empty(nbytes, dtype=torch.uint8), None
        i = self._slot = self._slot ^ 1
        if self._pin_ev[i] is not None:
            self._pin_ev[i].synchronize()
        if self._pinned[i] is None or self._pinned[i].numel() < nbytes:
            self._pinned[i] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8).pin_memory()
        return self._pinned[i][:nbytes], i

    def __call__(self, images, matrices, flips=None, bgr=False):
        B = len(images)
        if B == 0 or len(matrices) != B:
            raise _lib.PoseKernelError("DeviceCropper: need one matrix per image")
        desc = np.zeros(B, _CROP_DTYPE)
        sizes, off = [], 0
        for i, im in enumerate(images):
            im = im.numpy() if torch.is_tensor(im) else np.asarray(im)
            if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
                raise _lib.PoseKernelError(f"DeviceCropper: image {i} must be (H,W,3) uint8, got {im.dtype} {im.shape}")
            desc[i] = (off, im.shape[0], im.shape[1], int(bool(flips[i])) if flips is not None else 0, int(bool(bgr)), invert_affine(matrices[i]))
            sizes.append(im)
            off += (im.size + 15) // 16 * 16
        nd = desc.nbytes
        host, slot = self._staging(off + (nd + 15) // 16 * 16)          # [images | descriptor table]: ONE host-to-device copy
        # plain memcpy through a numpy view: a torch slice assignment fans every 0.9 MB image out over all intra-op threads (128 on the
        # GPU host: 21 ms per 64 images against 1.4 ms; scripts/probes/cropper_phases.py)
        hv = host.numpy()
        for i, im in enumerate(sizes):
            o = int(desc[i]["off"])
            np.copyto(hv[o:o + im.size], np.ascontiguousarray(im).reshape(-1))
        np.copyto(hv[off:off + nd], desc.view(np.uint8))
        dev_buf = host.to(self.device, non_blocking=True)
        if slot is not None:
            self._pin_ev[slot] = torch.cuda.Event()
            self._pin_ev[slot].record()
        src, dtab = dev_buf[:off], dev_buf[off:off + nd]
        out32 = torch.empty(B, 3, self.h, self.w, dtype=torch.float32, device=self.device) if self.nchw else None
        out16 = torch.empty(B, self.h, self.w, 8, dtype=torch.bfloat16, device=self.device) if self.nhwc8 else None
        call("pk_affine_crop_normalize", src, dtab, B, self.w, self.h, out32, out16, MEAN.ctypes.data, STD.ctypes.data, stream_ptr())
        dev_buf.record_stream(torch.cuda.current_stream())
        return out32, out16
