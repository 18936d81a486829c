#!/usr/bin/env python3
"""Inference script (drop-in for the reference's inference.py: `PoseInference(checkpoint, device, flip_test)` with
preprocess / predict / predict_batch / postprocess, same flags).

Pre- and post-processing run on the device: the BGR -> RGB swap, the affine crop around the bbox (scale x 1.25), ToTensor and the
mean/std normalisation of inference.py:64-110 are ONE kernel per batch (pk_affine_crop_normalize, OpenCV's 8-bit warpAffine arithmetic
as restated in oracle/warp.py), the model's flip-test inference and decode follow (PoseEstimator.inference), and the heat-px -> image
mapping of inference.py:142-175 is one kernel (pk_affine_coords).  `predict_batch` really batches (the reference loops over predict).
Visualisation (cv2 drawing) is outside the path: `visualize` needs OpenCV on the machine.
"""
import argparse
import os
import sys
import time
from typing import List, Optional, Tuple

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from infantposeestimation_gaussianbias_amd.configs import get_config  # noqa: E402
from infantposeestimation_gaussianbias_amd.datasets.transforms import DeviceCropper, get_affine_matrix  # noqa: E402
from infantposeestimation_gaussianbias_amd.models import build_model  # noqa: E402
from infantposeestimation_gaussianbias_amd.utils.postprocess import heatmap_to_image_coords  # noqa: E402


class PoseInference:
    def __init__(self, checkpoint: Optional[str] = None, device: str = 'cuda', flip_test: bool = True, config: Optional[str] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("PoseInference needs an MI355X: the hot path has no CPU implementation")
        self.device = torch.device(device)
        self.flip_test = flip_test
        self.cfg = get_config(config) if config else get_config()
        self.model = build_model(self.cfg).to(self.device).eval()
        if checkpoint and os.path.isfile(checkpoint):
            ckpt = torch.load(checkpoint, map_location="cpu", weights_only=True)
            self.model.load_state_dict(ckpt['model_state_dict'])
            print(f'Loaded checkpoint: {checkpoint}')
        self.input_size = self.cfg.data.input_size          # (w, h)
        self.flip_pairs = self.cfg.data.flip_pairs
        self._crop = DeviceCropper(self.input_size, self.device, nchw=True, nhwc8=False)

    # ---- inference.py:64-110
    @staticmethod
    def _center_scale(img, bbox):
        if bbox is None:
            h, w = img.shape[:2]
            bbox = np.array([0, 0, w, h])
        x1, y1, x2, y2 = bbox
        return np.array([(x1 + x2) / 2, (y1 + y2) / 2]), np.array([x2 - x1, y2 - y1]) * 1.25

    def preprocess_batch(self, imgs: List[np.ndarray], bboxes: Optional[List[Optional[np.ndarray]]] = None):
        """BGR uint8 images (+ bboxes) -> (B,3,H,W) normalised device tensor, centers (B,2), scales (B,2)."""
        cs = [self._center_scale(im, bboxes[i] if bboxes else None) for i, im in enumerate(imgs)]
        mats = [get_affine_matrix(c, s, self.input_size, 0) for c, s in cs]
        x, _ = self._crop(imgs, mats, None, bgr=True)
        return x, np.stack([c for c, _ in cs]), np.stack([s for _, s in cs])

    def preprocess(self, img: np.ndarray, bbox: Optional[np.ndarray] = None):
        x, c, s = self.preprocess_batch([img], [bbox])
        return x, c[0], s[0]

    # ---- inference.py:142-175
    def postprocess(self, keypoints, scores, center, scale):
        """heat-px keypoints (K,2) / (B,K,2) -> original image coordinates."""
        kp = torch.as_tensor(keypoints, dtype=torch.float32, device=self.device)
        single = kp.dim() == 2
        kp = kp[None] if single else kp
        c = torch.as_tensor(np.asarray(center, np.float32).reshape(-1, 2), device=self.device)
        s = torch.as_tensor(np.asarray(scale, np.float32).reshape(-1, 2), device=self.device)
        out = heatmap_to_image_coords(kp, c, s, self.input_size, self.cfg.data.heatmap_size).cpu().numpy()
        return (out[0] if single else out), scores

    @torch.no_grad()
    def predict_batch(self, imgs: List[np.ndarray], bboxes: Optional[List[np.ndarray]] = None) -> List[Tuple[np.ndarray, np.ndarray]]:
        x, centers, scales = self.preprocess_batch(imgs, bboxes)
        if self.flip_test:
            kp, sc = self.model.inference(x, flip=True, flip_pairs=self.flip_pairs)
        else:
            kp, sc = self.model.inference(x, flip=False)
        kp_img, _ = self.postprocess(kp, sc, centers, scales)
        sc = sc.cpu().numpy()
        return [(kp_img[i], sc[i]) for i in range(len(imgs))]

    @torch.no_grad()
    def predict(self, img: np.ndarray, bbox: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
        return self.predict_batch([img], [bbox])[0]

    def visualize(self, img, keypoints, scores, score_threshold: float = 0.3, output_path: Optional[str] = None):
        try:
            import cv2
        except ImportError as e:
            raise RuntimeError("visualize() draws with OpenCV, which is not installed (third-party, outside the inference path)") from e
        vis = img.copy()
        for (x, y), s in zip(keypoints, scores):
            if s >= score_threshold:
                cv2.circle(vis, (int(x), int(y)), 3, (0, 255, 0), -1)
        if output_path:
            cv2.imwrite(output_path, vis)
        return vis


def detect_persons(img: np.ndarray) -> List[np.ndarray]:
    """inference.py:262-276: the reference's placeholder detector = the whole image."""
    h, w = img.shape[:2]
    return [np.array([0, 0, w, h])]


def main(args):
    from PIL import Image
    pose = PoseInference(checkpoint=args.checkpoint, device=args.device, flip_test=not args.no_flip, config=args.config)
    rgb = np.asarray(Image.open(args.input).convert("RGB"))
    img = rgb[:, :, ::-1].copy()                          # the reference hands BGR (cv2.imread) to PoseInference
    bboxes = [np.array(args.bbox)] if args.bbox else detect_persons(img)
    t0 = time.time()
    results = pose.predict_batch([img] * len(bboxes), bboxes)
    print(f'Inference time: {(time.time() - t0) * 1000:.2f} ms')
    for kp, sc in results:
        for k, ((x, y), s) in enumerate(zip(kp, sc)):
            print(f'  kpt {k:2d}: ({x:8.2f}, {y:8.2f})  score {s:.3f}')
    if args.output:
        # inference.py:296-300 of the reference: draw and save.  Drawing is OpenCV's (visualize raises RuntimeError without cv2).
        vis = img
        for n, (kp, sc) in enumerate(results):
            vis = pose.visualize(vis, kp, sc, score_threshold=args.threshold, output_path=args.output if n == len(results) - 1 else None)
        print(f'Result saved to: {args.output}')


if __name__ == '__main__':
    p = argparse.ArgumentParser(description='Pose Estimation Inference')
    p.add_argument('--checkpoint', type=str, default=None)
    p.add_argument('--input', type=str, required=True)
    p.add_argument('--output', type=str, default=None)
    p.add_argument('--device', type=str, default='cuda')
    p.add_argument('--no_flip', action='store_true')
    p.add_argument('--threshold', type=float, default=0.3)
    p.add_argument('--bbox', type=float, nargs=4, default=None)
    p.add_argument('--config', type=str, default=None)
    main(p.parse_args())
