"""COCO keypoint dataset + batch builder (drop-in for the reference's datasets/coco_dataset.py: same record fields, same batch dict).

File I/O without the two packages the image lacks: the annotation file is plain JSON (read directly instead of through pycocotools,
same filtering as coco_dataset.py:64-115) and images are decoded with PIL instead of cv2.imread + cvtColor (both give RGB uint8; the JPEG
decoders may differ by an LSB -- unpinned, outside the path).  `__getitem__` returns the DECODED image and the transformed record; the
crop / normalise / target generation of a whole batch runs on the device in `DeviceBatcher` (pk_affine_crop_normalize + pk_gaussian_target).
"""
import copy
import json
import os
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from .generate_heatmap import generate_target
from .transforms import Compose, DeviceCropper, get_train_transforms, get_val_transforms


class COCOPoseDataset(torch.utils.data.Dataset):
    def __init__(self, data_root: str, ann_file: str, img_prefix: str, input_size=(192, 256), heatmap_size=(48, 64), sigma: float = 2.0,
                 num_keypoints: int = 17, transforms: Optional[Compose] = None, is_train: bool = True, flip_pairs=None):
        self.data_root = data_root
        self.ann_file = os.path.join(data_root, ann_file)
        self.img_prefix = os.path.join(data_root, img_prefix)
        self.input_size, self.heatmap_size = np.array(input_size), np.array(heatmap_size)
        self.sigma, self.num_keypoints, self.transforms, self.is_train = sigma, num_keypoints, transforms, is_train
        self.flip_pairs = flip_pairs or [(1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12), (13, 14), (15, 16)]
        with open(self.ann_file) as f:
            self.coco = json.load(f)
        self.db = self._load_annotations()
        print(f"Loaded {len(self.db)} samples from {self.ann_file}")

    def _load_annotations(self) -> List[Dict]:
        """coco_dataset.py:64-115: one record per person annotation with keypoints, bbox clipped to the image, scale x 1.25."""
        by_img: Dict[int, list] = {}
        for ann in self.coco.get("annotations", []):
            if not ann.get("iscrowd", 0):
                by_img.setdefault(ann["image_id"], []).append(ann)
        db = []
        for info in self.coco.get("images", []):
            for ann in by_img.get(info["id"], []):
                if ann.get("num_keypoints", 0) == 0:
                    continue
                x, y, w, h = ann["bbox"]
                if w <= 0 or h <= 0:
                    continue
                x1, y1, x2, y2 = max(0, x), max(0, y), min(info["width"], x + w), min(info["height"], y + h)
                if x2 <= x1 or y2 <= y1:
                    continue
                kp = np.array(ann["keypoints"]).reshape(-1, 3)
                db.append({"image_file": os.path.join(self.img_prefix, info["file_name"]), "image_id": info["id"], "ann_id": ann["id"],
                           "center": np.array([(x1 + x2) / 2, (y1 + y2) / 2], dtype=np.float32),
                           "scale": np.array([x2 - x1, y2 - y1], dtype=np.float32) * 1.25,
                           "bbox": np.array([x1, y1, x2, y2], dtype=np.float32), "keypoints": kp[:, :2].astype(np.float32),
                           "keypoints_visible": kp[:, 2].astype(np.float32), "area": ann.get("area", w * h)})
        return db

    def __len__(self) -> int:
        return len(self.db)

    def __getitem__(self, idx: int) -> Dict[str, Any]:
        from PIL import Image
        rec = copy.deepcopy(self.db[idx])
        with Image.open(rec["image_file"]) as im:
            img = np.asarray(im.convert("RGB"))
        data = dict(rec, img=img, img_width=img.shape[1], flip_pairs=self.flip_pairs, flip=False)
        if self.transforms is not None:
            data = self.transforms(data)
        return data


def collate_records(samples: List[Dict]) -> List[Dict]:
    return samples          # images have different sizes: the device batcher takes the list


class DeviceBatcher:
    """Iterates a loader of record lists and yields the reference's batch dict (coco_dataset.py:169-183) with everything on the device:
    `img` (B,3,H,W) fp32 normalised (+ `img_nhwc8`, the same pixels as bf16 8-channel NHWC = the stem's input), `target`, `target_weight`
    from the T1 kernel, `keypoints`, `keypoints_visible`, `meta`.

    Prefetch (the role of the reference's DataLoader workers + pin_memory, coco_dataset.py:253-306): batch n + 1 is staged into a pinned
    buffer, copied, cropped / normalised (pk_affine_crop_normalize) and turned into targets (pk_gaussian_target) on a SIDE stream while the
    consumer's stream runs step n; handing a batch out costs one event wait.  `prefetch=False` does the same work on the consumer's stream."""

    def __init__(self, loader, cfg, device="cuda", prefetch=True, nchw=True):
        self.loader, self.cfg, self.device = loader, cfg, torch.device(device)
        self.cropper = DeviceCropper(cfg.data.input_size, device, nchw=nchw)
        self.dataset = getattr(loader, "dataset", None)
        self.prefetch = prefetch and torch.cuda.is_available() and self.device.type == "cuda"
        self._stream = torch.cuda.Stream(device=self.device) if self.prefetch else None
        self._kp_pinned = [None, None]
        self._kp_ev = [None, None]
        self._kp_slot = 0

    def __len__(self):
        return len(self.loader)

    def _keypoints_to_device(self, samples):
        """keypoints + visibility through a small pinned buffer (a pageable .to(device) would block the host behind the side stream)."""
        kp = np.stack([s["keypoints"] for s in samples]).astype(np.float32)
        vis = np.stack([s["keypoints_visible"] for s in samples]).astype(np.float32)
        both = np.concatenate([kp.reshape(-1), vis.reshape(-1)])
        if not torch.cuda.is_available():
            t = torch.from_numpy(both).to(self.device)
        else:
            i = self._kp_slot = self._kp_slot ^ 1
            if self._kp_ev[i] is not None:
                self._kp_ev[i].synchronize()
            if self._kp_pinned[i] is None or self._kp_pinned[i].numel() < both.size:
                self._kp_pinned[i] = torch.empty(both.size, dtype=torch.float32).pin_memory()
            self._kp_pinned[i][:both.size] = torch.from_numpy(both)
            t = self._kp_pinned[i][:both.size].to(self.device, non_blocking=True)
            self._kp_ev[i] = torch.cuda.Event()
            self._kp_ev[i].record()
        return t[:kp.size].view(kp.shape), t[kp.size:].view(vis.shape)

    def _build(self, samples):
        d = self.cfg.data
        img32, img16 = self.cropper([s["img"] for s in samples], [s["matrix"] for s in samples], [s.get("flip", False) for s in samples])
        kp, vis = self._keypoints_to_device(samples)
        target, weight = generate_target(kp, vis, d.input_size, d.heatmap_size, d.sigma)
        f32 = lambda k: torch.from_numpy(np.stack([np.asarray(s[k], np.float32) for s in samples]))
        return {"img": img32, "img_nhwc8": img16, "target": target, "target_weight": weight, "keypoints": kp, "keypoints_visible": vis,
                "meta": {"image_id": torch.tensor([int(s["image_id"]) for s in samples]), "ann_id": torch.tensor([int(s["ann_id"]) for s in samples]),
                         "center": f32("center"), "scale": f32("scale"), "bbox": f32("bbox"),
                         "area": torch.tensor([float(s["area"]) for s in samples])}}

    def _prepare(self, samples):
        """Enqueue the device work of one batch on the side stream -> (batch, event that marks it ready)."""
        with torch.cuda.stream(self._stream):
            batch = self._build(samples)
            ev = torch.cuda.Event()
            ev.record()
        return batch, ev

    def _hand_out(self, pending):
        batch, ev = pending
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        for v in batch.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(cur)
        return batch

    def __iter__(self):
        if not self.prefetch:
            for samples in self.loader:
                yield self._build(samples)
            return
        it = iter(self.loader)
        samples = next(it, None)
        pending = self._prepare(samples) if samples is not None else None
        while pending is not None:
            batch = self._hand_out(pending)
            samples = next(it, None)                   # batch n + 1 goes to the side stream before batch n is handed out: its copy and
            pending = self._prepare(samples) if samples is not None else None      # kernels run beside the consumer's step n
            yield batch


def build_coco_dataloader(cfg, is_train: bool = True, device="cuda"):
    """coco_dataset.py:253-306 with the image work moved to the device."""
    d, t = cfg.data, cfg.train
    tf = get_train_transforms(d.input_size, t.flip_prob, t.rotation_factor, t.scale_factor) if is_train else get_val_transforms(d.input_size)
    ds = COCOPoseDataset(d.data_root, d.train_ann if is_train else d.val_ann, d.train_img_prefix if is_train else d.val_img_prefix, d.input_size,
                         d.heatmap_size, d.sigma, d.num_keypoints, tf, is_train, d.flip_pairs)
    loader = torch.utils.data.DataLoader(ds, batch_size=t.batch_size, shuffle=is_train, num_workers=t.num_workers, collate_fn=collate_records,
                                         drop_last=is_train, persistent_workers=t.num_workers > 0)
    return DeviceBatcher(loader, cfg, device)
