"""Configuration surface (drop-in for the reference's configs/config.py:15-130).

Same dataclass fields and defaults, so `cfg.data.input_size`, `cfg.model.backbone`, `cfg.train.lr` ... read
identically.  Extension over the reference: `get_config(path_or_name)` also accepts
  * a preset name ("hrformer_small", "hrformer_base", "hrnet_w32", "hrnet_w48", "hrnet_w18", "preemie"), or
  * a legacy-format yaml (reference configs/*.yaml, written by root config.py:135-224): `MODEL.NUM_JOINTS`,
    `MODEL.IMAGE_SIZE`, `MODEL.HEATMAP_SIZE`, `MODEL.SIGMA`, `TRAIN.*` are mapped onto the dataclass fields.
"""
import os
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

_COCO_NAMES = ("nose left_eye right_eye left_ear right_ear left_shoulder right_shoulder left_elbow right_elbow "
               "left_wrist right_wrist left_hip right_hip left_knee right_knee left_ankle right_ankle").split()


@dataclass
class DataConfig:
    data_root: str = 'data/coco/'
    train_ann: str = 'annotations/person_keypoints_train2017.json'
    val_ann: str = 'annotations/person_keypoints_val2017.json'
    train_img_prefix: str = 'train2017/'
    val_img_prefix: str = 'val2017/'
    input_size: Tuple[int, int] = (192, 256)     # (width, height)
    heatmap_size: Tuple[int, int] = (48, 64)     # (width, height)
    num_keypoints: int = 17
    sigma: float = 2.0
    keypoint_names: List[str] = field(default_factory=lambda: list(_COCO_NAMES))
    flip_pairs: List[Tuple[int, int]] = field(default_factory=lambda: [(i, i + 1) for i in range(1, 17, 2)])


@dataclass
class ModelConfig:
    backbone: str = 'hrformer_base'
    pretrained: bool = True
    in_channels: int = 3
    base_channels: int = 78
    head_type: str = 'fusion'
    head_in_channels: int = 78
    num_keypoints: int = 17
    hidden_dim: int = 256
    use_target_weight: bool = True
    use_fusion_loss: bool = True
    heatmap_loss_weight: float = 1.0
    offset_loss_weight: float = 1.0
    peak_loss_weight: float = 0.5
    variance_loss_weight: float = 0.1
    overlap_loss_weight: float = 0.05
    shape_loss_weight: float = 0.05
    target_sigma: float = 2.0


@dataclass
class TrainConfig:
    max_epochs: int = 210
    val_interval: int = 10
    batch_size: int = 32
    num_workers: int = 4
    optimizer: str = 'AdamW'
    lr: float = 5e-4
    weight_decay: float = 0.01
    betas: Tuple[float, float] = (0.9, 0.999)
    warmup_epochs: int = 5
    warmup_lr: float = 5e-7
    lr_milestones: List[int] = field(default_factory=lambda: [170, 200])
    lr_gamma: float = 0.1
    flip_prob: float = 0.5
    half_body_prob: float = 0.3
    rotation_factor: float = 40.0
    scale_factor: Tuple[float, float] = (0.5, 1.5)
    save_best: str = 'AP'
    checkpoint_dir: str = 'checkpoints/'
    device: str = 'cuda'
    fp16: bool = True        # "mixed precision on": bf16 on MI355X (no GradScaler needed)


@dataclass
class Config:
    data: DataConfig = field(default_factory=DataConfig)
    model: ModelConfig = field(default_factory=ModelConfig)
    train: TrainConfig = field(default_factory=TrainConfig)
    exp_name: str = 'hrformer_base_coco_256x192'
    seed: int = 42


_PRESETS = {
    # name: (backbone, head_type, channels, (W_in,H_in), (W_hm,H_hm), K, sigma)
    'hrformer_base': ('hrformer_base', 'fusion', 78, (192, 256), (48, 64), 17, 2.0),
    'hrformer_small': ('hrformer_small', 'fusion', 32, (192, 256), (48, 64), 17, 2.0),
    'hrnet_w32': ('hrnet_w32', 'heatmap', 32, (288, 384), (72, 96), 17, 2.0),      # root config.py:135-164
    'hrnet_w48': ('hrnet_w48', 'heatmap', 48, (288, 384), (72, 96), 17, 2.0),
    'hrnet_w18': ('hrnet_w18', 'heatmap', 18, (96, 128), (24, 32), 17, 2.0),       # BASELINE config 1
    'preemie': ('hrformer_base', 'fusion', 78, (288, 384), (72, 96), 13, 1.5),     # BASELINE config 5 (K=13, sigma 1.5)
}


def _apply_preset(cfg: Config, name: str) -> Config:
    bb, head, ch, insz, hmsz, K, sigma = _PRESETS[name]
    cfg.model.backbone, cfg.model.head_type = bb, head
    cfg.model.base_channels = cfg.model.head_in_channels = ch
    cfg.data.input_size, cfg.data.heatmap_size = insz, hmsz
    cfg.data.num_keypoints = cfg.model.num_keypoints = K
    cfg.data.sigma = cfg.model.target_sigma = sigma
    if K != 17:
        cfg.data.keypoint_names = [f'kpt_{i}' for i in range(K)]
        cfg.data.flip_pairs = []
    cfg.exp_name = f'{name}_{insz[1]}x{insz[0]}'
    return cfg


def _apply_legacy_yaml(cfg: Config, path: str) -> Config:
    import yaml
    with open(path) as f:
        y = yaml.safe_load(f) or {}
    m, t = y.get('MODEL', {}) or {}, y.get('TRAIN', {}) or {}
    if 'NUM_JOINTS' in m:
        K = int(m['NUM_JOINTS'])
        cfg.data.num_keypoints = cfg.model.num_keypoints = K
        if K != 17:
            cfg.data.keypoint_names = [f'kpt_{i}' for i in range(K)]
            cfg.data.flip_pairs = []
    if 'IMAGE_SIZE' in m:      # legacy order is [W, H] as written by root config.py (IMAGE_SIZE = [288, 384])
        cfg.data.input_size = (int(m['IMAGE_SIZE'][0]), int(m['IMAGE_SIZE'][1]))
    if 'HEATMAP_SIZE' in m:
        cfg.data.heatmap_size = (int(m['HEATMAP_SIZE'][0]), int(m['HEATMAP_SIZE'][1]))
    if 'SIGMA' in m:
        cfg.data.sigma = cfg.model.target_sigma = float(m['SIGMA'])
    name = str(m.get('NAME', ''))
    if 'w48' in name:
        cfg.model.backbone, cfg.model.head_type, cfg.model.base_channels = 'hrnet_w48', 'heatmap', 48
    elif 'w32' in name:
        cfg.model.backbone, cfg.model.head_type, cfg.model.base_channels = 'hrnet_w32', 'heatmap', 32
    if m.get('FUSED_HEAD'):
        cfg.model.head_type = 'fusion'
    cfg.model.head_in_channels = cfg.model.base_channels
    for src, dst, cast in (('BATCH_SIZE', 'batch_size', int), ('EPOCHS', 'max_epochs', int), ('LR', 'lr', float),
                           ('WEIGHT_DECAY', 'weight_decay', float), ('NUM_WORKERS', 'num_workers', int),
                           ('VAL_INTERVAL', 'val_interval', int)):
        if src in t:
            setattr(cfg.train, dst, cast(t[src]))
    cfg.exp_name = os.path.splitext(os.path.basename(path))[0]
    return cfg


def get_config(source: Optional[str] = None) -> Config:
    """`get_config()` == the reference default. `get_config('hrformer_small')` / `get_config('x.yaml')` extend it."""
    cfg = Config()
    if source is None:
        return cfg
    if source in _PRESETS:
        return _apply_preset(cfg, source)
    if os.path.isfile(source):
        return _apply_legacy_yaml(cfg, source)
    raise ValueError(f"Unknown config source: {source}")
