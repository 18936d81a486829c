"""Oracle (test infrastructure only): the input pipeline of the reference restated in numpy -- affine crop, flip, half-body / scale /
rotation augmentation, normalisation (datasets/transforms.py:22-300, datasets/coco_dataset.py:140-183, inference.py:64-140).

The image warp in the reference is `cv2.warpAffine(img, trans, (w, h), flags=cv2.INTER_LINEAR)` (transforms.py:42-47, 212-217) and the matrix
comes from `cv2.getAffineTransform`.  OpenCV (`opencv-python>=4.5.0`, requirements.txt) is a third-party dependency that is NOT in this
image, and the reference holds no fixture of a warped image: **parity unpinned vs cv2**.  What is restated here is OpenCV's published
algorithm for 8-bit images:
  * getAffineTransform: the 2x3 matrix that maps three source points onto three destination points (6x6 linear system, float64);
  * warpAffine without WARP_INVERSE_MAP inverts the matrix in float64, then walks the destination in FIXED POINT: AB_BITS = 10 fractional bits
    for the coordinates (rounded with cvRound = round-half-even), INTER_BITS = 5 bits of sub-pixel position, bilinear weights
    (32 - a)(32 - b) * 32 etc. that sum to 2^15, result = (sum + 2^14) >> 15; BORDER_CONSTANT with value 0 for taps outside the image.
The HIP kernel (pk_affine_crop_normalize) does the same integer arithmetic and is tested bit-exact against THIS file.
"""
import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], np.float32)
STD = np.array([0.229, 0.224, 0.225], np.float32)
UPPER_BODY_IDS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10]
LOWER_BODY_IDS = [11, 12, 13, 14, 15, 16]


# ------------------------------------------------------------------------------ matrices (transforms.py:58-104)
def _get_dir(src_point, rot_rad):
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return np.array([src_point[0] * cs - src_point[1] * sn, src_point[0] * sn + src_point[1] * cs])


def _third_point(a, b):
    d = a - b
    return b + np.array([-d[1], d[0]], dtype=np.float32)


def solve_affine(src, dst):
    """2x3 float64 matrix with M @ [x, y, 1] = dst for the three point pairs (what cv2.getAffineTransform returns)."""
    a, b = np.zeros((6, 6)), np.zeros(6)
    for i in range(3):
        a[i, 0:2], a[i, 2] = src[i], 1.0
        a[i + 3, 3:5], a[i + 3, 5] = src[i], 1.0
        b[i], b[i + 3] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(a, b).reshape(2, 3)


def get_affine_transform(center, scale, output_size, rot=0.0):
    src_w, (dst_w, dst_h) = scale[0], output_size
    src_dir = _get_dir([0, src_w * -0.5], np.pi * rot / 180)
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src, dst = np.zeros((3, 2), np.float32), np.zeros((3, 2), np.float32)
    src[0], src[1] = center, center + src_dir
    dst[0] = [dst_w * 0.5, dst_h * 0.5]
    dst[1] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir
    src[2], dst[2] = _third_point(src[0], src[1]), _third_point(dst[0], dst[1])
    return solve_affine(np.float32(src).astype(np.float64), np.float32(dst).astype(np.float64))


def invert_affine(m):
    m = np.asarray(m, np.float64).copy().reshape(6)
    d = m[0] * m[4] - m[1] * m[3]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[4] * d, m[0] * d
    m[0], m[1], m[3], m[4] = a11, m[1] * -d, m[3] * -d, a22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


# ------------------------------------------------------------------------------ warp (OpenCV imgwarp.cpp: WarpAffineInvoker + remapBilinear<uchar>)
def _sat_int(v):
    return np.clip(np.rint(v), -2147483648, 2147483647).astype(np.int64)


def warp_affine_u8(img, m_fwd, out_wh, flip=False):
    """img (H,W,C) uint8 -> (h,w,C) uint8; `flip` mirrors the source columns first (RandomFlip: img[:, ::-1])."""
    H, W, C = img.shape
    w, h = int(out_wh[0]), int(out_wh[1])
    mi = invert_affine(m_fwd)
    xs = np.arange(w, dtype=np.float64)
    adelta, bdelta = _sat_int(mi[0] * xs * 1024), _sat_int(mi[3] * xs * 1024)
    ys = np.arange(h, dtype=np.float64)
    x0 = _sat_int((mi[1] * ys + mi[2]) * 1024) + 16
    y0 = _sat_int((mi[4] * ys + mi[5]) * 1024) + 16
    X = (x0[:, None] + adelta[None, :]) >> 5
    Y = (y0[:, None] + bdelta[None, :]) >> 5
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
    a, b = X & 31, Y & 31
    wts = [(32 - a) * (32 - b) * 32, a * (32 - b) * 32, (32 - a) * b * 32, a * b * 32]
    acc = np.zeros((h, w, C), np.int64)
    for (dx, dy), wt in zip(((0, 0), (1, 0), (0, 1), (1, 1)), wts):
        xx, yy = sx + dx, sy + dy
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        xc = np.clip(xx, 0, W - 1)
        xc = (W - 1 - xc) if flip else xc
        v = img[np.clip(yy, 0, H - 1), xc].astype(np.int64) * ok[..., None]
        acc += v * wt[..., None]
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def normalize_chw(img_u8_hwc):
    """coco_dataset.py:158-163: float32 /255, (x - mean) / std, CHW."""
    t = img_u8_hwc.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)
    return (t - MEAN[:, None, None]) / STD[:, None, None]


# ------------------------------------------------------------------------------ keypoints / augmentation decisions
def affine_keypoints(keypoints, visible, m_fwd, input_size=None):
    """transforms.py:50-53 (TopdownAffine) / :219-227 (with rotation: a visible keypoint that leaves the crop becomes invisible)."""
    kp, vis = keypoints.copy(), visible.copy()
    for i in range(len(kp)):
        if vis[i] > 0:
            kp[i] = (m_fwd @ np.array([kp[i, 0], kp[i, 1], 1.0]))[:2]
            if input_size is not None and (kp[i, 0] < 0 or kp[i, 0] >= input_size[0] or kp[i, 1] < 0 or kp[i, 1] >= input_size[1]):
                vis[i] = 0
    return kp, vis


def flip_record(img_width, center, keypoints, visible, flip_pairs):
    """transforms.py:128-143 without the image (the warp mirrors the columns)."""
    center, kp, vis = center.copy(), keypoints.copy(), visible.copy()
    center[0] = img_width - center[0] - 1
    kp[:, 0] = img_width - kp[:, 0] - 1
    for a, b in flip_pairs:
        kp[[a, b]] = kp[[b, a]]
        vis[[a, b]] = vis[[b, a]]
    return center, kp, vis


def half_body(keypoints, visible, scale, take_upper_if_both, min_keypoints=3):
    """transforms.py:252-290: -> (center, scale) of the chosen half, or None when neither half has enough visible keypoints."""
    up = [keypoints[i] for i in UPPER_BODY_IDS if visible[i] > 0]
    lo = [keypoints[i] for i in LOWER_BODY_IDS if visible[i] > 0]
    if len(up) >= min_keypoints and len(lo) >= min_keypoints:
        sel = up if take_upper_if_both else lo
    elif len(up) >= min_keypoints:
        sel = up
    elif len(lo) >= min_keypoints:
        sel = lo
    else:
        return None
    sel = np.array(sel)
    c = sel.mean(axis=0)
    wh = sel.max(axis=0) - sel.min(axis=0)
    return c, np.maximum(wh * 1.5, scale * 0.5)


def train_sample(img, rec, input_size, rng, flip_prob=0.5, rotation_factor=40.0, scale_factor=(0.5, 1.5), flip_pairs=()):
    """get_train_transforms (transforms.py:293-310) on one record with the draws in the reference's order (np.random.random / uniform /
    randn on `rng`): RandomFlip, RandomHalfBody(0.3), RandomBBoxTransform(rotation_prob 0.6), TopdownAffineWithRotation, normalise."""
    center, scale = rec["center"].copy(), rec["scale"].copy()
    kp, vis = rec["keypoints"].copy(), rec["keypoints_visible"].copy()
    flip = rng.random() < flip_prob
    if flip:
        center, kp, vis = flip_record(img.shape[1], center, kp, vis, flip_pairs)
    if not (rng.random() > 0.3):
        up = sum(1 for i in UPPER_BODY_IDS if vis[i] > 0) >= 3
        lo = sum(1 for i in LOWER_BODY_IDS if vis[i] > 0) >= 3
        take_upper = (rng.random() < 0.5) if (up and lo) else True
        hb = half_body(kp, vis, scale, take_upper)
        if hb is not None:
            center, scale = hb
    scale = scale * rng.uniform(scale_factor[0], scale_factor[1])
    rot = float(np.clip(rng.randn() * rotation_factor, -rotation_factor * 2, rotation_factor * 2)) if rng.random() < 0.6 else 0.0
    m = get_affine_transform(center, scale, input_size, rot)
    crop = warp_affine_u8(img, m, input_size, flip=flip)
    kp, vis = affine_keypoints(kp, vis, m, input_size)
    return normalize_chw(crop), kp.astype(np.float32), vis.astype(np.float32), dict(center=center, scale=scale, rotation=rot, flip=flip, matrix=m)


def val_sample(img, rec, input_size):
    m = get_affine_transform(rec["center"], rec["scale"], input_size, 0.0)
    kp, vis = affine_keypoints(rec["keypoints"], rec["keypoints_visible"], m, None)
    return normalize_chw(warp_affine_u8(img, m, input_size)), kp.astype(np.float32), vis.astype(np.float32), dict(matrix=m)
