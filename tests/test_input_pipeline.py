"""f2 / f3: the input pipeline.  CPU part: the oracle's restatement of the reference's transforms (oracle/warp.py) against the
expressions the reference itself uses where they are available here (torch normalisation), its own invariants, and the product's host-side
transforms (same decisions, same RNG order).  GPU part (`-m gpu`): the crop kernel bit-exact against the oracle's integer warp, the fp32
normalisation bit-exact, the bf16 NHWC-8 output, the model consuming it, and PoseInference end to end.
cv2 is not in the image and the reference keeps no warped-image fixture: parity of the warp is UNPINNED vs cv2, pinned vs oracle/warp.py."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(rng, H, W, K=17):
    x1, y1 = rng.uniform(0, W * 0.4), rng.uniform(0, H * 0.4)
    x2, y2 = x1 + rng.uniform(W * 0.3, W * 0.55), y1 + rng.uniform(H * 0.3, H * 0.55)
    kp = np.stack([rng.uniform(x1, x2, K), rng.uniform(y1, y2, K)], 1).astype(np.float32)
    vis = rng.choice([0.0, 1.0, 2.0], K, p=[0.2, 0.3, 0.5]).astype(np.float32)
    return {"center": np.array([(x1 + x2) / 2, (y1 + y2) / 2], np.float32), "scale": np.array([x2 - x1, y2 - y1], np.float32) * 1.25,
            "keypoints": kp, "keypoints_visible": vis}


PAIRS = [(1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12), (13, 14), (15, 16)]


def test_oracle_normalisation_is_the_reference_expression():
    """coco_dataset.py:156-163 verbatim in torch vs oracle/warp.normalize_chw: bit for bit."""
    from oracle import warp as ow
    img = np.random.default_rng(0).integers(0, 256, (37, 29, 3), dtype=np.uint8)
    t = torch.from_numpy(img.transpose(2, 0, 1)).float() / 255.0
    mean, std = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    assert np.array_equal(((t - mean) / std).numpy().view(np.uint32), ow.normalize_chw(img).view(np.uint32))


def test_oracle_warp_invariants():
    from oracle import warp as ow
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (50, 40, 3), dtype=np.uint8)
    assert np.array_equal(ow.warp_affine_u8(img, np.array([[1., 0, 0], [0, 1., 0]]), (40, 50)), img)          # identity
    ref = np.zeros_like(img)
    ref[0:48, 3:40] = img[2:50, 0:37]
    assert np.array_equal(ow.warp_affine_u8(img, np.array([[1., 0, 3], [0, 1., -2]]), (40, 50)), ref)         # integer shift, zero border
    m = ow.get_affine_transform(np.array([20., 25.]), np.array([30., 40.]), (192, 256), 30.0)
    assert np.allclose(m[0, 0], m[1, 1]) and np.allclose(m[0, 1], -m[1, 0]) and np.allclose(np.hypot(m[0, 0], m[0, 1]), 192 / 30)   # similarity
    assert np.allclose(m @ np.array([20., 25., 1.]), [96., 128.])                                             # bbox centre -> crop centre
    assert np.array_equal(ow.warp_affine_u8(img[:, ::-1].copy(), m, (192, 256)), ow.warp_affine_u8(img, m, (192, 256), flip=True))
    half = ow.warp_affine_u8(img, np.array([[0.5, 0, 0], [0, 0.5, 0]]), (20, 25))                             # exact 2x decimation taps
    assert np.array_equal(half, img[0:50:2, 0:40:2])


def test_host_transforms_follow_the_oracle_decisions():
    """Product transforms (datasets/transforms.py) vs oracle/warp.train_sample on the same seeded numpy RNG: flips, half-body boxes,
    scale / rotation draws, matrices, transformed keypoints and visibility."""
    from infantposeestimation_gaussianbias_amd.datasets import transforms as T
    from oracle import warp as ow
    rng = np.random.default_rng(2)
    n_flip = n_rot = 0
    for i in range(60):
        H, W = int(rng.integers(60, 200)), int(rng.integers(60, 200))
        rec = _record(rng, H, W)
        r1, r2 = np.random.RandomState(100 + i), np.random.RandomState(100 + i)
        img = np.zeros((H, W, 3), np.uint8)
        _, kp_o, vis_o, info = ow.train_sample(img, rec, (48, 64), r1, flip_pairs=PAIRS)
        tf = T.get_train_transforms((48, 64), rng=r2)
        d = tf(dict(copy.deepcopy(rec), img_width=W, flip_pairs=PAIRS, flip=False))
        assert bool(d["flip"]) == bool(info["flip"]) and np.allclose(d["matrix"], info["matrix"], rtol=0, atol=1e-12)
        assert np.array_equal(d["keypoints"].astype(np.float32), kp_o) and np.array_equal(d["keypoints_visible"], vis_o)
        assert float(d.get("rotation", 0)) == float(info["rotation"])
        n_flip += int(info["flip"])
        n_rot += int(info["rotation"] != 0)
    assert 10 < n_flip < 50 and 15 < n_rot < 55
    rec = _record(rng, 100, 80)
    _, kp_o, vis_o, info = ow.val_sample(np.zeros((100, 80, 3), np.uint8), rec, (48, 64))
    d = T.get_val_transforms((48, 64))(dict(copy.deepcopy(rec), img_width=80))
    assert np.allclose(d["matrix"], info["matrix"], atol=1e-12) and np.array_equal(d["keypoints"].astype(np.float32), kp_o)
    assert np.allclose(T.invert_affine(d["matrix"]), ow.invert_affine(info["matrix"]), atol=0)


@pytest.mark.gpu
def test_device_cropper_bit_exact_vs_oracle():
    """pk_affine_crop_normalize on a ragged batch (different image sizes, flips, rotations, crops reaching outside the image, a BGR
    source): the fp32 NCHW output equals oracle warp + reference normalisation bit for bit; the NHWC-8 output is its bf16 rounding."""
    from infantposeestimation_gaussianbias_amd.datasets import transforms as T
    from oracle import warp as ow
    rng = np.random.default_rng(3)
    imgs, mats, flips, want = [], [], [], []
    for i in range(9):
        H, W = int(rng.integers(40, 300)), int(rng.integers(40, 300))
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        rec = _record(rng, H, W)
        if i == 0:
            rec["center"], rec["scale"] = np.array([2., 3.], np.float32), np.array([W * 1.5, H * 1.5], np.float32)      # mostly border
        x, kp, vis, info = ow.train_sample(img, rec, (48, 64), np.random.RandomState(i), flip_pairs=PAIRS)
        imgs.append(img)
        mats.append(info["matrix"])
        flips.append(info["flip"])
        want.append(x)
    out32, out16 = T.DeviceCropper((48, 64), "cuda")(imgs, mats, flips)
    torch.cuda.synchronize()
    got = out32.cpu().numpy()
    for i in range(len(imgs)):
        assert np.array_equal(got[i].view(np.uint32), want[i].view(np.uint32)), f"sample {i}: crop differs from the oracle"
    ref16 = torch.from_numpy(np.stack(want)).permute(0, 2, 3, 1).to(torch.bfloat16)
    assert torch.equal(out16[..., :3].cpu(), ref16) and float(out16[..., 3:].abs().max()) == 0.0
    # BGR source (inference.py:83): the same crop of the channel-swapped image
    o_bgr, _ = T.DeviceCropper((48, 64), "cuda", nhwc8=False)([im[:, :, ::-1].copy() for im in imgs], mats, flips, bgr=True)
    assert torch.equal(o_bgr, out32)
    with pytest.raises(Exception):
        T.DeviceCropper((48, 64), "cuda")([imgs[0].astype(np.float32)], mats[:1])


@pytest.mark.gpu
def test_model_takes_the_cropper_layout_and_pose_inference_runs():
    """The bf16 NHWC-8 batch goes straight into the stem (bitwise the same heatmaps as the fp32 NCHW batch through the conversion kernel);
    PoseInference.predict / predict_batch: preprocess + flip-test inference + image-space keypoints, batched == one at a time."""
    sys.path.insert(0, ROOT)
    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.datasets import transforms as T
    from infantposeestimation_gaussianbias_amd.models import build_model
    from oracle import warp as ow
    rng = np.random.default_rng(4)
    cfg = get_config("hrformer_small")
    torch.manual_seed(0)
    model = build_model(cfg).cuda().eval()
    imgs = [rng.integers(0, 256, (int(rng.integers(200, 400)), int(rng.integers(150, 300)), 3), dtype=np.uint8) for _ in range(3)]
    mats = [T.get_affine_matrix(np.array([im.shape[1] / 2, im.shape[0] / 2]), np.array([im.shape[1], im.shape[0]]) * 1.25, cfg.data.input_size) for im in imgs]
    x32, x16 = T.DeviceCropper(cfg.data.input_size, "cuda")(imgs, mats)
    with torch.no_grad():
        a, b = model(x32)["heatmaps"], model(x16)["heatmaps"]
    assert torch.equal(a, b)
    import inference as inf
    pi = inf.PoseInference(checkpoint=None, device="cuda", flip_test=True, config="hrformer_small")
    pi.model.load_state_dict(model.state_dict())
    bgr = [im[:, :, ::-1].copy() for im in imgs]
    x, c, s = pi.preprocess(bgr[0])
    want = ow.normalize_chw(ow.warp_affine_u8(imgs[0], ow.get_affine_transform(c, s, cfg.data.input_size), cfg.data.input_size))
    assert np.array_equal(x[0].cpu().numpy().view(np.uint32), want.view(np.uint32))
    batch = pi.predict_batch(bgr)
    single = [pi.predict(im) for im in bgr]
    for (k1, s1), (k2, s2) in zip(batch, single):
        assert k1.shape == (17, 2) and np.all(np.isfinite(k1)) and np.allclose(k1, k2, atol=2e-2) and np.allclose(s1, s2, atol=1e-3)
    kp_hm = np.array([[24.0, 32.0]] * 17, np.float32)                       # heatmap centre -> bbox centre (inference.py:158-170)
    out, _ = pi.postprocess(kp_hm.copy(), None, c, s)
    assert np.allclose(out, np.tile(c, (17, 1)), atol=1e-3)


def _record_loader(n_batches, B, cfg, seed=11):
    """Host-side record lists the way the COCO DataLoader hands them to DeviceBatcher (decoded uint8 image + transformed record)."""
    from infantposeestimation_gaussianbias_amd.datasets import transforms as T
    rng = np.random.default_rng(seed)
    tf = T.get_val_transforms(cfg.data.input_size)
    batches = []
    for _ in range(n_batches):
        recs = []
        for i in range(B):
            H, W = int(rng.integers(220, 420)), int(rng.integers(180, 340))
            r = _record(rng, H, W)
            r.update(img=rng.integers(0, 256, (H, W, 3), dtype=np.uint8), img_width=W, flip_pairs=PAIRS, flip=False, image_id=i, ann_id=i,
                     bbox=np.array([0, 0, W, H], np.float32), area=float(H * W))
            recs.append(tf(r))
        batches.append(recs)
    return batches


@pytest.mark.gpu
def test_device_batcher_prefetch_equals_the_synchronous_path():
    """Batch n + 1 prepared on a side stream with reused pinned staging buffers == the same work on the consumer's stream, bit for bit,
    also when the consumer overwrites nothing but is slow / fast to come back (buffer reuse is guarded by the copies' events)."""
    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.datasets.coco_dataset import DeviceBatcher
    cfg = get_config("hrformer_small")
    cfg.data.input_size, cfg.data.heatmap_size = (96, 128), (24, 32)
    loader = _record_loader(5, 6, cfg)
    want = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()} for b in DeviceBatcher(loader, cfg, prefetch=False)]
    got = []
    for i, b in enumerate(DeviceBatcher(loader, cfg, prefetch=True)):
        if i % 2:
            torch.cuda.synchronize()
        got.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()})
    torch.cuda.synchronize()
    assert len(got) == len(want) == 5
    for a, b in zip(got, want):
        for k in ("img", "img_nhwc8", "target", "target_weight", "keypoints", "keypoints_visible"):
            assert torch.equal(a[k], b[k]), k
        assert torch.equal(a["meta"]["center"], b["meta"]["center"])


@pytest.mark.gpu
def test_trainer_over_prefetching_batcher_sustains_the_resident_batch_rate():
    """a Trainer loop fed by DeviceBatcher (uint8 images on the host -> pinned staging -> crop / normalise / targets on a side
    stream) runs at >= 95 % of the img/s of the same loop on device-resident batches (BASELINE cfg 2 shape, hipGraph replay)."""
    import time
    sys.path.insert(0, ROOT)
    from infantposeestimation_gaussianbias_amd import engine
    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.datasets.coco_dataset import DeviceBatcher
    from infantposeestimation_gaussianbias_amd.models import build_model
    cfg = get_config("hrformer_small")
    B, steps = 64, 24
    torch.manual_seed(0)
    model = build_model(cfg).cuda()
    tr = engine.Trainer(model, cfg, iters_per_epoch=100, use_graph=True, graph_warmup=2, graph_streams=True)
    loader = _record_loader(4, B, cfg)
    batcher = DeviceBatcher(loader * (steps // 4 + 2), cfg, prefetch=True, nchw=False)
    it = iter(batcher)
    resident = [next(it) for _ in range(2)]
    for i in range(6):                              # eager warm-up + capture + first replays
        tr.step(resident[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(resident[i % 2])
    torch.cuda.synchronize()
    t_res = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    n = 0
    for batch in it:
        tr.step(batch)
        n += 1
        if n == steps:
            break
    torch.cuda.synchronize()
    t_fed = (time.perf_counter() - t0) / n
    print(f"resident {t_res * 1e3:.2f} ms/step, fed through the prefetching batcher {t_fed * 1e3:.2f} ms/step ({B / t_fed:.0f} img/s)")
    assert n == steps and t_res / t_fed >= 0.95, (t_res, t_fed)
