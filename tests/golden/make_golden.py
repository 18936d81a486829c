#!/usr/bin/env python3
"""Capture golden vectors from the reference's own PyTorch-CPU modules.

Runs ONLY in the build container (the reference tree cannot travel): it puts the reference
root (env POSE_REFERENCE_ROOT, default /root/reference) on sys.path, runs the hot-path
functions of SURVEY.md §8(a) on seeded inputs and writes small .npz/.json fixtures next to
this file.  Weights are never stored: they are regenerated from state_dict keys by
tests/golden/recipe.py on both sides.

Packages the reference imports but this image lacks (cv2, pycocotools, easydict,
tensorboard, torchvision) are replaced by MagicMock so the *modules* import; none of the
functions captured here touch them.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz, *.json
"""
import hashlib
import json
import os
import sys
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("POSE_REFERENCE_ROOT", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
sys.path.insert(0, REF)
for name in ("cv2", "pycocotools", "pycocotools.coco", "pycocotools.cocoeval", "easydict",
             "torch.utils.tensorboard", "torchvision", "torchvision.transforms", "matplotlib",
             "matplotlib.pyplot"):
    sys.modules.setdefault(name, mock.MagicMock())

from recipe import spec_of, synth_input, synth_state_dict  # noqa: E402

torch.set_num_threads(8)
torch.manual_seed(0)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def N(t):
    return t.detach().cpu().numpy()


def load_recipe(module, salt=0):
    spec = spec_of(module.state_dict())
    sd = {k: T(v) for k, v in synth_state_dict(spec, salt).items()}
    module.load_state_dict(sd, strict=True)
    return spec


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"  {name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays")


# ----------------------------------------------------------------------------- T1 / T2
def edge_keypoints(win, hin, stride):
    """Hand-picked input-pixel coordinates around every clipping rule of the generator."""
    xs = [0.0, 1.0, 22.0, 3.9, 4.0, win - 1.0, float(win), win + 3.0 * stride, win + 23.9,
          win + 24.0, win + 28.0, -1.0, -3.9, -23.9, -24.0, -27.9, -28.0, -30.0, -100.0, win / 2.0,
          win / 2.0 + 0.5 * stride, win / 2.0 + 0.49, 2.0 * stride + 1e-3, 6.0 * stride, 6.5 * stride]
    ys = [0.0, 1.0, 22.0, hin - 1.0, float(hin), hin + 23.9, hin + 24.0, -1.0, -23.9, -24.0, -28.0,
          -30.0, hin / 2.0, hin / 3.0, 5.5 * stride, 6.0 * stride, 4.5 * stride, 1.5 * stride]
    pts = [(x, ys[i % len(ys)]) for i, x in enumerate(xs)] + [(xs[(3 * i + 1) % len(xs)], y) for i, y in enumerate(ys)]
    return np.asarray(pts, dtype=np.float32)


def cap_t1():
    from datasets.coco_dataset import COCOPoseDataset
    out = {}
    cfgs = [(192, 256, 48, 64, 2.0, 17), (96, 128, 24, 32, 2.0, 17), (288, 384, 72, 96, 2.0, 17),
            (256, 256, 64, 64, 1.5, 13), (256, 256, 128, 128, 1.5, 13), (192, 256, 48, 64, 1.0, 17),
            (192, 256, 48, 64, 3.0, 17)]
    rng = np.random.default_rng(11)
    for ci, (win, hin, wh, hh, sigma, K) in enumerate(cfgs):
        ds = COCOPoseDataset.__new__(COCOPoseDataset)
        ds.input_size = np.array((win, hin))
        ds.heatmap_size = np.array((wh, hh))
        ds.sigma = sigma
        ds.num_keypoints = K
        edges = edge_keypoints(win, hin, win / wh)
        n_edge = (len(edges) + K - 1) // K
        nsamp = n_edge + (6 if wh <= 72 else 2)
        kps = np.zeros((nsamp, K, 2), np.float32)
        vis = np.zeros((nsamp, K), np.float32)
        flat = np.concatenate([edges, edges[: n_edge * K - len(edges)]], 0)
        kps[:n_edge] = flat.reshape(n_edge, K, 2)
        vis[:n_edge] = rng.choice([1.0, 2.0], size=(n_edge, K))
        kps[n_edge:, :, 0] = rng.uniform(-40, win + 40, size=(nsamp - n_edge, K))
        kps[n_edge:, :, 1] = rng.uniform(-40, hin + 40, size=(nsamp - n_edge, K))
        vis[n_edge:] = rng.choice([0.0, 1.0, 2.0], p=[0.15, 0.25, 0.6], size=(nsamp - n_edge, K))
        tg = np.zeros((nsamp, K, hh, wh), np.float32)
        tw = np.zeros((nsamp, K, 1), np.float32)
        for i in range(nsamp):
            tg[i], tw[i] = ds._generate_target(kps[i], vis[i])
        out[f"c{ci}_cfg"] = np.array([win, hin, wh, hh, sigma, K], np.float64)
        out[f"c{ci}_kp"] = kps
        out[f"c{ci}_vis"] = vis
        out[f"c{ci}_target"] = tg
        out[f"c{ci}_weight"] = tw
        out[f"c{ci}_sha256"] = np.frombuffer(hashlib.sha256(tg.tobytes()).digest(), np.uint8)
    out["n_cfg"] = np.array(len(cfgs))
    save("t1_target.npz", **out)


def cap_t2():
    sys.path.insert(0, os.path.join(REF, "data"))
    import pose_transforms as pt
    out = {}
    rng = np.random.default_rng(12)
    cfgs = [((256, 256), (64, 64), 2.0, 13), ((256, 192), (64, 48), 2.0, 17), ((256, 256), (128, 128), 1.5, 13)]
    for ci, (insz, hmsz, sigma, K) in enumerate(cfgs):
        gen = pt.GenerateTarget({"input_size": insz, "heatmap_size": hmsz, "sigma": sigma})
        n = 3
        kp = np.zeros((n, K, 2), np.float32)
        kp[..., 0] = rng.uniform(-10, insz[1] + 10, (n, K))
        kp[..., 1] = rng.uniform(-10, insz[0] + 10, (n, K))
        kp[0, 0] = (0.0, 0.0)
        kp[0, 1] = (insz[1] - 0.01, insz[0] - 0.01)
        kp[0, 2] = (float(insz[1]), 5.0)
        vis = rng.choice([0.0, 1.0, 2.0], size=(n, K)).astype(np.float32)
        hms = np.zeros((n, K, hmsz[0], hmsz[1]), np.float32)
        ws = np.zeros((n, K), np.float32)
        for i in range(n):
            r = gen({"keypoints": kp[i].copy(), "keypoints_visible": vis[i].copy()})
            hms[i], ws[i] = r["heatmaps"], r["keypoint_weights"]
        out[f"c{ci}_cfg"] = np.array([insz[0], insz[1], hmsz[0], hmsz[1], sigma, K], np.float64)
        out[f"c{ci}_kp"], out[f"c{ci}_vis"], out[f"c{ci}_heatmaps"], out[f"c{ci}_weights"] = kp, vis, hms, ws
    out["n_cfg"] = np.array(len(cfgs))
    save("t2_dense_target.npz", **out)


# ----------------------------------------------------------------------------- A1..A5
def grads_of(module, y, x, gy):
    module.zero_grad(set_to_none=True)
    (gx,) = torch.autograd.grad(y, x, gy, retain_graph=True)
    y.backward(gy)
    return N(gx), {k: N(p.grad) for k, p in module.named_parameters() if p.grad is not None}


def cap_attn():
    from models import hrformer as hf
    out, meta = {}, {}
    for tag, (C, h, H, W, B) in {"a": (32, 1, 9, 10, 2), "b": (64, 2, 8, 6, 2), "c": (78, 2, 12, 9, 1),
                                 "d": (128, 4, 7, 7, 1)}.items():
        # window attention alone on (B_w, 49, C)
        wa = hf.WindowAttention(C, 7, h)
        meta[f"wa_{tag}"] = {"C": C, "heads": h, "spec": load_recipe(wa, salt=1)}
        xw = T(synth_input(f"wa_{tag}", (3, 49, C))).requires_grad_(True)
        yw = wa(xw)
        gy = T(synth_input(f"wa_{tag}_gy", yw.shape))
        gx, pg = grads_of(wa, yw, xw, gy)
        out.update({f"wa_{tag}_x": N(xw), f"wa_{tag}_y": N(yw), f"wa_{tag}_gy": N(gy), f"wa_{tag}_gx": gx})
        out.update({f"wa_{tag}_g.{k}": v for k, v in pg.items()})
        # whole block on NCHW (padding path), drop_path = 0
        blk = hf.HRFormerBlock(C, h, window_size=7, mlp_ratio=4.0, drop_path=0.0)
        meta[f"blk_{tag}"] = {"C": C, "heads": h, "H": H, "W": W, "spec": load_recipe(blk, salt=2)}
        x = T(synth_input(f"blk_{tag}", (B, C, H, W))).requires_grad_(True)
        y = blk(x)
        gy = T(synth_input(f"blk_{tag}_gy", y.shape))
        gx, pg = grads_of(blk, y, x, gy)
        out.update({f"blk_{tag}_x": N(x), f"blk_{tag}_y": N(y), f"blk_{tag}_gy": N(gy), f"blk_{tag}_gx": gx})
        out.update({f"blk_{tag}_g.{k}": v for k, v in pg.items()})
    # window partition / reverse pure movement
    x = T(synth_input("wp", (2, 9, 10, 5)))
    wins, (Hp, Wp) = hf.window_partition(x, 7)
    out["wp_x"], out["wp_windows"], out["wp_pad"] = N(x), N(wins), np.array([Hp, Wp])
    out["wp_back"] = N(hf.window_reverse(wins, 7, 9, 10, Hp, Wp))
    # drop_path arithmetic with a fixed uniform draw
    out["dp_keep_formula"] = np.array([0.9])
    save("attn_blocks.npz", **out)
    return meta


def cap_modules():
    from models import hrformer as hf
    from models import hrnet as hn
    out, meta = {}, {}

    def run(tag, mod, xs, train, salt):
        meta[tag] = {"spec": load_recipe(mod, salt=salt), "train": train}
        mod.train(train)
        xs = [T(x).requires_grad_(True) for x in xs]
        ys = mod([x for x in xs]) if isinstance(xs, list) and len(xs) > 1 or tag.startswith(("hrm", "fm")) else mod(xs[0])
        ys = ys if isinstance(ys, (list, tuple)) else [ys]
        gys = [T(synth_input(f"{tag}_gy{i}", y.shape)) for i, y in enumerate(ys)]
        mod.zero_grad(set_to_none=True)
        tot = sum((y * g).sum() for y, g in zip(ys, gys))
        gxs = torch.autograd.grad(tot, xs, retain_graph=True)
        tot.backward()
        for i, x in enumerate(xs):
            out[f"{tag}_x{i}"], out[f"{tag}_gx{i}"] = N(x), N(gxs[i])
        for i, (y, g) in enumerate(zip(ys, gys)):
            out[f"{tag}_y{i}"], out[f"{tag}_gy{i}"] = N(y), N(g)
        for k, p in mod.named_parameters():
            if p.grad is not None:
                out[f"{tag}_g.{k}"] = N(p.grad)
        for k, b in mod.named_buffers():
            if "running" in k or "num_batches" in k:
                out[f"{tag}_buf.{k}"] = N(b)

    for train in (True, False):
        s = "tr" if train else "ev"
        run(f"basic_{s}", hn.BasicBlock(8, 8), [synth_input("basic", (2, 8, 6, 5))], train, 3)
        ds = torch.nn.Sequential(torch.nn.Conv2d(8, 16, 1, bias=False), torch.nn.BatchNorm2d(16))
        run(f"bneck_ds_{s}", hn.Bottleneck(8, 4, downsample=ds), [synth_input("bneck", (2, 8, 6, 5))], train, 4)
        run(f"bneck_{s}", hf.Bottleneck(16, 4), [synth_input("bneck2", (2, 16, 6, 5))], train, 5)
        run(f"hrm2_{s}", hn.HighResolutionModule(2, hn.BasicBlock, [1, 1], [4, 8]),
            [synth_input("hrm2_0", (2, 4, 8, 6)), synth_input("hrm2_1", (2, 8, 4, 3))], train, 6)
        run(f"hrm3_{s}", hn.HighResolutionModule(3, hn.BasicBlock, [1, 1, 1], [4, 8, 16]),
            [synth_input("hrm3_0", (2, 4, 8, 12)), synth_input("hrm3_1", (2, 8, 4, 6)),
             synth_input("hrm3_2", (2, 16, 2, 3))], train, 7)
        run(f"hrm4_{s}", hn.HighResolutionModule(4, hn.BasicBlock, [1, 1, 1, 1], [4, 8, 8, 8]),
            [synth_input("hrm4_0", (1, 4, 16, 8)), synth_input("hrm4_1", (1, 8, 8, 4)),
             synth_input("hrm4_2", (1, 8, 4, 2)), synth_input("hrm4_3", (1, 8, 2, 1))], train, 8)
        run(f"fm2_{s}", hf.HRFormerModule(2, "HRFORMERBLOCK", [1, 1], [16, 32], [1, 2], [4, 4], [7, 7], 0.0),
            [synth_input("fm2_0", (2, 16, 8, 6)), synth_input("fm2_1", (2, 32, 4, 3))], train, 9)
    # odd-size bilinear (9x7 <- 5x4, 3x2), as F.interpolate inside the exchange unit
    import torch.nn.functional as F
    src = T(synth_input("bil", (1, 3, 5, 4))).requires_grad_(True)
    up = F.interpolate(src, size=[9, 7], mode="bilinear", align_corners=False)
    g = T(synth_input("bil_g", up.shape))
    (gs,) = torch.autograd.grad(up, src, g)
    out["bil_src"], out["bil_up"], out["bil_g"], out["bil_gsrc"] = N(src), N(up), N(g), N(gs)
    save("modules.npz", **out)
    return meta


# ----------------------------------------------------------------------------- H1 / L1-L4
def synth_loss_inputs(tag, B, K, H, W, win, hin, peaky):
    rng = np.random.default_rng(abs(hash(tag)) % (2 ** 31) if False else int.from_bytes(hashlib.md5(tag.encode()).digest()[:4], "little"))
    hm = rng.standard_normal((B, K, H, W)).astype(np.float32) * 0.5
    gt = np.stack([rng.uniform(0, win, (B, K)), rng.uniform(0, hin, (B, K))], -1).astype(np.float32)
    tgt = np.zeros((B, K, H, W), np.float32)
    ys, xs = np.mgrid[0:H, 0:W]
    for b in range(B):
        for k in range(K):
            cx, cy = gt[b, k, 0] * W / win, gt[b, k, 1] * H / hin
            g = np.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / 8.0).astype(np.float32)
            tgt[b, k] = g
            if peaky:
                hm[b, k] += 6.0 * np.exp(-((xs - cx - 0.7) ** 2 + (ys - cy + 0.4) ** 2) / 6.0).astype(np.float32)
    off = rng.standard_normal((B, K, 2, H, W)).astype(np.float32)
    var = np.log1p(np.exp(rng.standard_normal((B, K, H, W)))).astype(np.float32) + 1.0
    w = rng.choice([0.0, 1.0, 2.0], p=[0.2, 0.3, 0.5], size=(B, K, 1)).astype(np.float32)
    return hm, off, var, tgt, w, gt


def cap_head_loss():
    from models import fusion_head as fh
    from models import pose_estimator as pe
    from models import losses as ls
    out, meta = {}, {}
    # H1 head forward/backward, train + eval BN
    for K in (17, 13):
        for train in (True, False):
            tag = f"head_k{K}_{'tr' if train else 'ev'}"
            head = fh.HeatmapRegressionHead(8, num_keypoints=K, hidden_dim=16)
            meta[tag] = {"spec": load_recipe(head, salt=20 + K), "K": K, "in": 8, "hidden": 16}
            head.train(train)
            x = T(synth_input(tag, (2, 8, 8, 6))).requires_grad_(True)
            o = head(x)
            ghm, goff, gvar = (T(synth_input(tag + n, o[k].shape)) for n, k in
                               (("_ghm", "heatmaps"), ("_goff", "offsets"), ("_gvar", "variances")))
            head.zero_grad(set_to_none=True)
            tot = (o["heatmaps"] * ghm).sum() + (o["offsets"] * goff).sum() + (o["variances"] * gvar).sum()
            (gx,) = torch.autograd.grad(tot, x, retain_graph=True)
            tot.backward()
            out.update({f"{tag}_x": N(x), f"{tag}_hm": N(o["heatmaps"]), f"{tag}_off": N(o["offsets"]),
                        f"{tag}_var": N(o["variances"]), f"{tag}_fw": N(o["fusion_weight"]),
                        f"{tag}_ghm": N(ghm), f"{tag}_goff": N(goff), f"{tag}_gvar": N(gvar), f"{tag}_gx": N(gx)})
            for k, p in head.named_parameters():
                if p.grad is not None:
                    out[f"{tag}_g.{k}"] = N(p.grad)
    # H2 plain heatmap head
    hh = pe.HeatmapHead(8, 17)
    meta["hmhead"] = {"spec": load_recipe(hh, salt=30)}
    x = T(synth_input("hmhead", (2, 8, 5, 4)))
    out["hmhead_x"], out["hmhead_y"] = N(x), N(hh(x))

    # L1/L2 FusionPoseLoss: 7 components + grads
    loss = fh.FusionPoseLoss(1.0, 1.0, 0.5, 0.1, 0.05, 0.05, 2.0, True)
    for tag, (B, K, H, W, win, hin, peaky) in {
        "l_small": (2, 17, 16, 12, 48, 64, False), "l_peaky": (2, 17, 16, 12, 48, 64, True),
        "l_k13": (3, 13, 16, 16, 64, 64, True), "l_full": (2, 17, 64, 48, 192, 256, True),
        "l_k5": (2, 5, 8, 8, 32, 32, False),
    }.items():
        hm, off, var, tgt, w, gt = synth_loss_inputs(tag, B, K, H, W, win, hin, peaky)
        if tag == "l_k5":
            w[1] = 0.0  # a sample with every joint invisible
        thm, toff, tvar = T(hm).requires_grad_(True), T(off).requires_grad_(True), T(var).requires_grad_(True)
        res = loss({"heatmaps": thm, "offsets": toff, "variances": tvar, "fusion_weight": torch.tensor(0.6)},
                   T(tgt), T(w), T(gt), input_size=(win, hin), heatmap_size=(H, W))
        names = ["heatmap_loss", "offset_loss", "peak_loss", "variance_loss", "overlap_loss", "shape_loss", "total_loss"]
        comps = np.array([float(res[n]) for n in names], np.float64)
        g = torch.autograd.grad(res["total_loss"], [thm, toff, tvar])
        out.update({f"{tag}_hm": hm, f"{tag}_off": off, f"{tag}_var": var, f"{tag}_tgt": tgt, f"{tag}_w": w,
                    f"{tag}_gt": gt, f"{tag}_size": np.array([win, hin, H, W]), f"{tag}_losses": comps,
                    f"{tag}_ghm": N(g[0]), f"{tag}_goff": N(g[1]), f"{tag}_gvar": N(g[2])})
        sa = fh.SoftArgmax2D()
        c, s = sa(T(hm))
        out[f"{tag}_softargmax"], out[f"{tag}_scores"] = N(c), N(s)
        # L3 / L4 on the same maps
        tp = T(hm).requires_grad_(True)
        l3 = pe.KeypointMSELoss(True)(tp, T(tgt), T(w))
        (g3,) = torch.autograd.grad(l3, tp)
        out[f"{tag}_l3"], out[f"{tag}_l3_g"] = np.array(float(l3)), N(g3)
        out[f"{tag}_l3_now"] = np.array(float(pe.KeypointMSELoss(False)(T(hm), T(tgt), T(w))))
        php = torch.relu(T(hm)).requires_grad_(True)  # morphology loss wants non-negative maps
        l4 = {
            "fused_mse": ls.FusedPoseLoss(True, "mse")(T(hm), T(tgt), T(w)),
            "fused_sl1": ls.FusedPoseLoss(True, "smoothl1")(T(hm), T(tgt), T(w)),
            "morph": ls.MorphologyShapeLoss(1.2, 0.5)(php, T(tgt), T(w)),
            "joints": ls.JointsMSELoss(True)(T(hm), T(tgt), T(w)),
            "joints_now": ls.JointsMSELoss(False)(T(hm), T(tgt), T(w)),
            "offreg_sl1": ls.OffsetRegressionLoss("smoothl1")(T(gt) * 0.1, T(gt[:, ::-1].copy()) * 0.1, T(w)),
            "offreg_l1": ls.OffsetRegressionLoss("l1")(T(gt) * 0.1, T(gt[:, ::-1].copy()) * 0.1, T(w)),
            "offreg_mse": ls.OffsetRegressionLoss("mse")(T(gt) * 0.1, T(gt[:, ::-1].copy()) * 0.1, T(w)),
        }
        (gm,) = torch.autograd.grad(l4["morph"], php)
        out[f"{tag}_l4_morph_g"] = N(gm)
        mean, var_ = ls.MorphologyShapeLoss().compute_spatial_statistics(torch.relu(T(hm)))
        out[f"{tag}_l4_mean"], out[f"{tag}_l4_var"] = N(mean), N(var_)
        for k, v in l4.items():
            out[f"{tag}_l4_{k}"] = np.array(float(v))
        # CombinedLoss through a plain-attribute stand-in for the EasyDict config
        class _C:  # noqa: N801
            class LOSS:
                MORPH_LAMBDA, MORPH_WEIGHT, REG_WEIGHT = 1.2, 0.15, 0.6
        tot, d = ls.CombinedLoss(_C)({"heatmaps": torch.relu(T(hm)), "coords": T(gt) * 0.1, "refined_coords": T(gt) * 0.11},
                                     {"heatmaps": T(tgt), "coords": T(gt[:, ::-1].copy()) * 0.1, "weights": T(w)})
        out[f"{tag}_l4_combined"] = np.array([float(tot)] + [float(d[k]) for k in ("heatmap", "morph", "regression", "refined")])
    save("head_loss.npz", **out)
    return meta


# ----------------------------------------------------------------------------- D1..D4
def cap_decode():
    from models import fusion_head as fh
    from models import pose_estimator as pe
    import utils.postprocess as pp
    out = {}
    rng = np.random.default_rng(21)
    for tag, (B, K, H, W) in {"d_small": (2, 17, 16, 12), "d_full": (2, 17, 64, 48), "d_sq": (1, 13, 32, 32)}.items():
        hm = (rng.standard_normal((B, K, H, W)) * 0.3).astype(np.float32)
        ys, xs = np.mgrid[0:H, 0:W]
        for b in range(B):
            for k in range(K):
                mode = (b * K + k) % 7
                cx, cy = rng.uniform(0, W - 1), rng.uniform(0, H - 1)
                if mode == 0:
                    cx, cy = 0.0, rng.uniform(0, H - 1)      # left border peak
                elif mode == 1:
                    cx, cy = W - 1.0, H - 1.0                # corner peak
                elif mode == 2:
                    cx, cy = 1.0, 1.0                        # Taylor boundary (px == 1 is excluded)
                hm[b, k] += 4.0 * np.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / 5.0).astype(np.float32)
                if mode == 3:                                # exact tie: two equal maxima -> first index wins
                    hm[b, k, 2, 3] = hm[b, k, 5, 1] = 9.0
                if mode == 4:                                # plateau: symmetric neighbours -> sign(0) = 0
                    hm[b, k] = np.round(hm[b, k] * 2) / 2
        off = rng.standard_normal((B, K, 2, H, W)).astype(np.float32)
        out[f"{tag}_hm"], out[f"{tag}_off"] = hm, off
        head = fh.HeatmapRegressionHead(4, num_keypoints=K, hidden_dim=8)
        for alpha_p, fw_p in ((0.5, 0.5), (-0.3, 1.2)):
            with torch.no_grad():
                head.subpixel_refine.alpha.fill_(alpha_p)
                head.fusion_weight.fill_(fw_p)
            o = {"heatmaps": T(hm), "offsets": T(off), "fusion_weight": torch.sigmoid(head.fusion_weight)}
            with torch.no_grad():
                c1, s1 = head.decode(o, apply_offset=True)
                c0, _ = head.decode(o, apply_offset=False)
                lg = head.subpixel_refine.local_refine(T(hm), fh.SoftArgmax2D()(T(hm))[0])
            sfx = f"a{alpha_p}_f{fw_p}"
            out[f"{tag}_d1_{sfx}"], out[f"{tag}_d1_nooff_{sfx}"], out[f"{tag}_d1_scores"] = N(c1), N(c0), N(s1)
            out[f"{tag}_d1_local"] = N(lg)
        k2, s2 = pe.PoseEstimator.decode_heatmaps(T(hm), shift=True)
        k2n, _ = pe.PoseEstimator.decode_heatmaps(T(hm), shift=False)
        out[f"{tag}_d2"], out[f"{tag}_d2_noshift"], out[f"{tag}_d2_scores"] = N(k2), N(k2n), N(s2)
        out[f"{tag}_argmax"] = N(torch.max(T(hm).view(B, K, -1), 2)[1]).astype(np.int32)
        p, mv = pp.get_max_preds(T(hm))
        ps, _ = pp.get_max_preds_with_subpixel(T(hm))
        out[f"{tag}_d3_max"], out[f"{tag}_d3_maxvals"], out[f"{tag}_d3_taylor"] = N(p), N(mv), N(ps)
        reg = T(rng.uniform(0, 1, (B, K, 2)).astype(np.float32))
        cen, sc = T(rng.uniform(50, 200, (B, 2)).astype(np.float32)), T(rng.uniform(80, 160, (B, 2)).astype(np.float32))
        f0, _ = pp.fused_decode(T(hm))
        f1, _ = pp.fused_decode(T(hm), reg.clone(), cen, sc, alpha=0.4)
        f2, _ = pp.fused_decode(T(hm), (reg * 100).clone(), None, None, alpha=0.4)
        out[f"{tag}_d3_fused0"], out[f"{tag}_d3_fused1"], out[f"{tag}_d3_fused2"] = N(f0), N(f1), N(f2)
        out[f"{tag}_d3_reg"], out[f"{tag}_d3_center"], out[f"{tag}_d3_scale"] = N(reg), N(cen), N(sc)
        rc = pp.coordinate_refinement(T(hm), ps.clone())
        out[f"{tag}_d3_refined"] = N(rc)
        fp, m = pp.filter_low_confidence(ps, mv, 0.3)
        out[f"{tag}_d3_filtered"], out[f"{tag}_d3_mask"] = N(fp), N(m)
        out[f"{tag}_d3_transformed"] = N(pp.transform_preds(ps, cen, sc, [640, 480]))
        class _Cfg:  # noqa: N801
            class TEST:
                FUSION_ALPHA = 0.4
        r = pp.postprocess_predictions({"heatmaps": T(hm), "coords": reg.clone()}, {"center": cen, "scale": sc}, _Cfg)
        out[f"{tag}_d3_pipeline"] = N(r["preds"])
    # train.py / validate.py transform_preds (B x K python loop in the reference)
    c = rng.uniform(0, 192, (3, 17, 2)).astype(np.float32)
    cen, sc = rng.uniform(100, 300, (3, 2)).astype(np.float32), rng.uniform(100, 300, (3, 2)).astype(np.float32)
    outp = c.copy()
    for i in range(3):
        for k in range(17):
            t = c[i, k].copy()
            t[0] = c[i, k, 0] / 192 * sc[i, 0] + cen[i, 0] - sc[i, 0] / 2
            t[1] = c[i, k, 1] / 256 * sc[i, 1] + cen[i, 1] - sc[i, 1] / 2
            outp[i, k] = t
    out["tp_in"], out["tp_center"], out["tp_scale"], out["tp_out"] = c, cen, sc, outp
    save("decode.npz", **out)


# ----------------------------------------------------------------------------- model level
def cap_models():
    import models
    from models import hrnet as hn
    from models import pose_estimator as pe
    out, meta = {}, {}
    specs = {}
    for name, (bb, K, head) in {"hrformer_small_fusion": ("hrformer_small", 17, "fusion"),
                                "hrnet_w32_heatmap": ("hrnet_w32", 17, "heatmap"),
                                "hrformer_base_fusion_k13": ("hrformer_base", 13, "fusion"),
                                "hrnet_w48_heatmap": ("hrnet_w48", 17, "heatmap")}.items():
        m = models.PoseEstimator(bb, K, False, head, True)
        specs[name] = spec_of(m.state_dict())
        specs[name + "#params"] = [k for k, _ in m.named_parameters()]
        if name == "hrformer_small_fusion":
            small = m
        if name == "hrnet_w32_heatmap":
            w32 = m
    # HRFormer-small eval forward, B=1, 256x192
    load_recipe(small, salt=40)
    small.eval()
    x = T(synth_input("small_eval", (1, 3, 256, 192)))
    with torch.no_grad():
        o = small(x)
    out["small_eval_hm"], out["small_eval_off"] = N(o["heatmaps"]), N(o["offsets"])[:, :, :, ::4, ::4]
    out["small_eval_var"], out["small_eval_fw"] = N(o["variances"])[:, :, ::4, ::4], N(o["fusion_weight"])
    with torch.no_grad():
        kp, sc = small.inference(x, flip=True, flip_pairs=[(1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12), (13, 14), (15, 16)])
        kp0, sc0 = small.inference(x, flip=False)
    out["small_eval_flip_kp"], out["small_eval_flip_sc"], out["small_eval_kp"], out["small_eval_sc"] = N(kp), N(sc), N(kp0), N(sc0)
    # HRFormer-small train-mode step (DropPath off), B=2 at 128x96: loss dict + grad norms of every parameter
    load_recipe(small, salt=40)
    small.train()
    for mod in small.modules():
        if mod.__class__.__name__ == "DropPath":
            mod.drop_prob = 0.0
    B, K, H, W, win, hin = 2, 17, 32, 24, 96, 128
    hm, off, var, tgt, w, gt = synth_loss_inputs("small_train", B, K, H, W, win, hin, True)
    x = T(synth_input("small_train", (B, 3, hin, win)))
    o = small(x, T(tgt), T(w), T(gt), input_size=(win, hin))
    small.zero_grad(set_to_none=True)
    o["loss"].backward()
    names = ["heatmap_loss", "offset_loss", "peak_loss", "variance_loss", "overlap_loss", "shape_loss", "total_loss"]
    out["small_train_losses"] = np.array([float(o["losses"][n]) for n in names])
    out["small_train_tgt"], out["small_train_w"], out["small_train_gt"] = tgt, w, gt
    out["small_train_hm"] = N(o["heatmaps"])
    gn = {k: float(p.grad.norm()) if p.grad is not None else -1.0 for k, p in small.named_parameters()}
    meta["small_train_gradnorm"] = gn
    meta["small_train_nograd"] = [k for k, v in gn.items() if v < 0]
    for k in ("backbone.conv1.weight", "backbone.stage2.0.branches.0.0.attn.relative_position_bias_table",
              "backbone.stage3.1.branches.1.0.attn.qkv.weight", "backbone.stage4.1.fuse_layers.0.3.0.weight",
              "head.heatmap_branch.3.weight", "head.offset_branch.3.bias", "backbone.stage2.0.branches.0.1.mlp.fc2.bias",
              "backbone.stage4.0.branches.3.1.norm2.weight", "backbone.layer1.0.bn2.weight"):
        out["small_train_g." + k] = N(dict(small.named_parameters())[k].grad)
    # BN running stats after that one train step
    sd = small.state_dict()
    for k in ("backbone.bn1.running_mean", "backbone.bn1.running_var", "head.shared_layers.1.running_var",
              "backbone.stage3.0.fuse_layers.0.1.1.running_mean"):
        out["small_train_buf." + k] = N(sd[k])

    # cfg 1: HRNet(base 18) + HeatmapHead + KeypointMSELoss, 128x96, B=4, 3 AdamW steps
    class Cfg1(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = hn.HRNet(base_channels=18)
            self.head = pe.HeatmapHead(18, 17)
            self.loss_fn = pe.KeypointMSELoss(True)
    c1 = Cfg1()
    specs["hrnet_w18_heatmap"] = spec_of(c1.state_dict())
    specs["hrnet_w18_heatmap#params"] = [k for k, _ in c1.named_parameters()]
    load_recipe(c1, salt=41)
    c1.train()
    hm, off, var, tgt, w, gt = synth_loss_inputs("cfg1", 4, 17, 32, 24, 96, 128, True)
    x = T(synth_input("cfg1", (4, 3, 128, 96)))
    decay, no_decay = [], []
    for n_, p in c1.named_parameters():
        (no_decay if ("bias" in n_ or "bn" in n_ or "norm" in n_) else decay).append(p)
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.01}, {"params": no_decay, "weight_decay": 0.0}],
                            lr=5e-4, betas=(0.9, 0.999))
    traj = []
    for step in range(3):
        opt.zero_grad()
        y = c1.head(c1.backbone(x))
        l = c1.loss_fn(y, T(tgt), T(w))
        l.backward()
        opt.step()
        traj.append(float(l))
        if step == 0:
            out["cfg1_hm0"] = N(y)
    out["cfg1_losses"], out["cfg1_tgt"], out["cfg1_w"] = np.array(traj), tgt, w
    out["cfg1_final_head_w"] = N(c1.head.final_layer.weight)
    out["cfg1_final_conv1_w"] = N(c1.backbone.conv1.weight)

    # HRNet-W32 eval, 128x96, B=1
    load_recipe(w32, salt=42)
    w32.eval()
    x = T(synth_input("w32_eval", (1, 3, 128, 96)))
    with torch.no_grad():
        o = w32(x)
        kp, sc = w32.inference(x, flip=False)
    out["w32_eval_hm"], out["w32_eval_kp"], out["w32_eval_sc"] = N(o["heatmaps"]), N(kp), N(sc)
    save("model_level.npz", **out)
    with open(os.path.join(HERE, "state_keys.json"), "w") as f:
        json.dump(specs, f, separators=(",", ":"))
    return meta


def cap_video():
    """utils/postprocess.py::temporal_smoothing / nms_pose (video post-processing)."""
    from utils import postprocess as pp
    rng = np.random.default_rng(77)
    out = {}
    traj = (np.cumsum(rng.normal(0, 1.5, (23, 17, 2)), 0) + rng.uniform(0, 48, (1, 17, 2))).astype(np.float32)
    out["ts_in"] = traj
    for w in (3, 5, 7):
        out[f"ts_gauss_w{w}"] = N(pp.temporal_smoothing(T(traj), w, "gaussian"))
        out[f"ts_avg_w{w}"] = N(pp.temporal_smoothing(T(traj), w, "moving_average"))
    short = traj[:2]
    out["ts_short_in"], out["ts_short_w5"] = short, N(pp.temporal_smoothing(T(short), 5, "gaussian"))
    # nms: clustered joints (several within 5 px of each other), ties in confidence, isolated joints
    preds = rng.uniform(0, 40, (6, 17, 2)).astype(np.float32)
    preds[:, 3] = preds[:, 2] + rng.uniform(-2, 2, (6, 2)).astype(np.float32)
    preds[:, 9] = preds[:, 2] + rng.uniform(-3, 3, (6, 2)).astype(np.float32)
    preds[:, 12] = preds[:, 11] + 1.0
    preds[1, 5] = preds[1, 4]
    conf = rng.uniform(0.1, 1.0, (6, 17, 1)).astype(np.float32)
    conf[1, 5] = conf[1, 4]
    out["nms_preds"], out["nms_conf"] = preds, conf
    for thr in (5.0, 2.0, 12.0):
        kept, mask = pp.nms_pose(T(preds), T(conf), thr)
        out[f"nms_out_t{int(thr)}"], out[f"nms_keep_t{int(thr)}"] = N(kept), N(mask).astype(np.uint8)
    save("video_post.npz", **out)


def cap_base():
    """HRFormer-base + fusion head, K=13 (BASELINE cfg 5 at reduced resolution): C=(78,156,312,624), head_dim 39 -- the
    configuration whose channel counts are not multiples of 8 (exercises the padded-twin path of the build)."""
    import models
    out, meta = {}, {}
    pairs = [(1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12)]
    m = models.PoseEstimator("hrformer_base", 13, False, "fusion", True)
    load_recipe(m, salt=44)
    m.eval()
    x = T(synth_input("base_eval", (1, 3, 128, 96)))
    with torch.no_grad():
        o = m(x)
        kp, sc = m.inference(x, flip=True, flip_pairs=pairs)
        kp0, sc0 = m.inference(x, flip=False)
    out["base_eval_hm"], out["base_eval_fw"] = N(o["heatmaps"]), N(o["fusion_weight"])
    out["base_eval_off"], out["base_eval_var"] = N(o["offsets"])[:, :, :, ::4, ::4], N(o["variances"])[:, :, ::4, ::4]
    out["base_eval_flip_kp"], out["base_eval_flip_sc"], out["base_eval_kp"], out["base_eval_sc"] = N(kp), N(sc), N(kp0), N(sc0)
    # train-mode step (DropPath off), B=2 at 128x96
    load_recipe(m, salt=44)
    m.train()
    for mod in m.modules():
        if mod.__class__.__name__ == "DropPath":
            mod.drop_prob = 0.0
    B, K, H, W, win, hin = 2, 13, 32, 24, 96, 128
    hm, off, var, tgt, w, gt = synth_loss_inputs("base_train", B, K, H, W, win, hin, True)
    x = T(synth_input("base_train", (B, 3, hin, win)))
    o = m(x, T(tgt), T(w), T(gt), input_size=(win, hin))
    m.zero_grad(set_to_none=True)
    o["loss"].backward()
    names = ["heatmap_loss", "offset_loss", "peak_loss", "variance_loss", "overlap_loss", "shape_loss", "total_loss"]
    out["base_train_losses"] = np.array([float(o["losses"][n]) for n in names])
    out["base_train_tgt"], out["base_train_w"], out["base_train_gt"] = tgt, w, gt
    out["base_train_hm"] = N(o["heatmaps"])
    gn = {k: float(p.grad.norm()) if p.grad is not None else -1.0 for k, p in m.named_parameters()}
    meta["base_train_gradnorm"] = gn
    meta["base_train_nograd"] = [k for k, v in gn.items() if v < 0]
    sd = dict(m.named_parameters())
    for k in ("backbone.stage2.0.branches.0.0.attn.qkv.weight", "backbone.stage2.0.branches.0.0.attn.proj.weight",
              "backbone.stage3.1.branches.1.0.attn.relative_position_bias_table", "backbone.stage4.0.branches.3.1.norm2.weight",
              "backbone.stage2.0.branches.1.0.mlp.fc1.weight", "backbone.transition1.0.0.weight", "head.shared_layers.0.weight"):
        out["base_train_g." + k] = N(sd[k].grad)
    for k in ("backbone.bn1.running_var", "backbone.stage3.0.fuse_layers.0.1.1.running_mean"):
        out["base_train_buf." + k] = N(dict(m.named_buffers())[k])
    save("model_base.npz", **out)
    return meta


# ----------------------------------------------------------------------------- S1
def cap_schedule():
    import train as rt  # reference train.py (tensorboard mocked)
    import models
    from configs.config import get_config
    cfg = get_config()
    m = models.PoseEstimator("hrformer_small", 17, False, "fusion", True)
    opt = rt.build_optimizer(m, cfg)
    id2name = {id(p): k for k, p in m.named_parameters()}
    groups = [[id2name[id(p)] for p in g["params"]] for g in opt.param_groups]
    sched = rt.build_scheduler(opt, cfg, num_iters_per_epoch=100)
    its = [0, 1, 10, 250, 499, 500, 501, 16999, 17000, 17001, 19999, 20000, 20999]
    lam = sched.lr_lambdas[0]
    meta = {"decay_names": groups[0], "no_decay_names": groups[1], "wd": [g["weight_decay"] for g in opt.param_groups],
            "lr_iters": its, "lr_factor": [float(lam(i)) for i in its], "iters_per_epoch": 100,
            "cfg": {"lr": cfg.train.lr, "warmup_lr": cfg.train.warmup_lr, "warmup_epochs": cfg.train.warmup_epochs,
                    "milestones": cfg.train.lr_milestones, "gamma": cfg.train.lr_gamma, "betas": list(cfg.train.betas),
                    "weight_decay": cfg.train.weight_decay, "max_epochs": cfg.train.max_epochs}}
    import dataclasses
    meta["default_config"] = dataclasses.asdict(cfg)
    return meta


# ----------------------------------------------------------------------------- round 2 additions: f4, cfg 4 training, f1
def cap_extra():
    """optimizer-state interchange (train.py:55-128,339-368), an HRNet-W32 heatmap-head train step (BASELINE cfg 4) and the
    COCOEvaluator bookkeeping (utils/metrics.py:61-106,107-270)."""
    import train as rt
    import models
    from configs.config import get_config
    from models import hrformer as hf
    meta, out = {}, {}
    # ---- f4: the reference's build_optimizer / build_scheduler on a small two-branch HRFormer module, three steps
    cfg = get_config()
    mod = hf.HRFormerModule(2, "HRFORMERBLOCK", [1, 1], [16, 32], [1, 2], [4, 4], [7, 7], 0.0)
    spec = load_recipe(mod, salt=21)
    mod.train()
    opt = rt.build_optimizer(mod, cfg)
    sched = rt.build_scheduler(opt, cfg, num_iters_per_epoch=2)
    xs = [T(synth_input("opt_x0", (2, 16, 8, 6))), T(synth_input("opt_x1", (2, 32, 4, 3)))]
    gys = None
    names = [k for k, _ in mod.named_parameters()]
    for step in range(3):
        opt.zero_grad()
        ys = mod([x.clone() for x in xs])
        if gys is None:
            gys = [T(synth_input(f"opt_gy{i}", y.shape)) for i, y in enumerate(ys)]
        loss = sum((y * g).sum() for y, g in zip(ys, gys))
        loss.backward()
        if step == 2:
            for k, p in mod.named_parameters():
                out["optim_grad3." + k] = N(p.grad).copy()
        opt.step()
        sched.step()
        if step == 1:           # state after two steps = what a checkpoint written there holds (train.py:351-357)
            sd = opt.state_dict()
            for idx, st in sd["state"].items():
                out[f"optim_state.{idx}.step"] = np.asarray(float(st["step"]))
                # (copies: the optimiser keeps updating these tensors in place in the third step)
                out[f"optim_state.{idx}.exp_avg"], out[f"optim_state.{idx}.exp_avg_sq"] = N(st["exp_avg"]).copy(), N(st["exp_avg_sq"]).copy()
            meta["optim_param_groups"] = [{k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in g.items()} for g in sd["param_groups"]]
            meta["optim_sched"] = {k: v for k, v in sched.state_dict().items() if k != "lr_lambdas"}
            for k, p in mod.named_parameters():
                out["optim_w2." + k] = N(p).copy()
            for k, b in mod.named_buffers():
                if b.is_floating_point():
                    out["optim_b2." + k] = N(b).copy()
    for k, p in mod.named_parameters():
        out["optim_w3." + k] = N(p)
    meta["optim_spec"], meta["optim_names"] = spec, names
    meta["optim_lr_after3"] = [g["lr"] for g in opt.param_groups]
    # ---- cfg 4: HRNet-W32 + HeatmapHead + KeypointMSELoss train step at fixture size (B=2, 128x96)
    w32 = models.PoseEstimator("hrnet_w32", 17, False, "heatmap", True)
    load_recipe(w32, salt=42)
    w32.train()
    hm, off, var, tgt, w, gt = synth_loss_inputs("w32_train", 2, 17, 32, 24, 96, 128, True)
    x = T(synth_input("w32_train", (2, 3, 128, 96)))
    o = w32(x, T(tgt), T(w))
    w32.zero_grad(set_to_none=True)
    o["loss"].backward()
    out["w32_train_loss"], out["w32_train_tgt"], out["w32_train_w"], out["w32_train_hm"] = np.asarray(float(o["loss"])), tgt, w, N(o["heatmaps"])
    gn = {k: float(p.grad.norm()) if p.grad is not None else -1.0 for k, p in w32.named_parameters()}
    meta["w32_train_gradnorm"] = gn
    meta["w32_train_nograd"] = [k for k, v in gn.items() if v < 0]
    for k in ("backbone.conv1.weight", "head.final_layer.weight", "head.final_layer.bias", "backbone.stage3.1.branches.1.2.conv2.weight",
              "backbone.stage4.2.fuse_layers.0.3.0.weight"):
        out["w32_train_g." + k] = N(dict(w32.named_parameters())[k].grad)
    # ---- f1: COCOEvaluator.update / compute_oks / manual evaluation on synthetic predictions
    from utils.metrics import COCOEvaluator
    rng = np.random.default_rng(5)
    B, K = 6, 17
    pk = (rng.uniform(0, 400, (B, K, 2))).astype(np.float32)
    ps = rng.uniform(-0.2, 1.0, (B, K)).astype(np.float32)
    ps[2] = -0.5                                   # an instance without any positive score
    ps[3, :5] = 0.0
    ids, anns = [11, 11, 12, 13, 13, 13], [1, 2, 3, 4, 5, 6]
    centers, scales = rng.uniform(50, 300, (B, 2)).astype(np.float32), rng.uniform(80, 200, (B, 2)).astype(np.float32)
    areas = rng.uniform(2000, 30000, B).astype(np.float32)
    bboxes = rng.uniform(0, 300, (B, 4)).astype(np.float32)
    ev = COCOEvaluator(ann_file=None, num_keypoints=K)
    ev.update(pk, ps, ids, anns, centers, scales, areas, bboxes)
    gts = []
    for i in range(B):
        kp3 = np.zeros((K, 3), np.float32)
        kp3[:, :2] = pk[i] + rng.normal(0, 6.0 if i % 2 else 1.5, (K, 2))
        kp3[:, 2] = rng.choice([0, 1, 2], K, p=[0.2, 0.3, 0.5])
        gts.append({"image_id": ids[i], "keypoints": kp3.flatten().tolist(), "area": float(areas[i])})
    metrics = ev.evaluate(gt_annotations=gts)
    out.update(eval_pk=pk, eval_ps=ps, eval_centers=centers, eval_scales=scales, eval_areas=areas, eval_bboxes=bboxes)
    out["eval_oks"] = np.array([ev.compute_oks(pk[i], np.asarray(gts[i]["keypoints"]).reshape(-1, 3)[:, :2],
                                               np.asarray(gts[i]["keypoints"]).reshape(-1, 3)[:, 2], gts[i]["area"]) for i in range(B)])
    meta["eval"] = {"image_ids": ids, "ann_ids": anns, "predictions": ev.predictions, "gts": gts,
                    "metrics": {k: float(v) for k, v in metrics.items()}}
    save("extra_r02.npz", **out)
    # ---- L1 with use_target_weight=False (fusion_head.py:653-657,708-712,739-743): values + input gradients
    from models import fusion_head as fh
    hm, off, var, tgt, w, gt = synth_loss_inputs("utw", 2, 17, 32, 24, 96, 128, True)
    hmv, offv, varv = T(hm).requires_grad_(True), T(off).requires_grad_(True), T(var).requires_grad_(True)
    lo = fh.FusionPoseLoss(use_target_weight=False)({"heatmaps": hmv, "offsets": offv, "variances": varv}, T(tgt), T(w), T(gt), (96, 128), (24, 32))
    lo["total_loss"].backward()
    names = ["heatmap_loss", "offset_loss", "peak_loss", "variance_loss", "overlap_loss", "shape_loss", "total_loss"]
    save("loss_utw_r02.npz", hm=hm, off=off, var=var, tgt=tgt, w=w, gt=gt, losses=np.array([float(lo[n]) for n in names]),
         g_hm=N(hmv.grad), g_off=N(offv.grad), g_var=N(varv.grad))
    return meta


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "extra":      # round-2 fixtures only (the others stay byte-identical)
        with open(os.path.join(HERE, "meta.json")) as f:
            meta = json.load(f)
        meta["extra"] = cap_extra()
        with open(os.path.join(HERE, "meta.json"), "w") as f:
            json.dump(meta, f, separators=(",", ":"))
        print("done (extra)")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "video":      # add the video post-processing fixtures only
        cap_video()
        print("done (video)")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "base":       # add the HRFormer-base fixtures without touching the others
        with open(os.path.join(HERE, "meta.json")) as f:
            meta = json.load(f)
        meta["base"] = cap_base()
        with open(os.path.join(HERE, "meta.json"), "w") as f:
            json.dump(meta, f, separators=(",", ":"))
        print("done (base)")
        return
    meta = {}
    print("capturing golden vectors from", REF)
    cap_t1()
    cap_t2()
    meta["attn"] = cap_attn()
    meta["modules"] = cap_modules()
    meta["head_loss"] = cap_head_loss()
    cap_decode()
    meta["models"] = cap_models()
    meta["schedule"] = cap_schedule()
    meta["base"] = cap_base()
    cap_video()
    meta["extra"] = cap_extra()
    meta["torch"] = torch.__version__
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, separators=(",", ":"))
    print("done")


if __name__ == "__main__":
    main()
