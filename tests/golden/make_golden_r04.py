#!/usr/bin/env python3
"""Round-4 additions to the golden vectors, captured from the reference's own modules (build container only, see make_golden.py):

  * the PUBLIC methods of models/fusion_head.py that round 3 only had inside the fused loss kernel: GaussianDistributionConstraint
    (compute_heatmap_variance / variance_alignment_loss / spatial_overlap_loss / distribution_shape_loss / forward, :405-575) called with
    ARBITRARY coordinates, FusionPoseLoss.heatmap_loss / offset_loss / peak_localization_loss (:637-743), LocalGaussianRefinement (:74-128),
    SoftArgmax2D with gradients (:24-71) -- values and input gradients;
  * window_partition / window_reverse (models/hrformer.py:67-114) on shapes that need padding;
  * HRNet-W48 + HeatmapHead eval forward (pose_estimator.py builders), 128x96, B = 1.

    python tests/golden/make_golden_r04.py            # writes tests/golden/public_r04.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (puts the reference on sys.path, stubs the absent third-party packages)
from recipe import synth_input  # noqa: E402

T, N = mg.T, mg.N


def leaf(a):
    return T(a).clone().requires_grad_(True)


def cap_constraint(out, tag, B, K, H, W, win, hin):
    from models import fusion_head as fh
    hm, off, var, tgt, w, gt = mg.synth_loss_inputs("pub_" + tag, B, K, H, W, win, hin, True)
    rng = np.random.default_rng(1234 + K)
    coords = np.stack([rng.uniform(0.5, W - 1.5, (B, K)), rng.uniform(0.5, H - 1.5, (B, K))], -1).astype(np.float32)
    out[f"{tag}.hm"], out[f"{tag}.off"], out[f"{tag}.var"], out[f"{tag}.tgt"] = hm, off, var, tgt
    out[f"{tag}.w"], out[f"{tag}.gt"], out[f"{tag}.coords"] = w, gt, coords
    gc = fh.GaussianDistributionConstraint(target_sigma=2.0, overlap_threshold=0.4)
    # compute_heatmap_variance with an upstream gradient on sigma
    h, c = leaf(hm), leaf(coords)
    sig = gc.compute_heatmap_variance(h, c)
    gs = T(synth_input(tag + "_gsig", tuple(sig.shape)))
    sig.backward(gs)
    out[f"{tag}.sigma"], out[f"{tag}.sigma_g"], out[f"{tag}.sigma_dhm"], out[f"{tag}.sigma_dc"] = N(sig), N(gs), N(h.grad), N(c.grad)
    # the three terms one by one (+ the variance term without the prediction branch), then forward() with mixed upstream weights
    for name, fn in (("t_var_nopred", lambda h, c, v: gc.variance_alignment_loss(h, c, T(w), None)),
                     ("t_var", lambda h, c, v: gc.variance_alignment_loss(h, c, T(w), v)),
                     ("t_ovl", lambda h, c, v: gc.spatial_overlap_loss(h, T(w))),
                     ("t_shape", lambda h, c, v: gc.distribution_shape_loss(h, T(w)))):
        h, c, v = leaf(hm), leaf(coords), leaf(var)
        val = fn(h, c, v)
        val.backward()
        out[f"{tag}.{name}"] = N(val)
        out[f"{tag}.{name}_dhm"] = N(h.grad)
        if c.grad is not None:
            out[f"{tag}.{name}_dc"] = N(c.grad)
        if v.grad is not None:
            out[f"{tag}.{name}_dvar"] = N(v.grad)
    h, c, v = leaf(hm), leaf(coords), leaf(var)
    d = gc(h, c, T(w), v)
    (1.0 * d["variance_loss"] + 0.7 * d["overlap_loss"] + 0.3 * d["shape_loss"]).backward()
    out[f"{tag}.fwd3"] = np.array([float(d["variance_loss"]), float(d["overlap_loss"]), float(d["shape_loss"])], np.float32)
    out[f"{tag}.fwd3_dhm"], out[f"{tag}.fwd3_dc"], out[f"{tag}.fwd3_dvar"] = N(h.grad), N(c.grad), N(v.grad)
    # FusionPoseLoss term methods with arbitrary predicted coordinates
    for utw in (True, False):
        fl = fh.FusionPoseLoss(use_target_weight=utw)
        u = "w" if utw else "nw"
        h = leaf(hm)
        val = fl.heatmap_loss(h, T(tgt), T(w))
        val.backward()
        out[f"{tag}.{u}.hm_loss"], out[f"{tag}.{u}.hm_loss_dhm"] = N(val), N(h.grad)
        o, c = leaf(off), leaf(coords)
        val = fl.offset_loss(o, c, T(gt), T(w), (win, hin), (H, W))
        val.backward()
        out[f"{tag}.{u}.off_loss"], out[f"{tag}.{u}.off_loss_doff"], out[f"{tag}.{u}.off_loss_dc"] = N(val), N(o.grad), N(c.grad)
        c = leaf(coords)
        val = fl.peak_localization_loss(c, T(gt), T(w), (win, hin), (H, W))
        val.backward()
        out[f"{tag}.{u}.peak_loss"], out[f"{tag}.{u}.peak_loss_dc"] = N(val), N(c.grad)
    # SoftArgmax2D with gradients on both outputs; LocalGaussianRefinement about the given coordinates
    h = leaf(hm)
    sa = fh.SoftArgmax2D()
    co, sc = sa(h)
    gco, gsc = T(synth_input(tag + "_gco", tuple(co.shape))), T(synth_input(tag + "_gsc", tuple(sc.shape)))
    (co * gco).sum().add((sc * gsc).sum()).backward()
    out[f"{tag}.sa_coords"], out[f"{tag}.sa_scores"], out[f"{tag}.sa_gco"], out[f"{tag}.sa_gsc"], out[f"{tag}.sa_dhm"] = (
        N(co), N(sc), N(gco), N(gsc), N(h.grad))
    with torch.no_grad():
        for r in (1, 2):
            out[f"{tag}.local_r{r}"] = N(fh.LocalGaussianRefinement(local_radius=r)(T(hm), T(coords)))
        # coordinates on / beyond the border: round-half-even + clamp of the patch centre
        edge = coords.copy()
        edge[:, 0] = (-0.7, 0.5)
        edge[:, 1] = (W - 0.5, H + 2.0)
        edge[:, 2] = (2.5, 3.5)
        out[f"{tag}.edge_coords"], out[f"{tag}.local_edge"] = edge, N(fh.LocalGaussianRefinement(local_radius=2)(T(hm), T(edge)))


def cap_windows(out):
    from models import hrformer as rh
    for tag, shape in (("wa", (2, 9, 10, 8)), ("wb", (1, 14, 7, 16)), ("wc", (3, 5, 3, 8))):
        x = T(synth_input("win_" + tag, shape))
        wins, (Hp, Wp) = rh.window_partition(x, 7)
        back = rh.window_reverse(wins, 7, shape[1], shape[2], Hp, Wp)
        out[f"{tag}.x"], out[f"{tag}.windows"], out[f"{tag}.hpwp"], out[f"{tag}.back"] = N(x), N(wins), np.array([Hp, Wp]), N(back)
        y = T(synth_input("winr_" + tag, tuple(wins.shape)))       # reverse of arbitrary windows (the pad rows are dropped)
        out[f"{tag}.rev_in"], out[f"{tag}.rev_out"] = N(y), N(rh.window_reverse(y, 7, shape[1], shape[2], Hp, Wp))


def cap_w48(out):
    import models
    m = models.PoseEstimator("hrnet_w48", 17, False, "heatmap", True)
    mg.load_recipe(m, salt=44)
    m.eval()
    x = T(synth_input("w48_eval", (1, 3, 128, 96)))
    with torch.no_grad():
        o = m(x)
        kp, sc = m.inference(x, flip=False)
    out["w48_eval_hm"], out["w48_eval_kp"], out["w48_eval_sc"] = N(o["heatmaps"]), N(kp), N(sc)


def main():
    torch.manual_seed(0)
    out = {}
    cap_constraint(out, "k17", 2, 17, 16, 12, 48, 64)
    cap_constraint(out, "k13", 3, 13, 12, 16, 64, 48)      # K = 13: skeleton pairs with an index >= K are skipped (:504-505)
    cap_windows(out)
    cap_w48(out)
    mg.save("public_r04.npz", **out)


def cap_deconv_head():
    """HeatmapHead with the optional deconv stack (models/pose_estimator.py:22-99): three ConvTranspose2d(stride 2) + BN + ReLU layers with
    kernels 4 / 2 / 4 (padding / output_padding as the reference derives them; kernel 3 gives output_padding -1 and raises there), then the 1x1 final layer; train-mode forward + every
    gradient, running statistics after that step, and the eval-mode forward.  -> tests/golden/deconv_r04.npz + its state_dict spec."""
    import json
    from models import pose_estimator as pe
    from recipe import spec_of
    torch.manual_seed(0)
    head = pe.HeatmapHead(32, 17, num_deconv_layers=3, num_deconv_filters=(48, 32, 24), num_deconv_kernels=(4, 2, 4))
    spec = mg.load_recipe(head, salt=51)
    out = {}
    x = T(synth_input("deconv_x", (2, 32, 6, 5))).requires_grad_(True)
    head.train()
    y = head(x)
    gy = T(synth_input("deconv_gy", tuple(y.shape)))
    y.backward(gy)
    out["y_train"], out["gx"] = N(y), N(x.grad)
    for k, p in head.named_parameters():
        out["g." + k] = N(p.grad)
    for k, v in head.state_dict().items():
        if "running" in k:
            out["buf." + k] = N(v)
    head.eval()
    with torch.no_grad():
        out["y_eval"] = N(head(x.detach()))
    mg.save("deconv_r04.npz", **out)
    with open(os.path.join(HERE, "deconv_r04.json"), "w") as f:
        json.dump({"spec": spec}, f, separators=(",", ":"))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "deconv":      # the deconv fixtures only (public_r04.npz stays byte-identical)
        cap_deconv_head()
    else:
        main()
        cap_deconv_head()
