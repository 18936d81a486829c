"""GPU parity tests (`-m gpu`): every call goes through the C-ABI (libposekernels.so) and is compared with the CPU
oracle on the same seeded inputs and with the golden vectors captured from the reference.

Bars: bit-exact for Gaussian targets and argmax indices; fp32 kernels (losses, decoders, optimiser) 1e-4 relative
or better; bf16 network ops are compared norm-wise against the fp32 oracle with the bf16 tolerance stated per test.
"""
import json
import math
import os

import numpy as np
import pytest
import torch

from conftest import rel_err
from recipe import synth_input, synth_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda"


def G(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV, dtype)


def C(t):
    return t.detach().float().cpu().numpy()


@pytest.fixture(scope="module")
def ops():
    from infantposeestimation_gaussianbias_amd import hipops
    return hipops


# ------------------------------------------------------------------------------------------------ T1 / T2
def test_t1_golden_bit_exact(golden, ops):
    from oracle import target as otgt
    z = golden("t1_target.npz")
    off_golden = []
    for ci in range(int(z["n_cfg"])):
        win, hin, wh, hh, sigma, K = z[f"c{ci}_cfg"]
        t, w = ops.gaussian_target(G(z[f"c{ci}_kp"]), G(z[f"c{ci}_vis"]), (win, hin), (int(wh), int(hh)), float(sigma))
        ot, ow = otgt.generate_target_batch(z[f"c{ci}_kp"], z[f"c{ci}_vis"], (win, hin), (wh, hh), float(sigma))
        assert np.array_equal(C(t).view(np.uint32), ot.view(np.uint32)), f"cfg {ci}: HIP != oracle bitwise"
        assert np.array_equal(C(w), ow)
        assert np.array_equal(C(w), z[f"c{ci}_weight"])
        ref = z[f"c{ci}_target"]
        if not np.array_equal(C(t).view(np.uint32), ref.view(np.uint32)):
            # Only possible if this host's numpy float32 exp differs from the capture host's (the LUT is built on the host with the
            # reference's own numpy expression).  Never silent: the test is reported as XFAIL with the measured distance.
            ulp = np.abs(C(t).view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64)).max()
            assert ulp <= 1 and np.array_equal(C(t) != 0, ref != 0), f"cfg {ci}: {ulp} ulp vs golden"
            off_golden.append((ci, int(ulp)))
    if off_golden:
        pytest.xfail(f"HIP == this host's oracle bit for bit, but this host's numpy exp is {off_golden} (cfg, ulp) away from the capture "
                     "host's golden targets")


def test_t1_full_size_properties(ops):
    """BASELINE size (B=64,K=17,64x48): bit-exact vs the oracle on a sample, plus size-independent properties."""
    from oracle import target as otgt
    g = torch.Generator().manual_seed(5)
    B, K = 64, 17
    kp = (torch.rand(B, K, 2, generator=g) * torch.tensor([232.0, 296.0]) - 20.0)
    vis = torch.randint(0, 3, (B, K), generator=g).float()
    t, w = ops.gaussian_target(kp.to(DEV), vis.to(DEV), (192, 256), (48, 64), 2.0)
    t, w = C(t), C(w)
    ot, ow = otgt.generate_target_batch(kp.numpy(), vis.numpy(), (192, 256), (48, 64), 2.0)
    assert np.array_equal(t.view(np.uint32), ot.view(np.uint32)) and np.array_equal(w, ow)
    assert t.min() >= 0 and t.max() <= 1.0
    assert ((t.reshape(B, K, -1) != 0).sum(-1) <= 169).all()                 # support never exceeds the 13x13 patch
    assert (t.reshape(B, K, -1).sum(-1)[vis.numpy() < 0.5] == 0).all()        # invisible joints draw nothing
    # idempotence / determinism: a second launch gives identical bytes
    t2, _ = ops.gaussian_target(kp.to(DEV), vis.to(DEV), (192, 256), (48, 64), 2.0)
    assert np.array_equal(C(t2).view(np.uint32), t.view(np.uint32))


def test_t2_dense_target(golden, ops):
    z = golden("t2_dense_target.npz")
    for ci in range(int(z["n_cfg"])):
        ih, iw, hh, hw, sigma, K = z[f"c{ci}_cfg"]
        h, w = ops.dense_target(G(z[f"c{ci}_kp"]), G(z[f"c{ci}_vis"]), (ih, iw), (int(hh), int(hw)), float(sigma))
        assert np.array_equal(C(w), z[f"c{ci}_weights"])
        assert np.abs(C(h) - z[f"c{ci}_heatmaps"]).max() < 2e-6        # device expf vs numpy exp: few-ulp class


# ------------------------------------------------------------------------------------------------ decoders
@pytest.mark.parametrize("tag", ["d_small", "d_full", "d_sq"])
def test_decoders_vs_golden(golden, ops, tag):
    from infantposeestimation_gaussianbias_amd.utils import postprocess as pp
    z = golden("decode.npz")
    hm, off = G(z[f"{tag}_hm"]), G(z[f"{tag}_off"])
    idx, mv, co = ops.argmax_decode(hm, 1)
    assert np.array_equal(C(idx).astype(np.int32), z[f"{tag}_argmax"])                 # integer-exact
    assert np.array_equal(C(co), z[f"{tag}_d2"]) and np.array_equal(C(mv), z[f"{tag}_d2_scores"])
    assert np.array_equal(C(ops.argmax_decode(hm, 0)[2]), z[f"{tag}_d2_noshift"])
    p, m = pp.get_max_preds(hm)
    assert np.array_equal(C(p), z[f"{tag}_d3_max"]) and np.array_equal(C(m), z[f"{tag}_d3_maxvals"])
    pt, _ = pp.get_max_preds_with_subpixel(hm)
    assert np.abs(C(pt) - z[f"{tag}_d3_taylor"]).max() < 1e-5
    for a, f in ((0.5, 0.5), (-0.3, 1.2)):
        sfx = f"a{a}_f{f}"
        c1, s1 = ops.softargmax_refine_decode(hm, off, torch.tensor([a], device=DEV), torch.tensor([f], device=DEV), 2)
        c0, _ = ops.softargmax_refine_decode(hm, None, torch.tensor([a], device=DEV), None, 2)
        assert np.abs(C(c1) - z[f"{tag}_d1_{sfx}"]).max() < 3e-4
        assert np.abs(C(c0) - z[f"{tag}_d1_nooff_{sfx}"]).max() < 3e-4
        assert np.array_equal(C(s1), z[f"{tag}_d1_scores"])
    f0, _ = pp.fused_decode(hm)
    f1, _ = pp.fused_decode(hm, G(z[f"{tag}_d3_reg"]), G(z[f"{tag}_d3_center"]), G(z[f"{tag}_d3_scale"]), alpha=0.4)
    f2, _ = pp.fused_decode(hm, G(z[f"{tag}_d3_reg"] * 100), None, None, alpha=0.4)
    assert np.abs(C(f0) - z[f"{tag}_d3_fused0"]).max() < 1e-5
    assert rel_err(C(f1), z[f"{tag}_d3_fused1"]) < 1e-5 and rel_err(C(f2), z[f"{tag}_d3_fused2"]) < 1e-5
    assert np.abs(C(pp.coordinate_refinement(hm, G(z[f"{tag}_d3_taylor"]))) - z[f"{tag}_d3_refined"]).max() < 2e-3
    tp = pp.transform_preds(G(z[f"{tag}_d3_taylor"]), G(z[f"{tag}_d3_center"]), G(z[f"{tag}_d3_scale"]), [640, 480])
    assert rel_err(C(tp), z[f"{tag}_d3_transformed"]) < 1e-6

    class _Cfg:
        class TEST:
            FUSION_ALPHA = 0.4
    r = pp.postprocess_predictions({"heatmaps": hm, "coords": G(z[f"{tag}_d3_reg"])},
                                   {"center": G(z[f"{tag}_d3_center"]), "scale": G(z[f"{tag}_d3_scale"])}, _Cfg)
    assert rel_err(C(r["preds"]), z[f"{tag}_d3_pipeline"]) < 1e-5


def test_eval_transform_and_flip_merge(golden, ops):
    from infantposeestimation_gaussianbias_amd.utils import postprocess as pp
    from oracle import decode as odec
    z = golden("decode.npz")
    # heat-px -> image: feed heat-px = input-px/4 so the composite equals train.py:328-336 on input-px
    out = pp.heatmap_to_image_coords(G(z["tp_in"]) / 4, G(z["tp_center"]), G(z["tp_scale"]), (192, 256), (48, 64))
    assert rel_err(C(out), z["tp_out"]) < 1e-6
    a, b = synth_input("fm_a", (2, 17, 16, 12)), synth_input("fm_b", (2, 17, 16, 12))
    pairs = [(i, i + 1) for i in range(1, 17, 2)]
    partner = torch.arange(17, dtype=torch.int32)
    for i, j in pairs:
        partner[i], partner[j] = j, i
    got = ops.flip_merge(G(a), G(b), partner.to(DEV))
    assert np.array_equal(C(got), odec.flip_merge(a, b, pairs))


def test_argmax_ties_nan_and_large_maps(ops):
    hm = torch.zeros(3, 2, 96, 72)
    hm[0, 0, 5, 7] = hm[0, 0, 80, 3] = 2.0          # tie -> lowest flat index
    hm[0, 1] = -1.0                                   # constant map -> index 0
    hm[1, 0, 95, 71] = 1.0                            # last element
    hm[2, 1, 40, 40] = float("nan")                   # NaN propagates as the maximum (torch.max semantics)
    idx, mv, _ = ops.argmax_decode(hm.to(DEV), 0)
    ref = torch.max(hm.view(3, 2, -1), 2)[1]
    assert torch.equal(idx.cpu().long(), ref)
    assert math.isnan(float(mv[2, 1]))


# ------------------------------------------------------------------------------------------------ losses
@pytest.mark.parametrize("tag", ["l_small", "l_peaky", "l_k13", "l_full", "l_k5"])
def test_fusion_loss_fwd_bwd_vs_golden(golden, ops, tag):
    z = golden("head_loss.npz")
    win, hin, H, W = (int(v) for v in z[f"{tag}_size"])
    hm, off, var = (G(z[f"{tag}_{n}"]).requires_grad_(True) for n in ("hm", "off", "var"))
    lam = torch.tensor([1.0, 1.0, 0.5, 0.1, 0.05, 0.05], device=DEV)
    vals = ops.fusion_loss(hm, off, var, G(z[f"{tag}_tgt"]), G(z[f"{tag}_w"]), G(z[f"{tag}_gt"]), (win, hin), 2.0, lam)
    assert np.allclose(C(vals), z[f"{tag}_losses"], rtol=1e-4, atol=1e-6), (C(vals), z[f"{tag}_losses"])
    vals[6].backward()
    assert rel_err(C(hm.grad), z[f"{tag}_ghm"]) < 2e-4
    assert rel_err(C(off.grad), z[f"{tag}_goff"]) < 2e-4
    assert rel_err(C(var.grad), z[f"{tag}_gvar"]) < 2e-4


def test_fusion_loss_full_batch_vs_oracle(ops):
    """B=64, K=17, 64x48 (BASELINE config 2 loss shapes) against the fp64 oracle; plus linearity in grad_output."""
    from oracle import losses as olos
    rng = np.random.default_rng(3)
    B, K, H, W = 64, 17, 64, 48
    hm = (rng.standard_normal((B, K, H, W)) * 0.7).astype(np.float32)
    off = rng.standard_normal((B, K, 2, H, W)).astype(np.float32)
    var = (np.abs(rng.standard_normal((B, K, H, W))) + 0.5).astype(np.float32)
    gt = np.stack([rng.uniform(0, 192, (B, K)), rng.uniform(0, 256, (B, K))], -1).astype(np.float32)
    vis = rng.choice([0.0, 1.0, 2.0], p=[.15, .25, .6], size=(B, K)).astype(np.float32)
    tgt, w = ops.gaussian_target(G(gt), G(vis), (192, 256), (48, 64), 2.0)
    lam = torch.tensor([1.0, 1.0, 0.5, 0.1, 0.05, 0.05], device=DEV)
    thm, toff, tvar = (G(a).requires_grad_(True) for a in (hm, off, var))
    vals = ops.fusion_loss(thm, toff, tvar, tgt, w, G(gt), (192, 256), 2.0, lam)
    (3.0 * vals[6]).backward()
    d = torch.float64
    ohm, ooff, ovar = (torch.from_numpy(a).to(d).requires_grad_(True) for a in (hm, off, var))
    ref = olos.fusion_pose_loss(ohm, ooff, ovar, tgt.cpu().to(d), w.cpu().to(d), torch.from_numpy(gt).to(d), (192, 256))
    got = C(vals)
    for i, n in enumerate(olos.NAMES):
        assert abs(got[i] - float(ref[n])) <= 1e-4 * max(1.0, abs(float(ref[n]))), n
    g = torch.autograd.grad(3.0 * ref["total_loss"], [ohm, ooff, ovar])
    assert rel_err(C(thm.grad), g[0].numpy()) < 2e-4
    assert rel_err(C(toff.grad), g[1].numpy()) < 2e-4
    assert rel_err(C(tvar.grad), g[2].numpy()) < 2e-4


@pytest.mark.parametrize("tag", ["l_small", "l_k13", "l_k5"])
def test_named_losses_vs_golden(golden, tag):
    from infantposeestimation_gaussianbias_amd.models import KeypointMSELoss, losses as L
    z = golden("head_loss.npz")
    hm, tgt, w, gt = (G(z[f"{tag}_{n}"]) for n in ("hm", "tgt", "w", "gt"))
    hp = hm.clone().requires_grad_(True)
    l3 = KeypointMSELoss(True)(hp, tgt, w)
    assert math.isclose(float(l3), float(z[f"{tag}_l3"]), rel_tol=1e-5)
    l3.backward()
    assert rel_err(C(hp.grad), z[f"{tag}_l3_g"]) < 1e-5
    assert math.isclose(float(KeypointMSELoss(False)(hm, tgt, w)), float(z[f"{tag}_l3_now"]), rel_tol=1e-5)
    assert math.isclose(float(L.FusedPoseLoss(True, "mse")(hm, tgt, w)), float(z[f"{tag}_l4_fused_mse"]), rel_tol=1e-5)
    assert math.isclose(float(L.FusedPoseLoss(True, "smoothl1")(hm, tgt, w)), float(z[f"{tag}_l4_fused_sl1"]), rel_tol=1e-5)
    assert math.isclose(float(L.JointsMSELoss(True)(hm, tgt, w)), float(z[f"{tag}_l4_joints"]), rel_tol=1e-5)
    assert math.isclose(float(L.JointsMSELoss(False)(hm, tgt, w)), float(z[f"{tag}_l4_joints_now"]), rel_tol=1e-5)
    assert math.isclose(float(L.MorphologyShapeLoss(1.2, 0.5)(torch.relu(hm), tgt, w)), float(z[f"{tag}_l4_morph"]), rel_tol=2e-4)
    mean, var = L.MorphologyShapeLoss().compute_spatial_statistics(torch.relu(hm))
    assert rel_err(C(mean), z[f"{tag}_l4_mean"]) < 1e-5 and rel_err(C(var), z[f"{tag}_l4_var"]) < 1e-4
    a, b = gt * 0.1, G(z[f"{tag}_gt"][:, ::-1].copy()) * 0.1
    for kind, key in (("smoothl1", "offreg_sl1"), ("l1", "offreg_l1"), ("mse", "offreg_mse")):
        assert math.isclose(float(L.OffsetRegressionLoss(kind)(a, b, w)), float(z[f"{tag}_l4_{key}"]), rel_tol=1e-5)

    class _C:
        class LOSS:
            MORPH_LAMBDA, MORPH_WEIGHT, REG_WEIGHT = 1.2, 0.15, 0.6
    tot, d = L.CombinedLoss(_C)({"heatmaps": torch.relu(hm), "coords": a, "refined_coords": gt * 0.11},
                                {"heatmaps": tgt, "coords": b, "weights": w})
    got = np.array([float(tot)] + [float(d[k]) for k in ("heatmap", "morph", "regression", "refined")])
    assert np.allclose(got, z[f"{tag}_l4_combined"], rtol=2e-4)
    # MorphologyShapeLoss backward (pk_spatial_stats_bwd) against autograd through the CPU oracle of the same formula
    from oracle import losses as olos
    hr = torch.relu(hm).detach().cpu().requires_grad_(True)
    olos.morphology_shape_loss(hr, tgt.cpu(), w.cpu(), 1.2, 0.5).backward()
    hd = torch.relu(hm).detach().requires_grad_(True)
    L.MorphologyShapeLoss(1.2, 0.5)(hd, tgt, w).backward()
    assert rel_err(C(hd.grad), hr.grad.numpy()) < 2e-4


def test_video_postprocess_kernels_vs_golden(golden):
    """pk_temporal_smooth / pk_nms_pose against the reference's outputs (float64 convolution stored as float32: exact up to the
    summation order of <= 7 doubles; suppression masks: identical)."""
    from infantposeestimation_gaussianbias_amd.utils import postprocess as pp
    z = golden("video_post.npz")
    for w in (3, 5, 7):
        assert np.abs(C(pp.temporal_smoothing(G(z["ts_in"]), w, "gaussian")) - z[f"ts_gauss_w{w}"]).max() < 2e-5
        assert np.abs(C(pp.temporal_smoothing(G(z["ts_in"]), w, "moving_average")) - z[f"ts_avg_w{w}"]).max() < 2e-5
    assert np.abs(C(pp.temporal_smoothing(G(z["ts_short_in"]), 5, "gaussian")) - z["ts_short_w5"]).max() < 2e-5
    with pytest.raises(Exception):
        pp.temporal_smoothing(G(z["ts_in"]), 4, "gaussian")                       # the reference fails on even windows too
    for thr in (5, 2, 12):
        kept, keep = pp.nms_pose(G(z["nms_preds"]), G(z["nms_conf"]), float(thr))
        assert keep.shape == (6, 17, 1) and keep.dtype == torch.bool
        assert np.array_equal(C(keep).astype(np.uint8), z[f"nms_keep_t{thr}"]) and np.array_equal(C(kept), z[f"nms_out_t{thr}"])


# ------------------------------------------------------------------------------------------------ optimiser
def test_fused_adamw_matches_oracle():
    from infantposeestimation_gaussianbias_amd import engine
    from oracle import optim as oopt
    torch.manual_seed(1)
    m = torch.nn.Sequential(torch.nn.Linear(33, 17), torch.nn.LayerNorm(17), torch.nn.Linear(17, 5)).to(DEV)
    extra = torch.nn.Linear(4, 4).to(DEV)
    m.add_module("dead", extra)
    ref = {n: p.detach().cpu().clone() for n, p in m.named_parameters()}
    st = {n: (torch.zeros_like(v), torch.zeros_like(v)) for n, v in ref.items()}
    opt = engine.FlatAdamW(m, lr=5e-4, weight_decay=0.01)
    x = torch.randn(64, 33, device=DEV)
    for step in range(1, 6):
        opt.zero_grad()
        loss = (m[2](m[1](m[0](x))) ** 2).mean()
        loss.backward()
        grads = {n: (p.grad.detach().cpu().clone() if p.grad is not None else None) for n, p in m.named_parameters()}
        lr = 5e-4 * step
        opt.set_lr(lr)
        opt.step(grad_scale=0.5)
        for n, g in grads.items():
            if g is None or n.startswith("dead"):
                continue
            oopt.adamw_step(ref[n], g * 0.5, st[n][0], st[n][1], step, lr, 0.0 if oopt.is_no_decay(n) else 0.01)
    for n, p in m.named_parameters():
        assert rel_err(C(p), ref[n].numpy()) < 5e-6, n           # includes the untouched grad-less 'dead' parameters


# ------------------------------------------------------------------------------------------------ model level
def _load(model, spec, salt):
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(spec, salt).items()}, strict=True)
    return model


def test_hrformer_small_eval_forward_vs_golden(golden):
    """Whole model, eval mode, recipe weights, B=1 256x192.  bf16 activations/weights vs the reference's fp32:
    norm-wise tolerance 3e-2 on heatmaps (44 blocks + 80 convs of bf16 rounding), decode within 0.5 px."""
    from infantposeestimation_gaussianbias_amd.models import PoseEstimator
    z, keys = golden("model_level.npz"), golden("state_keys.json")
    m = _load(PoseEstimator("hrformer_small", 17, False, "fusion", True), keys["hrformer_small_fusion"], 40).to(DEV).eval()
    x = G(synth_input("small_eval", (1, 3, 256, 192)))
    with torch.no_grad():
        o = m(x)
    assert o["heatmaps"].shape == (1, 17, 64, 48) and o["offsets"].shape == (1, 17, 2, 64, 48)
    assert rel_err(C(o["heatmaps"]), z["small_eval_hm"]) < 3e-2
    assert abs(float(o["fusion_weight"]) - float(z["small_eval_fw"])) < 1e-6
    kp, sc = m.inference(x, flip=True, flip_pairs=[(i, i + 1) for i in range(1, 17, 2)])
    assert np.abs(C(kp) - z["small_eval_flip_kp"]).max() < 1.0     # soft-argmax of low-contrast random-weight maps is noise-sensitive
    assert rel_err(C(sc), z["small_eval_flip_sc"]) < 5e-2


def test_hrformer_small_train_step_vs_golden(golden):
    """Train-mode forward + loss + backward (DropPath off) against the reference's fp32 numbers: losses within 3e-2, the 41
    grad-less parameters identical.  Gradient norms: rounding activations/gradients to bf16 at op boundaries in the CPU
    oracle already moves 20 of the 779 norms by more than 10 % (none by 25 %), so the bar is <= 60 beyond 10 %, none beyond 35 %."""
    from infantposeestimation_gaussianbias_amd.models import PoseEstimator
    z, keys, meta = golden("model_level.npz"), golden("state_keys.json"), golden("meta.json")["models"]
    m = _load(PoseEstimator("hrformer_small", 17, False, "fusion", True), keys["hrformer_small_fusion"], 40).to(DEV).train()
    m.backbone.drop_path_rate = 0.0
    x = G(synth_input("small_train", (2, 3, 128, 96)))
    o = m(x, G(z["small_train_tgt"]), G(z["small_train_w"]), G(z["small_train_gt"]), input_size=(96, 128))
    o["loss"].backward()
    got = np.array([float(o["losses"][n]) for n in ("heatmap_loss", "offset_loss", "peak_loss", "variance_loss", "overlap_loss",
                                                     "shape_loss", "total_loss")])
    assert np.allclose(got, z["small_train_losses"], rtol=3e-2, atol=1e-3), (got, z["small_train_losses"])
    nograd = sorted(k for k, p in m.named_parameters() if p.grad is None)
    assert nograd == sorted(meta["small_train_nograd"])
    bad, worse = [], []
    for k, p in m.named_parameters():
        gn = meta["small_train_gradnorm"][k]
        if gn > 1e-6 and abs(float(p.grad.norm()) - gn) > 0.1 * gn:
            bad.append((k, float(p.grad.norm()), gn))
        if gn > 1e-6 and abs(float(p.grad.norm()) - gn) > 0.35 * gn:
            worse.append((k, float(p.grad.norm()), gn))
    assert len(bad) <= 60 and not worse, (len(bad), worse[:10], bad[:10])
    sd = m.state_dict()
    for k in z:
        if k.startswith("small_train_buf."):
            assert rel_err(C(sd[k[16:]]), z[k]) < 2e-2, k


def test_hrformer_base_eval_forward_vs_golden(golden):
    """HRFormer-base (C = 78/156/312/624, head_dim 39) runs on the HIP kernels through its 8-aligned padded twin
    (models/padded.py): eval forward + flip-test inference against the reference's fp32 outputs (BASELINE cfg 5, K=13)."""
    from infantposeestimation_gaussianbias_amd import dispatch
    from infantposeestimation_gaussianbias_amd.models import PoseEstimator
    z, keys = golden("model_base.npz"), golden("state_keys.json")
    m = _load(PoseEstimator("hrformer_base", 13, False, "fusion", True), keys["hrformer_base_fusion_k13"], 44).to(DEV).eval()
    assert dispatch.backend_name(m).startswith("hip")
    x = G(synth_input("base_eval", (1, 3, 128, 96)))
    with torch.no_grad():
        o = m(x)
    assert o["heatmaps"].shape == (1, 13, 32, 24)
    assert rel_err(C(o["heatmaps"]), z["base_eval_hm"]) < 3e-2
    assert rel_err(C(o["offsets"])[:, :, :, ::4, ::4], z["base_eval_off"]) < 5e-2
    assert rel_err(C(o["variances"])[:, :, ::4, ::4], z["base_eval_var"]) < 3e-2
    kp, sc = m.inference(x, flip=True, flip_pairs=[(1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12)])
    assert np.abs(C(kp) - z["base_eval_flip_kp"]).max() < 1.0
    assert rel_err(C(sc), z["base_eval_flip_sc"]) < 5e-2
    # the public surface is still the reference's: real shapes in the state_dict, twin invisible
    assert m.state_dict()["backbone.stage2.0.branches.0.0.attn.qkv.weight"].shape == (234, 78)
    assert not any("_pk" in k or "twin" in k for k in m.state_dict())


@pytest.mark.parametrize("name,K,spec,salt", [("hrformer_base", 13, "hrformer_base_fusion_k13", 44), ("hrformer_small", 17, "hrformer_small_fusion", 40)])
def test_flip_inference_batched_and_stream_captured_match_two_pass_eager(golden, monkeypatch, name, K, spec, salt):
    """The serving path of bench.py --config hrformer_base_infer (BASELINE cfg 5; pose_estimator.py:275-329): the flip test as ONE forward
    over [x ; flip(x)] (eval mode: samples are independent), captured into a hipGraph WITH the concurrent branch streams, against the plain
    two-pass single-stream eager inference -- key points within 0.05 heat-map px, scores within 1e-2, and the captured graph replays to the
    same numbers on a second input."""
    from infantposeestimation_gaussianbias_amd import dispatch
    from infantposeestimation_gaussianbias_amd.models import PoseEstimator
    keys = golden("state_keys.json")
    m = _load(PoseEstimator(name, K, False, "fusion", True), keys[spec], salt).to(DEV).eval()
    pairs = [(1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12)]
    xs = [G(synth_input("flip_a", (3, 3, 128, 96))), G(synth_input("flip_b", (3, 3, 128, 96)))]
    monkeypatch.setenv("POSE_FUSED_WIDE_FORCE", "1")          # the wide fused halves at these small launches too
    ref = []
    monkeypatch.setenv("POSE_FLIP_BATCHED", "0")
    dispatch.set_streams(False)
    try:
        with torch.no_grad():
            for x in xs:
                kp, sc = m.inference(x, flip=True, flip_pairs=pairs)
                ref.append((C(kp), C(sc)))
        monkeypatch.setenv("POSE_FLIP_BATCHED", "1")
        dispatch.set_streams(True)
        with torch.no_grad():
            kp, sc = m.inference(xs[0], flip=True, flip_pairs=pairs)          # eager, batched, branch streams
        assert np.abs(C(kp) - ref[0][0]).max() < 0.05 and rel_err(C(sc), ref[0][1]) < 1e-2
        static = xs[0].clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            m.inference(static, flip=True, flip_pairs=pairs)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                out = m.inference(static, flip=True, flip_pairs=pairs)
        torch.cuda.current_stream().wait_stream(s)
        for x, (kp_r, sc_r) in zip(xs, ref):
            static.copy_(x)
            g.replay()
            torch.cuda.synchronize()
            assert np.abs(C(out[0]) - kp_r).max() < 0.05 and rel_err(C(out[1]), sc_r) < 1e-2
    finally:
        dispatch.set_streams(True)


def test_hrformer_base_train_step_vs_golden(golden):
    """Train-mode forward + loss + backward of HRFormer-base through the padded twin: losses, the set of grad-less parameters,
    gradient norms (same bf16 bar as HRFormer-small), selected gradient tensors incl. the head-structured qkv/proj weights,
    BatchNorm running statistics written back to the real buffers."""
    from infantposeestimation_gaussianbias_amd.models import PoseEstimator
    z, keys, meta = golden("model_base.npz"), golden("state_keys.json"), golden("meta.json")["base"]
    m = _load(PoseEstimator("hrformer_base", 13, False, "fusion", True), keys["hrformer_base_fusion_k13"], 44).to(DEV).train()
    m.backbone.drop_path_rate = 0.0
    x = G(synth_input("base_train", (2, 3, 128, 96)))
    o = m(x, G(z["base_train_tgt"]), G(z["base_train_w"]), G(z["base_train_gt"]), input_size=(96, 128))
    o["loss"].backward()
    got = np.array([float(o["losses"][n]) for n in ("heatmap_loss", "offset_loss", "peak_loss", "variance_loss", "overlap_loss",
                                                     "shape_loss", "total_loss")])
    assert np.allclose(got, z["base_train_losses"], rtol=3e-2, atol=1e-3), (got, z["base_train_losses"])
    nograd = sorted(k for k, p in m.named_parameters() if p.grad is None)
    assert nograd == sorted(meta["base_train_nograd"])
    bad, worse = [], []
    for k, p in m.named_parameters():
        gn = meta["base_train_gradnorm"][k]
        if gn > 1e-6 and abs(float(p.grad.norm()) - gn) > 0.1 * gn:
            bad.append((k, float(p.grad.norm()), gn))
        if gn > 1e-6 and abs(float(p.grad.norm()) - gn) > 0.35 * gn:
            worse.append((k, float(p.grad.norm()), gn))
    assert len(bad) <= 80 and not worse, (len(bad), worse[:10], bad[:10])
    for k in z:
        if k.startswith("base_train_g."):
            # Whole gradient tensors, incl. the head-structured qkv / proj weights: a wrong real<->twin mapping would give
            # an L2 distance of ~1.4 (uncorrelated).  bf16 through ~40 blocks at B=2 (BatchNorm over 2 samples) leaves
            # 0.15-0.33 here -- the same level HRFormer-small, which needs no twin, shows against ITS golden gradients
            # (0.18-0.44 for the early layers), while the norms agree to 2 %.
            g = C(dict(m.named_parameters())[k[13:]].grad)
            l2 = float(np.linalg.norm(g - z[k]) / np.linalg.norm(z[k]))
            assert g.shape == z[k].shape and l2 < 0.45 and abs(np.linalg.norm(g) / np.linalg.norm(z[k]) - 1) < 0.1, (k, l2)
    sd = m.state_dict()
    for k in z:
        if k.startswith("base_train_buf."):
            assert rel_err(C(sd[k[15:]]), z[k]) < 2e-2, k
    # a second backward accumulates into .grad like autograd does
    g0 = m.head.shared_layers["0"].weight.grad.clone()
    o2 = m(x, G(z["base_train_tgt"]), G(z["base_train_w"]), G(z["base_train_gt"]), input_size=(96, 128))
    o2["loss"].backward()
    assert rel_err(C(m.head.shared_layers["0"].weight.grad), C(2 * g0)) < 1e-2


def test_hrformer_base_trainer_steps_reduce_loss():
    """engine.Trainer on the padded twin: flat AdamW over the REAL parameters, gradients extracted into the flat buffer,
    twin re-embedded after every optimiser step; eager and hipGraph replay follow the same trajectory."""
    from infantposeestimation_gaussianbias_amd import dispatch, engine
    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.datasets import synthetic_batch
    from infantposeestimation_gaussianbias_amd.models import build_model
    cfg = get_config("hrformer_base")
    cfg.data.input_size, cfg.data.heatmap_size = (96, 128), (24, 32)
    batch = synthetic_batch(2, cfg.data.input_size, cfg.data.heatmap_size, cfg.model.num_keypoints, 2.0, DEV, seed=5)
    traj = {}
    try:
        for graph in (False, True):
            torch.manual_seed(0)
            model = build_model(cfg).to(DEV)
            model.backbone.drop_path_rate = 0.0
            tr = engine.Trainer(model, cfg, iters_per_epoch=2, use_graph=graph, graph_warmup=2, graph_streams=True)
            traj[graph] = [float(tr.step(batch)["loss"].detach()) for _ in range(6)]
            assert (tr._graph is not None) == graph
            assert sum(1 for a in tr.opt.active if not a) > 0 and dispatch.backend_name(model).startswith("hip")
    finally:
        dispatch.set_region_mode(False)
    assert np.all(np.isfinite(traj[False])) and traj[False][-1] < traj[False][0]
    assert np.allclose(traj[False], traj[True], rtol=5e-3), traj


def test_cfg1_hrnet_w18_trajectory_vs_golden(golden):
    """BASELINE config 1 (HRNet(18) + HeatmapHead + KeypointMSELoss, 128x96, B=4, three AdamW steps) on the HIP kernels through
    the 8-aligned padded twin (C = 18/36/72/144 -> 24/40/72/144), optimiser = fused AdamW over the REAL flat parameters."""
    from infantposeestimation_gaussianbias_amd import dispatch, engine
    from infantposeestimation_gaussianbias_amd.models import PoseEstimator
    z, keys = golden("model_level.npz"), golden("state_keys.json")
    m = _load(PoseEstimator("hrnet_w18", 17, False, "heatmap", True), keys["hrnet_w18_heatmap"], 41).to(DEV).train()
    assert dispatch.backend_name(m).startswith("hip")
    x = G(synth_input("cfg1", (4, 3, 128, 96)))
    # eval-mode forward against the fp32 CPU oracle with the same weights: validates the real<->twin embedding without the
    # BatchNorm-over-48-samples noise of the train-mode pass (branch 3 is 4x3 pixels at B=4)
    from oracle import nets as onet
    P = {k: torch.from_numpy(v) for k, v in synth_state_dict(keys["hrnet_w18_heatmap"], 41).items()}
    with torch.no_grad():
        ref = onet.pose_forward(x.cpu(), P, onet.Ctx(train=False))["heatmaps"]
        got = m.eval()(x)["heatmaps"]
    assert rel_err(C(got), ref.numpy()) < 3e-2
    m.train()
    opt = engine.FlatAdamW(m, lr=5e-4, weight_decay=0.01)
    losses = []
    for step in range(3):
        opt.zero_grad()
        o = m(x, G(z["cfg1_tgt"]), G(z["cfg1_w"]))
        if step == 0:
            print("cfg1 train-mode heatmap max-norm error", rel_err(C(o["heatmaps"]), z["cfg1_hm0"]))
        o["loss"].backward()
        losses.append(float(o["loss"].detach()))
        opt.step()
    print("cfg1 losses", losses, z["cfg1_losses"])
    assert math.isclose(losses[0], float(z["cfg1_losses"][0]), rel_tol=3e-2)
    assert np.allclose(losses, z["cfg1_losses"], rtol=8e-2), (losses, z["cfg1_losses"])   # Adam's sign-like first steps amplify bf16 noise
    assert sum(1 for a in opt.active if not a) > 0                                      # stage4's unused fuse layers stay grad-less
    assert m.state_dict()["backbone.stage2.0.branches.0.0.conv1.weight"].shape[0] == 18


def test_hrnet_w32_eval_vs_golden(golden):
    from infantposeestimation_gaussianbias_amd.models import PoseEstimator
    z, keys = golden("model_level.npz"), golden("state_keys.json")
    m = _load(PoseEstimator("hrnet_w32", 17, False, "heatmap", True), keys["hrnet_w32_heatmap"], 42).to(DEV).eval()
    with torch.no_grad():
        o = m(G(synth_input("w32_eval", (1, 3, 128, 96))))
    assert rel_err(C(o["heatmaps"]), z["w32_eval_hm"]) < 3e-2


def test_trainer_two_steps_loss_decreases_and_is_deterministic():
    from infantposeestimation_gaussianbias_amd import engine
    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.datasets import synthetic_batch
    from infantposeestimation_gaussianbias_amd.models import build_model
    cfg = get_config("hrformer_small")
    cfg.data.input_size, cfg.data.heatmap_size = (96, 128), (24, 32)
    batch = synthetic_batch(4, cfg.data.input_size, cfg.data.heatmap_size, 17, 2.0, DEV, seed=3)
    losses = []
    for rep in range(2):
        torch.manual_seed(0)
        model = build_model(cfg).to(DEV)
        model.backbone.drop_path_rate = 0.0
        tr = engine.Trainer(model, cfg, iters_per_epoch=2)      # short warm-up so the LR is non-trivial
        losses.append([float(tr.step(batch)["loss"]) for _ in range(8)])
        assert sum(1 for a in tr.opt.active if not a) == 41
    assert all(np.isfinite(losses[0]))
    assert losses[0][-1] < losses[0][0]
    assert math.isclose(losses[0][0], losses[1][0], rel_tol=2e-3)       # same weights, same batch: same first loss
    assert np.allclose(losses[0], losses[1], rtol=5e-2)                 # later steps: Adam amplifies reduction-order noise


def test_graph_replay_matches_eager_steps():
    """The captured hipGraph step (zero_grad + fwd + bwd + AdamW) follows the same loss trajectory as eager launches."""
    from infantposeestimation_gaussianbias_amd import engine
    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.datasets import synthetic_batch
    from infantposeestimation_gaussianbias_amd.models import build_model
    cfg = get_config("hrformer_small")
    cfg.data.input_size, cfg.data.heatmap_size = (96, 128), (24, 32)
    batches = [synthetic_batch(4, cfg.data.input_size, cfg.data.heatmap_size, 17, 2.0, DEV, seed=30 + i) for i in range(3)]
    traj = {}
    for mode in (False, True):
        torch.manual_seed(0)
        model = build_model(cfg).to(DEV)
        model.backbone.drop_path_rate = 0.0
        tr = engine.Trainer(model, cfg, iters_per_epoch=2, use_graph=mode, graph_warmup=2)
        traj[mode] = [float(tr.step(batches[i % 3])["loss"].detach()) for i in range(7)]
        assert (tr._graph is not None) == mode
    assert np.allclose(traj[False], traj[True], rtol=2e-3), traj


def test_branch_streams_are_bitwise_identical_to_single_stream():
    """Running the resolution branches on concurrent HIP streams must not change a single bit of the training
    trajectory (same kernels, same order of every reduction): a difference here would mean a cross-stream race."""
    from infantposeestimation_gaussianbias_amd import dispatch, engine
    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.datasets import synthetic_batch
    from infantposeestimation_gaussianbias_amd.models import build_model
    cfg = get_config("hrformer_small")
    cfg.data.input_size, cfg.data.heatmap_size = (96, 128), (24, 32)
    batches = [synthetic_batch(8, cfg.data.input_size, cfg.data.heatmap_size, 17, 2.0, DEV, seed=50 + i) for i in range(2)]
    runs = []
    for streams in ("1", "0", "1"):
        os.environ["POSE_STREAMS"] = streams
        try:
            torch.manual_seed(0)
            model = build_model(cfg).to(DEV)
            tr = engine.Trainer(model, cfg, iters_per_epoch=2)
            losses = [tr.step(batches[i % 2])["loss"].detach().clone() for i in range(6)]      # DropPath on (seeded)
            assert dispatch.streams_enabled() == (streams == "1")
            torch.cuda.synchronize()
            runs.append((torch.stack(losses).cpu(), tr.opt.flat.detach().cpu().clone()))
        finally:
            os.environ.pop("POSE_STREAMS", None)
    for losses, flat in runs[1:]:
        assert torch.equal(losses, runs[0][0]), (losses, runs[0][0])
        assert torch.equal(flat, runs[0][1])


def test_region_mode_gradients_match_stream_mode():
    """dispatch.parallel as ONE autograd node (region mode, the capturable variant) gives the same loss and the same
    parameter gradients as the eager multi-stream variant up to the order in which shared inputs' gradients are added."""
    from infantposeestimation_gaussianbias_amd import dispatch, engine
    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.datasets import synthetic_batch
    from infantposeestimation_gaussianbias_amd.models import build_model
    cfg = get_config("hrformer_small")
    cfg.data.input_size, cfg.data.heatmap_size = (96, 128), (24, 32)
    batch = synthetic_batch(4, cfg.data.input_size, cfg.data.heatmap_size, 17, 2.0, DEV, seed=11)
    res = {}
    try:
        for region in (False, True):
            torch.manual_seed(0)
            model = build_model(cfg).to(DEV)
            model.backbone.drop_path_rate = 0.0
            tr = engine.Trainer(model, cfg, iters_per_epoch=2)
            tr.step(batch)                                   # first step installs the gradient views
            dispatch.set_streams(True)
            dispatch.set_region_mode(region)
            out = tr._fwd_bwd(batch)
            torch.cuda.synchronize()
            res[region] = (float(out["loss"].detach()), tr.opt.grad.clone())
    finally:
        dispatch.set_region_mode(False)
    assert math.isclose(res[False][0], res[True][0], rel_tol=1e-6)
    a, b = res[False][1], res[True][1]
    assert float((a - b).norm() / a.norm()) < 2e-2           # bf16 re-association of fan-out gradient sums only


def test_graph_trainer_takes_an_odd_sized_batch_eagerly():
    """A batch whose shape differs from the captured one (short last batch of an epoch) is launched eagerly, then replays resume."""
    from infantposeestimation_gaussianbias_amd import dispatch, engine
    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.datasets import synthetic_batch
    from infantposeestimation_gaussianbias_amd.models import build_model
    cfg = get_config("hrformer_small")
    cfg.data.input_size, cfg.data.heatmap_size = (96, 128), (24, 32)
    full = synthetic_batch(4, cfg.data.input_size, cfg.data.heatmap_size, 17, 2.0, DEV, seed=1)
    short = synthetic_batch(3, cfg.data.input_size, cfg.data.heatmap_size, 17, 2.0, DEV, seed=2)
    try:
        torch.manual_seed(0)
        model = build_model(cfg).to(DEV)
        model.backbone.drop_path_rate = 0.0
        tr = engine.Trainer(model, cfg, iters_per_epoch=2, use_graph=True, graph_warmup=2, graph_streams=True)
        losses = [float(tr.step(full)["loss"].detach()) for _ in range(4)]
        assert tr._graph is not None
        losses.append(float(tr.step(short)["loss"].detach()))          # eager
        losses.append(float(tr.step(full)["loss"].detach()))           # replay again
        assert np.all(np.isfinite(losses)) and tr.opt.step_count == 6
    finally:
        dispatch.set_region_mode(False)


def test_capture_with_streams_but_without_region_mode_raises():
    """The combination that segfaulted hipStreamEndCapture in r01 (eager multi-stream autograd inside a capture) is refused
    with an exception before any cross-stream edge is recorded."""
    from infantposeestimation_gaussianbias_amd import _lib, dispatch
    dispatch.set_streams(True)
    dispatch.set_region_mode(False)
    x = torch.ones(8, device=DEV, requires_grad=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with pytest.raises(_lib.PoseKernelError, match="region mode"):
            with torch.cuda.graph(g, stream=s):
                dispatch.parallel([lambda ins: ins[0] * 2, lambda ins: ins[0] * 3], [[x], [x]])
    torch.cuda.synchronize()
    assert float(dispatch.parallel([lambda ins: ins[0] * 2, lambda ins: ins[0] * 3], [[x], [x]])[1].sum()) == 24.0   # eager still works


def test_eval_forward_between_graph_replays_sees_current_weights():
    """The captured pk_adamw_step rewrites the fp32 masters on every replay; an eval forward after ANY replay must use bf16
    copies packed from the current masters (they were one optimiser step behind from the second eval on)."""
    from infantposeestimation_gaussianbias_amd import dispatch, engine
    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.datasets import synthetic_batch
    from infantposeestimation_gaussianbias_amd.models import build_model
    cfg = get_config("hrformer_small")
    cfg.data.input_size, cfg.data.heatmap_size = (96, 128), (24, 32)
    batch = synthetic_batch(4, cfg.data.input_size, cfg.data.heatmap_size, 17, 2.0, DEV, seed=9)
    try:
        torch.manual_seed(0)
        model = build_model(cfg).to(DEV)
        tr = engine.Trainer(model, cfg, iters_per_epoch=2, use_graph=True, graph_warmup=2, graph_streams=True)
        for _ in range(4):
            tr.step(batch)
        assert tr._graph is not None
        for _ in range(2):                                   # replay, eval, replay, eval
            tr.step(batch)
            with torch.no_grad():
                got = model.eval()(batch["img"])["heatmaps"].clone()
            fresh = build_model(cfg).to(DEV)
            fresh.load_state_dict(model.state_dict())
            with torch.no_grad():
                ref = fresh.eval()(batch["img"])["heatmaps"]
            assert torch.equal(got, ref)
            model.train()
    finally:
        dispatch.set_region_mode(False)


def test_deferred_reductions_match_inline_reductions(monkeypatch):
    """Parameter gradients with the slab reductions postponed to ONE pk_reduce_many launch equal the ones produced by the
    per-layer reduce kernels (same slabs; the two kernels use different fixed fp32 summation trees: equal to rounding)."""
    from infantposeestimation_gaussianbias_amd import engine
    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.datasets import synthetic_batch
    from infantposeestimation_gaussianbias_amd.models import build_model
    cfg = get_config("hrformer_small")
    cfg.data.input_size, cfg.data.heatmap_size = (96, 128), (24, 32)
    batch = synthetic_batch(2, cfg.data.input_size, cfg.data.heatmap_size, 17, 2.0, DEV, seed=21)
    grads = {}
    for defer in ("0", "1"):
        monkeypatch.setenv("POSE_DEFER_REDUCE", defer)
        torch.manual_seed(0)
        model = build_model(cfg).to(DEV)
        model.backbone.drop_path_rate = 0.0
        tr = engine.Trainer(model, cfg, iters_per_epoch=2)
        tr._fwd_bwd(batch)
        tr.opt.install_grad_views()           # gradient sinks from here on (the views now hold the autograd-mode gradients)
        auto = tr.opt.grad.clone()
        tr._fwd_bwd(batch)
        torch.cuda.synchronize()
        grads[defer] = tr.opt.grad.clone()
        # sink-mode gradients (stored by the backward kernels) == autograd-mode gradients of the first pass
        assert float((auto - grads[defer]).norm() / auto.norm()) < 1e-5
    a, b = grads["0"], grads["1"]
    assert float((a - b).abs().max() / a.abs().max()) < 1e-6      # same slabs, two fixed fp32 summation trees


def test_graph_with_branch_streams_matches_eager_steps():
    """hipGraph capture WITH concurrent branch streams (region-mode fork/join) follows the loss trajectory of the same
    autograd structure launched eagerly."""
    from infantposeestimation_gaussianbias_amd import dispatch, engine
    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.datasets import synthetic_batch
    from infantposeestimation_gaussianbias_amd.models import build_model
    cfg = get_config("hrformer_small")
    cfg.data.input_size, cfg.data.heatmap_size = (96, 128), (24, 32)
    batches = [synthetic_batch(4, cfg.data.input_size, cfg.data.heatmap_size, 17, 2.0, DEV, seed=30 + i) for i in range(3)]
    traj = {}
    try:
        for mode in (False, True):
            torch.manual_seed(0)
            model = build_model(cfg).to(DEV)
            model.backbone.drop_path_rate = 0.0
            tr = engine.Trainer(model, cfg, iters_per_epoch=2, use_graph=mode, graph_warmup=2, graph_streams=True)
            traj[mode] = [float(tr.step(batches[i % 3])["loss"].detach()) for i in range(7)]
            assert (tr._graph is not None) == mode
    finally:
        dispatch.set_region_mode(False)
    assert np.allclose(traj[False], traj[True], rtol=2e-3), traj


def _dp_gpu_worker(rank, world, port, q, use_graph=False):
    import os
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from infantposeestimation_gaussianbias_amd import engine
    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.datasets import synthetic_batch
    from infantposeestimation_gaussianbias_amd.models import build_model
    cfg = get_config("hrformer_small")
    cfg.data.input_size, cfg.data.heatmap_size = (96, 128), (24, 32)
    shard = synthetic_batch(2, cfg.data.input_size, cfg.data.heatmap_size, 17, 2.0, "cuda", seed=50 + rank)
    torch.manual_seed(rank)                                  # ranks start from DIFFERENT weights: broadcast repairs it
    model = build_model(cfg).to("cuda")
    model.backbone.drop_path_rate = 0.0
    tr = engine.Trainer(model, cfg, iters_per_epoch=2, use_graph=use_graph, graph_warmup=1, graph_streams=use_graph, bucket_mb=4.0)
    init = tr.opt.flat.clone()
    if use_graph:
        # steps 1-2 eager (warm-up), step 3 captures forward+backward (RCCL/gloo and AdamW stay outside the graph), then replays
        losses = [float(tr.step(shard)["loss"].detach()) for _ in range(5)]
        torch.cuda.synchronize()
        assert tr._graph is not None and not tr._graph_has_opt
        q.put((rank, init.cpu().numpy(), np.asarray(losses), tr.opt.grad.cpu().numpy(), tr.opt.flat.cpu().numpy(), losses[-1]))
        dist.barrier()
        dist.destroy_process_group()
        return
    # local gradient of this shard from the broadcast weights, no exchange
    world_saved, tr.comm.world = tr.comm.world, 1
    tr._fwd_bwd(shard)
    tr.opt.install_grad_views()                              # first backward: adopt autograd's gradients into the flat buffer
    tr._fwd_bwd(shard)                                       # second pass: same kernels as every later step (gradient sinks,
    local = tr.opt.grad.clone()                              # deferred slab reductions) -> bitwise comparable with the step below
    tr.comm.world = world_saved
    tr.comm.finish()                                         # builds the buckets (and sums the probe gradients, which the step overwrites)
    # (the probe pass also advanced the BN running statistics; training-mode gradients do not depend on them)
    tr.comm.record_exposed = True
    out = tr.step(shard)
    torch.cuda.synchronize()
    ex = tr.comm.exposed_ms()                                # what bench.py --gpus N reports as comm.exposed_comm_ms_per_step
    assert ex is not None and ex > 0.0 and tr.comm.exposed_ms() is None
    # backward milestones launched the all-reduce of the head's / late stages' buckets before backward had finished
    assert len(tr.comm.buckets) >= 4 and 1 <= tr.comm.launched_early < len(tr.comm.buckets), (tr.comm.launched_early, len(tr.comm.buckets))
    # numpy arrays are pickled by value (tensors would travel as file descriptors the exiting worker takes with it)
    q.put((rank, init.cpu().numpy(), local.cpu().numpy(), tr.opt.grad.cpu().numpy(), tr.opt.flat.cpu().numpy(), float(out["loss"])))
    dist.barrier()
    dist.destroy_process_group()


def _run_two_ranks(use_graph):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_gpu_worker, args=(r, 2, port, q, use_graph)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=150) for _ in procs], key=lambda t: t[0])
    finally:
        for p in procs:
            p.join(20)
            if p.is_alive():            # a rank that died took its peer's collective with it: do not leave the survivor on the GPU
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    return res


def test_trainer_data_parallel_graph_replay_two_ranks_one_gpu():
    """N>1 graph mode: the captured hipGraph holds zero_grad+forward+backward only; the gradient exchange and AdamW run
    eagerly between replays.  Both ranks must hold identical summed gradients and weights after three steps, and the
    loss must move (the optimiser really ran on the replayed gradients)."""
    (_, i0, l0, g0, f0, _), (_, i1, l1, g1, f1, _) = _run_two_ranks(True)
    assert np.array_equal(i0, i1)
    assert np.array_equal(g0, g1) and np.array_equal(f0, f1)
    assert not np.array_equal(f0, i0)
    assert np.all(np.isfinite(l0)) and np.all(np.isfinite(l1)) and l0[4] != l0[2] != l0[0]


def test_trainer_data_parallel_two_ranks_one_gpu():
    """World-size-2 Trainer step (gloo carrying the HIP-produced flat gradient): summed gradient == sum of the ranks'
    local gradients, and both ranks hold identical weights before and after the step (reference: DDP in train.py)."""
    (_, i0, l0, g0, f0, loss0), (_, i1, l1, g1, f1, loss1) = _run_two_ranks(False)
    assert np.array_equal(i0, i1)                                    # broadcast of rank 0's initial state
    assert np.array_equal(g0, g1) and np.array_equal(f0, f1)         # same summed gradient, same updated weights
    assert np.array_equal(g0, l0 + l1)                               # the exchange is an exact two-term sum
    assert not np.array_equal(f0, i0) and np.isfinite(loss0) and np.isfinite(loss1)


def _dp_nccl_worker(rank, world, port, q):
    import os
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from infantposeestimation_gaussianbias_amd import engine
    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.datasets import synthetic_batch
    from infantposeestimation_gaussianbias_amd.models import build_model
    cfg = get_config("hrformer_small")
    cfg.data.input_size, cfg.data.heatmap_size = (96, 128), (24, 32)
    dev = f"cuda:{rank}"
    shard = synthetic_batch(2, cfg.data.input_size, cfg.data.heatmap_size, 17, 2.0, dev, seed=50 + rank)
    torch.manual_seed(rank)
    model = build_model(cfg).to(dev)
    model.backbone.drop_path_rate = 0.0
    tr = engine.Trainer(model, cfg, iters_per_epoch=2, use_graph=True, graph_warmup=2, graph_streams=True, bucket_mb=4.0)
    losses = [float(tr.step(shard)["loss"].detach()) for _ in range(6)]
    torch.cuda.synchronize()
    q.put((rank, np.asarray(losses), tr.opt.flat.cpu().numpy(), tr._graph is not None, tr._graph_has_comm, tr._graph_has_opt))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_data_parallel_rccl_whole_step_graph():
    """RCCL (`nccl` backend), one rank per GPU: the captured hipGraph holds forward + backward + the bucketed all-reduces (issued from
    backward milestones on RCCL's stream) + AdamW.  Needs two devices: skipped on the one-GPU test box, run wherever two are visible."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, l0, f0, g0, c0, o0), (_, l1, f1, g1, c1, o1) = res
    assert g0 and g1 and c0 == c1 and o0 == o1            # graph replay on both ranks, same capture form
    assert np.array_equal(f0, f1)                          # identical weights after six steps
    assert np.all(np.isfinite(l0)) and l0[-1] != l0[0]


# ------------------------------------------------------------------------------------------------ bf16-aware oracle
# The fp32 golden comparisons above carry the bf16 rounding of ~150 stored tensors (3e-2).  Here the CPU oracle rounds at the SAME
# storage points as the HIP path (oracle/nets.py: Ctx.q = bf16_storage rounds activations forward and their gradients backward; conv /
# linear weights are the bf16 compute copies; BatchNorm normalises the rounded conv output with fp32-accumulator statistics).
#
# What bounds the agreement is not where the rounding happens but that bf16 storage makes the random-weight, train-mode network CHAOTIC:
# the bf16-aware oracle run twice on inputs that differ by 1e-6 relative (one fp32 ulp) disagrees with ITSELF by 5 % (L2) on the heatmaps
# and by 15 % / 23 % / 32 % (median / 90th / 99th percentile over the 779 tensors) on the parameter gradients, while the fp32 oracle
# under the same perturbation moves by 1e-5 / 1e-3 (measured, r02: the numbers the HIP path shows against the oracle are the same 15 /
# 22 / 30 %).  A fixed tight bar would therefore test the noise, not the kernels.  The train-mode tests below measure that self-distance
# ("noise floor") in the test itself and require the HIP path to be no further from the oracle than 1.5x the oracle is from itself; the
# tight, noise-free evidence is per operator (tests/test_gpu_network_ops.py: 2e-3 .. 1e-2 against storage-aware references).
BF16_L2, BF16_MAX = 1.5e-2, 2.5e-2     # eval forward, whole model: relative L2 / max-norm (measured 5e-3 .. 1e-2 / 1.1e-2 .. 1.7e-2)


def _bf16_oracle(keys, salt, x, train):
    from oracle import nets as onet
    P32 = {k: torch.from_numpy(v).clone() for k, v in synth_state_dict(keys, salt).items()}
    for v in P32.values():
        if v.is_floating_point():
            v.requires_grad_(train)
    P = onet.bf16_weights(P32)
    ctx = onet.Ctx(train=train, q=onet.bf16_storage)
    return onet.pose_forward(x, P, ctx), P32, ctx


def _l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-30))


def _quant(vals):
    v = np.array(sorted(vals))
    return np.array([v[len(v) // 2], v[int(len(v) * 0.9)], v[int(len(v) * 0.99)]])


@pytest.mark.parametrize("name,bb,K,head,salt,shape", [("hrformer_small_fusion", "hrformer_small", 17, "fusion", 40, (1, 3, 256, 192)),
                                                       ("hrformer_base_fusion_k13", "hrformer_base", 13, "fusion", 44, (1, 3, 128, 96)),
                                                       ("hrnet_w18_heatmap", "hrnet_w18", 17, "heatmap", 41, (2, 3, 128, 96)),
                                                       ("hrnet_w32_heatmap", "hrnet_w32", 17, "heatmap", 42, (1, 3, 128, 96))])
def test_eval_forward_vs_bf16_aware_oracle(golden, name, bb, K, head, salt, shape):
    """Whole-model eval forward (running statistics: a fixed, well-conditioned function) against the oracle with bf16 rounding emulated
    at the kernels' storage points: relative L2 <= 1.5e-2 and max-norm <= 2.5e-2 on every output map."""
    from infantposeestimation_gaussianbias_amd.models import PoseEstimator
    keys = golden("state_keys.json")
    m = _load(PoseEstimator(bb, K, False, head, True), keys[name], salt).to(DEV).eval()
    x = G(synth_input("bf16_" + name, shape))
    with torch.no_grad():
        o = m(x)
        ref, _, _ = _bf16_oracle(keys[name], salt, x.cpu(), False)
    rep = {k: (round(_l2(C(o[k]), ref[k].numpy()), 5), round(rel_err(C(o[k]), ref[k].numpy()), 5)) for k in ("heatmaps", "offsets", "variances") if k in ref}
    print(name, "eval vs bf16-aware oracle (relative L2, max-norm)", rep)
    assert all(l2 < BF16_L2 and mx < BF16_MAX for l2, mx in rep.values()), rep


# analytically zero gradients (the last block's fc2 bias of branches 1-3 of the LAST module feeds only 1x1 conv + train-mode BatchNorm,
# which removes any per-channel constant): both sides hold rounding noise there, a relative distance means nothing
_ZERO_GRAD = ("stage4.1.branches.1.1.mlp.fc2.bias", "stage4.1.branches.2.1.mlp.fc2.bias", "stage4.1.branches.3.1.mlp.fc2.bias")


def test_hrformer_small_train_step_vs_bf16_aware_oracle(golden):
    """Train-mode forward + fused loss + backward (DropPath off, B = 4, 256x192) against the bf16-aware oracle, with the bar calibrated
    by the oracle's own sensitivity: losses, heatmaps, and EVERY parameter gradient by relative L2 distance and cosine similarity (median /
    90th / 99th percentile over the tensors) must be within 1.5x of the distance between two oracle runs whose inputs differ by 1e-6."""
    from infantposeestimation_gaussianbias_amd.models import PoseEstimator
    from oracle import losses as olos, train_step as ots
    keys = golden("state_keys.json")["hrformer_small_fusion"]
    m = _load(PoseEstimator("hrformer_small", 17, False, "fusion", True), keys, 40).to(DEV).train()
    m.backbone.drop_path_rate = 0.0
    img, tg, tw, kp = ots.synthetic_batch(4, (192, 256), (48, 64), 17, 2.0, seed=77)
    o = m(img.to(DEV), tg.to(DEV), tw.to(DEV), kp.to(DEV), input_size=(192, 256))
    o["loss"].backward()
    names = [k for k, _ in m.named_parameters()]

    def oracle_run(x):
        ref, P32, _ = _bf16_oracle(keys, 40, x, True)
        rl = olos.fusion_pose_loss(ref["heatmaps"], ref["offsets"], ref["variances"], tg, tw, kp, (192, 256))
        gr = torch.autograd.grad(rl["total_loss"], [P32[k] for k in names], allow_unused=True)
        return ref["heatmaps"].detach().numpy(), np.array([float(rl[n]) for n in olos.NAMES]), dict(zip(names, gr))

    hm_a, loss_a, g_a = oracle_run(img)
    hm_b, loss_b, g_b = oracle_run(img * (1 + 1e-6 * torch.randn(img.shape, generator=torch.Generator().manual_seed(1))))
    got = np.array([float(o["losses"][n]) for n in olos.NAMES])
    keep = [k for k in names if g_a[k] is not None and not k.endswith(_ZERO_GRAD) and float(g_a[k].norm()) > 1e-7]
    for k, p in m.named_parameters():
        assert (p.grad is None) == (g_a[k] is None), k                       # the same grad-less set
    hip = {k: C(dict(m.named_parameters())[k].grad) for k in keep}
    floor_l2, hip_l2 = _quant(_l2(g_b[k].numpy(), g_a[k].numpy()) for k in keep), _quant(_l2(hip[k], g_a[k].numpy()) for k in keep)
    floor_cos, hip_cos = min(_cos(g_b[k].numpy(), g_a[k].numpy()) for k in keep), min(_cos(hip[k], g_a[k].numpy()) for k in keep)
    floor_hm, hip_hm = _l2(hm_b, hm_a), _l2(C(o["heatmaps"]), hm_a)
    print("losses HIP / oracle / oracle'", got, loss_a, loss_b)
    print(f"heatmaps L2: HIP {hip_hm:.4f}, oracle self-distance {floor_hm:.4f}")
    print(f"gradient L2 (median, p90, p99) over {len(keep)} tensors: HIP {hip_l2}, oracle self-distance {floor_l2}; min cosine HIP {hip_cos:.4f}, self {floor_cos:.4f}")
    assert np.all(np.abs(got - loss_a) <= 3 * np.abs(loss_b - loss_a) + 1e-2 * np.abs(loss_a) + 1e-4), (got, loss_a, loss_b)
    assert hip_hm <= 1.5 * floor_hm + 5e-3
    assert np.all(hip_l2 <= 1.5 * floor_l2 + 1e-2), (hip_l2, floor_l2)
    assert hip_cos >= floor_cos - 0.05


def test_hrformer_small_expected_gradient_vs_bf16_aware_oracle(golden):
    """Well-conditioned whole-model gradient check (VERDICT r02 #1b): BatchNorm in eval mode (running statistics: no batch coupling),
    everything else exactly as in training (LayerNorm, attention, GELU, fused loss; DropPath is the identity in eval), B = 2 at 256x192.

    A single bf16 run cannot be held to a tight bar: storage rounding makes ANY bf16 implementation of this network disagree with a
    copy of itself whose input moved by 1e-5 relative -- measured on the oracle in this configuration: 5.4 % median / 7.7 % p90 / 12 %
    p99 per-tensor gradient L2 (train-mode BatchNorm: 15 / 23 / 32 %), of which ~4 % is zero-mean noise and the rest a bias common to
    all bf16 runs.  So the test compares EXPECTATIONS: the mean gradient over K dithered inputs (x * (1 + 1e-5 n_i)) on the HIP path
    against the mean over K differently dithered inputs on the bf16-aware oracle.  The common bias cancels, the noise shrinks by
    sqrt(K), and any systematic error of a kernel (a wrong factor on a low-magnitude gradient included) stays.  The noise left is
    measured in the same run as the distance d0 between the oracle's own two half-means (K/2 runs each; measured at K = 32: 2.6 % median,
    4.8 % p99, 6.5 % max -- two K-means are expected d0 / sqrt(2) apart); bars: every tensor within 3 % (fp32-accumulated head / stem
    weight gradients 1.5 %) or 2x its own d0, cosine >= 0.999 (or the half-vs-half cosine - 5e-4).
    Measured on MI355X (r03): HIP vs oracle 1.3 % median / 1.9 % p90 / 2.8 % p99 / 3.9 % max over 776 tensors, i.e. 0.7x the oracle's own
    half-vs-half distance (1.9 / 2.7 / 3.8 / 4.8 %) as two equally noisy means should be.  The only tensors further from the oracle than
    their noise are the variance branch's (d = 0.5-1 % at d0 = 0.1 %): its loss term sends the SAME value to every pixel, and the HIP
    path stores that gradient in bf16 before the 1x1 conv's backward -- one rounding error shared by all pixels instead of averaging out."""
    from infantposeestimation_gaussianbias_amd.models import PoseEstimator
    from oracle import losses as olos, train_step as ots
    keys = golden("state_keys.json")["hrformer_small_fusion"]
    K, B, size = 32, 2, (192, 256)      # 32 oracle runs of ~1.7 s on 8 cores
    m = _load(PoseEstimator("hrformer_small", 17, False, "fusion", True), keys, 40).to(DEV).eval()
    img, tg, tw, kp = ots.synthetic_batch(B, size, (48, 64), 17, 2.0, seed=77)
    names = [k for k, _ in m.named_parameters()]
    dither = lambda i: img * (1 + 1e-5 * torch.randn(img.shape, generator=torch.Generator().manual_seed(100 + i)))

    def oracle_run(x):
        from oracle import nets as onet
        P32 = {k: torch.from_numpy(v).clone() for k, v in synth_state_dict(keys, 40).items()}
        for k in names:
            P32[k].requires_grad_(True)
        ref = onet.pose_forward(x, onet.bf16_weights(P32), onet.Ctx(train=False, q=onet.bf16_storage))      # ... BatchNorm on running statistics
        rl = olos.fusion_pose_loss(ref["heatmaps"], ref["offsets"], ref["variances"], tg, tw, kp, size)
        gr = torch.autograd.grad(rl["total_loss"], [P32[k] for k in names], allow_unused=True)
        return {k: (None if g is None else g.double()) for k, g in zip(names, gr)}, float(rl["total_loss"])

    def hip_run(x):
        m.zero_grad(set_to_none=True)
        o = m(x.to(DEV), tg.to(DEV), tw.to(DEV), kp.to(DEV), input_size=size)
        o["loss"].backward()
        return {k: (None if p.grad is None else p.grad.detach().double().cpu()) for k, p in m.named_parameters()}, float(o["loss"])

    def mean_of(runs):
        acc, losses = None, []
        for g, l in runs:
            losses.append(l)
            if acc is None:
                acc = {k: (None if v is None else v.clone()) for k, v in g.items()}
            else:
                for k, v in g.items():
                    if v is not None:
                        acc[k] += v
        return {k: (None if v is None else v / len(losses)) for k, v in acc.items()}, float(np.mean(losses))

    half_a, loss_a = mean_of(oracle_run(dither(i)) for i in range(0, K, 2))
    half_b, loss_b = mean_of(oracle_run(dither(i)) for i in range(1, K, 2))
    hip, loss_h = mean_of(hip_run(dither(K + i)) for i in range(K))
    for k in names:
        assert (hip[k] is None) == (half_a[k] is None), k                      # the same grad-less set
    orc = {k: (None if half_a[k] is None else 0.5 * (half_a[k] + half_b[k])) for k in names}
    keep = [k for k in names if orc[k] is not None and not k.endswith(_ZERO_GRAD) and float(orc[k].norm()) > 1e-7]
    assert len(keep) >= 770
    bad, rows = [], []
    for k in keep:
        d, c = _l2(hip[k].numpy(), orc[k].numpy()), _cos(hip[k].numpy(), orc[k].numpy())
        d0, c0 = _l2(half_a[k].numpy(), half_b[k].numpy()), _cos(half_a[k].numpy(), half_b[k].numpy())
        tight = k.startswith("head.") or k.startswith("backbone.conv")       # fp32-accumulated head / stem weight gradients
        rows.append((d, d0, c, k))
        if d > max(1.5e-2 if tight else 3e-2, 2.0 * d0) or c < min(0.999, c0 - 5e-4):
            bad.append((k, round(d, 4), round(d0, 4), round(c, 5), round(c0, 5)))
    dist = np.array(sorted(r[0] for r in rows))
    self_d = np.array(sorted(r[1] for r in rows))
    print("largest distance / half-vs-half ratios:", sorted(((round(r[0] / max(r[1], 1e-9), 2), r[3]) for r in rows), reverse=True)[:5])
    print(f"expected-gradient distance HIP vs oracle over {len(keep)} tensors (median / p90 / p99 / max): "
          f"{dist[len(dist) // 2]:.4f} {dist[int(len(dist) * .9)]:.4f} {dist[int(len(dist) * .99)]:.4f} {dist[-1]:.4f}; oracle half-vs-half: "
          f"{self_d[len(dist) // 2]:.4f} {self_d[int(len(dist) * .9)]:.4f} {self_d[int(len(dist) * .99)]:.4f} {self_d[-1]:.4f}; losses {loss_h:.3f} / {0.5 * (loss_a + loss_b):.3f}")
    assert not bad, (len(bad), bad[:12])
    assert dist[len(dist) // 2] <= 2.5e-2 and dist[int(len(dist) * .99)] <= 5e-2, dist
    assert abs(loss_h - 0.5 * (loss_a + loss_b)) <= 3e-3 * abs(loss_a) + 3 * abs(loss_a - loss_b)


def test_cfg2_full_size_graph_step_is_reproducible_and_matches_eager():
    """BASELINE cfg 2 at its exact size (HRFormer-small + fusion head, 256x192, B = 64, bf16, DropPath 0.1, BatchNorm in train mode)
    through the path bench.py times: whole-step hipGraph replay with the resolution branches forked over HIP streams.  Three replayed
    steps: finite losses, bit-identical losses AND weights between two runs from the same seed (no atomics, fixed reduction orders, the
    DropPath draws come from the captured Philox state), the same trajectory as eager launches of the same autograd structure within
    5e-3, and a bounded HBM footprint."""
    from infantposeestimation_gaussianbias_amd import dispatch, engine
    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.datasets import synthetic_batch
    from infantposeestimation_gaussianbias_amd.models import build_model
    cfg = get_config("hrformer_small")
    assert tuple(cfg.data.input_size) == (192, 256) and tuple(cfg.data.heatmap_size) == (48, 64)
    cfg.train.batch_size = 64
    batch = synthetic_batch(64, cfg.data.input_size, cfg.data.heatmap_size, 17, cfg.data.sigma, DEV, seed=1234)
    torch.cuda.reset_peak_memory_stats()
    runs = {}
    try:
        for tag, graph in (("graph_a", True), ("graph_b", True), ("eager", False)):
            torch.manual_seed(42)
            model = build_model(cfg).to(DEV)
            assert model.backbone.drop_path_rate > 0
            tr = engine.Trainer(model, cfg, iters_per_epoch=1000, use_graph=graph, graph_warmup=2, graph_streams=True)
            losses = [tr.step(batch)["loss"].detach().clone() for _ in range(5)]           # 2 eager warm-up steps + capture + 3 replays
            torch.cuda.synchronize()
            assert (tr._graph is not None) == graph and dispatch.streams_enabled()
            runs[tag] = (torch.stack(losses).cpu(), tr.opt.flat.detach().cpu().clone())
            del tr, model
    finally:
        dispatch.set_region_mode(False)
    la, wa = runs["graph_a"]
    assert torch.isfinite(la).all() and torch.isfinite(wa).all()
    assert torch.equal(la, runs["graph_b"][0]) and torch.equal(wa, runs["graph_b"][1])
    le = runs["eager"][0]
    assert torch.equal(la[:2], le[:2])                                           # the two warm-up steps ARE eager launches
    assert np.allclose(la.numpy(), le.numpy(), rtol=5e-3), (la, le)
    gb = torch.cuda.max_memory_reserved() / 2 ** 30
    print(f"cfg 2 full size: losses {la.tolist()}, hbm_reserved {gb:.2f} GiB")
    assert gb < 24.0                                                             # measured 10-12 GiB for one trainer (B = 64) + the suite's residue


def test_cfg1_trajectory_vs_bf16_aware_oracle(golden):
    """BASELINE config 1 (HRNet-W18 + heatmap head + KeypointMSELoss, B=4, 128x96): three AdamW steps on the HIP path against three steps
    of the bf16-aware oracle from the same weights; bar = 2x the distance between two oracle trajectories whose inputs differ by 1e-6."""
    from infantposeestimation_gaussianbias_amd import engine
    from infantposeestimation_gaussianbias_amd.models import PoseEstimator
    from oracle import losses as olos, nets as onet, optim as oopt
    z, keys = golden("model_level.npz"), golden("state_keys.json")
    m = _load(PoseEstimator("hrnet_w18", 17, False, "heatmap", True), keys["hrnet_w18_heatmap"], 41).to(DEV).train()
    x = G(synth_input("cfg1", (4, 3, 128, 96)))
    opt = engine.FlatAdamW(m, lr=5e-4, weight_decay=0.01)
    pnames = [k for k, _ in m.named_parameters()]
    got = []
    for step in range(1, 4):
        opt.zero_grad()
        o = m(x, G(z["cfg1_tgt"]), G(z["cfg1_w"]))
        o["loss"].backward()
        got.append(float(o["loss"].detach()))
        opt.step()

    def oracle_traj(xc):
        P = {k: torch.from_numpy(v).clone() for k, v in synth_state_dict(keys["hrnet_w18_heatmap"], 41).items()}
        state, out = {}, []
        for step in range(1, 4):
            for k in pnames:
                P[k].requires_grad_(True)
            ctx = onet.Ctx(train=True, q=onet.bf16_storage)
            ro = onet.pose_forward(xc, onet.bf16_weights(P), ctx)
            loss = olos.keypoint_mse(ro["heatmaps"], torch.from_numpy(z["cfg1_tgt"]), torch.from_numpy(z["cfg1_w"]))
            grads = torch.autograd.grad(loss, [P[k] for k in pnames], allow_unused=True)
            out.append(float(loss.detach()))
            with torch.no_grad():
                for k, g in zip(pnames, grads):
                    P[k].requires_grad_(False)
                    if g is None:
                        continue
                    if k not in state:
                        state[k] = (torch.zeros_like(P[k]), torch.zeros_like(P[k]))
                    oopt.adamw_step(P[k], g, state[k][0], state[k][1], step, 5e-4, 0.0 if oopt.is_no_decay(k) else 0.01)
                onet.apply_bn_updates(P, ctx)
        return np.array(out)

    xc = x.cpu()
    want = oracle_traj(xc)
    want2 = oracle_traj(xc * (1 + 1e-6 * torch.randn(xc.shape, generator=torch.Generator().manual_seed(1))))
    floor = np.abs(want2 - want) / want
    dist = np.abs(np.array(got) - want) / want
    print("cfg1 trajectory HIP / oracle / oracle'", got, want, want2, "relative distance", dist, "self", floor)
    assert dist[0] < 5e-3                                   # first loss: forward only
    # the oracle's self-distance is ONE draw per step (measured 0.3 % / 1.0 % / 0.25 % on one box, 0.3 / 0.4 / 1.1 on another): the noise
    # scale of the later steps is the largest of the three, not the draw of that step
    assert np.all(dist <= 2 * floor.max() + 1.5e-2), (dist, floor)


def test_head_output_epilogue_meets_north_star_1e3(N=None):
    """north_star bar "heatmaps within 1e-3 rel fp32": the fp32-output epilogue of the head's 1x1 conv, given identical bf16 features and the
    bf16 compute copy of the weight, against fp32 PyTorch on those features."""
    from infantposeestimation_gaussianbias_amd import nnops
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(256, 17, 1)
    with torch.no_grad():
        conv.weight.copy_(conv.weight.to(torch.bfloat16).float())
    x = (torch.randn(4, 64, 48, 256) * 0.7).to(torch.bfloat16)
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), conv.weight, conv.bias)
    holder = torch.nn.Sequential(conv).to(DEV)
    with torch.no_grad(), nnops.use_weights(holder):
        out = nnops.head_out(x.to(DEV), holder[0])
    assert out.dtype == torch.float32 and rel_err(C(out), ref.detach().numpy()) < 1e-3


# ------------------------------------------------------------------------------------------------ f1 / f4 / cfg 4
def test_coco_evaluator_update_vs_reference(golden):
    """COCOEvaluator.update fed with DEVICE tensors (records + instance scores from pk_pose_records) against the prediction dicts the
    reference's evaluator built from the same arrays; the reference's OKS matching on top gives the same precision numbers."""
    from infantposeestimation_gaussianbias_amd.utils import COCOEvaluator
    z, meta = golden("extra_r02.npz"), golden("meta.json")["extra"]["eval"]
    ev = COCOEvaluator(ann_file=None, num_keypoints=17)
    ev.update(G(z["eval_pk"]), G(z["eval_ps"]), torch.tensor(meta["image_ids"]), meta["ann_ids"], G(z["eval_centers"]), G(z["eval_scales"]),
              G(z["eval_areas"]), G(z["eval_bboxes"]))
    ev2 = COCOEvaluator(ann_file=None, num_keypoints=17)      # the reference's calling convention: numpy arrays
    ev2.update(z["eval_pk"], z["eval_ps"], meta["image_ids"], meta["ann_ids"], z["eval_centers"], z["eval_scales"], z["eval_areas"], z["eval_bboxes"])
    for got in (ev.predictions, ev2.predictions):
        assert len(got) == len(meta["predictions"])
        for a, b in zip(got, meta["predictions"]):
            assert a["image_id"] == b["image_id"] and a["ann_id"] == b["ann_id"] and a["keypoints"] == b["keypoints"] and a["bbox"] == b["bbox"]
            assert a["area"] == b["area"] and abs(a["score"] - b["score"]) <= 1e-6 * max(1.0, abs(b["score"]))   # fp32 sum order vs numpy's pairwise mean
    m = ev.evaluate(gt_annotations=meta["gts"])
    assert all(abs(float(m[k]) - meta["metrics"][k]) < 1e-9 for k in ("AP", "AP50", "AP75")), (m, meta["metrics"])
    with pytest.raises(ValueError):
        ev.evaluate()


def test_optimizer_state_interchange_with_reference(golden):
    """f4: an AdamW state written by the reference (its build_optimizer grouping + per-iteration LambdaLR, two steps) loads into FlatAdamW;
    the third step from the reference's gradients lands on the reference's weights; FlatAdamW.state_dict() is accepted by
    torch.optim.AdamW built with the reference's two groups (train.py:55-97,339-368,426-435)."""
    from infantposeestimation_gaussianbias_amd import engine
    from infantposeestimation_gaussianbias_amd.models import hrformer
    z, meta = golden("extra_r02.npz"), golden("meta.json")["extra"]
    names = meta["optim_names"]
    mod = hrformer.HRFormerModule([16, 32], [1, 2], [1, 1], [4, 4], 0.0)
    assert [k for k, _ in mod.named_parameters()] == names
    sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(meta["optim_spec"], 21).items()}
    sd.update({k: torch.from_numpy(z["optim_w2." + k]) for k in names})
    sd.update({k[9:]: torch.from_numpy(z[k]) for k in z if k.startswith("optim_b2.")})
    mod.load_state_dict(sd)
    mod = mod.to(DEV)
    opt = engine.FlatAdamW(mod, lr=5e-4, weight_decay=0.01)
    groups = meta["optim_param_groups"]
    ref_sd = {"state": {}, "param_groups": groups}
    n_state = sum(1 for k in z if k.startswith("optim_state.") and k.endswith(".step"))
    for idx in range(n_state):
        ref_sd["state"][idx] = {"step": torch.tensor(float(z[f"optim_state.{idx}.step"])), "exp_avg": torch.from_numpy(z[f"optim_state.{idx}.exp_avg"]),
                                "exp_avg_sq": torch.from_numpy(z[f"optim_state.{idx}.exp_avg_sq"])}
    for k, p in mod.named_parameters():
        p.grad = G(z["optim_grad3." + k])
    opt.install_grad_views()
    opt.load_state_dict(ref_sd)
    assert opt.step_count == 2 and abs(opt.lr - groups[0]["lr"]) < 1e-12
    opt.step()
    torch.cuda.synchronize()
    for k, p in mod.named_parameters():
        assert rel_err(C(p), z["optim_w3." + k]) < 2e-6, k
    # and back: torch's AdamW with the reference's grouping takes our state
    mine = opt.state_dict()
    assert [len(g["params"]) for g in mine["param_groups"]] == [len(g["params"]) for g in groups]
    assert [g["weight_decay"] for g in mine["param_groups"]] == [g["weight_decay"] for g in groups]
    cpu = hrformer.HRFormerModule([16, 32], [1, 2], [1, 1], [4, 4], 0.0)
    decay = [p for n, p in cpu.named_parameters() if not engine.is_no_decay(n)]
    nodecay = [p for n, p in cpu.named_parameters() if engine.is_no_decay(n)]
    t_opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.01}, {"params": nodecay, "weight_decay": 0.0}], lr=5e-4)
    t_opt.load_state_dict({"state": {i: {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in st.items()} for i, st in mine["state"].items()},
                           "param_groups": mine["param_groups"]})
    assert len(t_opt.state_dict()["state"]) == len(mine["state"]) and float(t_opt.state_dict()["state"][0]["step"]) == 3.0


def test_hrnet_w32_train_step_vs_golden(golden):
    """BASELINE cfg 4 at fixture size: HRNet-W32 + heatmap head + KeypointMSELoss, train mode, B=2 128x96.  Against the reference's fp32
    numbers: loss 3e-2, the set of grad-less parameters.  Against the bf16-aware oracle, calibrated by its own sensitivity (train-mode
    BatchNorm over 2x4x3 samples on the last branch makes this fixture chaotic under bf16 storage): heatmaps and every parameter gradient."""
    from infantposeestimation_gaussianbias_amd.models import PoseEstimator
    from oracle import losses as olos
    z, keys, meta = golden("extra_r02.npz"), golden("state_keys.json"), golden("meta.json")["extra"]
    m = _load(PoseEstimator("hrnet_w32", 17, False, "heatmap", True), keys["hrnet_w32_heatmap"], 42).to(DEV).train()
    x = G(synth_input("w32_train", (2, 3, 128, 96)))
    o = m(x, G(z["w32_train_tgt"]), G(z["w32_train_w"]))
    o["loss"].backward()
    assert math.isclose(float(o["loss"].detach()), float(z["w32_train_loss"]), rel_tol=3e-2)
    assert sorted(k for k, p in m.named_parameters() if p.grad is None) == sorted(meta["w32_train_nograd"])
    names = [k for k, _ in m.named_parameters()]
    tgt, w = torch.from_numpy(z["w32_train_tgt"]), torch.from_numpy(z["w32_train_w"])

    def oracle_run(xc):
        ref, P32, _ = _bf16_oracle(keys["hrnet_w32_heatmap"], 42, xc, True)
        loss = olos.keypoint_mse(ref["heatmaps"], tgt, w)
        return ref["heatmaps"].detach().numpy(), float(loss.detach()), dict(zip(names, torch.autograd.grad(loss, [P32[k] for k in names], allow_unused=True)))

    xc = x.cpu()
    hm_a, loss_a, g_a = oracle_run(xc)
    hm_b, loss_b, g_b = oracle_run(xc * (1 + 1e-6 * torch.randn(xc.shape, generator=torch.Generator().manual_seed(1))))
    keep = [k for k in names if g_a[k] is not None and float(g_a[k].norm()) > 1e-7]
    hip = {k: C(dict(m.named_parameters())[k].grad) for k in keep}
    floor_l2, hip_l2 = _quant(_l2(g_b[k].numpy(), g_a[k].numpy()) for k in keep), _quant(_l2(hip[k], g_a[k].numpy()) for k in keep)
    floor_hm, hip_hm = _l2(hm_b, hm_a), _l2(C(o["heatmaps"]), hm_a)
    print(f"W32 train: loss HIP {float(o['loss'].detach()):.5f} oracle {loss_a:.5f} / {loss_b:.5f}; heatmaps L2 HIP {hip_hm:.4f} self {floor_hm:.4f}; "
          f"gradient L2 (median, p90, p99) HIP {hip_l2} self {floor_l2}")
    assert abs(float(o["loss"].detach()) - loss_a) <= 3 * abs(loss_b - loss_a) + 1e-2 * abs(loss_a)
    assert hip_hm <= 1.5 * floor_hm + 5e-3
    assert np.all(hip_l2 <= 1.5 * floor_l2 + 1e-2), (hip_l2, floor_l2)


def test_hrnet_w32_full_size_training_properties():
    """BASELINE cfg 4 at full size (384x288 -> 96x72, B = 16): two optimiser steps are finite, deterministic run to run, reduce the loss,
    and leave exactly the structurally unused parameters without a gradient."""
    from infantposeestimation_gaussianbias_amd import engine
    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.datasets import synthetic_batch
    from infantposeestimation_gaussianbias_amd.models import build_model
    cfg = get_config("hrnet_w32")
    assert cfg.data.input_size == (288, 384) and cfg.data.heatmap_size == (72, 96)
    batch = synthetic_batch(16, cfg.data.input_size, cfg.data.heatmap_size, 17, 2.0, DEV, seed=4)
    runs = []
    for _ in range(2):
        torch.manual_seed(0)
        model = build_model(cfg).to(DEV)
        tr = engine.Trainer(model, cfg, iters_per_epoch=2)
        losses = [float(tr.step(batch)["loss"]) for _ in range(3)]
        runs.append((losses, tr.opt.flat.detach().clone()))
        assert tr.step(batch)["heatmaps"].shape == (16, 17, 96, 72)
        n_dead = sum(1 for a in tr.opt.active if not a)
        assert 0 < n_dead < 60
    assert np.all(np.isfinite(runs[0][0])) and runs[0][0][-1] < runs[0][0][0]
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])


def test_fusion_loss_without_target_weight_vs_golden(golden):
    """FusionPoseLoss(use_target_weight=False) on the fused kernels (forward + hand-derived backward) against the reference's numbers."""
    from infantposeestimation_gaussianbias_amd.models.fusion_head import FusionPoseLoss
    z = golden("loss_utw_r02.npz")
    hm, off, var = (G(z[k]).requires_grad_(True) for k in ("hm", "off", "var"))
    out = FusionPoseLoss(use_target_weight=False).to(DEV)({"heatmaps": hm, "offsets": off, "variances": var}, G(z["tgt"]), G(z["w"]), G(z["gt"]),
                                                         (96, 128), (24, 32))
    got = np.array([float(out[n].detach()) for n in FusionPoseLoss.NAMES])
    assert np.allclose(got, z["losses"], rtol=1e-4, atol=1e-6), (got, z["losses"])
    out["total_loss"].backward()
    for t, k in ((hm, "g_hm"), (off, "g_off"), (var, "g_var")):
        assert rel_err(C(t.grad), z[k]) < 2e-4, k


def test_hrformer_without_relative_position_bias_vs_golden(golden):
    """with_rpe=False (hrformer.py:145-191): block forward + input / qkv-weight gradients (the unfused C = 64 training path and the fused
    forward) and the whole small-width backbone in eval mode against the vectors captured from the reference."""
    from infantposeestimation_gaussianbias_amd.models import hrformer
    from recipe import synth_input, synth_state_dict
    z, specs = golden("norpe_r03.npz"), golden("norpe_r03.json")
    blk = hrformer.HRFormerBlock(64, 2, 4.0, 0.0, with_rpe=False)
    blk.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(specs["block_spec"], 48).items()}, strict=True)
    blk = blk.to(DEV).train()
    x = G(synth_input("norpe_blk", (2, 64, 9, 10))).requires_grad_(True)
    y = blk(x)
    assert rel_err(C(y), z["blk_out"]) < 2e-2
    y.float().backward(G(synth_input("norpe_blk_gy", tuple(y.shape))))
    assert rel_err(C(x.grad), z["blk_gx"]) < 3e-2 and rel_err(C(blk.attn.qkv.weight.grad), z["blk_gqkv"]) < 3e-2
    with torch.no_grad():
        assert rel_err(C(blk.eval()(x.detach())), z["blk_out"]) < 2e-2                  # fused forward kernels (eval)
    bb = hrformer.HRFormer(with_rpe=False, in_channels=3, drop_path_rate=0.0, stage2_num_channels=(32, 64), stage2_num_heads=(1, 2),
                           stage3_num_channels=(32, 64, 128), stage3_num_heads=(1, 2, 4), stage4_num_channels=(32, 64, 128, 256),
                           stage4_num_heads=(1, 2, 4, 8))
    bb.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(specs["backbone_spec"], 47).items()}, strict=True)
    bb = bb.to(DEV).eval()
    with torch.no_grad():
        out = bb(G(synth_input("norpe_bb", (1, 3, 64, 64))))
    assert rel_err(C(out), z["bb_out"]) < 3e-2


def test_graph_replay_without_relative_position_bias_matches_eager():
    """HRFormer(with_rpe=False) through the captured hipGraph step: the fused attention backward must not upload a reduction
    table for the (constant, all-zero) bias table during capture; the replayed trajectory follows the eager one."""
    from infantposeestimation_gaussianbias_amd import engine
    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.datasets import synthetic_batch
    from infantposeestimation_gaussianbias_amd.models import PoseEstimator, hrformer
    cfg = get_config("hrformer_small")
    cfg.data.input_size, cfg.data.heatmap_size = (96, 128), (24, 32)
    batches = [synthetic_batch(4, cfg.data.input_size, cfg.data.heatmap_size, 17, 2.0, DEV, seed=60 + i) for i in range(3)]
    traj = {}
    for mode in (False, True):
        torch.manual_seed(0)
        bb = hrformer.HRFormer(with_rpe=False, in_channels=3, drop_path_rate=0.0, stage2_num_channels=(32, 64), stage2_num_heads=(1, 2),
                               stage3_num_channels=(32, 64, 128), stage3_num_heads=(1, 2, 4), stage4_num_channels=(32, 64, 128, 256),
                               stage4_num_heads=(1, 2, 4, 8))
        model = PoseEstimator.from_backbone(bb, 32, 17, "fusion", True).to(DEV)
        assert not any("relative_position" in k for k in model.state_dict())
        # (graph_streams=True in both runs: the eager run then has the autograd structure of the captured one -- one node per fork / join
        # region, shared-input gradients summed in one launch -- so the two trajectories differ by launch mechanics only)
        tr = engine.Trainer(model, cfg, iters_per_epoch=2, use_graph=mode, graph_warmup=2, graph_streams=True)
        traj[mode] = [float(tr.step(batches[i % 3])["loss"].detach()) for i in range(5)]
        assert (tr._graph is not None) == mode
    assert np.allclose(traj[False], traj[True], rtol=2e-3), traj


def _expected_gradient(golden, backbone, K_kp, head, spec_key, salt, size, hm_size, runs, min_keep):
    """The expected-gradient comparison of test_hrformer_small_expected_gradient_vs_bf16_aware_oracle for any model the oracle runs:
    mean parameter gradient over `runs` dithered inputs on the HIP path (for C % 8 != 0 models: through the 8-aligned padded twin, gradients
    extracted back into the real parameters) against the mean over `runs` differently dithered inputs on the bf16-aware oracle, BatchNorm on
    running statistics.  Every tensor within 3 % (or 2x the oracle's own half-vs-half distance), cosine >= 0.999."""
    from infantposeestimation_gaussianbias_amd.models import PoseEstimator
    from oracle import losses as olos, nets as onet, train_step as ots
    keys = golden("state_keys.json")[spec_key]
    m = _load(PoseEstimator(backbone, K_kp, False, head, True), keys, salt).to(DEV).eval()
    img, tg, tw, kp = ots.synthetic_batch(2, size, hm_size, K_kp, 2.0, seed=78)
    names = [k for k, _ in m.named_parameters()]
    dither = lambda i: img * (1 + 1e-5 * torch.randn(img.shape, generator=torch.Generator().manual_seed(300 + i)))

    def oracle_run(x):
        P32 = {k: torch.from_numpy(v).clone() for k, v in synth_state_dict(keys, salt).items()}
        for k in names:
            P32[k].requires_grad_(True)
        ref = onet.pose_forward(x, onet.bf16_weights(P32), onet.Ctx(train=False, q=onet.bf16_storage))
        if head == "fusion":
            loss = olos.fusion_pose_loss(ref["heatmaps"], ref["offsets"], ref["variances"], tg, tw, kp, size)["total_loss"]
        else:
            loss = olos.keypoint_mse(ref["heatmaps"], tg, tw)
        gr = torch.autograd.grad(loss, [P32[k] for k in names], allow_unused=True)
        return {k: (None if g is None else g.double()) for k, g in zip(names, gr)}, float(loss)

    def hip_run(x):
        m.zero_grad(set_to_none=True)
        o = m(x.to(DEV), tg.to(DEV), tw.to(DEV), kp.to(DEV), input_size=size) if head == "fusion" else m(x.to(DEV), tg.to(DEV), tw.to(DEV))
        o["loss"].backward()
        return {k: (None if p.grad is None else p.grad.detach().double().cpu()) for k, p in m.named_parameters()}, float(o["loss"])

    def mean_of(rs):
        acc, losses = None, []
        for g, l in rs:
            losses.append(l)
            if acc is None:
                acc = {k: (None if v is None else v.clone()) for k, v in g.items()}
            else:
                for k, v in g.items():
                    if v is not None:
                        acc[k] += v
        return {k: (None if v is None else v / len(losses)) for k, v in acc.items()}, float(np.mean(losses))

    half_a, loss_a = mean_of(oracle_run(dither(i)) for i in range(0, runs, 2))
    half_b, loss_b = mean_of(oracle_run(dither(i)) for i in range(1, runs, 2))
    hip, loss_h = mean_of(hip_run(dither(runs + i)) for i in range(runs))
    for k in names:
        assert (hip[k] is None) == (half_a[k] is None), k                      # the same grad-less set
    orc = {k: (None if half_a[k] is None else 0.5 * (half_a[k] + half_b[k])) for k in names}
    keep = [k for k in names if orc[k] is not None and not k.endswith(_ZERO_GRAD) and float(orc[k].norm()) > 1e-7]
    assert len(keep) >= min_keep, len(keep)
    bad, rows = [], []
    for k in keep:
        d, c = _l2(hip[k].numpy(), orc[k].numpy()), _cos(hip[k].numpy(), orc[k].numpy())
        d0, c0 = _l2(half_a[k].numpy(), half_b[k].numpy()), _cos(half_a[k].numpy(), half_b[k].numpy())
        rows.append((d, d0, c, k))
        if d > max(3e-2, 2.0 * d0) or c < min(0.999, c0 - 5e-4):
            bad.append((k, round(d, 4), round(d0, 4), round(c, 5), round(c0, 5)))
    dist = np.array(sorted(r[0] for r in rows))
    self_d = np.array(sorted(r[1] for r in rows))
    print(f"{backbone}: expected-gradient distance HIP vs oracle over {len(keep)} tensors (median / p90 / p99 / max): "
          f"{dist[len(dist) // 2]:.4f} {dist[int(len(dist) * .9)]:.4f} {dist[int(len(dist) * .99)]:.4f} {dist[-1]:.4f}; oracle half-vs-half: "
          f"{self_d[len(dist) // 2]:.4f} {self_d[int(len(dist) * .9)]:.4f} {self_d[int(len(dist) * .99)]:.4f} {self_d[-1]:.4f}; losses {loss_h:.4f} / "
          f"{0.5 * (loss_a + loss_b):.4f}")
    assert not bad, (len(bad), bad[:12])
    assert abs(loss_h - 0.5 * (loss_a + loss_b)) <= 3e-3 * abs(loss_a) + 3 * abs(loss_a - loss_b)


def test_hrformer_base_twin_expected_gradient_vs_bf16_aware_oracle(golden):
    """the padded twin (C = 78 -> 80, head_dim 39 -> 40) held to the expected-gradient bar instead of `l2 < 0.45`:
    HRFormer-base + fusion head, K = 13, 128x96, 16 runs per side."""
    _expected_gradient(golden, "hrformer_base", 13, "fusion", "hrformer_base_fusion_k13", 44, (96, 128), (24, 32), 16, 700)


def test_hrnet_w18_twin_expected_gradient_vs_bf16_aware_oracle(golden):
    """The same for HRNet-W18 + heatmap head (C = 18 -> 24 ...), BASELINE cfg 1's model, 128x96, 32 runs per side (0.5 s per oracle run; at 16
    runs one BatchNorm scale of a three-step down chain sat at 5.4 % against 2 x 2.4 % of its own noise)."""
    _expected_gradient(golden, "hrnet_w18", 17, "heatmap", "hrnet_w18_heatmap", 41, (96, 128), (24, 32), 32, 800)


def test_cfg5_full_size_flip_inference_graph_replay_matches_two_pass_eager(golden, monkeypatch, tmp_path):
    """BASELINE cfg 5 at its FULL size: HRFormer-base + fusion head at 384x288 -> 96x72, the infant key-point set and sigma
    read from a preemie_optimized-style yaml through get_config(path) (K = 13, sigma = 1.5), flip-test inference.  The served path (both flip
    passes as one batch, branch streams, hipGraph replay -- what bench.py --config hrformer_base_infer times, with the wide fused halves at
    their production launch sizes) against the plain two-pass single-stream eager inference: key points within 0.05 heat-map px, scores
    within 1e-2, on the captured input and on a second one; the T1 targets of that configuration keep the asymmetric sigma = 1.5 patch."""
    from infantposeestimation_gaussianbias_amd import dispatch, hipops
    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.models import PoseEstimator
    from oracle import target as otgt
    p = tmp_path / "preemie_optimized.yaml"
    p.write_text("MODEL:\n  NAME: 'pose_hrnet_w32'\n  NUM_JOINTS: 13\n  IMAGE_SIZE: [256, 256]\n  HEATMAP_SIZE: [128, 128]\n  SIGMA: 1.5\n"
                 "TRAIN:\n  BATCH_SIZE: 24\n  LR: 0.0005\n")
    ycfg = get_config(str(p))
    K, sigma = ycfg.data.num_keypoints, ycfg.data.sigma
    assert (K, sigma) == (13, 1.5)
    size, hm = (288, 384), (72, 96)
    B = 8
    kp = torch.rand(B, K, 2, generator=torch.Generator().manual_seed(5)) * torch.tensor([288.0, 384.0])
    vis = torch.randint(0, 3, (B, K), generator=torch.Generator().manual_seed(6)).float()
    t, w = hipops.gaussian_target(kp.to(DEV), vis.to(DEV), size, hm, sigma)
    ot, ow = otgt.generate_target_batch(kp.numpy(), vis.numpy(), size, hm, sigma)
    assert np.array_equal(C(t).view(np.uint32), ot.view(np.uint32)) and np.array_equal(C(w), ow)
    keys = golden("state_keys.json")
    m = _load(PoseEstimator("hrformer_base", K, False, "fusion", True), keys["hrformer_base_fusion_k13"], 44).to(DEV).eval()
    pairs = [(1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12)]
    xs = [G(synth_input("cfg5_a", (B, 3, 384, 288))), G(synth_input("cfg5_b", (B, 3, 384, 288)))]
    ref = []
    monkeypatch.setenv("POSE_FLIP_BATCHED", "0")
    dispatch.set_streams(False)
    try:
        with torch.no_grad():
            for x in xs:
                kpr, scr = m.inference(x, flip=True, flip_pairs=pairs)
                ref.append((C(kpr), C(scr)))
        assert ref[0][0].shape == (B, K, 2) and np.isfinite(ref[0][0]).all() and np.isfinite(ref[0][1]).all()
        monkeypatch.setenv("POSE_FLIP_BATCHED", "1")
        dispatch.set_streams(True)
        static = xs[0].clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            m.inference(static, flip=True, flip_pairs=pairs)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                out = m.inference(static, flip=True, flip_pairs=pairs)
        torch.cuda.current_stream().wait_stream(s)
        for x, (kp_r, sc_r) in zip(xs, ref):
            static.copy_(x)
            g.replay()
            torch.cuda.synchronize()
            assert np.abs(C(out[0]) - kp_r).max() < 0.05 and rel_err(C(out[1]), sc_r) < 1e-2
    finally:
        dispatch.set_streams(True)


def test_served_inference_is_bit_reproducible_with_branch_streams(golden):
    """The served cfg-5 path (both flip passes in one batch, branch streams, hipGraph replay) must return the SAME bits from replay to replay
    and the bits of the single-stream pass, whatever ran before it.  Round 4 found it did not: the LayerNorm sums of `k_attn_fwd_w` taken
    with `v_permlane16_swap` came out different in a few lanes of ~6 % of the replays (one bf16 ulp in a window's output, up to 0.4 px in
    a key point after the soft-argmax) whenever other streams' kernels shared the chip -- and only then, so no single-stream test saw it.
    The xor-16 butterfly steps now go through `ds_swizzle_b32` (pk_common.h): 0 of 150 replays differ (scripts/probes/cfg5_poison.py).
    Two inputs alternate so that a kernel reading a stale buffer of the previous replay would show as well."""
    from infantposeestimation_gaussianbias_amd import dispatch
    from infantposeestimation_gaussianbias_amd.models import PoseEstimator
    K, B = 13, 8
    keys = golden("state_keys.json")
    m = _load(PoseEstimator("hrformer_base", K, False, "fusion", True), keys["hrformer_base_fusion_k13"], 44).to(DEV).eval()
    pairs = [(1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12)]
    xs = [G(synth_input("cfg5_a", (B, 3, 384, 288))), G(synth_input("cfg5_b", (B, 3, 384, 288)))]
    dispatch.set_streams(False)
    try:
        with torch.no_grad():
            ref = [tuple(C(t) for t in m.inference(x, flip=True, flip_pairs=pairs)) for x in xs]
        dispatch.set_streams(True)
        static = xs[0].clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            m.inference(static, flip=True, flip_pairs=pairs)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                out = m.inference(static, flip=True, flip_pairs=pairs)
        torch.cuda.current_stream().wait_stream(s)
        bad = []
        for rep in range(24):
            for k, x in enumerate(xs):
                static.copy_(x)
                g.replay()
                torch.cuda.synchronize()
                if not (np.array_equal(C(out[0]), ref[k][0]) and np.array_equal(C(out[1]), ref[k][1])):
                    bad.append((rep, k, float(np.abs(C(out[0]) - ref[k][0]).max())))
        assert not bad, f"replays that differ from the single-stream result (replay, input, max |d key point|): {bad[:6]}"
    finally:
        dispatch.set_streams(True)


def _rccl_one_rank_worker(port, q):
    import os
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", POSE_GRAPH_COMM="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from infantposeestimation_gaussianbias_amd import engine
    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.datasets import synthetic_batch
    from infantposeestimation_gaussianbias_amd.models import build_model
    # (a) the bucketed exchange itself over RCCL: one rank, so the sum is the identity -- eager, with the exposed-time events
    cfg = get_config("hrformer_small")
    cfg.data.input_size, cfg.data.heatmap_size = (96, 128), (24, 32)
    shard = synthetic_batch(2, cfg.data.input_size, cfg.data.heatmap_size, 17, 2.0, "cuda", seed=50)
    torch.manual_seed(0)
    model = build_model(cfg).to("cuda")
    model.backbone.drop_path_rate = 0.0
    tr = engine.Trainer(model, cfg, iters_per_epoch=2, use_graph=True, graph_warmup=2, graph_streams=True, bucket_mb=4.0)
    tr.comm.world = 2                       # drive the N > 1 code paths (milestones, buckets, capture of the collectives) on ONE rank:
    tr._ms_cb = tr._milestone               # RCCL sums over the single rank, the optimiser halves the gradient (grad_scale = 1 / world)
    tr.comm.record_exposed = True
    losses = [float(tr.step(shard)["loss"].detach()) for _ in range(6)]
    torch.cuda.synchronize()
    q.put((np.asarray(losses), tr._graph is not None, bool(tr._graph_has_comm), bool(tr._graph_has_opt), len(tr.comm.buckets), tr.comm.launched_early,
           tr.comm.exposed_ms()))
    dist.destroy_process_group()


def test_rccl_gradient_exchange_and_graph_capture_on_one_rank():
    """RCCL itself has never run in this project's records (no multi-GPU box): this drives the N > 1 machinery -- bucketed all-reduces over
    the `nccl` backend issued from backward milestones, then the whole step INCLUDING the collectives captured into one hipGraph
    (POSE_GRAPH_COMM=1, thread-local capture mode) and replayed -- on a ONE-rank communicator.  It proves the call sequence, the capture and
    the replay against RCCL on this stack; it cannot prove cross-rank correctness (the gloo world-2 tests do that part)."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_one_rank_worker, args=(port, q))
    p.start()
    try:
        losses, has_graph, has_comm, has_opt, n_buckets, early, exposed = q.get(timeout=240)
    finally:
        p.join(30)
        if p.is_alive():
            p.kill()
    assert p.exitcode == 0
    print(f"one-rank RCCL: graph {has_graph}, collectives captured {has_comm}, optimiser captured {has_opt}, {n_buckets} buckets, "
          f"{early} launched from milestones in the last eager step, exposed {exposed}")
    assert has_graph and n_buckets >= 4 and np.all(np.isfinite(losses)) and losses[-1] != losses[0]
