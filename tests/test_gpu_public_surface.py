"""GPU parity of the reference's PUBLIC helper surface that is not on the fused training path (`-m gpu`; VERDICT r03 "missing" #2 / #6):
GaussianDistributionConstraint's methods on arbitrary coordinates, FusionPoseLoss.heatmap_loss / offset_loss / peak_localization_loss,
LocalGaussianRefinement, SoftArgmax2D with gradients (models/fusion_head.py:24-128,405-575,637-743), window_partition / window_reverse /
DropPath / drop_path (models/hrformer.py:15-35,67-114) and the HRNet-W48 builder.  Every value and gradient is compared with vectors captured
from the reference itself by tests/golden/make_golden_r04.py (fp32 kernels: 1e-4; row moves: bit-exact)."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from recipe import synth_input, synth_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


def G(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV, torch.float32)
    return t.requires_grad_(True) if grad else t


def C(t):
    return t.detach().float().cpu().numpy()


def close(a, b, tol=TOL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max()) <= tol * max(1.0, float(np.abs(b).max()))


DIMS = {"k17": (48, 64), "k13": (64, 48)}


@pytest.mark.parametrize("tag", ["k17", "k13"])
def test_gaussian_constraint_methods_vs_reference(golden, tag):
    from infantposeestimation_gaussianbias_amd.models.fusion_head import GaussianDistributionConstraint
    z = golden("public_r04.npz")
    gc = GaussianDistributionConstraint(target_sigma=2.0, overlap_threshold=0.4)
    assert gc.SKELETON[0] == (0, 1) and len(gc.SKELETON) == 16
    w = G(z[f"{tag}.w"])
    h, c = G(z[f"{tag}.hm"], True), G(z[f"{tag}.coords"], True)
    sig = gc.compute_heatmap_variance(h, c)
    assert close(C(sig), z[f"{tag}.sigma"])
    sig.backward(G(z[f"{tag}.sigma_g"]))
    assert close(C(h.grad), z[f"{tag}.sigma_dhm"]) and close(C(c.grad), z[f"{tag}.sigma_dc"])
    for name, fn in (("t_var_nopred", lambda h, c, v: gc.variance_alignment_loss(h, c, w, None)),
                     ("t_var", lambda h, c, v: gc.variance_alignment_loss(h, c, w, v)),
                     ("t_ovl", lambda h, c, v: gc.spatial_overlap_loss(h, w)),
                     ("t_shape", lambda h, c, v: gc.distribution_shape_loss(h, w))):
        h, c, v = G(z[f"{tag}.hm"], True), G(z[f"{tag}.coords"], True), G(z[f"{tag}.var"], True)
        val = fn(h, c, v)
        assert val.dim() == 0 and close(C(val), z[f"{tag}.{name}"]), (name, float(val), float(z[f"{tag}.{name}"]))
        val.backward()
        assert close(C(h.grad), z[f"{tag}.{name}_dhm"]), name
        if f"{tag}.{name}_dc" in z:
            assert close(C(c.grad), z[f"{tag}.{name}_dc"]), name
        if f"{tag}.{name}_dvar" in z:
            assert close(C(v.grad), z[f"{tag}.{name}_dvar"]), name
    h, c, v = G(z[f"{tag}.hm"], True), G(z[f"{tag}.coords"], True), G(z[f"{tag}.var"], True)
    d = gc(h, c, w, v)
    assert sorted(d) == ["overlap_loss", "shape_loss", "variance_loss"]
    got = np.array([float(d["variance_loss"]), float(d["overlap_loss"]), float(d["shape_loss"])])
    assert close(got, z[f"{tag}.fwd3"])
    (1.0 * d["variance_loss"] + 0.7 * d["overlap_loss"] + 0.3 * d["shape_loss"]).backward()
    assert close(C(h.grad), z[f"{tag}.fwd3_dhm"]) and close(C(c.grad), z[f"{tag}.fwd3_dc"]) and close(C(v.grad), z[f"{tag}.fwd3_dvar"])


@pytest.mark.parametrize("tag", ["k17", "k13"])
@pytest.mark.parametrize("utw", [True, False])
def test_fusion_loss_term_methods_vs_reference(golden, tag, utw):
    from infantposeestimation_gaussianbias_amd.models.fusion_head import FusionPoseLoss
    z = golden("public_r04.npz")
    fl = FusionPoseLoss(use_target_weight=utw).to(DEV)
    u = "w" if utw else "nw"
    win, hin = DIMS[tag]
    H, W = z[f"{tag}.hm"].shape[2:]
    w, gt = G(z[f"{tag}.w"]), G(z[f"{tag}.gt"])
    h = G(z[f"{tag}.hm"], True)
    val = fl.heatmap_loss(h, G(z[f"{tag}.tgt"]), w)
    val.backward()
    assert close(C(val), z[f"{tag}.{u}.hm_loss"]) and close(C(h.grad), z[f"{tag}.{u}.hm_loss_dhm"])
    o, c = G(z[f"{tag}.off"], True), G(z[f"{tag}.coords"], True)
    val = fl.offset_loss(o, c, gt, w, (win, hin), (H, W))
    val.backward()
    assert close(C(val), z[f"{tag}.{u}.off_loss"])
    assert close(C(o.grad), z[f"{tag}.{u}.off_loss_doff"]) and close(C(c.grad), z[f"{tag}.{u}.off_loss_dc"])
    c = G(z[f"{tag}.coords"], True)
    val = fl.peak_localization_loss(c, gt, w, (win, hin), (H, W))
    val.backward()
    assert close(C(val), z[f"{tag}.{u}.peak_loss"]) and close(C(c.grad), z[f"{tag}.{u}.peak_loss_dc"])


@pytest.mark.parametrize("tag", ["k17", "k13"])
def test_softargmax_gradients_and_local_refinement_vs_reference(golden, tag):
    from infantposeestimation_gaussianbias_amd.models.fusion_head import LocalGaussianRefinement, SoftArgmax2D, SubPixelRefinement
    z = golden("public_r04.npz")
    h = G(z[f"{tag}.hm"], True)
    co, sc = SoftArgmax2D()(h)
    assert close(C(co), z[f"{tag}.sa_coords"]) and np.array_equal(C(sc), z[f"{tag}.sa_scores"])
    ((co * G(z[f"{tag}.sa_gco"])).sum() + (sc * G(z[f"{tag}.sa_gsc"])).sum()).backward()
    assert close(C(h.grad), z[f"{tag}.sa_dhm"])
    hm, coords = G(z[f"{tag}.hm"]), G(z[f"{tag}.coords"])
    for r in (1, 2):
        assert close(C(LocalGaussianRefinement(local_radius=r)(hm, coords)), z[f"{tag}.local_r{r}"], 1e-5)
    assert close(C(LocalGaussianRefinement(2)(hm, G(z[f"{tag}.edge_coords"]))), z[f"{tag}.local_edge"], 1e-5)
    sp = SubPixelRefinement()
    assert isinstance(sp.soft_argmax, SoftArgmax2D) and isinstance(sp.local_refine, LocalGaussianRefinement)
    assert list(sp.state_dict()) == ["alpha"]                   # the reference's checkpoint keys (fusion_head.py:147-149)


@pytest.mark.parametrize("tag", ["wa", "wb", "wc"])
def test_window_partition_reverse_vs_reference(golden, tag):
    from infantposeestimation_gaussianbias_amd.models import hrformer
    z = golden("public_r04.npz")
    x = G(z[f"{tag}.x"], True)
    B, H, W, Cc = x.shape
    wins, (Hp, Wp) = hrformer.window_partition(x, 7)
    assert (Hp, Wp) == tuple(int(v) for v in z[f"{tag}.hpwp"])
    assert np.array_equal(C(wins), z[f"{tag}.windows"])
    back = hrformer.window_reverse(wins, 7, H, W, Hp, Wp)
    assert np.array_equal(C(back), z[f"{tag}.back"]) and np.array_equal(C(back), z[f"{tag}.x"])
    back.backward(torch.ones_like(back))                        # partition then reverse is the identity on the real pixels
    assert np.array_equal(C(x.grad), np.ones_like(z[f"{tag}.x"]))
    assert np.array_equal(C(hrformer.window_reverse(G(z[f"{tag}.rev_in"]), 7, H, W, Hp, Wp)), z[f"{tag}.rev_out"])
    xb = x.detach().to(torch.bfloat16)                          # the element type the network uses
    assert torch.equal(hrformer.window_reverse(hrformer.window_partition(xb, 7)[0], 7, H, W, Hp, Wp), xb)


def test_drop_path_semantics():
    from infantposeestimation_gaussianbias_amd.models import hrformer
    x = G(synth_input("dp", (64, 5, 6, 8)), True)
    assert hrformer.drop_path(x, 0.3, False) is x and hrformer.drop_path(x, 0.0, True) is x
    m = hrformer.DropPath(0.25)
    assert m.eval()(x) is x
    torch.manual_seed(3)
    y = m.train()(x)
    torch.manual_seed(3)
    mask = torch.floor(0.75 + torch.rand(64, dtype=torch.float32, device=DEV))
    # hrformer.py:19-23, same arithmetic order: IEEE division, then the mask (numpy: torch's device kernel multiplies by 1 / keep instead)
    mk = C(mask).reshape(64, 1, 1, 1)
    assert np.array_equal(C(y), (C(x) / np.float32(0.75)) * mk) and 0 < int(mask.sum()) < 64
    y.backward(torch.ones_like(y))
    assert np.array_equal(C(x.grad), (np.ones_like(C(x)) / np.float32(0.75)) * mk)


def test_hrnet_w48_eval_vs_golden(golden):
    from infantposeestimation_gaussianbias_amd.models import PoseEstimator
    z, keys = golden("public_r04.npz"), golden("state_keys.json")
    m = PoseEstimator("hrnet_w48", 17, False, "heatmap", True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(keys["hrnet_w48_heatmap"], 44).items()}, strict=True)
    m = m.to(DEV).eval()
    with torch.no_grad():
        x = G(synth_input("w48_eval", (1, 3, 128, 96)))
        o = m(x)
        kp, sc = m.inference(x, flip=False)
    assert rel_err(C(o["heatmaps"]), z["w48_eval_hm"]) < 3e-2
    assert kp.shape == (1, 17, 2) and rel_err(C(sc), z["w48_eval_sc"]) < 3e-2       # (arg-max positions of a random-weight net are not stable under bf16)


def test_heatmap_head_deconv_stack_vs_reference(golden):
    """HeatmapHead(num_deconv_layers=3, kernels 4 / 2 / 4) (pose_estimator.py:22-99; VERDICT r03 "missing" #5): train-mode forward, input and
    parameter gradients, running statistics and the eval-mode forward against the reference's own vectors; state_dict keys as the reference's
    nn.Sequential gives them.  bf16 activations through three deconv + BN layers: forward norm-wise 3e-2, gradients relative L2 0.15 (a wrong
    tap or class mapping gives > 1; measured 0.1 on the input gradient)."""
    from infantposeestimation_gaussianbias_amd.models.pose_estimator import HeatmapHead
    z, spec = golden("deconv_r04.npz"), golden("deconv_r04.json")["spec"]
    head = HeatmapHead(32, 17, num_deconv_layers=3, num_deconv_filters=(48, 32, 24), num_deconv_kernels=(4, 2, 4))
    assert list(head.state_dict()) == list(spec)
    head.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(spec, 51).items()}, strict=True)
    head = head.to(DEV).train()
    x = G(synth_input("deconv_x", (2, 32, 6, 5)), True)
    y = head(x)
    assert tuple(y.shape) == (2, 17, 48, 40) and rel_err(C(y), z["y_train"]) < 3e-2
    y.backward(G(synth_input("deconv_gy", tuple(y.shape))))
    l2 = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-30))
    assert l2(C(x.grad), z["gx"]) < 0.15, l2(C(x.grad), z["gx"])       # bf16 through three train-mode BatchNorm layers over 240 .. 3840 samples
    for k, p in head.named_parameters():
        assert p.grad is not None and l2(C(p.grad), z["g." + k]) < 0.15, (k, l2(C(p.grad), z["g." + k]))
    sd = head.state_dict()
    for k in z:
        if k.startswith("buf."):
            assert rel_err(C(sd[k[4:]]), z[k]) < 2e-2, k
    with torch.no_grad():
        assert rel_err(C(head.eval()(x.detach())), z["y_eval"]) < 3e-2
    with pytest.raises(ValueError):
        HeatmapHead(32, 17, num_deconv_layers=1, num_deconv_filters=(32,), num_deconv_kernels=(3,))


def test_empty_batches_give_empty_results_like_the_reference():
    """A frame without detections: the reference's tensor code (utils/postprocess.py:10-34, 139-184, 241-303; T1 per sample) returns empty
    (0, K, ...) results for B = 0.  Ours must too -- a zero-size grid would be a launch error -- and with the reference's shapes."""
    from infantposeestimation_gaussianbias_amd import hipops
    from infantposeestimation_gaussianbias_amd.utils import postprocess as pp
    K, H, W = 17, 64, 48
    hm = torch.zeros(0, K, H, W, device=DEV)
    preds, mv = pp.get_max_preds(hm)
    assert tuple(preds.shape) == (0, K, 2) and tuple(mv.shape) == (0, K, 1) and preds.dtype == torch.float32
    preds, mv = pp.get_max_preds_with_subpixel(hm)
    assert tuple(preds.shape) == (0, K, 2) and tuple(mv.shape) == (0, K, 1)
    assert tuple(pp.coordinate_refinement(hm, preds).shape) == (0, K, 2)
    p2, mask = pp.filter_low_confidence(preds, mv)
    assert tuple(p2.shape) == (0, K, 2) and tuple(mask.shape) == (0, K, 1)
    p3, keep = pp.nms_pose(preds, mv)
    assert tuple(p3.shape) == (0, K, 2) and tuple(keep.shape) == (0, K, 1) and keep.dtype == torch.bool
    c, s = torch.zeros(0, 2, device=DEV), torch.zeros(0, 2, device=DEV)
    assert tuple(pp.transform_preds(preds, c, s, [640, 480]).shape) == (0, K, 2)
    assert tuple(pp.heatmap_to_image_coords(preds, c, s, (192, 256), (48, 64)).shape) == (0, K, 2)
    co, sc = hipops.softargmax_refine_decode(hm, None, torch.ones(1, device=DEV), None)
    assert tuple(co.shape) == (0, K, 2) and tuple(sc.shape) == (0, K)
    rec, inst = hipops.pose_records(preds, mv.squeeze(-1))
    assert tuple(rec.shape) == (0, K, 3) and tuple(inst.shape) == (0,)
    t, w = hipops.gaussian_target(torch.zeros(0, K, 2, device=DEV), torch.zeros(0, K, device=DEV), (192, 256), (48, 64), 2.0)
    assert tuple(t.shape) == (0, K, 64, 48) and tuple(w.shape) == (0, K, 1)
    out = hipops.flip_merge(hm, hm, torch.arange(K, dtype=torch.int32, device=DEV))
    assert tuple(out.shape) == (0, K, H, W)
    # one real sample next to it still decodes as before (the early return is only for the empty case)
    one = torch.zeros(1, K, H, W, device=DEV)
    one[0, :, 10, 7] = 1.0
    preds1, mv1 = pp.get_max_preds(one)
    assert torch.equal(preds1[0, 0].cpu(), torch.tensor([7.0, 10.0])) and float(mv1[0, 0, 0]) == 1.0
    torch.cuda.synchronize()
