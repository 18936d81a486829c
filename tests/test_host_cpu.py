"""CPU-only checks of the host side: the C-ABI library loads and exports every declared symbol, the drop-in surface
(config fields, state_dict keys, error behaviour), the optimiser bookkeeping, and the N>1 gradient exchange over gloo."""
import dataclasses
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from infantposeestimation_gaussianbias_amd import _lib
    decl = _lib.declared_symbols()
    assert len(decl) >= 16
    for name in decl:
        assert hasattr(_lib.lib, name), name
    assert _lib.lib.pk_version() >= 100
    # validation failure path: negative code + message, nothing launched (no GPU needed)
    rc = _lib.lib.pk_gaussian_target(None, None, None, 0, None, None, 1, 1, 1, 1, 1.0, 1.0, 1.0, 1, 0, None)
    assert rc == -1 and b"null pointer" in _lib.lib.pk_last_error_string()


_NULL_CALLS = r"""
import ctypes, json, sys
sys.path.insert(0, sys.argv[1])
from infantposeestimation_gaussianbias_amd import _lib
out = {}
for name, (ret, args) in sorted(_lib.declared_symbols().items()):
    print("calling", name, flush=True)              # the last line names the culprit if the process dies
    rc = getattr(_lib.lib, name)(*[None if a is ctypes.c_void_p else 0 for a in args])
    out[name] = rc.decode() if isinstance(rc, bytes) else rc
print("RESULT " + json.dumps(out))
"""


def test_every_entry_point_survives_null_and_zero_arguments():
    """Drop-in boundary hygiene: every one of the header's entry points called with NULL pointers and zero sizes -- in a child process, so
    that a fault names its function instead of taking the test run down.  Launchers must return an error status with a message (their
    argument checks run before anything touches the device, so this needs no GPU); size / capability queries must return a number
    (three of them divided by a zero tile count until round 4)."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, "-c", _NULL_CALLS, ROOT], capture_output=True, text=True, timeout=300)
    last = (r.stdout.strip().splitlines() or ["(no output)"])[-1]
    assert r.returncode == 0, f"child died with {r.returncode} after: {last}\n{r.stderr[-400:]}"
    res = json.loads(last[len("RESULT "):])
    from infantposeestimation_gaussianbias_amd import _lib
    assert set(res) == set(_lib.declared_symbols())
    queries = {n for n in res if n.endswith(("_blocks", "_supported", "_slices", "_rows", "_tiles", "_groups", "_ws_floats", "_slab_floats", "_hidden_slice",
                                             "_cols")) or n in ("pk_version", "pk_last_error_string", "pk_sizeof_group_desc")}
    for n, rc in res.items():
        if n in queries:
            assert isinstance(rc, (int, str)) and (not isinstance(rc, int) or rc >= 0), (n, rc)
        else:
            assert isinstance(rc, int) and rc != 0, f"{n} accepted NULL / zero arguments (status {rc})"


def test_wide_fused_half_rules_and_argument_checks():
    """The forward-only wide fused halves (pk_ln_mlp_wide_fwd / pk_attn_block_wide_fwd): which shapes the library is built for, its
    "does the launch pay" rule (enough workgroups / windows), and the argument validation -- all host-side, nothing is launched."""
    from infantposeestimation_gaussianbias_amd import _lib
    L = _lib.lib
    for C in (80, 128, 160, 256, 320):
        assert L.pk_ln_mlp_wide_supported(C, 4 * C, 0) == 1
    assert L.pk_ln_mlp_wide_supported(64, 256, 0) == 0 and L.pk_ln_mlp_wide_supported(640, 2560, 0) == 0      # narrow: pk_ln_mlp_fwd; 640: unfused
    assert L.pk_ln_mlp_wide_supported(80, 100, 0) == 0                                                           # hidden % 32
    assert L.pk_ln_mlp_wide_supported(80, 320, 64 * 96 * 72) == 1 and L.pk_ln_mlp_wide_supported(160, 640, 64 * 48 * 36) == 1
    assert L.pk_ln_mlp_wide_supported(320, 1280, 32 * 24 * 18) == 0 and L.pk_ln_mlp_wide_supported(128, 512, 64 * 16 * 12) == 0   # too few workgroups
    assert L.pk_attn_block_wide_supported(80, 2, 0) == 1 and L.pk_attn_block_wide_supported(160, 4, 0) == 0
    assert L.pk_attn_block_wide_supported(80, 2, 32 * 154) == 1 and L.pk_attn_block_wide_supported(80, 2, 100) == 0
    rc = L.pk_ln_mlp_wide_fwd(None, None, None, None, None, None, None, None, None, 1, 80, 78, 320, 1, 1e-5, None)
    assert rc == -1 and b"null pointer" in L.pk_last_error_string()
    rc = L.pk_ln_mlp_wide_fwd(None, None, None, None, None, None, None, None, None, 1, 96, 96, 384, 1, 1e-5, None)
    assert rc == -2 and b"built for" in L.pk_last_error_string()
    rc = L.pk_attn_block_wide_fwd(None, None, None, None, None, None, None, None, None, None, None, 1, 1, 4, 160, 156, 0.16, 1e-5, None)
    assert rc == -2 and b"built for" in L.pk_last_error_string()


def test_ops_refuse_cpu_tensors():
    from infantposeestimation_gaussianbias_amd import _lib, hipops
    with pytest.raises(_lib.PoseKernelError):
        hipops.argmax_decode(torch.zeros(1, 1, 4, 4))


def test_config_surface_matches_reference_defaults(golden):
    from infantposeestimation_gaussianbias_amd.configs import get_config
    ref = golden("meta.json")["schedule"]["default_config"]
    got = dataclasses.asdict(get_config())
    norm = lambda d: {k: (norm(v) if isinstance(v, dict) else ([list(x) if isinstance(x, (list, tuple)) else x for x in v]
                                                               if isinstance(v, (list, tuple)) else v)) for k, v in d.items()}
    assert norm(got) == norm(ref)
    small = get_config("hrformer_small")
    assert small.model.backbone == "hrformer_small" and small.data.input_size == (192, 256)
    pre = get_config("preemie")
    assert pre.data.num_keypoints == 13 and pre.data.sigma == 1.5
    with pytest.raises(ValueError):
        get_config("no_such_thing")


def test_legacy_yaml_mapping(tmp_path):
    from infantposeestimation_gaussianbias_amd.configs import get_config
    p = tmp_path / "x.yaml"
    p.write_text("MODEL:\n  NAME: 'pose_hrnet_w32'\n  NUM_JOINTS: 13\n  IMAGE_SIZE: [256, 256]\n  HEATMAP_SIZE: [128, 128]\n  SIGMA: 1.5\n"
                 "TRAIN:\n  BATCH_SIZE: 24\n  LR: 0.0005\n")
    cfg = get_config(str(p))
    assert cfg.data.num_keypoints == 13 and cfg.data.heatmap_size == (128, 128) and cfg.data.sigma == 1.5
    assert cfg.model.backbone == "hrnet_w32" and cfg.train.batch_size == 24


@pytest.mark.parametrize("name,bb,K,head", [("hrformer_small_fusion", "hrformer_small", 17, "fusion"),
                                            ("hrnet_w32_heatmap", "hrnet_w32", 17, "heatmap"),
                                            ("hrformer_base_fusion_k13", "hrformer_base", 13, "fusion"),
                                            ("hrnet_w18_heatmap", "hrnet_w18", 17, "heatmap"),
                                            ("hrnet_w48_heatmap", "hrnet_w48", 17, "heatmap")])
def test_state_dict_contract(golden, name, bb, K, head):
    """Key names, order, shapes and dtypes equal the reference's (checkpoint drop-in)."""
    from infantposeestimation_gaussianbias_amd.models import PoseEstimator
    from recipe import spec_of, synth_state_dict
    spec = golden("state_keys.json")[name]
    m = PoseEstimator(bb, K, False, head, True)
    assert spec_of(m.state_dict()) == spec
    assert [k for k, _ in m.named_parameters()] == golden("state_keys.json")[name + "#params"]
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(spec, 1).items()}, strict=True)


def test_state_dict_contract_without_relative_position_bias(golden):
    """HRFormer(with_rpe=False) (hrformer.py:145-191): no table parameter, no index buffer -- keys, order, shapes as the reference's."""
    from infantposeestimation_gaussianbias_amd.models import hrformer
    from recipe import spec_of
    spec = golden("norpe_r03.json")["backbone_spec"]
    m = hrformer.HRFormer(with_rpe=False, in_channels=3, drop_path_rate=0.0, stage2_num_channels=(32, 64), stage2_num_heads=(1, 2),
                          stage3_num_channels=(32, 64, 128), stage3_num_heads=(1, 2, 4), stage4_num_channels=(32, 64, 128, 256),
                          stage4_num_heads=(1, 2, 4, 8))
    assert spec_of(m.state_dict()) == spec and not any("relative_position" in k for k in spec)


def test_unknown_backbone_raises():
    from infantposeestimation_gaussianbias_amd.models import PoseEstimator
    with pytest.raises(ValueError, match="Unknown backbone"):
        PoseEstimator("resnet50")


def test_optimizer_flags_and_schedule(golden):
    from infantposeestimation_gaussianbias_amd import engine
    from infantposeestimation_gaussianbias_amd.configs import get_config
    s = golden("meta.json")["schedule"]
    for n in s["decay_names"]:
        assert not engine.is_no_decay(n), n
    for n in s["no_decay_names"]:
        assert engine.is_no_decay(n), n
    cfg = get_config()
    for it, f in zip(s["lr_iters"], s["lr_factor"]):
        assert math.isclose(engine.lr_factor(it, s["iters_per_epoch"], cfg.train), f, rel_tol=1e-12)


def test_flat_buffers_alias_parameters():
    from infantposeestimation_gaussianbias_amd import engine
    m = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.LayerNorm(3), torch.nn.Linear(3, 2))
    before = [p.detach().clone() for p in m.parameters()]
    opt = engine.FlatAdamW(m)
    assert opt.numel % 4 == 0 and all(o % 4 == 0 for o in opt.offsets)
    for p, b in zip(m.parameters(), before):
        assert torch.equal(p, b) and p.data_ptr() >= opt.flat.data_ptr()
    fl = opt.flags.numpy()
    o_w, o_b = opt.offsets[0], opt.offsets[1]
    assert fl[o_w] == 3 and fl[o_b] == 2          # weight decays, bias does not; both active
    m[0].weight.data.add_(1.0)
    assert torch.equal(opt.flat[o_w:o_w + 15].view(3, 5), m[0].weight.data)
    # gradient adoption: a parameter that never gets a gradient becomes inactive
    x = torch.randn(4, 5)
    m[0](x).sum().backward()
    opt.install_grad_views()
    assert opt.active == [True, True, False, False, False, False]
    assert opt.flags.numpy()[opt.offsets[2]] == 1 and opt.flags.numpy()[opt.offsets[3]] == 0   # active bit cleared, decay bit kept
    assert m[0].weight.grad.data_ptr() == opt.grad_view(0).data_ptr()
    sd = opt.state_dict()
    assert len(sd["param_groups"]) == 2 and sd["param_groups"][1]["weight_decay"] == 0.0


# ---------------------------------------------------------------------------------------------- N>1 over gloo
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dp_worker(rank, world, port, q, overlap):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from infantposeestimation_gaussianbias_amd import engine
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 8), torch.nn.Tanh(), torch.nn.Linear(8, 1))
    unused = torch.nn.Linear(3, 3)                      # structurally unused parameters, like stage4's dead fuse layers
    model.add_module("unused", unused)
    if rank != 0:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)                              # diverge on purpose: broadcast must repair it
    opt = engine.FlatAdamW(model)
    comm = engine.GradientExchange(opt, bucket_mb=0.0001, overlap=overlap)  # ~26 elements per bucket -> several buckets
    comm.broadcast_initial_state()
    comm.record_exposed = True                          # exposed-communication timing is event-based: a CPU exchange records nothing
    g = torch.Generator().manual_seed(123)
    data = torch.randn(8, 6, generator=g)
    shard = data[rank * 4:(rank + 1) * 4]
    grads = []
    for step in range(3):                                # step 0 = hook-less first step, 1-2 = overlapped buckets
        opt.zero_grad()
        y = model[4](model[3](model[2](model[1](model[0](shard)))))
        (y ** 2).sum().backward()
        comm.finish()
        grads.append(opt.grad.clone())
    assert comm.exposed_ms() is None
    # numpy arrays are pickled by value (tensors travel as file descriptors that die with the exiting worker: flaky)
    q.put((rank, opt.flat.detach().numpy().copy(), [g.numpy().copy() for g in grads], len(comm.buckets), list(opt.active)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_gradient_exchange_world2_matches_single_process(overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res = [(r, torch.from_numpy(f), [torch.from_numpy(g) for g in gs], nb_, act) for r, f, gs, nb_, act in res]
    (_, flat0, g0, nb, active), (_, flat1, g1, _, _) = res
    assert nb >= 3 and active[-2:] == [False, False]
    assert torch.equal(flat0, flat1)                                 # broadcast made ranks identical
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)                                     # every rank holds the same summed gradient
    # single-process reference on the concatenated batch
    from infantposeestimation_gaussianbias_amd import engine
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 8), torch.nn.Tanh(), torch.nn.Linear(8, 1))
    model.add_module("unused", torch.nn.Linear(3, 3))
    opt = engine.FlatAdamW(model)
    data = torch.randn(8, 6, generator=torch.Generator().manual_seed(123))
    y = model[4](model[3](model[2](model[1](model[0](data)))))
    (y ** 2).sum().backward()
    opt.install_grad_views()
    for g in g0:
        assert torch.allclose(g, opt.grad, rtol=1e-5, atol=1e-6)     # sum over shards == gradient of the whole batch


def test_bench_gpus_flag_never_falls_back_to_one_gpu():
    """`python bench.py --gpus N` must start N ranks or fail: with fewer than N devices it exits non-zero before touching a GPU,
    and under a torchrun environment whose WORLD_SIZE disagrees with --gpus it refuses as well."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "9"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr and not r.stdout.strip()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()


def test_gradient_sink_requested_twice_raises():
    """Direct-store gradient sinks hold ONE gradient per step: a second request inside one epoch must raise, never overwrite."""
    from infantposeestimation_gaussianbias_amd import _lib, nnops
    p = torch.nn.Parameter(torch.zeros(4))
    p._pk_grad_sink = torch.zeros(4)
    nnops.begin_grad_epoch()
    assert nnops.grad_sink_of(p) is p._pk_grad_sink and nnops.sink_written(p)
    with pytest.raises(_lib.PoseKernelError):
        nnops.grad_sink_of(p)
    nnops.begin_grad_epoch()
    assert not nnops.sink_written(p)
    assert nnops.grad_sink_of(p) is p._pk_grad_sink


def test_coco_evaluator_compute_oks_and_manual_evaluate():
    """COCOEvaluator keeps the reference's public compute_oks / _manual_evaluate (utils/metrics.py:108-142,206-270)."""
    from infantposeestimation_gaussianbias_amd.utils.metrics import COCOEvaluator
    ev = COCOEvaluator()
    rng = np.random.default_rng(5)
    gt = rng.uniform(20, 200, (17, 2))
    vis = (rng.uniform(size=17) > 0.3).astype(np.float64) * 2
    pred = gt + rng.normal(0, 4.0, (17, 2))
    area = 90.0 * 120.0
    k = np.asarray(ev.oks_sigmas, np.float64)
    d = ((pred - gt) ** 2).sum(-1)
    want = np.exp(-(d / (2 * area * k ** 2 + np.spacing(1)))[vis > 0]).sum() / (vis > 0).sum()
    got = ev.compute_oks(pred, gt, vis, area)
    assert abs(got - want) < 1e-12 and 0.0 < got < 1.0
    assert ev.compute_oks(pred, gt, np.zeros(17), area) == 0.0
    assert ev.compute_oks(gt, gt, vis, area) == 1.0
    # one prediction, one ground truth: it is a hit at exactly the thresholds its OKS reaches
    ev.predictions.append({"image_id": 1, "ann_id": 1, "keypoints": np.concatenate([pred, np.ones((17, 1))], 1).flatten().tolist(),
                           "score": 0.9, "area": area, "bbox": [0, 0, 90, 120]})
    gts = [{"image_id": 1, "keypoints": np.concatenate([gt, vis[:, None]], 1).flatten().tolist(), "area": area}]
    m = ev._manual_evaluate(gts)
    assert abs(m["AP"] - float((ev.oks_thresholds <= got).mean())) < 1e-9 and m == ev.evaluate(gts)


@pytest.mark.parametrize("k", [2, 4])
def test_deconv_parity_class_decomposition_is_conv_transpose(k):
    """The stacked-3x3-conv + pixel-shuffle form the HIP path uses for ConvTranspose2d(k, stride 2) (nnops._deconv_taps / deconv_bn_relu,
    pose_estimator.py:47-69) is exactly torch's conv_transpose2d (fp64, CPU: host logic only); kernel 3 raises like the reference."""
    import torch.nn.functional as F
    from infantposeestimation_gaussianbias_amd import nnops
    torch.manual_seed(k)
    Cin, Cout, B, H, W = 5, 3, 2, 4, 3
    w = torch.randn(Cin, Cout, k, k, dtype=torch.float64)
    x = torch.randn(B, Cin, H, W, dtype=torch.float64)
    p = (k - 1) // 2
    want = F.conv_transpose2d(x, w, stride=2, padding=p, output_padding=k - 2 * p - 2)
    ws = torch.zeros(4 * Cout, Cin, 3, 3, dtype=torch.float64)
    for cls, tap, ky, kx in nnops._deconv_taps(k):
        ws[cls * Cout:(cls + 1) * Cout, :, tap // 3, tap % 3] = w[:, :, ky, kx].t()
    y = F.conv2d(x, ws, padding=1).view(B, 2, 2, Cout, H, W)                  # (b, py, px, c, i, j)
    got = y.permute(0, 3, 4, 1, 5, 2).reshape(B, Cout, 2 * H, 2 * W)         # out[b, c, 2i+py, 2j+px]
    assert want.shape == got.shape and torch.allclose(got, want, atol=1e-12)
    with pytest.raises(ValueError):
        nnops._deconv_taps(3)


def test_bench_profile_fields_carry_their_source_and_go_stale(tmp_path, monkeypatch):
    """the in-step fields bench.py reads from profiles/ name their source files and the fingerprint of the code they were
    captured with; a profile taken with other kernel sources / launch logic is reported stale (bench.py then drops those fields)."""
    sys.path.insert(0, ROOT)
    import bench
    prof, steps = bench.load_step_profile()
    assert steps >= 4 and any(k.startswith("k_conv8p") for k in prof)
    lps, us, union_ms = prof["k_conv8p(IgemmArgs)"]
    assert lps == 7 and us > 100 and union_ms is not None and union_ms * 1e3 < lps * us       # concurrent launches: union < sum
    src = bench.profile_source()
    assert src["files"][0].startswith("profiles/") and src["captured_at_code_sha16"] and src["current_code_sha16"]
    fake = tmp_path / "meta.json"
    fake.write_text('{"code_sha16": "0000000000000000", "command": "x"}')
    monkeypatch.setattr(bench, "PROFILE_META", str(fake))
    assert bench.profile_source()["stale"] is True
    e = bench._entry("k", "shape", "mfma", 1e-4, flops=1e9, trace="k_conv8p(IgemmArgs)", single_shape=True, prof=prof, pmc=({}, {}))
    assert e["frac_in_step_union"] > e["frac_in_step"] > 0
