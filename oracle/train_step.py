"""Oracle (test infrastructure only): one full training step of the reference algorithm on the host CPU.

Used by tests (trajectory parity) and by bench.py's `cpu_baseline` leg (kind "port": this restatement, pinned to the
reference by tests/test_oracle_golden.py, timed on the GPU box's host cores).  Never imported by the product package.
"""
import time

import numpy as np
import torch

from . import losses as olos
from . import nets as onet
from . import optim as oopt
from . import target as otgt


def init_params(spec, seed=0):
    """Random parameters for a {key: [shape, dtype]} spec (shapes are the state_dict contract; values are synthetic)."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    for k, (shape, dt) in spec.items():
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            P[k] = torch.zeros((), dtype=torch.int64)
        elif leaf == "relative_position_index":
            ws = int(round(shape[0] ** 0.5))
            ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
            ys, xs = ys.reshape(-1), xs.reshape(-1)
            P[k] = (ys[:, None] - ys[None, :] + ws - 1) * (2 * ws - 1) + (xs[:, None] - xs[None, :] + ws - 1)
        elif leaf == "running_var" or (leaf == "weight" and len(shape) == 1):
            P[k] = torch.ones(shape)
        elif leaf in ("running_mean", "bias"):
            P[k] = torch.zeros(shape)
        elif len(shape) == 4:
            P[k] = torch.randn(shape, generator=g) * (2.0 / (shape[0] * shape[2] * shape[3])) ** 0.5
        elif len(shape) == 2:
            P[k] = torch.randn(shape, generator=g) * 0.02
        else:
            P[k] = torch.full(shape, 0.5) if len(shape) == 0 else torch.randn(shape, generator=g) * 0.02
    return P


def synthetic_batch(B, input_size=(192, 256), heatmap_size=(48, 64), K=17, sigma=2.0, seed=1234):
    rng = np.random.default_rng(seed)
    W, H = input_size
    img = torch.from_numpy(rng.standard_normal((B, 3, H, W)).astype(np.float32))
    kp = (rng.uniform(0, 1, (B, K, 2)) * np.array([W, H])).astype(np.float32)
    vis = rng.choice([0.0, 1.0, 2.0], p=[0.15, 0.25, 0.6], size=(B, K)).astype(np.float32)
    tg, tw = otgt.generate_target_batch(kp, vis, input_size, heatmap_size, sigma)
    return img, torch.from_numpy(tg), torch.from_numpy(tw), torch.from_numpy(kp)


def train_step(P, pnames, state, step, batch, input_size, lr=5e-4, wd=0.01):
    img, tgt, w, kp = batch
    for k in pnames:
        P[k].requires_grad_(True)
    ctx = onet.Ctx(train=True)
    o = onet.pose_forward(img, P, ctx)
    if "offsets" in o:
        loss = olos.fusion_pose_loss(o["heatmaps"], o["offsets"], o["variances"], tgt, w, kp, input_size)["total_loss"]
    else:
        loss = olos.keypoint_mse(o["heatmaps"], tgt, w)
    grads = torch.autograd.grad(loss, [P[k] for k in pnames], allow_unused=True)
    with torch.no_grad():
        for k, g in zip(pnames, grads):
            P[k].requires_grad_(False)
            if g is None:
                continue
            if k not in state:
                state[k] = (torch.zeros_like(P[k]), torch.zeros_like(P[k]))
            oopt.adamw_step(P[k], g, state[k][0], state[k][1], step, lr, 0.0 if oopt.is_no_decay(k) else wd)
        onet.apply_bn_updates(P, ctx)
    return float(loss.detach())


def time_train_steps(spec, pnames, B=8, steps=3, warmup=1, input_size=(192, 256), heatmap_size=(48, 64), K=17, threads=None):
    """-> (images/sec, threads used). CPU fp32, all host cores unless `threads` is given."""
    if threads:
        torch.set_num_threads(threads)
    P = init_params(spec)
    batch = synthetic_batch(B, input_size, heatmap_size, K)
    state = {}
    for s in range(warmup):
        train_step(P, pnames, state, s + 1, batch, input_size)
    t0 = time.perf_counter()
    for s in range(steps):
        train_step(P, pnames, state, warmup + s + 1, batch, input_size)
    dt = time.perf_counter() - t0
    return B * steps / dt, torch.get_num_threads()


def flip_inference(P, img, flip_pairs):
    """PoseEstimator.inference with flip test (pose_estimator.py:275-329): two eval forwards, the flipped pass's heatmaps flipped
    back with left/right channels swapped and averaged, offsets of the un-flipped pass, head.decode (fusion_head.py:309-365)."""
    from . import decode as odec
    with torch.no_grad():
        ctx = onet.Ctx(train=False)
        o = onet.pose_forward(img, P, ctx)
        of = onet.pose_forward(torch.flip(img, dims=[-1]), P, ctx)
        hm = odec.flip_merge(o["heatmaps"].numpy(), of["heatmaps"].numpy(), flip_pairs)
        if "offsets" in o:
            return odec.fusion_decode(hm, o["offsets"].numpy(), float(P["head.subpixel_refine.alpha"]), float(torch.sigmoid(P["head.fusion_weight"])))
        return odec.argmax_decode(hm)


def time_flip_inference(spec, flip_pairs, B=4, steps=3, warmup=1, input_size=(288, 384), threads=None):
    """-> (images/sec, threads used): flip-test inference of the CPU oracle, fp32."""
    if threads:
        torch.set_num_threads(threads)
    P = init_params(spec)
    W, H = input_size
    img = torch.from_numpy(np.random.default_rng(1234).standard_normal((B, 3, H, W)).astype(np.float32))
    for _ in range(warmup):
        flip_inference(P, img, flip_pairs)
    t0 = time.perf_counter()
    for _ in range(steps):
        flip_inference(P, img, flip_pairs)
    dt = time.perf_counter() - t0
    return B * steps / dt, torch.get_num_threads()
