#!/usr/bin/env python3
"""Experiment: does hipGraph capture survive fork/join across HIP streams?  Each stage runs in a child process
(a crash in hipStreamEndCapture must not take the driver script down).

    python scripts/gpu_graph_streams.py            # runs stages A, B, C in children
    python scripts/gpu_graph_streams.py A|B|C      # one stage in this process
"""
import faulthandler
import os
import subprocess
import sys
import time

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def stage_a():
    import torch
    x = torch.randn(1024, 1024, device="cuda")
    s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    _ = (x @ x) @ x                      # library initialisation is not capturable
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s0):
        a = x @ x
        s1.wait_stream(s0)
        s2.wait_stream(s0)
        with torch.cuda.stream(s1):
            b = a @ x
        with torch.cuda.stream(s2):
            c = a + 1
        d = a * 2
        s0.wait_stream(s1)
        s0.wait_stream(s2)
        out = b + c + d
    g.replay()
    torch.cuda.synchronize()
    ref = (x @ x) @ x + (x @ x + 1) + (x @ x) * 2
    print("A: torch fork/join capture ok, max err", float((out - ref).abs().max()), flush=True)


def stage_t():
    """torch-only: autograd ACROSS streams inside a capture (the engine inserts its own events from its worker thread)."""
    import torch
    x = torch.randn(512, 512, device="cuda")
    ws = [torch.randn(512, 512, device="cuda", requires_grad=True) for _ in range(3)]
    s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()

    def fwd_bwd():
        for w in ws:
            w.grad = None
        a = x @ ws[0]
        s1.wait_stream(s0)
        s2.wait_stream(s0)
        with torch.cuda.stream(s1):
            b = torch.tanh(a @ ws[1])
        with torch.cuda.stream(s2):
            c = torch.tanh(a @ ws[2])
        s0.wait_stream(s1)
        s0.wait_stream(s2)
        loss = (b * c).sum()
        loss.backward()
        return loss

    with torch.cuda.stream(s0):
        for _ in range(3):
            fwd_bwd()
    torch.cuda.synchronize()
    ref = [w.grad.clone() for w in ws]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s0):
        fwd_bwd()
    print("T: captured", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print("T: autograd across streams in capture ok, grad err", [float((w.grad - r).abs().max()) for w, r in zip(ws, ref)], flush=True)


def _trainer(graph, streams, B):
    import torch
    from infantposeestimation_gaussianbias_amd import engine
    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.datasets import synthetic_batch
    from infantposeestimation_gaussianbias_amd.models import build_model
    cfg = get_config("hrformer_small")
    if B <= 8:
        cfg.data.input_size, cfg.data.heatmap_size = (96, 128), (24, 32)
    batch = synthetic_batch(B, cfg.data.input_size, cfg.data.heatmap_size, 17, 2.0, "cuda", seed=3)
    torch.manual_seed(0)
    model = build_model(cfg).to("cuda")
    model.backbone.drop_path_rate = 0.0
    return engine.Trainer(model, cfg, iters_per_epoch=2, use_graph=graph, graph_warmup=2, graph_streams=streams), batch


def stage_v(variant):
    """Bisect the capture crash: F = forward + loss only, G = fwd+bwd (no optimiser), J = full step + explicit join."""
    import torch
    from infantposeestimation_gaussianbias_amd import dispatch
    tr, batch = _trainer(True, True, 4)
    full = tr._fwd_bwd

    def fwd_only(b):
        tr.opt.zero_grad()
        return tr.model(b["img"], b["target"], b["target_weight"], gt_keypoints=b.get("keypoints"), input_size=tr.cfg.data.input_size)

    def joined(b):
        out = full(b)
        dispatch.join_side_streams()
        return out

    for i in range(2):
        tr.step(batch)                      # eager warm-up steps (with optimiser)
    if variant == "F":
        tr._fwd_bwd = fwd_only
    elif variant in ("J", "G"):
        tr._fwd_bwd = joined
    if variant in ("F", "G"):
        tr.opt.step = lambda *a, **k: None
    for i in range(3):
        out = tr.step(batch)
        torch.cuda.synchronize()
        print(variant, "step", i, float(out["loss"].detach()), flush=True)


def stage_b():
    """fwd+bwd+AdamW captured with branch streams, small batch; compare the loss trajectory with eager."""
    import torch
    tr, batch = _trainer(True, True, 4)
    traj = []
    for i in range(6):
        print("B: step", i, flush=True)
        traj.append(float(tr.step(batch)["loss"]))
    tr2, batch = _trainer(False, False, 4)
    ref = [float(tr2.step(batch)["loss"]) for _ in range(6)]
    print("B: graph+streams", traj, "\nB: eager        ", ref, flush=True)


def stage_c():
    """Full-size step (B=64, 256x192): time graph+streams replays."""
    import torch
    tr, batch = _trainer(True, True, 64)
    for i in range(4):
        tr.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for i in range(n):
        out = tr.step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"C: graph+streams B=64: {dt * 1e3:.2f} ms/step  {64 / dt:.1f} img/s  loss {float(out['loss']):.5f}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 2:
        st = sys.argv[1]
        if st in "FGJ":
            stage_v(st)
        elif st == "T":
            stage_t()
        else:
            {"A": stage_a, "B": stage_b, "C": stage_c}[st]()
        sys.exit(0)
    for st in (sys.argv[2] if len(sys.argv) > 2 else "ABC"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), st], timeout=400)
        print(f"stage {st}: exit code {r.returncode}", flush=True)
        if r.returncode != 0 and st in "BC":
            break
