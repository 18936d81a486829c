#!/usr/bin/env python3
"""Diagnostic: is the served cfg-5 inference path (batched flip, branch streams, hipGraph replay) bitwise reproducible from replay to replay,
and how far is it from the two-pass single-stream eager path?  (A replay-to-replay difference would be a race.)"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from infantposeestimation_gaussianbias_amd import dispatch  # noqa: E402
from infantposeestimation_gaussianbias_amd.models import PoseEstimator  # noqa: E402
from recipe import synth_input, synth_state_dict  # noqa: E402

DEV = torch.device("cuda:0")
K, B = 13, 8
keys = json.load(open(os.path.join(ROOT, "tests", "golden", "state_keys.json")))
m = PoseEstimator("hrformer_base", K, False, "fusion", True)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(keys["hrformer_base_fusion_k13"], 44).items()}, strict=True)
m = m.to(DEV).eval()
pairs = [(1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12)]
xs = [torch.from_numpy(synth_input(n, (B, 3, 384, 288))).to(DEV) for n in ("cfg5_a", "cfg5_b")]
C = lambda t: t.detach().float().cpu().numpy()
os.environ["POSE_FLIP_BATCHED"] = "0"
dispatch.set_streams(False)
ref = []
with torch.no_grad():
    for rep in range(2):
        for k, x in enumerate(xs):
            kp, sc = m.inference(x, flip=True, flip_pairs=pairs)
            if rep == 0:
                ref.append((C(kp), C(sc)))
            else:
                print(f"two-pass eager, input {k}: repeat identical = {np.array_equal(C(kp), ref[k][0]) and np.array_equal(C(sc), ref[k][1])}")
os.environ["POSE_FLIP_BATCHED"] = "1"
with torch.no_grad():
    for k, x in enumerate(xs):
        kp, sc = m.inference(x, flip=True, flip_pairs=pairs)
        d = np.abs(C(kp) - ref[k][0])
        print(f"batched eager (no streams), input {k}: max |dkp| = {d.max():.5f} at {np.unravel_index(d.argmax(), d.shape)}, > 0.01: {(d > 0.01).sum()}")
dispatch.set_streams(True)
with torch.no_grad():
    outs = []
    for rep in range(3):
        for k, x in enumerate(xs):
            kp, sc = m.inference(x, flip=True, flip_pairs=pairs)
            torch.cuda.synchronize()
            if rep == 0:
                outs.append((C(kp), C(sc)))
                d = np.abs(C(kp) - ref[k][0])
                print(f"batched eager + streams, input {k}: max |dkp| = {d.max():.5f}, > 0.01: {(d > 0.01).sum()}")
            else:
                print(f"batched eager + streams, input {k}, repeat {rep}: identical = {np.array_equal(C(kp), outs[k][0]) and np.array_equal(C(sc), outs[k][1])}")
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
first = {}
for rep in range(6):
    for k, x in enumerate(xs):
        static.copy_(x)
        g.replay()
        torch.cuda.synchronize()
        kp, sc = C(out[0]), C(out[1])
        if k not in first:
            first[k] = (kp, sc)
            d = np.abs(kp - ref[k][0])
            print(f"graph replay, input {k}: max |dkp| vs two-pass = {d.max():.5f} at {np.unravel_index(d.argmax(), d.shape)}, > 0.01: {(d > 0.01).sum()}; "
                  f"vs eager+streams identical = {np.array_equal(kp, outs[k][0])}")
        else:
            same = np.array_equal(kp, first[k][0]) and np.array_equal(sc, first[k][1])
            print(f"graph replay {rep}, input {k}: identical to first replay = {same}" + ("" if same else f"  max |dkp| = {np.abs(kp - first[k][0]).max():.5f}"))
