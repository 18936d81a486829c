#!/usr/bin/env python3
"""Diagnostic: does the cfg-5 inference result depend on the CONTENT of freshly allocated (torch.empty) device memory?  Every allocation is
pre-filled with 0x00 or 0xFF bytes (bf16 / fp32 NaN, int -1); single stream, eager.  A result that changes with the fill pattern means some
kernel reads words nobody wrote."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from infantposeestimation_gaussianbias_amd import dispatch, nnops  # noqa: E402
from infantposeestimation_gaussianbias_amd.models import PoseEstimator  # noqa: E402
from recipe import synth_input, synth_state_dict  # noqa: E402

DEV = torch.device("cuda:0")
K, B = 13, int(os.environ.get("PROBE_B", "8"))
keys = json.load(open(os.path.join(ROOT, "tests", "golden", "state_keys.json")))
m = PoseEstimator("hrformer_base", K, False, "fusion", True)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(keys["hrformer_base_fusion_k13"], 44).items()}, strict=True)
m = m.to(DEV).eval()
pairs = [(1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12)]
x = torch.from_numpy(synth_input("cfg5_a", (B, 3, 384, 288))).to(DEV)
C = lambda t: t.detach().float().cpu().numpy()
dispatch.set_streams(False)
_orig = torch.empty
FILL = [None]


def filled(*a, **k):
    t = _orig(*a, **k)
    if FILL[0] is not None and t.is_cuda and t.numel():
        t.view(-1).view(torch.uint8).fill_(FILL[0])
    return t


torch.empty = filled
res = {}
with torch.no_grad():
    for name, f in (("plain", None), ("zeros", 0), ("ones", 0xFF), ("plain2", None), ("ones2", 0xFF)):
        FILL[0] = f
        out = m(torch.cat([x, torch.flip(x, dims=[-1])], 0))
        torch.cuda.synchronize()
        res[name] = {k: C(v) for k, v in out.items() if torch.is_tensor(v)}
        hm = res[name]["heatmaps"]
        print(f"{name:7s}: heatmaps finite = {np.isfinite(hm).all()}, nan count = {np.isnan(hm).sum()}")
for a, b in (("plain", "zeros"), ("zeros", "ones"), ("plain", "plain2"), ("ones", "ones2")):
    for k in res[a]:
        same = np.array_equal(res[a][k], res[b][k], equal_nan=True)
        d = np.nanmax(np.abs(res[a][k] - res[b][k])) if not same else 0.0
        print(f"{a} vs {b}: {k}: identical = {same}  max |d| = {d:.3e}")
# the whole served path: forward + flip merge + decode, single stream and branch streams
for streams in (False, True):
    dispatch.set_streams(streams)
    got = {}
    with torch.no_grad():
        for name, f in (("plain", None), ("zeros", 0), ("ones", 0xFF), ("ones2", 0xFF), ("zeros2", 0), ("plain2", None)):
            FILL[0] = f
            kp, sc = m.inference(x, flip=True, flip_pairs=pairs)
            torch.cuda.synchronize()
            got[name] = (C(kp), C(sc))
    for a_, b_ in (("plain", "zeros"), ("zeros", "ones"), ("ones", "ones2"), ("zeros", "zeros2"), ("plain", "plain2")):
        same = np.array_equal(got[a_][0], got[b_][0], equal_nan=True) and np.array_equal(got[a_][1], got[b_][1], equal_nan=True)
        print(f"inference, streams={streams}: {a_} vs {b_}: identical = {same}  max |dkp| = {np.nanmax(np.abs(got[a_][0] - got[b_][0])):.4f} "
              f"nan = {np.isnan(got[b_][0]).sum()}")
