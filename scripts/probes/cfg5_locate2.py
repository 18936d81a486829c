#!/usr/bin/env python3
"""Diagnostic: record (input, output) of every fused-half / exchange call of the cfg-5 forward, single stream vs branch streams, and report
the first call whose output differs although its inputs are identical."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from infantposeestimation_gaussianbias_amd import dispatch, exchange, nnops  # noqa: E402
from infantposeestimation_gaussianbias_amd.models import PoseEstimator  # noqa: E402
from recipe import synth_input, synth_state_dict  # noqa: E402

DEV = torch.device("cuda:0")
K, B, N = 13, int(os.environ.get("PROBE_B", "8")), int(os.environ.get("PROBE_N", "40"))
keys = json.load(open(os.path.join(ROOT, "tests", "golden", "state_keys.json")))
m = PoseEstimator("hrformer_base", K, False, "fusion", True)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(keys["hrformer_base_fusion_k13"], 44).items()}, strict=True)
m = m.to(DEV).eval()
x = torch.from_numpy(synth_input("cfg5_a", (B, 3, 384, 288))).to(DEV)
x2 = torch.cat([x, torch.flip(x, dims=[-1])], 0)
log = []


def wrap(mod, name, n_in):
    orig = getattr(mod, name)

    def f(*a, **k):
        ins = [t.clone() for t in a[:n_in] if torch.is_tensor(t)] if n_in else [t.clone() for t in a[0]]
        out = orig(*a, **k)
        outs = out if isinstance(out, (list, tuple)) else [out]
        log.append((name, ins, [t.clone() for t in outs if torch.is_tensor(t)]))
        return out
    setattr(mod, name, f)


wrap(nnops, "attn_half_wide_forward", 1)
wrap(nnops, "mlp_half_wide_forward", 1)
wrap(exchange, "unit", 0)
orig_attn = nnops._AttnHalf.apply
dispatch.set_streams(False)
with torch.no_grad():
    m(x2)
    log.clear()
    m(x2)
torch.cuda.synchronize()
ref = list(log)
print(f"{len(ref)} recorded calls per forward: " + ", ".join(sorted({r[0] for r in ref})))
dispatch.set_streams(True)
hits = 0
for it in range(N):
    log.clear()
    with torch.no_grad():
        m(x2)
    torch.cuda.synchronize()
    for k, ((n0, i0, o0), (n1, i1, o1)) in enumerate(zip(ref, log)):
        same_in = all(torch.equal(a, b) for a, b in zip(i0, i1))
        same_out = all(torch.equal(a, b) for a, b in zip(o0, o1))
        if not (same_in and same_out):
            hits += 1
            d = [(a.float() - b.float()).abs() for a, b in zip(o0, o1)]
            q = max(range(len(d)), key=lambda j: float(d[j].max()))
            idx = torch.nonzero(d[q] > 0)
            print(f"run {it}: call {k} ({n1}, input shape {tuple(i1[0].shape)}): inputs identical = {same_in}, outputs identical = {same_out}; "
                  f"{int((d[q] > 0).sum())} elements differ, max |d| = {float(d[q].max()):.4g}, first idx {idx[0].tolist() if len(idx) else None}, "
                  f"last idx {idx[-1].tolist() if len(idx) else None}")
            break
print(f"{hits} of {N} runs differ")
