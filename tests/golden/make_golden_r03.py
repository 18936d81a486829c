#!/usr/bin/env python3
"""Round-3 additions to the golden vectors, captured from the reference's own modules (build container only, see make_golden.py):

  * HRFormer(with_rpe=False) (models/hrformer.py:145-191): state_dict spec of the small-width backbone (no relative-position table /
    index buffer) and an eval forward + input gradient of one HRFormerBlock built without the bias, on recipe weights.

    python tests/golden/make_golden_r03.py            # writes tests/golden/norpe_r03.npz + norpe_r03.json
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (puts the reference on sys.path, stubs the absent third-party packages)
from recipe import spec_of, synth_input  # noqa: E402

SMALL = dict(in_channels=3, drop_path_rate=0.0, stage2_num_channels=(32, 64), stage2_num_heads=(1, 2), stage3_num_channels=(32, 64, 128),
             stage3_num_heads=(1, 2, 4), stage4_num_channels=(32, 64, 128, 256), stage4_num_heads=(1, 2, 4, 8))


def main():
    from models import hrformer as rh
    torch.manual_seed(0)
    bb = rh.HRFormer(with_rpe=False, **SMALL).eval()
    spec = mg.load_recipe(bb, salt=47)
    x = mg.T(synth_input("norpe_bb", (1, 3, 64, 64)))
    with torch.no_grad():
        y = bb(x)
    y0 = y[0] if isinstance(y, (list, tuple)) else y
    out = {"bb_out": mg.N(y0)}
    blk = rh.HRFormerBlock(64, 2, window_size=7, with_rpe=False, drop_path=0.0).eval()
    bspec = mg.load_recipe(blk, salt=48)
    xb = mg.T(synth_input("norpe_blk", (2, 64, 9, 10))).requires_grad_(True)
    yb = blk(xb)
    gy = mg.T(synth_input("norpe_blk_gy", tuple(yb.shape)))
    yb.backward(gy)
    out.update(blk_out=mg.N(yb), blk_gx=mg.N(xb.grad), blk_gqkv=mg.N(blk.attn.qkv.weight.grad))
    mg.save("norpe_r03.npz", **out)
    with open(os.path.join(HERE, "norpe_r03.json"), "w") as f:
        json.dump({"backbone_spec": spec, "block_spec": bspec}, f, separators=(",", ":"))
    print("keys without the table:", not any("relative_position" in k for k in spec), len(spec))


if __name__ == "__main__":
    main()
