import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infantposeestimation_gaussianbias_amd import nnops
from scripts.bench_kernels import Holder, timeit
DEV, BF = "cuda", torch.bfloat16
B, H, W = 64, 64, 48
for Cin, Cout, k in [(64, 256, 1), (256, 64, 1), (64, 64, 3), (256, 256, 3), (64, 64, 1)]:
    conv = torch.nn.Conv2d(Cin, Cout, k, 1, k // 2, bias=False)
    m = Holder(c=conv).to(DEV)
    x = torch.randn(B, H, W, Cin, device=DEV).to(BF)
    with nnops.use_weights(m) as wc:
        wf = wc.fwd[id(m.c.weight)]
        for st in (False, True):
            sec = timeit(lambda: nnops._conv_raw(x, wf, Cout, k, 1, st), iters=50)
            print(f"conv {Cin}->{Cout} k{k} stats={st}: {sec * 1e6:7.1f} us", flush=True)
