"""How much of a training step is host enqueue time? (eager, branch streams on)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infantposeestimation_gaussianbias_amd import engine
from infantposeestimation_gaussianbias_amd.configs import get_config
from infantposeestimation_gaussianbias_amd.datasets import synthetic_batch
from infantposeestimation_gaussianbias_amd.models import build_model
cfg = get_config("hrformer_small"); cfg.train.batch_size = 64
model = build_model(cfg).to("cuda")
tr = engine.Trainer(model, cfg, iters_per_epoch=1000)
batch = synthetic_batch(64, (192, 256), (48, 64), 17, 2.0, "cuda", seed=1234)
for _ in range(5): tr.step(batch)
torch.cuda.synchronize()
host, total = [], []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(batch)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append(t1 - t0); total.append(t2 - t0)
print("host enqueue ms", sorted(host)[5] * 1e3, "step ms (enqueue+drain)", sorted(total)[5] * 1e3)
# forward-only / backward-only split
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): tr.step(batch)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
