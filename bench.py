#!/usr/bin/env python3
"""Headline benchmark: training images/sec (fwd + bwd + AdamW step), HRFormer-small + fusion head, 256x192, bf16.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME]

`--gpus N` without a torchrun environment starts N fresh ranks itself (one child process per GPU); under
`python -m torch.distributed.run --nproc-per-node N` it reads RANK / LOCAL_RANK / WORLD_SIZE.  Rank 0 prints ONE JSON line
(contract in the task statement).  Synthetic device-resident batches (seed 1234+rank), weak scaling (fixed per-GPU batch),
DropPath on, BN in train mode, every parameter updated.  Extra objects on the line:
  `roofline`      the dominant hand-written kernel of the step AND a `table` with the top aggregate kernels of the rocprof trace
                  (profiles/), each on its dominant shape, each timed live with HIP events on the launch stream;
  `cpu_baseline`  the CPU oracle (parity-pinned port of the reference) timed on this host's cores (rank 0, N=1 only).
Other BASELINE.json configs (not the headline metric): --config hrnet_w32_384 (cfg 4: HRNet-W32 heatmap head, 384x288, training) and
--config hrformer_base_infer (cfg 5: HRFormer-base + fusion head, K=13, 384x288, flip-test inference, hipGraph replay).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def log(msg):
    print(f"# [{time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16

CONFIGS = {
    "hrformer_small": dict(preset="hrformer_small", batch=64, mode="train", metric="images/sec (train fwd+bwd) HRFormer-S 256x192",
                           workload="HRFormer-small + fusion head, 256x192 -> 64x48, K=17, train step fwd+bwd+AdamW, DropPath 0.1, BN train"),
    "hrnet_w32_384": dict(preset="hrnet_w32", batch=32, mode="train", metric="images/sec (train fwd+bwd) HRNet-W32 384x288",
                          workload="HRNet-W32 + heatmap head (KeypointMSELoss), 384x288 -> 96x72, K=17, train step fwd+bwd+AdamW, BN train; "
                                   "per-GPU batch 32 = the reference's TrainConfig.batch_size default (configs/config.py; its configs/hrnet_w32.yaml is an empty file "
                                   "and BASELINE.json states no batch for this config)"),
    "hrformer_base_infer": dict(preset="preemie", batch=32, mode="infer", metric="images/sec (flip-test inference) HRFormer-B 384x288",
                                workload="HRFormer-base + fusion head, K=13, 384x288 -> 96x72, flip-test inference (forward of x and of flip(x) as one 2B-sample batch + flip merge + decode; branch streams inside the graph), "
                                         "8-aligned padded twin"),
}


def time_kernel(fn, iters=20, warmup=3):
    """Average device time of `fn()` (one launch sequence on torch's current stream == the stream the C-ABI launches on).  The
    `iters` calls are captured into one hipGraph and the replay is timed with HIP events: launched one by one from Python, kernels
    shorter than ~20 us are paced by the host (two C-ABI calls take longer to issue than a 12 us kernel takes to run)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    try:
        g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        # N > 1: the process group's watchdog thread may query events while we capture; only police the capturing thread's own calls
        # (as engine.Trainer._capture does); all collectives have completed by the time the probes run
        mode = "thread_local" if (dist.is_initialized() and dist.get_world_size() > 1) else "global"
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side, capture_error_mode=mode):
                for _ in range(iters):
                    fn()
        torch.cuda.current_stream().wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e-3
    except Exception as e:       # noqa: BLE001  (a refused capture falls back to host-paced launches)
        log(f"time_kernel: graph capture refused ({type(e).__name__}: {e}); timing host-paced launches")
        torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


PROFILE_TAG = "r04"
PROFILE_CSV = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_bench_kernel_stats.csv")      # rocprofv3 --kernel-trace --stats of `python bench.py --steps 4 --warmup 2`
PROFILE_UNION = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_kernel_union.json")         # scripts/trace_summary.py over the same trace: union of each kernel's intervals
PROFILE_META = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_profile_meta.json")          # scripts/profile_meta.py: fingerprint of the code the profile was taken with


def profile_source():
    """Where the in-step fields of the roofline rows come from, and whether the committed profile still describes this code: the fields
    read from profiles/ are nulled when the fingerprint of the kernel sources / launch logic differs from the one stored with the profile."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        import profile_meta
        now = profile_meta.fingerprint()
    except Exception:       # noqa: BLE001
        now = None
    try:
        with open(PROFILE_META) as f:
            meta = json.load(f)
    except (OSError, ValueError):
        meta = {}
    return {"files": [os.path.relpath(p, ROOT) for p in (PROFILE_CSV, PROFILE_UNION)], "captured_at_code_sha16": meta.get("code_sha16"),
            "current_code_sha16": now, "stale": not (now and meta.get("code_sha16") == now), "command": meta.get("command")}


def load_step_profile():
    """Per-kernel in-step numbers from the committed rocprof summary of THIS command: {kernel name: (launches per step, average us,
    union ms per step or None)}.  Steps in the trace = launches of k_adamw (one per step, eager warm-up steps and graph replays alike)."""
    import csv
    try:
        with open(PROFILE_CSV) as f:
            rows = list(csv.DictReader(f))
    except OSError:
        return {}, 0
    steps = max([int(r["Calls"]) for r in rows if r["Name"].startswith("k_adamw")] or [0])
    if not steps:
        return {}, 0
    try:
        with open(PROFILE_UNION) as f:
            uni = json.load(f)["kernels"]
    except (OSError, KeyError, ValueError):
        uni = {}
    return {r["Name"].replace("void ", ""): (int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3,
                                            (uni.get(r["Name"].replace("void ", "")) or {}).get("union_ms_per_step")) for r in rows}, steps


def load_pmc():
    """-> (isolated, step): L2-miss bytes per launch from the --pmc passes over scripts/prof_kernels.py.  `isolated` has one shape per kernel
    name (the roofline probes' shapes); `step` is the mean over every launch of that name in an eager step: exact only for single-shape kernels."""
    out = []
    for name in (f"{PROFILE_TAG}_pmc_traffic_isolated.json", f"{PROFILE_TAG}_pmc_traffic_step.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                out.append(json.load(f)["kernels"])
        except (OSError, KeyError, ValueError):
            out.append({})
    return out


def _entry(kernel, what, bound, sec, flops=None, bytes_=None, trace=None, single_shape=False, prof=None, pmc=None):
    """One roofline row.  `sec`: the probe's average launch duration measured live (isolated, contention-free).  `trace`: the kernel's
    name in the rocprof summary; `us_in_step_avg` / `launches_per_step` come from profiles/ (same command, whole step, concurrent
    streams), and `frac_in_step` is the same algorithmic work over that in-step average -- only where every launch of that kernel name
    in the step has the probe's shape (`single_shape`); template kernels that serve many shapes get null."""
    work = flops if bound == "mfma" else bytes_
    peak, unit, scale = (MFMA_BF16_PEAK_TFLOPS, "TFLOP/s", 1e12) if bound == "mfma" else (HBM_PEAK_GBS, "GB/s", 1e9)
    ach = work / sec / scale
    e = {"kernel": kernel, "shape": what, "bound": bound, "achieved": round(ach, 1), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
         "us_per_launch": round(sec * 1e6, 1), "traffic": None}
    if flops:
        e["algorithmic_flops"] = flops
    if bytes_:
        e["algorithmic_bytes"] = bytes_
    if pmc and trace:
        key = trace.split("(")[0]
        iso, stp = pmc
        if key in iso:
            e["traffic"] = iso[key]["traffic_bytes"]
        elif single_shape and key in stp:
            e["traffic"] = stp[key]["traffic_bytes"]
    if prof and trace and trace in prof:
        lps, us, union_ms = prof[trace]
        e["launches_per_step"] = round(lps, 2)
        e["us_in_step_avg"] = round(us, 1)
        e["ms_in_step"] = round(lps * us / 1e3, 3)
        e["frac_in_step"] = round(work / (us * 1e-6) / scale / peak, 4) if single_shape else None
        # instances that share the chip on concurrent streams (the head's three branches) each look slow per launch: total work of the
        # step's launches over the UNION of their intervals is what the kernel family delivers in the step
        e["union_ms_in_step"] = None if union_ms is None else round(union_ms, 3)
        e["frac_in_step_union"] = round(work * lps / (union_ms * 1e-3) / scale / peak, 4) if (single_shape and union_ms) else None
    elif prof and trace:
        log(f"roofline: kernel name {trace!r} not found in {os.path.basename(PROFILE_CSV)} (renamed instantiation?): in-step fields omitted")
    return e


def roofline_table(model, B, trainer=None):
    """The kernels that top the aggregate of the step (profiles/r04_trace_summary.txt), each on its dominant shape, timed live here and
    set beside its in-step numbers from profiles/.  Algorithmic work: GEMM-shaped kernels 2*M*N*K flops vs the dense
    bf16 MFMA peak; the shallow token GEMMs, the fused block kernels, BatchNorm and the slab reduction are HBM-bound: bytes = every
    operand read once + every result written once."""
    from infantposeestimation_gaussianbias_amd import _lib, nnops
    from infantposeestimation_gaussianbias_amd._lib import call, stream_ptr
    dev = next(model.parameters()).device
    BF = torch.bfloat16
    H, W = 64, 48
    M = B * H * W
    out = []
    prof, steps_in_trace = load_step_profile()
    pmc = load_pmc()
    src = profile_source()
    if src["stale"]:
        log(f"roofline: profiles/{PROFILE_TAG}_* were captured at code {src['captured_at_code_sha16']}, this tree is {src['current_code_sha16']}: "
            "in-step fields read from them are omitted (re-run scripts/gpu_r04_profile.sh)")
        prof, pmc = {}, ({}, {})
    E = lambda *a, **k: out.append(_entry(*a, prof=prof, pmc=pmc, **k))
    with nnops.use_weights(model) as wc:
        # 1. head conv 3x3 256->256 @64x48 (fusion_head.py:215,224,235), forward with the BN-statistics epilogue; the same kernel runs the
        #    data gradient: 8 launches of this shape per step
        conv = model.head.shared_layers["3"]
        C = conv.weight.shape[1]
        x = torch.randn(B, H, W, C, device=dev).to(BF)
        g = torch.randn(B, H, W, C, device=dev).to(BF)
        wf = wc.fwd[id(conv.weight)]
        flops = 2.0 * M * conv.weight.shape[0] * 9 * C
        sec = time_kernel(lambda: nnops._conv_raw(x, wf, conv.weight.shape[0], 3, 1, True))
        E("k_conv8p", "head conv3x3 256->256 @64x48, fwd + BN-stat epilogue (same kernel: data gradient)", "mfma", sec, flops=flops,
          bytes_=2.0 * (2 * M * C) + 2 * 9 * C * C, trace="k_conv8p(IgemmArgs)", single_shape=True)
        # Weight gradients as the step launches them: slabs only (dw = NULL), the split-M partial sums of ALL layers are reduced by
        # the step's one k_reduce_many launch.
        def wgrad_slabs(xx, gg, Mr, N_, Cin_, ks, geom, a_map=None):
            # geom: (B, H, W) of the conv input, or -- linear form with a window row map -- of the token grid the map partitions
            Bq, Hq, Wq = geom if geom else (0, 0, 0)
            Ho, Wo = (0, 0) if a_map is not None else (Hq, Wq)
            S = _lib.lib.pk_wgrad_slices(Mr, N_, Cin_, ks, 1, Hq, Wq, 1 if a_map is not None else 0)
            ws = torch.empty(S * N_ * (ks * ks * Cin_ + 1), device=dev)
            return lambda: call("pk_wgrad_bf16", xx, gg, ws, None, None, 0, a_map, None, None, 0, Mr, N_, Cin_, ks, 1, Bq, Hq, Wq, Ho, Wo, 0,
                                stream_ptr())
        # 2. the head conv's weight gradient: k_wgrad3 (256 x 256 tile, 5-stage LDS-DMA ring)
        sec = time_kernel(wgrad_slabs(x, g, M, C, C, 3, (B, H, W)))
        E("k_wgrad3", "head conv3x3 256->256 @64x48 weight gradient (slabs)", "mfma", sec, flops=flops, bytes_=2.0 * (2 * M * C),
          trace="k_wgrad3(WgradArgs)", single_shape=True)
        # 3. streaming weight gradient k_wgrad4<128,64>: qkv weight gradient of the branch-0 blocks (tokens x 96 x 32)
        Mw = B * 70 * 49
        u, dq = torch.randn(Mw, 32, device=dev).to(BF), torch.randn(Mw, 96, device=dev).to(BF)
        sec = time_kernel(wgrad_slabs(u, dq, Mw, 96, 32, 1, None))
        E("k_wgrad4<128,64>", "qkv weight gradient, branch 0: 219520 tokens x 96 x 32 (slabs)", "hbm", sec, flops=2.0 * Mw * 96 * 32,
          bytes_=2.0 * Mw * (96 + 32), trace="k_wgrad4<128, 64, false>(WgradArgs)")
        # 3b. k_wgrad4w<128,64> (windowed / row-scaled streaming kernel): qkv weight gradient of the branch-1 blocks, x gathered through
        # the 7x7 window partition (recomputed per DMA lane)
        amap1, nwin1 = nnops.window_rowmap(B, 32, 24, dev)
        Mw1 = amap1.numel()
        u1, dq1 = torch.randn(B * 32 * 24, 64, device=dev).to(BF), torch.randn(Mw1, 192, device=dev).to(BF)
        sec = time_kernel(wgrad_slabs(u1, dq1, Mw1, 192, 64, 1, (B, 32, 24), a_map=amap1))
        E("k_wgrad4w<128,64>", f"qkv weight gradient, branch 1: {Mw1} window tokens x 192 x 64, gathered rows (slabs)", "hbm", sec,
          flops=2.0 * Mw1 * 192 * 64, bytes_=2.0 * (Mw1 * 192 + B * 32 * 24 * 64), trace="k_wgrad4w<128, 64, 1>(WgradArgs)")      # MODE 1: A rows gathered through the window partition
        # 4. k_conv3h<64,64> (halo kernel): 3x3 conv 64->64 @64x48 (layer1), forward with BN statistics; HBM-bound by arithmetic
        x64 = torch.randn(B, H, W, 64, device=dev).to(BF)
        w64 = torch.randn(64, 9, 64, device=dev).to(BF)
        sec = time_kernel(lambda: nnops._conv_raw(x64, w64, 64, 3, 1, True))
        E("k_conv3h<64,64>", "conv3x3 64->64 @64x48 fwd + BN statistics (same kernel: data gradient)", "hbm", sec, flops=2.0 * M * 64 * 9 * 64,
          bytes_=2.0 * (2 * M * 64), trace="k_conv3h<64, 64>(IgemmArgs, int)", single_shape=True)
        # 4a. k_igemm2<128,64,4,1,32>: the most-launched GEMM tile of the step (token GEMMs and 1x1 convs with K <= 128); probe = 1x1 conv
        #     64->256 @64x48 data gradient (N = 64, K = 256 goes to the BK = 64 tile; this is its forward twin N = 256, K = 64, no statistics)
        w1 = torch.randn(256, 1, 64, device=dev).to(BF)
        sec = time_kernel(lambda: nnops._conv_raw(x64, w1, 256, 1, 1, False))
        E("k_igemm2<128,64,4,1,32>", "conv1x1 64->256 @64x48 (output-bound: 25 MB in, 100 MB out)", "hbm", sec, flops=2.0 * M * 256 * 64,
          bytes_=2.0 * (M * 64 + M * 256), trace="k_igemm2<128, 64, 4, 1, 32, 3>(IgemmArgs)")      # plain-addressing instantiation (LEAN = 3)
        # 4c. stride-2 data gradient (parity-grouped tap walk): stem conv2 64->64 s2, dx at 128x96
        g2 = torch.randn(B, H, W, 64, device=dev).to(BF)
        wd2 = torch.randn(64, 9, 64, device=dev).to(BF)
        sec = time_kernel(lambda: nnops._conv_dgrad(g2, wd2, 64, 3, 2, (128, 96)))
        E("k_igemm2<128,64,4,1,64> (dilated)", "data gradient of conv3x3 s2 64->64: dx 128x96 from dy 64x48, 2.25 of 9 taps per pixel", "hbm", sec,
          flops=2.0 * (4 * M) * 64 * 64 * 2.25, bytes_=2.0 * (M * 64 + 4 * M * 64), trace=None)
        # 4b. k_igemm2<128,32,4,1,64>: largest shape = 3x3 conv 256->32 @64x48 (transition1 branch 0 forward / data gradient of the head's 32->256 conv)
        w32 = torch.randn(32, 9, 256, device=dev).to(BF)
        sec = time_kernel(lambda: nnops._conv_raw(x, w32, 32, 3, 1, True))
        E("k_igemm2<128,32,4,1,64>", "conv3x3 256->32 @64x48 fwd + BN-stat epilogue", "mfma", sec, flops=2.0 * M * 32 * 9 * 256,
          bytes_=2.0 * (M * 256 + M * 32), trace="k_igemm2<128, 32, 4, 1, 64, 3>(IgemmArgs)")
        # 5./6. the fused block halves of branch 0 (C = 32): bytes = x in + y out (forward), x + dy in, dx out (backward)
        blk = model.backbone.stage2[0].branches[0][0]
        xb = torch.randn(B, H, W, 32, device=dev).to(BF)
        gy = torch.randn(B, H, W, 32, device=dev).to(BF)
        s = torch.ones(B, device=dev)
        m, a = blk.mlp, blk.attn
        if nnops.fused_mlp_enabled(32):
            with torch.no_grad():
                sec = time_kernel(lambda: nnops._MlpHalfFused.apply(xb, blk.norm2.weight, blk.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight,
                                                                    m.fc2.bias, s))
            E("k_mlp_fwd<32>", "LN2+fc1+GELU+fc2+residual, 196608 tokens x 32 (hidden 128 in registers)", "hbm", sec,
              flops=16.0 * M * 32 * 32, bytes_=4.0 * M * 32, trace="k_mlp_fwd<32>(MlpArgs)", single_shape=True)
        if nnops.fused_attn_enabled(32, 1):
            aargs = (blk.norm1.weight, blk.norm1.bias, a.relative_position_bias_table, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias, s, 1)
            y, o, lse, amap = nnops.attn_half_fused_forward(xb, *aargs, save=True)
            nw = amap.numel() // 49
            nb = _lib.lib.pk_attn_block_blocks(nw)
            lnp, rpb = torch.empty(nb * 64, device=dev), torch.empty(nb * 4 * 169, device=dev)
            dx, dqkv, u_w = torch.empty_like(xb), torch.empty(nw * 49, 96, device=dev, dtype=BF), torch.empty(nw * 49, 32, device=dev, dtype=BF)
            sec = time_kernel(lambda: call("pk_attn_block_bwd", gy, xb, amap, blk.norm1.weight, blk.norm1.bias, a.relative_position_bias_table,
                                           wc.fwd[id(a.qkv.weight)], a.qkv.bias, wc.dgrad[id(a.qkv.weight)], wc.dgrad[id(a.proj.weight)], s, o, lse,
                                           dx, dqkv, u_w, lnp, rpb, nw, nw // B, 1, 32, 1e-5, stream_ptr()))
            E("k_attn_bwd<32>", "attention-half backward, 4480 windows x 49 tokens x 32 (dx + dqkv + u)", "hbm", sec,
              bytes_=2.0 * (3 * M * 32 + nw * 49 * (32 + 96 + 32)), trace="k_attn_bwd<32>(AttnArgs, float*, float*)", single_shape=True)
    # 7. BatchNorm on the head-sized tensors (196608 x 256: 100 MB each): apply (raw -> y), backward reduce (dy, y, raw read), backward apply
    C2 = 256
    raw, y, dy = (torch.randn(M, C2, device=dev).to(BF) for _ in range(3))
    vec = lambda: torch.rand(C2, device=dev) + 0.5
    scale, shift, mean, rstd, gamma = vec(), vec(), vec(), vec(), vec()
    yo = torch.empty_like(raw)
    maskb = torch.empty(M * C2 // 8, dtype=torch.uint8, device=dev)       # ReLU bit mask: written by the forward, read by both backward passes
    sec = time_kernel(lambda: call("pk_bn_act", raw, scale, shift, None, yo, M, C2, 1, maskb, stream_ptr()))
    E("k_bn_act", "scale/shift + ReLU, 196608 x 256 (raw in, y + bit mask out)", "hbm", sec, bytes_=4.0 * M * C2 + M * C2 / 8, trace=next((k for k in prof if k.startswith("k_bn_act(")), None))
    nbb = _lib.lib.pk_bn_bwd_blocks(M)
    part, sums, dga, dbe = torch.empty(nbb, 2, C2, device=dev), torch.empty(2 * C2, device=dev), torch.empty(C2, device=dev), torch.empty(C2, device=dev)
    draw = torch.empty_like(raw)
    sec = time_kernel(lambda: call("pk_bn_bwd", dy, None, raw, mean, rstd, gamma, part, sums, dga, dbe, draw, None, M, C2, 1, maskb, stream_ptr()))
    E("k_bn_bwd_reduce + k_sum_partials + k_bn_bwd_apply", "BatchNorm backward, 196608 x 256: two passes over (dy, raw, ReLU bit mask) + dx out (three launches)", "hbm",
      sec, bytes_=2.0 * M * C2 * (2 + 2 + 1) + 2 * M * C2 / 8, trace=None)
    # 8. the step's slab reduction (ONE launch for all layers): bytes = every slab read once + every gradient written once
    tabs = [(k, t) for k, t in nnops._TABLES.items()]
    if tabs:
        key, tab = max(tabs, key=lambda kt: sum(r[3] * r[4] for r in kt[0]))
        slab_bytes = 4.0 * sum(r[3] * r[4] + r[4] for r in key)
        sec = time_kernel(lambda: call("pk_reduce_many", tab["desc"], tab["blk_desc"], tab["blk_first"], tab["nb"], stream_ptr()), iters=5)
        E("k_reduce_many", f"slab reduction of the whole step: {len(key)} rows, {tab['nb']} workgroups", "hbm", sec, bytes_=slab_bytes,
          trace="k_reduce_many(ReduceDesc const*, int const*, int const*)")       # (the trace also holds the first warm-up step's per-layer reductions)
    # 9. input pipeline (f2; not part of the synthetic step): inverse-affine crop + normalise of 64 640x480 uint8 images -> bf16 NHWC-8
    try:
        import numpy as np
        from infantposeestimation_gaussianbias_amd.datasets import transforms as T
        imgs = [np.random.default_rng(i).integers(0, 255, (480, 640, 3), dtype=np.uint8) for i in range(4)] * (B // 4)
        mats = [np.array([[0.4, 0.0, -32.0], [0.0, 0.4, 32.0]], np.float64)] * len(imgs)
        crop = T.DeviceCropper((192, 256), dev, nchw=False, nhwc8=True)
        for _ in range(2):           # both pinned staging buffers (58 MB each) are allocated on first use: not the steady state
            crop(imgs, mats)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            crop(imgs, mats)
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / 4
        E("k_affine_crop_normalize (+ host staging)", "64 x 640x480x3 uint8 -> 64 x 256x192 bf16 NHWC-8: wall time incl. the pinned-host copy (PCIe)", "hbm",
          sec, bytes_=float(len(imgs) * (480 * 640 * 3 + 256 * 192 * 16)), trace=None)
    except Exception as e:       # noqa: BLE001  (the probe is informational)
        log(f"input-pipeline probe skipped: {type(e).__name__}: {e}")
    for e in out:
        e["steps_in_trace"] = steps_in_trace
        e["profile_source"] = src       # the fields us_in_step_avg / launches_per_step / ms_in_step / frac_in_step* / traffic come from these files
    return out


def cpu_baseline(cfg_name, c, cfg):
    """The CPU oracle (parity-pinned port of the reference, fp32 PyTorch-CPU) on this host's cores: a bounded sample of the same workload."""
    from oracle import train_step as ots
    with open(os.path.join(ROOT, "tests", "golden", "state_keys.json")) as f:
        keys = json.load(f)
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))      # the GPU box gives one job a 16-CPU share of a much larger host
    if cfg_name == "hrformer_small":
        ips, threads = ots.time_train_steps(keys["hrformer_small_fusion"], keys["hrformer_small_fusion#params"], B=8, steps=5, warmup=2,
                                            input_size=c["input"], heatmap_size=c["heatmap"], K=c["K"], threads=cores)
        sample = "HRFormer-small fusion 256x192 train step (fwd+bwd+AdamW), fp32 PyTorch-CPU oracle, B=8, 2 warm-up + 5 timed steps"
    elif cfg_name == "hrnet_w32_384":
        ips, threads = ots.time_train_steps(keys["hrnet_w32_heatmap"], keys["hrnet_w32_heatmap#params"], B=4, steps=3, warmup=1,
                                            input_size=c["input"], heatmap_size=c["heatmap"], K=c["K"], threads=cores)
        sample = "HRNet-W32 heatmap head 384x288 train step (fwd+bwd+AdamW), fp32 PyTorch-CPU oracle, B=4, 1 warm-up + 3 timed steps"
    else:
        pairs = cfg.data.flip_pairs or [(1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12)]
        ips, threads = ots.time_flip_inference(keys["hrformer_base_fusion_k13"], pairs, B=4, steps=3, warmup=1, input_size=c["input"], threads=cores)
        sample = "HRFormer-base fusion K=13 384x288 flip-test inference, fp32 PyTorch-CPU oracle, B=4, 1 warm-up + 3 timed passes"
    return {"value": round(ips, 3), "unit": "images/sec", "cores": threads, "kind": "port", "sample": sample}


def roofline_other(cfg_name, model, B, c):
    """--config hrnet_w32_384 / hrformer_base_infer: the dominant hand-written kernel of that configuration, timed live on its dominant shape."""
    from infantposeestimation_gaussianbias_amd import nnops
    dev = next(model.parameters()).device
    BF = torch.bfloat16
    if cfg_name == "hrnet_w32_384":
        # branch 0 BasicBlock conv3x3 32->32 @96x72 (hrnet.py:24-52): 16 blocks x 2 convs forward + as many data gradients
        H, W, C = 96, 72, 32
        x, w = torch.randn(B, H, W, C, device=dev).to(BF), torch.randn(C, 9, C, device=dev).to(BF)
        sec = time_kernel(lambda: nnops._conv_raw(x, w, C, 3, 1, True))
        return [_entry("k_conv3h<32,32>", f"BasicBlock conv3x3 32->32 @96x72, B={B}, fwd + BN-stat epilogue", "hbm", sec,
                       flops=2.0 * B * H * W * C * 9 * C, bytes_=2.0 * 2 * B * H * W * C)]
    # HRFormer-base twin.  The flip test runs x and flip(x) as ONE batch: every launch sees 2B samples.
    try:
        with open(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_pmc_traffic_wide.json")) as f:
            wide_pmc = json.load(f)["kernels"]        # L2-miss bytes per launch at these shapes (scripts/gpu_pmc_traffic_wide.sh)
    except (OSError, KeyError, ValueError):
        wide_pmc = {}

    def traffic_of(prefix):
        hit = [v["traffic_bytes"] for k, v in wide_pmc.items() if k.startswith(prefix)]
        return hit[0] if len(hit) == 1 else None

    from infantposeestimation_gaussianbias_amd.models.hrformer import HRFormerBlock
    B2 = 2 * B
    H, W, C = 96, 72, 256
    x, w = torch.randn(B2, H, W, C, device=dev).to(BF), torch.randn(C, 9, C, device=dev).to(BF)
    sec = time_kernel(lambda: nnops._conv_raw(x, w, C, 3, 1, False))
    rows = [_entry("k_conv8p", f"head conv3x3 256->256 @96x72, {B2} samples, forward (eval: no statistics)", "mfma", sec,
                   flops=2.0 * B2 * H * W * C * 9 * C, bytes_=2.0 * 2 * B2 * H * W * C)]
    del x, w
    # the forward-only fused halves of the wide branches (k_attn_fwd_w / k_mlp_fwd_w): algorithmic bytes = x read + y written + the weights once
    for Cb, heads, Hb, Wb in ((80, 2, 96, 72), (160, 4, 48, 36)):
        blk = HRFormerBlock(Cb, heads).to(dev).eval()
        blk.c_real, blk.attn_scale = Cb - 2, float(Cb // heads - 1) ** -0.5
        xb = (torch.randn(B2, Hb, Wb, Cb, device=dev) * 1.5).to(BF)
        a, m = blk.attn, blk.mlp
        M = B2 * Hb * Wb
        with torch.no_grad(), nnops.use_weights(blk):
            if nnops.wide_attn_enabled(Cb, heads, B2 * -(-Hb // 7) * -(-Wb // 7)):
                sec = time_kernel(lambda: nnops.attn_half_wide_forward(xb, blk.norm1.weight, blk.norm1.bias, nnops.rel_table(a, heads), a.qkv.weight,
                                                                       a.qkv.bias, a.proj.weight, a.proj.bias, None, heads, blk.c_real, blk.attn_scale))
                rows.append(_entry(f"k_attn_fwd_w (C={Cb})", f"attention half of the block, {M} tokens of C={Cb} ({heads} heads of 40), 7x7 windows", "hbm", sec,
                                   flops=2.0 * M * Cb * 4 * Cb + 4.0 * M * 49 * Cb, bytes_=2.0 * 2 * M * Cb + 2.0 * 4 * Cb * Cb))
                rows[-1]["traffic"] = traffic_of("k_attn_fwd_w<")
            if nnops.wide_mlp_enabled(Cb, 4 * Cb, M):
                sec = time_kernel(lambda: nnops.mlp_half_wide_forward(xb, blk.norm2.weight, blk.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias,
                                                                      None, blk.c_real))
                rows.append(_entry(f"k_mlp_fwd_w (C={Cb})", f"MLP half of the block, {M} tokens of C={Cb}, hidden {4 * Cb}", "hbm", sec,
                                   flops=2.0 * M * Cb * 8 * Cb, bytes_=2.0 * 2 * M * Cb + 2.0 * 8 * Cb * Cb))
                rows[-1]["traffic"] = traffic_of(f"k_mlp_fwd_w<{-(-Cb // 32)}, ")
        del blk, xb
    return rows


class InferRunner:
    """Flip-test inference as one replayed hipGraph (static input buffer); falls back to eager launches when capture is refused."""

    def __init__(self, model, x, pairs):
        self.model, self.x, self.pairs = model, x.clone(), pairs
        self.graph, self.out = None, None
        with torch.no_grad():
            for _ in range(2):
                self.out = model.inference(self.x, flip=True, flip_pairs=pairs)
        torch.cuda.synchronize()
        if os.environ.get("POSE_GRAPH", "1") != "0":
            try:
                from infantposeestimation_gaussianbias_amd import dispatch
                # forward-only fork / join of the branch streams is star-shaped and captures fine (the refused case is backward's
                # side-stream <-> side-stream edges); branches 2-3 of the wide models are launch-latency-bound and fill the bubbles of
                # branches 0-1: cfg 5 23.5 -> 19.2 ms.  POSE_INFER_STREAMS=0: one stream.
                dispatch.set_streams(os.environ.get("POSE_INFER_STREAMS", "1") != "0")
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s), torch.no_grad():
                    self.model.inference(self.x, flip=True, flip_pairs=pairs)       # warm the side stream's allocator pool
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=s):
                        self.out = self.model.inference(self.x, flip=True, flip_pairs=pairs)
                torch.cuda.current_stream().wait_stream(s)
                self.graph = g
            except Exception as e:       # noqa: BLE001  (capture refusals differ by ROCm version; the eager path is always valid)
                log(f"inference capture failed ({type(e).__name__}: {e}); timing eager launches")
                torch.cuda.synchronize()

    def step(self, x):
        if self.graph is not None:
            self.x.copy_(x, non_blocking=True)
            self.graph.replay()
            return self.out
        with torch.no_grad():
            return self.model.inference(x, flip=True, flip_pairs=self.pairs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="hrformer_small", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--eager", action="store_true",
                    help="launch every kernel from the host each step instead of replaying the captured hipGraph")
    ap.add_argument("--single-stream", action="store_true", help="do not run the resolution branches on concurrent HIP streams")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: start N ranks ourselves, one fresh child process per GPU, BEFORE anything in this
        # process touches the GPU (device_count() does not initialise it).  The children print the one JSON line (rank 0).
        import socket
        import subprocess
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with --nproc-per-node {args.gpus} (or without torchrun)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU implementation")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from infantposeestimation_gaussianbias_amd import _lib, dispatch, engine
    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.datasets import synthetic_batch
    from infantposeestimation_gaussianbias_amd.models import build_model

    c = dict(CONFIGS[args.config])
    torch.manual_seed(42)
    cfg = get_config(c["preset"])
    c.update(input=tuple(cfg.data.input_size), heatmap=tuple(cfg.data.heatmap_size), K=cfg.data.num_keypoints)
    B = c["batch"]
    cfg.train.batch_size = B
    model = build_model(cfg).to(dev)
    use_graph = not (args.eager or os.environ.get("POSE_GRAPH", "1") == "0")
    streams = not (args.single_stream or os.environ.get("POSE_STREAMS", "1") == "0")
    if not streams:
        os.environ["POSE_STREAMS"] = "0"
    batch = synthetic_batch(B, c["input"], c["heatmap"], c["K"], cfg.data.sigma, dev, seed=1234 + rank)
    calls_per_step, launch, trainer = None, None, None

    if c["mode"] == "train":
        # default: the whole step (zero_grad + fwd + bwd + gradient exchange + AdamW) is one hipGraph whose branches fork/join across HIP streams
        trainer = engine.Trainer(model, cfg, iters_per_epoch=1000, use_graph=use_graph, graph_warmup=2, graph_streams=streams)
        args.warmup = max(args.warmup, 4) if use_graph else args.warmup      # 2 eager steps + capture + 1 replay before timing
        step = lambda: trainer.step(batch)
    else:
        model.eval()
        dispatch.set_streams(False)
        runner = InferRunner(model, batch["img"], cfg.data.flip_pairs or [(1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12)])
        step = lambda: runner.step(batch["img"])

    comm_check = None
    if world > 1:
        # self-verifying multi-GPU record: RCCL sums a ones-tensor over the ranks before anything is timed, and every rank reports
        # the device it drives (name + PCI bus id), gathered over the same communicator
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        pr = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local, "device": pr.name, "pci_bus_id": getattr(pr, "pci_bus_id", None),
                "pci_device_id": getattr(pr, "pci_device_id", None), "uuid": str(getattr(pr, "uuid", ""))}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        comm_check = {"backend": dist.get_backend(), "rccl_ranks": int(ones.item()), "ranks": gathered}
        if int(ones.item()) != world:
            raise SystemExit(f"bench.py: all_reduce of ones over {world} ranks returned {ones.item()}")

    out = None
    for i in range(args.warmup):
        t_w = time.perf_counter()
        c0 = _lib.CALLS[0]
        out = step()
        if i == 1:
            calls_per_step = _lib.CALLS[0] - c0      # second eager warm-up step: every kernel sequence issued from the host
        if rank == 0 and i < 3:
            torch.cuda.synchronize()
            log(f"warm-up step {i}: {time.perf_counter() - t_w:.3f} s")
    if world > 1:
        dist.barrier()
        if trainer is not None:
            trainer.comm.record_exposed = True       # events around the exchange on the compute stream: exposed communication per step
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if c["mode"] == "train":
        final = float(out["loss"].detach())
        launch = ("hipGraph replay" if trainer._graph is not None else "eager launches") + \
                 (", branches on concurrent HIP streams" if dispatch.streams_enabled() else ", single stream")
        if world > 1:
            launch += f", gradient all-reduce (RCCL) {'inside the graph' if getattr(trainer, '_graph_has_comm', False) else 'between replays'}"
    else:
        kp, sc = out
        final = float(torch.isfinite(kp).all())
        launch = "hipGraph replay" if runner.graph is not None else "eager launches"

    if rank == 0:
        log(f"timed region: {dt:.3f} s for {args.steps} steps -> {B * world * args.steps / dt:.1f} img/s")
        roof = None
        if not args.no_roofline:
            try:
                if world > 1:
                    time.sleep(0.5)          # one watchdog polling period: the timed region's collectives are retired
                table = roofline_table(model, B, trainer) if args.config == "hrformer_small" else roofline_other(args.config, model, B, c)
                roof = dict(table[0])
                roof["table"] = table
            except Exception as e:       # noqa: BLE001  (the probes must never cost the measurement: the line is printed regardless)
                log(f"roofline probes failed ({type(e).__name__}: {e}); reporting the timed region without them")
                table = []
            for e in table:
                log(f"roofline: {e['kernel'][:48]:48s} {e['us_per_launch']:8.1f} us isolated ({e.get('us_in_step_avg', '-')} in step)  "
                    f"{e['achieved']:8.1f} {e['unit']} = {e['frac'] * 100:5.1f} % of {e['bound']} peak")
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.config, c, cfg)
            log(f"cpu baseline: {cpu}")
        line = {
            "metric": c["metric"], "value": round(B * world * args.steps / dt, 2),
            "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": c["workload"], "global_batch": B * world, "per_gpu_batch": B, "parallelism": f"dp{world}", "launch": launch},
            "roofline": roof, "cpu_baseline": cpu, ("final_loss" if c["mode"] == "train" else "outputs_finite"): round(final, 5),
            "hbm_reserved_gb": round(torch.cuda.max_memory_reserved(dev) / 2 ** 30, 2),
            "c_abi_calls_per_step": calls_per_step, "backend": dispatch.backend_name(model),
        }
        if comm_check is not None:
            comm_check["collectives_captured_in_graph"] = bool(c["mode"] == "train" and getattr(trainer, "_graph_has_comm", False))
            comm_check["buckets_launched_from_backward_milestones"] = int(trainer.comm.launched_early) if c["mode"] == "train" else None
            # time per step the compute stream is held by the exchange (bucket launches that did not overlap + the wait for all of them);
            # None when the collectives are captured inside the graph (POSE_GRAPH_COMM=1): then nothing of them is on the host's path
            ex = trainer.comm.exposed_ms() if c["mode"] == "train" else None
            comm_check["exposed_comm_ms_per_step"] = None if ex is None else round(ex, 3)
            comm_check["gradient_bytes_per_step"] = int(trainer.opt.numel * 4) if c["mode"] == "train" else None
            comm_check["buckets"] = len(trainer.comm.buckets) if c["mode"] == "train" else None
            line["comm"] = comm_check
        print(json.dumps(line))
    if world > 1:
        # rank 0 spends seconds on the roofline probes after the timed region: the other ranks wait here instead of tearing the
        # communicator down under it
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
