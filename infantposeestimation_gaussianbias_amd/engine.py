"""Training engine (S1): flat parameter storage, fused AdamW, per-iteration LR schedule, data-parallel gradient exchange.

MI355X-first layout: all parameters live in ONE fp32 buffer, all gradients in another (each `param.data` / `param.grad`
is a view), so
  * the optimiser is one HIP launch over the whole model (pk_adamw_step) instead of ~820 per-tensor updates,
  * the gradient all-reduce runs on a few large contiguous slices (RCCL over xGMI is per-link bound: few, big messages),
    issued from autograd hooks while backward is still producing earlier layers' gradients,
  * averaging over ranks is folded into the optimiser kernel (grad_scale = 1/world).
Semantics follow train.py:55-128: AdamW(lr 5e-4, betas (.9,.999), wd .01 on the "decay" group), parameters whose name
contains 'bias' / 'bn' / 'norm' are not decayed; parameters that never receive a gradient (41 tensors for
HRFormer-small) are left untouched exactly as torch.optim.AdamW skips `grad is None`.
"""
import math
import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import hipops, nnops


def is_no_decay(name: str) -> bool:
    return "bias" in name or "bn" in name or "norm" in name          # train.py:66


def lr_factor(it: int, iters_per_epoch: int, cfg_train) -> float:
    """LambdaLR factor, stepped per iteration: linear warm-up over warmup_epochs, then x gamma at each milestone epoch."""
    warm = cfg_train.warmup_epochs * iters_per_epoch
    if it < warm:
        r = cfg_train.warmup_lr / cfg_train.lr
        return r + (1 - r) * it / warm
    f = 1.0
    for m in cfg_train.lr_milestones:
        if it >= m * iters_per_epoch:
            f *= cfg_train.lr_gamma
    return f


class FlatAdamW:
    """Owns the flat buffers of `model` and applies the fused optimiser step."""

    ALIGN = 4            # elements: every tensor starts 16-byte aligned inside the flat buffers

    def __init__(self, model: torch.nn.Module, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, direct_grads=False):
        self.model = model
        self.direct_grads = direct_grads
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError("model has no trainable parameters")
        dev = named[0][1].device
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += -(-p.numel() // self.ALIGN) * self.ALIGN
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        flags = torch.zeros(off, dtype=torch.uint8)
        for n, p, o in zip(self.names, self.params, self.offsets):
            self.flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + p.numel()].view_as(p)
            flags[o:o + p.numel()] = 2 | (0 if is_no_decay(n) else 1)
        self.flags = flags.to(dev)
        self.active = [True] * len(self.params)
        self._grads_installed = False
        self._sinks_armed = False          # True once a whole step has run with the gradient sinks installed
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.lr_dev = torch.full((1,), lr, dtype=torch.float32, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.step_count = 0

    # -- gradient storage ---------------------------------------------------------------------------------
    def grad_view(self, i):
        p, o = self.params[i], self.offsets[i]
        return self.grad[o:o + p.numel()].view_as(p)

    def install_grad_views(self):
        """Adopt the gradients autograd produced on the first backward: parameters still without a gradient are the
        structurally unused ones -> marked inactive (flag bit1 cleared) so the optimiser never touches them."""
        inactive = []
        for i, p in enumerate(self.params):
            v = self.grad_view(i)
            if p.grad is None:
                self.active[i] = False
                inactive.append(i)
                v.zero_()
            else:
                v.copy_(p.grad)
            p.grad = v
        if inactive:
            fl = self.flags.cpu()
            for i in inactive:
                o, n = self.offsets[i], self.params[i].numel()
                fl[o:o + n] &= 0xFD
            self.flags.copy_(fl)
        self._grads_installed = True
        if self.direct_grads:
            # backward kernels now store parameter gradients straight into these views (no autograd accumulation)
            for i, p in enumerate(self.params):
                p._pk_grad_sink = self.grad_view(i)

    def zero_grad(self):
        if self._grads_installed:
            if not self.direct_grads:          # with the gradient sink every active gradient is overwritten, not accumulated
                self.grad.zero_()
            else:
                nnops.begin_grad_epoch()       # a sink may be stored to once per epoch (checked in nnops.grad_sink_of)
        else:
            for p in self.params:
                p.grad = None

    # -- update ------------------------------------------------------------------------------------------------
    def set_lr(self, lr: float):
        self.lr = lr
        self.lr_dev.fill_(lr)

    def step(self, grad_scale: float = 1.0):
        if self.flat.is_cuda:
            nnops.finalize_deferred()          # postponed parameter-gradient reductions (no-op when the Trainer already ran them)
        if not self._grads_installed:
            self.install_grad_views()
        elif self.direct_grads and self._sinks_armed:
            # direct-store sinks are not zeroed between steps: an active parameter whose sink no backward kernel asked for in
            # this step (a batch that skipped a branch) must not have last step's gradient applied again
            for i, p in enumerate(self.params):
                if self.active[i] and not nnops.sink_written(p):
                    self.grad_view(i).zero_()
        self._sinks_armed = self.direct_grads
        self.step_count += 1
        self.step_dev.add_(1)
        hipops.adamw_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.flags, None, self.lr_dev, self.step_dev,
                          self.betas[0], self.betas[1], self.eps, self.weight_decay, grad_scale)
        wc = getattr(self.model, "_pk_weight_cache", None)
        if wc is not None:
            wc.mark_dirty()          # the kernel rewrote the fp32 masters behind torch's back: bf16 copies are stale
        tw = getattr(self.model, "_pk_twin", None)
        if tw:
            tw.mark_dirty()          # ... and so is the padded twin's embedded copy

    # -- checkpoint format of the reference (train.py:351-357): per-parameter state keyed by index ---------------
    def state_dict(self):
        decay = [i for i, n in enumerate(self.names) if not is_no_decay(n)]
        nodecay = [i for i, n in enumerate(self.names) if is_no_decay(n)]
        order = decay + nodecay
        remap = {old: new for new, old in enumerate(order)}
        state = {}
        for i, p in enumerate(self.params):
            if not self.active[i] or self.step_count == 0:
                continue
            o, n = self.offsets[i], p.numel()
            state[remap[i]] = {"step": torch.tensor(float(self.step_count)),
                               "exp_avg": self.exp_avg[o:o + n].view_as(p).clone(),
                               "exp_avg_sq": self.exp_avg_sq[o:o + n].view_as(p).clone()}
        common = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "amsgrad": False, "maximize": False,
                  "foreach": None, "capturable": False, "differentiable": False, "fused": None, "initial_lr": self.lr}
        groups = [dict(common, weight_decay=self.weight_decay, params=[remap[i] for i in decay]),
                  dict(common, weight_decay=0.0, params=[remap[i] for i in nodecay])]
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        decay = [i for i, n in enumerate(self.names) if not is_no_decay(n)]
        nodecay = [i for i, n in enumerate(self.names) if is_no_decay(n)]
        order = decay + nodecay
        steps = 0
        for new, old in enumerate(order):
            st = sd["state"].get(new)
            if st is None:
                continue
            o, n = self.offsets[old], self.params[old].numel()
            self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps = max(steps, int(float(st["step"])))
        self.step_count = steps
        self.step_dev.fill_(steps)
        if sd.get("param_groups"):
            self.set_lr(float(sd["param_groups"][0]["lr"]))


class WarmupMultiStepLR:
    """Per-iteration schedule of train.py:100-128 with LambdaLR's state_dict fields the reference checkpoints carry."""

    def __init__(self, optimizer: FlatAdamW, cfg_train, iters_per_epoch: int):
        self.opt, self.cfg, self.ipe = optimizer, cfg_train, iters_per_epoch
        self.base_lr = cfg_train.lr
        self.last_epoch = 0
        optimizer.set_lr(self.base_lr * lr_factor(0, iters_per_epoch, cfg_train))

    def step(self):
        self.last_epoch += 1
        self.opt.set_lr(self.base_lr * lr_factor(self.last_epoch, self.ipe, self.cfg))

    def get_last_lr(self):
        return [self.opt.lr, self.opt.lr]

    def state_dict(self):
        return {"base_lrs": [self.base_lr, self.base_lr], "last_epoch": self.last_epoch, "_step_count": self.last_epoch + 1,
                "_last_lr": self.get_last_lr(), "lr_lambdas": [None]}

    def load_state_dict(self, sd):
        self.last_epoch = int(sd["last_epoch"])
        self.opt.set_lr(self.base_lr * lr_factor(self.last_epoch, self.ipe, self.cfg))


class GradientExchange:
    """Data-parallel gradient sum over ranks on the flat gradient buffer (RCCL `nccl` backend on GPUs, gloo on CPU).

    Buckets are contiguous slices of the flat buffer, built from the END (backward produces the last layers first).  A bucket's
    all-reduce is launched asynchronously (RCCL runs it on its own stream) as soon as every ACTIVE parameter in it has its final
    gradient of the step, so the exchange of the head's / late stages' gradients overlaps with the rest of backward:
      * gradient-sink mode (FlatAdamW(direct_grads=True), the Trainer's default): `flush_ready()` is called from backward
        MILESTONES (nnops.backward_milestone: hooks on the stage-boundary tensors, fired when backward has passed them) after the
        postponed slab reductions registered so far have been run; readiness = the parameter's sink was stored to this step;
      * plain autograd mode (`overlap=True`): post-accumulate hooks count a bucket's gradients.
    `finish()` launches whatever is left and waits.  Parameters that never receive a gradient (41 tensors for HRFormer-small,
    SURVEY §8e) are skipped consistently on all ranks (the active set is structural)."""

    def __init__(self, opt: FlatAdamW, bucket_mb: float = 16.0, group=None, overlap: bool = False):
        self.opt, self.group, self.overlap = opt, group, overlap
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
        self.buckets: List[Dict] = []
        self._hooks = []
        self._pending = []
        self._armed = False
        self.suspended = False          # True while a hipGraph without collectives is being captured: milestones must not launch any
        self.launched_early = 0         # buckets launched from milestones / hooks in the last step (observability, tests)
        # exposed communication: time the COMPUTE stream spends in finish() -- launching the buckets that did not overlap with backward
        # and waiting for all of them -- measured with events on that stream when `record_exposed` is set (bench.py --gpus N reports it)
        self.record_exposed = False
        self._exposed = []

    def broadcast_initial_state(self):
        if self.world == 1:
            return
        dist.broadcast(self.opt.flat, 0, group=self.group)
        for b in self.opt.model.buffers():
            dist.broadcast(b, 0, group=self.group)

    def _build(self):
        opt = self.opt
        hi, cur, members = opt.numel, opt.numel, []
        plan = []
        for i in reversed(range(len(opt.params))):
            members.append(i)
            cur = opt.offsets[i]
            if hi - cur >= self.bucket_elems or i == 0:
                plan.append((cur, hi, members))
                hi, members = cur, []
        self.buckets = [{"lo": lo, "hi": hi_, "members": [i for i in m if opt.active[i]], "need": sum(1 for i in m if opt.active[i]), "got": 0}
                        for lo, hi_, m in plan]
        owner = {}
        for b, (_, _, m) in enumerate(plan):
            for i in m:
                owner[i] = b
        if self.overlap and not opt.direct_grads:
            for i, p in enumerate(opt.params):
                if opt.active[i]:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(owner[i])))
        self._armed = True

    def _make_hook(self, b):
        def hook(_param):
            bk = self.buckets[b]
            bk["got"] += 1
            if bk["got"] == bk["need"] and not self.suspended:
                self.launched_early += 1
                self._launch(bk)
        return hook

    def _launch(self, bk):
        bk["got"] = -1     # launched
        self._pending.append(dist.all_reduce(self.opt.grad[bk["lo"]:bk["hi"]], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def flush_ready(self):
        """Gradient-sink mode: launch, in backward order, every bucket whose active parameters have all been stored to this step.
        The caller has run nnops.finalize_deferred() first, so a stored-to sink holds its final value."""
        if self.world == 1 or not self._armed or self.suspended or not self.opt.direct_grads:
            return
        from . import nnops as _nn
        for bk in self.buckets:
            if bk["got"] == -1:
                continue
            if not all(_nn.sink_written(self.opt.params[i]) for i in bk["members"]):
                break               # buckets are ordered by backward time: later ones cannot be complete either
            self.launched_early += 1
            self._launch(bk)

    def finish(self):
        """Call after backward: flush buckets that did not fire (first step, or only inactive members), wait for all."""
        if self.world == 1:
            return
        if not self.opt._grads_installed:
            self.opt.install_grad_views()
        if not self._armed:
            self._build()
        timed = self.record_exposed and self.opt.grad.is_cuda and not torch.cuda.is_current_stream_capturing()
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for bk in self.buckets:
            if bk["got"] != -1:
                self._launch(bk)
        for w in self._pending:
            w.wait()
        self._pending.clear()
        if timed:
            e1.record()
            self._exposed.append((e0, e1))
        for bk in self.buckets:
            bk["got"] = 0

    def exposed_ms(self):
        """Mean time per step the compute stream was held by the gradient exchange since the last call (None: nothing recorded, e.g.
        the collectives live inside a captured graph).  Synchronises."""
        if not self._exposed:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._exposed]
        self._exposed = []
        return sum(ms) / len(ms)

    @property
    def grad_scale(self):
        return 1.0 / self.world


class Trainer:
    """One optimisation step of train.py::train_one_epoch (:169-187) without its per-step host syncs.

    `use_graph=True` captures zero_grad + forward + backward + gradient exchange + optimiser of a fixed-shape batch into a hipGraph
    after `graph_warmup` eager steps and replays it: ~1 300 kernel launches per step become one graph launch, which removes the
    host as the bottleneck.  With N>1 ranks over RCCL the bucketed all-reduces are captured too: they are issued from backward
    milestones on RCCL's own stream, so inside the replayed graph they run beside the remaining backward kernels, and AdamW follows
    in the same graph.  When the collectives cannot be captured (gloo, or a capture the runtime refuses) the graph holds
    zero_grad + forward + backward only and the exchange + AdamW run between replays.  The LR is written to a device scalar before
    each replay, so the per-iteration schedule advances normally.  Batches are copied into static input buffers."""

    def __init__(self, model, cfg, iters_per_epoch: int, bucket_mb: float = 8.0, use_graph: bool = False, graph_warmup: int = 3,
                 overlap_comm: bool = True, graph_streams: bool = False):
        self.model, self.cfg = model, cfg
        self.overlap_comm = overlap_comm
        # concurrent branch streams: on for eager steps; inside a captured graph only on request (fork/join capture across
        # streams, see scripts/gpu_graph_streams.py)
        self._want_streams = graph_streams if use_graph else True
        self._region = graph_streams               # with use_graph=False: eager launches, same autograd structure as the graph
        t = cfg.train
        if t.optimizer != "AdamW":
            raise ValueError(f"Unknown optimizer: {t.optimizer}")   # the fused kernel implements the reference default only
        # gradients are stored directly by the backward kernels (gradient sinks); the exchange is driven by backward milestones
        self.opt = FlatAdamW(model, t.lr, tuple(t.betas), 1e-8, t.weight_decay, direct_grads=True)
        self.sched = WarmupMultiStepLR(self.opt, t, iters_per_epoch)
        self.comm = GradientExchange(self.opt, bucket_mb, overlap=overlap_comm)
        self.comm.broadcast_initial_state()
        self._ms_cb = None
        if self.comm.world > 1 and overlap_comm:
            self._ms_cb = self._milestone
        # (Measured and dropped for one GPU: running the postponed slab reductions of the finished stages on an auxiliary stream at
        # the same milestones -- 20.29 -> 20.79 ms per step; the reduction competes with the backward kernels for HBM and queues.
        # Round 3, same experiment on the 17.2 ms step: 17.0-17.4 ms without, 17.5-17.8 ms with, four alternating runs each.)
        # >= 2 eager steps before capture: step 1 installs the gradient sinks, step 2 builds the descriptor tables that depend on
        # them (deferred reductions, padded-twin extraction) -- table uploads are host->device copies and cannot be captured
        self.use_graph, self.graph_warmup = use_graph, max(2, graph_warmup)
        # autograd binds each AccumulateGrad node to the stream that was current when it was created; warm-up and
        # capture therefore run on ONE dedicated side stream, otherwise backward would sync with the (non-capturing)
        # default stream in the middle of the capture
        self._stream = torch.cuda.Stream() if (use_graph and torch.cuda.is_available()) else None
        self._eager_steps = 0
        self._graph = None
        self._static = None
        self._static_out = None
        self._graph_has_opt = self._graph_has_comm = False

    def _milestone(self):
        """Backward has passed a stage boundary: everything registered so far belongs to layers whose backward is complete."""
        if not self.comm._armed:
            return
        # (also while a forward+backward-only graph is being captured: the partial reduction tables must be the ones the eager
        # warm-up steps built -- a new table cannot be uploaded during capture; only the collectives are held back)
        if not self._region:
            # eager branch streams: the hook fires on the stream of whichever boundary producer ran last, while the weight-gradient
            # kernels of the other branches were enqueued on side streams autograd does not order against this one.  (Region mode
            # joins every side stream at the end of each region's backward, so nothing is outstanding there.)
            from . import dispatch
            dispatch.join_side_streams(torch.cuda.current_stream())
        nnops.finalize_deferred()
        self.comm.flush_ready()

    # -- eager path --------------------------------------------------------------------------------------------
    def _fwd_bwd(self, batch):
        nnops._PENDING.clear()         # reductions registered by a step that did not finish (exception) must not leak into this one
        self.opt.zero_grad()
        # the device input pipeline hands over the stem's own layout (bf16 NHWC, 8-channel pixels): no conversion launch
        x = batch["img_nhwc8"] if batch.get("img_nhwc8") is not None else batch["img"]
        out = self.model(x, batch["target"], batch["target_weight"], gt_keypoints=batch.get("keypoints"),
                         input_size=self.cfg.data.input_size)
        out["loss"].backward()
        nnops.finalize_deferred()      # ONE launch: all postponed slab reductions of parameter gradients (the rest of them after milestones)
        tw = getattr(self.model, "_pk_twin", None)
        if tw:
            tw.grads_to_real()         # padded twin: no-op when the end-of-backward callback already extracted the gradients
        return out

    def _eager_step(self, batch):
        if not self.model.training:
            self.model.train()
        self.comm.launched_early = 0
        out = self._fwd_bwd(batch)
        self.comm.finish()
        self.opt.step(self.comm.grad_scale)
        self.sched.step()
        return out

    # -- graph path --------------------------------------------------------------------------------------------
    def _comm_capturable(self):
        return (self.comm.world > 1 and dist.get_backend(self.comm.group) == "nccl" and os.environ.get("POSE_GRAPH_COMM", "0") == "1")

    def _capture(self, batch):
        keys = [k for k in ("img_nhwc8" if batch.get("img_nhwc8") is not None else "img", "target", "target_weight", "keypoints")
                if batch.get(k) is not None]
        self._static = {k: batch[k].clone() for k in keys}
        world = self.comm.world
        attempts = [True, False] if self._comm_capturable() else [world == 1]
        for whole_step in attempts:
            torch.cuda.synchronize()
            mode = "global"
            if world > 1:
                # The process group's watchdog thread polls the events of outstanding collectives (cudaEventQuery); under the
                # default global capture mode such a call from another thread invalidates the capture.  All collectives have
                # completed (synchronize above); give the watchdog one polling period to retire them, and only police the
                # capturing threads' own calls.
                import time
                time.sleep(0.5)
                mode = "thread_local"
            self.comm.suspended = world > 1 and not whole_step      # no collectives inside a forward+backward-only capture
            steps_before = self.opt.step_count
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, stream=self._stream, capture_error_mode=mode):
                    out = self._fwd_bwd(self._static)
                    if whole_step:
                        self.comm.finish()
                        self.opt.step(self.comm.grad_scale)
            except Exception as e:      # noqa: BLE001  (a refused RCCL capture must not end the run: fall back to exchange between replays)
                if not (whole_step and world > 1):
                    raise
                import sys
                print(f"# engine: capturing the step with its RCCL all-reduces failed ({type(e).__name__}: {e}); capturing forward+backward only",
                      file=sys.stderr, flush=True)
                self.opt.step_count = steps_before
                self.comm._pending.clear()
                for bk in self.comm.buckets:
                    bk["got"] = 0
                nnops._PENDING.clear()
                continue
            finally:
                self.comm.suspended = False
            if whole_step:
                self.opt.step_count -= 1          # the captured optimiser launch did not execute; _graph_step counts the replays
            self._graph, self._static_out = g, out
            self._graph_has_opt, self._graph_has_comm = whole_step, whole_step and world > 1
            break
        nnops.freeze_workspaces(self.model)   # the graph holds the slab-workspace pointers: they must not be reallocated

    def _graph_step(self, batch):
        for k, v in self._static.items():
            v.copy_(batch[k], non_blocking=True)
        self._graph.replay()
        if self._graph_has_opt:
            self.opt.step_count += 1          # the captured pk_adamw_step already advanced the device-side counter
            # ... and rewrote the fp32 masters behind torch's back: the bf16 compute copies (and a padded twin's embedded
            # parameters) are one step behind until the next forward repacks them
            wc = getattr(self.model, "_pk_weight_cache", None)
            if wc is not None:
                wc.mark_dirty()
            tw = getattr(self.model, "_pk_twin", None)
            if tw:
                tw.mark_dirty()
                twc = getattr(tw.twin, "_pk_weight_cache", None)
                if twc is not None:
                    twc.mark_dirty()
        else:
            self.comm.finish()
            self.opt.step(self.comm.grad_scale)
        self.sched.step()
        return self._static_out

    def step(self, batch):
        from . import dispatch
        dispatch.set_streams(self._want_streams)
        dispatch.set_region_mode(self._region)    # capturable fork/join (one autograd node per parallel region)
        nnops.set_milestone_callback(self._ms_cb)  # process-wide hook: the trainer that steps owns it
        if not self.use_graph:
            return self._eager_step(batch)
        if self._graph is None:
            if self._eager_steps < self.graph_warmup:
                self._eager_steps += 1
                self._stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._stream):
                    out = self._eager_step(batch)
                torch.cuda.current_stream().wait_stream(self._stream)
                return out
            self.model.train()
            self._capture(batch)
            # the capture itself does not execute: run this step through the graph
        if any(k not in batch or batch[k] is None or batch[k].shape != v.shape for k, v in self._static.items()):
            # a batch the captured graph was not recorded for (e.g. the short last batch of an epoch): launch it eagerly on the
            # capture stream, with the same autograd structure and the same optimiser path as the replayed steps
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                out = self._eager_step(batch)
            torch.cuda.current_stream().wait_stream(self._stream)
            return out
        return self._graph_step(batch)
