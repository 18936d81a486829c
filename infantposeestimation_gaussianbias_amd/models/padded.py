"""Padded twins: HIP execution of models whose channel counts are not multiples of 8.

The kernels move 16-byte (8 x bf16) channel chunks, so HRFormer-base (C = 78/156/312/624, head_dim 39 -- hrformer.py:779-825)
and HRNet-W18 (C = 18/36/72/144) cannot be fed to them directly.  Instead of a second (PyTorch-ROCm) code path, such a model
gets a *twin* with 8-aligned shapes (HRFormer-base: C = 80/160/320/640 = heads x 40; W18: 24/40/72/144) whose parameters are
the real ones embedded in zero padding:

  * plain tensors are zero-extended at the end of every dimension;
  * attention weights are head-structured: qkv rows (3, heads, 39->40, C->Cp), proj columns (C->Cp, heads, 39->40);
  * zero weights/biases keep every padded activation channel exactly zero through conv/BN/LN/GELU/attention/residuals;
  * the two places where the padded width would change the arithmetic take the real value explicitly: LayerNorm statistics
    (`c_real`, pk_layernorm_*) and the softmax scale 39^-0.5 (`attn_scale`, pk_window_attn_*).

The real module stays the public surface (state_dict keys/shapes, optimiser, checkpoints).  Before a forward the real
parameters/buffers are embedded into the twin's flat fp32 storage (ONE pk_embed_boxes launch, skipped when nothing changed);
backward kernels store the twin's gradients into a flat twin gradient buffer (gradient sink); after backward ONE launch
extracts the real boxes into the real `.grad` tensors (or the engine's flat gradient buffer), and BatchNorm running statistics
travel back the same way.  Parameters the forward never touches (the dead fuse layers of the last module) keep `grad = None`,
exactly like the reference.
"""
import numpy as np
import torch

from .. import nnops
from .._lib import call, stream_ptr

_EMBED_DTYPE = np.dtype([("real", "<i8"), ("off", "<i8"), ("r", "<i4", (4,)), ("p", "<i4", (4,)), ("numel", "<i8")])


def _up8(n):
    return -(-n // 8) * 8


# ------------------------------------------------------------------------------------------------ twin construction
def _twin_backbone(real):
    from .hrformer import HRFormer, HRFormerBlock
    from .hrnet import HRNet
    if isinstance(real, HRFormer):
        kw = dict(real._ctor)
        for s in (2, 3, 4):
            ch, hd = kw[f"stage{s}_num_channels"], kw[f"stage{s}_num_heads"]
            if any(c % h for c, h in zip(ch, hd)):
                return None
            kw[f"stage{s}_num_channels"] = tuple(h * _up8(c // h) for c, h in zip(ch, hd))
        twin = HRFormer(**kw)
        for m_r, m_t in zip(real.modules(), twin.modules()):
            if isinstance(m_r, HRFormerBlock) and m_t.dim != m_r.dim:
                m_t.c_real = m_r.dim                              # LayerNorm statistics over the real channels
                m_t.attn_scale = float(m_r.dim // m_r.heads) ** -0.5   # q scale of the real head dimension
        return twin
    if isinstance(real, HRNet):
        return HRNet(real.in_channels, real.base_channels, _stage_channels=[_up8(c) for c in real.stage_channels])
    return None


def _build_twin(real):
    from .pose_estimator import PoseEstimator
    if isinstance(real, PoseEstimator):
        bb = _twin_backbone(real.backbone)
        if bb is None:
            return None
        return PoseEstimator.from_backbone(bb, bb.out_channels, real.num_keypoints, real.head_type, real.use_fusion_loss)
    return _twin_backbone(real)


def _box(name, t_real, t_twin, attn_meta):
    """-> (r[4], p[4]) describing how `t_real` sits inside `t_twin`."""
    r, p = list(t_real.shape), list(t_twin.shape)
    prefix = name.rsplit(".attn.", 1)[0] if ".attn." in name else None
    if prefix is not None and prefix in attn_meta and r != p:
        heads, d, dp = attn_meta[prefix]
        if name.endswith("attn.qkv.weight"):
            return [3, heads, d, r[1]], [3, heads, dp, p[1]]
        if name.endswith("attn.qkv.bias"):
            return [1, 3, heads, d], [1, 3, heads, dp]
        if name.endswith("attn.proj.weight"):
            return [1, r[0], heads, d], [1, p[0], heads, dp]
    if len(r) > 4:
        raise ValueError(f"{name}: rank {len(r)} tensors are not supported by the padded twin")
    while len(r) < 4:
        r.insert(0, 1)
        p.insert(0, 1)
    if any(a > b for a, b in zip(r, p)):
        raise ValueError(f"{name}: real shape {t_real.shape} does not fit the twin's {t_twin.shape}")
    return r, p


class _Table:
    """Device-side descriptor table for pk_embed_boxes."""

    def __init__(self, rows, dev):
        self.n = len(rows)
        if not rows:
            return
        desc = np.zeros(len(rows), dtype=_EMBED_DTYPE)
        blk_desc, blk_first, nb = [], [], 0
        for i, (ptr, off, r, p) in enumerate(rows):
            numel = int(np.prod(r))
            desc[i] = (ptr, off, r, p, numel)
            k = -(-numel // 1024)
            blk_desc += [i] * k
            blk_first += [nb] * k
            nb += k
        self.desc = torch.from_numpy(desc.view(np.uint8)).to(dev)
        self.blk_desc = torch.tensor(blk_desc, dtype=torch.int32, device=dev)
        self.blk_first = torch.tensor(blk_first, dtype=torch.int32, device=dev)
        self.nb = nb

    def run(self, base, direction):
        if self.n:
            call("pk_embed_boxes", base, self.desc, self.blk_desc, self.blk_first, self.nb, direction, stream_ptr())


class PaddedTwin:
    def __init__(self, real, twin):
        from .hrformer import HRFormerBlock
        self.real, self.twin = real, twin
        dev = next(real.parameters()).device
        self.device = dev
        twin.to(dev)
        self.attn_meta = {}
        real_mods = dict(real.named_modules())
        for n, m in twin.named_modules():
            if isinstance(m, HRFormerBlock):
                mr = real_mods[n]
                self.attn_meta[n] = (mr.heads, mr.dim // mr.heads, m.dim // m.heads)
        # ---- flat fp32 storage of the twin: [parameters | float buffers], zero-initialised (= the padding)
        self.rp, self.tp = dict(real.named_parameters()), dict(twin.named_parameters())
        self.rb = {k: v for k, v in real.named_buffers() if v.is_floating_point()}
        self.tb = {k: v for k, v in twin.named_buffers() if v.is_floating_point()}
        if set(self.rp) != set(self.tp) or set(self.rb) != set(self.tb):
            raise RuntimeError("padded twin: parameter/buffer names of the twin differ from the real model")
        self.off, off = {}, 0
        for k, t in list(self.tp.items()) + [("#" + k, t) for k, t in self.tb.items()]:
            self.off[k] = off
            off += -(-t.numel() // 4) * 4
        self.n_param_elems = self.off["#" + next(iter(self.tb))] if self.tb else off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.n_param_elems, dtype=torch.float32, device=dev)
        for k, t in self.tp.items():
            o = self.off[k]
            t.data = self.flat[o:o + t.numel()].view_as(t)
            t._pk_grad_sink = self.grad[o:o + t.numel()].view_as(t)       # backward kernels store the gradient here
            t._pk_used = False
        for k, t in self.tb.items():
            o = self.off["#" + k]
            t.data = self.flat[o:o + t.numel()].view_as(t)
        self.boxes = {k: _box(k, self.rp[k], self.tp[k], self.attn_meta) for k in self.tp}
        self.boxes.update({"#" + k: _box(k, self.rb[k], self.tb[k], self.attn_meta) for k in self.tb})
        self._sig_params = self._sig_bufs = self._sig_grads = None
        self._t_params = self._t_bufs = self._t_over = self._t_acc = None
        self._dirty = True
        self._pending = False
        self.nbt = [(dict(real.named_buffers())[k], v) for k, v in twin.named_buffers() if k.endswith("num_batches_tracked")]

    # ---- real -> twin
    def mark_dirty(self):
        self._dirty = True

    def sync_to_twin(self):
        sp = tuple((t.data_ptr(), t._version) for t in self.rp.values())
        sb = tuple((t.data_ptr(), t._version) for t in self.rb.values())
        ptrs_p, ptrs_b = tuple(a for a, _ in sp), tuple(a for a, _ in sb)
        if self._t_params is None or ptrs_p != tuple(a for a, _ in self._sig_params):
            self._t_params = _Table([(self.rp[k].data_ptr(), self.off[k], *self.boxes[k]) for k in self.tp], self.device)
            self._dirty = True
        if self._t_bufs is None or ptrs_b != tuple(a for a, _ in self._sig_bufs):
            self._t_bufs = _Table([(self.rb[k].data_ptr(), self.off["#" + k], *self.boxes["#" + k]) for k in self.tb], self.device)
            self._dirty = True
        if self._dirty or sp != self._sig_params or sb != self._sig_bufs:
            self._t_params.run(self.flat, 0)
            self._t_bufs.run(self.flat, 0)
            for r, t in self.nbt:
                t.copy_(r)
            nnops.weight_cache(self.twin).mark_dirty()
            self._sig_params, self._sig_bufs, self._dirty = sp, sb, False
        if self.twin.training != self.real.training:
            self.twin.train(self.real.training)
        rb, tb = getattr(self.real, "backbone", self.real), getattr(self.twin, "backbone", self.twin)
        if hasattr(rb, "drop_path_rate"):
            tb.drop_path_rate = rb.drop_path_rate

    # ---- twin -> real
    def buffers_to_real(self):
        """BatchNorm running statistics updated by a training-mode forward of the twin."""
        if self._t_bufs is not None:
            self._t_bufs.run(self.flat, 1)
            for r, t in self.nbt:
                r.copy_(t)
            # the real buffers changed through a raw pointer: re-read their versions so the next sync does not re-embed
            self._sig_bufs = tuple((t.data_ptr(), t._version) for t in self.rb.values())

    def grads_to_real(self):
        """Scatter the twin's gradients into the real parameters' gradients.  Parameters carrying a gradient sink
        (engine.FlatAdamW) are overwritten there; otherwise `.grad` is accumulated into (allocated when None), like autograd."""
        if not self._pending:
            return
        self._pending = False
        nnops.finalize_deferred()                # postponed slab reductions write into the twin's gradient buffer first
        used = [k for k, t in self.tp.items() if t._pk_used]
        dst = {}
        for k in used:
            p = self.rp[k]
            sink = nnops.grad_sink_of(p)
            if sink is not None:
                dst[k] = (sink, 1)
            else:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                dst[k] = (p.grad, 2)
        sig = tuple((k, d.data_ptr(), m) for k, (d, m) in dst.items())
        if sig != self._sig_grads:
            self._t_over = _Table([(d.data_ptr(), self.off[k], *self.boxes[k]) for k, (d, m) in dst.items() if m == 1], self.device)
            self._t_acc = _Table([(d.data_ptr(), self.off[k], *self.boxes[k]) for k, (d, m) in dst.items() if m == 2], self.device)
            self._sig_grads = sig
        self._t_over.run(self.grad, 1)
        self._t_acc.run(self.grad, 2)

    # ---- forward through the twin
    def run(self, *args, **kwargs):
        self.sync_to_twin()
        if torch.is_grad_enabled():
            nnops.begin_grad_epoch()             # the twin's gradient sinks are extracted after every backward and may be stored to again
        out = self.twin(*args, **kwargs)
        if self.real.training:
            self.buffers_to_real()
        if torch.is_grad_enabled():
            self._pending = True
            root = out.get("loss") if isinstance(out, dict) else out
            if root is None and isinstance(out, dict):
                root = next((v for v in out.values() if torch.is_tensor(v) and v.requires_grad), None)
            if root is not None and root.requires_grad:
                root.register_hook(self._queue_extract)
        return out

    def _queue_extract(self, grad):
        # runs when backward reaches the output: ask the engine to call us once the whole backward pass has finished
        torch.autograd.Variable._execution_engine.queue_callback(self.grads_to_real)
        return grad


def twin_for(model):
    """The padded twin of `model`, or None when the HIP kernels take the model as it is (or no twin can be built)."""
    state = getattr(model, "_pk_twin", None)
    if state is False:
        return None
    dev = next(model.parameters()).device
    if state is not None and state.device == dev:
        return state
    if dev.type != "cuda":
        return None                                            # decided again once the model lives on the GPU
    if nnops.supported(model):
        object.__setattr__(model, "_pk_twin", False)
        return None
    twin = _build_twin(model)
    if twin is None or not nnops.supported(twin):
        object.__setattr__(model, "_pk_twin", False)
        return None
    state = PaddedTwin(model, twin)
    object.__setattr__(model, "_pk_twin", state)          # plain attribute: the twin must not appear in state_dict()/parameters()
    return state
