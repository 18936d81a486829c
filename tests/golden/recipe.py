"""Deterministic synthetic weights shared by make_golden.py and the parity tests.

The golden fixtures cannot ship model weights (46 MB for HRFormer-small), so both the
capture script (which loads them into the *reference* modules) and the tests (which load
them into this repo's modules / oracle) regenerate every tensor from its state_dict key:
`seed = crc32(key) ^ salt` -> numpy Generator -> values chosen by the tensor's role.

This file is test data plumbing only (no reference code, no product code).
"""
import zlib

import numpy as np


def _rng(key: str, salt: int) -> np.random.Generator:
    return np.random.default_rng((zlib.crc32(key.encode()) ^ (salt * 2654435761)) & 0xFFFFFFFF)


def synth_tensor(key: str, shape, dtype: str = "float32", salt: int = 0) -> np.ndarray:
    """One tensor of a state_dict, chosen by the role its key names."""
    shape = tuple(int(s) for s in shape)
    g = _rng(key, salt)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "relative_position_index":
        ws = int(round(shape[0] ** 0.5))
        ys, xs = np.meshgrid(np.arange(ws), np.arange(ws), indexing="ij")
        ys, xs = ys.reshape(-1), xs.reshape(-1)
        idx = (ys[:, None] - ys[None, :] + ws - 1) * (2 * ws - 1) + (xs[:, None] - xs[None, :] + ws - 1)
        return idx.astype(np.int64)
    if leaf == "running_mean":
        return (0.1 * g.standard_normal(shape)).astype(np.float32)
    if leaf == "running_var":
        return g.uniform(0.5, 1.5, shape).astype(np.float32)
    if leaf == "relative_position_bias_table":
        return (0.5 * g.standard_normal(shape)).astype(np.float32)
    if leaf in ("alpha", "fusion_weight"):
        return np.asarray(g.uniform(-0.5, 1.0), dtype=np.float32).reshape(shape)
    if leaf == "bias":
        return (0.05 * g.standard_normal(shape)).astype(np.float32)
    if leaf == "weight":
        if len(shape) == 1:  # BatchNorm / LayerNorm scale
            return g.uniform(0.6, 1.4, shape).astype(np.float32)
        if len(shape) == 4:  # conv OIHW: he-like so activations stay O(1)
            fan_in = shape[1] * shape[2] * shape[3]
            return (g.standard_normal(shape) * np.sqrt(1.0 / fan_in)).astype(np.float32)
        if len(shape) == 2:  # linear (out, in)
            return (g.standard_normal(shape) * np.sqrt(1.0 / shape[1])).astype(np.float32)
    return (0.1 * g.standard_normal(shape)).astype(np.float32)


def synth_state_dict(spec: dict, salt: int = 0) -> dict:
    """spec: {key: [shape, dtype]} -> {key: ndarray}."""
    return {k: synth_tensor(k, v[0], v[1], salt) for k, v in spec.items()}


def spec_of(state_dict) -> dict:
    """{key: [shape, dtype-name]} of a torch state_dict."""
    return {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in state_dict.items()}


def synth_input(tag: str, shape, scale: float = 1.0, salt: int = 0) -> np.ndarray:
    return (scale * _rng("input:" + tag, salt).standard_normal(tuple(shape))).astype(np.float32)
