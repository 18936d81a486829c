#!/usr/bin/env python3
"""Fingerprint of the code a committed profile was taken with: sha256 over the kernel sources, their build flags and the host modules that decide which
kernels are launched.  `python scripts/profile_meta.py write <out.json> "<command>"` stores it next to the profile; bench.py recomputes it
and marks the in-step fields it reads from profiles/ as stale (null) when it differs."""
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "infantposeestimation_gaussianbias_amd")


def fingerprint():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")) + glob.glob(os.path.join(PKG, "csrc", "*.h")) + [os.path.join(PKG, "csrc", "Makefile")] +
                   [os.path.join(PKG, f) for f in ("nnops.py", "exchange.py", "dispatch.py", "engine.py")] +
                   glob.glob(os.path.join(PKG, "models", "*.py")))
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "write":
        with open(sys.argv[2], "w") as f:
            json.dump({"code_sha16": fingerprint(), "command": sys.argv[3] if len(sys.argv) > 3 else None}, f, indent=1)
    print(fingerprint())
