"""ctypes binding of libposekernels.so (the C-ABI declared in include/posekernels.h).

There is NO fallback: if the shared library is missing or a symbol cannot be resolved this module
raises, and every op in the package fails with it.  Build with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C infantposeestimation_gaussianbias_amd/csrc`.
"""
import ctypes
import os
import re

# PyTorch's wheel carries its own libamdhip64.so; libposekernels.so is linked against the SONAME only.  Whichever is loaded first
# serves both: with torch first there is ONE HIP runtime in the process (the one that owns torch's device context and streams).
# The other order loads /opt/rocm's copy for the kernels and torch's for everything else, and every launch then fails with
# "no ROCm-capable device is detected" (seen with build() + smoke() in one process).
import torch  # noqa: F401  (must precede the CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
# POSE_KERNELS_LIB: another build of the same library (A/B measurements of two builds on one box); it must export every declared symbol
LIB_PATH = os.environ.get("POSE_KERNELS_LIB") or os.path.join(_HERE, "csrc", "libposekernels.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "posekernels.h")

_CT = {
    "int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double, "int64_t": ctypes.c_int64,
    "int32_t": ctypes.c_int32,
}


class PoseKernelError(RuntimeError):
    pass


def declared_symbols(header_path=HEADER_PATH):
    """Parse `int pk_xxx(...)` prototypes out of the public header -> {name: [ctypes argtypes]}."""
    with open(header_path) as f:
        src = re.sub(r"/\*.*?\*/", " ", f.read(), flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|const char\s*\*)\s+(pk_\w+)\s*\(([^)]*)\)\s*;", src):
        args = []
        body = m.group(3).strip()
        if body and body != "void":
            for a in body.split(","):
                a = a.strip()
                if "*" in a:
                    args.append(ctypes.c_void_p)
                else:
                    ty = a.replace("const", "").split()[0]
                    args.append(_CT[ty])
        out[m.group(2)] = (m.group(1), args)
    return out


def _load():
    if not os.path.exists(LIB_PATH):
        raise PoseKernelError(
            f"{LIB_PATH} not found: the HIP extension is not built. There is no CPU/PyTorch fallback for the hot path; "
            "run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc, cross-compiles gfx950 without a GPU).")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (ret, args) in declared_symbols().items():
        fn = getattr(lib, name)          # AttributeError here == header/library mismatch: fail loudly
        fn.argtypes = args
        fn.restype = ctypes.c_char_p if "char" in ret else ctypes.c_int
    return lib


lib = _load()


CALLS = [0]      # number of C-ABI calls made so far (bench.py reports the per-step count; one call = one to three kernel launches)


def call(name, *args):
    """Invoke a pk_* entry; tensors are passed as data_ptr(), None as NULL. Raises on any non-zero status."""
    CALLS[0] += 1
    conv = []
    for a in args:
        if a is None:
            conv.append(None)
        elif hasattr(a, "data_ptr"):
            conv.append(a.data_ptr())
        else:
            conv.append(a)
    rc = getattr(lib, name)(*conv)
    if rc != 0:
        raise PoseKernelError(f"{name} failed (code {rc}): {lib.pk_last_error_string().decode()}")


_raw_stream = None


def stream_ptr():
    """Raw hipStream_t of torch's current stream on the current device (fast path: one C call, no Stream object)."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream
