"""ctypes binding of libsimseg_hip.so.  Prototypes are parsed from include/simseg_hip.h so that the header is the
single source of truth for the C ABI.  There is no CPU fallback: `call()` raises if the library is missing."""
import ctypes
import os
import re
import threading

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "simseg_hip.h")
LIB_PATH = os.environ.get("SIMSEG_AMD_LIB") or os.path.join(HERE, "libsimseg_hip.so")      # (override: same-box A/B runs of two builds)

_CT = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64, "float": ctypes.c_float}


def parse_header(path=HEADER):
    """-> {name: (restype, [(argtype, argname), ...])} for every function the header declares."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"(const char\*|int64_t|int)\s+(simseg_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        alist = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    alist.append((ctypes.c_void_p, a.split("*")[-1].strip()))
                else:
                    ty, nm = a.rsplit(" ", 1)
                    alist.append((_CT[ty], nm))
        out[name] = (ctypes.c_char_p if "char" in ret else (ctypes.c_int64 if ret == "int64_t" else ctypes.c_int), alist)
    return out


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m simseg_amd.build` (hipcc, gfx950). "
                "simseg_amd has no CPU or eager-PyTorch fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (ret, args) in parse_header().items():
            fn = getattr(lib, name, None)
            if fn is None:
                if name.startswith("simseg_debug_"):     # an older build in an A/B run may lack a debug hook
                    continue
                raise AttributeError(f"{LIB_PATH} does not export {name}: header / library mismatch")
            fn.restype = ret
            fn.argtypes = [a for a, _ in args]
        _lib = lib
    return _lib


_HALF = threading.local()     # .seen: 16-bit flavour of the tensors handed to ptr() since the last call(); .cur: what this thread's library side is set to


def note_half(dtype):
    """A call whose 16-bit tensors do not pass through ptr() (pointer tables): say which flavour they are.  Both flavours are recorded
    (bit 0 = bf16, bit 1 = fp16) so that call() can refuse a mix instead of silently reading one type as the other."""
    if dtype is torch.float16:
        _HALF.seen = getattr(_HALF, "seen", 0) | 2
    elif dtype is torch.bfloat16:
        _HALF.seen = getattr(_HALF, "seen", 0) | 1


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Remembers whether a bf16 or an fp16 tensor went by: dtype code 1 of the C ABI means
    "the 16-bit type the calling thread selected" (simseg_set_half_type), and call() selects it from what its arguments were."""
    if t is None:
        return None
    note_half(t.dtype)
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def raw(name, *args):
    """Entry points that return a value (sizes), not a status."""
    return getattr(load(), name)(*args)


def call(name, *args):
    lib = load()
    seen = getattr(_HALF, "seen", 0)
    if seen:
        _HALF.seen = 0
        if seen == 3:
            # either the arguments of this call mix the two 16-bit types, or an exception between a wrapper's ptr() calls and its call()
            # left a selection behind: refuse loudly (the selection is cleared, so the next call starts clean)
            raise RuntimeError(f"{name}: bf16 and fp16 tensors were handed to one call (or an earlier call failed half-way); "
                               "dtype code 1 of the C ABI means ONE 16-bit type per call")
        if getattr(_HALF, "cur", 1) != seen:          # (thread-local on both sides: autograd's backward threads start at bf16 like the library)
            if lib.simseg_set_half_type(seen) != 0:
                raise RuntimeError(lib.simseg_last_error().decode())
            _HALF.cur = seen
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed: {lib.simseg_last_error().decode()}")


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("simseg_amd ops run on MI355X only (tensor is on %s); there is no CPU fallback" % t.device)
