"""ctypes binding of libick.so — the only way the Python host reaches the HIP kernels.

The prototypes are read from include/ick.h (the C-ABI contract), so the binding cannot
drift from the header.  There is NO fallback: if the library is missing or fails to
load, importing this module's `lib()` raises — the product path never computes on a
CPU/eager substitute (a silent fallback would void every parity claim).
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

# torch bundles its own libamdhip64.so.7 / libhsa-runtime64 and loads them by path.  Importing torch FIRST makes
# libick.so's DT_NEEDED libamdhip64.so.7 resolve to that already-loaded runtime; the other order puts two HIP
# runtimes in one process and launches fail with hipErrorNoDevice.  (A non-torch C-ABI consumer simply gets
# /opt/rocm's runtime.)
import torch  # noqa: F401  (load order matters)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libick.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "ick.h")

ABI_VERSION = 8
OP_NT, OP_NN, OP_TN, OP_CONV_FWD, OP_CONV_FWD_C4, OP_CONV_DGRAD, OP_CONV_WGRAD, OP_CONV_DGRAD_S2 = range(8)
ACT_NONE, ACT_RELU, ACT_GELU, ACT_TANH = range(4)
ACT_POST_RESIDUAL = 16


class IckGemm(ctypes.Structure):
    _fields_ = [
        ("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p),
        ("bias", ctypes.c_void_p), ("residual", ctypes.c_void_p),
        ("stat_sum", ctypes.c_void_p), ("stat_sq", ctypes.c_void_p),
        ("op", ctypes.c_int32), ("act", ctypes.c_int32),
        ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
        ("lda", ctypes.c_int64), ("ldb", ctypes.c_int64), ("ldc", ctypes.c_int64), ("ldr", ctypes.c_int64),
        ("batch_outer", ctypes.c_int32), ("batch_inner", ctypes.c_int32),
        ("sAo", ctypes.c_int64), ("sAi", ctypes.c_int64), ("sBo", ctypes.c_int64), ("sBi", ctypes.c_int64),
        ("sCo", ctypes.c_int64), ("sCi", ctypes.c_int64),
        ("splitk", ctypes.c_int32), ("accumulate", ctypes.c_int32), ("alpha", ctypes.c_float),
        ("Nb", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("Cin", ctypes.c_int32),
        ("Ho", ctypes.c_int32), ("Wo", ctypes.c_int32), ("Cout", ctypes.c_int32),
        ("R", ctypes.c_int32), ("S", ctypes.c_int32), ("stride", ctypes.c_int32), ("pad", ctypes.c_int32),
        ("tile", ctypes.c_int32), ("stat_copies", ctypes.c_int32), ("stat_stride", ctypes.c_int64), ("col_scale", ctypes.c_void_p),
        ("kchunk", ctypes.c_int32), ("io16", ctypes.c_int32), ("a_absmax", ctypes.c_void_p),
    ]


_CTYPE = {"int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "long": ctypes.c_int64,
          "float": ctypes.c_float, "double": ctypes.c_double, "uint64_t": ctypes.c_uint64, "uint32_t": ctypes.c_uint32}


def parse_header(path: str = HEADER) -> Dict[str, Tuple[str, List[str]]]:
    """{symbol: (return type, [argument types])} for every `ick_*` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|const char\*)\s+(ick_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    types.append("ptr")
                else:
                    types.append(a.replace("const ", "").split()[0])
        out[name] = (ret, types)
    return out


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m imagecaptioner_amd.build` "
                           f"(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    for name, (ret, types) in parse_header().items():
        fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = ctypes.c_char_p if ret != "int" else ctypes.c_int
        fn.argtypes = [ctypes.c_void_p if t == "ptr" else _CTYPE[t] for t in types]
    if L.ick_abi_version() != ABI_VERSION:
        raise RuntimeError("libick.so ABI version mismatch")
    _lib = L
    return L


class IckError(RuntimeError):
    pass


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().ick_last_error()
        raise IckError(f"{what or 'libick'} failed (rc={rc}): {msg.decode() if msg else ''}")
