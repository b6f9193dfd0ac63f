from __future__ import annotations

import ctypes
import os

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcclip_hip.so")
ABI_VERSION = 3
_lib = None


class CclipError(RuntimeError):
    pass


def load_library() -> ctypes.CDLL:
    """Load the in-tree HIP library.  Missing library = hard error (no CPU / eager fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CclipError(
            f"{LIB_PATH} is missing: build it with `python construction-clip_amd/csrc/build.py` "
            "(or __graft_entry__.build()).  This package has no non-HIP compute path.")
    l = ctypes.CDLL(LIB_PATH)
    l.cclip_abi_version.restype = ctypes.c_int
    if l.cclip_abi_version() != ABI_VERSION:
        raise CclipError(f"libcclip_hip.so ABI {l.cclip_abi_version()} != binding ABI {ABI_VERSION}; rebuild")
    _lib = l
    return l


class _LazyLib:
    def __getattr__(self, name):
        return getattr(load_library(), name)


lib = _LazyLib()

_ERR = {1: "argument/shape/alignment contract violated (nothing launched)", 2: "HIP launch error"}


def check(status: int, what: str) -> None:
    if status != 0:
        raise CclipError(f"{what}: libcclip_hip status {status} ({_ERR.get(status, 'unknown')})")
