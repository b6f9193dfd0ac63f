"""ctypes binding of libcclip_hip.so (C ABI: include/cclip_hip.h) - the only compute backend.

There is no fallback: if the library is missing or a launcher reports an error this raises.
"""
from ._lib import lib, load_library, check, LIB_PATH  # noqa: F401
