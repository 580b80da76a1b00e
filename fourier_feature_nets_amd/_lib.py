"""ctypes binding of libffn_hip.so (the C ABI declared in include/ffn_hip.h).

There is deliberately no fallback: if the shared library is missing or a tensor is not
on a GPU, the call raises.  The kernels are the product; nothing here computes on the CPU.
"""

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# FFN_HIP_LIBRARY points at an alternative build of the same ABI (kernel experiments)
LIB_PATH = os.environ.get("FFN_HIP_LIBRARY") or os.path.join(_HERE, "libffn_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ffn_hip.h")
ABI_VERSION = 3

_lib = None

c_f = ctypes.c_float
c_i = ctypes.c_int
c_i64 = ctypes.c_int64
c_p = ctypes.c_void_p


class FfnError(RuntimeError):
    """A libffn_hip entry point returned a non-zero status."""


def declared_symbols():
    """Names of every entry point declared in include/ffn_hip.h."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ffn_[a-z0-9_]+)\s*\(", text)))


def load():
    """Loads the library once; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libffn_hip.so not found at %s -- build it with "
            "`python -m fourier_feature_nets_amd.build` (needs hipcc); there is no CPU "
            "fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.ffn_abi_version.restype = c_i
    lib.ffn_last_error_string.restype = ctypes.c_char_p
    if lib.ffn_abi_version() != ABI_VERSION:
        raise RuntimeError("libffn_hip.so ABI %d != expected %d; rebuild"
                           % (lib.ffn_abi_version(), ABI_VERSION))
    for name in declared_symbols():
        getattr(lib, name)  # AttributeError here means header and library disagree
    _lib = lib
    return lib


def call(name, *args):
    """Invokes an entry point and raises FfnError on a non-zero return."""
    lib = load()
    fn = getattr(lib, name)
    fn.restype = c_i
    status = fn(*args)
    if status != 0:
        raise FfnError("%s failed (%d): %s" % (name, status,
                                               lib.ffn_last_error_string().decode()))
