"""ctypes binding of libmrn_hip.so.

Signatures are parsed from include/mrn_hip.h, so the header is the single source of truth for the C ABI.
There is no CPU fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os
import re

import torch  # noqa: F401  -- must come first: torch's bundled libamdhip64 has to be the process's HIP runtime before
#                              libmrn_hip.so (NEEDED libamdhip64.so.7) is dlopen'ed, otherwise a second runtime
#                              without a device gets bound ("no ROCm-capable device is detected")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MRN_LIB_PATH") or os.path.join(_HERE, "csrc", "libmrn_hip.so")   # (override: A/B builds)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mrn_hip.h")

_CTYPES = {
    "int": ctypes.c_int,
    "int64_t": ctypes.c_int64,
    "float": ctypes.c_float,
    "void*": ctypes.c_void_p,
    "float*": ctypes.c_void_p,
    "const float*": ctypes.c_void_p,
    "const int64_t*": ctypes.c_void_p,
    "int64_t*": ctypes.c_void_p,
    "const int*": ctypes.c_void_p,
    "int*": ctypes.c_void_p,
    "const void*": ctypes.c_void_p,
    "const void* const*": ctypes.c_void_p,
    "const char*": ctypes.c_char_p,
}


def parse_header(path=HEADER_PATH):
    """Return {name: (restype, [argtypes], [argnames])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = {}
    for m in re.finditer(r"(?m)^\s*((?:const\s+)?\w+\s*\*?)\s*(mrn_\w+)\s*\(([^;{]*?)\)\s*;", text):
        ret = re.sub(r"\s+", " ", m.group(1)).strip().replace(" *", "*")
        name = m.group(2)
        args = m.group(3).strip()
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = re.sub(r"\s+", " ", a).strip()
                mm = re.match(r"(.*?)(\w+)$", a)
                ty = mm.group(1).strip().replace(" *", "*")
                argtypes.append(ty)
                argnames.append(mm.group(2))
        protos[name] = (ret, argtypes, argnames)
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self._protos = None

    def load(self):
        if self._dll is not None:
            return self._dll
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m mrn_amd.build` (there is no CPU fallback)")
        dll = ctypes.CDLL(LIB_PATH)
        self._protos = parse_header()
        for name, (ret, argtypes, _) in self._protos.items():
            fn = getattr(dll, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = _CTYPES[ret]
            fn.argtypes = [_CTYPES[t] for t in argtypes]
        self._dll = dll
        return dll

    def call(self, name, *args):
        dll = self.load()
        fn = getattr(dll, name)
        rc = fn(*args)
        if fn.restype is ctypes.c_int and rc != 0:
            msg = dll.mrn_last_error().decode()
            raise RuntimeError(f"{name} failed (code {rc}): {msg}")
        return rc


LIB = _Lib()


def call(name, *args):
    return LIB.call(name, *args)
