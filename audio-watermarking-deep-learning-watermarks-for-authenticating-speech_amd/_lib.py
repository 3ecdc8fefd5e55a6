"""ctypes binding of libwm_hip.so (the C ABI declared in include/wm_hip.h).

The prototypes are read from the header itself, so the header is the single source of
truth for the boundary.  There is NO CPU fallback: if the library (or a GPU) is missing,
every entry point raises -- the HIP path is the product.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG_DIR)
HEADER = os.path.join(_ROOT, "include", "wm_hip.h")
LIB_PATH = os.environ.get("WM_HIP_LIB", os.path.join(_PKG_DIR, "libwm_hip.so"))   # override: kernel experiments only

_CT = {
    "int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double, "long long": ctypes.c_longlong,
    "wm_stream_t": ctypes.c_void_p,
}


def parse_header(path: str = HEADER) -> Dict[str, List[Tuple[str, str]]]:
    """{function name: [(c type, arg name), ...]} for every `int wm_*(...)` prototype in the header."""
    text = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    protos: Dict[str, List[Tuple[str, str]]] = {}
    for m in re.finditer(r"\bint\s+(wm_\w+)\s*\(([^)]*)\)\s*;", text):
        args = []
        for a in m.group(2).split(","):
            a = " ".join(a.split())
            mm = re.match(r"(.*?)(\w+)$", a)
            ctype, name = mm.group(1).strip(), mm.group(2)
            args.append((ctype, name))
        protos[m.group(1)] = args
    return protos


def _ctype(c: str):
    if c.endswith("*"):
        return ctypes.c_void_p
    return _CT[c.replace("const ", "")]


class _Lib:
    def __init__(self):
        self._dll = None
        self.protos = parse_header()

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the watermark hot path.")
            dll = ctypes.CDLL(LIB_PATH)
            for name, args in self.protos.items():
                fn = getattr(dll, name)          # AttributeError here == header/library mismatch
                fn.restype = ctypes.c_int
                fn.argtypes = [_ctype(c) for c, _ in args]
            self._dll = dll
        return self._dll

    def __getattr__(self, name):
        if name.startswith("wm_"):
            fn = getattr(self.load(), name)

            def call(*a, _fn=fn, _name=name):
                rc = _fn(*a)
                if rc != 0:
                    raise RuntimeError(f"{_name} failed with hipError {rc}")
            call.__name__ = name
            setattr(self, name, call)
            return call
        raise AttributeError(name)


lib = _Lib()
