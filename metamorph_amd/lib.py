"""ctypes binding of libmm355.so (the C ABI declared in include/mm355.h).

The library is the only compute path of this package: if it cannot be loaded every op raises
`Mm355Unavailable` -- there is NO eager / CPU fallback (a silent fallback would void the parity claims).
"""
from __future__ import annotations

import ctypes
import os
import re
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint32, c_void_p

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MM355_LIB_PATH") or os.path.join(PKG, "lib", "libmm355.so")   # (override: timing-only ablation builds, tools/)
HEADER = os.path.join(os.path.dirname(PKG), "include", "mm355.h")


class Mm355Unavailable(RuntimeError):
    pass


class Mm355Error(RuntimeError):
    pass


_lib = None
_load_error = None

_CTYPE = {
    "int": c_int, "int64_t": c_int64, "uint32_t": c_uint32, "float": c_float,
}


def _parse_header():
    """Return {name: [ctype, ...]} for every `int mm355_*(...)` prototype in include/mm355.h."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int64_t|int|const char\*)\s+(mm355_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    types.append(c_void_p)
                else:
                    base = a.replace("const", "").split()[0]
                    types.append(_CTYPE[base])
        protos[name] = ({"int": c_int, "int64_t": c_int64}.get(ret, c_char_p), types)
    return protos


def exported_symbols():
    return sorted(_parse_header().keys())


def load():
    global _lib, _load_error
    if _lib is not None:
        return _lib
    if _load_error is not None:
        raise Mm355Unavailable(_load_error)
    if not os.path.exists(LIB_PATH):
        _load_error = (f"{LIB_PATH} is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950).  metamorph_amd has no non-HIP fallback.")
        raise Mm355Unavailable(_load_error)
    # PyTorch-ROCm wheels bundle their own libamdhip64; it must be the HIP runtime this library binds to (one
    # runtime per process: streams and device pointers are not interchangeable between two copies).  Importing
    # torch first makes the dynamic linker resolve our DT_NEEDED libamdhip64.so.N to the already-loaded copy.
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        _load_error = f"cannot load {LIB_PATH}: {e}"
        raise Mm355Unavailable(_load_error)
    for name, (ret, types) in _parse_header().items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = ret
        fn.argtypes = types
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        msg = load().mm355_strerror(code)
        raise Mm355Error(f"{what}: {msg.decode() if msg else code} ({code})")
