"""ctypes binding of libavsr_hip.so (the gfx950 HIP kernels; C ABI in include/avsr_hip.h).

The product path loads exactly one file, ``auto_avsr_amd/libavsr_hip.so``, and
raises if it is missing -- there is no eager / PyTorch / CPU fallback.  The
prototypes are parsed from the public header so that the header stays the single
source of truth for the ABI.
"""
import ctypes
import os
import re

# torch wheels bundle their own libamdhip64; it must be the HIP runtime of the process, so it has to be loaded
# before libavsr_hip.so resolves its libamdhip64.so.7 dependency (two runtimes in one process = "no device")
import torch  # noqa: F401  (side effect: loads torch/lib/libamdhip64.so)

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "avsr_hip.h")
LIB_PATH = os.path.join(_HERE, "libavsr_hip.so")

_SCALARS = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "uint64_t": ctypes.c_uint64,
    "int64_t": ctypes.c_int64,
    "avsr_stream_t": ctypes.c_void_p,
    "uint8_t": ctypes.c_uint8,
}


def parse_header(path=HEADER):
    """Return {name: (restype, [argtypes], [argnames])} for every avsr_* prototype."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(avsr_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef"):
            continue
        restype = ctypes.c_char_p if "char" in ret else (ctypes.c_int64 if "int64_t" in ret else ctypes.c_int)
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                nm = re.findall(r"\w+", a)[-1]
                ty = a[: a.rfind(nm)].strip()
                if "*" in ty:
                    argtypes.append(ctypes.c_void_p)
                else:
                    base = ty.replace("const", "").strip()
                    argtypes.append(_SCALARS[base])
                argnames.append(nm)
        protos[name] = (restype, argtypes, argnames)
    return protos


class AvsrLibraryError(RuntimeError):
    pass


class _Lib:
    def __init__(self, path):
        if not os.path.exists(path):
            raise AvsrLibraryError(
                f"{path} not found: build it with `python -m auto_avsr_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no fallback path."
            )
        self.path = path
        self.cdll = ctypes.CDLL(path)
        self.protos = parse_header()
        for name, (restype, argtypes, _) in self.protos.items():
            fn = getattr(self.cdll, name, None)
            if fn is None:
                raise AvsrLibraryError(f"{path} does not export {name} declared in avsr_hip.h")
            fn.restype = restype
            fn.argtypes = argtypes
        self.is_emulator = bool(self.cdll.avsr_is_emulator())
        self._fns = {name: (getattr(self.cdll, name), restype is ctypes.c_int)
                     for name, (restype, _, _) in self.protos.items()}

    def call(self, name, *args):
        fn, status = self._fns[name]
        rc = fn(*args)
        if not status:
            return rc
        if rc != 0:
            raise AvsrLibraryError(f"{name} failed ({rc}): {self.cdll.avsr_last_error().decode()}")


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib(LIB_PATH)
    return _lib


def _install_for_tests(path):
    """Test hook (tests/conftest.py): point the binding at another build of the same ABI."""
    global _lib
    _lib = _Lib(path)
    return _lib
