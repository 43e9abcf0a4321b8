"""ctypes binding of libmaavss_hip.so.

The argument types of every entry point are derived from include/maavss.h (the single source of
truth of the C-ABI), so the header, the library and this binding cannot drift apart silently.
There is NO fallback: if the library is missing or an entry point fails, a RuntimeError is raised.
"""
import ctypes
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
HEADER = os.path.join(_ROOT, "include", "maavss.h")
LIB_PATH = os.path.join(_PKG, "lib", "libmaavss_hip.so")

_SCALARS = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64,
            "float": ctypes.c_float, "double": ctypes.c_double, "uint32_t": ctypes.c_uint32}


def header_abi_version(path=HEADER):
    m = re.search(r"#define\s+MAAVSS_ABI_VERSION\s+(\d+)", open(path).read())
    if not m:
        raise MaavssError(f"{path}: no MAAVSS_ABI_VERSION")
    return int(m.group(1))


def parse_header(path=HEADER):
    """-> {name: (restype, [(argtype, argname)])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int64_t|int)\s+(maavss_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if "char" in ret else (ctypes.c_int64 if ret == "int64_t" else ctypes.c_int)
        argl = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    argl.append((ctypes.c_void_p, a.split("*")[-1].strip()))
                else:
                    ty, nm = a.rsplit(" ", 1)
                    ty = ty.replace("const ", "").strip()
                    argl.append((_SCALARS[ty], nm))
        protos[name] = (restype, argl)
    return protos


class MaavssError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.isfile(LIB_PATH):
            raise MaavssError(
                f"{LIB_PATH} not found: the HIP library is the product path and has no fallback. "
                f"Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C maavss_amd/csrc`.")
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, (restype, args) in self.protos.items():
            fn = getattr(self.cdll, name)      # AttributeError if the .so lacks a declared symbol
            fn.restype = restype
            fn.argtypes = [t for t, _ in args]
        have, want = self.cdll.maavss_version(), header_abi_version()
        if have != want:
            raise MaavssError(f"{LIB_PATH} reports ABI version {have}, include/maavss.h declares {want}: stale build "
                              f"(entry points changed meaning between versions, see INTEGRATION.md) -- rebuild with `make -C maavss_amd/csrc`")

    def call(self, name, *args):
        rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            raise MaavssError(f"{name} failed (status {rc}): {self.cdll.maavss_last_error().decode()}")

    def query(self, name, *args):
        """Entry points that return a size/count instead of a status."""
        return getattr(self.cdll, name)(*args)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


class KernelTimer:
    """Optional per-entry-point timing with HIP events recorded on the launch stream (torch's current
    stream is the stream every entry point launches on).  Used by bench.py for the roofline numbers."""

    def __init__(self, only=None):
        self.records = []          # (name, args, start_event, end_event)
        self.only = None if only is None else frozenset(only)   # time these entry points only (an event pair costs ~3 us of stream time)

    def summary(self):
        import torch
        torch.cuda.synchronize()
        out = {}
        for name, args, e0, e1 in self.records:
            d = out.setdefault(name, {"calls": 0, "ms": 0.0, "args": []})
            d["calls"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["args"].append(args)
        return out


_timer = None


def set_timer(timer):
    global _timer
    _timer = timer


def call(name, *args):
    if _timer is None or (_timer.only is not None and name not in _timer.only):
        lib().call(name, *args)
        return
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib().call(name, *args)
    e1.record()
    _timer.records.append((name, args, e0, e1))


def query(name, *args):
    return lib().query(name, *args)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MaavssError("maavss_amd ops run on the MI355X only (got a CPU tensor); there is no CPU fallback")
