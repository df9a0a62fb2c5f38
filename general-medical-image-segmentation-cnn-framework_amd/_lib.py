"""ctypes binding of libmi355seg.so (the C-ABI declared in include/mi355seg.h).

The prototypes are parsed from the header itself so the binding cannot drift from it.
There is NO fallback: if the shared library is missing or a symbol is absent this module
raises, and every op built on it fails loudly (the product path never routes through
PyTorch eager kernels or the CPU oracle).
"""
import ctypes
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
HEADER = os.path.join(_ROOT, "include", "mi355seg.h")
# MI355SEG_LIB_PATH: load another build of the same library (A/B timing of kernel variants); it must export every symbol too
LIB_PATH = os.environ.get("MI355SEG_LIB_PATH") or os.path.join(_PKG, "libmi355seg.so")

_CTYPE = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "long long": ctypes.c_longlong,
    "size_t": ctypes.c_size_t,
}


class Mi355SegError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> {name: (restype_str, [(ctype_str, argname), ...])} for every declared function."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = {}
    for m in re.finditer(r"(const char\*|size_t|int)\s+(mi355seg_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        parsed = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                parsed.append((mm.group(1).strip(), mm.group(2)))
        protos[name] = (ret, parsed)
    return protos


def _to_ctype(t):
    if "*" in t:
        return ctypes.c_void_p
    return _CTYPE[t.replace("const ", "").strip()]


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise Mi355SegError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or `make -C {os.path.join(_PKG, 'csrc')}`). There is no CPU / eager fallback.")
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, (ret, args) in self.protos.items():
            try:
                fn = getattr(self.cdll, name)
            except AttributeError as e:
                raise Mi355SegError(f"libmi355seg.so does not export {name} (declared in include/mi355seg.h)") from e
            fn.argtypes = [_to_ctype(t) for t, _ in args]
            fn.restype = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "const char*": ctypes.c_char_p}[ret]
        self.cdll.mi355seg_last_error.restype = ctypes.c_char_p

    def call(self, name, *args):
        """Call an int-returning entry point; raise with the library's message on failure."""
        rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            msg = self.cdll.mi355seg_last_error().decode(errors="replace")
            raise Mi355SegError(f"{name} failed (rc={rc}): {msg}")

    def query(self, name, *args):
        return getattr(self.cdll, name)(*args)


_LIB = None


def lib() -> _Lib:
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB
