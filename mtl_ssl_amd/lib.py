"""ctypes binding of libmtlssl_hip.so (the C ABI declared in include/mtlssl_hip.h).

The prototypes are parsed from the header itself, so the header is the single source of
truth for the boundary. The product path fails loudly if the library is missing: there is
no CPU fallback (the CPU oracle under oracle/ is test infrastructure only).
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "mtlssl_hip.h")
LIB_PATH = os.environ.get("MTLSSL_LIB_PATH") or os.path.join(HERE, "libmtlssl_hip.so")   # the override is for A/B measurements of two builds


class ConvDesc(ctypes.Structure):
    """mtlssl_conv_desc."""
    _fields_ = [(n, ctypes.c_int32) for n in
                ("N", "H", "W", "C", "K", "R", "S", "OH", "OW", "stride", "dilation",
                 "pad_t", "pad_l", "ldy")]


_SCALARS = {
    "int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64,
    "uint32_t": ctypes.c_uint32, "float": ctypes.c_float,
    "mtlssl_stream_t": ctypes.c_void_p, "mtlssl_comm_t": ctypes.c_void_p,
}


def _ctype(decl):
    decl = decl.strip()
    if "*" in decl:
        if "mtlssl_conv_desc" in decl:
            return ctypes.POINTER(ConvDesc)
        return ctypes.c_void_p
    base = decl.replace("const", "").split()
    return _SCALARS[base[0]]


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function declared in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|int64_t|const char\*)\s+(mtlssl_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        restype = {"int": ctypes.c_int, "int64_t": ctypes.c_int64,
                   "const char*": ctypes.c_char_p}[ret]
        argtypes = []
        if args.strip() not in ("", "void"):
            for a in args.split(","):
                a = a.strip()
                # drop the parameter name
                a = re.sub(r"\s*\w+$", "", a) if not a.endswith("*") else a
                argtypes.append(_ctype(a))
        protos[name] = (restype, argtypes)
    return protos


class MtlsslError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise MtlsslError(
                "%s not found: build it with `python -m mtl_ssl_amd.build` "
                "(there is no CPU fallback for the product path)" % LIB_PATH)
        # The library must bind to the SAME HIP runtime as the process that owns the device memory
        # it is handed. PyTorch wheels bundle their own libamdhip64; loading it first makes the
        # dynamic linker resolve this library's DT_NEEDED libamdhip64 to that copy (same soname)
        # instead of opening /opt/rocm's as a second, uninitialised runtime.
        import torch  # noqa: F401
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        # an older build (MTLSSL_LIB_PATH, tools/ab_lib.sh) called through this header would get shifted arguments
        want = int(re.search(r"#define\s+MTLSSL_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
        self.cdll.mtlssl_abi_version.restype = ctypes.c_int
        have = self.cdll.mtlssl_abi_version()
        if have != want:
            raise MtlsslError("%s has ABI version %d, include/mtlssl_hip.h declares %d: rebuild it with "
                              "`python -m mtl_ssl_amd.build`" % (LIB_PATH, have, want))
        for name, (restype, argtypes) in self.protos.items():
            fn = getattr(self.cdll, name)          # AttributeError if the symbol is missing
            fn.restype = restype
            fn.argtypes = argtypes
            if restype is ctypes.c_int and name not in ("mtlssl_abi_version", "mtlssl_conv2d_tile_config", "mtlssl_conv2d_num_dispatches",
                                                        "mtlssl_conv2d_set_winograd", "mtlssl_conv2d_set_fp32_engine", "mtlssl_conv2d_filter_xf_variant",
                                                        "mtlssl_conv2d_set_pointwise"):
                setattr(self, name[len("mtlssl_"):], self._checked(fn, name))
            else:
                setattr(self, name[len("mtlssl_"):], fn)

    def _checked(self, fn, name):
        last_error = self.cdll.mtlssl_last_error

        def call(*args):
            rc = fn(*args)
            if rc != 0:
                raise MtlsslError("%s failed (%d): %s" % (name, rc, last_error().decode()))
        return call


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()
