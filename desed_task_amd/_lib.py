"""ctypes binding of libsed_hip.so (the C-ABI in include/sed_hip.h).

The product path has NO fallback: if the HIP library is missing, `get()` raises.  The signatures are
parsed from include/sed_hip.h so the header stays the single source of truth.

`use_library()` lets the CPU test-suite inject the fiber-emulator build of the *same kernel sources*
(tests/emu) to exercise the host logic without a GPU; nothing in this package ever calls it.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_HEADER = os.path.join(os.path.dirname(_HERE), "include", "sed_hip.h")
LIB_PATH = os.path.join(_HERE, "libsed_hip.so")

_CTYPES = {"int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double, "long long": ctypes.c_longlong,
           "unsigned": ctypes.c_uint, "unsigned int": ctypes.c_uint}

ERRORS = {-1: "bad argument", -2: "kernel launch failed", -3: "unsupported configuration"}


def parse_header(path=_HEADER):
    """-> {name: [ctypes argtypes]} for every `int sed_*(...)` prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|long long)\s+(sed_\w+)\s*\(([^)]*)\)\s*;", text):
        args = []
        for a in m.group(2).split(","):
            a = a.strip()
            if "*" in a:
                args.append(ctypes.c_void_p)
            else:
                ty = re.sub(r"\bconst\b", "", a).strip()
                ty = ty.rsplit(None, 1)[0].strip()
                args.append(_CTYPES[ty])
        protos[m.group(1)] = args
    return protos


class _Lib:
    def __init__(self, path, is_emulator=False):
        self.path = path
        self.is_emulator = is_emulator
        import torch  # noqa: F401  -- first: libsed_hip.so must resolve libamdhip64 to the runtime torch already loaded;
        # dlopen-ing it before torch brings in a second HIP runtime and every launch on a torch stream then fails
        self._dll = ctypes.CDLL(path)
        self.protos = parse_header()
        for name, argtypes in self.protos.items():
            fn = getattr(self._dll, name)      # AttributeError if the library misses a declared symbol
            fn.argtypes = argtypes
            fn.restype = ctypes.c_longlong if name.endswith("_floats") else ctypes.c_int

    def value(self, name, *args):
        """For the few entry points that return a count instead of a status."""
        return getattr(self._dll, name)(*args)

    def call(self, name, *args):
        rc = getattr(self._dll, name)(*args)
        if rc != 0:
            raise RuntimeError("%s failed: %s (rc=%d)" % (name, ERRORS.get(rc, "?"), rc))


_lib = None


def get():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "desed_task_amd: %s is missing -- build it with `python -m desed_task_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        _lib = _Lib(LIB_PATH)
    return _lib


def use_library(path, is_emulator=True):
    """TEST HOOK: bind a different build of the same C-ABI (the CPU fiber emulator)."""
    global _lib
    _lib = _Lib(path, is_emulator=is_emulator) if path is not None else None
    return _lib


TUNING_KEYS = {"glu_grid_cap": 0, "glu_bwd128_split": 1, "convb_ck": 2, "convb_mp": 3, "block0_nocenter": 4, "glu_fwd128": 5, "wgrad_narrow": 6, "wgrad_cap": 7, "attn_valu": 8, "wgrad_wide": 9, "gru_lds_kb": 10, "mel_taps_mem": 11, "convb_tpw": 12, "mel_wave": 13, "gemm_ntn": 14, "linear_tiles": 15}


_tuning = {}


def set_tuning(key, value):
    """Tests / sweep tools: override a kernel tuning choice of the bound library (0 = built-in choice).  See sed_set_tuning."""
    get().call("sed_set_tuning", TUNING_KEYS[key], int(value))
    _tuning[key] = int(value)


def get_tuning(key):
    """The value last given to set_tuning (0 = built-in choice): for choices made on the host side of an entry point."""
    return _tuning.get(key, 0)


def check_tensor(t, name="tensor"):
    """Device policy: CUDA (ROCm) tensors only -- unless the emulator build was injected by the tests."""
    lib = get()
    if lib.is_emulator:
        if t.is_cuda:
            raise RuntimeError("%s: emulator build bound but tensor is on the GPU" % name)
    elif not t.is_cuda:
        raise RuntimeError("%s must live on the MI355X (cuda) device: there is no CPU path" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)


def stream_ptr(t):
    if t.is_cuda:
        import torch
        return torch.cuda.current_stream(t.device).cuda_stream
    return 0
