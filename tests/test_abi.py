"""Host-side checks that need no GPU: the C-ABI library builds, loads, and exports exactly what include/sed_hip.h
declares; the product path refuses to run without the HIP library or on CPU tensors (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

from desed_task_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    path = build.build(verbose=False)                       # hipcc cross-compiles gfx950 without a GPU
    protos = _lib.parse_header()
    assert len(protos) >= 25
    dll = ctypes.CDLL(path)                                 # loads on a GPU-less host (no compute calls made)
    for name in protos:
        assert hasattr(dll, name), "libsed_hip.so does not export %s" % name
    # ... and NOTHING else: the dynamic symbol table is exactly the header (no kernel stubs, no template instantiations)
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    exported = sorted(ln.split()[-1] for ln in nm.splitlines() if ln.strip())
    assert exported == sorted(protos), sorted(set(exported) ^ set(protos))[:10]
    # every prototype cites the reference call site it replaces or says it has none
    text = open(os.path.join(ROOT, "include", "sed_hip.h")).read()
    assert "sed_trainer.py" in text and "CRNN.py" in text and "CNN.py" in text and "RNN.py" in text


def test_header_prototypes_parse_to_ctypes():
    protos = _lib.parse_header()
    args = protos["sed_mel_fwd"]
    assert args[0] is ctypes.c_void_p and args[2] is ctypes.c_int and args[-1] is ctypes.c_void_p
    assert protos["sed_adam_step"][4] is ctypes.c_longlong and protos["sed_adam_step"][5] is ctypes.c_float


def test_no_cpu_fallback():
    """With the real library bound, CPU tensors are rejected; with no library the binding raises."""
    saved = _lib._lib
    try:
        _lib.use_library(None)
        lib = _lib.get()
        assert not lib.is_emulator
        with pytest.raises(RuntimeError, match="no CPU path"):
            _lib.check_tensor(torch.zeros(4), "x")
        from desed_task_amd import features
        with pytest.raises(RuntimeError):
            features.take_log(torch.ones(2, 4, 5))
        _lib.use_library(None)
        real = _lib.LIB_PATH
        _lib.LIB_PATH = real + ".missing"
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            _lib.get()
        _lib.LIB_PATH = real
    finally:
        _lib._lib = saved


def test_product_never_imports_oracle_or_emulator():
    pkg = os.path.join(ROOT, "desed_task_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "tests.emu" not in src and "build_emu" not in src, f
