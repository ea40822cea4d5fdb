"""Helpers for the CPU tests: build + bind the fiber-emulator build of the kernels (tests/emu)."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))


def bind_emulator():
    import build_emu
    from desed_task_amd import _lib
    path = build_emu.build()
    cur = _lib._lib
    if cur is None or not cur.is_emulator:
        _lib.use_library(path, is_emulator=True)
    return _lib.get()


def emu_threads(n):
    """Workgroups of a launch run on n OS threads; 1 = sequential launches and atomics (bit-reproducible runs)."""
    import ctypes
    from desed_task_amd import _lib
    lib = ctypes.CDLL(_lib.get().path)
    lib.emu_get_threads.restype = ctypes.c_int
    prev = lib.emu_get_threads()
    lib.emu_set_threads(int(n))
    return prev


@pytest.fixture(scope="module")
def emu():
    return bind_emulator()


@pytest.fixture
def emu_sequential(emu):
    """The emulator with in-order workgroups: two runs of the same launches are arithmetically identical."""
    prev = emu_threads(1)
    yield emu
    if prev > 0:
        emu_threads(prev)
    else:
        emu_threads(int(os.environ.get("SED_EMU_THREADS", min(8, os.cpu_count() or 1))))
