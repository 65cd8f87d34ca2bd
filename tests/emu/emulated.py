"""TEST INFRASTRUCTURE ONLY: run the package's Python wrappers against the host-interpreted
build of the kernel sources (tests/emu/build_emu.py) instead of libscint_hip.so on a GPU.

`install(monkeypatch)` swaps, for the duration of one test, the library handle, the GPU check and
the stream getter of scintools_amd for emulated ones; tensors then live in host memory (the
interpreter's "device" memory) and every C-ABI call executes the kernels block by block on the
CPU.  Nothing here is reachable from the product: scintools_amd never imports tests/, and outside
these tests it still raises ScintHipError without a GPU (tests/test_cabi_cpu.py::test_no_cpu_fallback).
"""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

_emu_lib = None


def load():
    """Build (if stale) and load the interpreted library with the product's ctypes signatures."""
    global _emu_lib
    if _emu_lib is None:
        import build_emu
        from scintools_amd import _lib
        lib = ctypes.CDLL(build_emu.build())
        for name, (argtypes, restype) in _lib._SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = restype
        _emu_lib = lib
    return _emu_lib


def install(monkeypatch):
    import torch
    from scintools_amd import _lib, arcfit, device, dynspec, ththmod
    lib = load()
    cpu = torch.device("cpu")
    monkeypatch.setattr(_lib, "_lib", lib)
    monkeypatch.setattr(_lib, "require_gpu", lambda: lib)
    monkeypatch.setattr(device, "require_gpu", lambda: cpu)
    monkeypatch.setattr(device, "stream_ptr", lambda: None)
    monkeypatch.setattr(device, "workspace", device._PerThreadWorkspace())
    for mod in (ththmod, dynspec, arcfit):
        monkeypatch.setattr(mod, "require_gpu", lambda: cpu)
        monkeypatch.setattr(mod, "stream_ptr", lambda: None)
        monkeypatch.setattr(mod, "workspace", device.workspace)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    return lib
