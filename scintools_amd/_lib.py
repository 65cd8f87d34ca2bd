"""ctypes binding of libscint_hip.so (the C ABI in include/scint_hip.h).

There is no CPU fallback: if the library or a GPU is missing, every compute
entry point raises.  Loading the library itself needs only the HIP runtime, so
``symbols()`` works in a GPU-less container (the CPU test-suite checks that the
library exports everything the header declares).
"""
import ctypes
import os
import re
from ctypes import POINTER, c_char_p, c_double, c_int32, c_int64, c_size_t, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libscint_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(HERE), "include", "scint_hip.h")

SCINT_OK = 0
SCINT_E_NOCONV = 4
SCINT_E_EMPTY = 5
SCINT_E_NONFINITE = 6


class ScintHipError(RuntimeError):
    """A call into libscint_hip.so failed (message from scint_last_error)."""


class CsGeom(ctypes.Structure):
    """Mirror of scint_cs_geom."""
    _fields_ = [("ntau", c_int64), ("nfd", c_int64),
                ("tau0", c_double), ("dtau", c_double),
                ("fd0", c_double), ("dfd", c_double),
                ("tau_max", c_double), ("fd_max", c_double),
                ("tau1_step", c_double), ("fd1_step", c_double)]


_P = c_void_p  # device pointers travel as integers
_SIGNATURES = {
    "scint_version": ([], c_int32),
    "scint_last_error": ([c_char_p, c_size_t], c_int32),
    "scint_device_count": ([], c_int32),
    "scint_profile_begin": ([], c_int32),
    "scint_profile_end": ([POINTER(c_double), POINTER(c_double), POINTER(c_int64), c_int32], c_int32),
    "scint_sspec_workspace_bytes": ([c_int64, c_int64, POINTER(c_size_t)], c_int32),
    "scint_sspec": ([_P, c_int64, c_int64, _P, _P, c_int32, c_int32, _P, _P, _P, _P, c_size_t, _P], c_int32),
    "scint_cs_workspace_bytes": ([c_int64, c_int64, c_int64, POINTER(c_size_t)], c_int32),
    "scint_cs": ([_P, c_int64, c_int64, c_int64, c_double, c_int64, c_int64, c_int32, _P, _P, c_size_t, _P], c_int32),
    "scint_mean": ([_P, c_int64, POINTER(c_double), _P], c_int32),
    "scint_thth_map": ([_P, POINTER(CsGeom), _P, c_int64, _P, c_int64, c_double, c_int32, _P, _P], c_int32),
    "scint_sweep_precision": ([c_int32], c_int32),
    "scint_sweep_schedule": ([c_int32, c_int32, c_int32], c_int32),
    "scint_sweep_stats": ([POINTER(c_double)], c_int32),
    "scint_sweep_workgroups": ([c_int32, c_int32], c_int32),
    "scint_eval_sweep_workspace_bytes": ([c_int64, c_int64, c_int64, c_int32, POINTER(c_size_t)], c_int32),
    "scint_eval_sweep": ([_P, POINTER(CsGeom), _P, c_int64, _P, POINTER(c_int32), POINTER(c_double), c_int64,
                          c_double, c_int32, c_int64, _P, _P, _P, _P, c_size_t, _P], c_int32),
    "scint_eval_sweep_multi_workspace_bytes": ([c_int64, c_int64, c_int64, c_int32, c_int64, POINTER(c_size_t)],
                                               c_int32),
    "scint_eval_sweep_multi": ([_P, c_int64, c_int64, POINTER(c_int32), POINTER(CsGeom), _P, c_int64, _P,
                                POINTER(c_int32), POINTER(c_double), c_int64, c_double, c_int32, c_int64,
                                _P, _P, _P, _P, c_size_t, _P], c_int32),
    "scint_eigvec_sweep_workspace_bytes": ([c_int64, c_int64, c_int64, c_int32, POINTER(c_size_t)], c_int32),
    "scint_eigvec_sweep": ([_P, POINTER(CsGeom), _P, c_int64, _P, POINTER(c_int32), POINTER(c_double), c_int64,
                            c_double, c_int32, c_int64, _P, _P, c_int64, _P, _P, _P, c_size_t, _P], c_int32),
    "scint_sweep_keep": ([_P, c_int64, POINTER(c_double), c_int64, c_double, c_double, _P, _P, _P], c_int32),
    "scint_eigvec_sweep_multi_workspace_bytes": ([c_int64, c_int64, c_int64, c_int32, c_int64, POINTER(c_size_t)],
                                                 c_int32),
    "scint_eigvec_sweep_multi": ([_P, c_int64, c_int64, POINTER(c_int32), POINTER(CsGeom), _P, c_int64, _P,
                                  POINTER(c_int32), POINTER(c_double), c_int64, c_double, c_int32, c_int64,
                                  _P, _P, c_int64, _P, _P, _P, c_size_t, _P], c_int32),
    "scint_chisq_sweep_workspace_bytes": ([c_int64, c_int64, c_int64, c_int32, c_int64, c_int64, c_int64, c_int64,
                                           POINTER(c_size_t)], c_int32),
    "scint_chisq_sweep": ([_P, POINTER(CsGeom), _P, c_int64, _P, POINTER(c_int32), POINTER(c_double), c_int64,
                           c_double, c_int32, c_int64, _P, POINTER(c_int32), _P, c_int64, c_int64, _P, c_double, _P, _P, _P, c_int64,
                           _P, _P, _P, c_size_t, _P], c_int32),
    "scint_chisq_sweep_last_route": ([POINTER(c_int32), POINTER(c_int64)], c_int32),
    "scint_eigh_top_workspace_bytes": ([c_int64, c_int32, POINTER(c_size_t)], c_int32),
    "scint_eigh_top": ([_P, c_int64, _P, c_double, c_int32, _P, _P, _P, _P, _P, c_size_t, _P], c_int32),
    "scint_rev_map_workspace_bytes": ([POINTER(c_size_t)], c_int32),
    "scint_rev_map": ([_P, _P, _P, c_int32, _P, c_int64, POINTER(CsGeom), c_double, c_int32, _P, _P, c_size_t, _P],
                      c_int32),
    "scint_model_workspace_bytes": ([c_int64, c_int64, POINTER(c_size_t)], c_int32),
    "scint_model_from_recov": ([_P, c_int64, c_int64, _P, _P, c_size_t, _P], c_int32),
    "scint_ifft2_shifted": ([_P, c_int64, c_int64, c_double, c_int64, c_int64, _P, _P, c_size_t, _P], c_int32),
    "scint_mosaic_workspace_bytes": ([c_int64, c_int64, c_int64, POINTER(c_size_t)], c_int32),
    "scint_mosaic_phase": ([_P, c_int64, _P, c_int64, c_int64, _P, c_int64, _P, _P, c_int32, _P, c_size_t, _P, _P], c_int32),
    "scint_mosaic_add": ([_P, c_int64, _P, c_int64, c_int64, _P, c_int64, _P, _P, c_int32, POINTER(c_double), _P], c_int32),
    "scint_chunk_cut": ([_P, c_int64, c_int64, _P, _P, c_int64, c_int64, c_int64, c_int32, _P, _P, _P, c_size_t, _P], c_int32),
    "scint_cs_batch": ([_P, c_int64, c_int64, c_int64, c_int64, POINTER(c_double), POINTER(c_int64), c_int32, _P, _P, c_size_t, _P], c_int32),
    "scint_retrieval_tail_workspace_bytes": ([c_int64, c_int64, c_int64, POINTER(c_size_t)], c_int32),
    "scint_retrieval_tail": ([_P, _P, POINTER(c_int32), POINTER(c_int32), POINTER(CsGeom), POINTER(c_double), c_int64, c_int64, c_int64, c_int64,
                              c_double, _P, _P, c_size_t, _P], c_int32),
    "scint_gs_workspace_bytes": ([c_int64, c_int64, POINTER(c_size_t)], c_int32),
    "scint_gerchberg_saxton": ([_P, c_int64, c_int64, _P, _P, c_int64, c_int64, c_int32, _P, c_size_t, _P], c_int32),
    "scint_acf_workspace_bytes": ([c_int64, c_int64, POINTER(c_size_t)], c_int32),
    "scint_acf": ([_P, c_int64, c_int64, c_int32, c_int32, _P, _P, c_size_t, _P], c_int32),
    "scint_chisq": ([_P, c_int64, _P, c_int64, c_int64, _P, c_double, _P, _P], c_int32),
    "scint_spline_resample": ([_P, c_int64, c_int64, c_int32, _P, _P, _P, _P, POINTER(c_double), c_int64, c_int64,
                               _P, _P, c_int64, _P, _P, c_size_t, _P], c_int32),
    "scint_norm_sspec": ([_P, c_int64, c_int64, _P, _P, c_int64, c_int64, c_double, c_double, c_int64, c_int64,
                          _P, _P, _P, c_int64, _P, _P, _P, _P], c_int32),
    "scint_masked_colavg_workspace_bytes": ([c_int64, c_int64, POINTER(c_size_t)], c_int32),
    "scint_masked_colavg": ([_P, _P, c_int64, c_int64, _P, _P, _P, _P, _P, c_size_t, _P], c_int32),
    "scint_row_nanmean": ([_P, c_int64, c_int64, c_int64, c_int64, _P, c_int64, c_int64, _P, _P], c_int32),
    "scint_block_std": ([_P, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, _P, _P, c_size_t, _P], c_int32),
    "scint_fft2_workspace_bytes": ([c_int64, c_int64, POINTER(c_size_t)], c_int32),
    "scint_fft2": ([_P, _P, c_int64, c_int64, _P, c_size_t, _P], c_int32),
}

_lib = None
ABI_VERSION = 107          # scint_version() of the library these signatures describe (csrc/capi.hip)


def header_symbols():
    """Names of every function declared in include/scint_hip.h."""
    with open(HEADER_PATH) as fh:
        text = fh.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(scint_[a-z0-9_]+)\s*\(", text)))


def load():
    """Load (once) and return the ctypes library.  Raises ScintHipError if the
    shared object has not been built (python -m scintools_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ScintHipError(
            f"{LIB_PATH} is missing: build it with `python -m scintools_amd.build` "
            "(there is no CPU fallback)")
    # PyTorch-ROCm ships its own HIP runtime; it must be in the process BEFORE our library is
    # loaded so that both bind the same runtime (loading ours first leaves the process with two
    # runtimes and ours then sees no device).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (argtypes, restype) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    # The signatures above describe ONE version of the C ABI (an argument added in the middle of a list shifts every
    # pointer after it): a stale build must fail here, not corrupt memory in its first call.
    got = lib.scint_version()
    if got != ABI_VERSION:
        raise ScintHipError(f"{LIB_PATH} implements version {got} of the C ABI, this package binds version {ABI_VERSION}: "
                            "rebuild it with `python -m scintools_amd.build`")
    _lib = lib
    return lib


def last_error():
    buf = ctypes.create_string_buffer(1024)
    load().scint_last_error(buf, len(buf))
    return buf.value.decode(errors="replace")


def check(rc, what=""):
    if rc != SCINT_OK:
        raise ScintHipError(f"{what or 'libscint_hip call'} failed (status {rc}): {last_error()}")


def require_gpu():
    """Fail loudly when no MI355X is visible -- the product never computes on the CPU."""
    lib = load()
    if lib.scint_device_count() < 1:
        raise ScintHipError("no HIP device visible: scintools_amd has no CPU fallback")
    return lib
