"""Device plumbing: PyTorch-ROCm owns HBM buffers and the HIP stream, nothing else.

No torch op runs inside the hot loop -- tensors are allocated here and their
``data_ptr()`` is handed to the HIP kernels through the C ABI.
"""
import threading

import numpy as np
import torch

from . import _lib


def require_gpu():
    _lib.require_gpu()
    if not torch.cuda.is_available():
        raise _lib.ScintHipError("torch.cuda is not available: scintools_amd has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr():
    """hipStream_t of torch's current stream (kernels and torch copies stay ordered)."""
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


def to_device(x, dtype):
    """NumPy array / torch tensor -> contiguous device tensor of `dtype`."""
    dev = require_gpu()
    if isinstance(x, torch.Tensor):
        t = x.to(device=dev, dtype=dtype)
    else:
        np_dtype = {torch.float64: np.float64, torch.complex128: np.complex128,
                    torch.int32: np.int32, torch.int64: np.int64, torch.uint8: np.uint8}[dtype]
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np_dtype)).to(dev)
    return t.contiguous()


def empty(shape, dtype):
    return torch.empty(shape, dtype=dtype, device=require_gpu())


class Workspace:
    """Grow-only scratch buffer handed to the library (which never allocates HBM
    except its cached twiddle tables)."""

    def __init__(self):
        self._buf = None

    def get(self, nbytes):
        if self._buf is None or self._buf.numel() < nbytes or self._buf.device != require_gpu():
            self._buf = None
            self._buf = torch.empty(int(nbytes), dtype=torch.uint8, device=require_gpu())
        return self._buf

    def release(self):
        self._buf = None


class _PerThreadWorkspace:
    """One grow-only scratch buffer per Python thread (each thread drives its own stream)."""

    def __init__(self):
        self._local = threading.local()

    def _ws(self):
        ws = getattr(self._local, "ws", None)
        if ws is None:
            ws = self._local.ws = Workspace()
        return ws

    def get(self, nbytes):
        return self._ws().get(nbytes)

    def release(self):
        self._ws().release()


workspace = _PerThreadWorkspace()


class DeviceBacked:
    """Attribute that stays in HBM until the host first looks at it.

    The reference keeps its spectra as NumPy attributes (``self.sspec`` ...).  Here a kernel's
    output is parked as a device tensor; internal consumers take it straight from HBM
    (``DeviceBacked.tensor``).  The first host read copies it down ONCE and hands ownership to
    the host array -- the device copy is dropped, so in-place edits of the NumPy array are
    always honoured by later calls (they re-upload it).  Assignment stores a host value."""

    def __init__(self, name):
        self.name = name
        self.slot = "_devbacked_" + name

    def __get__(self, obj, owner=None):
        if obj is None:
            return self
        state = obj.__dict__.get(self.slot)
        if state is None:
            raise AttributeError(f"'{type(obj).__name__}' object has no attribute '{self.name}'")
        if state[0] is None:
            state[0] = state[1].cpu().numpy()
            state[1] = None
        return state[0]

    def __set__(self, obj, value):
        obj.__dict__[self.slot] = [value, None]

    def __delete__(self, obj):
        obj.__dict__.pop(self.slot, None)

    # -- internal side ---------------------------------------------------------------
    def park(self, obj, tensor):
        """Store a kernel output without copying it to the host."""
        obj.__dict__[self.slot] = [None, tensor]

    def present(self, obj):
        return obj.__dict__.get(self.slot) is not None

    def tensor(self, obj, dtype=torch.float64):
        """Device tensor of the current value: the parked one, or an upload of the host array."""
        state = obj.__dict__.get(self.slot)
        if state is None:
            raise AttributeError(f"'{type(obj).__name__}' object has no attribute '{self.name}'")
        if state[1] is not None:
            return state[1]
        return to_device(np.asarray(state[0]), dtype)

    def shape(self, obj):
        state = obj.__dict__[self.slot]
        return tuple(state[1].shape) if state[1] is not None else np.shape(state[0])
