"""Time the arc-normalisation row on the GPU next to the CPU oracle (a restatement of the
reference; test infrastructure, hence this script lives under tests/).

    python tests/tools/time_arcfit.py [size] [--cpu]

Prints one JSON line: per-stage GPU wall times (host logic and PCIe copies included), the
norm_sspec kernel's algorithmic bytes / time, and with --cpu the oracle's times.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from scintools_amd import _lib  # noqa: E402
from scintools_amd.dynspec import Dynspec  # noqa: E402
from scintools_amd.synth import arc_dynspec  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
cpu = "--cpu" in sys.argv


class Obj:
    pass


dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=3, nimg=64)
o = Obj()
o.dyn, o.freqs, o.times = dyn, freqs, times


def timed(fn, reps=3):
    best = 1e99
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


d = Dynspec(dyn=o, verbose=False)
out = {"size": size}
out["scale_dyn_lambda_s"] = timed(lambda: d.scale_dyn())
out["calc_sspec_lamsteps_s"] = timed(lambda: d.calc_sspec(lamsteps=True))
eta = d.beta[len(d.beta) // 2] / (0.5 * d.fdop.max())**2
out["norm_sspec_s"] = timed(lambda: d.norm_sspec(eta=eta, lamsteps=True))
out["norm_sspec_shape"] = [len(d.normsspec_tdel), len(np.ravel(d.normsspec_fdop))]
out["fit_arc_s"] = timed(lambda: d.fit_arc(lamsteps=True, numsteps=1e4))
out["betaeta"] = float(d.betaeta)


def chain():
    """What prep_thetatheta's fallback runs: dynspec -> lambda resample -> sspec -> fit_arc, with
    every intermediate parked in HBM (only the dynamic spectrum crosses PCIe)."""
    e = Dynspec(dyn=o, verbose=False)
    e.fit_arc(lamsteps=True, numsteps=1e4)
    return e


out["chain_dyn_to_betaeta_s"] = timed(chain)

# kernel-only time of the norm_sspec row kernel through the library's hipEvent profiler
lib = _lib.load()
import ctypes  # noqa: E402
from scintools_amd.device import empty, ptr, stream_ptr, to_device  # noqa: E402
sspec_t = to_device(d.lamsspec, torch.float64)
nrow, nc = d.lamsspec.shape
fdop_t, y_t = to_device(d.fdop, torch.float64), to_device(d.beta, torch.float64)
nx = out["norm_sspec_shape"][1]
x_t = to_device(np.linspace(-5, 5, nx), torch.float64)
nr = nrow - 2
norm_t, mask_t, pow_t = empty((nr, nx), torch.float64), empty((nr, nx), torch.uint8), empty((nr,), torch.float64)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    ev0.record()
    rc = lib.scint_norm_sspec(ptr(sspec_t), nc, nc, ptr(fdop_t), ptr(y_t), 1, nr, float(eta), 5.0, nc // 2, nc // 2,
                              None, ptr(x_t), None, nx, ptr(norm_t), ptr(mask_t), ptr(pow_t), stream_ptr())
    _lib.check(rc, "scint_norm_sspec")
    ev1.record()
    torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1)
algo = 8.0 * nr * nc + 9.0 * nr * nx
out["norm_kernel_ms"] = ms
out["norm_kernel_algorithmic_GBps"] = algo / ms / 1e6

if cpu:
    from oracle import arcfit_oracle as ao  # noqa: E402
    t0 = time.perf_counter()
    r = ao.calc_sspec_lam(dyn, freqs, d.dt, d.df)
    out["cpu_calc_sspec_lamsteps_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    ao.norm_sspec(r["lamsspec"], r["beta"], r["tdel"], r["fdop"], d.freq, eta, lamsteps=True)
    out["cpu_norm_sspec_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    fa = ao.fit_arc(r["lamsspec"], r["beta"], r["tdel"], r["beta"], r["fdop"], d.freq, lamsteps=True, numsteps=1e4)
    out["cpu_fit_arc_s"] = time.perf_counter() - t0
    out["cpu_betaeta"] = float(fa["sides"][0]["eta"])
print(json.dumps(out))
