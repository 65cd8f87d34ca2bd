#!/usr/bin/env python
"""How fast is the oracle (the "port" that bench.py times as `cpu_baseline`) next to the reference itself?

    python tests/golden/time_port_vs_reference.py          # build container only: needs /root/reference

The bench box has no /root/reference, so `cpu_baseline.kind` is "port"; this script records, once per round in the
build container, the time of the UNMODIFIED reference's `scintools.ththmod.Eval_calc` (ththmod.py:371-401; astropy
replaced by the stand-in of tests/golden/refshim, as for the goldens) beside `oracle.thth_oracle.Eval_calc` on the
same conjugate spectrum, curvature by curvature, with the values' relative difference.  Result:
tests/golden/port_vs_reference_timing.json, which bench.py quotes in `cpu_baseline.sample` (missing 4).
Both run in this process with the BLAS threads the container gives them; the ratio, not the absolute times, is
what carries over to another host.
"""
import json
import os
import sys
import time
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "refshim"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import astropy.units as u  # noqa: E402  (the shim)
import scintools.ththmod as ref  # noqa: E402
from oracle import thth_oracle as port  # noqa: E402
from scintools_amd.synth import arc_dynspec  # noqa: E402

warnings.simplefilter("ignore")


def main(size=2048, factors=(0.5, 1.0, 3.5), reps=3):
    dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=2, nimg=64)
    dyn -= dyn.mean()
    fd, tau = port.fft_axis(times, 1000.0, 0), port.fft_axis(freqs, 1.0, 0)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, size)
    CS = port.conjugate_spectrum(dyn, 0)
    rows = []
    for f in factors:
        eta = f * eta_true
        t_ref, t_port = [], []
        for _ in range(reps):           # interleaved, so that both see the same machine state
            t0 = time.perf_counter()
            v_ref = ref.Eval_calc(CS, tau * u.us, fd * u.mHz, eta * u.s**3, edges * u.mHz)
            t_ref.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            v_port = port.Eval_calc(CS, tau, fd, eta, edges)
            t_port.append(time.perf_counter() - t0)
        rows.append({"eta_over_true": f, "reference_s": float(np.median(t_ref)), "port_s": float(np.median(t_port)),
                     "port_over_reference": float(np.median(t_port) / np.median(t_ref)),
                     "rel_diff": float(abs(v_port - v_ref) / abs(v_ref))})
        print(rows[-1], flush=True)
    out = {"what": "reference scintools.ththmod.Eval_calc vs oracle.thth_oracle.Eval_calc, same CS, median of %d" % reps,
           "size": size, "host_cores": os.cpu_count(), "rows": rows,
           "port_over_reference_time": float(np.mean([r["port_over_reference"] for r in rows])),
           "max_rel_diff": float(max(r["rel_diff"] for r in rows))}
    with open(os.path.join(HERE, "port_vs_reference_timing.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:2]))
