#!/usr/bin/env python
"""Write a reference-`Simulation` screen (oracle/sim_oracle.py: bit-identical restatement of
scint_sim.Simulation with the SURVEY.md 8d settings) as an input file for `bench.py --dyn-npz`:

    python tests/tools/make_sim_input.py SIZE SEED OUT.npz

4096^2 takes about three minutes of host time (SIZE FFT pairs of SIZE x 128)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
from oracle import sim_oracle  # noqa: E402

size, seed, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
sim = sim_oracle.baseline_dynspec(size, seed, workers=sim_oracle.default_workers())
np.savez(out, dyn=sim.dyn, freqs=sim.freqs, times=sim.times, eta=sim.eta)
print(out, sim.dyn.shape, "eta", sim.eta, "sha256", sim_oracle.checksum(sim.dyn)[:16])
