"""Deviation of the phase-retrieval chain from the reference run in tests/golden/retrieval.npz (numbers quoted in
tests/test_gpu_parity.py::test_phase_retrieval_vs_reference_golden).  python tools/retrieval_deviation.py on a GPU box."""
import os, sys, numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from scintools_amd.dynspec import Dynspec
g = np.load(os.path.join(REPO, "tests/golden/retrieval.npz")); f = np.load(os.path.join(REPO, "tests/golden/fit_thetatheta.npz"))
n = int(g["nchan"])
class B: dyn, freqs, times, dt, df = f["dspec"][:n], f["freq"][:n], f["time"], float(f["dt"]), float(f["df"])
al = lambda a, r: a * np.exp(-1j * np.angle(np.vdot(r, a)))
d = Dynspec(dyn=B(), verbose=False); d.prep_thetatheta(cwf=64, edges_lim=.3, eta_min=30, eta_max=50, nedge=128); d.calc_wavefield()
print("ththeta rel", abs(d.ththeta - float(g["ththeta"])) / float(g["ththeta"]), "eta_evo rel", np.abs(d.eta_evo / g["eta_evo"] - 1).max())
for cf, key in ((0, "chunk0"), (3, "chunk3")):
    r = g[key]; print(key, np.abs(al(d.chunks[cf, 0], r) - r).max() / np.abs(r).max())
r = g["wavefield"]; print("wavefield", np.abs(al(d.wavefield, r) - r).max() / np.abs(r).max())
d.gerchberg_saxton(niter=2); r = g["wavefield_gs"]; print("wavefield_gs", np.abs(al(d.wavefield, r) - r).max() / np.abs(r).max())
