"""Time the rank-1 rev_map (API path, the whole image) over the curvatures of the headline sweep, and the inverse model FFT.

    python tools/time_revmap.py [size] [eta / eta_true ...]        (default 4096; 0.25 0.5 1 2 4)

The cost of an image depends on the curvature: the pairs of a Doppler column fall on fewer delay rows the flatter the arc is
(more LDS-atomic collisions per row), and above the crop (eta > ~2.9 eta_true at the headline grid) N shrinks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scintools_amd import ththmod as thth
from scintools_amd.synth import arc_dynspec
args = [a for a in sys.argv[1:]]
size = int(args[0]) if args else 4096
factors = [float(a) for a in args[1:]] or [0.25, 0.5, 1.0, 2.0, 4.0]
dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=3, nimg=64); dyn -= dyn.mean()
fd = thth.fft_axis(times, 1000.0); tau = thth.fft_axis(freqs, 1.0)
edges = np.linspace(-fd.max() / 2, fd.max() / 2, size)
cs = thth.conjugate_spectrum(dyn, 0, pad_value=0.0)
grid = thth._Grid(tau, fd, edges)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
etas = np.array(factors) * eta_true
w, V, info = thth.eigvec_sweep(cs, tau, fd, etas, edges)
w_t = info["w_dev"]
rec = None
for k, e in enumerate(etas):
    keep = grid.keep(e)
    n = len(keep)
    th_red = thth._theta_centres(grid.edges_red(keep)); th_t = thth.to_device(th_red, torch.float64)
    band = 2 * abs(e) * (th_red**2).max() / (tau[1] - tau[0])
    rec = thth._rev_map_dev(grid.geom, th_t, n, e, True, vec_t=V[k], w_t=w_t[k:k + 1])
    ms = t(lambda: thth._rev_map_dev(grid.geom, th_t, n, e, True, vec_t=V[k], w_t=w_t[k:k + 1]))
    print(f"rev_map rank-1 eta/eta_true {factors[k]:5.2f}  N {n:5d}  delay band ~{min(band, len(tau)):6.0f} rows  {ms:.3f} ms", flush=True)
print("model fft ms", t(lambda: thth._model_dev(rec)))
