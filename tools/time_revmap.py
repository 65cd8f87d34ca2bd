"""Time the rank-1 rev_map (API path) and the inverse model FFT on a 4096^2 conjugate spectrum."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scintools_amd import ththmod as thth
from scintools_amd.synth import arc_dynspec
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=3, nimg=64); dyn -= dyn.mean()
fd = thth.fft_axis(times, 1000.0); tau = thth.fft_axis(freqs, 1.0)
edges = np.linspace(-fd.max() / 2, fd.max() / 2, size)
cs = thth.conjugate_spectrum(dyn, 0, pad_value=0.0)
grid = thth._Grid(tau, fd, edges)
e = eta_true
keep = grid.keep(e)
w, V, info = thth.eigvec_sweep(cs, tau, fd, np.array([e]), edges)
th_red = thth._theta_centres(grid.edges_red(keep)); th_t = thth.to_device(th_red, torch.float64)
w_t = info["w_dev"]
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
n = len(keep)
rec = thth._rev_map_dev(grid.geom, th_t, n, e, True, vec_t=V[0], w_t=w_t[0:1])
print("rev_map rank-1 ms", t(lambda: thth._rev_map_dev(grid.geom, th_t, n, e, True, vec_t=V[0], w_t=w_t[0:1])))
print("model fft ms", t(lambda: thth._model_dev(rec)))
