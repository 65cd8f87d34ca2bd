import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scintools_amd import ththmod as thth
from scintools_amd.synth import arc_dynspec
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=3, nimg=64); dyn -= dyn.mean()
fd = thth.fft_axis(times, 1000.0); tau = thth.fft_axis(freqs, 1.0)
edges = np.linspace(-fd.max()/2, fd.max()/2, size)
cs = thth.conjugate_spectrum(dyn, 0, pad_value=0.0)
grid = thth._Grid(tau, fd, edges)
d_t = thth.to_device(dyn, torch.float64)
def sync(): torch.cuda.synchronize()
etas = np.geomspace(0.5, 2.0, 8) * eta_true
for rep in range(2):
    t0 = time.perf_counter()
    chis = [thth.chisq_calc(d_t, cs, tau, fd, e, edges, 1.0) for e in etas]
    sync(); t1 = time.perf_counter()
print('chisq_calc per eta ms', 1e3*(t1-t0)/len(etas), chis[:3])
# breakdown
e = eta_true
for rep in range(2):
    sync(); t=[time.perf_counter()]
    keep = grid.keep(e); red = thth._thth_dev(cs, grid, e, keep, True); sync(); t.append(time.perf_counter())
    w, V, it = thth._eigh_top_dev(red, None, True); sync(); t.append(time.perf_counter())
    th_red = thth._theta_centres(grid.edges_red(keep)); th_t = thth.to_device(th_red, torch.float64)
    w_t = thth.to_device(np.array([w]), torch.float64)
    recov = thth._rev_map_dev(grid.geom, th_t, len(keep), e, True, vec_t=V, w_t=w_t); sync(); t.append(time.perf_counter())
    model = thth._model_dev(recov); sync(); t.append(time.perf_counter())
print('gather %.2f  eigh %.2f (%d it)  rev_map %.2f  model %.2f ms' % (*(1e3*np.diff(t)), it))

etas = np.geomspace(0.5, 2.0, 32) * eta_true
for rep in range(2):
    sync(); t0 = time.perf_counter()
    chis, info = thth.chisq_sweep(d_t, cs, tau, fd, etas, edges, 1.0, return_info=True)
    sync(); t1 = time.perf_counter()
print('chisq_sweep per eta ms', 1e3*(t1-t0)/len(etas), 'iters mean', info['iters'].mean(), chis[:3], 'argmin eta/eta_true', etas[np.nanargmin(chis)]/eta_true)
for rep in range(2):
    sync(); t0 = time.perf_counter()
    w, V, info = thth.eigvec_sweep(cs, tau, fd, etas, edges)
    sync(); t1 = time.perf_counter()
print('eigvec_sweep per eta ms', 1e3*(t1-t0)/len(etas))
