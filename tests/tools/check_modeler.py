import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scintools_amd import ththmod as thth
from oracle import thth_oracle as to
from scintools_amd.synth import arc_dynspec
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=3, nimg=64); dyn -= dyn.mean()
fd = thth.fft_axis(times, 1000.0); tau = thth.fft_axis(freqs, 1.0)
edges = np.linspace(-fd.max()/2, fd.max()/2, size)
CS = to.conjugate_spectrum(dyn, 0)
t0=time.time(); ref = to.modeler(CS, tau, fd, eta_true, edges); print('oracle modeler s', time.time()-t0)
t0=time.time(); got = thth.modeler(CS, tau, fd, eta_true, edges); print('gpu modeler s', time.time()-t0)
for k, name in ((0,'thth_red'),(1,'thth2'),(2,'recov'),(3,'model')):
    print(name, got[k].shape, 'max err', np.abs(got[k]-ref[k]).max(), 'ref max', np.abs(ref[k]).max())
print('w', got[5], ref[5], 'V overlap', abs(np.vdot(ref[6], got[6])))
print('chisq gpu', thth.chisq_calc(dyn, CS, tau, fd, eta_true, edges, 1.0), 'oracle', to.chisq_calc(dyn, CS, tau, fd, eta_true, edges, 1.0), 'sum dyn^2', (dyn**2).sum())
