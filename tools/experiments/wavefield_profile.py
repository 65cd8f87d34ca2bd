"""Experiment (round 6): where the host time of calc_wavefield goes (cProfile by cumulative time), the bench's 961-chunk case."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scintools_amd.dynspec import Dynspec
from scintools_amd.synth import arc_dynspec
size, cw = 4096, 256
dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=3, nimg=64)
class O:
    pass
o = O(); o.dyn, o.freqs, o.times, o.name = dyn, freqs, times, "arc"; o.dt, o.df = float(times[1]-times[0]), float(freqs[1]-freqs[0])
d = Dynspec(dyn=o, process=False, verbose=False)
d.prep_thetatheta(cwf=cw, cwt=cw, eta_min=0.5*eta_true, eta_max=2*eta_true, npad=3)
d.fit_thetatheta()
def once():
    if type(d).chunks.present(d): del d.chunks      # (hasattr would copy the parked gigabyte to the host first)
    d.calc_wavefield(); torch.cuda.synchronize()
once()
t0 = time.perf_counter(); once(); print("calc_wavefield", time.perf_counter() - t0, "s")
pr = cProfile.Profile(); pr.enable(); once(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
