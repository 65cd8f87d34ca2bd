"""Experiment (round 6): where the host time of Dynspec.fit_arc(lamsteps=True) goes (cProfile by cumulative time), the bench's 4096^2 case."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scintools_amd.dynspec import Dynspec
from scintools_amd.synth import arc_dynspec
size = 4096
dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=3, nimg=64)
class O:
    pass
def once():
    o = O(); o.dyn, o.freqs, o.times, o.name = dyn, freqs, times, "arc"; o.dt, o.df = float(times[1]-times[0]), float(freqs[1]-freqs[0])
    d = Dynspec(dyn=o, process=False, verbose=False)
    d.fit_arc(lamsteps=True, numsteps=1e4); torch.cuda.synchronize()
once()
t0 = time.perf_counter(); once(); print("fit_arc", time.perf_counter() - t0, "s")
pr = cProfile.Profile(); pr.enable(); once(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(32)
