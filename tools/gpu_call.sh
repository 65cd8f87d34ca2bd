#!/bin/bash
# scratch per-call script (round 6): chi^2 from the back-map's accumulators (SCINT_CHISQ_FUSE=1, default) against the written image (=0)
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_modeler_fullsize.py tests/test_gpu_edges.py tests/test_gpu_stopping_rule.py -m gpu -q -x -k "rev or model or chisq" > $O/r06y_pytest.log 2>&1; tail -3 $O/r06y_pytest.log
{
echo "# python bench.py --objective chisq --steps 4 --warmup 1 --headline-only   (eta/s, ms per step; interleaved)"
for rep in 1 2 3; do for d in 0 1; do
  v=$(SCINT_CHISQ_FUSE=$d timeout 600 python bench.py --objective chisq --steps 4 --warmup 1 --headline-only 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1), round(d['ms_per_step'],1))")
  echo "SCINT_CHISQ_FUSE=$d  $v"
done; done
python - <<'PY'
import numpy as np, torch, time
from scintools_amd import ththmod as thth
from scintools_amd.synth import arc_dynspec
import os
size=4096
dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=3, nimg=64); dyn -= dyn.mean()
fd = thth.fft_axis(times, 1000.0); tau = thth.fft_axis(freqs, 1.0)
edges = np.linspace(-fd.max()/2, fd.max()/2, size)
cs = thth.to_device(thth.conjugate_spectrum(dyn, 0, pad_value=0.0))
etas = np.linspace(0.25, 4.0, 256) * eta_true
os.environ["SCINT_CHISQ_FUSE"]="1"
a, ia = thth.chisq_sweep(dyn, cs, tau, fd, etas, edges, float(dyn.size), return_info=True)
os.environ["SCINT_CHISQ_FUSE"]="0"
b, ib = thth.chisq_sweep(dyn, cs, tau, fd, etas, edges, float(dyn.size), return_info=True)
print("fused route:", ia["fused"], "redone", ia["redone"], "| unfused:", ib["fused"], ib["redone"], "| max rel diff", np.nanmax(np.abs(a-b)/np.abs(b)), "failed", int(np.sum(ia["status"]!=0)))
PY
} > $O/r06_chisq_fuse_ab.txt 2>&1
cat $O/r06_chisq_fuse_ab.txt
