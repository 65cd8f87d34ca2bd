#!/bin/bash
# scratch per-call script (round 6): the chi^2 objective on a reference Simulation screen (4096^2), fused route against the written image
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 600 python tests/tools/make_sim_input.py 4096 3 /tmp/sim4096.npz > /dev/null 2>&1
{
echo "# python bench.py --dyn-npz /tmp/sim4096.npz (tests/tools/make_sim_input.py 4096 3) --objective chisq --steps 3 --warmup 1 --headline-only"
for rep in 1 2; do for d in 0 1; do
  SCINT_CHISQ_FUSE=$d timeout 600 python bench.py --dyn-npz /tmp/sim4096.npz --objective chisq --steps 3 --warmup 1 --headline-only --cpu-pool 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('SCINT_CHISQ_FUSE=$d', round(d['value'],1), round(d['ms_per_step'],1), 'failed', d.get('failed_etas'))"
done; done
python - <<'PY'
import numpy as np, os
from scintools_amd import ththmod as thth
z = np.load('/tmp/sim4096.npz'); dyn = z['dyn'] - z['dyn'].mean(); freqs, times = z['freqs'], z['times']
fd = thth.fft_axis(times, 1000.0); tau = thth.fft_axis(freqs, 1.0)
edges = np.linspace(-fd.max()/2, fd.max()/2, dyn.shape[0])
cs = thth.to_device(thth.conjugate_spectrum(dyn, 0, pad_value=0.0))
eta0 = float(z['eta']) if 'eta' in z.files else np.abs(tau).max()/(fd.max()/2)**2
etas = np.linspace(0.25, 4.0, 64) * eta0
os.environ['SCINT_CHISQ_FUSE']='1'; a, ia = thth.chisq_sweep(dyn, cs, tau, fd, etas, edges, float(dyn.size), return_info=True)
os.environ['SCINT_CHISQ_FUSE']='0'; b, ib = thth.chisq_sweep(dyn, cs, tau, fd, etas, edges, float(dyn.size), return_info=True)
print('screen: fused', ia['fused'], 'redone', ia['redone'], 'max rel diff', np.nanmax(np.abs(a-b)/np.abs(b)), 'failed', int(np.sum(ia['status']!=0)))
PY
} > $O/r06_chisq_fuse_screen.txt 2>&1
cat $O/r06_chisq_fuse_screen.txt
