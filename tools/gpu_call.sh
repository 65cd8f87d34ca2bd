#!/bin/bash
# scratch (round 6): the small-N regime (fit_thetatheta: N ~ 1200, matrices L3-resident): block rows per mat-vec workgroup and strip lengths
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cp scintools_amd/libscint_hip.so /tmp/default.so
one() { timeout 300 python bench.py --workload fit_thetatheta --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); mv=d['kernels'].get('pk2_matvec_kernel',{})
print('%.4f s' % d['value'], [round(t,3) for t in d['seconds_all']], 'mat-vec busy', round(mv.get('busy_share_of_wall',0),3), 'GB/s in flight', round(mv.get('GBs_in_flight',0)))"; }
{
for rep in 1 2; do
  for v in default rows4 rows2 rows16; do
    if [ $v = default ]; then cp /tmp/default.so scintools_amd/libscint_hip.so; else cp variants/$v.so scintools_amd/libscint_hip.so; fi
    echo "== $v: $(one)"
  done
done
cp /tmp/default.so scintools_amd/libscint_hip.so
for s in 4 6 8; do echo "== default, SCINT_STRIP_LEN=$s: $(SCINT_STRIP_LEN=$s one)"; done
cp variants/rows4.so scintools_amd/libscint_hip.so
for s in 4 7; do echo "== rows4, SCINT_STRIP_LEN=$s: $(SCINT_STRIP_LEN=$s one)"; done
cp /tmp/default.so scintools_amd/libscint_hip.so
} | tee $O/r06q_small_n_shape_ab.txt
