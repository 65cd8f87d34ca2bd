#!/bin/bash
# scratch per-call script (round 6): calc_wavefield after sharing grids / crop tables between the chunks of a frequency row
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -q -x -k "retrieval or mosaic or wavefield or chunk or multi" > $O/r06q_pytest.log 2>&1; tail -2 $O/r06q_pytest.log
timeout 600 python bench.py --workload wavefield --steps 3 --warmup 1 > $O/r06_wl_wavefield_shared_grids.json 2> $O/r06q.err; python -c "
import json
d=json.loads([l for l in open('$O/r06_wl_wavefield_shared_grids.json') if l.startswith('{')][-1])
print(d['value'], d['seconds_all'], d.get('parity_sample'), {k: round(v['busy_share_of_wall'],3) for k,v in d['kernels'].items()})"
timeout 600 python bench.py --workload fit_thetatheta --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fit_thetatheta', d['value'], d.get('parity_sample'))"
timeout 300 python bench.py --workload tutorial_fit --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('tutorial', d['value'], d['parity'])"
