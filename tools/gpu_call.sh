#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -q -x -k "retrieval or mosaic or nans_goes" > $O/r06u_pytest.log 2>&1; tail -2 $O/r06u_pytest.log
timeout 600 python bench.py --workload wavefield --steps 3 --warmup 1 > $O/r06u_wl_wavefield.json 2> $O/r06u_wl.err; python -c "
import json
d=json.loads([l for l in open('$O/r06u_wl_wavefield.json') if l.startswith('{')][-1])
print(d['value'], d['seconds_all'], d.get('parity_sample'), {k: round(v['busy_share_of_wall'],3) for k,v in d['kernels'].items()})"
tail -2 $O/r06u_wl.err
timeout 300 python bench.py --workload tutorial_fit --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('tutorial', d['value'], d['parity'])"
