#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "retrieval or mosaic" > $O/r06n_pytest.log 2>&1; tail -3 $O/r06n_pytest.log
timeout 600 python bench.py --workload wavefield --steps 3 --warmup 1 > $O/r06n_wl_wavefield.json 2> $O/r06n_wl.err; python -c "
import json
d=json.loads([l for l in open('$O/r06n_wl_wavefield.json') if l.startswith('{')][-1])
print(d['value'], d['seconds_all'], d.get('parity_sample'), {k: round(v['busy_share_of_wall'],3) for k,v in d['kernels'].items()})"
tail -2 $O/r06n_wl.err
timeout 600 python tools/experiments/wavefield_profile.py 2>&1 | grep -v amdgpu | cut -c1-150 | head -40 > $O/r06n_wavefield_profile.txt; head -36 $O/r06n_wavefield_profile.txt
