#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 600 python bench.py --workload fit_arc --steps 5 --warmup 2 > $O/r06_wl_fit_arc.json 2> $O/r06q.err; python -c "
import json
d=json.loads([l for l in open('$O/r06_wl_fit_arc.json') if l.startswith('{')][-1])
print(d['value'], d['seconds_all'], d.get('betaeta'), d.get('cpu_baseline',{}).get('value'), d.get('parity'))"
tail -3 $O/r06q.err
