#!/bin/bash
# check cadence of the block sweep (repeated, interleaved)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do for ce in 4 3 2 5; do
  SCINT_CHECK_EVERY=$ce timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --modeler-steps 0 > $O/v_bench_ce${ce}_$rep.json 2>> $O/v_bench.err
done; done
python - <<'PY'
import json
for ce in (2,3,4,5):
    out=[]
    for rep in (1,2,3):
        d=json.loads([l for l in open(f'gpurun_out/v_bench_ce{ce}_{rep}.json') if l.startswith('{')][-1])
        out.append(round(d['value'],1))
    print('check_every', ce, out, 'passes', d['config']['lanczos_steps_mean'])
PY
