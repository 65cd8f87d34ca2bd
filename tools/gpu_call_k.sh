#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --modeler-steps 0 > $O/k_bench.json 2>> $O/k_bench.err
timeout 300 python bench.py --objective chisq --steps 2 --warmup 1 --no-cpu-baseline > $O/k_bench_chisq.json 2>> $O/k_bench.err
timeout 300 python -m pytest tests/test_gpu_edges.py tests/test_gpu_modeler_fullsize.py -m gpu -q -x > $O/k_pytest.log 2>&1; tail -2 $O/k_pytest.log
python - <<'PY'
import json
for f in ('k_bench.json','k_bench_chisq.json'):
    d=json.loads([l for l in open('gpurun_out/'+f) if l.startswith('{')][-1]); r=d['roofline']
    print(f, round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'mv GB/s', round(r['achieved']), 'share', round(r['share_of_step_time'],3), 'steps', d['config']['lanczos_steps_mean'])
PY
