#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for hf in 2 0; do
  SCINT_MV2_HALF=$hf timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --modeler-steps 0 > $O/l_bench_half$hf.json 2>> $O/l_bench.err
done
SCINT_MV2_HALF=2 timeout 300 python -m pytest tests/test_gpu_edges.py tests/test_gpu_stopping_rule.py -m gpu -q -x > $O/l_pytest.log 2>&1; tail -2 $O/l_pytest.log
python - <<'PY'
import json
for f in ('l_bench_half2.json','l_bench_half0.json'):
    d=json.loads([l for l in open('gpurun_out/'+f) if l.startswith('{')][-1]); r=d['roofline']
    print(f, round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'mv GB/s', round(r['achieved']), 'share', round(r['share_of_step_time'],3), 'steps', d['config']['lanczos_steps_mean'])
PY
