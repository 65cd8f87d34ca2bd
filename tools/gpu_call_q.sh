#!/bin/bash
# complex-to-real model transform + removal of the losing mat-vec variants: parity, chi^2 sweep and headline rates
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_modeler_fullsize.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/q_pytest.log 2>&1; tail -3 $O/q_pytest.log
timeout 300 python bench.py --objective chisq --steps 2 --warmup 1 --no-cpu-baseline > $O/q_bench_chisq.json 2>> $O/q_bench.err
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --modeler-steps 0 > $O/q_bench.json 2>> $O/q_bench.err
python - <<'PY'
import json
for f in ('q_bench_chisq.json','q_bench.json'):
    try:
        d=json.loads([l for l in open('gpurun_out/'+f) if l.startswith('{')][-1]); r=d['roofline']
        print(f, d['metric'], round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'frac', round(r['frac'],3), 'steps', d['config'].get('lanczos_steps_mean'))
    except Exception as e: print(f, 'failed', e)
PY
tail -3 $O/q_bench.err
python tools/time_fft.py 4096 sspec cs > $O/q_fft.txt 2>&1; grep -v amdgpu $O/q_fft.txt
