#!/bin/bash
# scratch (round 6): index-compressed passes in the real sweep, on / off and span thresholds
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stopping_rule.py -m gpu -q -x -k "sweep or eval or Eval or stopping or calibrated or single_search" > $O/r06r_pytest.log 2>&1; tail -3 $O/r06r_pytest.log
{
for rep in 1 2; do
for v in "SCINT_SWEEP_INDEXED=0" "SCINT_SWEEP_INDEXED=1" "SCINT_SWEEP_INDEXED=1 SCINT_SWEEP_INDEXED_SPAN=40" "SCINT_SWEEP_INDEXED=1 SCINT_SWEEP_INDEXED_SPAN=100" "SCINT_SWEEP_INDEXED=1 SCINT_SWEEP_INDEXED_SPAN=1000"; do
  echo "== $v: $(env $v timeout 300 python bench.py --steps 5 --warmup 2 --headline-only 2>/dev/null | python tools/bench_line.py /dev/stdin 2>&1 | tr '\n' ' ' | cut -c1-300)"
done
done
} | tee $O/r06r_indexed_ab.txt
