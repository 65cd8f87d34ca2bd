#!/bin/bash
# after the check-cadence change: full GPU suite + the driver's bench command
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/w_pytest.log 2>&1; echo "pytest rc=$?" >> $O/w_pytest.log
grep -E "passed|failed|^FAILED|rc=" $O/w_pytest.log | tail -6
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/w_bench_n1.json 2> $O/w_bench.err; echo "bench rc=$?"
head -c 250 $O/w_bench_n1.json; echo
