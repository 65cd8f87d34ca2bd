#!/bin/bash
# Round-2 GPU call H: block Lanczos for the eigenvector / chi^2 sweeps too.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/h_pytest.log 2>&1; echo "pytest rc=$?" >> $O/h_pytest.log
grep -E "passed|failed|^FAILED|rc=" $O/h_pytest.log | tail -15
SCINT_LANCZOS_BLOCK=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_stopping_rule.py -m gpu -q > $O/h_pytest_blk1.log 2>&1; tail -2 $O/h_pytest_blk1.log
timeout 300 python bench.py --objective chisq --steps 2 --warmup 1 --no-cpu-baseline > $O/h_bench_chisq.json 2>> $O/h_bench.err
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --modeler-steps 0 > $O/h_bench.json 2>> $O/h_bench.err
timeout 300 python tools/time_modeler.py 4096 > $O/h_modeler.txt 2>&1
cd $R; tail -3 $O/h_bench.err; tail -4 $O/h_modeler.txt
