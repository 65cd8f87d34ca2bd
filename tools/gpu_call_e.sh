#!/bin/bash
# Round-2 GPU call E: two pipelined groups on two streams; faster reductions.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $O/e_pytest.log 2>&1; echo "pytest rc=$?" >> $O/e_pytest.log
tail -5 $O/e_pytest.log
for b in 48 56 70; do
  timeout 300 python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --modeler-steps 0 > $O/e_bench_b$b.json 2>> $O/e_bench.err
done
SCINT_SWEEP_DEPTH=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --modeler-steps 0 > $O/e_bench_depth1.json 2>> $O/e_bench.err
timeout 300 python bench.py --objective chisq --steps 2 --warmup 1 --no-cpu-baseline > $O/e_bench_chisq.json 2>> $O/e_bench.err
timeout 300 python tools/time_fft.py > $O/e_fft.txt 2>&1
timeout 300 python tools/time_modeler.py 4096 > $O/e_modeler.txt 2>&1
cd $R; tail -3 $O/e_bench.err; grep sspec $O/e_fft.txt; tail -4 $O/e_modeler.txt
