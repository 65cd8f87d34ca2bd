#!/bin/bash
# Round-2 GPU call F: check cadence, 1024-thread single-slab rev_map, slot count, config 4 (64 observations).
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $O/f_pytest.log 2>&1; echo "pytest rc=$?" >> $O/f_pytest.log
tail -5 $O/f_pytest.log
for ce in 1 2 4; do
  SCINT_CHECK_EVERY=$ce timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --modeler-steps 0 > $O/f_bench_ce$ce.json 2>> $O/f_bench.err
done
timeout 300 python bench.py --batch 90 --steps 5 --warmup 2 --no-cpu-baseline --modeler-steps 0 > $O/f_bench_b90.json 2>> $O/f_bench.err
timeout 300 python bench.py --objective chisq --steps 2 --warmup 1 --no-cpu-baseline > $O/f_bench_chisq.json 2>> $O/f_bench.err
timeout 300 python tools/time_modeler.py 4096 > $O/f_modeler.txt 2>&1
timeout 300 python bench.py --size 2048 --obs-total 64 --steps 2 --warmup 1 --no-cpu-baseline --modeler-steps 0 > $O/f_bench_cfg4.json 2>> $O/f_bench.err
cd $R; tail -3 $O/f_bench.err; tail -4 $O/f_modeler.txt
