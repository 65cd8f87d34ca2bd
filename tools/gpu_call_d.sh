#!/bin/bash
# Round-2 GPU call D: two interleaved slot halves (reduce/check on an aux stream), rev_map slab pre-test.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $O/d_pytest.log 2>&1; echo "pytest rc=$?" >> $O/d_pytest.log
tail -6 $O/d_pytest.log
for b in 54 70 94; do
  timeout 300 python bench.py --batch $b --steps 4 --warmup 1 --no-cpu-baseline --modeler-steps 0 > $O/d_bench_b$b.json 2>> $O/d_bench.err
done
SCINT_SWEEP_HALVES=1 timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --modeler-steps 0 > $O/d_bench_onehalf.json 2>> $O/d_bench.err
timeout 300 python bench.py --objective chisq --steps 2 --warmup 1 --no-cpu-baseline > $O/d_bench_chisq.json 2>> $O/d_bench.err
timeout 300 python tools/time_modeler.py 4096 > $O/d_modeler.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/d_prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --modeler-steps 0 > $O/d_prof.log 2>&1
db=$(find $O/d_prof -name "*.db" | head -1); python $R/tools/rocpd_summary.py $db $O/d_kernel_stats.csv $O/d_kernel_overlap.json > /dev/null
find $O -name "*.db" -size +30M -delete
cd $R; tail -3 $O/d_bench.err; tail -5 $O/d_modeler.txt
