#!/bin/bash
# Round-2 GPU call C: pipelined single-stream scheduler, windowed rev_map, one-call chi^2 sweep.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/c_pytest.log 2>&1; echo "pytest rc=$?" >> $O/c_pytest.log
tail -25 $O/c_pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 > $O/c_bench.json 2> $O/c_bench.err; echo "bench rc=$?"
for b in 44 52 60 70; do
  timeout 300 python bench.py --batch $b --steps 4 --warmup 1 --no-cpu-baseline --modeler-steps 0 > $O/c_bench_b$b.json 2>> $O/c_bench.err
done
SCINT_SWEEP_DEPTH=1 timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --modeler-steps 0 > $O/c_bench_depth1.json 2>> $O/c_bench.err
timeout 300 python tools/time_modeler.py 4096 > $O/c_modeler.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/c_prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --modeler-steps 0 > $O/c_prof.log 2>&1
db=$(find $O/c_prof -name "*.db" | head -1); python $R/tools/rocpd_summary.py $db $O/c_kernel_stats.csv $O/c_kernel_overlap.json > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $O/c_prof_mod -o bench -- python $R/bench.py --objective chisq --steps 2 --warmup 1 --no-cpu-baseline > $O/c_prof_mod.log 2>&1
db=$(find $O/c_prof_mod -name "*.db" | head -1); python $R/tools/rocpd_summary.py $db $O/c_kernel_stats_mod.csv $O/c_kernel_overlap_mod.json > /dev/null
find $O -name "*.db" -size +30M -delete
cd $R; head -c 400 $O/c_bench.json; echo; tail -3 $O/c_bench.err; tail -5 $O/c_modeler.txt
