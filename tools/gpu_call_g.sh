#!/bin/bash
# Round-2 GPU call G: two-vector (block) Lanczos for the eigenvalue sweep; transposed chi^2 tail.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/g_pytest.log 2>&1; echo "pytest rc=$?" >> $O/g_pytest.log
grep -E "passed|failed|^FAILED|rc=" $O/g_pytest.log | tail -15
for blk in 2 1; do
  SCINT_LANCZOS_BLOCK=$blk timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --modeler-steps 0 > $O/g_bench_blk$blk.json 2>> $O/g_bench.err
done
timeout 300 python bench.py --batch 48 --steps 5 --warmup 2 --no-cpu-baseline --modeler-steps 0 > $O/g_bench_b48.json 2>> $O/g_bench.err
timeout 300 python bench.py --objective chisq --steps 2 --warmup 1 --no-cpu-baseline > $O/g_bench_chisq.json 2>> $O/g_bench.err
timeout 300 python tools/time_modeler.py 4096 > $O/g_modeler.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/g_prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --modeler-steps 0 > $O/g_prof.log 2>&1
db=$(find $O/g_prof -name "*.db" | head -1); python $R/tools/rocpd_summary.py $db $O/g_kernel_stats.csv $O/g_kernel_overlap.json > /dev/null
find $O -name "*.db" -size +30M -delete
cd $R; tail -3 $O/g_bench.err; tail -4 $O/g_modeler.txt
