#!/bin/bash
# scratch: closing checks of the mixed sweep (tests of the sweeps with the final library, kernel trace of the mixed bench)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 70 python -m pytest tests/test_gpu_zz_mixed.py tests/test_gpu_edges.py tests/test_gpu_parity.py -m gpu -q -x --durations=4 > $O/r03mx_tests.log 2>&1; echo "pytest rc=$?" >> $O/r03mx_tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=|Error" $O/r03mx_tests.log | tail -8
( cd /tmp && timeout 45 rocprofv3 --kernel-trace --stats -d $O/r03mx_prof -o bench -- python $R/bench.py --precision mixed --steps 3 --warmup 1 --no-cpu-baseline --modeler-steps 0 > $O/r03mx_prof.log 2>&1 )
db=$(find $O/r03mx_prof -name "*.db" | head -1)
python tools/rocpd_summary.py $db $O/r03mx_kernel_stats.csv $O/r03mx_kernel_overlap.json > /dev/null
head -10 $O/r03mx_kernel_stats.csv | cut -c1-200
python tools/bench_line.py $O/r03mx_prof.log
rm -rf $O/r03mx_prof
