#!/bin/bash
# Round-2 GPU call I: half-strip block mat-vec vs full-strip, two tail lanes; traces of the chi^2 sweep.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/i_pytest.log 2>&1; echo "pytest rc=$?" >> $O/i_pytest.log
grep -E "passed|failed|^FAILED|rc=" $O/i_pytest.log | tail -15
for hf in 1 0; do
  SCINT_MV2_HALF=$hf timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --modeler-steps 0 > $O/i_bench_half$hf.json 2>> $O/i_bench.err
done
timeout 300 python bench.py --objective chisq --steps 2 --warmup 1 --no-cpu-baseline > $O/i_bench_chisq.json 2>> $O/i_bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/i_prof_mod -o bench -- python $R/bench.py --objective chisq --steps 2 --warmup 1 --no-cpu-baseline > $O/i_prof_mod.log 2>&1
db=$(find $O/i_prof_mod -name "*.db" | head -1); python $R/tools/rocpd_summary.py $db $O/i_kernel_stats_mod.csv $O/i_kernel_overlap_mod.json > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $O/i_prof_tm -o tm -- python $R/tools/time_modeler.py 4096 > $O/i_prof_tm.log 2>&1
db=$(find $O/i_prof_tm -name "*.db" | head -1); python $R/tools/rocpd_summary.py $db $O/i_kernel_stats_tm.csv > /dev/null
find $O -name "*.db" -size +30M -delete
cd $R; tail -3 $O/i_bench.err
