#!/bin/bash
# Round-2 GPU call A: full GPU test-suite, baseline bench lines and kernel traces (run from the repo root).
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch" 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $O/a_pytest.log 2>&1; echo "pytest rc=$?" >> $O/a_pytest.log
tail -30 $O/a_pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 > $O/a_bench.json 2> $O/a_bench.err; echo "bench rc=$?"
SCINT_SWEEP_GROUPS=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --modeler-steps 0 > $O/a_bench_g1.json 2>> $O/a_bench.err
timeout 300 python bench.py --size 2048 --obs-total 8 --steps 3 --warmup 1 --no-cpu-baseline --modeler-steps 0 > $O/a_bench_cfg4x8.json 2>> $O/a_bench.err
# MALL-residency probes: few resident curvatures, short strips
for cfg in "1 4" "2 4" "2 8" "4 8" "8 16"; do set -- $cfg
  SCINT_STRIP_LEN=$2 timeout 200 python bench.py --neta 64 --batch $1 --steps 2 --warmup 1 --no-cpu-baseline --modeler-steps 0 > $O/a_probe_b$1_s$2.json 2>> $O/a_bench.err
done
cd /tmp
for g in 2 1; do
  SCINT_SWEEP_GROUPS=$g timeout 300 rocprofv3 --kernel-trace --stats -d $O/a_prof_g$g -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --modeler-steps 0 > $O/a_prof_g$g.log 2>&1
  db=$(find $O/a_prof_g$g -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $db $O/a_kernel_stats_g$g.csv $O/a_kernel_overlap_g$g.json > /dev/null
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/a_prof_mod -o bench -- python $R/bench.py --objective chisq --steps 2 --warmup 1 --no-cpu-baseline > $O/a_prof_mod.log 2>&1
db=$(find $O/a_prof_mod -name "*.db" | head -1); python $R/tools/rocpd_summary.py $db $O/a_kernel_stats_mod.csv $O/a_kernel_overlap_mod.json > /dev/null
find $O -name "*.db" -size +30M -delete
cd $R; head -c 600 $O/a_bench.json; echo; tail -5 $O/a_bench.err
