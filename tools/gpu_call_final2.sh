#!/bin/bash
# Round-2 closing measurement (PMC passes and the Simulation screen of gpu_call_final.sh are not repeated:
# the mat-vec and gather kernels are unchanged since that run).
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
TAG=${1:-m}
timeout 900 python -m pytest tests -m gpu -q --durations=8 > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.log
grep -E "passed|failed|^FAILED|rc=" $O/${TAG}_pytest.log | tail -8
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"
timeout 300 python bench.py --size 2048 --obs-total 64 --steps 2 --warmup 1 --no-cpu-baseline --modeler-steps 0 > $O/${TAG}_bench_cfg4_64obs.json 2>> $O/${TAG}_bench.err
timeout 300 python bench.py --size 8192 --neta 64 --steps 2 --warmup 1 --no-cpu-baseline --modeler-steps 0 > $O/${TAG}_bench_cfg5_8192.json 2>> $O/${TAG}_bench.err
timeout 300 python bench.py --npad 3 --steps 2 --warmup 1 --no-cpu-baseline --modeler-steps 0 > $O/${TAG}_bench_npad3.json 2>> $O/${TAG}_bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --modeler-steps 0 > $O/${TAG}_prof.log 2>&1
db=$(find $O/${TAG}_prof -name "*.db" | head -1); python $R/tools/rocpd_summary.py $db $O/${TAG}_bench_kernel_stats.csv $O/${TAG}_bench_kernel_overlap.json > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_mod -o bench -- python $R/bench.py --objective chisq --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_prof_mod.log 2>&1
db=$(find $O/${TAG}_prof_mod -name "*.db" | head -1); python $R/tools/rocpd_summary.py $db $O/${TAG}_modeler_kernel_stats.csv $O/${TAG}_modeler_kernel_overlap.json > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_fft -o fft -- python $R/tools/time_fft.py > $O/${TAG}_prof_fft.log 2>&1
db=$(find $O/${TAG}_prof_fft -name "*.db" | head -1); python $R/tools/rocpd_summary.py $db $O/${TAG}_fft_kernel_stats.csv > /dev/null
find $O -name "*.db" -size +20M -delete
cd $R
head -c 300 $O/${TAG}_bench_n1.json; echo; tail -3 $O/${TAG}_bench.err; 
