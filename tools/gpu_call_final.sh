#!/bin/bash
# Round-2 measurement call: full GPU suite, the driver's bench command, kernel traces, PMC passes,
# config 4 / 5 lines and a reference-Simulation screen at the headline size.
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
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace -d $O/${TAG}_pmc_$c -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --modeler-steps 0 > $O/${TAG}_pmc_$c.log 2>&1
done
python $R/tools/pmc_summary.py $(find $O/${TAG}_pmc_FETCH_SIZE -name "*.db" | head -1) $(find $O/${TAG}_pmc_WRITE_SIZE -name "*.db" | head -1) $O/${TAG}_pmc_FETCH_SIZE.log $O/${TAG}_pmc_summary.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --modeler-steps 0 (256 eta, 4096^2)" > $O/${TAG}_pmc_summary.txt 2>&1
find $O -name "*.db" -size +20M -delete
cd $R
timeout 600 python tests/tools/make_sim_input.py 4096 3 /tmp/sim4096.npz > $O/${TAG}_sim_input.txt 2>&1
timeout 300 python bench.py --dyn-npz /tmp/sim4096.npz --steps 3 --warmup 1 --cpu-pool 0 --cpu-sample 2 --modeler-steps 0 > $O/${TAG}_bench_sim4096.json 2>> $O/${TAG}_bench.err
head -c 300 $O/${TAG}_bench_n1.json; echo; tail -3 $O/${TAG}_bench.err; cat $O/${TAG}_pmc_summary.txt | head -12
