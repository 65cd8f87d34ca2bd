#!/bin/bash
# Second GPU call for the wide-block paths: kernel trace and PMC traffic of ONE variant (the winner of
# tools/gpu_wide_blocks.sh), chosen by the environment of this call, e.g.
#   gpurun --timeout 1200 -- 'SCINT_LANCZOS_BLOCK=8 bash tools/gpu_wide_profile.sh b8'
# Writes gpurun_out/wp_<tag>_*: bench line (20 steps), per-kernel statistics with interval unions and
# co-residency (tools/rocpd_summary.py), FETCH_SIZE / WRITE_SIZE per kernel (tools/pmc_kernels.py).
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
TAG=${1:-variant}
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/wp_${TAG}_bench_n1.json 2> $O/wp_${TAG}_bench.err; echo "bench rc=$?"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/wp_${TAG}_prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --modeler-steps 0 > $O/wp_${TAG}_prof.log 2>&1
db=$(find $O/wp_${TAG}_prof -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $db $O/wp_${TAG}_kernel_stats.csv $O/wp_${TAG}_kernel_overlap.json > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace -d $O/wp_${TAG}_pmc_$c -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --modeler-steps 0 > $O/wp_${TAG}_pmc_$c.log 2>&1
done
python $R/tools/pmc_kernels.py $(find $O/wp_${TAG}_pmc_FETCH_SIZE -name "*.db" | head -1) $(find $O/wp_${TAG}_pmc_WRITE_SIZE -name "*.db" | head -1) $O/wp_${TAG}_pmc_kernels.json "bench.py --steps 1 --warmup 0 --no-cpu-baseline --modeler-steps 0 ($TAG)" > $O/wp_${TAG}_pmc.txt 2>&1
find $O -name "*.db" -size +20M -delete
cd $R
head -c 400 $O/wp_${TAG}_bench_n1.json; echo; head -20 $O/wp_${TAG}_kernel_stats.csv; head -20 $O/wp_${TAG}_pmc.txt
