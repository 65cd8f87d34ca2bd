#!/bin/bash
# scratch: HBM traffic counters of the mixed sweep (two --pmc passes, --kernel-trace only)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
CMD="bench.py --precision mixed --steps 1 --warmup 0 --no-cpu-baseline --modeler-steps 0"
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 35 rocprofv3 --pmc $c --kernel-trace -d $O/r03mx_pmc_$c -o pmc -- python $R/$CMD > $O/r03mx_pmc_$c.log 2>&1 )
done
python tools/pmc_kernels.py $(find $O/r03mx_pmc_FETCH_SIZE -name "*.db" | head -1) $(find $O/r03mx_pmc_WRITE_SIZE -name "*.db" | head -1) \
   $O/r03mx_pmc_kernels.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python $CMD (256 eta, 4096^2)" 2>&1 | head -20
grep "^{" $O/r03mx_pmc_FETCH_SIZE.log | tail -1 > $O/r03mx_pmc_benchline.json
rm -rf $O/r03mx_pmc_FETCH_SIZE $O/r03mx_pmc_WRITE_SIZE
