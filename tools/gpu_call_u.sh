#!/bin/bash
# HBM traffic of the calc_sspec 4096^2 kernels (PMC, separate passes)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/u_pmc_$c -o pmc -- python $R/tools/time_fft.py 4096 sspec > $O/u_pmc_$c.log 2>&1
done
cd $R/tools
python pmc_kernels.py $(find $O/u_pmc_FETCH_SIZE -name "*.db" | head -1) $(find $O/u_pmc_WRITE_SIZE -name "*.db" | head -1) $O/u_pmc_sspec4096.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python tools/time_fft.py 4096 sspec"
find $O -name "*.db" -size +5M -delete
