#!/bin/bash
# scratch script of one GPU call (round 6, call 1): persistent calc_sspec kernels against round 5's, interleaved
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
{
for rep in 1 2; do
  for v in "1 1" "2 1" "1 2" "2 2"; do
    set -- $v
    echo "== rows=$1 cols=$2 (rep $rep)"
    SCINT_SSPEC_ROWS=$1 SCINT_SSPEC_COLS=$2 timeout 120 python tools/time_fft.py 4096 8192 2048 sspec prewhite
  done
done
} > $O/r06a_sspec_ab.txt 2>&1
tail -40 $O/r06a_sspec_ab.txt
timeout 600 python -m pytest tests -m gpu -q -x -k "sspec or arcfit or secondary" > $O/r06a_pytest_sspec.log 2>&1; tail -3 $O/r06a_pytest_sspec.log
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $O/r06a_prof_fft -o fft -- python $R/tools/time_fft.py 4096 8192 sspec > $O/r06a_prof_fft.log 2>&1 )
db=$(find $O/r06a_prof_fft -name "*.db" | head -1)
python tools/rocpd_summary.py $db $O/r06a_fft_kernel_stats.csv $O/r06a_fft_kernel_overlap.json > /dev/null 2>&1
head -12 $O/r06a_fft_kernel_stats.csv | cut -c1-200
