export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O; R=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_arcfit.py -m gpu -q -x -k "sspec or arc or norm" 2>&1 | tail -2
timeout 300 python tools/time_fft.py sspec prewhite 2>&1 | tail -6
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/s15_prof -o fft -- python $R/tools/time_fft.py 4096 sspec > $O/s15_prof.log 2>&1 )
db=$(find $O/s15_prof -name "*.db" | head -1); python tools/rocpd_summary.py $db $O/s15.csv > /dev/null
grep -E 'sspec' $O/s15.csv | awk -F'",' '{split($2,a,","); n=$1; sub(/.*scint::/,"",n); sub(/[<(].*/,"",n); print n, a[3]}'
find $O -name "*.db" -delete
