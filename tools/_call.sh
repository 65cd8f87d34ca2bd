export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O; R=$PWD
SCINT_SSPEC_ROWS1=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "sspec" 2>&1 | tail -2
for T in 1 0; do
SCINT_SSPEC_ROWS1=$T timeout 300 python tools/time_fft.py sspec prewhite 2>&1 | grep sspec | tr '\n' ';'; echo
( cd /tmp && SCINT_SSPEC_ROWS1=$T timeout 300 rocprofv3 --kernel-trace --stats -d $O/s27_prof_$T -o fft -- python $R/tools/time_fft.py 4096 sspec > $O/s27_prof_$T.log 2>&1 )
db=$(find $O/s27_prof_$T -name "*.db" | head -1); python tools/rocpd_summary.py $db $O/s27_$T.csv > /dev/null
echo "rows1=$T: $(grep -E 'sspec_' $O/s27_$T.csv | awk -F'",' '{split($2,a,","); n=$1; sub(/.*scint::/,"",n); sub(/[<(].*/,"",n); print n, a[3]}' | tr '\n' ' ')"
done
find $O -name "*.db" -size +5M -delete
