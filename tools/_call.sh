export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O; R=$PWD
SCINT_SSPEC_ROWS8=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "sspec" 2>&1 | tail -2
for T in 1 0; do
SCINT_SSPEC_ROWS8=$T timeout 300 python tools/time_fft.py sspec prewhite 2>&1 | grep sspec | tr '\n' ';'; echo
( cd /tmp && SCINT_SSPEC_ROWS8=$T timeout 300 rocprofv3 --kernel-trace --stats -d $O/s29_prof_$T -o fft -- python $R/tools/time_fft.py 4096 8192 sspec > $O/s29_prof_$T.log 2>&1 )
db=$(find $O/s29_prof_$T -name "*.db" | head -1); python tools/rocpd_summary.py $db $O/s29_$T.csv > /dev/null
echo "rows8=$T: $(grep -E 'sspec_' $O/s29_$T.csv | awk -F'",' '{split($2,a,","); n=$1; sub(/.*scint::/,"",n); sub(/\(.*/,"",n); print n, a[3]}' | tr '\n' ' ')"
done
find $O -name "*.db" -size +5M -delete
