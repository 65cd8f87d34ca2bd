#!/bin/bash
# split exchange for every length (SCINT_FFT_SPLIT=2), two-pass columns at every size; sspec kernel breakdown
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
SCINT_FFT_SPLIT=2 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_arcfit.py -m gpu -q -x > $O/p_pytest.log 2>&1; tail -2 $O/p_pytest.log
SCINT_FFT_SPLIT=2 python tools/time_fft.py > $O/p_fft_split2.txt 2>&1
SCINT_FFT_TWO_PASS=2 python tools/time_fft.py > $O/p_fft_twopass2.txt 2>&1
SCINT_FFT_SPLIT=2 SCINT_FFT_TWO_PASS=2 python tools/time_fft.py > $O/p_fft_both.txt 2>&1
for f in split2 twopass2 both; do echo "--- $f"; grep -v amdgpu $O/p_fft_$f.txt; done
cd /tmp
for sz in 4096 2048; do
timeout 300 rocprofv3 --kernel-trace --stats -d $O/p_prof_$sz -o fft -- python $R/tools/time_fft.py $sz sspec > $O/p_prof_$sz.log 2>&1
db=$(find $O/p_prof_$sz -name "*.db" | head -1); python $R/tools/rocpd_summary.py $db $O/p_sspec${sz}_kernels.csv > /dev/null
cut -c1-150 $O/p_sspec${sz}_kernels.csv | head -12
done
rm -rf $O/p_prof_4096 $O/p_prof_2048
