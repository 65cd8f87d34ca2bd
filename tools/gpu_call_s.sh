#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
SCINT_FFT_PAIR_SPLIT=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "sspec or conjugate or acf or fullsize" > $O/s_pytest.log 2>&1; tail -3 $O/s_pytest.log
python tools/time_fft.py 2048 4096 sspec cs cs3 > $O/s_fft0.txt 2>&1
SCINT_FFT_PAIR_SPLIT=1 python tools/time_fft.py 2048 4096 sspec cs cs3 > $O/s_fft1.txt 2>&1
echo "--- pair store"; grep -v amdgpu $O/s_fft0.txt; echo "--- pair split"; grep -v amdgpu $O/s_fft1.txt
