#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_arcfit.py -m gpu -q -x > $O/n_pytest.log 2>&1; tail -2 $O/n_pytest.log
python tools/time_fft.py > $O/n_fft_small1.txt 2>&1
SCINT_FFT_SMALL_BLOCK=0 python tools/time_fft.py > $O/n_fft_small0.txt 2>&1
python tools/time_revmap.py > $O/n_rev.txt 2>&1
grep -v amdgpu $O/n_fft_small1.txt; echo ---; grep -v amdgpu $O/n_fft_small0.txt; tail -2 $O/n_rev.txt
