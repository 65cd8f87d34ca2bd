#!/bin/bash
# long-row real-to-complex path: parity + 8192^2 timings
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/r_pytest.log 2>&1; tail -3 $O/r_pytest.log
python tools/time_fft.py 8192 4096 > $O/r_fft.txt 2>&1; grep -v amdgpu $O/r_fft.txt
