#!/bin/bash
# FFT instruction diet + tile order + split exchange: parity tests, then timings of the variants
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_arcfit.py tests/test_gpu_edges.py tests/test_gpu_modeler_fullsize.py -m gpu -q -x > $O/o_pytest.log 2>&1; tail -3 $O/o_pytest.log
python tools/time_fft.py > $O/o_fft_default.txt 2>&1
SCINT_FFT_TILE_ORDER=0 python tools/time_fft.py > $O/o_fft_order0.txt 2>&1
SCINT_FFT_SPLIT=0 python tools/time_fft.py > $O/o_fft_split0.txt 2>&1
for f in default order0 split0; do echo "--- $f"; grep -v amdgpu $O/o_fft_$f.txt; done
