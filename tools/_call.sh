export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_edges.py tests/test_gpu_parity.py tests/test_gpu_multirank.py -m gpu -q -x 2>&1 | tail -2
bash tools/gpu_run.sh quick s28
bash tools/gpu_run.sh quick s28b
