export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_edges.py tests/test_gpu_parity.py tests/test_gpu_stopping_rule.py -m gpu -q -x -k "not headline" 2>&1 | tail -3
bash tools/gpu_run.sh quick s20
SCINT_SWEEP_GROUPS=1 bash tools/gpu_run.sh quick s20g1
