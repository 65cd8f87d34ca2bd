export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_edges.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
bash tools/gpu_run.sh quick s24
bash tools/gpu_run.sh quick s24b
bash tools/gpu_run.sh trace s24 > /dev/null
python tools/timeline.py $(find $O/s24_prof_bench -name "*.db" | head -1) --top 6
