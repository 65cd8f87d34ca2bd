export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_modeler_fullsize.py tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -q -x 2>&1 | tail -3
timeout 100 python tools/time_revmap.py 2>&1 | tail -2
timeout 300 python bench.py --objective chisq --steps 3 --warmup 1 --no-cpu-baseline > $O/c7_chisq.json 2> $O/c7.err; python tools/bench_line.py $O/c7_chisq.json
