export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_modeler_fullsize.py tests/test_gpu_multirank.py tests/test_gpu_bench_contract.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python bench.py --objective chisq --steps 3 --warmup 1 --no-cpu-baseline > $O/s18_chisq.json 2> $O/s18_chisq.err; python tools/bench_line.py $O/s18_chisq.json
