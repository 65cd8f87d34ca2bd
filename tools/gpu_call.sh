#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_contract.py -m gpu -q -x > $O/r06l_contract.log 2>&1; tail -3 $O/r06l_contract.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06l_bench_n1.json 2> $O/r06l_bench.err ) 2>&1 | grep real
tail -c 1500 $O/r06l_bench_n1.json; echo; tail -3 $O/r06l_bench.err
