#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 45 python -m pytest tests/test_gpu_bench_contract.py -m gpu -q -x > $O/r03mx_contract.log 2>&1; echo "pytest rc=$?" >> $O/r03mx_contract.log
tail -6 $O/r03mx_contract.log
