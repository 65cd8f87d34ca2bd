#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 20 python bench.py --precision mixed --steps 5 --warmup 1 --no-cpu-baseline --modeler-steps 0 > $O/r03mx_slim.json 2> $O/r03mx_slim.err
python tools/bench_line.py $O/r03mx_slim.json
