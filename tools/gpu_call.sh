#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
for abl in 64; do echo "== abl=$abl"; SCINT_SSPEC_ABL=$abl timeout 100 python tools/experiments/rows2_phase_times.py 4096 2>&1 | grep -v amdgpu; done | tee $O/r06i_cols2_phase_times.txt
