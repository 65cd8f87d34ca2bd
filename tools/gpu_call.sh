#!/bin/bash
# scratch (round 6, call 10): branch-free grouped logarithms, means folded into the column kernel
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cp scintools_amd/libscint_hip.so /tmp/default.so
{
for rep in 1 2; do
for v in default lg2 lg8 lg16; do
  if [ $v = default ]; then cp /tmp/default.so scintools_amd/libscint_hip.so; else cp variants/$v.so scintools_amd/libscint_hip.so; fi
  echo "== $v: $(timeout 100 python tools/time_fft.py 4096 8192 2048 sspec prewhite 2>&1 | grep -v amdgpu | cut -c1-40 | tr '\n' ' ')"
done
done
cp /tmp/default.so scintools_amd/libscint_hip.so
echo "== default (4 logarithms per group)"; SCINT_SSPEC_ABL=32 timeout 100 python tools/experiments/rows2_phase_times.py 4096 2>&1 | grep -v amdgpu
} | tee $O/r06j_log_groups_ab.txt
timeout 600 python -m pytest tests -m gpu -q -x -k "sspec or arcfit or secondary" > $O/r06j_pytest_sspec.log 2>&1; tail -2 $O/r06j_pytest_sspec.log
