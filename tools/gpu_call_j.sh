#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python tools/time_revmap.py > $O/j_rev_default.txt 2>&1
SCINT_REV_NOLO=1 python tools/time_revmap.py > $O/j_rev_nolo.txt 2>&1
SCINT_REV_SMALL=1 python tools/time_revmap.py > $O/j_rev_small.txt 2>&1
SCINT_REV_SMALL=1 SCINT_REV_NOLO=1 python tools/time_revmap.py > $O/j_rev_small_nolo.txt 2>&1
tail -n 2 $O/j_rev_*.txt
python tools/time_fft.py > $O/j_fft.txt 2>&1; grep -v amdgpu $O/j_fft.txt
