#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
{
for rep in 1 2; do for d in 0 1; do echo "== direct=$d: $(SCINT_SSPEC_DIRECT=$d timeout 100 python tools/time_fft.py 4096 8192 sspec 2>&1 | grep -v amdgpu | cut -c1-40 | tr '\n' ' ')"; done; done
for d in 0 1; do
( cd /tmp && SCINT_SSPEC_DIRECT=$d timeout 100 rocprofv3 --kernel-trace --stats -d $O/r06k_d$d -o fft -- python $R/tools/time_fft.py 4096 8192 sspec > $O/r06k_d$d.log 2>&1 )
python tools/rocpd_summary.py $(find $O/r06k_d$d -name "*.db" | head -1) $O/r06k_d$d.csv $O/r06k_d$d.json > /dev/null 2>&1
echo "direct=$d"; grep -E "cols2|prep" $O/r06k_d$d.csv | cut -d, -f1-5 | cut -c1-110
done
} | tee $O/r06k_direct_ab.txt
