#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/r06t_prof_wf -o wf -- python $R/bench.py --workload wavefield --steps 2 --warmup 1 > $O/r06t_prof_wf.log 2>&1 )
db=$(find $O/r06t_prof_wf -name "*.db" | head -1)
python tools/rocpd_summary.py $db $O/r06t_wf_kernel_stats.csv $O/r06t_wf_kernel_overlap.json > /dev/null 2>&1
head -16 $O/r06t_wf_kernel_stats.csv | cut -c1-150
