#!/bin/bash
# scratch per-call script (round 6): delay rows per workgroup of the fused diagonal back-map (LDS beside two mat-vec workgroups: <= 16 KiB = 768 rows)
R=$PWD; O=$R/gpurun_out; mkdir -p $O
run() { timeout 600 python bench.py --objective chisq --steps 4 --warmup 1 --headline-only "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1), round(d['ms_per_step'],1))"; }
{
echo "# python bench.py --objective chisq --steps 4 --warmup 1 --headline-only  (eta/s, ms per step), libraries swapped in, interleaved"
for rep in 1 2 3; do for v in head slab768 slab1536; do cp variants/$v.so scintools_amd/libscint_hip.so; echo "$v: $(run)"; done; done
cp variants/head.so scintools_amd/libscint_hip.so
} > $O/r06_chisq_fused_slab_ab.txt 2>&1
cat $O/r06_chisq_fused_slab_ab.txt
