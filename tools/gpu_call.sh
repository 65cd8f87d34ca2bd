#!/bin/bash
# scratch per-call script (round 6): the diagonal back-map with its per-diagonal geometry slimmed (one pass, hardware reciprocal) against HEAD
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
run() { timeout 600 python bench.py --objective chisq --steps 4 --warmup 1 --headline-only "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1), round(d['ms_per_step'],1))"; }
{
echo "# python bench.py --objective chisq --steps 4 --warmup 1 --headline-only  (eta/s, ms per step), libraries swapped in, interleaved"
for rep in 1 2 3; do for v in slim slim2; do cp variants/$v.so scintools_amd/libscint_hip.so; echo "$v: $(run)"; done; done
for v in slim slim2; do cp variants/$v.so scintools_amd/libscint_hip.so; echo "## $v: python tools/time_revmap.py 4096 0.25 1 4"; timeout 300 python tools/time_revmap.py 4096 0.25 1 4 2>&1 | grep rev_map; done
cp variants/slim2.so scintools_amd/libscint_hip.so
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d $O/r06z2_ctr -o ctr -- python $R/tools/time_revmap.py 4096 0.25 1 4 > $O/r06z2_ctr.log 2>&1 )
echo "## slim2: rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU of python tools/time_revmap.py 4096 0.25 1 4"
python tools/pmc_any.py $(find $O/r06z2_ctr -name "*.db" | head -1) rev_diag
} > $O/r06_revmap_slim2_ab.txt 2>&1
cat $O/r06_revmap_slim2_ab.txt
