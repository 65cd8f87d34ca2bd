#!/bin/bash
# scratch per-call script (round 6): the chi^2 objective with the fused tail -- tail lanes, curvatures per tail batch, resident curvatures
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cp scintools_amd/libscint_hip.so /tmp/head.so
run() { timeout 600 python bench.py --objective chisq --steps 4 --warmup 1 --headline-only "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1), round(d['ms_per_step'],1))"; }
{
echo "# python bench.py --objective chisq --steps 4 --warmup 1 --headline-only  (eta/s, ms per step), library variants swapped in, interleaved"
for rep in 1 2; do
  for v in head lanes1 lanes3 revbatch16; do
    if [ $v = head ]; then cp /tmp/head.so scintools_amd/libscint_hip.so; else cp variants/$v.so scintools_amd/libscint_hip.so; fi
    echo "$v: $(run)"
  done
done
cp /tmp/head.so scintools_amd/libscint_hip.so
echo "# resident curvatures (--batch), HEAD library"
for b in 48 64 80 96 112 128; do echo "batch $b: $(run --batch $b)"; done
} > $O/r06_chisq_tail_knobs.txt 2>&1
cat $O/r06_chisq_tail_knobs.txt
