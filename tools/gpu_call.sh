#!/bin/bash
# scratch per-call script (round 6, back-map A/B): slab rows and stride limit of the diagonal kernel
R=$PWD; O=$R/gpurun_out; mkdir -p $O
{
echo "# python tools/time_revmap.py 4096 0.25 0.5 1 2 4   (one image, API layout, GPU to itself)"
for cfg in "1024 4" "512 4" "2048 4" "1024 2" "1024 8" "1024 1"; do set -- $cfg; echo "## SCINT_DIAG_SLAB_ROWS=$1 SCINT_DIAG_STRIDE=$2"; SCINT_DIAG_SLAB_ROWS=$1 SCINT_DIAG_STRIDE=$2 timeout 300 python tools/time_revmap.py 4096 0.25 0.5 1 2 4 2>&1 | grep "rev_map"; done
echo "# python bench.py --objective chisq --steps 4 --warmup 1 --headline-only   (eta/s, ms per step; interleaved)"
for rep in 1 2; do for cfg in "1024 4" "512 4" "2048 4" "1024 2" "1024 8"; do set -- $cfg
  v=$(SCINT_DIAG_SLAB_ROWS=$1 SCINT_DIAG_STRIDE=$2 timeout 600 python bench.py --objective chisq --steps 4 --warmup 1 --headline-only 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1), round(d['ms_per_step'],1))")
  echo "slab $1 stride $2:  $v"
done; done
} > $O/r06_revmap_diag_knobs.txt 2>&1
cat $O/r06_revmap_diag_knobs.txt
