#!/bin/bash
# scratch per-call script (round 6, back-map A/B): general kernel (SCINT_REV_DIAG=0) against the diagonal kernel
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_modeler_fullsize.py tests/test_gpu_edges.py -m gpu -q -x -k "rev or model or chisq" > $O/r06v_pytest.log 2>&1; tail -3 $O/r06v_pytest.log
{
echo "# python tools/time_revmap.py 4096 0.25 0.5 1 2 4   (one image, API layout, GPU to itself)"
for d in 0 1; do echo "## SCINT_REV_DIAG=$d"; SCINT_REV_DIAG=$d timeout 300 python tools/time_revmap.py 4096 0.25 0.5 1 2 4 2>&1 | grep "rev_map"; done
echo "# python bench.py --objective chisq --steps 4 --warmup 1 --headline-only   (eta/s, ms per step; interleaved)"
for rep in 1 2 3; do for d in 0 1; do
  v=$(SCINT_REV_DIAG=$d timeout 600 python bench.py --objective chisq --steps 4 --warmup 1 --headline-only 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1), round(d['ms_per_step'],1))")
  echo "SCINT_REV_DIAG=$d  $v"
done; done
} > $O/r06_revmap_diag_ab.txt 2>&1
cat $O/r06_revmap_diag_ab.txt
