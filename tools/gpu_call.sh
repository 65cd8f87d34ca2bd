#!/bin/bash
# scratch (round 6): the two bounded experiments on the headline's bytes
R=$PWD; O=$R/gpurun_out; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/probes/pk3_probe.hip -o /tmp/pk3_probe && timeout 120 /tmp/pk3_probe | tee $O/r06o_pk3_probe.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/probes/indexed_pass_probe.hip -o /tmp/indexed_pass_probe && timeout 120 /tmp/indexed_pass_probe | tee $O/r06o_indexed_pass_probe.txt
