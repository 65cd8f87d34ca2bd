#!/bin/bash
# scratch: interleaved A/B of the mixed sweep's variants (one GPU box)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
Q="--precision mixed --steps 5 --warmup 1 --no-cpu-baseline --modeler-steps 0"
run() {  # tag, lib, extra args..., env via VAR=..
  tag=$1; lib=$2; shift; shift
  cp variants/$lib.so scintools_amd/libscint_hip.so
  timeout 60 python bench.py $Q "$@" > $O/ab_$tag.json 2> $O/ab_$tag.err || echo "rc=$? $tag"
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/ab_$tag.json") if l.startswith("{")][-1]); r=d["roofline"]; m=r.get("mixed",{})
    print(f"$tag: {d['value']:.0f} eta/s  passes {d['config']['lanczos_steps_mean']:.2f} cert {m.get('certificate_passes_mean',0):.2f} failed {d['config']['failed_etas']} batch {d['config']['batch']}  mv32 {r['achieved']:.0f} GB/s share {r['share_of_step_time']:.3f} launch {r['avg_launch_ms']:.3f} ms")
except Exception as e:
    print("$tag: no line", e)
PY
}
for round in 1 2; do
  run base_$round base
  run sep_$round sep
  run r2_$round r2
  run nofence_$round nofence
  run tol05_$round tol05
  run b100_$round base --batch 100
  run b180_$round base --batch 180
  run b240_$round base --batch 240
  SCINT_CHECK_EVERY=3 run ce3_$round base
done
cp variants/base.so scintools_amd/libscint_hip.so
