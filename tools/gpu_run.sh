#!/bin/bash
# The ONE script behind every GPU measurement of this repository (profiles/README.md names the
# sub-command and tag of each committed file).  Run from the repository root on the GPU box:
#   gpurun --timeout 900 -- 'bash tools/gpu_run.sh <sub-command> [tag] [extra bench.py arguments]'
# Output goes to gpurun_out/<tag>_*; summaries worth keeping are copied into profiles/ by hand.
#   suite    [tag]          pytest -m gpu (whole suite, durations)
#   bench    [tag] [args]   the driver's bench command (--gpus 1 --steps 20 --warmup 5) + extra args
#   quick    [tag] [args]   bench.py --steps 5 --warmup 2 --headline-only (no CPU baseline, no modeler / mixed / one-slot-group legs) + extra args
#   configs  [tag]          BASELINE configs 4 / 5, npad = 3 and the 4096^2 reference-Simulation screen
#   trace    [tag] [args]   rocprofv3 --kernel-trace --stats of a 3-step bench -> per-kernel stats + interval unions
#   modeler  [tag]          the same trace of the chi^2 (modeler) objective
#   pmc      [tag] [args]   FETCH_SIZE / WRITE_SIZE in separate --pmc passes of a 1-step bench (256 eta)
#   pmc_modeler [tag]       the same two PMC passes of the chi^2 (modeler) objective -> <tag>_pmc_modeler_summary.json (whole-step traffic / algorithmic)
#   mixedev  [tag]          evidence set of the mixed sweep with THIS library: kernel trace, FETCH/WRITE PMC, six hardware counters
#   counters [tag] [args]   two --pmc passes (instruction counts + activity; cycles + waits) of a 1-step bench, tools/pmc_any.py
#   fft      [tag]          kernel trace + PMC passes of tools/time_fft.py (calc_sspec and CS, 2048^2 .. 8192^2)
#   sspecroof [tag]         FETCH / WRITE / instruction-count passes of calc_sspec alone at 4096^2 and 8192^2 -> <tag>_sspec_roofline_<size>.json (bench.py: sspec.roofline)
#   probes   [tag]          tools/probes/*.hip (stream ceiling, the round-2 and round-3 mat-vec loops with their parts switchable, f64 MFMA layout)
#   mixed    [tag] [args]   the mixed-precision sweep: a 3-step bench with its --mixed-steps leg (rate, bytes by operand, curve against
#                           the float64 one), then tests/test_gpu_zz_mixed.py
#   workloads [tag]         bench.py --workload fit_thetatheta | wavefield | tutorial_fit | fit_arc (the (f) rows end to end, CPU port beside them)
#   all      [tag]          suite, bench, configs, trace, modeler, pmc, pmc_modeler, counters, fft, sspecroof, workloads, mixed  (the closing call of a round)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
CMD=${1:-all}; TAG=${2:-m}; shift; shift || true
EXTRA="$*"
QUICK="--headline-only"

suite() {
  timeout 1200 python -m pytest tests -m gpu -q --durations=8 > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.log
  grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/${TAG}_pytest.log | tail -8
}
bench() {
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 $EXTRA > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"
  head -c 400 $O/${TAG}_bench_n1.json; echo; tail -2 $O/${TAG}_bench.err
}
quick() {
  timeout 300 python bench.py --steps 5 --warmup 2 $QUICK $EXTRA > $O/${TAG}_quick.json 2> $O/${TAG}_quick.err; echo "quick rc=$?"
  python tools/bench_line.py $O/${TAG}_quick.json
}
configs() {
  timeout 300 python bench.py --size 2048 --obs-total 64 --steps 2 --warmup 1 $QUICK > $O/${TAG}_bench_cfg4_64obs.json 2>> $O/${TAG}_bench.err
  timeout 300 python bench.py --size 8192 --neta 64 --steps 2 --warmup 1 $QUICK > $O/${TAG}_bench_cfg5_8192.json 2>> $O/${TAG}_bench.err
  timeout 300 python bench.py --npad 3 --steps 2 --warmup 1 $QUICK > $O/${TAG}_bench_npad3.json 2>> $O/${TAG}_bench.err
  timeout 600 python tests/tools/make_sim_input.py 4096 3 /tmp/sim4096.npz > $O/${TAG}_sim_input.txt 2>&1
  timeout 300 python bench.py --dyn-npz /tmp/sim4096.npz --steps 3 --warmup 1 --cpu-pool 0 --cpu-sample 2 --modeler-steps 0 > $O/${TAG}_bench_sim4096.json 2>> $O/${TAG}_bench.err
  for f in cfg4_64obs cfg5_8192 npad3 sim4096; do python tools/bench_line.py $O/${TAG}_bench_$f.json; done
}
trace_of() {  # name, bench arguments
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_$1 -o bench -- python $R/bench.py $2 > $O/${TAG}_prof_$1.log 2>&1 )
  db=$(find $O/${TAG}_prof_$1 -name "*.db" | head -1)
  python tools/rocpd_summary.py $db $O/${TAG}_$1_kernel_stats.csv $O/${TAG}_$1_kernel_overlap.json > /dev/null
  head -12 $O/${TAG}_$1_kernel_stats.csv | cut -c1-220
  python tools/timeline.py $db --top 8 > $O/${TAG}_$1_timeline.txt 2>&1; head -9 $O/${TAG}_$1_timeline.txt
}
trace()   { trace_of bench "--steps 3 --warmup 1 $QUICK $EXTRA"; }
modeler() { trace_of modeler "--objective chisq --steps 2 --warmup 1 --no-cpu-baseline"; }
pmc_of() {  # name, command (relative to the repo root), description
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace -d $O/${TAG}_pmc_$1_$c -o pmc -- python $R/$2 > $O/${TAG}_pmc_$1_$c.log 2>&1 )
  done
  python tools/pmc_kernels.py $(find $O/${TAG}_pmc_$1_FETCH_SIZE -name "*.db" | head -1) $(find $O/${TAG}_pmc_$1_WRITE_SIZE -name "*.db" | head -1) \
      $O/${TAG}_pmc_$1_kernels.json "$3" > $O/${TAG}_pmc_$1.txt 2>&1
  head -24 $O/${TAG}_pmc_$1.txt
}
pmc() {
  pmc_of bench "bench.py --steps 1 --warmup 0 $QUICK $EXTRA" "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 1 --warmup 0 $QUICK $EXTRA (256 eta, 4096^2)"
  python tools/pmc_summary.py $(find $O/${TAG}_pmc_bench_FETCH_SIZE -name "*.db" | head -1) $(find $O/${TAG}_pmc_bench_WRITE_SIZE -name "*.db" | head -1) \
      $O/${TAG}_pmc_bench_FETCH_SIZE.log $O/${TAG}_pmc_summary.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 1 --warmup 0 $QUICK (256 eta, 4096^2)" > $O/${TAG}_pmc_summary.txt 2>&1
  head -12 $O/${TAG}_pmc_summary.txt
}
pmc_modeler() {
  pmc_of modeler "bench.py --objective chisq --steps 1 --warmup 0 --no-cpu-baseline" "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --objective chisq --steps 1 --warmup 0 --no-cpu-baseline (256 eta, 4096^2)"
  python tools/pmc_modeler_summary.py $(find $O/${TAG}_pmc_modeler_FETCH_SIZE -name "*.db" | head -1) $(find $O/${TAG}_pmc_modeler_WRITE_SIZE -name "*.db" | head -1) \
      $O/${TAG}_pmc_modeler_FETCH_SIZE.log $O/${TAG}_pmc_modeler_summary.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --objective chisq --steps 1 --warmup 0 --no-cpu-baseline (256 eta, 4096^2)" > $O/${TAG}_pmc_modeler_summary.txt 2>&1
  head -12 $O/${TAG}_pmc_modeler_summary.txt
}
counters_of() {  # name, command: the counters of profiles/r03_*_counters.txt in two passes (8 SQ slots + 2 GRBM slots per pass on gfx950)
  i=0
  for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/${TAG}_ctr_$1_$i -o ctr -- python $R/$2 > $O/${TAG}_ctr_$1_$i.log 2>&1 )
    python tools/pmc_any.py $(find $O/${TAG}_ctr_$1_$i -name "*.db" | head -1) >> $O/${TAG}_$1_counters_raw.txt 2>&1
  done
  grep -E "matvec|cert_resid|gather|rev_gather|sspec" $O/${TAG}_$1_counters_raw.txt | head -40
}
counters() { rm -f $O/${TAG}_bench_counters_raw.txt; counters_of bench "bench.py --steps 1 --warmup 0 $QUICK $EXTRA"; }
mixedev() {
  trace_of mixed "--precision mixed --steps 3 --warmup 1 $QUICK"
  pmc_of mixed "bench.py --precision mixed --steps 1 --warmup 0 $QUICK" "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --precision mixed --steps 1 --warmup 0 $QUICK"
  grep '"metric"' $O/${TAG}_pmc_mixed_FETCH_SIZE.log | tail -1 > $O/${TAG}_mixed_pmc_benchline.json
  rm -f $O/${TAG}_mixed_counters_raw.txt; counters_of mixed "bench.py --precision mixed --steps 1 --warmup 0 $QUICK"
}
fft() {
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_fft -o fft -- python $R/tools/time_fft.py > $O/${TAG}_prof_fft.log 2>&1 )
  db=$(find $O/${TAG}_prof_fft -name "*.db" | head -1); python tools/rocpd_summary.py $db $O/${TAG}_fft_kernel_stats.csv > /dev/null
  grep -E "sspec|cs " $O/${TAG}_prof_fft.log | head -12
  pmc_of fft "tools/time_fft.py" "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python tools/time_fft.py"
}
sspecroof() {  # HBM bytes + instruction counts of calc_sspec's kernels at 4096^2 and 8192^2 -> <tag>_sspec_roofline_<size>.json (tools/sspec_roofline.py)
  for n in 4096 8192; do
    i=0
    for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
      i=$((i+1))
      ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/${TAG}_sspecroof_${n}_$i -o p -- python $R/tools/time_fft.py $n sspec > $O/${TAG}_sspecroof_${n}_$i.log 2>&1 )
    done
    python tools/sspec_roofline.py $(find $O/${TAG}_sspecroof_${n}_1 -name "*.db" | head -1) $(find $O/${TAG}_sspecroof_${n}_2 -name "*.db" | head -1) \
        $(find $O/${TAG}_sspecroof_${n}_3 -name "*.db" | head -1) $n $O/${TAG}_sspec_roofline_$n.json
  done
}
workloads() {  # the (f) rows end to end: bench.py --workload X -> <tag>_wl_X.json (one line each, CPU port on a sample beside the GPU time)
  for w in fit_thetatheta wavefield tutorial_fit fit_arc; do
    timeout 900 python bench.py --workload $w --steps 3 --warmup 1 > $O/${TAG}_wl_$w.json 2> $O/${TAG}_wl_$w.err; echo "workload $w rc=$?"
    python -c "
import json
d=json.loads([l for l in open('$O/${TAG}_wl_$w.json') if l.startswith('{')][-1])
print('  ', round(d['value'],4), 's; cpu port', round(d['cpu_baseline']['value'],1), 's; x', round(d['speedup_vs_cpu_baseline']), {k: round(v['busy_share_of_wall'],3) for k,v in d.get('kernels',{}).items()})"
  done
}
mixed() {
  timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --modeler-steps 0 --mixed-steps 3 $EXTRA > $O/${TAG}_mixed.json 2> $O/${TAG}_mixed.err; echo "bench rc=$?"
  python tools/bench_line.py $O/${TAG}_mixed.json; python -c "
import json,sys
d=json.loads([l for l in open('$O/${TAG}_mixed.json') if l.startswith('{')][-1]); m=d.get('mixed_precision',{})
print({k:m.get(k) for k in ('value','speedup_vs_f64','failed_etas','max_rel_diff_vs_f64_curve','lanczos_steps_mean','certificate_passes_mean')}); print(m.get('matvec32')); print(m.get('matvec64'))"
  tail -3 $O/${TAG}_mixed.err
  timeout 170 python -m pytest tests/test_gpu_zz_mixed.py -m gpu -q -x --durations=5 > $O/${TAG}_mixed_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_mixed_pytest.log
  grep -E "passed|failed|^FAILED|^ERROR|rc=|Error|assert" $O/${TAG}_mixed_pytest.log | tail -12
}
probes() {
  for p in stream_probe pk2_probe pk2e_probe mfma_f64_probe; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/probes/$p.hip -o /tmp/$p 2> /dev/null && timeout 120 /tmp/$p | tee $O/${TAG}_$p.txt
  done
}
case $CMD in
  suite|bench|quick|configs|trace|modeler|pmc|pmc_modeler|counters|mixedev|fft|sspecroof|workloads|probes|mixed) $CMD ;;
  all) suite; bench; configs; trace; modeler; pmc; pmc_modeler; counters; fft; sspecroof; workloads; mixed ;;
  *) echo "unknown sub-command $CMD"; exit 2 ;;
esac
find $O -name "*.db" -size +20M -delete
