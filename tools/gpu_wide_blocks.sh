#!/bin/bash
# First GPU call for the opt-in wide-block Lanczos paths: none of them has
# run on a GPU.  Correctness first (one small test per variant), then the headline bench per variant,
# interleaved with the default so that box-to-box variation cancels.  ~6 GPU-minutes.
#   gpurun --timeout 900 -- 'bash tools/gpu_wide_blocks.sh'
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
# 0. the matrix-core instruction itself: operand layout and rate on this GPU
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/probes/mfma_f64_probe.hip -o /tmp/mfma_probe 2> /dev/null \
  && timeout 60 /tmp/mfma_probe | tee $O/wb_mfma_probe.txt
SCINT_TEST_WIDE_BLOCKS=1 timeout 300 python -m pytest tests/test_gpu_edges.py -m gpu -q -k "wide_blocks or tiny" \
    > $O/wb_pytest.log 2>&1; echo "pytest rc=$?" >> $O/wb_pytest.log
grep -E "passed|failed|^FAILED|rc=" $O/wb_pytest.log | tail -8
run() {  # tag, block, matvec mode
  SCINT_LANCZOS_BLOCK=$2 SCINT_MATVEC_MFMA=$3 timeout 240 python bench.py --steps 5 --warmup 2 --no-cpu-baseline \
      --modeler-steps 0 > $O/wb_bench_$1.json 2>> $O/wb_bench.err || echo "bench $1 failed rc=$?"
}
for rep in 1 2; do
  run default_$rep 2 0
  SCINT_PK2_PREFETCH=1 run default_uncond_$rep 2 0   # same kernel with unconditional prefetch loads (exact wait counts)
  run b8_$rep 8 0                                  # banded mat-vec (4 block rows per workgroup)
  SCINT_Q_BAND=2 run b8band2_$rep 8 0              # two block rows per workgroup (64 KiB of LDS)
  SCINT_Q_BAND=1 run b8strip4_$rep 8 0             # plain strips of 4 tiles
  SCINT_Q_BAND=1 SCINT_Q_STRIP=8 run b8strip8_$rep 8 0
  SCINT_Q_BAND=4 run b4band_$rep 4 2
  SCINT_Q_BAND=4 run b2band_$rep 2 2               # the default two-vector recurrence on the matrix cores, banded
  run b4q_$rep 4 2
  run b4m_$rep 4 1
  run b4v_$rep 4 0
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob('gpurun_out/wb_bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as exc:
        print(f, 'no line', exc); continue
    c, r = d['config'], d['roofline']
    print(f.split('wb_bench_')[1][:-5], round(d['value'], 1), 'eta/s  passes', round(c['lanczos_steps_mean'], 2),
          'failed', c['failed_etas'], 'fit', c['eta_fit_over_true'], 'matvec GB/s', round(r['achieved']),
          'share', round(r['share_of_step_time'], 3))
PY
