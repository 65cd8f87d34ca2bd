set -u
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/probes/stream_probe.hip -o /tmp/stream_probe && timeout 120 /tmp/stream_probe | tee $O/r03_stream_probe.txt
B="python bench.py --warmup 2 --no-cpu-baseline --modeler-steps 0"
timeout 200 $B --steps 5 > $O/t_obs1.json 2>$O/t.err
timeout 200 $B --steps 3 --obs 4 > $O/t_obs4.json 2>>$O/t.err
timeout 200 $B --steps 5 --neta 1024 > $O/t_neta1024.json 2>>$O/t.err
SCINT_STRIP_LEN=8 timeout 200 $B --steps 5 > $O/t_strip8.json 2>>$O/t.err
SCINT_STRIP_LEN=4 timeout 200 $B --steps 5 > $O/t_strip4.json 2>>$O/t.err
python - <<'PY'
import glob, json
for f in sorted(glob.glob('gpurun_out/t_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as exc:
        print(f, 'no line', exc); continue
    c, r = d['config'], d['roofline']
    print(f, round(d['value'], 1), 'eta/s  passes', round(c['lanczos_steps_mean'], 2), 'matvec GB/s', round(r['achieved']), 'share', round(r['share_of_step_time'], 3), 'ms', round(d['ms_per_step'],1))
PY
tail -3 $O/t.err
