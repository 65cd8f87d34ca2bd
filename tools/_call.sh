export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/probes/pk2e_probe.hip -o /tmp/pk2e_probe
/tmp/pk2e_probe 50 | tee $O/s8_pk2e_probe.txt
timeout 900 python -m pytest tests/test_gpu_edges.py tests/test_gpu_parity.py tests/test_gpu_stopping_rule.py -m gpu -q -x 2>&1 | tail -3
bash tools/gpu_run.sh quick s8
SCINT_SWEEP_GROUPS=1 bash tools/gpu_run.sh quick s8g1
