export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O
bash tools/gpu_run.sh quick s30nt
SCINT_SWEEP_GROUPS=1 bash tools/gpu_run.sh quick s30nt_g1
bash tools/gpu_run.sh quick s30nt_b
