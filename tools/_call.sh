export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O
for CE in 2 4; do SCINT_CHECK_EVERY=$CE bash tools/gpu_run.sh quick s25_ce$CE; done
for B in 48 96 128; do bash tools/gpu_run.sh quick s25_b$B --batch $B; done
bash tools/gpu_run.sh quick s25_ref
