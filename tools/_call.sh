export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O
bash tools/gpu_run.sh suite r03b
bash tools/gpu_run.sh bench r03b
