export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O
bash tools/gpu_run.sh suite s19
python tools/time_wavefield.py 2>&1 | tail -2
python tools/_retr_dev.py 2>&1 | tail -6
