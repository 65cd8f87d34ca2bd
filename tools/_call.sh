export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
for L in 4 8; do
SCINT_STRIP_LEN=$L SCINT_SWEEP_GROUPS=1 bash tools/gpu_run.sh quick s3g1_len$L
done
SCINT_STRIP_LEN=8 bash tools/gpu_run.sh quick s3_len8
