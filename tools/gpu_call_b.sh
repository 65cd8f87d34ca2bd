#!/bin/bash
# Round-2 GPU call B: new packed gather (two variants), order-independent rev_map, two-pass column FFT.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/b_pytest.log 2>&1; echo "pytest rc=$?" >> $O/b_pytest.log
tail -25 $O/b_pytest.log
SCINT_GATHER_DEEP=0 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py -m gpu -q -x > $O/b_pytest_shallow.log 2>&1; echo "shallow rc=$?" >> $O/b_pytest_shallow.log; tail -3 $O/b_pytest_shallow.log
SCINT_FFT_TWO_PASS=0 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fft or conjugate or sspec" > $O/b_pytest_fftold.log 2>&1; tail -2 $O/b_pytest_fftold.log
for deep in 1 0; do for g in 1 2; do
  SCINT_GATHER_DEEP=$deep SCINT_SWEEP_GROUPS=$g timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --modeler-steps 0 > $O/b_bench_deep${deep}_g$g.json 2>> $O/b_bench.err
done; done
for tp in 1 0; do SCINT_FFT_TWO_PASS=$tp timeout 300 python tools/time_fft.py > $O/b_fft_tp$tp.txt 2>&1; done
timeout 300 python tools/time_modeler.py 4096 > $O/b_modeler.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/b_prof_mod -o bench -- python $R/bench.py --objective chisq --steps 2 --warmup 1 --no-cpu-baseline > $O/b_prof_mod.log 2>&1
db=$(find $O/b_prof_mod -name "*.db" | head -1); python $R/tools/rocpd_summary.py $db $O/b_kernel_stats_mod.csv $O/b_kernel_overlap_mod.json > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $O/b_prof_fft -o fft -- python $R/tools/time_fft.py > $O/b_prof_fft.log 2>&1
db=$(find $O/b_prof_fft -name "*.db" | head -1); python $R/tools/rocpd_summary.py $db $O/b_kernel_stats_fft.csv > /dev/null
find $O -name "*.db" -size +30M -delete
cd $R; cat $O/b_fft_tp1.txt; cat $O/b_modeler.txt | tail -5
