#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 600 python tools/experiments/wavefield_profile.py 2>&1 | grep -v amdgpu | cut -c1-150 | tee $O/r06m_wavefield_profile.txt | head -70
