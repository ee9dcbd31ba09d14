#!/bin/bash
TAG=${1:-sweep}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python3 tools/gemm_sweep.py --json $OUT/sweep.json "$@" 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep.log
