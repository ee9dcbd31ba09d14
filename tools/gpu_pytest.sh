#!/bin/bash
# Usage: bash tools/gpu_pytest.sh <tag> [pytest args...]
TAG=${1:-pt}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest "$@" --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 60 $OUT/pytest.log | cut -c1-400
