#!/bin/bash
# residual prefetch in the bf16 GEMM epilogue: bit-identity test, isolated A/B (burst + sustained), C5 A/B (NOMAD_BF16_RPRE=0 / 1)
TAG=${1:-rpre}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bf16.py -q -m gpu -x -k "192_column or epilogues" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -n 2 $OUT/pytest.log
timeout 300 python3 tools/lib_gemm_yardstick.py --f32 "" --bf16 c5_out,c5_fc2,c5h_out,c5h_fc2 > $OUT/yardstick.jsonl 2> $OUT/err.log
python3 -c "
import json
for l in open('$OUT/yardstick.jsonl'):
    d=json.loads(l); print(d['shape'], 'vendor', round(d['lib_ms']*1e3,1), 'tile16', round(d['ours_ms']*1e3,1), 'rpre', round(d['rpre_ms']*1e3,1), 'n192', round(d.get('n192_ms',0)*1e3,1))"
timeout 200 python3 tools/sustained_gemm.py c5_fc2,c5_out 16,57 2>>$OUT/err.log | tee $OUT/sustained.jsonl
for rep in 1 2; do for m in 0 1; do
  export NOMAD_BF16_RPRE=$m
  timeout 300 python3 bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/c5_$m.$rep.json 2> $OUT/c5_$m.$rep.err; echo "c5 rpre=$m rep $rep exit $?"
  python3 -c "import json,sys; d=json.loads(open('$OUT/c5_$m.$rep.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('kernel_time_ms_per_step'))"
done; done
