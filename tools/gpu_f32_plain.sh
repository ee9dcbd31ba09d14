#!/bin/bash
# fp32 GEMM with the plain / residual-prefetch epilogue: kernel + parity tests, then headline A/B (NOMAD_F32_PLAIN_EPI=0 / 1), alternating
TAG=${1:-f32plain}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -x > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -n 2 $OUT/pytest.log
for rep in 1 2; do for m in 0 1; do
  export NOMAD_F32_PLAIN_EPI=$m
  timeout 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-also --live-traffic off > $OUT/c2_$m.$rep.json 2> $OUT/c2_$m.$rep.err; echo -n "c2 plain_epi=$m rep $rep exit $?  "
  python3 -c "import json,sys; d=json.loads(open('$OUT/c2_$m.$rep.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_gemm_launches']['achieved'], d.get('kernel_time_ms_per_step'))"
done; done
