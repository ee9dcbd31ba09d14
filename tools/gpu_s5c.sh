#!/bin/bash
# round 5, trip c: persistent bf16 GEMM with the interleaved epilogue (tile 60) vs epilogue between tiles (63) vs shipped-until-now (58)
TAG=${1:-s5c}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python3 tools/p9_ab.py --tiles 58,63,60 --shapes c5_qkv,c5_out,c5_fc1,c5_fc2,c5_conv4,c5_conv2,c5_fc1_nogelu,c5h_out,c5h_fc2,c5_k128,c5_k256,one_tile,eight_tiles > $OUT/p9_ab.jsonl 2> $OUT/p9_ab.err; echo "p9_ab exit $?" | tee -a $OUT/summary.txt
cat $OUT/p9_ab.jsonl; tail -3 $OUT/p9_ab.err
timeout 600 python3 tools/p9_timeline.py > $OUT/p9_timeline.jsonl 2> $OUT/p9_timeline.err; echo "p9_timeline exit $?" | tee -a $OUT/summary.txt
cat $OUT/p9_timeline.jsonl; tail -3 $OUT/p9_timeline.err
timeout 1200 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_precision_vs_oracle.py -q -m gpu -x --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 5 $OUT/pytest.log
for rep in 1 2; do for v in 0 1; do
  NOMAD_DIAG_LIB=1 NOMAD_BF16_P9=$v timeout 600 python bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/bench_c5_p9_${v}_$rep.json 2> $OUT/bench_c5_${v}_$rep.err
  echo "P9=$v rep $rep: $(python3 -c "import json,sys; d=json.load(open('$OUT/bench_c5_p9_${v}_$rep.json')); print(d['value'], d['ms_per_step'])")" | tee -a $OUT/summary.txt
done; done
