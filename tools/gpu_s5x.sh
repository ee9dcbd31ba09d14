#!/bin/bash
# round 5, trip x: attention kernels without the compiler's vmcnt(0) in front of their LDS reads (fp32: ext-vector loads; bf16: asm
# transposing reads) - tests, the kernels alone, C5 bench A/B (NOMAD_BF16_ATTN_V3 = 3: builtin reads), the default bench
TAG=${1:-s5x}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_precision_vs_oracle.py tests/test_gpu_race_screen.py -q -m gpu -x --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 3 $OUT/pytest.log
for v in 3 2 3 2; do NOMAD_BF16_ATTN_V3=$v timeout 300 python3 tools/attn_bf16_ab.py >> $OUT/attn_ab.jsonl 2>> $OUT/attn_ab.err; done
cat $OUT/attn_ab.jsonl
for i in 1 2; do timeout 300 python3 tools/attn_f32_time.py >> $OUT/attn_f32.jsonl 2>> $OUT/attn_f32.err; done
timeout 300 python3 tools/attn_f32_time.py --B 32 --T 1499 >> $OUT/attn_f32.jsonl 2>> $OUT/attn_f32.err
cat $OUT/attn_f32.jsonl
for rep in 1 2; do for v in 3 2; do
  NOMAD_DIAG_LIB=1 NOMAD_BF16_ATTN_V3=$v timeout 600 python bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/bench_c5_$v_$rep.json 2> $OUT/bench_c5_$v_$rep.err
  echo "ATTN_V3=$v rep $rep: $(python3 -c "import json,sys; d=json.load(open('$OUT/bench_c5_$v_$rep.json')); print(d['value'], d['ms_per_step'])")" | tee -a $OUT/summary.txt
done; done
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt
python3 -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['roofline'])"
