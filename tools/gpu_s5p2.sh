#!/bin/bash
# round 5, trip p2: the bf16 pos-conv with its input slab resident in LDS (posconv_bf16_slab.hip.h) - bf16 tests, configs[4] bench
# A/B against the grouped GEMM it replaces (NOMAD_BF16_POSCONV_SLAB=0, diag library), kernel stats of configs[4]
TAG=${1:-s5p2}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_precision_vs_oracle.py tests/test_gpu_race_screen.py -q -m gpu -x --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 15 $OUT/pytest.log
for rep in 1 2 3; do for v in 0 1; do
  NOMAD_DIAG_LIB=1 NOMAD_BF16_POSCONV_SLAB=$v timeout 600 python bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/bench_c5_${v}_$rep.json 2> $OUT/bench_c5_${v}_$rep.err
  echo "slab=$v rep $rep: $(python3 -c "import json,sys; d=json.load(open('$OUT/bench_c5_${v}_$rep.json')); print(d['value'], d['ms_per_step'], d.get('bf16_max_abs_score_diff_vs_f32'))")" | tee -a $OUT/summary.txt
done; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -o c5 -- python3 $ROOTDIR/bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 5 --warmup 2 --no-cpu-baseline --no-profile > $OUT/prof_bench_c5.json 2> $OUT/prof_c5.err); echo "rocprof c5 exit $?" | tee -a $OUT/summary.txt
f=$(find $OUT/prof_c5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/c5_kernel_stats.csv && head -8 $OUT/c5_kernel_stats.csv | cut -c1-160
rm -rf $OUT/prof_c5
