#!/bin/bash
# round-2 GPU trip: bf16x3 layer outputs (tests + C4 / larger loss batches with the clean branch on the split-bf16 forward)
out=gpurun_out/r02c; mkdir -p $out
python -m pytest tests/test_gpu_bf16x3.py -q -x -k "layer_outputs or forward_loss or vs_fp32_path" -s 2>&1 | grep -E "bf16x3|forward\(\)|passed|failed|Error|error" | tail -20 > $out/tests.txt
for prec in fp32 bf16x3; do
  python tools/bench_c4.py --precision $prec 2>/dev/null | tail -1 >> $out/c4.txt
  python tools/bench_c4.py --precision $prec --batch 32 --samples 64000 --steps 10 2>/dev/null | tail -1 >> $out/c4.txt
  python tools/bench_c4.py --precision $prec --batch 16 --samples 32000 --steps 10 2>/dev/null | tail -1 >> $out/c4.txt
done
cat $out/tests.txt $out/c4.txt
