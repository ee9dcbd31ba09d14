#!/bin/bash
# bf16x3 GEMM: one instantiation (run-time output format, small epilogue): tests, then bench --dtype bf16x3 A/B (NOMAD_X3_PLAIN_EPI=0 / 1)
TAG=${1:-x3plain}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_precision_vs_oracle.py -q -m gpu -x > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -n 2 $OUT/pytest.log
for m in 0 1; do
NOMAD_X3_PLAIN_EPI=$m python3 - <<'PY' > $OUT/emb_$m.txt 2>>$OUT/err.log
import torch, hashlib
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
eng = Engine(seeded_state_dict(0), 0)
g = torch.Generator().manual_seed(3)
for B, N in ((256, 64000), (5, 70000), (33, 160000)):
    wav = (0.1 * torch.randn(B, N, generator=g)).clamp(-1, 1).cuda()
    e = eng.embed_bf16x3(wav).cpu()
    print(B, N, hashlib.sha1(e.numpy().tobytes()).hexdigest())
PY
done
cmp $OUT/emb_0.txt $OUT/emb_1.txt && echo "bf16x3 embeddings bit-identical in both modes"; cat $OUT/emb_1.txt
for rep in 1 2; do for m in 0 1; do
  export NOMAD_X3_PLAIN_EPI=$m
  timeout 300 python3 bench.py --dtype bf16x3 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/x3_$m.$rep.json 2> $OUT/x3_$m.$rep.err; echo -n "x3 plain_epi=$m rep $rep exit $?  "
  python3 -c "import json,sys; d=json.loads(open('$OUT/x3_$m.$rep.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('kernel_time_ms_per_step'))"
done; done
