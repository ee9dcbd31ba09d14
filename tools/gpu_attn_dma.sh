#!/bin/bash
# K / V staging of the bf16 attention by LDS-DMA: bit-equality of the embeddings, then C5 A/B (NOMAD_BF16_ATTN_DMA=0 / 1 / 2), alternating
TAG=${1:-attndma}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for m in 0 1 2; do
NOMAD_BF16_ATTN_DMA=$m python3 - <<'PY' > $OUT/emb_$m.txt 2>>$OUT/err.log
import torch, hashlib
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
eng = Engine(seeded_state_dict(0), 0)
g = torch.Generator().manual_seed(3)
for B, N in ((32, 480000), (5, 70000), (40, 64000)):
    wav = (0.1 * torch.randn(B, N, generator=g)).clamp(-1, 1).cuda()
    e = eng.embed_bf16(wav).cpu()
    print(B, N, hashlib.sha1(e.numpy().tobytes()).hexdigest())
PY
done
cmp $OUT/emb_0.txt $OUT/emb_1.txt && cmp $OUT/emb_0.txt $OUT/emb_2.txt && echo "embeddings bit-identical in all three modes"; cat $OUT/emb_0.txt
for rep in 1 2; do for m in 0 1 2; do
  export NOMAD_BF16_ATTN_DMA=$m
  timeout 300 python3 bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/c5_$m.$rep.json 2> $OUT/c5_$m.$rep.err; echo "c5 attn_dma=$m rep $rep exit $?"
  python3 -c "import json,sys; d=json.loads(open('$OUT/c5_$m.$rep.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('kernel_time_ms_per_step'))"
done; done
