#!/usr/bin/env python3
"""Round 6: how far is the bf16 path from the fp32 path, embedding by embedding?  Three batches (4 s clips x 64, 1 s clips x 64, 30 s clips x 8) on the
seeded and on the 'peaky' weights: max |difference| and minimum cosine of the 256-d unit embeddings, and of the NOMAD scores against 16 references.
Used to compare builds of the bf16-output GELU (NOMAD_LIB_VARIANT=gelu1 / gelu2 / main).
Usage: [NOMAD_LIB_VARIANT=..] python tools/bf16_accuracy.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from nomad_amd.engine import Engine  # noqa: E402
from nomad_amd.weights import seeded_state_dict  # noqa: E402


def main():
    rows = {"variant": os.environ.get("NOMAD_LIB_VARIANT", "main")}
    for wname, kw in (("seeded", {}), ("peaky", dict(seed=1, qk_gain=6.0))):
        eng = Engine(seeded_state_dict(**kw) if kw else seeded_state_dict(0), 0)
        for bname, B, n in (("4s", 64, 64000), ("1s", 64, 16384), ("30s", 8, 480000)):
            g = torch.Generator().manual_seed(B + n)
            wav = (0.1 * torch.randn(B, n, generator=g)).clamp(-1, 1).cuda()
            e32 = eng.embed(wav)
            e16 = eng.embed_bf16(wav)
            nref = min(16, B // 2)
            s32 = eng.pairwise(e32[nref:], e32[:nref])
            s16 = eng.pairwise(e16[nref:], e16[:nref])
            s32 = s32[0] if isinstance(s32, (tuple, list)) else s32
            s16 = s16[0] if isinstance(s16, (tuple, list)) else s16
            rows[f"{wname}_{bname}"] = {"emb_max_abs": float((e32 - e16).abs().max()),
                                        "emb_rms": float((e32 - e16).pow(2).mean().sqrt()),
                                        "emb_min_cos": float(torch.nn.functional.cosine_similarity(e32, e16, dim=1).min()),
                                        "score_max_abs": float((s32.double() - s16.double()).abs().max())}
        eng.close()
    print(json.dumps(rows), flush=True)


if __name__ == "__main__":
    main()
