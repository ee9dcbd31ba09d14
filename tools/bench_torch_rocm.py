"""What the stock PyTorch-ROCm stack does with this workload on the same MI355X.

Not part of the product and not the oracle: it times HuggingFace ``transformers``' ``Wav2Vec2Model``
(the same wav2vec 2.0 BASE architecture fairseq gives the reference, /root/reference/src/nomad_audio/nomad.py:214-231),
random-initialised, fp32, ``torch.no_grad()``, on ``cuda:0`` through rocBLAS/hipBLASLt/MIOpen, followed by the
reference's head (mean_t -> ReLU -> Linear(768,256) -> L2 normalise) and ``torch.cdist`` in float64 - the op
sequence a reference user runs when they call ``Nomad(device='cuda').predict`` on this box, minus file IO.

Prints one JSON object per batch size: clips/s for 4 s clips.  ``--sdpa`` also times the "sdpa" attention
implementation (fairseq's own MultiheadAttention is the eager bmm+softmax kind, which is the default here).
"""
import argparse
import json
import time

import torch


def build(attn):
    from transformers import Wav2Vec2Config, Wav2Vec2Model
    cfg = Wav2Vec2Config()  # defaults == wav2vec2-base
    cfg.apply_spec_augment = False
    cfg._attn_implementation = attn
    torch.manual_seed(0)
    m = Wav2Vec2Model(cfg).eval().cuda()
    head = torch.nn.Linear(768, 256).cuda()
    return m, head


@torch.no_grad()
def embed(m, head, wav):
    x = m(wav).last_hidden_state
    x = torch.relu(x.mean(1))
    return torch.nn.functional.normalize(head(x), dim=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,8,32,64,128,256")
    ap.add_argument("--clips", type=int, default=256, help="clips per timed pass (the bench.py step size)")
    ap.add_argument("--refs", type=int, default=32)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--sdpa", action="store_true")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    torch.backends.cuda.matmul.allow_tf32 = False
    g = torch.Generator().manual_seed(0)
    wav = (0.1 * torch.randn(a.clips, 64000, generator=g)).cuda()
    res = []
    for attn in (["eager", "sdpa"] if a.sdpa else ["eager"]):
        m, head = build(attn)
        for B in (int(b) for b in a.batches.split(",")):
            try:
                def step():
                    emb = torch.cat([embed(m, head, wav[i:i + B]) for i in range(0, a.clips, B)])
                    d = torch.cdist(emb[a.refs:].double(), emb[:a.refs].double())
                    return d.mean(1)
                step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    step()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / a.steps
                row = {"impl": "transformers Wav2Vec2Model fp32, torch " + torch.__version__, "attention": attn,
                       "forward_batch": B, "clips_per_pass": a.clips, "ms_per_pass": round(dt * 1e3, 2),
                       "clips_per_s": round(a.clips / dt, 1),
                       "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
            except torch.OutOfMemoryError:
                row = {"attention": attn, "forward_batch": B, "error": "out of memory"}
                torch.cuda.empty_cache()
            res.append(row)
            print(json.dumps(row), flush=True)
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
