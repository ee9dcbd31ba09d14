#!/usr/bin/env python3
"""Mixed-length scoring (the real `predict` workload: every file has its own length): one ragged batch vs the
reference-style per-file loop, same engine, same clips."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from nomad_amd.engine import Engine  # noqa: E402
from nomad_amd.weights import seeded_state_dict  # noqa: E402

eng = Engine(seeded_state_dict(0), 0)
g = torch.Generator().manual_seed(0)
n_clips = 256
lens = torch.randint(16000, 8 * 16000, (n_clips,), generator=g).tolist()      # 1 s .. 8 s
waves = [(0.1 * torch.randn(n, generator=g)).clamp(-1, 1).cuda() for n in lens]


def ragged():
    return eng.embed_ragged(waves)


def per_file():
    return torch.cat([eng.embed(w[None, :]) for w in waves])


def ragged_x3():
    return eng.embed_ragged(waves, precision="bf16x3")


out = {"workload": f"{n_clips} clips, uniform random lengths 1-8 s ({sum(lens) / 16000:.0f} s of audio)"}
ref = None
for name, fn in (("ragged_batch", ragged), ("per_file_loop", per_file)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        r = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    out[name] = {"clips_per_s": round(n_clips / dt, 1), "audio_s_per_s": round(sum(lens) / 16000 / dt, 1)}
    if ref is None:
        ref = r
    else:
        out["bit_identical"] = bool(torch.equal(ref, r))
fn = ragged_x3
r = fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    r = fn()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
out["ragged_batch_bf16x3"] = {"clips_per_s": round(n_clips / dt, 1), "audio_s_per_s": round(sum(lens) / 16000 / dt, 1),
                              "max_abs_err_vs_fp32": float((r - ref).abs().max())}
print(json.dumps(out))

# long recordings of mixed lengths (config C5 through predict): fp32 vs bf16 ragged batches
n_long = 48
lens = torch.randint(10 * 16000, 40 * 16000, (n_long,), generator=g).tolist()   # 10 s .. 40 s
waves = [(0.1 * torch.randn(n, generator=g)).clamp(-1, 1).cuda() for n in lens]
out = {"workload": f"{n_long} recordings, uniform random lengths 10-40 s ({sum(lens) / 16000:.0f} s of audio)"}
for name, kw in (("ragged_fp32", {}), ("ragged_bf16x3", {"precision": "bf16x3"}), ("ragged_bf16", {"bf16": True})):
    eng.embed_ragged(waves, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        r = eng.embed_ragged(waves, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    out[name] = {"clips_per_s": round(n_long / dt, 1), "audio_s_per_s": round(sum(lens) / 16000 / dt, 1)}
print(json.dumps(out))
