#!/usr/bin/env python3
"""Race hunt: the loop of tests/test_gpu_parity.py::test_repeat_runs_are_bit_identical, which failed twice in ~40 runs of the
suite in round 2 (embed_bf16 of 64 clips of 4 s differing from its first result) - fp32 embeds in between, matmuls on another
stream - repeated many times; on a mismatch: which clips differ and by how much.
Usage: python tools/race_hunt_bf16.py [rounds] [bf16_split_rows] [f32_split_rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
eng = Engine(seeded_state_dict(0), 0)
if len(sys.argv) > 2:
    eng.BF16_SPLIT_ROWS = int(sys.argv[2])
if len(sys.argv) > 3:
    eng.F32_SPLIT_ROWS = int(sys.argv[3])
gen = torch.Generator().manual_seed(33)
if os.environ.get("HUNT_BIG_FIRST") == "1":   # what the suite does before the screen: a 512-clip batch (a main workspace of tens of GB),
    big = (0.1 * torch.randn(512, 64000, generator=gen)).clamp(-1, 1).cuda()   # long clips, ragged batches, many streams
    eng.embed(big)
    eng.embed_bf16(big[:64])
    eng.embed_bf16x3(big[:64])
    del big
    long_ = (0.1 * torch.randn(2, 480000, generator=gen)).clamp(-1, 1).cuda()
    eng.embed(long_); eng.embed_bf16(long_)
    eng.embed_ragged([long_[0, :50000], long_[1, :123456], long_[0, :8000]], precision="bf16")
    del long_
    _streams = [torch.cuda.Stream() for _ in range(40)]
    torch.cuda.synchronize()
parts = set(filter(None, os.environ.get("HUNT_PARTS", "").split(",")))  # what else of the parity file's life before the screen:
if parts:                                                                 # a = second engine, b = predict (threads), c = autograd, d = degenerate inputs
    import tempfile
    from nomad_amd.nomad import Nomad
    if "a" in parts:
        eng2 = Engine(seeded_state_dict(1, qk_gain=6.0), 0)
        eng2.embed((0.1 * torch.randn(4, 30000, generator=gen)).clamp(-1, 1).cuda())
    if "b" in parts or "c" in parts:
        nmd = Nomad(weights=seeded_state_dict(0))
    if "b" in parts:
        wavs = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "wavs")
        with tempfile.TemporaryDirectory() as d:
            nmd.predict("dir", os.path.join(wavs, "nmr-data"), os.path.join(wavs, "test-data"), results_path=d)
    if "c" in parts:
        est = (0.1 * torch.randn(2, 1, 16384, generator=gen)).cuda().requires_grad_(True)
        nmd.forward(est, (0.1 * torch.randn(2, 1, 16384, generator=gen)).cuda()).backward()
    if "d" in parts:
        for kind in (torch.zeros(2, 16000), torch.full((2, 16000), 0.5), torch.ones(2, 16000)):
            eng.embed(kind.cuda())
    torch.cuda.synchronize()
wav = (0.1 * torch.randn(64, 64000, generator=gen)).clamp(-1, 1).cuda()
ref = eng.embed(wav).clone()
ref16 = eng.embed_bf16(wav).clone()
side = torch.cuda.Stream()
junk = torch.randn(4096, 4096, device="cuda")
bad32 = bad16 = calls16 = 0
for rd in range(rounds):
    for it in range(12):
        with torch.cuda.stream(side):
            for _ in range(4):
                junk = junk @ junk * 1e-3
        out = eng.embed(wav)
        if not torch.equal(out, ref):
            bad32 += 1
        if it % 4 == 0:
            calls16 += 1
            o16 = eng.embed_bf16(wav)
            if not torch.equal(o16, ref16):
                bad16 += 1
                rows = torch.nonzero((o16 != ref16).any(dim=1)).flatten().tolist()
                print(f"round {rd} it {it}: bf16 result differs in clips {rows[:16]}{'...' if len(rows) > 16 else ''} ({len(rows)} of 64), "
                      f"max|diff| {(o16 - ref16).abs().max().item():.3e}, finite {bool(torch.isfinite(o16).all())}", flush=True)
torch.cuda.synchronize()
print(f"bf16_split_rows={eng.BF16_SPLIT_ROWS} f32_split_rows={eng.F32_SPLIT_ROWS}: fp32 mismatches {bad32}/{rounds * 12}, bf16 mismatches {bad16}/{calls16}", flush=True)
