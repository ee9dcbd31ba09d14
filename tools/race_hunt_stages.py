#!/usr/bin/env python3
"""Race hunt, second stage: WHICH kernel of the bf16 forward produces the run-to-run difference that two co-scheduled bf16
forwards show ("bf16 nondeterminism")?  Same loop as tools/race_hunt_bf16.py - fp32 embeds in between, matmuls on
another stream, prior activity in the process (HUNT_PARTS) - but the two halves are launched here, and with HUNT_CKSUM=1
(libnomad_diag.so) every stage of both forwards leaves a per-clip checksum (nomad_diag_set_cksum); on a mismatch the FIRST
stage whose checksum differs from the reference call's names the kernel.
Usage: [HUNT_PARTS=a,b,c,d] [HUNT_CKSUM=1] [HUNT_DIAG=1] [HUNT_SHAPE=64x64000] python tools/race_hunt_stages.py [rounds]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 80
cksum = os.environ.get("HUNT_CKSUM", "0") == "1"
diag = cksum or os.environ.get("HUNT_DIAG", "0") == "1"
Bc, Nc = (int(v) for v in os.environ.get("HUNT_SHAPE", "64x64000").split("x"))
order = os.environ.get("HUNT_ORDER", "side_first")   # which half is launched first
eng = Engine(seeded_state_dict(0), 0, diag=diag)
lib = eng.lib
gen = torch.Generator().manual_seed(33)

STAGES = (["gn_sums", "gn_scale", "gn_shift"] + [f"conv{i}" for i in range(7)] + ["feature_ln", "proj(xpad)", "posconv+res", "encoder_ln"]
          + [f"L{l}.{n}" for l in range(12) for n in ("qkv", "attn", "out_proj+res", "ln1", "fc1", "fc2+res", "ln2")] + ["emb"])
KERNEL = {"gn_sums": "wav_stats/fold", "gn_scale": "gn_fold", "gn_shift": "gn_fold", "conv0": "conv0_gn_gelu<bf16>", "feature_ln": "layernorm<2,bf16>",
          "proj(xpad)": "gemm_bf16 (+zero_pad_rows)", "posconv+res": "gemm_bf16<128,64> grouped", "encoder_ln": "layernorm<3,bf16>",
          "qkv": "gemm_bf16", "attn": "attention_bf16_v2", "out_proj+res": "gemm_bf16", "ln1": "layernorm<3,bf16>", "fc1": "gemm_bf16",
          "fc2+res": "gemm_bf16", "ln2": "layernorm<3,bf16>", "emb": "head_pool/head"}
NST = len(STAGES)
snap = cksum and os.environ.get("HUNT_SNAP", "0") == "1"      # keep copies of scale / shift / conv0 output of both halves
if cksum:
    lib.nomad_diag_set_cksum.restype = C.c_int
    lib.nomad_diag_set_cksum.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.nomad_diag_set_snapshot.restype = C.c_int
    lib.nomad_diag_set_snapshot.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
SNAP_STAGES = (1, 2, 3)

parts = set(filter(None, os.environ.get("HUNT_PARTS", "").split(",")))
if parts:
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

wav = (0.1 * torch.randn(Bc, Nc, generator=gen)).clamp(-1, 1).cuda()
h = Bc // 2
SEGS = 16 * (Bc - h)
lib.nomad_enable_bf16(eng.ctx)


def bf16_split(w):
    """Engine.embed_bf16 with the two-stream split on; -> (emb, [table of the caller's-stream half, table of the side half])."""
    emb = torch.empty(Bc, 256, dtype=torch.float32, device="cuda")
    tabs = [torch.zeros(NST, SEGS, dtype=torch.int64, device="cuda") for _ in range(2)] if cksum else None
    cur, st = torch.cuda.current_stream(), eng.side_stream(1)
    L0 = (Nc - 10) // 5 + 1
    snaps = [[torch.empty(n, dtype=torch.uint8, device="cuda") for n in (2048 * (Bc - h), 2048 * (Bc - h), 1024 * L0 * (Bc - h))]
             for _ in range(2)] if snap else None

    def arm(k):
        if cksum:
            lib.nomad_diag_set_cksum(eng.ctx, tabs[k].data_ptr(), NST, SEGS)
        if snap:
            for slot, stg in enumerate(SNAP_STAGES):
                lib.nomad_diag_set_snapshot(eng.ctx, slot, stg, snaps[k][slot].data_ptr(), snaps[k][slot].numel())

    def side_half():
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            arm(1)
            eng._embed_bf16_into(w[h:], emb[h:], side=1)

    def main_half():
        arm(0)
        eng._embed_bf16_into(w[:h], emb[:h], side=False)

    if order == "side_first":
        side_half(); main_half()
    else:
        st.wait_stream(cur)
        main_half(); side_half()
    cur.wait_stream(st)
    return emb, tabs, snaps


def describe_snapshots(k, snaps):
    """How the kept copies of half k differ from the reference call's: scale, shift (fp32 [B][512]) and conv0's output."""
    L0 = (Nc - 10) // 5 + 1
    for slot, name in enumerate(("scale", "shift")):
        a, b = snaps[k][slot].view(torch.float32), refsnaps[k][slot].view(torch.float32)
        n = int((a.view(torch.int32) != b.view(torch.int32)).sum())
        print(f"        {name} copy: {n} of {a.numel()} floats differ" + (f", max|diff| {float((a - b).abs().max()):.3e}" if n else ""), flush=True)
    o = snaps[k][2].view(torch.bfloat16).view(-1, L0, 512)
    r = refsnaps[k][2].view(torch.bfloat16).view(-1, L0, 512)
    d = o.view(torch.int16) != r.view(torch.int16)
    n = int(d.sum())
    if not n:
        print("        conv0 output copy: identical to the reference copy (the checksum kernel read something else)", flush=True)
        return
    clips = torch.nonzero(d.flatten(1).any(dim=1)).flatten().tolist()
    frames = d.any(dim=2)
    b0 = clips[0]
    fr = torch.nonzero(frames[b0]).flatten().tolist()
    ch = torch.nonzero(d[b0, fr[0]]).flatten().tolist()
    diff = (o.float() - r.float()).abs()
    rel = (diff / r.float().abs().clamp_min(1e-3))[d]
    print(f"        conv0 output copy: {n} elements differ, clips {clips[:8]}{'...' if len(clips) > 8 else ''} ({len(clips)}), differing frames per clip "
          f"{[int(frames[c].sum()) for c in clips[:8]]} of {L0}; clip {b0}: frames {fr[:10]}{'...' if len(fr) > 10 else ''}; frame {fr[0]}: {len(ch)} channels "
          f"{ch[:12]}{'...' if len(ch) > 12 else ''}; max|diff| {float(diff.max()):.3e}; relative diff median {float(rel.median()):.2e} max {float(rel.max()):.2e}", flush=True)
    chan_hist = d.sum(dim=(0, 1))
    top = torch.topk(chan_hist, 8)
    print(f"        channels with most differences: {top.indices.tolist()} counts {top.values.tolist()}; channels touched {int((chan_hist > 0).sum())} of 512", flush=True)


ref = eng.embed(wav).clone()
ref16, reftabs, refsnaps = bf16_split(wav)
ref16 = ref16.clone()
torch.cuda.synchronize()
side = torch.cuda.Stream()
junk = torch.randn(4096, 4096, device="cuda")
bad32 = bad16 = calls16 = 0
first_stage_hist = {}
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
            o16, tabs, snaps = bf16_split(wav)
            differs = not torch.equal(o16, ref16)
            tdiff = cksum and any(not torch.equal(tabs[k], reftabs[k]) for k in range(2))
            if differs or tdiff:
                bad16 += 1
                rows = torch.nonzero((o16 != ref16).any(dim=1)).flatten().tolist()
                print(f"round {rd} it {it}: emb differs in clips {rows[:8]}{'...' if len(rows) > 8 else ''} ({len(rows)} of {Bc}), "
                      f"max|diff| {(o16 - ref16).abs().max().item():.3e}", flush=True)
                if cksum:
                    for k, name in ((0, "caller's-stream half"), (1, "side-stream half")):
                        d = (tabs[k] != reftabs[k])
                        st_bad = torch.nonzero(d.any(dim=1)).flatten().tolist()
                        if not st_bad:
                            continue
                        s0 = st_bad[0]
                        segs = torch.nonzero(d[s0]).flatten().tolist()
                        stage = STAGES[s0]
                        kern = KERNEL.get(stage, KERNEL.get(stage.split(".")[-1], "gemm_bf16"))
                        first_stage_hist[stage] = first_stage_hist.get(stage, 0) + 1
                        print(f"    {name}: FIRST differing stage {s0} = {stage} [{kern}], segments {segs[:12]}{'...' if len(segs) > 12 else ''} "
                              f"({len(segs)}); {len(st_bad)} of {NST} stages differ; next: {[STAGES[i] for i in st_bad[1:4]]}", flush=True)
                        if snap:
                            describe_snapshots(k, snaps)
torch.cuda.synchronize()
print(f"lib={'diag' if diag else 'product'} cksum={int(cksum)} shape={Bc}x{Nc} order={order} parts={sorted(parts)}: "
      f"fp32 mismatches {bad32}/{rounds * 12}, bf16 mismatches {bad16}/{calls16}; first-stage histogram {first_stage_hist}", flush=True)
