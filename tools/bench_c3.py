#!/usr/bin/env python3
"""Config C3 (BASELINE.json configs[2]): 10 000 degraded x 1 000 non-matching references, clips sharded across
the GPUs of one node, ONE all-gather of the reference embeddings, then each rank's distance slab + row means.

    python tools/bench_c3.py                                         # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_c3.py

Strong scaling: the total work is fixed.  Rank r embeds deg[partition(10000, W, r)] and ref[partition(1000, W, r)]
in batches of 256 synthetic 4 s clips (generated on the GPU per batch), keeps only the (n,256) embeddings."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from nomad_amd.dist import all_gather_rows, partition  # noqa: E402
from nomad_amd.engine import Engine  # noqa: E402
from nomad_amd.weights import seeded_state_dict  # noqa: E402

N_DEG, N_REF, N_SAMPLES, BATCH = int(os.environ.get("C3_DEG", 10000)), int(os.environ.get("C3_REF", 1000)), 64000, 256
PRECISION = os.environ.get("C3_PRECISION", "fp32")   # fp32 | bf16x3 | bf16


def embed_range(eng, start, stop, seed_base):
    out = []
    for s in range(start, stop, BATCH):
        n = min(BATCH, stop - s)
        g = torch.Generator(device="cuda").manual_seed(seed_base + s)       # clip content depends on the global index only
        wav = (0.1 * torch.randn(n, N_SAMPLES, generator=g, device="cuda")).clamp(-1, 1)
        out.append({"fp32": eng.embed, "bf16x3": eng.embed_bf16x3, "bf16": eng.embed_bf16}[PRECISION](wav))
    return torch.cat(out) if out else torch.empty(0, 256, device="cuda")


def main():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    eng = Engine(seeded_state_dict(0), local)
    embed_range(eng, 0, BATCH, 3_000_000)                                    # warm-up (workspace, weight copies, clocks)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    ds, de = partition(N_DEG, world, rank)
    rs, re_ = partition(N_REF, world, rank)
    deg_emb = embed_range(eng, ds, de, 1_000_000)
    ref_emb = embed_range(eng, rs, re_, 2_000_000)
    ref_all = all_gather_rows(ref_emb)                                       # the one data-path collective
    dist_slab, mean = eng.pairwise(deg_emb, ref_all, want_matrix=True)
    scores = all_gather_rows(mean)                                           # output only: 10 000 float64
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        assert scores.shape == (N_DEG,) and torch.isfinite(scores).all() and ref_all.shape == (N_REF, 256)
        print(json.dumps({"config": f"C3: {N_DEG} deg x {N_REF} ref, 16 kHz x 4 s, {PRECISION}, clip-sharded x{world}",
                          "n_gpus": world, "seconds": round(dt, 3), "clips_per_s": round((N_DEG + N_REF) / dt, 1),
                          "pairs": N_DEG * N_REF, "scaling": "strong",
                          "score_mean": float(scores.mean()), "score_min": float(scores.min())}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
