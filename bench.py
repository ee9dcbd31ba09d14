#!/usr/bin/env python3
"""NOMAD scoring throughput on MI355X: clips/s embedded + N x M NOMAD distances.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N --steps K --warmup W        # N > 1 from a plain shell: spawns the N ranks itself (below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Started WITHOUT the launcher's environment (no RANK) and with --gpus N > 1, this process touches no GPU: it starts
`python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a child (one rank per GPU, RCCL),
relays rank 0's single JSON line and exits with the child's status.

One "step" (per rank) = one pass of the hot path over one batch of synthetic input (BASELINE.json
configs[1]): 256 clips of 16 kHz x 4 s (224 degraded + 32 non-matching references) through the
wav2vec 2.0 BASE backbone + 768->256 head in fp32, one all-gather (RCCL) of the reference
embeddings, then this rank's (224 x 32*N) float64 distance slab + row means (the NOMAD scores).
Per-GPU work is fixed as N grows (weak scaling); value = all ranks' clips / max-over-ranks time.
Inputs are resident in HBM before the timed region.  Weights: real nomad_best_model.pt when present,
otherwise a seeded random parameter set of the same architecture (no network for checkpoints).

Also on the JSON line:
  roofline      the dominant kernel (fp32 MFMA GEMM, all launches): algorithmic FLOPs / summed
                hipEvent durations of those launches inside the timed region, vs the 157.3 TF fp32 MFMA peak
  cpu_baseline  the CPU oracle (PyTorch fp32 restatement of the reference path, batch-1 loop like
                nomad.py:171-183 + float64 cdist) timed on this box's host cores over a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL otherwise fails with hipIpcGetMemHandle: invalid argument); the
# boxes export it already - this only covers a shell that lost it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

FLOP_PER_CLIP_4S = 56.925e9      # BASELINE.md section 2 (2*MAC, no recompute)
FLOP_LAYERS_4S = 35.264e9        # the 12 encoder layers ("attention-GEMM" subset)
PEAK_FP32_MFMA = 157.3e12        # MI355X_MICROARCH.md chip table


def cpu_baseline(sd, n_samples, budget_s=20.0, max_clips=256):
    """The CPU oracle on this box's host cores: batch-1 loop like nomad.py:171-183 + float64 cdist.
    The thread count is calibrated first (torch's default of one thread per core is pathological on
    many-core hosts at batch 1); `cores` reports the threads actually used."""
    import numpy as np
    import torch
    from oracle import nomad_oracle as O
    ncpu = os.cpu_count() or 1
    g = torch.Generator().manual_seed(0)
    wav = (0.1 * torch.randn(max_clips, n_samples, generator=g)).clamp(-1, 1)
    # what the reference itself would do: torch's DEFAULT thread count (it never calls set_num_threads), a few clips - reported next
    # to the calibrated figure so that a reader can tell "the reference's CPU path as shipped" from "the same path, threads tuned"
    default_threads = torch.get_num_threads()
    default_leg = None
    try:
        with torch.no_grad():
            O.triplet_forward(sd, wav[:1])
            t0, nd = time.perf_counter(), 0
            while nd < 8 and (nd < 2 or time.perf_counter() - t0 < budget_s / 5):
                O.triplet_forward(sd, wav[nd:nd + 1])
                nd += 1
            default_leg = {"value": round(nd / (time.perf_counter() - t0), 3), "unit": "clips/s", "cores": default_threads, "clips": nd,
                           "note": "torch's default thread count, as the reference would run"}
    except Exception as ex:  # noqa: BLE001
        default_leg = {"error": str(ex)[:120]}
    t_start = time.perf_counter()
    best_threads, best_dt = None, None
    with torch.no_grad():
        for c in [c for c in (8, 16, 32, 64) if c <= ncpu] or [ncpu]:
            torch.set_num_threads(c)
            O.triplet_forward(sd, wav[:1])  # warm-up at this thread count
            t0 = time.perf_counter()
            O.triplet_forward(sd, wav[:1])
            dt = time.perf_counter() - t0
            if best_dt is None or dt < best_dt:
                best_threads, best_dt = c, dt
            if time.perf_counter() - t_start > budget_s / 2:
                break
        torch.set_num_threads(best_threads)
        embs = []
        t0 = time.perf_counter()
        n = 0
        while n < max_clips and (n < 2 or time.perf_counter() - t_start < budget_s):
            embs.append(O.triplet_forward(sd, wav[n:n + 1])[0].numpy())
            n += 1
        e = np.stack(embs)
        half = max(1, n // 8)
        O.pairwise(e[half:], e[:half])
        dt = time.perf_counter() - t0
        # the batched leg SURVEY.md section 8(d) asks for: the same oracle on batches of 8 clips (what a user who batches the
        # reference's per-file loop by hand would get), bounded to a few seconds, threads re-calibrated for the larger GEMMs
        batched = None
        try:
            bt, bdt = None, None
            for c in [c for c in (16, 32, 64) if c <= ncpu] or [ncpu]:
                torch.set_num_threads(c)
                O.triplet_forward(sd, wav[:8])
                t1 = time.perf_counter()
                O.triplet_forward(sd, wav[:8])
                d1 = time.perf_counter() - t1
                if bdt is None or d1 < bdt:
                    bt, bdt = c, d1
            torch.set_num_threads(bt)
            t1, nb = time.perf_counter(), 0
            while nb + 8 <= max_clips and (nb < 16 or time.perf_counter() - t1 < budget_s / 3):
                O.triplet_forward(sd, wav[nb:nb + 8])
                nb += 8
            batched = {"value": round(nb / (time.perf_counter() - t1), 3), "unit": "clips/s", "batch": 8, "cores": bt, "clips": nb}
        except Exception as ex:  # noqa: BLE001
            batched = {"error": str(ex)[:120]}
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": round(n / dt, 3), "unit": "clips/s", "cores": best_threads, "kind": "port", "cpu_model": cpu_model,
            "batched_b8": batched, "torch_default_threads": default_leg,
            "sample": f"{n} clips of {n_samples} samples, batch-1 loop (as nomad.py:171-183) + float64 cdist "
                      f"{n - half}x{half}, {dt:.1f} s wall; host has {ncpu} logical CPUs, thread count calibrated "
                      f"over 8/16/32/64"}


PRECISION_NOTE = {"f32": "fp32", "bf16": "bf16 storage / fp32 accumulate",
                  "bf16x3": "bf16x3 (hi/lo-split operands, 3 bf16 MFMA products, fp32 accumulate; fp32 attention / norms)"}


def spawn_ranks(n: int, argv) -> int:
    """Parent of a multi-GPU run started from a plain shell: N fresh rank processes under torch.distributed.run (this
    process has not touched the GPU and never will - no exec of an initialised process), rank 0's stdout relayed."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + [a for a in argv if a != "--spawn"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    if res.returncode != 0:
        print(f"bench.py: the {n}-rank run failed (status {res.returncode}); does this box have {n} GPUs?", file=sys.stderr)
    return res.returncode if res.returncode != 0 or lines else 1


def c3_sharded_scores(embed_fn, pairwise_fn, wav, n_deg_local, batch, use_pg, stage_ms=None):
    """One rank's part of configs[2]: embed this rank's clips (its slice of the degraded set, then its slice of the references)
    in batches, ONE all-gather of the reference embeddings, this rank's distance slab + row means, one gather of the scores
    -> (all scores in global order, all reference embeddings, this rank's slab).  Shard sizes may differ between ranks.
    stage_ms (dict, optional): receives the device times of the all-gather, the distance stage and the score gather (events on the
    current stream; BASELINE.md section 4, row C3: "all-gather + distance time")."""
    import torch
    from nomad_amd.dist import all_gather_rows
    emb = torch.cat([embed_fn(wav[i:i + batch]) for i in range(0, wav.shape[0], batch)])
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if stage_ms is not None else None
    if ev:
        ev[0].record()
    ref_all = all_gather_rows(emb[n_deg_local:].contiguous(), force_collective=use_pg)     # the one data-path collective
    if ev:
        ev[1].record()
    d, mean = pairwise_fn(emb[:n_deg_local].contiguous(), ref_all, True)
    if ev:
        ev[2].record()
    scores = all_gather_rows(mean, force_collective=use_pg)
    if ev:
        ev[3].record()
        ev[3].synchronize()
        stage_ms.update(allgather_ms=ev[0].elapsed_time(ev[1]), pairwise_ms=ev[1].elapsed_time(ev[2]), scores_gather_ms=ev[2].elapsed_time(ev[3]))
    return scores, ref_all, d


def agree_or_raise(err, use_pg, what):
    """All ranks learn whether ANY rank failed a local set-up step, before the leg's first collective: a rank that failed alone
    must not leave the others waiting inside an all-gather.  Raises on every rank if one did."""
    if use_pg:
        import torch
        import torch.distributed as dist
        bad = torch.tensor([0 if err is None else 1], dtype=torch.int32, device="cuda")
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()) and err is None:
            err = RuntimeError(f"{what}: another rank could not set the leg up")
    if err is not None:
        raise err


def time_c3(eng, world, rank, use_pg, fence, n_deg=10000, n_ref=1000, batch=256):
    """BASELINE.json configs[2]: 10 000 degraded x 1 000 non-matching references (16 kHz x 4 s, fp32), clips sharded over the
    ranks, ONE all-gather of the reference embeddings, each rank's distance slab + row means, one gather of the scores.
    Strong scaling (the total is fixed); waveforms resident in HBM before the timed region."""
    import torch
    from nomad_amd.dist import partition
    (ds, de), (rs, re_) = partition(n_deg, world, rank), partition(n_ref, world, rank)
    g = torch.Generator(device="cuda").manual_seed(3000 + rank)
    # everything that can fail on ONE rank only (the 2.8 GB of waveforms, the first embed) happens before the ranks agree to go
    # on: a rank that failed here must not leave the others waiting inside the all-gather
    err = None
    try:
        wav = (0.1 * torch.randn(de - ds + re_ - rs, 64000, generator=g, device="cuda")).clamp(-1, 1)
        eng.embed(wav[:batch])
    except Exception as e:  # noqa: BLE001
        err = e
    if use_pg:
        import torch.distributed as dist
        bad = torch.tensor([0 if err is None else 1], dtype=torch.int32, device="cuda")
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()) and err is None:
            err = RuntimeError("configs[2]: another rank could not set the leg up")
    if err is not None:
        raise err
    fence()
    t0 = time.perf_counter()
    stage_ms = {}
    scores, ref_all, d = c3_sharded_scores(eng.embed, eng.pairwise, wav, de - ds, batch, use_pg, stage_ms)
    fence()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    st = torch.tensor([stage_ms.get("allgather_ms", 0.0), stage_ms.get("pairwise_ms", 0.0), stage_ms.get("scores_gather_ms", 0.0)], dtype=torch.float64, device="cuda")
    if use_pg:
        import torch.distributed as dist
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        dist.all_reduce(st, op=dist.ReduceOp.MAX)
    ok = bool(scores.shape[0] == n_deg and ref_all.shape[0] == n_ref and torch.isfinite(scores).all().item())
    dt = float(dt.item())
    return {"workload": f"configs[2]: {n_deg} deg x {n_ref} ref clips of 64000 samples, fp32, clip-sharded x{world}, one all-gather of "
                        f"the ref embeddings, {n_deg}x{n_ref} float64 distances + means, end to end",
            "dtype": "f32", "value": round((n_deg + n_ref) / dt, 2), "unit": "clips/s", "seconds": round(dt, 3), "pairs": n_deg * n_ref,
            "allgather_ms": round(float(st[0].item()), 3), "pairwise_ms": round(float(st[1].item()), 3),
            "scores_gather_ms": round(float(st[2].item()), 3),
            "stage_note": "device time (events on the launch stream, max over ranks) of the all-gather of the reference embeddings, of this "
                          "rank's distance slab + row means, and of the gather of the scores; the rest of `seconds` is the embedding of the clips",
            "scaling": "strong", "finite": ok}


def rank_census(world, rank, local_rank, use_pg, my_elapsed, steps, ref_rows_seen, ref_rows_local):
    """What proves, from the ONE line rank 0 prints, that the collective really spanned `world` ranks on `world` different GPUs
    (VERDICT r5 item 5): an all-reduce of ones, every rank's device (name, PCI bus, uuid where the runtime has one) and hostname,
    the min / max over ranks of the timed loop, and the row count of the all-gathered reference embeddings against what the ranks
    put in.  Runs after the timed region; gloo / CPU tensors work too (tests/test_dist_cpu.py calls it at world size 8)."""
    import socket
    import torch
    import torch.distributed as dist
    cuda = torch.cuda.is_available() and (not use_pg or dist.get_backend() == "nccl")
    dev = torch.device("cuda", local_rank) if cuda else torch.device("cpu")
    me = {"rank": rank, "local_rank": local_rank, "host": socket.gethostname(), "pid": os.getpid()}
    if cuda:
        pr = torch.cuda.get_device_properties(local_rank)
        me.update(device=pr.name, pci_bus_id=getattr(pr, "pci_bus_id", None), uuid=str(getattr(pr, "uuid", "")) or None,
                  hbm_gb=round(pr.total_memory / 2 ** 30, 1))
    ones = torch.ones(1, dtype=torch.float64, device=dev)
    tmin = torch.tensor([my_elapsed], dtype=torch.float64, device=dev)
    tmax = tmin.clone()
    rows = torch.tensor([float(ref_rows_local)], dtype=torch.float64, device=dev)
    everyone = [me]
    if use_pg:
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(rows, op=dist.ReduceOp.SUM)
        everyone = [None] * world
        dist.all_gather_object(everyone, me)
    gpus = {(m["host"], m.get("pci_bus_id"), m.get("uuid"), m["local_rank"]) for m in everyone}
    return {"ranks_seen": int(round(float(ones.item()))), "world_size": world,
            "distinct_devices": len(gpus), "ranks": everyone,
            "rank_ms_per_step": {"min": round(1e3 * float(tmin.item()) / steps, 3), "max": round(1e3 * float(tmax.item()) / steps, 3)},
            "ref_rows_all_gathered": int(ref_rows_seen), "ref_rows_contributed_sum": int(round(float(rows.item()))),
            "consistent": bool(int(round(float(ones.item()))) == world and len(gpus) == world
                               and int(ref_rows_seen) == int(round(float(rows.item()))))}


def time_c4(sd, device, steps=10, warmup=3, batch=32, samples=16384):
    """BASELINE.json configs[3]: nomad.forward() as an auxiliary loss (nomad_loss_test.py:60-79 shapes: 2 x (32,1,16384),
    T = 50) - per-step latency of the loss forward and of forward + backward to `estimate`, one GPU.  fp32 (the reference's
    arithmetic) and, next to it, Nomad(precision="bf16x3") (three bf16 MFMA products per fp32 product in every GEMM)."""
    import torch
    from nomad_amd.nomad import Nomad
    g = torch.Generator().manual_seed(0)
    clean_h = (0.1 * torch.randn(batch, 1, samples, generator=g)).clamp(-1, 1)
    noise_h = 0.02 * torch.randn(batch, 1, samples, generator=g)
    # LossNetLayers owns a freshly initialised, never loaded Linear(768, 256) (nomad.py:238-241): seeded here and shared by the
    # two Nomad objects below, so that their losses are comparable
    head_w = (torch.rand(256, 768, generator=g) * 2 - 1) / 768 ** 0.5
    head_b = (torch.rand(256, generator=g) * 2 - 1) / 768 ** 0.5
    out = {"workload": f"configs[3]: nomad.forward() on 2 x ({batch},1,{samples}) (T=50), d loss / d estimate through the whole "
                       f"backbone (feature_grad_mult 0.1)", "dtype": "f32", "steps": steps, "warmup": warmup, "finite": True}
    losses = {}
    for prec in ("fp32", "bf16x3"):
        nmd = Nomad(device=device, weights=sd, precision=prec)
        nmd.lossnet_layers.embedding_weight = head_w.to(nmd.DEVICE).contiguous()
        nmd.lossnet_layers.embedding_bias = head_b.to(nmd.DEVICE).contiguous()
        clean = clean_h.to(nmd.DEVICE)
        est0 = (clean + noise_h.to(nmd.DEVICE)).clamp(-1, 1)

        def fwd():
            return nmd.forward(est0, clean)

        def fwd_bwd():
            est = est0.clone().requires_grad_(True)
            nmd.forward(est, clean).backward()
            return est.grad

        res = {}
        for name, fn in (("forward_ms", fwd), ("forward_backward_ms", fwd_bwd)):
            for _ in range(warmup):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                r = fn()
            torch.cuda.synchronize()
            res[name] = round(1e3 * (time.perf_counter() - t0) / steps, 3)
            out["finite"] = bool(torch.isfinite(r).all().item()) and out["finite"]
        losses[prec] = float(fwd())
        if prec == "fp32":
            # the same step replayed from ONE captured HIP graph (Nomad.graphed_loss: for training loops with a fixed batch shape):
            # same kernels, same order, same bits - only the ~440 launches' host time and the gaps between dependent kernels go
            try:
                eager_grad = fwd_bwd().clone()
                graphed = nmd.graphed_loss(est0, clean)
                for _ in range(warmup):
                    graphed.step(est0, clean)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    gl, gg = graphed.step(est0, clean)
                torch.cuda.synchronize()
                res["forward_backward_graph_replay_ms"] = round(1e3 * (time.perf_counter() - t0) / steps, 3)
                res["graph_replay_bit_identical_to_eager"] = bool(torch.equal(gg, eager_grad))
                del graphed
            except Exception as e:  # noqa: BLE001
                res["forward_backward_graph_replay_ms"] = None
                res["graph_replay_error"] = str(e)[:200]
        nmd.engine.close()
        if prec == "fp32":
            out.update(res)
        else:
            res["loss_rel_diff_vs_f32"] = abs(losses["bf16x3"] - losses["fp32"]) / abs(losses["fp32"])
            out["precision_bf16x3"] = res
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU per step")
    ap.add_argument("--refs", type=int, default=32, help="of which non-matching references")
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--dtype", choices=("f32", "bf16", "bf16x3"), default="f32",
                    help="f32 = the BASELINE metric (configs[1]); bf16 = the long-form config C5 path "
                         "(use with --seconds 30 --batch 32); bf16x3 = fp32-class scores from three bf16 MFMA "
                         "products over hi/lo-split operands (an extra line, never the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the extra bf16x3 measurement after the fp32 one")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with hipEvents")
    ap.add_argument("--weights", choices=("seeded", "peaky"), default="seeded",
                    help="without a real checkpoint: seeded = PyTorch-default-style random init (near-zero attention logits); "
                         "peaky = the same with q/k gain 6 (logit sigma ~ 6: softmax rows with a few dominant keys, as "
                         "trained models have) - the attention kernels are data dependent")
    ap.add_argument("--live-traffic", choices=("auto", "off"), default="auto",
                    help="auto: a 1-GPU headline run takes the two rocprofv3 --pmc passes behind roofline.traffic itself (child "
                         "processes, ~15 s); off: replay profiles/pmc_traffic.json")
    ap.add_argument("--spawn", action="store_true",
                    help="start the ranks through torch.distributed.run even for --gpus 1 (what --gpus N > 1 does by itself)")
    ap.add_argument("--single-stream", action="store_true",
                    help="no two-stream batch split: kernels run alone, as in the pass the roofline is timed in - the command "
                         "the rocprofv3 kernel summaries under profiles/ are taken with")
    args = ap.parse_args()
    if "RANK" not in os.environ and (args.gpus > 1 or args.spawn):
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))   # before anything touches the GPU in this process

    import torch
    import torch.distributed as dist
    from nomad_amd import build
    from nomad_amd.dist import ShardedScorer
    from nomad_amd.engine import Engine
    from nomad_amd.weights import find_checkpoint, load_checkpoint, num_frames, seeded_state_dict

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: nomad_amd has no CPU path")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, this box has {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    # Under the distributed launcher (RANK / WORLD_SIZE in the environment) the RCCL process group is created at EVERY
    # world size, 1 included, and the all-gather of the reference embeddings is really issued - so a 1-GPU box
    # exercises the same collective path the 2/4/8-GPU runs take.  A plain `python bench.py` has no process group.
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    use_pg = world > 1 or launched
    # RCCL prints a version banner on this process's stdout when its communicator comes up (at init or at the first
    # collective); the contract is ONE JSON line on rank 0's stdout, so fd 1 points at stderr until that line is printed
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    if use_pg:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if rank == 0:
        build.build_library()  # no-op when the in-tree .so is up to date; other ranks wait before dlopen
    if use_pg:
        dist.barrier()

    ckpt = find_checkpoint()
    peaky = dict(seed=1, qk_gain=6.0)
    sd = load_checkpoint(ckpt) if ckpt else (seeded_state_dict(**peaky) if args.weights == "peaky" else seeded_state_dict(0))
    eng = Engine(sd, local_rank)
    if args.single_stream:
        eng.F32_SPLIT_ROWS = eng.X3_SPLIT_ROWS = eng.BF16_SPLIT_ROWS = 0
    embed_fn = {"f32": eng.embed, "bf16": eng.embed_bf16, "bf16x3": eng.embed_bf16x3}[args.dtype]
    scorer = ShardedScorer(embed_fn, eng.pairwise, equal_shards=True, force_collective=use_pg)

    n_samples = int(round(args.seconds * 16000))
    B, n_ref = args.batch, args.refs
    g = torch.Generator().manual_seed(1000 + rank)
    wav = (0.1 * torch.randn(B, n_samples, generator=g)).clamp(-1, 1).cuda()  # resident in HBM
    deg_wav, ref_wav = wav[:B - n_ref], wav[B - n_ref:]

    def step():
        return scorer.score(deg_wav, ref_wav, want_matrix=True)

    def fence():
        torch.cuda.synchronize()
        if use_pg:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    profile = not args.no_profile
    # Every precision embeds a batch as two halves on two streams (Engine.embed*): launches overlap, so per-kernel event
    # durations no longer add up to wall time.  Its roofline comes from a second, profiled pass with the split off;
    # `value` from the un-profiled pass the product actually runs.
    split_attr = {"bf16x3": "X3_SPLIT_ROWS", "bf16": "BF16_SPLIT_ROWS", "f32": "F32_SPLIT_ROWS"}.get(args.dtype)
    split_prof = profile and split_attr is not None and getattr(eng, split_attr)
    if profile and not split_prof:
        eng.profile_enable(True)
        eng.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mean, d, ref_all_seen = step()
    fence()
    elapsed = time.perf_counter() - t0
    my_elapsed = elapsed
    prof = eng.profile_read() if profile and not split_prof else None
    if profile and not split_prof:
        eng.profile_enable(False)
    if split_prof:
        keep = getattr(eng, split_attr)
        setattr(eng, split_attr, 0)
        step()
        fence()
        eng.profile_enable(True)
        eng.profile_reset()
        for _ in range(args.steps):
            step()
        fence()
        prof = eng.profile_read()
        eng.profile_enable(False)
        setattr(eng, split_attr, keep)
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if use_pg:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    assert torch.isfinite(mean).all()
    census = rank_census(world, rank, local_rank, use_pg, my_elapsed, args.steps, ref_all_seen.shape[0], n_ref)

    # After the headline measurement (never part of `value`): the same workload through the bf16x3 precision mode,
    # timed the same way, and how far its scores are from the fp32 ones just computed.
    also = None
    if args.dtype == "f32" and not args.no_also:
        try:
            sc3 = ShardedScorer(eng.embed_bf16x3, eng.pairwise, equal_shards=True, force_collective=use_pg)
            for _ in range(args.warmup):
                m3, _, _ = sc3.score(deg_wav, ref_wav, want_matrix=True)
            fence()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                m3, _, _ = sc3.score(deg_wav, ref_wav, want_matrix=True)
            fence()
            t3 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
            if use_pg:
                dist.all_reduce(t3, op=dist.ReduceOp.MAX)
            also = {"precision": "bf16x3 (hi/lo-split bf16 operands, 3 bf16 MFMA products per fp32 product, fp32 accumulate)",
                    "value": round(world * B * args.steps / float(t3.item()), 2), "unit": "clips/s",
                    "ms_per_step": round(1e3 * float(t3.item()) / args.steps, 3),
                    "max_abs_score_diff_vs_f32": float((m3 - mean).abs().max().item())}
        except Exception as e:  # the headline line must not depend on the extra mode
            also = {"precision": "bf16x3", "error": str(e)[:200]}
    # ... and BASELINE.json configs[4] (long-form 30 s clips, bf16, batch 32 per GPU), a few steps (~25 ms each), so that
    # the driver's run carries a number for it too.  Only next to the default headline workload.
    also_c5 = None
    if args.dtype == "f32" and not args.no_also and n_samples == 64000 and B == 256:
        try:
            g5 = torch.Generator().manual_seed(2000 + rank)
            wav5 = (0.1 * torch.randn(32, 480000, generator=g5)).clamp(-1, 1).cuda()
            sc5 = ShardedScorer(eng.embed_bf16, eng.pairwise, equal_shards=True, force_collective=use_pg)
            K5, W5 = 10, 3   # (the steps are ~17 ms: 10 + 3 of them, like `bench.py --dtype bf16 --seconds 30 --steps 10 --warmup 3`, cost 0.2 s)
            for _ in range(W5):
                m5, _, _ = sc5.score(wav5[:28], wav5[28:], want_matrix=True)
            fence()
            t1 = time.perf_counter()
            for _ in range(K5):
                m5, _, _ = sc5.score(wav5[:28], wav5[28:], want_matrix=True)
            fence()
            t5 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
            if use_pg:
                dist.all_reduce(t5, op=dist.ReduceOp.MAX)
            v5 = world * 32 * K5 / float(t5.item())
            # the same loop with the batch on ONE stream (Engine.BF16_SPLIT_ROWS = 0): what a caller who submits one forward at a time
            # gets; since round 6 the N = 768 GEMMs of such a forward run the persistent kernel's 192-row tile mode
            keep5 = eng.BF16_SPLIT_ROWS
            eng.BF16_SPLIT_ROWS = 0
            try:
                for _ in range(W5):
                    m5s, _, _ = sc5.score(wav5[:28], wav5[28:], want_matrix=True)
                fence()
                t1 = time.perf_counter()
                for _ in range(K5):
                    m5s, _, _ = sc5.score(wav5[:28], wav5[28:], want_matrix=True)
                fence()
                t5s = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
                if use_pg:
                    dist.all_reduce(t5s, op=dist.ReduceOp.MAX)
                c5_one = {"value": round(world * 32 * K5 / float(t5s.item()), 2), "unit": "clips/s", "steps": K5, "warmup": W5,
                          "ms_per_step": round(1e3 * float(t5s.item()) / K5, 3),
                          "scores_bit_equal_to_two_streams": bool(torch.equal(m5s, m5))}
            finally:
                eng.BF16_SPLIT_ROWS = keep5
            # the same 32 x 480 000 batch through bf16x3 (fp32-class scores from the bf16 matrix cores) and, once, through fp32:
            # what long-form scores within the north star's 1e-4 cost next to the bf16 figure
            c5_x3 = None
            try:
                # what can fail on ONE rank only (the workspaces of the extra bf16x3 / fp32 passes over 32 x 480 000 samples) happens
                # locally, without a collective, and the ranks agree before the sharded scoring below (ShardedScorer.score all-gathers)
                err5 = None
                try:
                    eng.embed_bf16x3(wav5)
                    eng.embed(wav5)
                    fence_local = torch.cuda.synchronize
                    fence_local()
                except Exception as e:  # noqa: BLE001
                    err5 = e
                agree_or_raise(err5, use_pg, "configs[4] fp32-class leg")
                sc5x = ShardedScorer(eng.embed_bf16x3, eng.pairwise, equal_shards=True, force_collective=use_pg)
                for _ in range(2):
                    m5x, _, _ = sc5x.score(wav5[:28], wav5[28:], want_matrix=True)
                fence()
                t1 = time.perf_counter()
                for _ in range(3):
                    m5x, _, _ = sc5x.score(wav5[:28], wav5[28:], want_matrix=True)
                fence()
                t5x = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
                sc5f = ShardedScorer(eng.embed, eng.pairwise, equal_shards=True, force_collective=use_pg)
                m5f, _, _ = sc5f.score(wav5[:28], wav5[28:], want_matrix=True)
                fence()
                t1 = time.perf_counter()
                m5f, _, _ = sc5f.score(wav5[:28], wav5[28:], want_matrix=True)
                fence()
                t5f = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
                if use_pg:
                    dist.all_reduce(t5x, op=dist.ReduceOp.MAX)
                    dist.all_reduce(t5f, op=dist.ReduceOp.MAX)
                c5_x3 = {"precision": "bf16x3 (hi/lo-split bf16 operands, 3 bf16 MFMA products per fp32 product, fp32 accumulate)",
                         "value": round(world * 32 * 3 / float(t5x.item()), 2), "unit": "clips/s", "steps": 3,
                         "ms_per_step": round(1e3 * float(t5x.item()) / 3, 3),
                         "max_abs_score_diff_vs_f32": float((m5x - m5f).abs().max().item()),
                         "bf16_max_abs_score_diff_vs_f32": float((m5 - m5f).abs().max().item()),
                         "f32_same_batch": {"value": round(world * 32 / float(t5f.item()), 2), "unit": "clips/s", "steps": 1,
                                            "ms_per_step": round(1e3 * float(t5f.item()), 3)}}
            except Exception as e:
                c5_x3 = {"precision": "bf16x3", "error": str(e)[:200]}
            also_c5 = {"workload": "configs[4]: batch=32 x 480000 samples (T=1499) per GPU, bf16 storage / fp32 accumulate, "
                                   "28 deg x 4*N ref float64 distances + means",
                       "dtype": "bf16", "value": round(v5, 2), "unit": "clips/s", "steps": K5, "warmup": W5,
                       "ms_per_step": round(1e3 * float(t5.item()) / K5, 3),
                       "model_frac_of_mfma_peak": round(v5 * 500.044e9 / world / 2.5e15, 4),
                       "finite": bool(torch.isfinite(m5).all().item()),
                       "one_stream": c5_one,
                       "fp32_class_scores_same_batch": c5_x3,
                       "context": "STATIC notes, not measured in this run: this forward holds a shader clock of 1997-2090 MHz of the 2400 nominal "
                                  "(profiles/r05_clock_c5.jsonl, r06_clock_c5.jsonl); alone on the GPU the shipped persistent bf16 GEMM does 1016-1146 TFLOP/s on QKV, "
                                  "903-1018 on fc1 + GELU, 985-1108 on fc2 (192-row tiles), hipBLASLt (no epilogue) 933-967 / 949-1041 / 1115-1180 "
                                  "(profiles/r05_vendor_yardstick_bf16.jsonl, r06_gemm_bf16_p9_short_ab.jsonl); per-shape times inside the forward: "
                                  "profiles/r06_c5_layer_table_*.json; what the GEMM's epilogues cost per tile: profiles/r06_gemm_bf16_p9_epilogue_cost.txt"}
            del wav5
        except Exception as e:
            also_c5 = {"workload": "configs[4]", "error": str(e)[:200]}
    # ... BASELINE.json configs[2] (10 000 x 1 000, end to end, sharded over the ranks of this run) and configs[3]
    # (nomad.forward() latency, one GPU: rank 0), and the attention kernels' sensitivity to the weights: the headline
    # workload and configs[4] once more on the "peaky" seeded weights.  All after the headline, never part of `value`.
    also_c3 = also_c4 = also_peaky = None
    headline = args.dtype == "f32" and not args.no_also and n_samples == 64000 and B == 256
    if headline:
        try:
            also_c3 = time_c3(eng, world, rank, use_pg, fence)
        except Exception as e:
            also_c3 = {"workload": "configs[2]", "error": str(e)[:200]}
        if rank == 0:
            try:
                also_c4 = time_c4(sd, local_rank)
            except Exception as e:
                also_c4 = {"workload": "configs[3]", "error": str(e)[:200]}
    if headline and not ckpt and args.weights == "seeded":
        engp = None
        keep = (eng.F32_SPLIT_ROWS, eng.BF16_SPLIT_ROWS)
        try:
            engp = Engine(seeded_state_dict(**peaky), local_rank)
            engp.F32_SPLIT_ROWS = engp.BF16_SPLIT_ROWS = 0    # kernels run alone: the per-kernel times below are theirs
            scp = ShardedScorer(engp.embed, engp.pairwise, equal_shards=True, force_collective=use_pg)
            res = {"weights": "seeded_state_dict(seed=1, qk_gain=6.0): attention logits with sigma ~ 6 instead of ~ 0"}
            g5 = torch.Generator().manual_seed(2000 + rank)
            wav5 = (0.1 * torch.randn(32, 480000, generator=g5)).clamp(-1, 1).cuda()
            sc5 = ShardedScorer(engp.embed_bf16, engp.pairwise, equal_shards=True, force_collective=use_pg)
            # the same loops on the headline's (seeded) engine with its batch split off: the like-for-like reference
            eng.F32_SPLIT_ROWS = eng.BF16_SPLIT_ROWS = 0
            sc5s = ShardedScorer(eng.embed_bf16, eng.pairwise, equal_shards=True, force_collective=use_pg)
            sc2s = ShardedScorer(eng.embed, eng.pairwise, equal_shards=True, force_collective=use_pg)
            for tag, e_, loops in (("peaky", engp, (("c2", lambda: scp.score(deg_wav, ref_wav, want_matrix=True), B, args.steps),
                                                   ("c5", lambda: sc5.score(wav5[:28], wav5[28:], want_matrix=True), 32, 5))),
                                   ("seeded", eng, (("c2", lambda: sc2s.score(deg_wav, ref_wav, want_matrix=True), B, args.steps),
                                                    ("c5", lambda: sc5s.score(wav5[:28], wav5[28:], want_matrix=True), 32, 5)))):
                for key, fn, n_clips, k in loops:
                    for _ in range(2):
                        fn()
                    fence()
                    e_.profile_enable(True)
                    e_.profile_reset()
                    t1 = time.perf_counter()
                    for _ in range(k):
                        fn()
                    fence()
                    tp = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
                    pp = e_.profile_read()
                    e_.profile_enable(False)
                    if use_pg:
                        dist.all_reduce(tp, op=dist.ReduceOp.MAX)
                    row = {"value": round(world * n_clips * k / float(tp.item()), 2), "unit": "clips/s",
                           "ms_per_step": round(1e3 * float(tp.item()) / k, 3),
                           "attention_ms_per_step": round(pp["attention_mfma"]["ms"] / k, 3), "single_stream": True}
                    if tag == "peaky":
                        res[key] = row
                    else:
                        res[key]["seeded_weights_same_loop"] = row
            del wav5
            also_peaky = res
        except Exception as e:
            also_peaky = {"error": str(e)[:200]}
        finally:   # whatever happened above: the headline engine gets its split back, the second engine is released
            eng.F32_SPLIT_ROWS, eng.BF16_SPLIT_ROWS = keep
            if engp is not None:
                engp.profile_enable(False)
                engp.close()

    if rank == 0:
        clips = world * B * args.steps
        value = clips / elapsed
        T = num_frames(n_samples)
        flop_clip = {64000: FLOP_PER_CLIP_4S, 480000: 500.044e9}.get(n_samples)
        # which BASELINE.json config this command line is (anything else is labelled as what it is)
        if args.dtype == "f32" and n_samples == 64000 and B == 256:
            cfg_name = "configs[1]"
        elif args.dtype == "bf16" and n_samples == 480000:
            cfg_name = "configs[4] (long-form 30 s clips, bf16)"
        else:
            cfg_name = "custom (not a BASELINE.json config)"
        # bf16x3 executes 3 bf16 MFMA flops per algorithmic (fp32-equivalent) flop: its ceiling is a third of the bf16 peak
        peak = {"f32": PEAK_FP32_MFMA, "bf16": 2.5e15, "bf16x3": 2.5e15 / 3}[args.dtype]
        out = {
            "metric": "clips/sec embedded + NxM NOMAD distances, 16kHz x 4s batches",
            "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic (0.1*randn waveforms, seed 1000+rank; " +
                    ("real nomad_best_model.pt" if ckpt else f"{args.weights} random-init wav2vec2-base + head weights") + ")",
            "config": {"workload": f"{cfg_name}: batch={B} x {n_samples} samples (T={T}) per GPU, wav2vec2-base + "
                                   f"projection head {PRECISION_NOTE[args.dtype]}, {B - n_ref} deg x {n_ref}*N "
                                   f"ref float64 distances + means",
                       "clips_per_gpu_per_step": B, "deg_per_gpu": B - n_ref, "ref_total": n_ref * world,
                       "parallelism": f"clip-sharded x{world}, all-gather of ref embeddings",
                       "collective": (f"RCCL all_gather_into_tensor executed every step (torch.distributed nccl, "
                                      f"version {'.'.join(map(str, torch.cuda.nccl.version()))})" if use_pg
                                      else "none (single process, no process group)")},
        }
        out["ranks_seen"] = census["ranks_seen"]
        out["rank_census"] = census
        if flop_clip:
            out["model_tflops_per_gpu"] = round(value * flop_clip / world / 1e12, 2)
            out["model_frac_of_mfma_peak"] = round(value * flop_clip / world / peak, 4)
            if n_samples == 64000:
                out["encoder_layers_frac_of_mfma_peak"] = round(value * FLOP_LAYERS_4S / world / peak, 4)
        traffic_tab = None
        traffic_live = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if args.dtype == "f32" and n_samples == 64000 and B == 256:
            # HBM-side bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
            # (fabric-side counters, Infinity-Cache hits included).  On a 1-GPU headline run the two passes are taken NOW, as
            # child processes (tools/pmc_traffic.py:collect, ~15 s; this process is idle meanwhile); otherwise, or if that
            # fails, the table committed under profiles/ (same passes, taken by tools/gpu_pmc_traffic.sh) is replayed and says so.
            # (never from inside a profiler: a bench.py that is itself the target of rocprofv3 would start nested profiled runs)
            under_profiler = any(k in os.environ for k in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_METRICS_PATH")) or \
                "rocprof" in os.environ.get("LD_PRELOAD", "")
            if world == 1 and prof and args.live_traffic != "off" and not under_profiler:
                try:
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    import pmc_traffic
                    scratch = os.path.join(ROOT, "gpurun_out", "bench_live_pmc")
                    try:
                        os.makedirs(scratch, exist_ok=True)
                    except OSError:
                        import tempfile
                        scratch = tempfile.mkdtemp(prefix="bench_live_pmc_")
                    traffic_tab = pmc_traffic.collect(scratch)
                    traffic_live = "measured in this run"
                except Exception as e:  # noqa: BLE001 - any failure falls back to the committed table, labelled
                    traffic_live = f"live passes failed ({str(e)[:120]})"
                    traffic_tab = None
            if traffic_tab is None and os.path.isfile(tfile):
                traffic_tab = json.load(open(tfile))
        if prof:
            def rate(cls):
                return cls["flops"] / (cls["ms"] * 1e-3) / 1e12 if cls["ms"] > 0 else 0.0
            allg, big, fine = prof["gemm_mfma_all"], prof["gemm_mfma_256x128"], prof["gemm_mfma_128x64"]
            dom = big if big["ms"] >= fine["ms"] else fine      # the dominant kernel = the instantiation with most time
            if args.dtype == "bf16x3":
                kname = "gemm_bf16x3_kernel 256x256 (3 x v_mfma_f32_16x16x32_bf16 per fp32-equivalent product)"
            elif args.dtype == "bf16":
                kname = ("gemm_bf16_8phase_kernel 256x256 (v_mfma_f32_16x16x32_bf16) + gemm_bf16_glds_kernel 128x128/256x256"
                         if dom is big else "gemm_bf16_glds_kernel 128x64 (v_mfma_f32_32x32x16_bf16)")
            else:
                kname = ("gemm_f32_glds_kernel<256,128,16,4,2,3> + gemm_f32_mixed_kernel (256x128 tiles, 128x128 tiles for the rows of the last round)"
                         if dom is big else "gemm_f32_glds_kernel<128,128,32,4,2 / n48>") + " (v_mfma_f32_16x16x4_f32)"
            ach = rate(dom)
            traffic = alg_bytes = None
            if traffic_tab:
                row = traffic_tab["gemm_256x128" if dom is big else "gemm_128x64"]
                traffic = round(row["hbm_bytes_per_launch"])
                alg_bytes = round(row.get("algorithmic_bytes_per_launch", 0)) or None
            out["roofline"] = {"bound": "mfma", "kernel": kname,
                               "achieved": round(ach, 2), "peak": round(peak / 1e12, 1), "unit": "TFLOP/s",
                               "frac": round(ach * 1e12 / peak, 4), "traffic": traffic,
                               "traffic_source": (None if traffic is None else
                                                  "LIVE: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, two separate passes of "
                                                  "`bench.py --steps 1 --warmup 1 --single-stream` run as child processes at the end of this run "
                                                  "(tools/pmc_traffic.py:collect); fabric-side bytes per launch of the dominant kernel"
                                                  if traffic_live == "measured in this run" else
                                                  f"profiles/pmc_traffic.json (STATIC: the same two passes taken earlier "
                                                  f"({traffic_tab.get('taken', 'round 2')}) by tools/gpu_pmc_traffic.sh, not measured in this run"
                                                  f"{'; ' + traffic_live if traffic_live else ''})"),
                               "algorithmic_bytes_per_launch": alg_bytes,
                               "launches": dom["launches"], "avg_launch_ms": round(dom["ms"] / max(dom["launches"], 1), 4),
                               "algorithmic_gflop_per_launch": round(dom["flops"] / max(dom["launches"], 1) / 1e9, 3),
                               "share_of_step_time": round(dom["ms"] / args.steps / (1e3 * elapsed / args.steps), 4),
                               "all_gemm_launches": {"achieved": round(rate(allg), 2), "launches": allg["launches"],
                                                     "frac": round(rate(allg) * 1e12 / peak, 4)},
                               "other_instantiation": {"achieved": round(rate(fine if dom is big else big), 2),
                                                       "launches": (fine if dom is big else big)["launches"]}}
            out["kernel_time_ms_per_step"] = {k: round(v["ms"] / args.steps, 3) for k, v in prof.items()}
            if split_prof:
                out["roofline"]["note"] = ("per-kernel timings from a second pass with the two-stream batch split off "
                                           "(kernels run alone); value / ms_per_step from the pass with it on")
        if also:
            out["also_measured"] = also
        if also_c5:
            out["also_measured_c5"] = also_c5
        if also_c3:
            out["also_measured_c3"] = also_c3
        if also_c4:
            out["also_measured_c4"] = also_c4
        if also_peaky:
            out["also_measured_peaky"] = also_peaky
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, n_samples)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
