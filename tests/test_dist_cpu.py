"""CPU, gloo, world_size 2: the sharded scoring path (nomad_amd/dist.py) equals the unsharded one.
The GPU compute callables are replaced by deterministic CPU stand-ins; what is under test is the
partitioning, the all-gather (equal and unequal shard sizes) and the slab bookkeeping."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nomad_amd.dist import ShardedScorer, all_gather_rows, partition


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_embed(wav):  # (B,N) -> (B,256) unit norm, deterministic per clip, batch-invariant
    g = torch.Generator().manual_seed(0)
    proj = torch.randn(wav.shape[1], 256, generator=g, dtype=torch.float64)
    e = wav.double() @ proj
    return torch.nn.functional.normalize(e, dim=1).float()


def _fake_pairwise(deg, ref, want_matrix):
    d = torch.cdist(deg.double(), ref.double())
    return (d if want_matrix else None), d.mean(dim=1)


def _worker(rank, world, port, n_deg, n_ref, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(1234)
    deg = torch.randn(n_deg, 320, generator=g)
    ref = torch.randn(n_ref, 320, generator=g)
    ds, de = partition(n_deg, world, rank)
    rs, re_ = partition(n_ref, world, rank)
    scorer = ShardedScorer(_fake_embed, _fake_pairwise, equal_shards=(n_ref % world == 0))
    mean, d, ref_all = scorer.score(deg[ds:de], ref[rs:re_], want_matrix=True)
    full = scorer.gather_scores(mean)
    rows = all_gather_rows(d)
    if rank == 0:
        np.savez(os.path.join(out_dir, "out.npz"), mean=full.numpy(), dist=rows.numpy(), ref=ref_all.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _run(n_deg, n_ref, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_deg, n_ref, str(tmp_path)), nprocs=2, join=True)
    out = np.load(os.path.join(tmp_path, "out.npz"))
    g = torch.Generator().manual_seed(1234)
    deg = torch.randn(n_deg, 320, generator=g)
    ref = torch.randn(n_ref, 320, generator=g)
    d, m = _fake_pairwise(_fake_embed(deg), _fake_embed(ref), True)
    assert np.array_equal(out["ref"], _fake_embed(ref).numpy())
    assert np.abs(out["dist"] - d.numpy()).max() < 1e-12
    assert np.abs(out["mean"] - m.numpy()).max() < 1e-12


def test_sharded_equals_unsharded_even(tmp_path):
    _run(12, 6, tmp_path)


def test_sharded_equals_unsharded_ragged(tmp_path):
    _run(13, 5, tmp_path)  # unequal shards: 7+6 deg, 3+2 ref


# ---- bench.py's configs[2] leg (10 000 x 1 000 sharded over the ranks of the run) at world size 4, unequal shards ----------
def _c3_worker(rank, world, port, n_deg, n_ref, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(99)
    deg = torch.randn(n_deg, 320, generator=g)
    ref = torch.randn(n_ref, 320, generator=g)
    (ds, de), (rs, re_) = partition(n_deg, world, rank), partition(n_ref, world, rank)
    wav = torch.cat([deg[ds:de], ref[rs:re_]])
    pw = lambda a, b, m: _fake_pairwise(a, b, m)                                               # noqa: E731
    scores, ref_all, slab = bench.c3_sharded_scores(_fake_embed, pw, wav, de - ds, 2, True)   # batches of 2 clips
    if rank == 0:
        np.savez(os.path.join(out_dir, "c3.npz"), scores=scores.numpy(), ref=ref_all.numpy())
    assert slab.shape == (de - ds, n_ref)
    dist.barrier()
    dist.destroy_process_group()


def test_bench_c3_leg_sharded_over_four_ranks_equals_unsharded(tmp_path):
    n_deg, n_ref, world = 10, 7, 4                      # shards 3+3+2+2 degraded, 2+2+2+1 references
    mp.spawn(_c3_worker, args=(world, _free_port(), n_deg, n_ref, str(tmp_path)), nprocs=world, join=True)
    out = np.load(os.path.join(tmp_path, "c3.npz"))
    g = torch.Generator().manual_seed(99)
    deg = torch.randn(n_deg, 320, generator=g)
    ref = torch.randn(n_ref, 320, generator=g)
    _, m = _fake_pairwise(_fake_embed(deg), _fake_embed(ref), True)
    assert np.array_equal(out["ref"], _fake_embed(ref).numpy())
    assert out["scores"].shape == (n_deg,) and np.abs(out["scores"] - m.numpy()).max() < 1e-12


def _c3_census_worker(rank, world, port, n_deg, n_ref, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7)
    deg = torch.randn(n_deg, 64, generator=g)
    ref = torch.randn(n_ref, 64, generator=g)
    (ds, de), (rs, re_) = partition(n_deg, world, rank), partition(n_ref, world, rank)
    wav = torch.cat([deg[ds:de], ref[rs:re_]])
    emb = lambda w: torch.nn.functional.normalize(torch.cat([w.double(), w.double() ** 2], 1)[:, :128].repeat(1, 2), dim=1).float()   # noqa: E731
    pw = lambda a, b, m: _fake_pairwise(a, b, m)                                                                                       # noqa: E731
    scores, ref_all, slab = bench.c3_sharded_scores(emb, pw, wav, de - ds, 256, True)
    census = bench.rank_census(world, rank, rank, True, 0.001 * (rank + 1), 1, ref_all.shape[0], re_ - rs)
    assert slab.shape == (de - ds, n_ref)
    if rank == 0:
        import json
        np.savez(os.path.join(out_dir, "c3w8.npz"), scores=scores.numpy(), ref=ref_all.numpy())
        json.dump(census, open(os.path.join(out_dir, "census.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_c3_leg_world_8_uneven_shards_and_rank_census(tmp_path):
    """configs[2]'s sharded leg at the world size the driver's 8-GPU run uses, with shards that do not divide
    (10 001 x 1 003: 1251 / 1250 degraded, 126 / 125 references per rank), and the census bench.py prints so that its line is
    self-verifying: ranks_seen from an all-reduce, the all-gathered reference rows against the sum of what the ranks put in,
    min / max of the per-rank timed loops."""
    n_deg, n_ref, world = 10001, 1003, 8
    mp.spawn(_c3_census_worker, args=(world, _free_port(), n_deg, n_ref, str(tmp_path)), nprocs=world, join=True)
    out = np.load(os.path.join(tmp_path, "c3w8.npz"))
    import json
    census = json.load(open(os.path.join(tmp_path, "census.json")))
    g = torch.Generator().manual_seed(7)
    deg = torch.randn(n_deg, 64, generator=g)
    ref = torch.randn(n_ref, 64, generator=g)
    emb = lambda w: torch.nn.functional.normalize(torch.cat([w.double(), w.double() ** 2], 1)[:, :128].repeat(1, 2), dim=1).float()   # noqa: E731
    _, m = _fake_pairwise(emb(deg), emb(ref), True)
    assert np.array_equal(out["ref"], emb(ref).numpy())
    assert out["scores"].shape == (n_deg,) and np.abs(out["scores"] - m.numpy()).max() < 1e-12
    assert census["ranks_seen"] == 8 and census["world_size"] == 8 and len(census["ranks"]) == 8
    assert sorted(r["rank"] for r in census["ranks"]) == list(range(8))
    assert census["ref_rows_all_gathered"] == n_ref == census["ref_rows_contributed_sum"]
    assert census["rank_ms_per_step"] == {"min": 1.0, "max": 8.0}
    assert census["distinct_devices"] == 8 and census["consistent"]


# ---- data-parallel fine-tuning: gradient averaging across ranks (gloo, world_size 2) ------------------------------
def _ddp_worker(rank, world, port, out):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nomad_amd.train import allreduce_mean_gradients

    class FakeEngine:  # the two calls allreduce_mean_gradients makes on an Engine
        def __init__(self):
            self.grad = torch.arange(10, dtype=torch.float32) * (rank + 1)
        def train_read(self, what):
            assert what == 1
            return self.grad.clone()
        def train_write(self, what, flat):
            assert what == 1
            self.grad = flat.clone()
    eng = FakeEngine()
    allreduce_mean_gradients(eng)
    out.put((rank, eng.grad.tolist()))
    dist.destroy_process_group()


def test_allreduce_mean_gradients_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400) + 17
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [i * 1.5 for i in range(10)]  # mean of 1x and 2x
    assert res[0] == want and res[1] == want


# ---- force_collective: a group of ONE rank still issues the collectives (what the 1-GPU RCCL test relies on) ----------
def _ws1_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = {"n": 0}
    real = dist.all_gather_into_tensor

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    dist.all_gather_into_tensor = counting
    g = torch.Generator().manual_seed(5)
    deg, ref = torch.randn(5, 320, generator=g), torch.randn(3, 320, generator=g)
    res = {}
    for name, force, equal in (("skip", False, True), ("equal", True, True), ("sizes", True, False)):
        calls["n"] = 0
        sc = ShardedScorer(_fake_embed, _fake_pairwise, equal_shards=equal, force_collective=force)
        mean, d, ref_all = sc.score(deg, ref, want_matrix=True)
        res[name] = (calls["n"], mean.clone(), d.clone(), ref_all.clone())
    np.savez(os.path.join(out_dir, "ws1.npz"), calls=np.array([res[k][0] for k in ("skip", "equal", "sizes")]),
             same=np.array([all(torch.equal(res["skip"][i], res[k][i]) for i in (1, 2, 3)) for k in ("equal", "sizes")]))
    dist.destroy_process_group()


def test_force_collective_in_a_group_of_one(tmp_path):
    mp.spawn(_ws1_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    out = np.load(os.path.join(tmp_path, "ws1.npz"))
    assert out["calls"].tolist() == [0, 1, 2]     # none / the all-gather / size exchange + all-gather
    assert out["same"].all()


# ---- Nomad.predict inside a torch.distributed job: files shard across ranks ------------------------------------------
class _FileEngine:
    """CPU stand-in for nomad_amd.engine.Engine: a clip's "embedding" is a deterministic function of its samples."""
    device = torch.device("cpu")

    def pack_ragged_host(self, waves):
        lens = [int(w.shape[0]) for w in waves]
        host = np.zeros((len(waves), max(lens)), dtype=np.float32)
        for i, w in enumerate(waves):
            host[i, :lens[i]] = np.asarray(w)
        return host, lens

    def embed_ragged(self, waves, precision=None, packed=None):
        host, lens = packed
        host = host.numpy() if not isinstance(host, np.ndarray) else host
        out = np.zeros((len(lens), 256), dtype=np.float32)
        for i, n in enumerate(lens):
            x = host[i, :n].astype(np.float64)
            out[i, :3] = n / 1e4, x.sum(), np.abs(x).mean()
        return out

    def fetch_async(self, emb):
        class F:
            def result(self_inner):
                return emb
        return F()

    def pairwise(self, deg, ref, want_matrix=False):
        d = torch.cdist(deg.double(), ref.double())
        return (d if want_matrix else None), d.mean(dim=1)


def _file_nomad():
    from nomad_amd import wavio
    from nomad_amd.nomad import Nomad
    n = Nomad.__new__(Nomad)
    n.engine, n.precision, n.group, n.model = _FileEngine(), "fp32", None, None
    n.load_processing = lambda p, trim=False: torch.from_numpy(wavio.load_processing(p, 16000, trim))
    return n


def _write_dirs(root, n_deg, n_ref):
    import struct
    rng = np.random.default_rng(5)
    for sub, n in (("nmr", n_ref), ("deg", n_deg)):
        os.makedirs(os.path.join(root, sub))
        for i in range(n):
            x = (0.2 * rng.standard_normal(int(rng.integers(300, 3000))) * 32767).astype("<i2").tobytes()
            with open(os.path.join(root, sub, f"{sub}{i:03d}.wav"), "wb") as f:
                f.write(b"RIFF" + struct.pack("<I", 36 + len(x)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16) +
                        b"data" + struct.pack("<I", len(x)) + x)


def _predict_worker(rank, world, port, root):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = os.path.join(root, "out_dist")
    avg, dm = _file_nomad().predict("dir", os.path.join(root, "nmr"), os.path.join(root, "deg"), results_path=out)
    avg.to_pickle(os.path.join(root, f"avg_rank{rank}.pkl"))     # every rank returns the full tables
    dm.to_pickle(os.path.join(root, f"dm_rank{rank}.pkl"))
    dist.barrier()
    dist.destroy_process_group()


def test_predict_shards_files_across_ranks(tmp_path):
    """world_size 2 (gloo): each rank embeds its slice of both directories, the all-gathers put the tables back in
    listing order, each rank computes its slab of the distance matrix, rank 0 writes the CSV files - byte-identical to
    the single-process run; 7 degraded files do not divide evenly, 1 reference file leaves a rank with none."""
    import pandas as pd
    for n_deg, n_ref in ((7, 4), (3, 1)):
        root = str(tmp_path / f"d{n_deg}_{n_ref}")
        os.makedirs(root)
        _write_dirs(root, n_deg, n_ref)
        os.makedirs(os.path.join(root, "out_dist"))
        os.makedirs(os.path.join(root, "out_single"))
        mp.spawn(_predict_worker, args=(2, _free_port(), root), nprocs=2, join=True)
        avg, dm = _file_nomad().predict("dir", os.path.join(root, "nmr"), os.path.join(root, "deg"),
                                        results_path=os.path.join(root, "out_single"))
        assert avg.shape == (n_deg, 1) and dm.shape == (n_deg, n_ref)
        for r in range(2):
            assert pd.read_pickle(os.path.join(root, f"avg_rank{r}.pkl")).equals(avg)
            assert pd.read_pickle(os.path.join(root, f"dm_rank{r}.pkl")).equals(dm)
        for name in ("nomad_avg.csv", "nomad_scores.csv"):
            assert open(os.path.join(root, "out_dist", name), "rb").read() == open(os.path.join(root, "out_single", name), "rb").read()


def test_predict_with_an_empty_file_list(tmp_path):
    """csv mode with no rows (an empty directory already fails in get_embeddings, exactly like the reference's
    nomad.py:148-151): cdist + np.mean(axis=1) of the reference give an empty matrix for no degraded files and NaN means for
    no references; the engine's kernels take no empty operands, so predict answers that itself."""
    import pandas as pd
    root = str(tmp_path)
    _write_dirs(root, 3, 2)
    full = lambda sub: pd.DataFrame({"filename": sorted(os.path.join(root, sub, f) for f in os.listdir(os.path.join(root, sub)))})
    empty = pd.DataFrame({"filename": []})
    for name, df in (("nmr.csv", full("nmr")), ("deg.csv", full("deg")), ("none.csv", empty)):
        df.to_csv(os.path.join(root, name), index=False)
    os.makedirs(os.path.join(root, "o1"))
    os.makedirs(os.path.join(root, "o2"))
    n = _file_nomad()
    avg, dm = n.predict("csv", os.path.join(root, "nmr.csv"), os.path.join(root, "none.csv"), results_path=os.path.join(root, "o1"))
    assert avg.shape == (0, 1) and dm.shape == (0, 2)
    avg, dm = n.predict("csv", os.path.join(root, "none.csv"), os.path.join(root, "deg.csv"), results_path=os.path.join(root, "o2"))
    assert avg.shape == (3, 1) and dm.shape == (3, 0) and avg["NOMAD"].isna().all()
    assert open(os.path.join(root, "o2", "nomad_avg.csv")).read().count("\n") == 4


def _predict_bad_worker(rank, world, port, root):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _file_nomad().predict("dir", os.path.join(root, "nmr"), os.path.join(root, "deg"), results_path=os.path.join(root, "out"))
        outcome = "no error"
    except ValueError as e:
        outcome = f"ValueError: {e}"
    except RuntimeError as e:
        outcome = f"RuntimeError: {e}"
    with open(os.path.join(root, f"outcome{rank}.txt"), "w") as f:
        f.write(outcome)
    dist.barrier()
    dist.destroy_process_group()


def test_a_bad_file_on_one_rank_fails_every_rank_instead_of_hanging(tmp_path):
    """The last degraded file (rank 1's slice) is not a wav: rank 1 raises the front end's ValueError, rank 0 - whose own
    files were fine - raises too instead of waiting in the all-gather."""
    root = str(tmp_path)
    _write_dirs(root, 6, 2)
    os.makedirs(os.path.join(root, "out"))
    bad = sorted(os.listdir(os.path.join(root, "deg")))[-1]
    open(os.path.join(root, "deg", bad), "wb").write(b"RIFFxxxxWAVEnope")
    listing = os.listdir(os.path.join(root, "deg"))
    mp.spawn(_predict_bad_worker, args=(2, _free_port(), root), nprocs=2, join=True)
    outcomes = [open(os.path.join(root, f"outcome{r}.txt")).read() for r in range(2)]
    owner = 0 if listing.index(bad) < 3 else 1                        # os.listdir order decides whose slice it is in
    assert outcomes[owner].startswith("ValueError") and bad in outcomes[owner]
    assert outcomes[1 - owner].startswith("RuntimeError") and f"[{owner}]" in outcomes[1 - owner]
