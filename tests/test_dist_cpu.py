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
