"""Child process of tests/test_gpu_rccl.py: the multi-GPU scoring path on the REAL collective backend (RCCL via
torch.distributed "nccl") in a process group of one rank - the only RCCL execution a 1-GPU box can give.

Checks, with the real Engine:
  * the sharded scorer with the all-gather really issued (force_collective) equals the unsharded result bit for bit;
  * the same on a non-default stream, with fresh inputs every iteration: the collective must be ordered after the
    engine's embedding kernels and before its distance kernel on whatever stream the engine launches on;
  * the unequal-shard path (size exchange + padded all-gather) and the final gather of the scores;
  * Nomad.predict on the shipped example files with its file sharding forced on: same tables, byte-identical CSV files.
Prints RCCL_WS1_OK on success.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from nomad_amd.dist import ShardedScorer, all_gather_rows
    from nomad_amd.engine import Engine
    from nomad_amd.weights import seeded_state_dict

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == 1 and torch.cuda.is_available()
    torch.cuda.set_device(local_rank)
    if "MASTER_ADDR" not in os.environ:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = os.environ.get("NOMAD_TEST_PORT", "29547")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert dist.get_backend() == "nccl"
    print("RCCL version", torch.cuda.nccl.version(), flush=True)

    eng = Engine(seeded_state_dict(0), local_rank)
    sharded = ShardedScorer(eng.embed, eng.pairwise, equal_shards=True, force_collective=True)
    ragged = ShardedScorer(eng.embed, eng.pairwise, equal_shards=False, force_collective=True)
    side = torch.cuda.Stream()
    gen = torch.Generator().manual_seed(0)
    n_coll = 0
    for it in range(6):
        wav = (0.1 * torch.randn(12, 16000, generator=gen)).clamp(-1, 1).cuda()
        deg, ref = wav[:8], wav[8:]
        # unsharded truth: no process-group call at all
        emb = eng.embed(wav)
        d0, m0 = eng.pairwise(emb[:8].contiguous(), emb[8:].contiguous(), True)
        torch.cuda.synchronize()
        for scorer in (sharded, ragged):
            for stream in (None, side):
                if stream is None:
                    m, d, ref_all = scorer.score(deg, ref, want_matrix=True)
                    scores = scorer.gather_scores(m)
                else:
                    stream.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(stream):
                        m, d, ref_all = scorer.score(deg, ref, want_matrix=True)
                        scores = scorer.gather_scores(m)
                    torch.cuda.current_stream().wait_stream(stream)
                n_coll += 2 if scorer is sharded else 4
                torch.cuda.synchronize()
                assert torch.equal(ref_all, emb[8:]), (it, "ref embeddings after the all-gather")
                assert torch.equal(d, d0) and torch.equal(m, m0) and torch.equal(scores, m0), (it, "scores")
    # a bare all-gather of a tensor the previous kernel on this stream has just written
    x = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
    for k in range(8):
        x = x * 1.0001 + k
        y = all_gather_rows(x, equal=True, force_collective=True)
        assert y.data_ptr() != x.data_ptr() and torch.equal(y, x)
    # Nomad.predict on the shipped example files with its file sharding switched on in this group of one rank
    # (NOMAD_FORCE_COLLECTIVE=1): the embedding tables and the distance slab go through RCCL all-gathers and must come back
    # identical to the run without a collective
    import tempfile
    from nomad_amd.nomad import Nomad
    wavs = os.path.join(ROOT, "tests", "golden", "wavs")
    nmd = Nomad(weights=seeded_state_dict(0), device=local_rank)
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(d + "/a")
        os.makedirs(d + "/b")
        os.environ.pop("NOMAD_FORCE_COLLECTIVE", None)
        avg0, dm0 = nmd.predict("dir", os.path.join(wavs, "nmr-data"), os.path.join(wavs, "test-data"), results_path=d + "/a")
        os.environ["NOMAD_FORCE_COLLECTIVE"] = "1"
        avg1, dm1 = nmd.predict("dir", os.path.join(wavs, "nmr-data"), os.path.join(wavs, "test-data"), results_path=d + "/b")
        os.environ.pop("NOMAD_FORCE_COLLECTIVE", None)
        assert avg1.equals(avg0) and dm1.equals(dm0) and dm1.shape == (2, 4)
        for name in ("nomad_avg.csv", "nomad_scores.csv"):
            assert open(f"{d}/a/{name}", "rb").read() == open(f"{d}/b/{name}", "rb").read()
    nmd.engine.close()
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    eng.close()
    print(f"RCCL_WS1_OK collectives={n_coll + 8} predict_sharded=ok", flush=True)


if __name__ == "__main__":
    main()
