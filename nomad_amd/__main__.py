"""``python -m nomad_amd --mode dir --nmr P --deg P`` - CLI of the reference
(/root/reference/src/nomad_audio/__main__.py:4-18); ``--device`` is honoured here (the reference
parses it and ignores it).

Several GPUs: ``python -m torch.distributed.run --nproc-per-node N -m nomad_amd --mode dir ...`` - one process per GPU,
the files shard across the ranks (RCCL all-gather of the embeddings, ``Nomad.predict``), rank 0 writes the CSV files."""
import argparse
import os

from .nomad import Nomad


def main():
    ap = argparse.ArgumentParser(prog="nomad_amd")
    ap.add_argument("--mode", type=str, default="dir", help="Choose mode dir or csv")
    ap.add_argument("--nmr", "--nmr_path", dest="nmr", type=str, help="Path to non-matching reference files")
    ap.add_argument("--deg", "--test_path", dest="deg", type=str, help="Path to test files")
    ap.add_argument("--results_path", type=str, default=None)
    ap.add_argument("--device", type=str, default=None)
    ap.add_argument("--weights", type=str, default=None, help="checkpoint path, or 'seeded'")
    ap.add_argument("--precision", type=str, default="fp32", choices=("fp32", "bf16x3", "bf16"),
                    help="bf16x3: split-operand bf16 MFMA, scores within ~1e-6 of fp32 at over twice the speed; "
                         "bf16: fastest, for long recordings (scores within ~5e-4 of fp32)")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = 0
    if world > 1 or "LOCAL_RANK" in os.environ:      # started by torch.distributed.run: one process per GPU
        import torch
        import torch.distributed as dist
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local_rank)
        a.device = f"cuda:{local_rank}"
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        rank = dist.get_rank()
    nomad_avg, _ = Nomad(device=a.device, weights=a.weights, precision=a.precision).predict(a.mode, a.nmr, a.deg, a.results_path)
    if rank == 0:
        print("Nomad average scores, printing top 5 test files")
        print(nomad_avg.head())
    if world > 1 or "LOCAL_RANK" in os.environ:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
