#!/usr/bin/env python3
"""CPU baseline of one triplet fine-tuning step: torch autograd + Adam on the CPU oracle, the reference's shape
(3 x (bs,1,160000)).  Test infrastructure (it imports oracle/): the GPU timing of the same step is tools/bench_train.py.
Usage: python tests/manual/train_step_cpu_baseline.py [--bs 8] [--seconds 10]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nomad_amd.weights import seeded_state_dict
from oracle import nomad_oracle as O


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=10.0)
    a = ap.parse_args()
    n = int(a.seconds * 16000)
    sd = seeded_state_dict(0)
    g = torch.Generator().manual_seed(0)
    A, P, N = [(0.1 * torch.randn(a.bs, n, generator=g)).clamp(-1, 1) for _ in range(3)]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    opt, params = O.make_adam(sd, lr=1e-4)
    live = dict(sd)
    live.update(params)
    t0 = time.perf_counter()
    loss = torch.nn.TripletMarginLoss(margin=0.2)(*(O.triplet_forward(live, w) for w in (A, P, N)))
    opt.zero_grad()
    loss.backward()
    opt.step()
    print(json.dumps({"workload": f"triplet step 3x({a.bs},1,{n})", "cpu_oracle_step_s": time.perf_counter() - t0,
                      "cpu_threads": torch.get_num_threads(), "loss": float(loss)}))


if __name__ == "__main__":
    main()
