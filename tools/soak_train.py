#!/usr/bin/env python3
"""Determinism soak of the fine-tuning step: two runs of N optimisation steps (dropout + LayerDrop on, merged branches)
from the same initial state with the same host seeds must end in bit-identical parameters and Adam moments."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.train import ExponentialLR, Training
from nomad_amd.weights import seeded_state_dict
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
g = torch.Generator().manual_seed(0)
batches = [[(0.1 * torch.randn(4, 1, 48000, generator=g)).clamp(-1, 1) for _ in range(3)] for _ in range(3)]
finals = []
for run in range(2):
    tr = Training(dict(experiment_name="soak", checkpoint_path="seeded", margin=0.2), engine=Engine(seeded_state_dict(0), 0))
    tr.margin, tr.lr_scheduler = 0.2, ExponentialLR([1e-5, 1e-4], 0.99)
    losses = []
    for i in range(n):
        A, P, N = batches[i % 3]
        losses.append(tr.train_step(A, P, N).item())
    finals.append((tr.engine.train_read(0).clone(), tr.engine.train_read(2).clone(), tr.engine.train_read(3).clone(), losses))
    tr.engine.close()
same = all(torch.equal(a, b) for a, b in zip(finals[0][:3], finals[1][:3])) and finals[0][3] == finals[1][3]
print(f"soak_train: {n} steps x 2 runs, losses {['%.4f' % l for l in finals[0][3][:4]]}..., bit-identical = {same}")
sys.exit(0 if same else 1)
