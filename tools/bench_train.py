"""Triplet fine-tuning step on the reference's training shape (src/config/train_triplet.yaml: train_bs 8, clips
trimmed to 10 s): three forwards + TripletMarginLoss + backward to every trainable parameter + Adam, timed on the GPU.
Usage: python tools/bench_train.py [--bs 8] [--seconds 10] [--steps 5] [--eval-mode]
(the CPU-autograd timing of the same step lives with the other oracle users: tests/manual/train_step_cpu_baseline.py)"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--eval-mode", action="store_true", help="no dropout / LayerDrop")
    ap.add_argument("--separate", action="store_true", help="three separate forward/backward calls (no merged batch)")
    ap.add_argument("--train-convnet", action="store_true", help="freeze_convnet: False - the conv feature extractor trains too")
    ap.add_argument("--gemm-precision", choices=("fp32", "bf16x3"), default="fp32",
                    help="bf16x3: every GEMM of the step as three bf16 MFMA products over hi / lo halves (Engine.gemm_precision)")
    ap.add_argument("--out", type=str, default="")
    args = ap.parse_args()
    from nomad_amd.train import Training
    from nomad_amd.weights import num_frames, seeded_state_dict
    from nomad_amd.engine import Engine
    n = int(args.seconds * 16000)
    g = torch.Generator().manual_seed(0)
    A, P, N = [(0.1 * torch.randn(args.bs, 1, n, generator=g)).clamp(-1, 1).cuda() for _ in range(3)]
    cfg = dict(experiment_name="bench", checkpoint_path="seeded", margin=0.2, lr=1e-4, lr_decay_factor=0.99,
               gemm_precision=args.gemm_precision)
    reg = dict(dropout=0.0, attention_dropout=0.0, dropout_input=0.0, encoder_layerdrop=0.0) if args.eval_mode else None
    sd = seeded_state_dict(0)
    tr = Training(cfg, engine=Engine(sd, 0), regularisation=reg, merge_branches=not args.separate)
    from nomad_amd.train import ExponentialLR
    tr.margin, tr.lr_scheduler = 0.2, ExponentialLR([1e-5, 1e-4], 0.99)
    eng = tr.engine
    eng.train_set_convnet(args.train_convnet)
    for _ in range(args.warmup):
        loss = tr.train_step(A, P, N)
    torch.cuda.synchronize()
    eng.profile_enable(True)
    eng.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tr.train_step(A, P, N)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    prof = eng.profile_read()
    eng.profile_enable(False)
    T = num_frames(n)
    fwd_flop = 3 * args.bs * (56.925e9 * T / 199.0)  # ~linear in T except the T^2 attention term (small)
    res = {"workload": f"triplet step 3x({args.bs},1,{n}) T={T}", "mode": "eval-arith" if args.eval_mode else "train (dropout+layerdrop)",
           "branches": "separate calls" if args.separate else "merged 3B batch",
           "conv_feature_extractor": "trainable" if args.train_convnet else "frozen", "gemm_precision": args.gemm_precision,
           "ms_per_step": dt * 1e3, "triplets_per_s": args.bs / dt, "loss": loss.item(),
           "approx_model_tflops": 3 * fwd_flop / dt / 1e12,
           "classes_ms_per_step": {k: v["ms"] / args.steps for k, v in prof.items()},
           "classes_launches_per_step": {k: v["launches"] / args.steps for k, v in prof.items()}}
    line = json.dumps(res)
    print(line)
    if args.out:
        open(args.out, "w").write(line + "\n")


if __name__ == "__main__":
    main()
