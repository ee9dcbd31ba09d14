"""Triplet fine-tuning on the HIP engine: the host-side mirror of the reference's training script
(/root/reference/src/training/train_triplet.py, config /root/reference/src/config/train_triplet.yaml,
data set /root/reference/src/dataloader/triplet_dataloader.py).

Same names, same config keys, same control flow:

    train_obj = Training("src/config/train_triplet.yaml")     # train_triplet.py:44-110
    train_obj.training_loop()                                  # :161-205

Every FLOP of a step - three forwards, nn.TripletMarginLoss, backward to every trainable parameter, Adam - runs in
libnomad_hip.so (nomad_embed_train / nomad_triplet_loss / nomad_train_backward / nomad_train_adam_step); this file
is data loading, the epoch loop, the learning-rate schedule and checkpoint writing.

Differences, on purpose:
* none in the freeze switches: ``freeze_convnet: True`` (the shipped config; backbone at 1e-5, head at ``lr``),
  ``freeze_convnet: False`` (the conv feature extractor trains too, and - as in train_triplet.py:96 - ONE Adam group at
  ``lr`` for every parameter) and ``freeze_all: True`` are all supported.
* ``checkpoint_path`` may be a NOMAD-layout state dict (keys of nomad_best_model.pt), a fairseq ``wav2vec_small.pt``
  ({'model': state_dict}; the head is then initialised like ``nn.Linear`` under ``torch.manual_seed(0)``), or the word
  ``seeded`` (random weights, for tests).  fairseq itself is not needed.
* model.train() regularisation (fairseq BASE config: dropout 0.1, attention_dropout 0.1, dropout_input 0.1,
  encoder_layerdrop 0.05) uses the engine's counter-based masks; torch's RNG stream of the reference's device cannot
  be reproduced on any other device, so runs are statistically, not bit-wise, equivalent to the reference's.
"""
from __future__ import annotations

import os
import random
from datetime import datetime
from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from . import wavio
from .engine import Engine
from .weights import EMB_DIM, EMBED_DIM, check_state_dict, expected_shapes, load_checkpoint, seeded_state_dict

SEED = 0  # train_triplet.py:29-33

# fairseq wav2vec 2.0 BASE pre-training config (what wav2vec_small.pt carries in its cfg)
W2V_BASE_REGULARISATION = dict(dropout=0.1, attention_dropout=0.1, dropout_input=0.1, encoder_layerdrop=0.05)


def load_processing(filepath, target_sr: int = 16000, trim: bool = False) -> torch.Tensor:
    """triplet_dataloader.py:8-29: load, mono mix, resample to 16 kHz, optionally trim to 10 s.  -> (1, N) fp32."""
    return torch.from_numpy(wavio.load_processing(filepath, target_sr, trim))


class TripletDataset(torch.utils.data.Dataset):
    """triplet_dataloader.py:31-83: csv with Anchor / Positive / Negative (and db) columns under ``root``."""

    def __init__(self, config, data_mode="train_df", level=None):
        super().__init__()
        import pandas as pd
        self.config = config
        self.root = self.config["root"]
        self.dataset = pd.read_csv(self.config[data_mode])
        if level is not None:
            self.dataset = self.dataset[self.dataset["db"].isin(level)]
        self.dataset = self.dataset.drop_duplicates()

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, index):
        row = self.dataset.iloc[index]
        trim = self.config["trim"]
        return tuple(load_processing(os.path.join(self.root + row[c]), trim=trim) for c in ("Anchor", "Positive", "Negative"))

    def collate_fn(self, batch):  # zero padding at batch level
        A, P, N = zip(*batch)
        return self.zero_pad_wav(A), self.zero_pad_wav(P), self.zero_pad_wav(N)

    @staticmethod
    def zero_pad_wav(wavs):
        max_len = max(w.shape[1] for w in wavs)
        return torch.stack([torch.nn.functional.pad(w, (0, max_len - w.shape[1]), "constant", 0) for w in wavs], dim=0)


def load_pretrained(path: str, allow_unsafe_pickle: bool = None) -> Dict[str, torch.Tensor]:
    """State dict in the NOMAD checkpoint layout from ``checkpoint_path`` (see module docstring).

    The file is read with the tensors-only unpickler.  A fairseq checkpoint that pickles its config objects needs the full
    unpickler - which executes whatever the file contains - so that is used only when the caller asks for it
    (``allow_unsafe_pickle=True``, config key ``allow_unsafe_pickle``, or ``NOMAD_ALLOW_UNSAFE_PICKLE=1``); otherwise the
    restricted loader's error is re-raised with that hint.  (The reference always uses the full one: nomad.py:58.)"""
    if path == "seeded":
        return seeded_state_dict(0)
    if allow_unsafe_pickle is None:
        allow_unsafe_pickle = os.environ.get("NOMAD_ALLOW_UNSAFE_PICKLE", "0") == "1"
    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:
        if not allow_unsafe_pickle:
            raise RuntimeError(f"{path}: not loadable with torch.load(weights_only=True) ({type(e).__name__}: {str(e)[:200]}).  If this is a "
                               "trusted fairseq checkpoint with pickled config objects, pass allow_unsafe_pickle=True (config key "
                               "'allow_unsafe_pickle', or NOMAD_ALLOW_UNSAFE_PICKLE=1) to load it with the full unpickler.") from e
        obj = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(obj, dict) and "model" in obj and isinstance(obj["model"], dict):  # fairseq checkpoint
        want = expected_shapes()
        sd = {}
        for k, v in obj["model"].items():
            k2 = "ssl_model." + k
            if k2 in want:
                sd[k2] = v.detach().to(torch.float32).contiguous()
        g = torch.Generator().manual_seed(SEED)  # nn.Linear default init (kaiming_uniform a=sqrt(5)): U(-1/sqrt(in), +)
        bound = 1.0 / np.sqrt(EMBED_DIM)
        sd["embedding_layer.1.weight"] = (torch.rand(EMB_DIM, EMBED_DIM, generator=g) * 2 - 1) * bound
        sd["embedding_layer.1.bias"] = (torch.rand(EMB_DIM, generator=g) * 2 - 1) * bound
        sd.setdefault("ssl_model.mask_emb", torch.zeros(EMBED_DIM))
        check_state_dict(sd)
        return sd
    return load_checkpoint(path)


def allreduce_mean_gradients(engine, group=None) -> None:
    """Data-parallel fine-tuning (not in the reference, which is single-GPU): average the flat gradient vector over
    the ranks - ONE all-reduce of 85 M floats (RCCL over xGMI; gloo in the CPU test) between backward and Adam.
    One process per GPU; each rank's DataLoader must yield its own shard (DistributedSampler)."""
    world = dist.get_world_size(group)
    if world == 1:
        return
    g = engine.train_read(1)
    dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
    g.div_(world)
    engine.train_write(1, g)


class ExponentialLR:
    """torch.optim.lr_scheduler.ExponentialLR over the two learning rates (train_triplet.py:110)."""

    def __init__(self, lrs: Sequence[float], gamma: float):
        self.lrs, self.gamma = list(lrs), gamma

    def step(self):
        self.lrs = [lr * self.gamma for lr in self.lrs]

    def get_last_lr(self):
        return list(self.lrs)


class Training:
    def __init__(self, config_file, device: int = 0, engine: Optional[Engine] = None,
                 regularisation: Optional[dict] = None, merge_branches: bool = True, group=None):
        import yaml
        if isinstance(config_file, dict):
            self.config = dict(config_file)
        else:
            with open(config_file) as file:
                self.config = yaml.load(file, Loader=yaml.FullLoader)
        if not torch.cuda.is_available():
            raise RuntimeError("nomad_amd.train needs an MI355X: the engine has no CPU path")
        self.DEVICE = torch.device("cuda", device)
        print(f"Device: {self.DEVICE}")
        random.seed(SEED)
        np.random.seed(SEED)
        torch.manual_seed(SEED)
        if self.config.get("eval_w2v"):
            raise NotImplementedError("eval_w2v (raw wav2vec features) is outside the NOMAD hot path")
        self.engine = engine if engine is not None else Engine(
            load_pretrained(self.config["checkpoint_path"], self.config.get("allow_unsafe_pickle")), device)
        # gemm_precision (not a reference key): "bf16x3" forms the products of every GEMM of the step - forward, dX and the
        # split-K dW - as three bf16 MFMA products over hi / lo halves, accumulated in fp32 (Engine.gemm_precision).  Only set
        # when the config names it: an engine handed in by the caller (shared with a Nomad(precision="bf16x3"), say) keeps its mode
        if "gemm_precision" in self.config:
            self.engine.gemm_precision = self.config["gemm_precision"]
        self.engine.train_enable()
        # freeze_all (train_triplet.py:76-79): feature extractor and encoder frozen; what is left trainable is
        # post_extract_proj, the feature LayerNorm and the embedding layer
        training = self.config["experiment_name"] == "Training"
        self.engine.train_set_frozen(bool(training and self.config.get("freeze_all")))
        # freeze_convnet: False (train_triplet.py:71-73): the conv feature extractor gets gradients too (unless freeze_all
        # froze it again, :76-78)
        self.train_convnet = bool(training and not self.config.get("freeze_convnet", True) and not self.config.get("freeze_all"))
        self.engine.train_set_convnet(self.train_convnet)
        self.reg = dict(W2V_BASE_REGULARISATION)
        self.reg.update(regularisation or {})
        self._rng = np.random.RandomState(SEED)  # LayerDrop draws + per-call dropout seeds
        self.merge_branches = merge_branches     # A/P/N as one 3B-clip launch sequence when their padded lengths agree
        self.group = group                       # torch.distributed group for data-parallel training (None: default group)
        if self.config["experiment_name"] == "Training":
            self.current_level = self.config.get("current_level")
            g = torch.Generator()
            g.manual_seed(SEED)
            self.train_set = TripletDataset(self.config, data_mode="train_df", level=self.current_level)
            self.train_sampler = None
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
                self.train_sampler = torch.utils.data.distributed.DistributedSampler(
                    self.train_set, num_replicas=dist.get_world_size(group), rank=dist.get_rank(group), shuffle=True, seed=SEED)
            self.train_loader = torch.utils.data.DataLoader(
                self.train_set, batch_size=self.config["train_bs"], shuffle=self.train_sampler is None,
                sampler=self.train_sampler, num_workers=self.config["num_workers"],
                collate_fn=self.train_set.collate_fn, generator=g, pin_memory=True)
            self.valid_set = TripletDataset(self.config, data_mode="valid_df", level=self.current_level)
            self.valid_loader = torch.utils.data.DataLoader(
                self.valid_set, batch_size=self.config["val_bs"], shuffle=False, num_workers=self.config["num_workers"],
                collate_fn=self.valid_set.collate_fn, pin_memory=True)
            self.margin = float(self.config["margin"])
            # train_triplet.py:96-107: Adam; with freeze_convnet the pretrained parameters at 1e-5 and embedding_layer at
            # `lr`, otherwise the optimiser is not overwritten and every parameter runs at `lr`
            lr = float(self.config["lr"])
            body_lr = 1e-5 if self.config.get("freeze_convnet", True) else lr
            self.lr_scheduler = ExponentialLR([body_lr, lr], float(self.config["lr_decay_factor"]))

    # ---- one optimisation step (train_triplet.py:117-131) ---------------------------------------------
    def _draw(self) -> dict:
        """model.train() randomness of ONE forward call: LayerDrop mask (np.random.random() > layerdrop keeps the
        layer, fairseq TransformerEncoder.extract_features) and the seed of its dropout masks."""
        mask = 0
        for l in range(12):
            if self.reg["encoder_layerdrop"] <= 0 or self._rng.random_sample() > self.reg["encoder_layerdrop"]:
                mask |= 1 << l
        return dict(dropout=self.reg["dropout"], attention_dropout=self.reg["attention_dropout"],
                    dropout_input=self.reg["dropout_input"], seed=int(self._rng.randint(0, 2 ** 62)), layer_mask=mask)

    def train_step(self, A: torch.Tensor, P: torch.Tensor, N: torch.Tensor, training: bool = True) -> torch.Tensor:
        """A_embs = model(A); P_embs = model(P); N_embs = model(N); loss = criterion(...); zero_grad; backward; step.
        Returns the loss as a 1-element device tensor (no host sync here)."""
        eng = self.engine
        wavs = [w.to(self.DEVICE, torch.float32, non_blocking=True).squeeze(1).contiguous() for w in (A, P, N)]
        if not training:
            # model.eval(): clips are independent and the engine is batch invariant bit for bit, so the three
            # forwards of train_triplet.py:146-148 run as ONE batch of 3B clips when the lengths agree
            if wavs[0].shape == wavs[1].shape == wavs[2].shape:
                B = wavs[0].shape[0]
                e = eng.embed(torch.cat(wavs, dim=0))
                embs = [e[:B], e[B:2 * B], e[2 * B:]]
            else:
                embs = [eng.embed(w) for w in wavs]
            return eng.triplet_loss(embs[0].contiguous(), embs[1].contiguous(), embs[2].contiguous(), self.margin,
                                    want_grad=False)[0]
        draws = [self._draw() for _ in wavs]
        if self.merge_branches and wavs[0].shape == wavs[1].shape == wavs[2].shape:
            # The three forwards as ONE batch of 3B clips: every clip goes through the same arithmetic as in its own
            # call, LayerDrop stays per branch (the engine runs a layer per branch where the draws disagree), and
            # the dropout masks of the three branches are disjoint slices of one counter-based stream.
            B = wavs[0].shape[0]
            d = dict(draws[0], layer_mask=0xFFF)
            eng.train_set_stochastic(**d)
            eng.train_set_branches([x["layer_mask"] for x in draws])
            w = torch.cat(wavs, dim=0)
            emb, layers, saved = eng.embed_train(w)
            loss, da, dp, dn = eng.triplet_loss(emb[:B].contiguous(), emb[B:2 * B].contiguous(), emb[2 * B:].contiguous(),
                                                self.margin)
            eng.train_zero_grad()
            eng.train_backward(w, layers, saved, torch.cat([da, dp, dn], dim=0))
            eng.train_set_branches(None)
        else:
            outs = []
            for w, d in zip(wavs, draws):
                eng.train_set_stochastic(**d)
                outs.append(eng.embed_train(w))
            loss, da, dp, dn = eng.triplet_loss(outs[0][0], outs[1][0], outs[2][0], self.margin)
            eng.train_zero_grad()
            for w, (_, layers, saved), g, d in zip(wavs, outs, (da, dp, dn), draws):
                eng.train_set_stochastic(**d)
                eng.train_backward(w, layers, saved, g)
        eng.train_set_stochastic()
        if self.group is not None or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            allreduce_mean_gradients(eng, self.group)  # data parallel: every rank saw its own triplets
        lr_body, lr_head = self.lr_scheduler.get_last_lr()
        eng.adam_step(lr_body, lr_head)
        return loss

    def train(self, model=None, dataloader=None, optimizer=None, criterion=None) -> float:
        dataloader = dataloader if dataloader is not None else self.train_loader
        total = torch.zeros(1, device=self.DEVICE)
        for A, P, N in dataloader:
            total += self.train_step(A, P, N, training=True)
        return total.item() / max(len(dataloader), 1)

    def eval(self, model=None, dataloader=None, criterion=None) -> float:
        dataloader = dataloader if dataloader is not None else self.valid_loader
        total = torch.zeros(1, device=self.DEVICE)
        for A, P, N in dataloader:
            total += self.train_step(A, P, N, training=False)
        return total.item() / max(len(dataloader), 1)

    def save(self, path: str):
        torch.save(self.engine.train_state_dict(), path)

    # ---- train_triplet.py:161-205 --------------------------------------------------------------------------
    def training_loop(self):
        import yaml
        dt_string = datetime.now().strftime("%d-%m-%Y_%H-%M-%S")
        self.PATH_DIR = os.path.join("out-models", self.config["out_dir"], dt_string)
        os.makedirs(self.PATH_DIR, exist_ok=True)
        with open(os.path.join(self.PATH_DIR, "config.yaml"), "w") as file:
            yaml.dump(self.config, file)
        best_valid_loss = np.inf
        counter = 0
        rank0 = not (dist.is_available() and dist.is_initialized()) or dist.get_rank(self.group) == 0
        for i in range(self.config["num_epochs"]):
            if getattr(self, "train_sampler", None) is not None:
                self.train_sampler.set_epoch(i)
            train_loss = self.train()
            valid_loss = self.eval()  # every rank evaluates the whole validation set: identical numbers, no exchange
            if valid_loss < best_valid_loss:
                if rank0:
                    self.save(os.path.join(self.PATH_DIR, "best_model.pt"))
                best_valid_loss = valid_loss
                print("Saved Weights Success")
                counter = 0
            else:
                counter += 1
            if (counter + 1) % self.config["lr_decay_step"] == 0:
                self.lr_scheduler.step()
            print(f"COUNTER:  {counter}/{self.config['patience']}")
            print(f"LR: {self.lr_scheduler.get_last_lr()}")
            if counter > self.config["patience"]:
                print("Stop training, counter greater than patience")
                break
            print(f"EPOCHS: {i + 1} train_loss : {train_loss}")
            print(f"EPOCHS: {i + 1} valid_loss : {valid_loss}")
            print("\n")
        return best_valid_loss


def main(argv=None):
    """``python -m nomad_amd.train --config_file cfg.yaml``: the 'Training' branch of /root/reference/main.py:7-46
    (the evaluation experiments - quality_nmr, valid_rank, intensity, quality_fr - are plotting / statistics scripts
    around ``predict`` and are not part of this build)."""
    import argparse
    import yaml
    ap = argparse.ArgumentParser(prog="python -m nomad_amd.train")
    ap.add_argument("--config_file", type=str, required=True)
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)
    with open(args.config_file) as file:
        config = yaml.load(file, Loader=yaml.FullLoader)
    if config["experiment_name"] != "Training":
        raise SystemExit(f"experiment_name {config['experiment_name']!r}: only 'Training' is implemented here")
    Training(args.config_file, device=args.device).training_loop()


if __name__ == "__main__":
    main()
