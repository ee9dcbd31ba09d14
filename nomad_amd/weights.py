"""NOMAD / wav2vec 2.0 BASE parameter sets in the reference's state-dict key layout.

The reference builds ``TripletModel(ssl_model, 768, 256)`` and calls
``load_state_dict(torch.load('pt-models/nomad_best_model.pt'))``
(/root/reference/src/nomad_audio/nomad.py:63-65), so the on-disk artefact is a plain
``{name: tensor}`` dict whose keys are ``ssl_model.<fairseq Wav2Vec2Model key>`` plus
``embedding_layer.1.{weight,bias}`` (nomad.py:219-222).  This module

* lists that key layout with shapes (``expected_shapes``),
* loads a real checkpoint when one is on disk (``load_checkpoint``), and
* generates a deterministic, seeded random parameter set of the same layout
  (``seeded_state_dict``) for benches and parity tests - there is no network in the build
  environment, so the real weights can not be downloaded there.

Nothing in here touches the GPU; the engine repacks these tensors at ``nomad_create`` time.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from typing import Dict, Optional

import torch

# wav2vec 2.0 BASE architecture constants (fairseq ``wav2vec_small.pt`` config)
CONV_LAYERS = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2
CONV_DIM = 512
EMBED_DIM = 768
FFN_DIM = 3072
NUM_LAYERS = 12
NUM_HEADS = 12
HEAD_DIM = 64
POS_CONV_K = 128
POS_CONV_GROUPS = 16
EMB_DIM = 256


def num_frames(n_samples: int) -> int:
    """Number of encoder frames T for a waveform of ``n_samples`` (conv stack output length)."""
    length = n_samples
    for _, k, s in CONV_LAYERS:
        length = (length - k) // s + 1
    return length


def conv_lengths(n_samples: int):
    """Output length of each of the 7 conv layers."""
    out = []
    length = n_samples
    for _, k, s in CONV_LAYERS:
        length = (length - k) // s + 1
        out.append(length)
    return out


def expected_shapes() -> "OrderedDict[str, tuple]":
    """Key -> shape of the NOMAD checkpoint."""
    d: "OrderedDict[str, tuple]" = OrderedDict()
    p = "ssl_model."
    d[p + "mask_emb"] = (EMBED_DIM,)
    cin = 1
    for i, (cout, k, _) in enumerate(CONV_LAYERS):
        d[p + f"feature_extractor.conv_layers.{i}.0.weight"] = (cout, cin, k)
        cin = cout
    d[p + "feature_extractor.conv_layers.0.2.weight"] = (CONV_DIM,)
    d[p + "feature_extractor.conv_layers.0.2.bias"] = (CONV_DIM,)
    d[p + "layer_norm.weight"] = (CONV_DIM,)
    d[p + "layer_norm.bias"] = (CONV_DIM,)
    d[p + "post_extract_proj.weight"] = (EMBED_DIM, CONV_DIM)
    d[p + "post_extract_proj.bias"] = (EMBED_DIM,)
    d[p + "encoder.pos_conv.0.bias"] = (EMBED_DIM,)
    d[p + "encoder.pos_conv.0.weight_g"] = (1, 1, POS_CONV_K)
    d[p + "encoder.pos_conv.0.weight_v"] = (EMBED_DIM, EMBED_DIM // POS_CONV_GROUPS, POS_CONV_K)
    for l in range(NUM_LAYERS):
        q = p + f"encoder.layers.{l}."
        for name in ("k_proj", "v_proj", "q_proj", "out_proj"):
            d[q + f"self_attn.{name}.weight"] = (EMBED_DIM, EMBED_DIM)
            d[q + f"self_attn.{name}.bias"] = (EMBED_DIM,)
        d[q + "self_attn_layer_norm.weight"] = (EMBED_DIM,)
        d[q + "self_attn_layer_norm.bias"] = (EMBED_DIM,)
        d[q + "fc1.weight"] = (FFN_DIM, EMBED_DIM)
        d[q + "fc1.bias"] = (FFN_DIM,)
        d[q + "fc2.weight"] = (EMBED_DIM, FFN_DIM)
        d[q + "fc2.bias"] = (EMBED_DIM,)
        d[q + "final_layer_norm.weight"] = (EMBED_DIM,)
        d[q + "final_layer_norm.bias"] = (EMBED_DIM,)
    d[p + "encoder.layer_norm.weight"] = (EMBED_DIM,)
    d[p + "encoder.layer_norm.bias"] = (EMBED_DIM,)
    d["embedding_layer.1.weight"] = (EMB_DIM, EMBED_DIM)
    d["embedding_layer.1.bias"] = (EMB_DIM,)
    return d


def seeded_state_dict(seed: int = 0, qk_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Deterministic random parameters in the checkpoint layout (fp32, CPU).

    Every tensor is random (never zeros / ones exactly) so that each affine path and bias is
    exercised.  ``qk_gain`` scales the q/k projection weights to make the attention softmax
    peaky in tests (gain 1 gives near-uniform attention with 0.02-std weights).
    """
    g = torch.Generator().manual_seed(seed)

    def randn(*shape, std=1.0):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * std

    sd: Dict[str, torch.Tensor] = OrderedDict()
    for key, shape in expected_shapes().items():
        leaf = key.split(".")[-1]
        if key.endswith("mask_emb"):
            t = torch.rand(*shape, generator=g, dtype=torch.float32)
        elif "conv_layers" in key and key.endswith(".0.weight"):
            fan_in = shape[1] * shape[2]
            t = randn(*shape, std=math.sqrt(2.0 / fan_in))
        elif "pos_conv.0.weight_v" in key:
            t = randn(*shape, std=math.sqrt(4.0 / (POS_CONV_K * EMBED_DIM)))
        elif "pos_conv.0.weight_g" in key:
            t = None  # filled below from weight_v
        elif "layer_norm" in key or "conv_layers.0.2" in key:
            t = 1.0 + randn(*shape, std=0.1) if leaf == "weight" else randn(*shape, std=0.1)
        elif key.startswith("embedding_layer"):
            bound = 1.0 / math.sqrt(EMBED_DIM)
            t = (torch.rand(*shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
        elif leaf == "weight":
            std = 0.02 * (qk_gain if (".q_proj." in key or ".k_proj." in key) else 1.0)
            t = randn(*shape, std=std)
        else:  # biases
            t = randn(*shape, std=0.02)
        sd[key] = t
    v = sd["ssl_model.encoder.pos_conv.0.weight_v"]
    norm = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()  # weight_norm(dim=2): norm over dims (0,1)
    sd["ssl_model.encoder.pos_conv.0.weight_g"] = norm * (1.0 + randn(1, 1, POS_CONV_K, std=0.1))
    return sd


def check_state_dict(sd: Dict[str, torch.Tensor]) -> None:
    """Raise ``KeyError`` / ``ValueError`` if ``sd`` is not a complete NOMAD parameter set."""
    for key, shape in expected_shapes().items():
        if key.endswith("mask_emb"):
            continue  # unused at inference (mask=False, nomad.py:226)
        if key not in sd:
            raise KeyError(f"NOMAD checkpoint is missing parameter {key!r}")
        if tuple(sd[key].shape) != tuple(shape):
            raise ValueError(f"{key}: shape {tuple(sd[key].shape)} != expected {shape}")


def load_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    """Load ``nomad_best_model.pt`` (a torch zip-pickle state dict, nomad.py:65) as fp32 CPU tensors."""
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    # torch>=2 weight_norm parametrisation names -> the fairseq-era names
    ren = {
        "ssl_model.encoder.pos_conv.0.parametrizations.weight.original0": "ssl_model.encoder.pos_conv.0.weight_g",
        "ssl_model.encoder.pos_conv.0.parametrizations.weight.original1": "ssl_model.encoder.pos_conv.0.weight_v",
    }
    out = OrderedDict()
    for k, v in sd.items():
        out[ren.get(k, k)] = v.detach().to(torch.float32).contiguous()
    check_state_dict(out)
    return out


def find_checkpoint() -> Optional[str]:
    """Where the reference keeps its weights: ``./pt-models/nomad_best_model.pt`` (nomad.py:28)."""
    for cand in (os.environ.get("NOMAD_CHECKPOINT"), os.path.join("pt-models", "nomad_best_model.pt")):
        if cand and os.path.isfile(cand):
            return cand
    return None


# fairseq wav2vec 2.0 BASE pre-training config (the cfg stored inside wav2vec_small.pt, which
# load_model_ensemble_and_task keeps when the reference builds its model, nomad.py:58): the gradient entering the
# conv feature extractor is scaled by this (GradMultiply on the extractor output, Wav2Vec2Model.forward)
W2V_BASE_FEATURE_GRAD_MULT = 0.1


def feature_grad_mult_of(obj) -> Optional[float]:
    """``feature_grad_mult`` of a loaded fairseq checkpoint object (``{'cfg': {'model': {...}}}`` in fairseq >= 0.10.2,
    ``{'args': Namespace}`` before), or None when the object carries no model config (a plain state dict)."""
    if not isinstance(obj, dict):
        return None
    cfg = obj.get("cfg")
    model = None
    if cfg is not None:
        model = cfg.get("model") if hasattr(cfg, "get") else getattr(cfg, "model", None)
    if model is None:
        model = obj.get("args")
    if model is None:
        return None
    v = model.get("feature_grad_mult") if hasattr(model, "get") else getattr(model, "feature_grad_mult", None)
    return None if v is None else float(v)


def find_feature_grad_mult(checkpoint: Optional[str] = None) -> float:
    """The value the reference's model would carry.  The wav2vec 2.0 BASE constant 0.1 unless the caller names a fairseq
    checkpoint EXPLICITLY (argument, or ``$NOMAD_W2V_CHECKPOINT``): that file is then read with ``weights_only=True``
    (no code in the pickle is ever executed) and its ``feature_grad_mult`` used when the restricted unpickler can
    reach it - fairseq checkpoints that pickle omegaconf / argparse objects cannot be read that way and give 0.1 too.
    Nothing is read from the working directory implicitly: constructing ``Nomad()`` never unpickles a file the caller
    did not name."""
    cand = checkpoint or os.environ.get("NOMAD_W2V_CHECKPOINT")
    if cand and os.path.isfile(cand):
        try:
            v = feature_grad_mult_of(torch.load(cand, map_location="cpu", weights_only=True))
        except Exception:  # classes the restricted unpickler refuses: fall back to the published BASE value
            v = None
        if v is not None:
            return v
    return W2V_BASE_FEATURE_GRAD_MULT
