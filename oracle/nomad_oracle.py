"""CPU oracle for the NOMAD scoring hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product (``nomad_amd``) never imports it and has no CPU fallback.

What it restates (plain PyTorch fp32 on CPU; numpy float64 for the distance stage):

* ``backbone``      - fairseq ``Wav2Vec2Model.forward(wav, mask=False, features_only=True)`` as
  called at /root/reference/src/nomad_audio/nomad.py:226 and :245.  fairseq (``>=0.12.2``,
  un-pinned above, /root/reference/requirements.txt:4) is a third-party dependency that is absent
  from /root/reference and from the build image; the algorithm restated here is the published
  wav2vec 2.0 BASE architecture (``wav2vec_small.pt`` config: conv_feature_layers
  [(512,10,5)]+[(512,3,2)]*4+[(512,2,2)]*2, conv_bias=False, extractor_mode=default (GroupNorm on
  layer 0), 768-d / 12 layers / 12 heads / FFN 3072, post-LN (layer_norm_first=False),
  conv_pos=128, conv_pos_groups=16, GELU).  SURVEY.md section 3.2 is the op-by-op spec.
* ``triplet_forward`` - ``TripletModel.forward`` (nomad.py:224-231).
* ``lossnet_forward`` - ``LossNetLayers.forward`` (nomad.py:243-258).
* ``nomad_loss``      - ``NomadLoss.forward`` (nomad.py:267-282).
* ``pairwise``        - ``scipy.spatial.distance.cdist`` + ``np.mean(axis=1)`` (nomad.py:108-111).
* ``load_wav``        - ``Nomad.load_processing`` for PCM wav at 16 kHz (nomad.py:192-212).

PINNING STATUS (round 4).  The reference holds no programmatic golden vectors for this path; its only pinned values are the
3-decimal README tables (README.md:69-81), which need the real ``nomad_best_model.pt`` (downloaded at import time by the
reference; not available offline), and ``nomad.py`` cannot be imported here (``import fairseq`` / ``torchaudio`` + two downloads at
module level).  What the oracle IS pinned to, by fixtures a committed script (``oracle/make_golden.py``) generated in the build
container and ``tests/test_oracle.py`` / ``tests/test_reference_classes.py`` check:

* ``head`` / ``triplet_forward`` (squeeze, time mean, ReLU, Linear, normalize): the reference's own Python -
  ``src/models/networks.py`` imported as a file (``TripletModel``, ``Origw2v``; ``tests/golden/ref_networks.npz``) and the
  ``TripletModel`` of ``nomad.py`` itself (below); also run live over this module's backbone, bit-equal.
* ``lossnet_forward`` / ``nomad_loss`` / the gradient of ``Nomad.forward``: the reference's own ``LossNetLayers``, ``NomadLoss``
  and ``Nomad.forward`` - the ClassDef nodes of ``nomad.py`` compiled WITHOUT importing the module, run over the HF backbone
  (``tests/golden/ref_nomad_classes.npz``: 13 outputs, loss, d loss / d estimate at feature_grad_mult 1.0 and 0.1).
* ``pairwise`` and the result tables: SciPy ``cdist`` + ``np.mean`` (what nomad.py:108-111 calls), and the reference's own
  ``Nomad.predict`` / ``get_embeddings`` / ``get_embeddings_csv`` executed the same way - DataFrames, the BYTES of both CSV files in
  dir and csv mode, the default ``results-csv/<timestamp>/`` paths, every exception message (same fixture).  Only
  ``load_processing`` (torchaudio) is replaced there, by a PCM-16 reader returning what ``torchaudio.load`` returns.
* ``backbone`` (fairseq ``Wav2Vec2Model``, >99.9 % of the arithmetic): HuggingFace ``transformers`` ``Wav2Vec2Model`` - an
  independent implementation of the same published architecture - with identical seeded weights through the fairseq->HF key
  map, plus the two fairseq-only semantics restated from its published source (``GradMultiply`` 0.1, pad-to-multiple-of-2).
  **Parity against fairseq itself is unpinned**: fairseq is in neither /root/reference nor the image.
* Parity against the README table is **unpinned** until real weights are available (slot: ``tests/test_readme_table.py``).
"""
from __future__ import annotations

import struct
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

CONV_STRIDES = (5, 2, 2, 2, 2, 2, 2)
NUM_LAYERS = 12
NUM_HEADS = 12
P = "ssl_model."


def fold_pos_conv_weight(sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """weight_norm(dim=2) of fairseq's pos_conv: w = v * (g / ||v||_(0,1)), g of shape (1,1,K)."""
    v = sd[P + "encoder.pos_conv.0.weight_v"]
    g = sd[P + "encoder.pos_conv.0.weight_g"]
    norm = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    return v * (g / norm)


def feature_extractor(sd: Dict[str, torch.Tensor], wav: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
    """7-layer conv feature extractor -> (B, T, 512) (before layer_norm).  wav: (B, N) raw amplitude."""
    x = wav[:, None, :]
    for i, stride in enumerate(CONV_STRIDES):
        w = sd[P + f"feature_extractor.conv_layers.{i}.0.weight"]
        x = F.conv1d(x, w, stride=stride)
        if i == 0:  # Fp32GroupNorm(512 groups, 512 channels), affine, eps 1e-5
            x = F.group_norm(x, 512, sd[P + "feature_extractor.conv_layers.0.2.weight"],
                             sd[P + "feature_extractor.conv_layers.0.2.bias"], eps=1e-5)
        x = F.gelu(x)
        if taps is not None:
            taps[f"conv{i}"] = x.transpose(1, 2).contiguous()
    return x.transpose(1, 2)


def encoder_layer(sd: Dict[str, torch.Tensor], l: int, x: torch.Tensor, taps: Optional[dict] = None,
                  stoch: Optional["Stochastic"] = None, key_padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One post-LN TransformerSentenceEncoderLayer (layer_norm_first=False). x: (B,T,768).
    stoch=None is eval mode; otherwise fairseq's train-mode dropouts with the engine's counter-based masks.
    key_padding_mask (B,T) bool, True = padded key: fairseq MultiheadAttention fills those score columns with -inf
    before the softmax (``attn_weights.masked_fill(key_padding_mask[:, None, None, :], -inf)``)."""
    q_ = P + f"encoder.layers.{l}."
    B, T, C = x.shape
    hd = C // NUM_HEADS
    q = F.linear(x, sd[q_ + "self_attn.q_proj.weight"], sd[q_ + "self_attn.q_proj.bias"]) * (hd ** -0.5)
    k = F.linear(x, sd[q_ + "self_attn.k_proj.weight"], sd[q_ + "self_attn.k_proj.bias"])
    v = F.linear(x, sd[q_ + "self_attn.v_proj.weight"], sd[q_ + "self_attn.v_proj.bias"])
    q = q.view(B, T, NUM_HEADS, hd).transpose(1, 2)
    k = k.view(B, T, NUM_HEADS, hd).transpose(1, 2)
    v = v.view(B, T, NUM_HEADS, hd).transpose(1, 2)
    scores = q @ k.transpose(-1, -2)
    if key_padding_mask is not None:
        scores = scores.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    probs = torch.softmax(scores, dim=-1)
    if stoch is not None:
        probs = probs * stoch.mult(2 + 3 * l, probs.shape, stoch.attention_dropout)
    a = probs @ v
    a = a.transpose(1, 2).reshape(B, T, C)
    if taps is not None and l == 0:
        taps["attn0"] = a
    a = F.linear(a, sd[q_ + "self_attn.out_proj.weight"], sd[q_ + "self_attn.out_proj.bias"])
    if stoch is not None:
        a = a * stoch.mult(3 + 3 * l, a.shape, stoch.dropout)
    x = F.layer_norm(x + a, (C,), sd[q_ + "self_attn_layer_norm.weight"], sd[q_ + "self_attn_layer_norm.bias"], 1e-5)
    h = F.gelu(F.linear(x, sd[q_ + "fc1.weight"], sd[q_ + "fc1.bias"]))
    h = F.linear(h, sd[q_ + "fc2.weight"], sd[q_ + "fc2.bias"])
    if stoch is not None:
        h = h * stoch.mult(4 + 3 * l, h.shape, stoch.dropout)
    x = F.layer_norm(x + h, (C,), sd[q_ + "final_layer_norm.weight"], sd[q_ + "final_layer_norm.bias"], 1e-5)
    return x


class GradMultiply(torch.autograd.Function):
    """fairseq/modules/grad_multiply.py: identity in the forward, gradient times ``scale`` in the backward."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return x.new(x)

    @staticmethod
    def backward(ctx, grad):
        return grad * ctx.scale, None


def backbone(sd: Dict[str, torch.Tensor], wav: torch.Tensor, taps: Optional[dict] = None,
             stoch: Optional["Stochastic"] = None, feature_grad_mult: float = 1.0,
             required_seq_len_multiple: int = 1) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """wav (B,N) -> (x (B,T,768), [12 layer outputs (B,T,768)]).

    The reference gets ``layer_results`` as (T,B,768) tuples and permutes them to (B,T,768)
    (nomad.py:248); the list returned here is already in that (B,T,768) form.

    feature_grad_mult: fairseq ``Wav2Vec2Model.forward`` - ``features = self.feature_extractor(source)`` then, if
    ``feature_grad_mult != 1.0``, ``features = GradMultiply.apply(features, self.feature_grad_mult)`` (train and eval
    alike; with 0 the extractor runs under ``torch.no_grad()``).  wav2vec 2.0 BASE (wav2vec_small.pt): 0.1.  The
    forward values do not depend on it; 1.0 (default here) is the plain chain rule.

    required_seq_len_multiple: fairseq >= 0.12 ``TransformerEncoder.extract_features`` pads the frame axis with zero
    frames to a multiple of this (config default 2) AFTER the positional conv and encoder LayerNorm, passes a
    key-padding mask over the pad frames to every layer, and strips the pad frames from ``x`` and from every entry of
    ``layer_results`` afterwards.  Real frames never attend to a pad frame (their score is -inf, exp(-inf) = 0 adds
    exactly nothing to the softmax sum), so it changes no value on real frames: 1 (default, no padding) and 2 agree -
    tests/test_oracle.py::test_pad_to_multiple_is_a_noop checks that.
    """
    if feature_grad_mult > 0:
        x = feature_extractor(sd, wav, taps)
        if feature_grad_mult != 1.0:
            # fairseq multiplies the (B,512,T) tensor before the transpose; a transpose commutes with an elementwise scale
            x = GradMultiply.apply(x, feature_grad_mult)
    else:
        with torch.no_grad():
            x = feature_extractor(sd, wav, taps)
    x = F.layer_norm(x, (512,), sd[P + "layer_norm.weight"], sd[P + "layer_norm.bias"], 1e-5)
    x = F.linear(x, sd[P + "post_extract_proj.weight"], sd[P + "post_extract_proj.bias"])
    if stoch is not None:  # dropout_input
        x = x * stoch.mult(0, x.shape, stoch.dropout_input)
    if taps is not None:
        taps["proj"] = x
    w = fold_pos_conv_weight(sd)
    pc = F.conv1d(x.transpose(1, 2), w, sd[P + "encoder.pos_conv.0.bias"], padding=64, groups=16)
    pc = pc[:, :, :-1]  # SamePad: even kernel drops the last frame
    x = x + F.gelu(pc).transpose(1, 2)
    x = F.layer_norm(x, (768,), sd[P + "encoder.layer_norm.weight"], sd[P + "encoder.layer_norm.bias"], 1e-5)
    if stoch is not None:  # TransformerEncoder: F.dropout(x, p=self.dropout) after the LayerNorm
        x = x * stoch.mult(1, x.shape, stoch.dropout)
    if taps is not None:
        taps["enc_in"] = x
    T_real, kpm = x.shape[1], None
    pad = (-T_real) % max(1, int(required_seq_len_multiple))
    if pad:
        if stoch is not None:
            raise NotImplementedError("the padded variant restates the eval-mode call only")
        x = F.pad(x, (0, 0, 0, pad), value=0.0)  # pad_to_multiple(x, multiple, dim=-2, value=0)
        kpm = torch.zeros(x.shape[0], x.shape[1], dtype=torch.bool)
        kpm[:, -pad:] = True
    layers = []
    for l in range(NUM_LAYERS):
        if stoch is None:
            x = encoder_layer(sd, l, x, taps, stoch, kpm)
        elif stoch.branch_masks is None:
            if (stoch.layer_mask >> l) & 1:  # LayerDrop: a dropped layer is the identity
                x = encoder_layer(sd, l, x, taps, stoch)
        else:
            # merged batch of equal branches (anchor | positive | negative), LayerDrop decided per branch: clips are
            # independent, so "the layer on the kept branches" = the layer on everything, kept where it applies
            nb = len(stoch.branch_masks)
            keep = torch.tensor([(m >> l) & 1 for m in stoch.branch_masks], dtype=torch.bool)
            if bool(keep.any()):
                keep_rows = keep.repeat_interleave(x.shape[0] // nb)[:, None, None]
                x = torch.where(keep_rows, encoder_layer(sd, l, x, taps, stoch), x)
        layers.append(x)
    if pad:  # "undo padding": x[:, :-pad_length] and every layer result likewise
        x = x[:, :T_real]
        layers = [y[:, :T_real] for y in layers]
    return x, layers


def head(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """mean over time -> ReLU -> Linear(768,256) -> L2 normalise (nomad.py:228-230)."""
    e = F.linear(F.relu(torch.mean(x, 1)), w, b)
    return F.normalize(e, dim=1)


def triplet_forward(sd: Dict[str, torch.Tensor], wav: torch.Tensor, stoch: Optional["Stochastic"] = None,
                    feature_grad_mult: float = 1.0) -> torch.Tensor:
    """``TripletModel.forward`` (nomad.py:224-231): wav (B,1,N) or (B,N) -> (B,256) unit-norm.
    feature_grad_mult: see ``backbone`` (only matters for gradients that enter the conv feature extractor)."""
    if wav.dim() == 3:
        wav = wav.squeeze(1)
    x, _ = backbone(sd, wav, stoch=stoch, feature_grad_mult=feature_grad_mult)
    return head(x, sd["embedding_layer.1.weight"], sd["embedding_layer.1.bias"])


def lossnet_forward(sd: Dict[str, torch.Tensor], wav: torch.Tensor,
                    emb_w: torch.Tensor, emb_b: torch.Tensor, feature_grad_mult: float = 1.0,
                    required_seq_len_multiple: int = 1) -> List[torch.Tensor]:
    """``LossNetLayers.forward`` (nomad.py:243-258): 12 layer outputs (B,T,768) + embedding (B,256).

    ``emb_w/emb_b`` are LossNetLayers' OWN embedding layer (nomad.py:238-241), which the reference
    never loads from the checkpoint (it stays randomly initialised) - callers inject it.
    ``feature_grad_mult`` / ``required_seq_len_multiple``: see ``backbone`` (the reference's model has 0.1 and,
    with fairseq >= 0.12, 2).
    """
    if wav.dim() == 3:
        wav = wav.squeeze(1)
    x, layers = backbone(sd, wav, feature_grad_mult=feature_grad_mult, required_seq_len_multiple=required_seq_len_multiple)
    return list(layers) + [head(x, emb_w, emb_b)]


def nomad_loss(nomad_ref: Sequence[torch.Tensor], nomad_test: Sequence[torch.Tensor]) -> torch.Tensor:
    """``NomadLoss.forward`` (nomad.py:267-282): sum over 13 entries of mean |test - ref|."""
    loss = 0.0
    for i in range(13):
        loss = loss + F.l1_loss(nomad_test[i], nomad_ref[i])
    return loss


def pairwise(test_emb: np.ndarray, nmr_emb: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """cdist (Euclidean, float64, difference form) + row mean, as nomad.py:108-111."""
    a = np.asarray(test_emb, dtype=np.float64)
    b = np.asarray(nmr_emb, dtype=np.float64)
    d = np.sqrt(((a[:, None, :] - b[None, :, :]) ** 2).sum(-1))
    return d, d.mean(axis=1)


def load_wav(path: str) -> Tuple[np.ndarray, int]:
    """Minimal RIFF/WAVE reader: (channels, N) float32 in [-1,1) (int16/32768 like torchaudio.load), sr."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos = 12
    fmt = None
    while pos + 8 <= len(data):
        cid = data[pos:pos + 4]
        size = struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            tag, ch, sr, _, _, bits = fmt
            if tag == 1 and bits == 16:
                x = np.frombuffer(body, dtype="<i2").astype(np.float32) / 32768.0
            elif tag == 1 and bits == 32:
                x = np.frombuffer(body, dtype="<i4").astype(np.float32) / 2147483648.0
            elif tag == 3 and bits == 32:
                x = np.frombuffer(body, dtype="<f4").astype(np.float32)
            else:
                raise ValueError(f"{path}: unsupported wav format tag={tag} bits={bits}")
            return x.reshape(-1, ch).T.copy(), sr
        pos += 8 + size + (size & 1)
    raise ValueError(f"{path}: no data chunk")


def load_processing(path: str) -> torch.Tensor:
    """``Nomad.load_processing`` (nomad.py:192-212) for 16 kHz input: (1,N) fp32 mono."""
    x, sr = load_wav(path)
    if x.shape[0] > 1:
        x = ((x[0] + x[1]) / 2)[None, :]
    if sr != 16000:
        raise NotImplementedError("oracle covers 16 kHz input only (all reference fixtures are 16 kHz)")
    return torch.from_numpy(np.ascontiguousarray(x))


# ---- triplet fine-tuning step (/root/reference/src/training/train_triplet.py:112-133) ---------------------------
# Checked in tests against torch's own autograd / torch.optim.Adam, which ARE what the reference executes for
# these lines (nn.TripletMarginLoss, loss.backward(), torch.optim.Adam); the backbone underneath is the restatement
# above, with its pinning status.

def trainable_keys(sd: Dict[str, torch.Tensor], freeze_convnet: bool = True) -> List[str]:
    """Parameters train_triplet.py leaves trainable: with freeze_convnet: True (src/config/train_triplet.yaml)
    everything outside ``ssl_model.feature_extractor``, with False (train_triplet.py:71-73) those too.
    ``mask_emb`` gets no gradient with mask=False."""
    return [k for k in sd if (not freeze_convnet or "feature_extractor" not in k) and not k.endswith("mask_emb")]


class Stochastic:
    """model.train() regularisation of one forward call, with the ENGINE's mask generator restated
    (nomad_amd/csrc/dropout.hip.h: keep <=> hash(seed, site, element) >= round(p * 2^32)).

    Which elements fairseq's own dropout would drop depends on torch's RNG stream of the device it runs on and is
    not reproducible anywhere else; what the restatement pins is WHERE dropout is applied and HOW (sites, scaling,
    softmax normaliser untouched, LayerDrop as identity) - fairseq wav2vec2.py / transformer_sentence_encoder_layer
    semantics - given the same masks on both sides."""

    def __init__(self, seed: int, dropout: float = 0.1, attention_dropout: float = 0.1, dropout_input: float = 0.1,
                 layer_mask: int = 0xFFF, branch_masks: Optional[Sequence[int]] = None):
        self.seed, self.dropout, self.attention_dropout = int(seed), float(dropout), float(attention_dropout)
        self.dropout_input, self.layer_mask = float(dropout_input), int(layer_mask)
        self.branch_masks = list(branch_masks) if branch_masks is not None else None  # one LayerDrop mask per branch

    @staticmethod
    def _fmix32(h: np.ndarray) -> np.ndarray:
        h = h ^ (h >> np.uint32(16))
        h = h * np.uint32(0x85EBCA6B)
        h = h ^ (h >> np.uint32(13))
        h = h * np.uint32(0xC2B2AE35)
        return h ^ (h >> np.uint32(16))

    def mult(self, site: int, shape, p: float) -> torch.Tensor:
        """Per-element multiplier (1/(1-p) kept, 0 dropped) for a C-ordered tensor of ``shape``."""
        p32 = np.float32(p)
        if p32 <= 0:
            return torch.ones(tuple(shape))
        t = float(p32) * 4294967296.0
        threshold = np.uint32(4294967295 if t >= 4294967295.0 else int(t + 0.5))
        n = int(np.prod(shape))
        idx = np.arange(n, dtype=np.uint64)
        lo, hi = np.uint32(self.seed & 0xFFFFFFFF), np.uint32((self.seed >> 32) & 0xFFFFFFFF)
        with np.errstate(over="ignore"):
            h = self._fmix32(idx.astype(np.uint32) ^ lo ^ np.uint32((site * 0x9E3779B9) & 0xFFFFFFFF))
            h = self._fmix32(h + (idx >> np.uint64(32)).astype(np.uint32) * np.uint32(0x85EBCA77) + hi)
        scale = np.float32(1.0) / (np.float32(1.0) - p32)
        return torch.from_numpy(np.where(h >= threshold, scale, np.float32(0)).astype(np.float32).reshape(tuple(shape)))


def triplet_step_grads(sd: Dict[str, torch.Tensor], A: torch.Tensor, Pw: torch.Tensor, N: torch.Tensor,
                       margin: float, stoch: Optional[Sequence[Optional[Stochastic]]] = None,
                       freeze_convnet: bool = True, feature_grad_mult: float = 0.1
                       ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """One A/P/N forward + nn.TripletMarginLoss(margin) + backward (train_triplet.py:121-128).
    stoch: None = eval-mode arithmetic, or one Stochastic per branch (A, P, N).
    freeze_convnet=False (train_triplet.py:71-73): the conv feature extractor's parameters are differentiated too, and
    their gradients carry fairseq's ``feature_grad_mult`` (0.1 in wav2vec_small.pt's config).
    -> (loss, {key: d loss / d parameter})."""
    st = list(stoch) if stoch is not None else [None, None, None]
    sd = {k: v.clone() for k, v in sd.items()}
    keys = trainable_keys(sd, freeze_convnet)
    for k in keys:
        sd[k].requires_grad_(True)
    fgm = 1.0 if freeze_convnet else feature_grad_mult
    ea, ep, en = (triplet_forward(sd, A, st[0], fgm), triplet_forward(sd, Pw, st[1], fgm), triplet_forward(sd, N, st[2], fgm))
    loss = torch.nn.TripletMarginLoss(margin=margin)(ea, ep, en)
    grads = torch.autograd.grad(loss, [sd[k] for k in keys])
    return loss.detach(), dict(zip(keys, grads))


def make_adam(sd: Dict[str, torch.Tensor], lr: float, lr_body: float = 1e-5):
    """The reference's optimiser (train_triplet.py:98-107): Adam, backbone at 1e-5, embedding_layer at ``lr``.
    -> (optimizer, {key: Parameter})."""
    head = ("embedding_layer.1.weight", "embedding_layer.1.bias")
    params = {k: torch.nn.Parameter(sd[k].clone()) for k in trainable_keys(sd)}
    body = [p for k, p in params.items() if k not in head]
    opt = torch.optim.Adam([{"params": body, "lr": lr_body}, {"params": [params[k] for k in head]}], lr=lr)
    return opt, params
