"""Generate tests/golden/*.npz from an implementation that is INDEPENDENT of the oracle.

Run in the build container only:  python oracle/make_golden.py

The reference (/root/reference/src/nomad_audio/nomad.py) cannot be imported here (fairseq,
torchaudio and the network download of its weights are all unavailable).  What is importable is
HuggingFace ``transformers``' ``Wav2Vec2Model`` - a separate implementation of the same upstream
wav2vec 2.0 BASE architecture that fairseq implements.  This script loads the SAME seeded
parameter set (``nomad_amd.weights.seeded_state_dict``) into that model through the published
fairseq->HF key map, runs the reference's head (nomad.py:228-230) and SciPy's ``cdist`` /
``np.mean`` (nomad.py:108-111) on top, and stores inputs/outputs as small fixtures.  Neither
``transformers`` nor anything under /root/reference travels with the fixtures.

The one piece of the reference's OWN Python that imports with plain torch is ``src/models/networks.py`` (``TripletModel``,
the twin of nomad.py:214-231, and ``Origw2v``): the embedding fixtures below are produced by THAT class's forward, built
around the HF backbone through an adapter that answers fairseq's call ``ssl_model(wav, mask=False, features_only=True)``
with ``{'x': last_hidden_state}`` (``reference_models``).  So squeeze -> mean over time -> ReLU -> Linear -> normalize
 are pinned to the reference's code, the backbone (a2) to HF.

Fixtures written (data only):
  tests/golden/wavs/*.wav          the reference's six example clips (data/nmr-data, data/test-data)
  tests/golden/hf_example_wavs.npz embeddings (6,256), 2x4 distance matrix + means, per-layer checksums
  tests/golden/hf_tiny.npz         batch of 3 synthetic clips of 6000 samples: full 12 layer outputs + embeddings
  tests/golden/hf_loss.npz         nomad.forward() pins: two (2,1,16384) inputs -> loss and d loss/d estimate
                                   (HF layers + torch L1 + torch autograd)
  tests/golden/ref_networks.npz    (``python oracle/make_golden.py refnet``) reference TripletModel / Origw2v outputs on the tiny
                                   batch (both weight sets) and on the six example clips
  tests/golden/ref_nomad_classes.npz (``python oracle/make_golden.py refnomad``) outputs of the reference's OWN ``LossNetLayers``,
                                   ``NomadLoss``, ``Nomad.forward`` / ``predict`` / ``get_embeddings(_csv)`` (nomad.py:82-189,
                                   233-282), compiled from the file's ClassDef nodes without importing the module: 13 layer
                                   outputs + loss + gradients on the hf_loss.npz inputs; DataFrames, CSV bytes (dir and csv
                                   mode), default result paths and exception messages of ``predict`` on the example clips
  tests/golden/hf_grad_fgm.npz     (``python oracle/make_golden.py fgm``) gradients with fairseq's
                                   ``GradMultiply(features, feature_grad_mult)`` hooked onto the HF model's
                                   feature-extractor output: d loss/d estimate of the hf_loss.npz inputs at 0.1, and
                                   the gradient of a smooth functional of the 13 outputs of one 42 000-sample clip
                                   (T = 131: three attention tiles) at 0.1 and 1.0
"""
import os
import shutil
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nomad_amd.weights import seeded_state_dict  # noqa: E402

REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")


def hf_model_from_state_dict(sd):
    from transformers import Wav2Vec2Config, Wav2Vec2Model
    cfg = Wav2Vec2Config()  # defaults == wav2vec2-base
    cfg.apply_spec_augment = False
    m = Wav2Vec2Model(cfg).eval()
    hf = {}
    p = "ssl_model."
    for i in range(7):
        hf[f"feature_extractor.conv_layers.{i}.conv.weight"] = sd[p + f"feature_extractor.conv_layers.{i}.0.weight"]
    hf["feature_extractor.conv_layers.0.layer_norm.weight"] = sd[p + "feature_extractor.conv_layers.0.2.weight"]
    hf["feature_extractor.conv_layers.0.layer_norm.bias"] = sd[p + "feature_extractor.conv_layers.0.2.bias"]
    hf["feature_projection.layer_norm.weight"] = sd[p + "layer_norm.weight"]
    hf["feature_projection.layer_norm.bias"] = sd[p + "layer_norm.bias"]
    hf["feature_projection.projection.weight"] = sd[p + "post_extract_proj.weight"]
    hf["feature_projection.projection.bias"] = sd[p + "post_extract_proj.bias"]
    hf["encoder.pos_conv_embed.conv.bias"] = sd[p + "encoder.pos_conv.0.bias"]
    hf["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = sd[p + "encoder.pos_conv.0.weight_g"]
    hf["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = sd[p + "encoder.pos_conv.0.weight_v"]
    hf["encoder.layer_norm.weight"] = sd[p + "encoder.layer_norm.weight"]
    hf["encoder.layer_norm.bias"] = sd[p + "encoder.layer_norm.bias"]
    for l in range(12):
        a, b = p + f"encoder.layers.{l}.", f"encoder.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            for wb in ("weight", "bias"):
                hf[b + f"attention.{n}.{wb}"] = sd[a + f"self_attn.{n}.{wb}"]
        for wb in ("weight", "bias"):
            hf[b + f"layer_norm.{wb}"] = sd[a + f"self_attn_layer_norm.{wb}"]
            hf[b + f"feed_forward.intermediate_dense.{wb}"] = sd[a + f"fc1.{wb}"]
            hf[b + f"feed_forward.output_dense.{wb}"] = sd[a + f"fc2.{wb}"]
            hf[b + f"final_layer_norm.{wb}"] = sd[a + f"final_layer_norm.{wb}"]
    hf["masked_spec_embed"] = sd[p + "mask_emb"]
    missing, unexpected = m.load_state_dict(hf, strict=False)
    assert not unexpected, unexpected
    assert all("masked_spec_embed" in k for k in missing), missing
    return m


@torch.no_grad()
def hf_forward(m, wav):
    out = m(wav, output_hidden_states=True)
    # hidden_states[0] is the encoder input (after pos-conv + LN); [1..12] are the layer outputs
    return out.last_hidden_state, list(out.hidden_states[1:])


class _FairseqCallAdapter(torch.nn.Module):
    """What the reference's wrappers call (networks.py:16,31; nomad.py:226): ``ssl_model(wav, mask=False,
    features_only=True)['x']`` - answered by the HF model's last hidden state."""

    def __init__(self, hf):
        super().__init__()
        self.hf = hf

    def forward(self, wav, mask=False, features_only=True):
        assert mask is False and features_only is True
        return {"x": self.hf(wav).last_hidden_state}


def reference_models(hf, sd):
    """(TripletModel, Origw2v) of /root/reference/src/models/networks.py:4-34, imported from the reference tree (build
    container only), around ``hf``; the seeded head goes into ``embedding_layer.1`` as nomad.py:63-65 loads it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("reference_networks", os.path.join(REF, "src", "models", "networks.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ssl = _FairseqCallAdapter(hf)
    tm = mod.TripletModel(ssl, 768, 256).eval()
    with torch.no_grad():
        tm.embedding_layer[1].weight.copy_(sd["embedding_layer.1.weight"])
        tm.embedding_layer[1].bias.copy_(sd["embedding_layer.1.bias"])
    return tm, mod.Origw2v(ssl, 768).eval()


def hf_head(x, w, b):
    e = torch.nn.functional.linear(torch.relu(x.mean(1)), w, b)
    return torch.nn.functional.normalize(e, dim=1)


def read_wav(path):
    from scipy.io import wavfile
    sr, x = wavfile.read(path)
    assert sr == 16000 and x.dtype == np.int16 and x.ndim == 1
    return torch.from_numpy(x.astype(np.float32) / 32768.0)[None, :]


def main():
    from scipy.spatial.distance import cdist
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(os.path.join(GOLD, "wavs", "nmr-data"), exist_ok=True)
    os.makedirs(os.path.join(GOLD, "wavs", "test-data"), exist_ok=True)
    names = {"nmr-data": ["FI53_04", "FL67_01", "MJ57_01", "MJ60_10"],
             "test-data": ["445-123860-0012_NOISE_15", "6563-285357-0042_OPUS_64k"]}
    for d, ns in names.items():
        for n in ns:
            shutil.copyfile(os.path.join(REF, "data", d, n + ".wav"), os.path.join(GOLD, "wavs", d, n + ".wav"))

    sd = seeded_state_dict(0)
    m = hf_model_from_state_dict(sd)
    we, be = sd["embedding_layer.1.weight"], sd["embedding_layer.1.bias"]
    ref_tm, _ = reference_models(m, sd)   # embeddings come from the reference's TripletModel.forward (networks.py:14-21)

    def ref_embed(model, w):              # (B, N) -> (B, 256): the reference takes (B, 1, N) and squeezes it itself
        with torch.no_grad():
            return model(w[:, None, :])

    # 1. the reference's own example: 2 deg x 4 ref
    embs, checks = {}, {}
    for d, ns in names.items():
        for n in ns:
            wav = read_wav(os.path.join(GOLD, "wavs", d, n + ".wav"))
            x, layers = hf_forward(m, wav)
            embs[n] = ref_embed(ref_tm, wav)[0].numpy()
            assert np.abs(embs[n] - hf_head(x, we, be)[0].numpy()).max() < 1e-7
            checks[n] = np.array([[float(l.double().sum()), float(l.double().abs().sum())] for l in layers])
    nmr = np.stack([embs[n] for n in names["nmr-data"]])
    deg = np.stack([embs[n] for n in names["test-data"]])
    dm = cdist(deg, nmr)
    np.savez(os.path.join(GOLD, "hf_example_wavs.npz"),
             nmr_names=np.array(names["nmr-data"]), deg_names=np.array(names["test-data"]),
             nmr_emb=nmr, deg_emb=deg, dist=dm, mean=dm.mean(axis=1),
             layer_checks=np.stack([checks[n] for n in names["nmr-data"] + names["test-data"]]))
    print("example wavs: dist\n", dm, "\nmean", dm.mean(axis=1))

    # 2. tiny synthetic batch with full layer outputs
    g = torch.Generator().manual_seed(123)
    wav = (0.1 * torch.randn(3, 6000, generator=g)).clamp(-1, 1)
    x, layers = hf_forward(m, wav)
    np.savez(os.path.join(GOLD, "hf_tiny.npz"), wav=wav.numpy(),
             layers=torch.stack(layers).numpy(), emb=ref_embed(ref_tm, wav).numpy())

    # 3. peaky-attention variant (qk_gain 6) on the same tiny batch
    sd2 = seeded_state_dict(1, qk_gain=6.0)
    m2 = hf_model_from_state_dict(sd2)
    x, layers = hf_forward(m2, wav)
    np.savez(os.path.join(GOLD, "hf_tiny_peaky.npz"), wav=wav.numpy(),
             last=x.numpy(), layer0=layers[0].numpy(),
             emb=ref_embed(reference_models(m2, sd2)[0], wav).numpy())

    # 4. nomad.forward() pins (LossNetLayers' own embedding layer is seeded explicitly)
    g = torch.Generator().manual_seed(7)
    clean = (0.1 * torch.randn(2, 1, 16384, generator=g)).clamp(-1, 1)
    est = (clean + 0.02 * torch.randn(2, 1, 16384, generator=g)).clamp(-1, 1)
    lw = (torch.rand(256, 768, generator=g) * 2 - 1) / 768 ** 0.5
    lb = (torch.rand(256, generator=g) * 2 - 1) / 768 ** 0.5
    est.requires_grad_(True)
    outs = []
    with torch.enable_grad():
        for w_ in (est, clean):
            o = m(w_.squeeze(1), output_hidden_states=True)
            outs.append(list(o.hidden_states[1:]) + [hf_head(o.last_hidden_state, lw, lb)])
        loss = sum(torch.nn.functional.l1_loss(a, b) for a, b in zip(outs[0], outs[1]))
        (grad,) = torch.autograd.grad(loss, est)
    est = est.detach()
    np.savez(os.path.join(GOLD, "hf_loss.npz"), estimate=est.numpy(), clean=clean.numpy(),
             emb_w=lw.numpy(), emb_b=lb.numpy(), loss=np.float64(loss.detach()), grad=grad.numpy(),
             terms=np.array([float(torch.nn.functional.l1_loss(a, b)) for a, b in zip(outs[0], outs[1])]))
    print("loss", float(loss))


def main_refnet():
    """tests/golden/ref_networks.npz: outputs of the reference's own TripletModel / Origw2v (networks.py) - the pin of SURVEY.md
    section 8 rows a1 / a3 / a4 to reference code."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    g = torch.Generator().manual_seed(123)
    wav = (0.1 * torch.randn(3, 6000, generator=g)).clamp(-1, 1)      # the hf_tiny.npz batch
    out = {"wav": wav.numpy()}
    for tag, sd in (("seed0", seeded_state_dict(0)), ("peaky", seeded_state_dict(1, qk_gain=6.0))):
        tm, ow = reference_models(hf_model_from_state_dict(sd), sd)
        with torch.no_grad():
            out[f"emb_{tag}"] = tm(wav[:, None, :]).numpy()            # TripletModel.forward: (B,1,N) -> (B,256)
            out[f"pooled_{tag}"] = ow(wav[:, None, :]).numpy()         # Origw2v.forward: (B,1,N) -> (B,768), mean over time
            if tag == "seed0":
                names = [("nmr-data", n) for n in ("FI53_04", "FL67_01", "MJ57_01", "MJ60_10")] + \
                        [("test-data", n) for n in ("445-123860-0012_NOISE_15", "6563-285357-0042_OPUS_64k")]
                out["example_names"] = np.array([n for _, n in names])
                out["example_emb"] = np.stack([tm(read_wav(os.path.join(GOLD, "wavs", d, n + ".wav"))[:, None, :])[0].numpy()
                                               for d, n in names])
    np.savez(os.path.join(GOLD, "ref_networks.npz"), **out)
    old = np.load(os.path.join(GOLD, "hf_tiny.npz"))
    print("ref TripletModel vs hf_tiny.npz emb: max|diff|", float(np.abs(out["emb_seed0"] - old["emb"]).max()))


class _GradMultiply(torch.autograd.Function):
    """fairseq/modules/grad_multiply.py (GradMultiply): identity forward, grad * scale backward."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return x.new(x)

    @staticmethod
    def backward(ctx, grad):
        return grad * ctx.scale, None


def smooth_functional_weights(B, T, seed):
    """Upstream gradients of the smooth test functional sum_i <out_i, G_i>; tests regenerate them from the seed."""
    g = torch.Generator().manual_seed(seed)
    G_layers = torch.randn(12, B, T, 768, generator=g) / (B * T * 768)
    G_emb = torch.randn(B, 256, generator=g) / (B * 256)
    return G_layers, G_emb


def main_fgm():
    """Gradient pins with fairseq's feature_grad_mult: HF has no such knob, so the GradMultiply is attached as a
    forward hook on HF's feature extractor (whose output is what fairseq calls ``features``)."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sd = seeded_state_dict(0)
    m = hf_model_from_state_dict(sd)
    scale = {"v": 1.0}
    m.feature_extractor.register_forward_hook(
        lambda mod, inp, out: out if scale["v"] == 1.0 else _GradMultiply.apply(out, scale["v"]))

    def hf_outputs(w):
        o = m(w, output_hidden_states=True)
        return list(o.hidden_states[1:]), o.last_hidden_state

    # (a) the hf_loss.npz inputs (same generator sequence as main() section 4), feature_grad_mult = 0.1
    g = torch.Generator().manual_seed(7)
    clean = (0.1 * torch.randn(2, 1, 16384, generator=g)).clamp(-1, 1)
    est = (clean + 0.02 * torch.randn(2, 1, 16384, generator=g)).clamp(-1, 1)
    lw = (torch.rand(256, 768, generator=g) * 2 - 1) / 768 ** 0.5
    lb = (torch.rand(256, generator=g) * 2 - 1) / 768 ** 0.5
    old = np.load(os.path.join(GOLD, "hf_loss.npz"))
    assert np.array_equal(old["estimate"], est.numpy()) and np.array_equal(old["emb_w"], lw.numpy())
    est.requires_grad_(True)
    scale["v"] = 0.1
    outs = []
    for w_ in (est, clean):
        layers, last = hf_outputs(w_.squeeze(1))
        outs.append(layers + [hf_head(last, lw, lb)])
    loss = sum(torch.nn.functional.l1_loss(a, b) for a, b in zip(outs[0], outs[1]))
    (grad_l1,) = torch.autograd.grad(loss, est)
    assert abs(float(loss.detach()) - float(old["loss"])) < 1e-6
    print("L1 loss grad at 0.1 vs 0.1 * stored grad: max rel",
          float((grad_l1 - 0.1 * torch.from_numpy(old["grad"])).abs().max() / grad_l1.abs().max()))

    # (b) one 42 000-sample clip (T = 131 frames: odd, three 64-key attention tiles), smooth functional
    gen = torch.Generator().manual_seed(21)
    wav = (0.1 * torch.randn(1, 42000, generator=gen)).clamp(-1, 1)
    T = 131
    G_layers, G_emb = smooth_functional_weights(1, T, 22)
    grads = {}
    for mult in (0.1, 1.0):
        scale["v"] = mult
        w = wav.clone().requires_grad_(True)
        layers, last = hf_outputs(w)
        assert layers[0].shape == (1, T, 768)
        s = sum((layers[i] * G_layers[i]).sum() for i in range(12)) + (hf_head(last, lw, lb) * G_emb).sum()
        (grads[mult],) = torch.autograd.grad(s, w)
    print("smooth functional: grad(0.1) vs 0.1 * grad(1.0): max rel",
          float((grads[0.1] - 0.1 * grads[1.0]).abs().max() / grads[0.1].abs().max()))
    np.savez(os.path.join(GOLD, "hf_grad_fgm.npz"), grad_l1_fgm01=grad_l1.numpy(), long_wav=wav.numpy(),
             long_grad_fgm01=grads[0.1].numpy(), long_grad_fgm1=grads[1.0].numpy(), g_seed=np.int64(22))  # head: emb_w / emb_b of hf_loss.npz


class _FairseqFullAdapter(torch.nn.Module):
    """fairseq's ``Wav2Vec2Model.forward(wav, mask=False, features_only=True)`` result as nomad.py:226-248 reads it: ``x`` (B,T,C)
    and ``layer_results`` - one tuple per encoder layer whose element 0 is that layer's output in fairseq's (T,B,C) layout."""

    def __init__(self, hf):
        super().__init__()
        self.hf = hf

    def forward(self, wav, mask=False, features_only=True):
        assert mask is False and features_only is True
        o = self.hf(wav, output_hidden_states=True)
        return {"x": o.last_hidden_state, "layer_results": [(h.transpose(0, 1), None) for h in o.hidden_states[1:]]}


REF_NOMAD_PY_SHA256 = "484dfd800773a987b7979bef205b11a612f65d98c264a6b42406a46f9d2839bd"   # /root/reference/src/nomad_audio/nomad.py as reviewed


def reference_nomad_namespace(fixed_now=None):
    """The reference's OWN classes ``Nomad``, ``TripletModel``, ``LossNetLayers``, ``NomadLoss`` (nomad.py:35-282), compiled from
    the ClassDef nodes of /root/reference/src/nomad_audio/nomad.py WITHOUT importing the module - so none of its import-time
    side effects (``import fairseq`` / ``torchaudio``, the two ``urlretrieve`` downloads) run.  Build container only; no text of
    the file is stored anywhere.  The namespace holds what the class bodies name: torch, nn, F, np, pd, cdist, tqdm, an ``os``
    whose ``listdir`` is sorted (the reference lists directories in file-system order; a fixture needs a defined one) and a
    ``datetime`` whose ``now()`` is fixed (the default ``results-csv/<timestamp>/`` paths)."""
    import ast
    import datetime as _dt
    import pandas as pd
    from scipy.spatial.distance import cdist

    class _SortedOS:
        def __getattr__(self, name):
            return getattr(os, name)

        @staticmethod
        def listdir(path):
            return sorted(os.listdir(path))

    class _FixedDatetime:
        @staticmethod
        def now():
            return fixed_now or _dt.datetime(2024, 1, 2, 3, 4, 5)

    # The class bodies below are EXECUTED (with a real os: predict() writes CSVs and changes directory).  Only the file that was read
    # and reviewed when this script was written may run: anything else - an edited or replaced reference tree - is refused.
    # Regenerate fixtures in a throwaway container without credentials or network only.
    import hashlib
    with open(os.path.join(REF, "src", "nomad_audio", "nomad.py"), "rb") as f:
        raw = f.read()
    digest = hashlib.sha256(raw).hexdigest()
    if digest != REF_NOMAD_PY_SHA256:
        raise RuntimeError(f"{REF}/src/nomad_audio/nomad.py has sha256 {digest}, not the reviewed {REF_NOMAD_PY_SHA256}: refusing to execute it")
    tree = ast.parse(raw.decode())
    wanted = ("Nomad", "TripletModel", "LossNetLayers", "NomadLoss")
    body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in wanted]
    assert sorted(n.name for n in body) == sorted(wanted)
    ns = {"torch": torch, "nn": torch.nn, "F": torch.nn.functional, "np": np, "pd": pd, "cdist": cdist,
          "tqdm": (lambda it: it), "os": _SortedOS(), "datetime": _FixedDatetime}
    exec(compile(ast.Module(body=body, type_ignores=[]), "<classes of the reference nomad.py>", "exec"), ns)
    return ns


def reference_nomad(ns, hf, sd, loss_head=None):
    """An instance of the reference's ``Nomad`` without its ``__init__`` (which loads fairseq checkpoints): the attributes
    ``__init__`` sets (nomad.py:39-80) built from the same classes around the HF backbone, and ``load_processing`` - the one
    method that needs torchaudio - replaced by a PCM-16 reader that returns what ``torchaudio.load`` returns for such a file."""
    ssl = _FairseqFullAdapter(hf)
    n = ns["Nomad"].__new__(ns["Nomad"])
    n.DEVICE = "cpu"
    n.model = ns["TripletModel"](ssl, 768, 256)
    with torch.no_grad():
        n.model.embedding_layer[1].weight.copy_(sd["embedding_layer.1.weight"])
        n.model.embedding_layer[1].bias.copy_(sd["embedding_layer.1.bias"])
    n.model.eval()
    n.lossnet_layers = ns["LossNetLayers"](ssl, 768, 256)
    if loss_head is not None:
        with torch.no_grad():
            n.lossnet_layers.embedding_layer[1].weight.copy_(loss_head[0])
            n.lossnet_layers.embedding_layer[1].bias.copy_(loss_head[1])
    n.nomad_loss = ns["NomadLoss"]()
    n.nomad_loss.eval()

    def load_processing(filepath, target_sr=16000, trim=False):
        if isinstance(filepath, np.ndarray):     # nomad.py:194-195
            filepath = filepath[0]
        return read_wav(filepath)
    n.load_processing = load_processing
    return n


def _bytes(path):
    with open(path, "rb") as f:
        return np.frombuffer(f.read(), dtype=np.uint8)


def main_refnomad():
    """tests/golden/ref_nomad_classes.npz: what the reference's own ``LossNetLayers.forward`` / ``NomadLoss.forward`` /
    ``Nomad.forward`` (nomad.py:142-146, 233-282) return on the hf_loss.npz inputs, and what its own ``Nomad.predict`` /
    ``get_embeddings`` / ``get_embeddings_csv`` (nomad.py:82-189) return and WRITE for the example directories - DataFrames, the
    bytes of both CSV files in both modes, the default result paths, and every exception message of the argument checks."""
    import tempfile
    import pandas as pd
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sd = seeded_state_dict(0)
    hf = hf_model_from_state_dict(sd)
    scale = {"v": 1.0}
    hf.feature_extractor.register_forward_hook(
        lambda mod, inp, out: out if scale["v"] == 1.0 else _GradMultiply.apply(out, scale["v"]))
    old = np.load(os.path.join(GOLD, "hf_loss.npz"))
    est, clean = torch.from_numpy(old["estimate"]), torch.from_numpy(old["clean"])
    lw, lb = torch.from_numpy(old["emb_w"]), torch.from_numpy(old["emb_b"])
    ns = reference_nomad_namespace()
    n = reference_nomad(ns, hf, sd, loss_head=(lw, lb))
    out = {}

    # ---- LossNetLayers.forward, NomadLoss.forward, Nomad.forward + autograd -------------------------------------------
    with torch.no_grad():
        a = n.lossnet_layers(est)            # (B,1,N) -> 12 x (B,T,768) + (B,256)
        b = n.lossnet_layers(clean)
        loss = n.nomad_loss(b, a)
    assert len(a) == 13 and a[0].shape == (2, 50, 768) and a[12].shape == (2, 256)
    out["loss"] = np.float64(loss)
    out["terms"] = np.array([float(torch.nn.functional.l1_loss(x, y)) for x, y in zip(a, b)])
    for tag, lst in (("est", a), ("clean", b)):
        out[f"checks_{tag}"] = np.array([[float(t.double().sum()), float(t.double().abs().sum())] for t in lst])
        out[f"emb_{tag}"] = lst[12].numpy()
        out[f"layers_{tag}_sample"] = np.stack([t.reshape(-1)[::97].numpy() for t in lst[:12]])
    n.nomad_loss.only_embedding = True       # the reference's other branch indexes element 13 of a 13-element list (nomad.py:271-272)
    try:
        n.nomad_loss(b, a)
        out["only_embedding_error"] = np.array("")
    except Exception as e:  # noqa: BLE001
        out["only_embedding_error"] = np.array(type(e).__name__)
    n.nomad_loss.only_embedding = False
    for mult, key in ((1.0, "grad_fgm1"), (0.1, "grad_fgm01")):
        scale["v"] = mult
        e = est.clone().requires_grad_(True)
        l = n.forward(e, clean)              # Nomad.forward (nomad.py:142-146)
        (g,) = torch.autograd.grad(l, e)
        out[key] = g.numpy()
        assert abs(float(l.detach()) - float(out["loss"])) < 1e-6
    scale["v"] = 1.0
    print("reference NomadLoss", float(out["loss"]), "hf_loss.npz", float(old["loss"]),
          "| terms max diff", float(np.abs(out["terms"] - old["terms"]).max()),
          "| grad max rel diff", float(np.abs(out["grad_fgm1"] - old["grad"]).max() / np.abs(old["grad"]).max()),
          "| only_embedding ->", str(out["only_embedding_error"]))

    # ---- Nomad.predict / get_embeddings / get_embeddings_csv on the example directories ------------------------------
    nmr_dir, deg_dir = os.path.join(GOLD, "wavs", "nmr-data"), os.path.join(GOLD, "wavs", "test-data")
    with tempfile.TemporaryDirectory() as tmp:
        df_avg, df_dm = n.predict("dir", nmr_dir, deg_dir, results_path=tmp)
        out["dir_avg_csv"], out["dir_scores_csv"] = _bytes(os.path.join(tmp, "nomad_avg.csv")), _bytes(os.path.join(tmp, "nomad_scores.csv"))
        out["avg_index"], out["avg_values"] = np.array(list(df_avg.index)), df_avg["NOMAD"].to_numpy()
        out["avg_index_name"], out["avg_columns"] = np.array(df_avg.index.name), np.array(list(df_avg.columns))
        out["dm_index"], out["dm_columns"], out["dm_values"] = np.array(list(df_dm.index)), np.array(list(df_dm.columns)), df_dm.to_numpy()
        emb = n.get_embeddings(deg_dir)      # the embeddings table itself (nomad.py:148-163)
        out["emb_columns"] = np.array([str(c) for c in emb.columns])
        out["emb_filenames_rel"] = np.array([os.path.relpath(p, GOLD) for p in emb["filename"]])
        out["emb_values"] = emb.drop("filename", axis=1).to_numpy(dtype=np.float32)
        # csv mode: the same files listed in two csv files, in REVERSE order (the listing order must be kept)
        lists = {}
        for tag, d in (("nmr", nmr_dir), ("deg", deg_dir)):
            lists[tag] = os.path.join(tmp, f"{tag}.csv")
            pd.DataFrame({"filename": [os.path.join(d, f) for f in sorted(os.listdir(d), reverse=True)]}).to_csv(lists[tag], index=False)
        os.makedirs(os.path.join(tmp, "csvmode"))
        n.predict("csv", lists["nmr"], lists["deg"], results_path=os.path.join(tmp, "csvmode"))
        out["csv_avg_csv"] = _bytes(os.path.join(tmp, "csvmode", "nomad_avg.csv"))
        out["csv_scores_csv"] = _bytes(os.path.join(tmp, "csvmode", "nomad_scores.csv"))
        # default result paths (results_path=None) under the fixed clock, relative to the working directory
        cwd = os.getcwd()
        os.makedirs(os.path.join(tmp, "cwd"))
        os.chdir(os.path.join(tmp, "cwd"))
        try:
            n.predict("dir", nmr_dir, deg_dir)
            made = sorted(os.path.join(r, f) for r, _, fs in os.walk(".") for f in fs)
        finally:
            os.chdir(cwd)
        out["default_paths"] = np.array([p[2:] for p in made])
        out["fixed_now"] = np.array("2024-01-02 03:04:05")
        # the argument checks (nomad.py:83-99) and the csv without a 'filename' column (nomad.py:157-158)
        bad = os.path.join(tmp, "bad.csv")
        pd.DataFrame({"file": ["a.wav"]}).to_csv(bad, index=False)
        cases = {"nmr_none": ("dir", None, deg_dir), "deg_none": ("dir", nmr_dir, None),
                 "dir_nmr_missing": ("dir", "/nonexistent/nmr", deg_dir), "dir_deg_missing": ("dir", nmr_dir, "/nonexistent/deg"),
                 "csv_nmr_missing": ("csv", "/nonexistent/nmr.csv", lists["deg"]), "csv_deg_missing": ("csv", lists["nmr"], "/nonexistent/deg.csv"),
                 "bad_mode": ("zip", nmr_dir, deg_dir)}
        msgs = {}
        for key, args in cases.items():
            try:
                n.predict(*args)
                msgs[key] = None
            except Exception as e:  # noqa: BLE001
                msgs[key] = [type(e).__name__, str(e)]
        try:
            n.get_embeddings(bad)
            msgs["csv_without_filename_column"] = None
        except Exception as e:  # noqa: BLE001
            msgs["csv_without_filename_column"] = [type(e).__name__, str(e)]
    import json
    out["messages_json"] = np.array(json.dumps(msgs))
    np.savez(os.path.join(GOLD, "ref_nomad_classes.npz"), **out)
    g = np.load(os.path.join(GOLD, "hf_example_wavs.npz"))
    print("reference predict table vs hf_example_wavs.npz: max |diff|",
          float(np.abs(out["dm_values"] - np.round(g["dist"], 3)).max()), float(np.abs(out["avg_values"] - np.round(g["mean"], 3)).max()))
    print(out["dir_scores_csv"].tobytes().decode())
    print(out["dir_avg_csv"].tobytes().decode())
    print(out["default_paths"], msgs)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "refnomad":
        main_refnomad()
    elif len(sys.argv) > 1 and sys.argv[1] == "fgm":
        main_fgm()
    elif len(sys.argv) > 1 and sys.argv[1] == "refnet":
        main_refnet()
    else:
        main()
