"""The reference's OWN classes as the checker.

``tests/golden/ref_nomad_classes.npz`` was produced by ``oracle/make_golden.py refnomad``: the ClassDef nodes ``Nomad``,
``TripletModel``, ``LossNetLayers`` and ``NomadLoss`` of /root/reference/src/nomad_audio/nomad.py compiled without importing the
module, run over the HF backbone with the seeded weights.  Held against it here:

* CPU: the oracle's restatement of ``LossNetLayers.forward`` / ``NomadLoss.forward`` (13 outputs, loss, gradients);
* CPU: ``nomad_amd.Nomad.predict`` / ``get_embeddings`` / ``get_embeddings_csv`` - the host layer, over a fake engine that
  returns the reference's embeddings - byte for byte against the CSV files the reference wrote (dir and csv mode), the default
  result paths under a fixed clock, the DataFrames and every exception message;
* GPU (``-m gpu``): the same CSV bytes and the loss / layer outputs / gradient from the real engine.
"""
import datetime as _dt
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLD
from oracle import nomad_oracle as O

REFN = os.path.join(GOLD, "ref_nomad_classes.npz")
NMR_DIR, DEG_DIR = os.path.join(GOLD, "wavs", "nmr-data"), os.path.join(GOLD, "wavs", "test-data")
torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))


@pytest.fixture(scope="module")
def ref():
    return np.load(REFN)


@pytest.fixture(scope="module")
def loss_inputs():
    g = np.load(os.path.join(GOLD, "hf_loss.npz"))
    return {k: torch.from_numpy(g[k]) for k in ("estimate", "clean", "emb_w", "emb_b")}


def _check_13_outputs(outs, ref, tag, tol):
    assert len(outs) == 13
    for i, t in enumerate(outs):
        t = t.detach().double().cpu()
        scale = ref[f"checks_{tag}"][i, 1] / t.numel()                      # mean |value| of this output
        assert abs(float(t.sum()) - ref[f"checks_{tag}"][i, 0]) < tol * scale * t.numel() ** 0.5 * 4, (tag, i)
        assert abs(float(t.abs().sum()) - ref[f"checks_{tag}"][i, 1]) < tol * t.numel(), (tag, i)
        if i < 12:
            got = t.reshape(-1)[::97].numpy()
            assert np.abs(got - ref[f"layers_{tag}_sample"][i]).max() < tol * max(1.0, np.abs(ref[f"layers_{tag}_sample"][i]).max()), (tag, i)
        else:
            assert np.abs(t.numpy() - ref[f"emb_{tag}"]).max() < tol, tag


# ---- CPU: the oracle against the reference's LossNetLayers / NomadLoss / Nomad.forward ---------------------------------
def test_oracle_lossnet_and_loss_match_the_reference_classes(sd0, ref, loss_inputs):
    li = loss_inputs
    with torch.no_grad():
        a = O.lossnet_forward(sd0, li["estimate"], li["emb_w"], li["emb_b"])
        b = O.lossnet_forward(sd0, li["clean"], li["emb_w"], li["emb_b"])
        loss = O.nomad_loss(b, a)
    _check_13_outputs(a, ref, "est", 2e-5)
    _check_13_outputs(b, ref, "clean", 2e-5)
    assert abs(float(loss) - float(ref["loss"])) < 2e-5
    terms = np.array([float(torch.nn.functional.l1_loss(x, y)) for x, y in zip(a, b)])
    assert np.abs(terms - ref["terms"]).max() < 5e-6


@pytest.mark.parametrize("mult,key", [(1.0, "grad_fgm1"), (0.1, "grad_fgm01")])
def test_oracle_gradient_matches_the_reference_forward(sd0, ref, loss_inputs, mult, key):
    li = loss_inputs
    est = li["estimate"].clone().requires_grad_(True)
    a = O.lossnet_forward(sd0, est, li["emb_w"], li["emb_b"], feature_grad_mult=mult)
    with torch.no_grad():
        b = O.lossnet_forward(sd0, li["clean"], li["emb_w"], li["emb_b"])
    (g,) = torch.autograd.grad(O.nomad_loss(b, a), est)
    want = ref[key]
    # the L1 sign makes the gradient piecewise constant in the layer outputs: a handful of elements flip between implementations
    assert np.abs(g.numpy() - want).max() < 6e-3 * np.abs(want).max()
    cos = float((g.numpy().ravel() @ want.ravel()) / (np.linalg.norm(g.numpy()) * np.linalg.norm(want)))
    assert cos > 0.9999
    if mult == 0.1:
        assert np.abs(want - 0.1 * ref["grad_fgm1"]).max() < 1e-5 * np.abs(want).max()   # GradMultiply scales the whole dX (fp32 rounding of the scaled chain)


# ---- CPU: the product's host layer (predict / get_embeddings / get_embeddings_csv) over a fake engine -------------------
class _RefEmbeddingEngine:
    """Stands where the HIP engine stands: 'embeds' a clip by looking its length up in the reference's embedding table
    (the six example clips all differ in length); distances by the CPU oracle."""

    device = torch.device("cpu")

    def __init__(self, ref):
        from nomad_amd import wavio
        g = np.load(os.path.join(GOLD, "ref_networks.npz"))
        self.by_len = {}
        dirs = {"FI53_04": "nmr-data", "FL67_01": "nmr-data", "MJ57_01": "nmr-data", "MJ60_10": "nmr-data",
                "445-123860-0012_NOISE_15": "test-data", "6563-285357-0042_OPUS_64k": "test-data"}
        for name, emb in zip(g["example_names"], g["example_emb"]):
            n = wavio.load_processing(os.path.join(GOLD, "wavs", dirs[str(name)], f"{name}.wav"), 16000, False).shape[1]
            assert n not in self.by_len
            self.by_len[n] = emb

    def pack_ragged_host(self, waves):
        lens = [int(np.asarray(w).reshape(-1).shape[0]) for w in waves]
        return object(), lens

    def embed_ragged(self, waves, precision=None, packed=None):
        return np.stack([self.by_len[n] for n in packed[1]])

    def fetch_async(self, emb):
        class F:
            def result(self_inner):
                return emb
        return F()

    def pairwise(self, deg, ref, want_matrix=True):
        d, m = O.pairwise(deg.numpy(), ref.numpy())
        return torch.from_numpy(d), torch.from_numpy(m)


@pytest.fixture()
def host_nomad(ref, monkeypatch):
    import importlib
    product = importlib.import_module("nomad_amd.nomad")   # (the attribute nomad_amd.nomad is the lazy singleton INSTANCE)
    n = product.Nomad.__new__(product.Nomad)
    n.engine, n.precision, n.model = _RefEmbeddingEngine(ref), "fp32", None
    n.NATIVE_WAV_THREADS = 0                               # decode through load_processing (no C library in this test)
    real_listdir = os.listdir
    monkeypatch.setattr(product.os, "listdir", lambda p: sorted(real_listdir(p)))   # the order the fixture was generated with

    class Fixed(_dt.datetime):
        @classmethod
        def now(cls, tz=None):
            return cls(2024, 1, 2, 3, 4, 5)
    monkeypatch.setattr(product, "datetime", Fixed)
    return n


def _assert_tables(df_avg, df_dm, ref):
    assert df_avg.index.name == str(ref["avg_index_name"]) and list(df_avg.columns) == [str(c) for c in ref["avg_columns"]]
    assert list(df_avg.index) == [str(x) for x in ref["avg_index"]] and np.array_equal(df_avg["NOMAD"].to_numpy(), ref["avg_values"])
    assert list(df_dm.index) == [str(x) for x in ref["dm_index"]] and list(df_dm.columns) == [str(x) for x in ref["dm_columns"]]
    assert np.array_equal(df_dm.to_numpy(), ref["dm_values"])


def test_predict_host_layer_writes_the_reference_bytes(host_nomad, ref, tmp_path, monkeypatch):
    import pandas as pd
    n = host_nomad
    out = tmp_path / "dir"
    out.mkdir()
    df_avg, df_dm = n.predict("dir", NMR_DIR, DEG_DIR, results_path=str(out))
    _assert_tables(df_avg, df_dm, ref)
    assert (out / "nomad_avg.csv").read_bytes() == ref["dir_avg_csv"].tobytes()
    assert (out / "nomad_scores.csv").read_bytes() == ref["dir_scores_csv"].tobytes()
    # the embeddings table of get_embeddings: column labels, file order, values
    emb = n.get_embeddings(DEG_DIR)
    assert [str(c) for c in emb.columns] == [str(c) for c in ref["emb_columns"]]
    assert [os.path.relpath(p, GOLD) for p in emb["filename"]] == [str(p) for p in ref["emb_filenames_rel"]]
    assert np.array_equal(emb.drop("filename", axis=1).to_numpy(dtype=np.float32), ref["emb_values"])
    # csv mode, files listed in reverse order
    lists = {}
    for tag, d in (("nmr", NMR_DIR), ("deg", DEG_DIR)):
        lists[tag] = str(tmp_path / f"{tag}.csv")
        pd.DataFrame({"filename": [os.path.join(d, f) for f in sorted(os.listdir(d), reverse=True)]}).to_csv(lists[tag], index=False)
    out2 = tmp_path / "csv"
    out2.mkdir()
    n.predict("csv", lists["nmr"], lists["deg"], results_path=str(out2))
    assert (out2 / "nomad_avg.csv").read_bytes() == ref["csv_avg_csv"].tobytes()
    assert (out2 / "nomad_scores.csv").read_bytes() == ref["csv_scores_csv"].tobytes()
    # default result paths under the fixture's clock
    monkeypatch.chdir(tmp_path)
    n.predict("dir", NMR_DIR, DEG_DIR)
    made = sorted(os.path.relpath(os.path.join(r, f), str(tmp_path)) for r, _, fs in os.walk(str(tmp_path / "results-csv")) for f in fs)
    assert made == [str(p) for p in ref["default_paths"]]
    assert (tmp_path / made[0]).read_bytes() == ref["dir_avg_csv"].tobytes()
    assert (tmp_path / made[1]).read_bytes() == ref["dir_scores_csv"].tobytes()


def test_predict_argument_checks_raise_the_reference_messages(host_nomad, ref, tmp_path):
    import pandas as pd
    n = host_nomad
    msgs = json.loads(str(ref["messages_json"]))
    lists = {}
    for tag, d in (("nmr", NMR_DIR), ("deg", DEG_DIR)):
        lists[tag] = str(tmp_path / f"{tag}.csv")
        pd.DataFrame({"filename": [os.path.join(d, f) for f in sorted(os.listdir(d))]}).to_csv(lists[tag], index=False)
    cases = {"nmr_none": ("dir", None, DEG_DIR), "deg_none": ("dir", NMR_DIR, None),
             "dir_nmr_missing": ("dir", "/nonexistent/nmr", DEG_DIR), "dir_deg_missing": ("dir", NMR_DIR, "/nonexistent/deg"),
             "csv_nmr_missing": ("csv", "/nonexistent/nmr.csv", lists["deg"]), "csv_deg_missing": ("csv", lists["nmr"], "/nonexistent/deg.csv"),
             "bad_mode": ("zip", NMR_DIR, DEG_DIR)}
    assert set(cases) | {"csv_without_filename_column"} == set(msgs)
    for key, args in cases.items():
        with pytest.raises(Exception) as ei:
            n.predict(*args)
        assert [type(ei.value).__name__, str(ei.value)] == msgs[key], key
    bad = str(tmp_path / "bad.csv")
    pd.DataFrame({"file": ["a.wav"]}).to_csv(bad, index=False)
    with pytest.raises(Exception) as ei:
        n.get_embeddings(bad)
    assert [type(ei.value).__name__, str(ei.value)] == msgs["csv_without_filename_column"]


def test_nomad_loss_only_embedding_branch_behaves_like_the_reference(ref):
    """nomad.py:270-273 indexes element 13 of the 13-element lists LossNetLayers returns: an IndexError in the reference."""
    from nomad_amd.nomad import NomadLoss
    assert str(ref["only_embedding_error"]) == "IndexError"
    nl = NomadLoss(engine=None)
    nl.only_embedding = True
    lst = [torch.zeros(1, 2, 768)] * 12 + [torch.zeros(1, 256)]
    with pytest.raises(IndexError):
        nl(lst, lst)


# ---- GPU: the real engine behind the same surface ------------------------------------------------------------------------
@pytest.mark.gpu
def test_gpu_predict_writes_the_reference_bytes(built_lib, sd0, ref, tmp_path, monkeypatch):
    import pandas as pd
    import importlib
    product = importlib.import_module("nomad_amd.nomad")   # (the attribute nomad_amd.nomad is the lazy singleton INSTANCE)
    real_listdir = os.listdir
    monkeypatch.setattr(product.os, "listdir", lambda p: sorted(real_listdir(p)))
    nmd = product.Nomad(weights=sd0)
    out = tmp_path / "dir"
    out.mkdir()
    df_avg, df_dm = nmd.predict("dir", NMR_DIR, DEG_DIR, results_path=str(out))
    _assert_tables(df_avg, df_dm, ref)
    assert (out / "nomad_avg.csv").read_bytes() == ref["dir_avg_csv"].tobytes()
    assert (out / "nomad_scores.csv").read_bytes() == ref["dir_scores_csv"].tobytes()
    emb = nmd.get_embeddings(DEG_DIR)
    assert [os.path.relpath(p, GOLD) for p in emb["filename"]] == [str(p) for p in ref["emb_filenames_rel"]]
    assert np.abs(emb.drop("filename", axis=1).to_numpy(dtype=np.float32) - ref["emb_values"]).max() < 1e-5
    lists = {}
    for tag, d in (("nmr", NMR_DIR), ("deg", DEG_DIR)):
        lists[tag] = str(tmp_path / f"{tag}.csv")
        pd.DataFrame({"filename": [os.path.join(d, f) for f in sorted(os.listdir(d), reverse=True)]}).to_csv(lists[tag], index=False)
    out2 = tmp_path / "csv"
    out2.mkdir()
    nmd.predict("csv", lists["nmr"], lists["deg"], results_path=str(out2))
    assert (out2 / "nomad_avg.csv").read_bytes() == ref["csv_avg_csv"].tobytes()
    assert (out2 / "nomad_scores.csv").read_bytes() == ref["csv_scores_csv"].tobytes()


@pytest.mark.gpu
def test_gpu_lossnet_loss_and_gradient_vs_the_reference_classes(built_lib, sd0, ref, loss_inputs):
    from nomad_amd.nomad import Nomad
    li = loss_inputs
    nmd = Nomad(weights=sd0, feature_grad_mult=0.1)
    nmd.lossnet_layers.embedding_weight = li["emb_w"].cuda()
    nmd.lossnet_layers.embedding_bias = li["emb_b"].cuda()
    _check_13_outputs(nmd.lossnet_layers(li["estimate"].cuda()), ref, "est", 1e-4)
    _check_13_outputs(nmd.lossnet_layers(li["clean"].cuda()), ref, "clean", 1e-4)
    est = li["estimate"].cuda().requires_grad_(True)
    loss = nmd.forward(est, li["clean"].cuda())
    assert abs(loss.item() - float(ref["loss"])) < 1e-4
    loss.backward()
    g, want = est.grad.cpu().numpy(), ref["grad_fgm01"]
    assert np.abs(g - want).max() < 6e-3 * np.abs(want).max()
    assert float((g.ravel() @ want.ravel()) / (np.linalg.norm(g) * np.linalg.norm(want))) > 0.9999
