"""GPU: the triplet fine-tuning step (/root/reference/src/training/train_triplet.py:112-133)
against torch autograd / torch.optim.Adam on the CPU oracle: loss, every parameter gradient, the Adam update,
and that the derived kernel-layout weights follow the master copy."""
import os

import numpy as np
import pytest
import torch

from oracle import nomad_oracle as O

pytestmark = pytest.mark.gpu

MARGIN = 0.2


@pytest.fixture(scope="module")
def sd_train():
    from nomad_amd.weights import seeded_state_dict
    return seeded_state_dict(3, qk_gain=3.0)


@pytest.fixture(scope="module")
def teng(built_lib, sd_train):
    from nomad_amd.engine import Engine
    eng = Engine({k: v.clone() for k, v in sd_train.items()}, 0)
    eng.train_enable()
    yield eng
    eng.close()


def _triplet_batch(B, n, seed):
    g = torch.Generator().manual_seed(seed)
    return [(0.1 * torch.randn(B, n, generator=g)).clamp(-1, 1) for _ in range(3)]


def _gpu_step_grads(eng, A, P, N, margin):
    """zero_grad + the reference's three forwards + loss + backward -> (loss, flat grads)."""
    eng.train_zero_grad()
    outs = [eng.embed_train(w.cuda()) for w in (A, P, N)]
    loss, da, dp, dn = eng.triplet_loss(outs[0][0], outs[1][0], outs[2][0], margin)
    for w, (emb, layers, saved), d in zip((A, P, N), outs, (da, dp, dn)):
        eng.train_backward(w.cuda(), layers, saved, d)
    return loss.cpu(), eng.train_read(1)


def test_segment_table_covers_trainable_keys(teng, sd_train):
    from nomad_amd.weights import expected_shapes
    shapes = expected_shapes()
    segs = teng.train_segments()
    total, head = teng.train_param_count()
    assert sorted(k for k, _, _ in segs) == sorted(O.trainable_keys(sd_train, freeze_convnet=False))   # every parameter the reference can train
    spans = sorted((o, n) for _, o, n in segs)
    assert spans[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert spans[-1][0] + spans[-1][1] == total
    for k, o, n in segs:
        assert n == int(np.prod(shapes[k])) and o % 4 == 0
    assert head == dict((k, o) for k, o, _ in segs)["embedding_layer.1.weight"]
    # the master copy starts as the checkpoint
    got = teng.train_unflatten(teng.train_read(0))
    for k in got:
        assert torch.equal(got[k], sd_train[k]), k


@pytest.mark.parametrize("B,margin", [(5, 0.2), (3, 1.0), (9, 0.05)])
def test_triplet_loss_and_gradient(teng, B, margin):
    g = torch.Generator().manual_seed(B)
    a, p, n = (torch.nn.functional.normalize(torch.randn(B, 256, generator=g), dim=1).requires_grad_(True) for _ in range(3))
    ref = torch.nn.TripletMarginLoss(margin=margin)(a, p, n)
    ga, gp, gn = torch.autograd.grad(ref, (a, p, n))
    loss, da, dp, dn = teng.triplet_loss(a.detach().cuda(), p.detach().cuda(), n.detach().cuda(), margin)
    assert abs(loss.item() - ref.item()) < 2e-6
    for got, want in ((da, ga), (dp, gp), (dn, gn)):
        assert (got.cpu() - want).abs().max().item() < 1e-6
    loss2, none_a, _, _ = teng.triplet_loss(a.detach().cuda(), p.detach().cuda(), n.detach().cuda(), margin, want_grad=False)
    assert none_a is None and loss2.item() == loss.item()


@pytest.mark.parametrize("B,n,margin", [(2, 8000, 1.0), (3, 5000, MARGIN), (2, 48000, 1.0)])  # last: T = 149, 3 attention tiles
def test_parameter_gradients_match_autograd(teng, sd_train, B, n, margin):
    A, P, N = _triplet_batch(B, n, seed=B)
    ref_loss, ref = O.triplet_step_grads(sd_train, A, P, N, margin)
    assert ref_loss.item() > 0  # some triplet is active, otherwise the test is vacuous
    loss, flat = _gpu_step_grads(teng, A, P, N, margin)
    assert abs(loss.item() - ref_loss.item()) < 2e-5
    got = teng.train_unflatten(flat)
    # per-tensor bound: 2e-4 of that tensor's largest gradient, plus 1e-6 of the largest gradient anywhere
    # (k_proj.bias has an exactly-zero gradient - softmax is shift invariant - so both sides hold rounding noise)
    top = max(v.abs().max().item() for v in ref.values())
    worst = ("", 0.0)
    for k, want in ref.items():
        err = (got[k] - want).abs().max().item() / (2e-4 * want.abs().max().item() + 1e-6 * top)
        if err > worst[1]:
            worst = (k, err)
    assert worst[1] < 1.0, worst
    # global direction: cosine over the whole parameter vector
    # freeze_convnet: True (the default): the conv feature extractor's slices of the gradient vector stay exactly zero
    for k, v in got.items():
        if k not in ref:
            assert "feature_extractor" in k and torch.count_nonzero(v).item() == 0, k
    w = torch.cat([(ref[k] if k in ref else torch.zeros(n)).reshape(-1) for k, _, n in teng.train_segments()]).double()
    gflat = flat.cpu().double()
    cos = (w @ gflat / (w.norm() * gflat.norm())).item()
    assert cos > 0.999999, cos


def test_gradients_accumulate_and_repeat_bit_identically(teng):
    A, P, N = _triplet_batch(2, 5000, seed=11)
    _, g1 = _gpu_step_grads(teng, A, P, N, 1.0)
    _, g2 = _gpu_step_grads(teng, A, P, N, 1.0)
    assert torch.equal(g1, g2)
    # a second backward without zero_grad doubles the gradient (A branch only here)
    teng.train_zero_grad()
    emb, layers, saved = teng.embed_train(A.cuda())
    d = torch.randn(2, 256, generator=torch.Generator().manual_seed(0)).cuda()
    teng.train_backward(A.cuda(), layers, saved, d)
    once = teng.train_read(1)
    teng.train_backward(A.cuda(), layers, saved, d)
    twice = teng.train_read(1)
    assert torch.allclose(twice, 2 * once, rtol=1e-6, atol=1e-12)


def test_adam_matches_torch_and_weights_follow(built_lib, sd_train):
    """Three optimiser steps with fixed gradients vs torch.optim.Adam (two learning rates, train_triplet.py:98-107);
    afterwards the forward and the gradients of the updated model match the oracle on the updated state dict
    (fused q/k/v, weight-normed pos-conv and the backward's transposed copies were all rebuilt)."""
    from nomad_amd.engine import Engine
    eng = Engine({k: v.clone() for k, v in sd_train.items()}, 0)
    try:
        eng.train_enable()
        lr, lr_body = 1e-2, 1e-3  # large steps so that stale derived weights would be obvious
        opt, params = O.make_adam(sd_train, lr=lr, lr_body=lr_body)
        g = torch.Generator().manual_seed(5)
        for step in range(3):
            grads = {k: torch.randn(p.shape, generator=g) * 1e-3 for k, p in params.items()}
            for k, p in params.items():
                p.grad = grads[k].clone()
            opt.step()
            eng.train_write(1, eng.train_flatten(grads, fill=0.0))     # frozen conv feature extractor: zero gradient
            eng.adam_step(lr_body, lr)
        got = eng.train_unflatten(eng.train_read(0))
        for k, p in params.items():
            assert (got[k] - p.detach()).abs().max().item() < 2e-6 * max(1.0, p.detach().abs().max().item()), k
        new_sd = eng.train_state_dict()
        assert set(new_sd) == set(sd_train)
        wav = _triplet_batch(2, 6000, seed=2)[0]
        want = O.triplet_forward(new_sd, wav)
        have = eng.embed(wav.cuda()).cpu()
        assert (have - want).abs().max().item() < 2e-5
        A, P, N = _triplet_batch(2, 5000, seed=4)
        ref_loss, ref = O.triplet_step_grads(new_sd, A, P, N, 1.0)
        loss, flat = _gpu_step_grads(eng, A, P, N, 1.0)
        assert abs(loss.item() - ref_loss.item()) < 5e-5
        gg = eng.train_unflatten(flat)
        top = max(v.abs().max().item() for v in ref.values())
        for k, want_g in ref.items():
            assert (gg[k] - want_g).abs().max().item() < 5e-4 * want_g.abs().max().item() + 1e-6 * top, k
    finally:
        eng.close()


# ---- model.train(): dropout / attention dropout / dropout_input / LayerDrop -------------------------------------
def _gpu_step_grads_stoch(eng, A, P, N, margin, stochs):
    """Like _gpu_step_grads, with one regularisation setting per branch, re-set before each backward."""
    def apply(st):
        eng.train_set_stochastic(st.dropout, st.attention_dropout, st.dropout_input, st.seed, st.layer_mask)
    eng.train_zero_grad()
    outs = []
    for w, st in zip((A, P, N), stochs):
        apply(st)
        outs.append(eng.embed_train(w.cuda()))
    loss, da, dp, dn = eng.triplet_loss(outs[0][0], outs[1][0], outs[2][0], margin)
    for w, (emb, layers, saved), d, st in zip((A, P, N), outs, (da, dp, dn), stochs):
        apply(st)
        eng.train_backward(w.cuda(), layers, saved, d)
    eng.train_set_stochastic()  # back to eval-mode arithmetic
    return loss.cpu(), [o[0].cpu() for o in outs], eng.train_read(1)


@pytest.mark.parametrize("case", ["all", "residual_only", "attention_only", "input_only", "layerdrop_only"])
def test_train_mode_matches_oracle_with_the_same_masks(teng, sd_train, case):
    B, n, margin = 2, 6000, 1.0
    A, P, N = _triplet_batch(B, n, seed=21)
    kw = {"all": dict(dropout=0.1, attention_dropout=0.1, dropout_input=0.1),
          "residual_only": dict(dropout=0.2, attention_dropout=0.0, dropout_input=0.0),
          "attention_only": dict(dropout=0.0, attention_dropout=0.3, dropout_input=0.0),
          "input_only": dict(dropout=0.0, attention_dropout=0.0, dropout_input=0.25),
          "layerdrop_only": dict(dropout=0.0, attention_dropout=0.0, dropout_input=0.0)}[case]
    masks = (0xFFF, 0xFFF, 0xFFF) if case not in ("all", "layerdrop_only") else (0xFFF & ~(1 << 3), 0xFFE, 0x7FF & ~(1 << 6))
    stochs = [O.Stochastic(seed=(0x1234567 << 20) + 977 * i, layer_mask=m, **kw) for i, m in enumerate(masks)]
    ref_loss, ref = O.triplet_step_grads(sd_train, A, P, N, margin, stochs)
    loss, embs, flat = _gpu_step_grads_stoch(teng, A, P, N, margin, stochs)
    for e, w, st in zip(embs, (A, P, N), stochs):
        assert (e - O.triplet_forward(sd_train, w, st)).abs().max().item() < 2e-5
    assert abs(loss.item() - ref_loss.item()) < 5e-5
    got = teng.train_unflatten(flat)
    top = max(v.abs().max().item() for v in ref.values())
    worst = ("", 0.0)
    for k, want in ref.items():
        err = (got[k] - want).abs().max().item() / (2e-4 * want.abs().max().item() + 1e-6 * top)
        if err > worst[1]:
            worst = (k, err)
    assert worst[1] < 1.0, worst


def test_train_mode_differs_from_eval_and_is_seed_deterministic(teng):
    A = _triplet_batch(2, 6000, seed=5)[0].cuda()
    teng.train_set_stochastic()
    e_eval = teng.embed_train(A)[0].clone()
    assert torch.equal(e_eval, teng.embed(A))  # training-mode forward without regularisation == scoring forward
    teng.train_set_stochastic(0.1, 0.1, 0.1, seed=7)
    e1 = teng.embed_train(A)[0].clone()
    e2 = teng.embed_train(A)[0].clone()
    teng.train_set_stochastic(0.1, 0.1, 0.1, seed=8)
    e3 = teng.embed_train(A)[0].clone()
    teng.train_set_stochastic()
    assert torch.equal(e1, e2) and not torch.equal(e1, e3) and not torch.equal(e1, e_eval)
    assert torch.equal(teng.embed(A), e_eval)  # scoring is never regularised


def test_dropout_keep_rate(teng):
    """The counter-based generator drops about p of the elements at every site (statistics of the oracle's
    restated mask; the engine uses the same masks, as the parity test above shows)."""
    st = O.Stochastic(seed=99)
    for site in (0, 1, 2, 17, 37):
        m = st.mult(site, (4, 50, 768), 0.1)
        assert abs((m == 0).float().mean().item() - 0.1) < 0.005
    a, b = st.mult(3, (1000,), 0.5), st.mult(4, (1000,), 0.5)
    assert 0.35 < ((a == 0) == (b == 0)).float().mean().item() < 0.65  # sites are independent


# ---- the Training class (host-side mirror of train_triplet.py) ------------------------------------------------------
def _write_wav(path, x, sr=16000):
    import struct
    pcm = (np.clip(x, -1, 1) * 32767).astype("<i2").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVEfmt " +
                struct.pack("<IHHIIHH", 16, 1, 1, sr, sr * 2, 2, 16) + b"data" + struct.pack("<I", len(pcm)) + pcm)


def _toy_dataset(tmp_path, n_rows=6):
    import pandas as pd
    rng = np.random.RandomState(0)
    rows = []
    for i in range(n_rows):
        base = 0.1 * rng.randn(6000 + 700 * (i % 3))
        names = {}
        for role, noise in (("Anchor", 0.0), ("Positive", 0.01), ("Negative", 0.2)):
            name = f"/{role}_{i}.wav"
            _write_wav(str(tmp_path) + name, base + noise * rng.randn(base.size))
            names[role] = name
        rows.append(dict(db=1 + i % 2, **names))
    csv = str(tmp_path / "triplets.csv")
    pd.DataFrame(rows).to_csv(csv, index=False)
    return csv


def _config(tmp_path, csv, **over):
    cfg = dict(experiment_name="Training", out_dir="toy", training_script="src.training.train_triplet", root=str(tmp_path),
               train_df=csv, valid_df=csv, sampling_rate=16000, train_bs=3, val_bs=3, test_bs=1, lr=1e-4,
               lr_decay_factor=0.5, lr_decay_step=1, num_epochs=2, num_workers=0, emb_dim=256, patience=200,
               checkpoint_path="seeded", ssl_out_dim=768, margin=0.2, freeze_convnet=True, freeze_all=False,
               current_level=[1, 2], trim=True, eval_w2v=False)
    cfg.update(over)
    return cfg


def test_training_class_step_matches_torch_training_step(tmp_path):
    """One Training.train_step in eval-mode arithmetic == the reference's lines 121-131 executed by torch on the CPU
    oracle: zero_grad, three forwards, TripletMarginLoss, backward, Adam(1e-5 backbone / lr head)."""
    from nomad_amd.train import Training
    csv = _toy_dataset(tmp_path)
    tr = Training(_config(tmp_path, csv), regularisation=dict(dropout=0.0, attention_dropout=0.0, dropout_input=0.0,
                                                              encoder_layerdrop=0.0))
    try:
        sd = tr.engine.train_state_dict()
        A, P, N = next(iter(tr.valid_loader))
        assert A.shape[0] == 3 and A.dim() == 3 and A.shape == P.shape == N.shape  # zero padded to the batch maximum
        opt, params = O.make_adam(sd, lr=1e-4)
        sd_live = dict(sd)
        sd_live.update(params)
        loss_ref = torch.nn.TripletMarginLoss(margin=0.2)(*(O.triplet_forward(sd_live, w) for w in (A, P, N)))
        opt.zero_grad()
        loss_ref.backward()
        opt.step()
        loss = tr.train_step(A, P, N)
        assert abs(loss.item() - loss_ref.item()) < 2e-5
        got = tr.engine.train_unflatten(tr.engine.train_read(0))
        for k, p in params.items():
            # first Adam step moves every coordinate by ~lr * sign(g); compare the update itself
            upd_ref, upd = p.detach() - sd[k], got[k] - sd[k]
            lr_k = 1e-4 if k.startswith("embedding_layer") else 1e-5
            frac_bad = ((upd - upd_ref).abs() > 0.05 * lr_k).float().mean().item()
            assert frac_bad < 2e-3, (k, frac_bad)  # sign flips only where |g| is at rounding level
    finally:
        tr.engine.close()


def test_training_loop_runs_and_saves_a_loadable_checkpoint(tmp_path, monkeypatch):
    from nomad_amd.train import Training
    from nomad_amd.weights import load_checkpoint
    csv = _toy_dataset(tmp_path)
    monkeypatch.chdir(tmp_path)
    tr = Training(_config(tmp_path, csv, lr=1e-3))
    try:
        before = tr.eval()
        best = tr.training_loop()
        assert np.isfinite(best)
        assert tr.lr_scheduler.get_last_lr()[1] < 1e-3  # lr_decay_step = 1: ExponentialLR stepped
        path = os.path.join(tr.PATH_DIR, "best_model.pt")
        sd = load_checkpoint(path)  # same key layout as nomad_best_model.pt, loadable by the scoring path
        assert os.path.isfile(os.path.join(tr.PATH_DIR, "config.yaml"))
        changed = [k for k in O.trainable_keys(sd) if not torch.equal(sd[k], tr.engine._state_dict[k])]
        frozen_same = all(torch.equal(sd[k], tr.engine._state_dict[k]) for k in sd if "feature_extractor" in k)
        assert len(changed) > 150 and frozen_same
        after = tr.eval()
        assert after < before + 0.05  # two tiny epochs: the validation loss does not blow up
    finally:
        tr.engine.close()


# ---- A/P/N as one merged batch with per-branch LayerDrop ----------------------------------------------------------
@pytest.mark.parametrize("masks", [(0xFFF, 0xFFF, 0xFFF), (0xFFF & ~(1 << 3), 0xFFE, 0x7FF & ~(1 << 3)), (0x0F0, 0xF0F, 0xFFF),
                                   (0xFFE & ~(1 << 5), 0xFFE & ~(1 << 5), 0x7FE & ~(1 << 5))])  # last: layers 0 and 5 dropped by all
def test_merged_branches_match_oracle(teng, sd_train, masks):
    B, n, margin = (2, 6000, 1.0) if masks[0] != 0x0F0 else (2, 40000, 1.0)  # one case with several attention tiles (T = 124)
    A, P, N = _triplet_batch(B, n, seed=31)
    st = O.Stochastic(seed=(77 << 33) + 5, dropout=0.1, attention_dropout=0.1, dropout_input=0.1, branch_masks=masks)
    # oracle: one forward over the concatenated batch, loss on its three thirds
    sd = {k: v.clone() for k, v in sd_train.items()}
    keys = O.trainable_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    e = O.triplet_forward(sd, torch.cat([A, P, N]), st)
    ref_loss = torch.nn.TripletMarginLoss(margin=margin)(e[:B], e[B:2 * B], e[2 * B:])
    grads = torch.autograd.grad(ref_loss, [sd[k] for k in keys], allow_unused=True)
    ref = {k: (gk if gk is not None else torch.zeros_like(sd[k])) for k, gk in zip(keys, grads)}  # None: layer dropped by all
    # engine: merged batch, per-branch masks
    w = torch.cat([A, P, N]).cuda()
    teng.train_set_stochastic(st.dropout, st.attention_dropout, st.dropout_input, st.seed, 0xFFF)
    teng.train_set_branches(list(masks))
    try:
        emb, layers, saved = teng.embed_train(w)
        loss, da, dp, dn = teng.triplet_loss(emb[:B].contiguous(), emb[B:2 * B].contiguous(), emb[2 * B:].contiguous(), margin)
        teng.train_zero_grad()
        teng.train_backward(w, layers, saved, torch.cat([da, dp, dn]))
        flat = teng.train_read(1)
    finally:
        teng.train_set_branches(None)
        teng.train_set_stochastic()
    assert (emb.cpu() - e.detach()).abs().max().item() < 2e-5
    assert abs(loss.item() - ref_loss.item()) < 5e-5
    got = teng.train_unflatten(flat)
    top = max(v.abs().max().item() for v in ref.values())
    for k, want in ref.items():
        assert (got[k] - want).abs().max().item() < 2e-4 * want.abs().max().item() + 1e-6 * top, k
    dropped_by_all = [l for l in range(12) if not any((m >> l) & 1 for m in masks)]
    for l in dropped_by_all:  # a layer no branch ran gets exactly no gradient
        assert all(float(got[k].abs().max()) == 0.0 for k in got if f"encoder.layers.{l}." in k)


def test_merged_equals_separate_calls_in_eval_arithmetic(teng):
    """Without regularisation the merged batch and the reference's three calls are the same function: embeddings
    bit-identical (batch invariance), gradients equal up to the summation order of the weight-gradient GEMMs."""
    B = 2
    A, P, N = _triplet_batch(B, 6000, seed=41)
    _, g_sep = _gpu_step_grads(teng, A, P, N, 1.0)
    w = torch.cat([A, P, N]).cuda()
    teng.train_set_branches([0xFFF] * 3)
    try:
        emb, layers, saved = teng.embed_train(w)
        for i, x in enumerate((A, P, N)):
            assert torch.equal(emb[i * B:(i + 1) * B], teng.embed(x.cuda()))
        loss, da, dp, dn = teng.triplet_loss(emb[:B].contiguous(), emb[B:2 * B].contiguous(), emb[2 * B:].contiguous(), 1.0)
        teng.train_zero_grad()
        teng.train_backward(w, layers, saved, torch.cat([da, dp, dn]))
        g_m = teng.train_read(1)
    finally:
        teng.train_set_branches(None)
    assert (g_m - g_sep).abs().max().item() < 1e-5 * g_sep.abs().max().item()


def test_training_api_error_paths(engine, teng):
    """Same conventions as the rest of the C ABI: negative status + message, nothing thrown inside the library."""
    from nomad_amd._lib import NomadHipError
    wav = _triplet_batch(2, 5000, seed=1)[0].cuda()
    with pytest.raises(NomadHipError, match="nomad_train_enable"):   # the shared scoring engine never enabled training
        engine.train_zero_grad()
    with pytest.raises(NomadHipError, match="probabilities"):
        teng.train_set_stochastic(dropout=1.0)
    with pytest.raises(NomadHipError, match="branches"):
        teng.train_set_branches([0xFFF] * 5)
    teng.train_set_branches([0xFFF] * 3)
    try:
        with pytest.raises(NomadHipError, match="equal branches"):  # 2 clips cannot be 3 equal branches
            teng.embed_train(wav)
    finally:
        teng.train_set_branches(None)
    emb, layers, saved = teng.embed_train(wav)
    with pytest.raises(NomadHipError, match="workspace|saved"):
        teng.lib.nomad_train_backward  # noqa: B018 (exists)
        from nomad_amd import _lib
        _lib.check(teng.lib.nomad_train_backward(teng.ctx, wav.data_ptr(), 2, 5000, layers.data_ptr(), saved.data_ptr(),
                                                 16, emb.data_ptr(), saved.data_ptr(), 16, None), "nomad_train_backward")
    with pytest.raises(ValueError):
        teng.train_write(1, torch.zeros(7, device="cuda"))


def test_optimisation_actually_learns(built_lib, sd_train):
    """Overfit four fixed triplets that the untrained model gets WRONG (positive = unrelated clip, negative = the anchor
    plus a little noise): with the reference's optimiser at larger learning rates the triplet loss must fall steadily."""
    from nomad_amd.engine import Engine
    from nomad_amd.train import ExponentialLR, Training
    g = torch.Generator().manual_seed(123)
    A = (0.1 * torch.randn(4, 1, 8000, generator=g)).clamp(-1, 1)
    P = (0.1 * torch.randn(4, 1, 8000, generator=g) * torch.linspace(0.2, 2.0, 8000)).clamp(-1, 1)
    N = (A + 0.03 * torch.randn(4, 1, 8000, generator=g)).clamp(-1, 1)
    eng = Engine({k: v.clone() for k, v in sd_train.items()}, 0)
    try:
        tr = Training(dict(experiment_name="overfit", checkpoint_path="seeded", margin=0.2), engine=eng,
                      regularisation=dict(dropout=0.0, attention_dropout=0.0, dropout_input=0.0, encoder_layerdrop=0.0))
        tr.margin, tr.lr_scheduler = 0.2, ExponentialLR([2e-5, 2e-3], 1.0)
        losses = [tr.train_step(A, P, N).item() for _ in range(40)]
        assert losses[0] > 0.2                      # wrong way round at the start: d(a,p) > d(a,n)
        assert all(torch.isfinite(torch.tensor(losses)))
        assert min(losses[-5:]) < 0.5 * losses[0], losses[::5]
        # the scoring path sees the fine-tuned model: the ordering of the distances has improved for every triplet
        ea, ep, en = (eng.embed(x.squeeze(1).cuda()) for x in (A, P, N))
        gap_after = (ea - ep).norm(dim=1) - (ea - en).norm(dim=1)
        assert float(gap_after.mean()) < losses[0] - 0.2
    finally:
        eng.close()


def test_train_step_with_branches_of_different_lengths(built_lib, sd_train):
    """The reference pads anchor, positive and negative batches separately, so their lengths may differ: then the step
    falls back to three forward/backward calls - same numbers as torch on the CPU oracle."""
    from nomad_amd.engine import Engine
    from nomad_amd.train import ExponentialLR, Training
    g = torch.Generator().manual_seed(9)
    A, P, N = [(0.1 * torch.randn(2, 1, n, generator=g)).clamp(-1, 1) for n in (6000, 7300, 5200)]
    eng = Engine({k: v.clone() for k, v in sd_train.items()}, 0)
    try:
        tr = Training(dict(experiment_name="x", checkpoint_path="seeded", margin=1.0), engine=eng,
                      regularisation=dict(dropout=0.0, attention_dropout=0.0, dropout_input=0.0, encoder_layerdrop=0.0))
        tr.margin, tr.lr_scheduler = 1.0, ExponentialLR([1e-5, 1e-4], 0.99)
        ref_loss, ref = O.triplet_step_grads(sd_train, A.squeeze(1), P.squeeze(1), N.squeeze(1), 1.0)
        loss = tr.train_step(A, P, N)
        assert abs(loss.item() - ref_loss.item()) < 2e-5
        got = eng.train_unflatten(eng.train_read(1))  # gradients of the step just taken are still in the vector
        top = max(v.abs().max().item() for v in ref.values())
        for k, want in ref.items():
            assert (got[k] - want).abs().max().item() < 2e-4 * want.abs().max().item() + 1e-6 * top, k
    finally:
        eng.close()


# ---- freeze_convnet: False (train_triplet.py:71-73): the conv feature extractor trains too --------------------------------
@pytest.mark.parametrize("B,n,fgm", [(2, 8000, 0.1), (3, 5000, 1.0), (2, 33000, 0.1)])   # last: L_1 = 3299 > 6 column blocks of 512
def test_convnet_gradients_match_autograd(teng, sd_train, B, n, fgm):
    """Every parameter gradient with the conv feature extractor trainable - conv1..6 weights (dW as split-K GEMMs over
    transposed im2col operands), conv0 weight and the GroupNorm affine - against the oracle's autograd, including
    fairseq's feature_grad_mult on everything that enters the extractor; the other gradients must not change."""
    A, P, N = _triplet_batch(B, n, seed=B + 40)
    ref_loss, ref = O.triplet_step_grads(sd_train, A, P, N, 1.0, freeze_convnet=False, feature_grad_mult=fgm)
    assert ref_loss.item() > 0
    _, flat_frozen = _gpu_step_grads(teng, A, P, N, 1.0)
    old = teng.feature_grad_mult
    teng.train_set_convnet(True)
    teng.feature_grad_mult = fgm
    try:
        loss, flat = _gpu_step_grads(teng, A, P, N, 1.0)
        _, flat2 = _gpu_step_grads(teng, A, P, N, 1.0)
    finally:
        teng.train_set_convnet(False)
        teng.feature_grad_mult = old
    assert torch.equal(flat, flat2)                                  # deterministic
    assert abs(loss.item() - ref_loss.item()) < 2e-5
    got, frozen = teng.train_unflatten(flat), teng.train_unflatten(flat_frozen)
    top = max(v.abs().max().item() for k, v in ref.items() if "feature_extractor" in k)
    worst = ("", 0.0)
    for k, want in ref.items():
        if "feature_extractor" in k:
            assert want.abs().max().item() > 0, k
            err = (got[k] - want).abs().max().item() / (5e-5 * want.abs().max().item() + 1e-6 * top)   # measured: 1e-5
            if err > worst[1]:
                worst = (k, err)
        else:
            assert torch.equal(got[k], frozen[k]), k                  # same kernels, same order as with the extractor frozen
    print(f"conv feature extractor gradients B={B} n={n} fgm={fgm}: worst {worst[0]} at {worst[1]:.2f} of its bound")
    assert worst[1] < 1.0, worst


def test_convnet_weights_follow_the_master_copy(built_lib, sd_train):
    """An Adam step with the extractor trainable moves the conv weights, and the forward / dX kernels' derived copies
    follow: the embedding afterwards equals the oracle's on the updated state dict."""
    from nomad_amd.engine import Engine
    eng = Engine({k: v.clone() for k, v in sd_train.items()}, 0)
    try:
        eng.train_enable()
        eng.train_set_convnet(True)
        A, P, N = _triplet_batch(2, 8000, seed=8)
        _gpu_step_grads(eng, A, P, N, 1.0)
        eng.adam_step(1e-3, 1e-3)
        new_sd = eng.train_state_dict()
        moved = [k for k in new_sd if "feature_extractor" in k and not torch.equal(new_sd[k], sd_train[k])]
        assert len(moved) == 9, moved                                  # 7 conv weights + GroupNorm weight and bias
        with torch.no_grad():
            want = O.triplet_forward(new_sd, A)
        got = eng.embed(A.cuda()).cpu()
        assert (got - want).abs().max().item() < 2e-5
        assert (eng.embed_bf16x3(A.cuda()).cpu() - want).abs().max().item() < 2e-5     # the split copies were rebuilt too
        # the dX chain's transposed conv weights: gradient w.r.t. the waveform through the updated extractor
        g32 = _wav_grad(eng, A)
        ref = _wav_grad_oracle(new_sd, A, eng.feature_grad_mult)
        assert (g32 - ref).abs().max().item() < 2e-3 * ref.abs().max().item()
    finally:
        eng.close()


def _wav_grad(eng, wav):
    """d sum(emb * fixed) / d wav through nomad_embed_train + nomad_embed_backward."""
    g = torch.Generator().manual_seed(1)
    d = torch.randn(wav.shape[0], 256, generator=g).cuda()
    emb, layers, saved = eng.embed_train(wav.cuda())
    return eng.embed_backward(wav.cuda(), layers, saved, None, d).cpu()


def _wav_grad_oracle(sd, wav, fgm):
    g = torch.Generator().manual_seed(1)
    d = torch.randn(wav.shape[0], 256, generator=g)
    w = wav.clone().requires_grad_(True)
    emb = O.triplet_forward(sd, w, None, fgm)
    (emb * d).sum().backward()
    return w.grad


def test_freeze_convnet_false_through_the_training_class(tmp_path):
    """Training(config with freeze_convnet: False).train_step: the extractor's tensors change too, and - as in the
    reference, which does not overwrite its optimiser then (train_triplet.py:96-107) - every parameter runs at `lr`."""
    from nomad_amd.train import Training
    csv = _toy_dataset(tmp_path)
    tr = Training(_config(tmp_path, csv, freeze_convnet=False), regularisation=dict(dropout=0.0, attention_dropout=0.0,
                                                                                   dropout_input=0.0, encoder_layerdrop=0.0))
    try:
        assert tr.train_convnet and tr.lr_scheduler.get_last_lr() == [1e-4, 1e-4]
        before = tr.engine.train_state_dict()
        A, P, N = next(iter(tr.valid_loader))
        assert tr.train_step(A, P, N).item() > 0
        after = tr.engine.train_state_dict()
        changed = {k for k in before if not torch.equal(before[k], after[k])}
        assert {k for k in before if "feature_extractor" in k} <= changed
        assert "ssl_model.encoder.layers.0.fc1.weight" in changed and "embedding_layer.1.weight" in changed
        k = "ssl_model.feature_extractor.conv_layers.3.0.weight"       # first Adam step: |update| ~ lr wherever g != 0
        upd = (after[k] - before[k]).abs()
        assert 0.5e-4 < upd.max().item() < 1.5e-4
    finally:
        tr.engine.close()


# ---- freeze_all: True (train_triplet.py:76-79) ------------------------------------------------------------------------
FREEZE_ALL_TRAINABLE = ("ssl_model.post_extract_proj.weight", "ssl_model.post_extract_proj.bias", "ssl_model.layer_norm.weight",
                        "ssl_model.layer_norm.bias", "embedding_layer.1.weight", "embedding_layer.1.bias")


def test_freeze_all_trains_only_what_the_reference_leaves_trainable(teng, sd_train):
    """The reference's freeze_all freezes ssl_model.feature_extractor and ssl_model.encoder: post_extract_proj, the
    feature LayerNorm and embedding_layer stay trainable and their gradients still flow THROUGH the frozen encoder.
    Those gradients must equal the unfrozen run's (and the oracle's); every encoder gradient must be exactly zero; an
    Adam step then moves only the trainable tensors."""
    A, P, N = _triplet_batch(2, 8000, seed=3)
    ref_loss, ref = O.triplet_step_grads(sd_train, A, P, N, 1.0)
    _, flat_full = _gpu_step_grads(teng, A, P, N, 1.0)
    full = teng.train_unflatten(flat_full)
    teng.train_set_frozen(True)
    try:
        loss, flat = _gpu_step_grads(teng, A, P, N, 1.0)
        got = teng.train_unflatten(flat)
    finally:
        teng.train_set_frozen(False)
    assert abs(loss.item() - ref_loss.item()) < 2e-5
    top = max(v.abs().max().item() for v in ref.values())
    for k, v in got.items():
        if k in FREEZE_ALL_TRAINABLE:
            assert torch.equal(v, full[k]), k                       # the same kernels in the same order
            assert (v - ref[k]).abs().max().item() < 2e-4 * ref[k].abs().max().item() + 1e-6 * top, k
            assert v.abs().max().item() > 0, k
        else:
            assert torch.count_nonzero(v).item() == 0, k


def test_freeze_all_through_the_training_class(tmp_path):
    """Training(config with freeze_all: True).train_step: only the six trainable tensors change."""
    from nomad_amd.train import Training
    csv = _toy_dataset(tmp_path)
    tr = Training(_config(tmp_path, csv, freeze_all=True), regularisation=dict(dropout=0.0, attention_dropout=0.0,
                                                                              dropout_input=0.0, encoder_layerdrop=0.0))
    try:
        before = tr.engine.train_state_dict()
        A, P, N = next(iter(tr.valid_loader))
        loss = tr.train_step(A, P, N)
        assert loss.item() > 0
        after = tr.engine.train_state_dict()
        changed = {k for k in before if not torch.equal(before[k], after[k])}
        assert changed and changed <= set(FREEZE_ALL_TRAINABLE), changed
        assert {"embedding_layer.1.weight", "ssl_model.post_extract_proj.weight"} <= changed
    finally:
        tr.engine.close()
