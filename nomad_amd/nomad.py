"""Drop-in ``Nomad`` surface over the MI355X HIP engine.

Mirrors /root/reference/src/nomad_audio/nomad.py: same class and method names, argument meaning,
return values, CSV side effects and exception messages -

* ``Nomad(device=None)``                 nomad.py:36-80
* ``predict(mode, nmr, deg, results_path) -> (df_avg_nomad, df_dm)``   nomad.py:82-140
* ``forward(estimate, clean) -> loss``   nomad.py:142-146
* ``get_embeddings(path)`` / ``get_embeddings_csv(model, file_names)``  nomad.py:148-189
* ``load_processing(filepath, target_sr, trim)``                       nomad.py:192-212
* ``TripletModel`` / ``LossNetLayers`` / ``NomadLoss``                 nomad.py:214-282

What differs, deliberately:

* all arithmetic (wav2vec 2.0 BASE backbone, head, distances, L1 loss) runs in libnomad_hip.so on
  a gfx950 GPU; there is NO CPU path - ``device='cpu'`` raises.
* nothing is downloaded at import time (no network in production clusters); the checkpoint is read
  from ``pt-models/nomad_best_model.pt`` (the reference's location, nomad.py:28) or from
  ``$NOMAD_CHECKPOINT``; ``Nomad(weights=...)`` accepts a state dict or ``'seeded'``.
* ``get_embeddings_csv`` packs clips of ANY lengths into ragged batches (one launch sequence, no
  padding in the arithmetic) instead of the reference's batch-1 loop with a device sync per clip
  (nomad.py:171-183); every clip sees exactly the arithmetic it sees at batch 1 (bit-identical).
* the distance matrix is computed on the GPU (float64, difference form) instead of SciPy.
"""
from __future__ import annotations

import os
from datetime import datetime
from typing import Dict, List, Union

import numpy as np
import pandas as pd
import torch

from . import wavio
from .dist import partition
from .engine import Engine
from .weights import find_checkpoint, find_feature_grad_mult, load_checkpoint, seeded_state_dict

SSL_OUT_DIM = 768
EMB_DIM = 256


def _resolve_device(device) -> int:
    if device is None:
        # the reference falls back to the CPU here (nomad.py:40-43); this build has no CPU path, so a host without a GPU gets the
        # same clear message an explicit device='cpu' gets, not a failure somewhere inside nomad_create
        if not torch.cuda.is_available():
            raise RuntimeError("NOMAD (MI355X build) runs on a HIP GPU only and no GPU is visible to PyTorch "
                               "(device=None resolves to 'cuda'; there is no CPU path)")
        device = "cuda"
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError(f"NOMAD (MI355X build) runs on a HIP GPU only; device={device!r} is not supported "
                           "(there is no CPU path)")
    return dev.index if dev.index is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)


class TripletModel:
    """``TripletModel.forward(wav, lengths=None)`` (nomad.py:224-231) on the engine."""

    def __init__(self, engine: Engine):
        self.engine = engine

    def eval(self):
        return self

    def __call__(self, wav: torch.Tensor, lengths=None) -> torch.Tensor:
        return self.forward(wav, lengths)

    def forward(self, wav: torch.Tensor, lengths=None) -> torch.Tensor:
        return self.engine.embed(wav.to(self.engine.device, torch.float32).contiguous())


def _takes_bf16x3(precision: str, wav: torch.Tensor) -> bool:
    return precision == "bf16x3" and wav.numel() >= BF16X3_MIN_SAMPLES


class LossNetLayers:
    """``LossNetLayers.forward(wav)`` (nomad.py:243-258): 12 layer outputs (B,T,768) + embedding.

    Owns its own ``Linear(768,256)`` like the reference (nomad.py:238-241) - freshly initialised and
    never loaded from the checkpoint - exposed as ``embedding_weight`` / ``embedding_bias`` so callers
    that need reproducibility can set it.
    """

    def __init__(self, engine: Engine, ssl_out_dim: int = SSL_OUT_DIM, emb_dim: int = EMB_DIM, precision: str = "fp32"):
        self.engine = engine
        self.precision = precision     # "bf16x3": batches of at least BF16X3_MIN_SAMPLES samples take the split-bf16 forward
        lin = torch.nn.Linear(ssl_out_dim, emb_dim)
        self.embedding_weight = lin.weight.detach().to(engine.device).contiguous()
        self.embedding_bias = lin.bias.detach().to(engine.device).contiguous()

    def __call__(self, wav):
        return self.forward(wav)

    def forward(self, wav: torch.Tensor) -> List[torch.Tensor]:
        wav = wav.to(self.engine.device, torch.float32).contiguous()
        fwd = self.engine.embed_bf16x3 if _takes_bf16x3(self.precision, wav) else self.engine.embed
        emb, layers = fwd(wav, head=(self.embedding_weight, self.embedding_bias), want_layers=True)
        return [layers[i] for i in range(12)] + [emb]


class NomadLoss:
    """``NomadLoss.forward(nomad_ref, nomad_test)`` (nomad.py:267-282): sum of 13 L1 means."""

    def __init__(self, engine: Engine):
        self.engine = engine
        self.L = 13
        self.only_embedding = False

    def eval(self):
        return self

    def __call__(self, nomad_ref, nomad_test):
        return self.forward(nomad_ref, nomad_test)

    def forward(self, nomad_ref, nomad_test) -> torch.Tensor:
        if self.only_embedding:
            # the reference's other branch (nomad.py:270-273) reads element 13 of the lists - one past what LossNetLayers
            # returns, an IndexError there and here; with longer lists it is the L1 distance of that entry alone
            ref, test = nomad_ref[13], nomad_test[13]
            return (test - ref).abs().mean()
        ref_layers = _stack_layers(nomad_ref[:12])
        test_layers = _stack_layers(nomad_test[:12])
        return self.engine.l1_loss(test_layers, ref_layers, nomad_test[12].contiguous(), nomad_ref[12].contiguous())


def _stack_layers(layers) -> torch.Tensor:
    """The 12 tensors LossNetLayers returns are views of one (12,B,T,768) buffer: reuse it."""
    base = layers[0]._base if layers[0]._base is not None else None
    if base is not None and base.dim() == 4 and base.shape[0] == 12 and all(
            l._base is base and l.data_ptr() == base[i].data_ptr() for i, l in enumerate(layers)):
        return base
    return torch.stack([l.contiguous() for l in layers])


class _NomadLossFn(torch.autograd.Function):
    """loss = NomadLoss(LossNetLayers(clean), LossNetLayers(estimate)); backward = d loss / d estimate and, when ``clean`` requires a
    gradient too, d loss / d clean - the reference's ``forward`` (nomad.py:142-146) is differentiable in both arguments."""

    @staticmethod
    def forward(ctx, estimate, clean, nomad):
        eng = nomad.engine
        head = (nomad.lossnet_layers.embedding_weight, nomad.lossnet_layers.embedding_bias)
        est = estimate.detach().to(eng.device, torch.float32).contiguous()
        cln = clean.detach().to(eng.device, torch.float32).contiguous()
        need_grad = estimate.requires_grad
        need_clean_grad = clean.requires_grad
        # the two forwards are independent: at training batch sizes (32 x 1 s) one of them fills less than half
        # of the chip, so the clean branch runs concurrently on a side stream with its own workspace
        cur = torch.cuda.current_stream(eng.device)
        # precision="bf16x3": the branches that carry no gradient (always `clean`; `estimate` too under no_grad) run
        # the split-bf16 forward (layer outputs within ~1e-5 of fp32); the branch that is differentiated stays on fp32 BUFFERS
        # (embed_train), its GEMM products in whatever Engine.gemm_precision says - "bf16x3" when the Nomad was made with
        # precision="bf16x3", exact fp32 otherwise
        fwd = eng.embed_bf16x3 if _takes_bf16x3(nomad.precision, cln) else eng.embed
        saved_c = None
        if need_clean_grad:
            # the uncommon case (the speech-enhancement example differentiates `estimate` only, nomad_loss_test.py:69): the clean branch
            # keeps its activations too, on the caller's stream - the training-mode forward has one workspace per context
            c_emb, c_layers, saved_c = eng.embed_train(cln, head)
        else:
            side = eng.side_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                c_emb, c_layers = fwd(cln, head=head, want_layers=True, side=True)
        if need_grad:
            e_emb, e_layers, saved = eng.embed_train(est, head)
        else:
            e_emb, e_layers = fwd(est, head=head, want_layers=True)
            saved = None
        if not need_clean_grad:
            cur.wait_stream(side)
            for t in (c_emb, c_layers, cln):
                t.record_stream(cur)
        loss = eng.l1_loss(e_layers, c_layers, e_emb, c_emb)
        if need_grad or need_clean_grad:
            ctx.nomad = nomad
            ctx.shapes = (estimate.shape, clean.shape)
            ctx.save_for_backward(est, e_layers, e_emb, cln, c_layers, c_emb, saved, saved_c)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        est, e_layers, e_emb, cln, c_layers, c_emb, saved, saved_c = ctx.saved_tensors
        eng = ctx.nomad.engine
        head = (ctx.nomad.lossnet_layers.embedding_weight, ctx.nomad.lossnet_layers.embedding_bias)
        dwav = dcln = None
        if saved is not None and ctx.needs_input_grad[0]:
            dl, de = eng.l1_loss_backward(e_layers, c_layers, e_emb, c_emb, grad_out)
            dwav = eng.embed_backward(est, e_layers, saved, dl, de, head).reshape(ctx.shapes[0])
        if saved_c is not None and ctx.needs_input_grad[1]:
            # |e - c| is symmetric: the gradient with respect to the clean side is the same kernel with the arguments swapped
            dl, de = eng.l1_loss_backward(c_layers, e_layers, c_emb, e_emb, grad_out)
            dcln = eng.embed_backward(cln, c_layers, saved_c, dl, de, head).reshape(ctx.shapes[1])
        return dwav, dcln, None


class GraphedLoss:
    """``nomad.forward`` + its backward to ``estimate`` for ONE input shape, captured once as a HIP graph and replayed per step.

    A configs[3] training step (2 x (32,1,16384)) is ~440 kernel launches of 5-90 us: issued one by one, the host is still launching
    when the first two thirds of the step's GPU work are done (profiles/r04_c4_posconv_splitk_ab.txt).  A training loop with a fixed
    batch shape can capture the launch sequence once (``torch.cuda.graph``; every entry point on the path is capture-safe: the
    library allocates nothing and queries no event after the first call of a shape) and replay it - same kernels, same order,
    same bits (``tests/test_gpu_backward.py::test_graphed_loss_replays_bit_identically``).

    ``loss = graphed(estimate, clean)`` is differentiable with respect to ``estimate`` like ``nomad.forward``: the replay computes the
    loss AND d loss / d estimate; the autograd node hands the stored gradient (times the incoming one) on.

    Restrictions (checked): a captured graph holds its kernel arguments BY VALUE - the dropout / LayerDrop masks, the step counter and
    the addresses of ``lossnet_layers.embedding_weight`` / ``embedding_bias`` of the moment of capture.  So capture is refused while the
    engine has dropout or LayerDrop switched on (``Engine.train_set_stochastic`` / ``train_set_branches``: every replay would reuse one
    set of masks), and ``step`` raises when the head tensors have been replaced since (the graph would read freed memory) or
    stochastic mode has been switched on; "same bits as eager" is a statement about the frozen, eval-mode loss path only."""

    def _state(self):
        ll = self.nomad.lossnet_layers
        return (ll.embedding_weight.data_ptr(), ll.embedding_bias.data_ptr())

    def _check_not_stochastic(self, what: str):
        eng = self.nomad.engine
        if getattr(eng, "stochastic", False) or getattr(eng, "stochastic_branches", False):
            raise RuntimeError(f"GraphedLoss: {what} while the engine has dropout / LayerDrop switched on - a HIP graph replays the masks "
                               "and the step counter of its capture; use nomad.forward() for stochastic passes")

    def __init__(self, nomad: "Nomad", estimate: torch.Tensor, clean: torch.Tensor, warmup: int = 3):
        self.nomad = nomad
        self._check_not_stochastic("capture")
        # the graph reads the head through these addresses: keep the tensors alive and notice a replacement
        self._head = (nomad.lossnet_layers.embedding_weight, nomad.lossnet_layers.embedding_bias)
        self._head_ptrs = self._state()
        dev = nomad.engine.device
        self._est = estimate.detach().to(dev, torch.float32).clone().requires_grad_(True)
        self._cln = clean.detach().to(dev, torch.float32).clone()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):           # warm-up off the capture: workspaces, kernel attributes, autograd buffers
            for _ in range(max(1, warmup)):
                self._est.grad = None
                nomad.forward(self._est, self._cln).backward()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self._graph = torch.cuda.CUDAGraph()
        self._est.grad = None
        with torch.cuda.graph(self._graph):
            self._loss = nomad.forward(self._est, self._cln)
            self._loss.backward()
        self._grad = self._est.grad

    def step(self, estimate: torch.Tensor, clean: torch.Tensor):
        """-> (loss, d loss / d estimate): views of the graph's static outputs, valid until the next step."""
        if estimate.shape != self._est.shape or clean.shape != self._cln.shape:
            raise ValueError(f"GraphedLoss was captured for {tuple(self._est.shape)} / {tuple(self._cln.shape)}")
        self._check_not_stochastic("replay")
        if self._state() != self._head_ptrs:
            raise RuntimeError("GraphedLoss: lossnet_layers.embedding_weight / embedding_bias were replaced after capture (the graph holds "
                               "their old addresses); update them in place (copy_) or capture a new graph with nomad.graphed_loss()")
        with torch.no_grad():
            self._est.copy_(estimate)
            self._cln.copy_(clean)
        self._graph.replay()
        return self._loss, self._grad

    def __call__(self, estimate: torch.Tensor, clean: torch.Tensor) -> torch.Tensor:
        return _GraphedLossFn.apply(estimate, clean, self)


class _GraphedLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, estimate, clean, graphed):
        loss, grad = graphed.step(estimate.detach(), clean.detach())
        ctx.save_for_backward(grad.clone())
        ctx.shape = estimate.shape
        return loss.clone()

    @staticmethod
    def backward(ctx, grad_out):
        (g,) = ctx.saved_tensors
        return (g * grad_out).reshape(ctx.shape), None, None


# precision="bf16x3": from about 2 500 frames per batch (12 clips of 4 s, 50 of 1 s) the split-storage forward (256 x 256 tiles,
# embed_bf16x3) is the fastest; below that the batch stays on fp32 buffers with the same three-product arithmetic in its
# small-tile GEMMs (Engine.gemm_precision = "bf16x3").  Measured on MI355X (tools/bench_small_batch.py,
# profiles/r03_small_batch.jsonl; fp32 / x3 products on fp32 buffers / split storage, ms): 8 clips of 4 s 5.96 / 4.33 / 5.32,
# 16 clips 9.44 / 6.64 / 5.84, 32 clips of 1 s (config C4's branch) 5.75 / 4.18 / 5.22, 64 clips of 1 s 9.15 / 6.38 / 5.77.
BF16X3_MIN_SAMPLES = 800_000


def _write_rounded_csv(df: pd.DataFrame, path: str) -> None:
    """``df.to_csv(path, index=False)`` for the scores table (first column = labels, the rest 3-decimal floats), byte for
    byte, without pandas' per-cell float formatting: 10 000 x 1 000 scores take pandas 6-9 s to write and this 1 s.
    A value rounded to 3 decimals is k / 1000 for an integer k, and pandas prints a float by its shortest round-trip
    repr, so a table of ``repr(k / 1000.0)`` covers every cell.  Anything else (NaN, values off the grid, huge
    values, labels that need CSV quoting) goes through pandas itself."""
    vals = df.iloc[:, 1:].to_numpy(dtype=np.float64, copy=False) if df.shape[1] > 1 else np.zeros((len(df), 0))
    labels = [str(x) for x in df.iloc[:, 0].tolist()]
    header = [str(c) for c in df.columns]
    special = set(',"\r\n')
    ok = (vals.size > 0 and bool(np.isfinite(vals).all()) and not any(special & set(x) for x in labels + header)
          and all(isinstance(x, str) for x in df.iloc[:, 0].tolist()))
    if ok:
        idx = np.rint(vals * 1000.0)
        ok = bool(idx.min() >= 0 and idx.max() <= 100000 and np.array_equal(idx / 1000.0, vals))
    if not ok:
        df.to_csv(path, index=False)
        return
    lut = np.array([repr(k / 1000.0) for k in range(int(idx.max()) + 1)], dtype=object)
    cells = lut[idx.astype(np.int64)]
    with open(path, "w", newline="") as f:
        f.write(",".join(header) + "\n")
        f.write("\n".join(labels[i] + "," + ",".join(cells[i]) for i in range(len(labels))))
        f.write("\n")


class _StagingRing:
    """``slots`` pinned host buffers reused round-robin by the packer thread (grown on demand, never per batch).

    A slot is rewritten only after the H2D copy that read it last has completed: the consumer records an event right
    after it enqueued the copy (``uploaded``), ``take`` waits for it.  With ``max_alive`` staged batches in flight and
    ``max_alive + 1`` slots that wait is already over in practice (the batch three back has had its results fetched)."""

    def __init__(self, slots: int, device=None):
        self.bufs = [None] * slots
        self.events = [None] * slots
        self.taken = 0
        self.device = device   # the engine's device: the H2D copy runs on ITS current stream, whatever device is current

    def take(self, rows: int, stride: int):
        slot = self.taken % len(self.bufs)
        self.taken += 1
        ev, self.events[slot] = self.events[slot], None
        if ev is not None:
            ev.synchronize()
        need = rows * stride
        if self.bufs[slot] is None or self.bufs[slot].numel() < need:
            self.bufs[slot] = None
            self.bufs[slot] = torch.empty(need, dtype=torch.float32, pin_memory=torch.cuda.is_available())
        return slot, self.bufs[slot][:need].view(rows, stride)

    def uploaded(self, slot: int) -> None:
        if torch.cuda.is_available():
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.events[slot] = ev


def _dist_info(group=None):
    """(world size, rank, use collectives) of the torch.distributed job this process belongs to - (1, 0, False) outside one.
    NOMAD_FORCE_COLLECTIVE=1 runs the collectives in a group of one rank too (how the RCCL path is exercised on a
    single-GPU box)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0, False
    world = dist.get_world_size(group)
    return world, dist.get_rank(group), world > 1 or os.environ.get("NOMAD_FORCE_COLLECTIVE") == "1"


def _staged_batches(paths, load, pack, max_batch_samples: int, decode_threads: int, max_alive: int, native_threads: int = 0,
                    target_sr: int = 16000, device=None):
    """Generator of (row indices, (staging buffer, lengths), uploaded) batches over ``paths`` in order, built ahead of
    the consumer; the consumer calls ``uploaded()`` once it has enqueued the H2D copy of the staging buffer.

    A packer thread groups consecutive files into batches of at most ``max_batch_samples`` samples (a longer file is a
    batch of its own; a batch also closes before rows x longest row would exceed twice that, which bounds the
    staging buffer when one long file sits among short ones).

    * ``native_threads > 0``: file headers are probed through the C ABI (``nomad_wav_probe``); the files it can decode are
      converted - and, at another sample rate, resampled to ``target_sr`` - by ``nomad_wav_read_rows`` on that many plain
      host threads, straight into a slot of a ring of pinned staging buffers - no per-file Python, no GIL, no second copy.
    * every other file (an encoding or a header the native reader does not take, an unreadable file) goes through
      ``load(path) -> (1, N) array`` on ``decode_threads`` Python worker threads with a bounded look-ahead, so results
      and exceptions are those of ``load``.  A batch without native files is packed by ``pack(list of 1-D arrays)``.

    At most ``max_alive`` packed batches exist between the packer and the consumer: the packer takes a token before
    it starts a batch, and the token returns when the consumer asks for the batch AFTER the next one (by then the
    consumer has fetched that batch's results).  Exceptions raised by ``load`` / ``pack`` re-raise in the consumer."""
    import queue
    import threading
    from concurrent.futures import ThreadPoolExecutor

    if not paths:
        return
    tokens = threading.Semaphore(max_alive)
    out: "queue.Queue" = queue.Queue()
    stop = threading.Event()
    lookahead = max(4 * decode_threads, 16)
    ring = _StagingRing(max_alive + 1, device) if native_threads > 0 else None
    probe_chunk = 2048

    def packer():
        try:
            with ThreadPoolExecutor(max_workers=max(1, decode_threads)) as pool:
                infos, fast = {}, {}          # index -> WavInfo / frames of the files the native reader takes
                probed = 0
                futs, scan = {}, 0            # index -> future of load(path); next file to look at for submission

                def probe_upto(upto):
                    nonlocal probed
                    while ring is not None and probed < min(upto, len(paths)):
                        chunk = [str(p) for p in paths[probed:probed + probe_chunk]]
                        inf, status = wavio.probe(chunk, native_threads)
                        for k in range(len(chunk)):
                            if status[k] == 0 and inf[k].sample_rate > 0 and inf[k].frames > 0:
                                infos[probed + k] = inf[k]
                                fast[probed + k] = (int(inf[k].frames) if inf[k].sample_rate == target_sr
                                                    else wavio.frames_at(inf[k], target_sr))
                        probed += len(chunk)

                def top_up(i):                 # keep up to `lookahead` Python decodes in flight, in file order
                    nonlocal scan
                    scan = max(scan, i)
                    while scan < len(paths) and len(futs) < lookahead and scan < i + probe_chunk:
                        probe_upto(scan + 1)
                        if scan not in fast:
                            futs[scan] = pool.submit(load, paths[scan])
                        scan += 1

                def flush(idxs, items, lens):
                    if not any(isinstance(it, int) for it in items):
                        return idxs, pack(items), (lambda: None)
                    stride = (max(lens) + 3) // 4 * 4
                    slot, host = ring.take(len(idxs), stride)
                    rows = [r for r, it in enumerate(items) if isinstance(it, int)]
                    wavio.read_rows([str(paths[items[r]]) for r in rows], [infos.pop(items[r]) for r in rows], rows, host,
                                    native_threads, target_sr)
                    for r, it in enumerate(items):
                        if not isinstance(it, int):
                            host[r, :lens[r]] = torch.as_tensor(it, dtype=torch.float32).reshape(-1)
                    return idxs, (host, lens), (lambda: ring.uploaded(slot))

                idxs, items, lens, total = [], [], [], 0
                tokens.acquire()
                for i in range(len(paths)):
                    if stop.is_set():
                        return
                    top_up(i)
                    if i in fast:
                        item, n = i, fast.pop(i)                  # native: converted when the batch is flushed
                    else:
                        w = futs.pop(i).result()
                        item = w[0] if getattr(w, "ndim", 1) == 2 else w
                        n = int(item.shape[0])
                    if idxs and (total + n > max_batch_samples or (len(idxs) + 1) * max(max(lens), n) > 2 * max_batch_samples):
                        out.put(flush(idxs, items, lens))
                        idxs, items, lens, total = [], [], [], 0
                        tokens.acquire()
                    idxs.append(i)
                    items.append(item)
                    lens.append(n)
                    total += n
                out.put(flush(idxs, items, lens))
            out.put(None)
        except BaseException as e:  # noqa: BLE001 - handed to the consumer
            out.put(e)

    th = threading.Thread(target=packer, name="nomad-packer", daemon=True)
    th.start()
    handed = 0
    try:
        while True:
            item = out.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            handed += 1
            if handed >= max_alive:      # the consumer is done with batch (handed - max_alive + 1): its slot is free again
                tokens.release()
            yield item
    finally:
        stop.set()
        for _ in range(max_alive + 1):   # unblock a packer waiting for a slot
            tokens.release()
        th.join(timeout=30)


class Nomad:
    def __init__(self, device=None, weights: Union[None, str, Dict[str, torch.Tensor]] = None, precision: str = "fp32",
                 feature_grad_mult: Union[None, float] = None, group=None):
        """feature_grad_mult: fairseq's ``Wav2Vec2Model.feature_grad_mult`` - ``forward()``'s gradient w.r.t. ``estimate``
        passes through ``GradMultiply(features, feature_grad_mult)`` at the conv feature extractor's output, exactly as
        in the reference's backbone (built from wav2vec_small.pt, whose config says 0.1; nomad.py:58).  None = 0.1, or
        the value in the fairseq checkpoint ``$NOMAD_W2V_CHECKPOINT`` names (read with ``weights_only=True``; nothing is
        unpickled that the caller did not name); 1.0 = plain chain rule.

        precision of the embeddings of ``predict`` / ``get_embeddings*``:
        "fp32"   the reference's arithmetic (fp32 MFMA), scores within 1e-4 of the reference;
        "bf16x3" GEMM operands split into hi + lo bf16 planes, three bf16 MFMA products per fp32 product, fp32
                 accumulation / softmax / norms: scores within ~1e-6 of the fp32 path, 2.7x as fast on full batches
                 (batches of fewer than BF16X3_MIN_SAMPLES samples stay on fp32 buffers, with the same three-product
                 arithmetic in their GEMMs: ``Engine.gemm_precision``); ``forward()``'s differentiated branch and its
                 backward run that way too;
        "bf16"   bf16 storage, fp32 accumulation: scores within ~5e-4 of fp32, fastest on long recordings.
        ``forward()`` (the training loss) is fp32 unless precision is "bf16x3"."""
        if precision not in ("fp32", "bf16x3", "bf16"):
            raise ValueError("precision must be 'fp32', 'bf16x3' or 'bf16'")
        self.precision = precision
        dev_index = _resolve_device(device)
        self.DEVICE = f"cuda:{dev_index}"
        print(f"NOMAD running on: {self.DEVICE}")
        if weights is None:
            path = find_checkpoint()
            if path is None:
                raise FileNotFoundError(
                    "NOMAD weights not found: place nomad_best_model.pt under ./pt-models/ (the reference "
                    "downloads it there, nomad.py:27-33) or set $NOMAD_CHECKPOINT; for synthetic benchmarks "
                    "pass weights='seeded'")
            sd = load_checkpoint(path)
        elif isinstance(weights, str) and weights == "seeded":
            sd = seeded_state_dict(0)
        elif isinstance(weights, str):
            sd = load_checkpoint(weights)
        else:
            sd = weights
        self.group = group   # torch.distributed group predict() shards its files over (None: the default group, if any)
        self.engine = Engine(sd, dev_index)
        self.engine.feature_grad_mult = find_feature_grad_mult() if feature_grad_mult is None else float(feature_grad_mult)
        if precision == "bf16x3":
            # everything that stays on fp32 buffers in this mode - batches too small for the split-storage path, and the
            # differentiated branch of forward() with its backward - forms its GEMM products as three bf16 MFMA products
            # as well (hi / lo halves split in registers): ~1e-6 on scores, ~5x shorter K loops on the small-M problems
            self.engine.gemm_precision = "bf16x3"
        self.model = TripletModel(self.engine)
        self.lossnet_layers = LossNetLayers(self.engine, SSL_OUT_DIM, EMB_DIM, precision)
        self.nomad_loss = NomadLoss(self.engine)

    # ------------------------------------------------------------------------------------------
    def predict(self, mode="dir", nmr="data/nmr-data", deg="data/test-data", results_path=None):
        if nmr is None:
            raise Exception("nmr_path not specified, you need to pass a valid value to nmr_path")
        if deg is None:
            raise Exception("test_path not specified, you need to pass a valide value to test_path")

        if mode == "dir":
            if os.path.isdir(nmr) == False:  # noqa: E712 (message parity with the reference)
                raise Exception(f"Path to the non-matching reference files {nmr} does not exist")
            if os.path.isdir(deg) == False:  # noqa: E712
                raise Exception(f"Path to the test files {deg} does not exist")
        elif mode == "csv":
            if os.path.isfile(nmr) == False:  # noqa: E712
                raise Exception(f"File {nmr} does not exist")
            if os.path.isfile(deg) == False:  # noqa: E712
                raise Exception(f"File {deg} does not exist")
        else:
            raise Exception(f"Mode value {mode} is not valid. Valid values are dir and csv")

        print(f"Compute non-matching reference embeddings from {nmr}")
        nmr_embeddings = self.get_embeddings(nmr).set_index("filename")

        print(f"Compute degraded embeddings from {deg}")
        test_embeddings = self.get_embeddings(deg).set_index("filename")

        # Pairwise distance matrix + average NOMAD score, on the GPU (cdist + np.mean, nomad.py:108-111)
        deg_t = torch.from_numpy(np.ascontiguousarray(test_embeddings.to_numpy(dtype=np.float32))).to(self.engine.device)
        ref_t = torch.from_numpy(np.ascontiguousarray(nmr_embeddings.to_numpy(dtype=np.float32))).to(self.engine.device)
        world, rank, collective = _dist_info(getattr(self, "group", None))
        if deg_t.shape[0] == 0 or ref_t.shape[0] == 0:
            # an empty directory: what cdist + np.mean(axis=1) give the reference - an empty matrix, and NaN means when
            # there is no reference to average over (the engine's kernels take no empty operands)
            collective = False
            dist = torch.zeros(deg_t.shape[0], ref_t.shape[0], dtype=torch.float64)
            mean = torch.full((deg_t.shape[0],), float("nan"), dtype=torch.float64)
        elif collective:   # this rank's slab of the matrix (its slice of the degraded files x all references), then one gather
            lo, hi = partition(deg_t.shape[0], world, rank)
            if hi > lo:
                dist, mean = self.engine.pairwise(deg_t[lo:hi].contiguous(), ref_t, want_matrix=True)
            else:
                dist = torch.zeros(0, ref_t.shape[0], dtype=torch.float64, device=deg_t.device)
                mean = torch.zeros(0, dtype=torch.float64, device=deg_t.device)
            dist, mean = self._all_gather_rows(dist), self._all_gather_rows(mean)
        else:
            dist, mean = self.engine.pairwise(deg_t, ref_t, want_matrix=True)
        distance_matrix = dist.cpu().numpy()
        avg_nomad = mean.cpu().numpy()

        test_files = [x.split("/")[-1].split(".")[0] for x in test_embeddings.index]
        df_avg_nomad = pd.DataFrame({"Test File": test_files, "NOMAD": avg_nomad}).set_index("Test File").round(3)

        df_dm = pd.DataFrame(distance_matrix).round(3)
        df_dm["Test File"] = test_files
        df_dm.set_index("Test File", inplace=True)
        df_dm.columns = [x.split("/")[-1].split(".")[0] for x in nmr_embeddings.index]

        if results_path is None:
            dt_string = datetime.now().strftime("%d-%m-%Y_%H-%M-%S")
            out_dir = os.path.join("results-csv", dt_string)
            if rank == 0:
                os.makedirs(out_dir, exist_ok=True)
            results_avg_path = os.path.join(out_dir, f"{dt_string}_nomad_avg.csv")
            results_scores_path = os.path.join(out_dir, f"{dt_string}_nomad_scores.csv")
        else:
            results_avg_path = os.path.join(results_path, "nomad_avg.csv")
            results_scores_path = os.path.join(results_path, "nomad_scores.csv")

        if rank == 0:   # every rank returns the tables, one writes the files
            df_avg_nomad.reset_index().to_csv(results_avg_path, index=False)
            _write_rounded_csv(df_dm.reset_index(), results_scores_path)
        return df_avg_nomad, df_dm

    def forward(self, estimate, clean):
        """NOMAD loss (nomad.py:142-146), differentiable w.r.t. ``estimate``.

        The whole forward and backward run in the HIP engine (``torch.autograd.Function`` glue only).  As in the
        reference, the gradient that reaches ``estimate`` carries fairseq's ``feature_grad_mult`` (0.1 for wav2vec 2.0
        BASE; ``Nomad(feature_grad_mult=...)`` / ``self.engine.feature_grad_mult``).  The backbone is frozen: the reference would also accumulate parameter gradients nobody reads
        (the freeze is commented out at nomad.py:74-76); ``clean`` receives no gradient."""
        return _NomadLossFn.apply(estimate, clean, self)

    def graphed_loss(self, estimate: torch.Tensor, clean: torch.Tensor) -> "GraphedLoss":
        """``forward`` + backward for inputs of this shape, captured as one HIP graph (see ``GraphedLoss``): for training loops with
        a fixed batch shape - the per-step launch overhead of ~440 small kernels disappears, the bits do not change."""
        return GraphedLoss(self, estimate, clean)

    def _all_gather_rows(self, x: torch.Tensor) -> torch.Tensor:
        """Rows of every rank, in rank order, on every rank (nccl = RCCL needs device tensors, gloo takes host ones)."""
        import torch.distributed as dist
        from .dist import all_gather_rows
        if dist.get_backend(getattr(self, "group", None)) == "nccl":
            x = x.to(self.engine.device)
        return all_gather_rows(x.contiguous(), getattr(self, "group", None), force_collective=True)

    def get_embeddings(self, path):
        if os.path.isdir(path):
            data = pd.DataFrame(os.listdir(path))
            data.columns = ["filename"]
            data["filename"] = [os.path.join(path, x) for x in data["filename"]]
        elif os.path.isfile(path):
            data = pd.read_csv(path)
            if "filename" not in data.columns:
                raise Exception("File {path} not including a column called filename. Please pass a csv file with a "
                                "column called filename that includes the absolute filpaths of the waveforms.")
        else:
            raise Exception(f"Path {path} does not exist")
        return self.get_embeddings_csv(self.model, data)

    def _embed_files_into(self, paths, embeddings: np.ndarray, max_batch_samples: int) -> None:
        """The file pipeline over ``paths`` -> rows of ``embeddings`` (same order)."""
        pending = None                                   # (row indices, host copy in flight) of the previous batch
        for idxs, packed, uploaded in _staged_batches(paths, lambda p: self.load_processing(p, trim=False),
                                                      self.engine.pack_ragged_host, max_batch_samples, self.DECODE_THREADS,
                                                      self.PIPELINE_BATCHES, self.NATIVE_WAV_THREADS,
                                                      device=getattr(self.engine, "device", None)):
            prec = self.precision
            if prec == "bf16x3" and sum(packed[1]) < BF16X3_MIN_SAMPLES:
                # a handful of files does not fill the 256 x 256 tiles of the split-storage path: they take the fp32-buffer forward,
                # which is the faster one there.  Its GEMM products follow Engine.gemm_precision, and Nomad(precision="bf16x3") has
                # set that to "bf16x3" - so these embeddings are bf16x3-class (~1e-6 from fp32) too, not exact fp32
                prec = "fp32"
            emb = self.engine.embed_ragged(None, precision=prec, packed=packed)   # asynchronous
            uploaded()                                                             # the staging slot is free once the copy is done
            fetch = self.engine.fetch_async(emb)                                   # D2H enqueued right behind it
            if pending is not None:
                embeddings[pending[0]] = pending[1].result()                        # waits for the PREVIOUS batch only
            pending = (idxs, fetch)
        if pending is not None:
            embeddings[pending[0]] = pending[1].result()

    def get_embeddings_csv(self, model, file_names, root=False, max_batch_samples: int = 256 * 64000):
        """Embeddings for every row of ``file_names`` (a DataFrame with the path in column 0).

        The reference embeds one file per iteration with a device sync each time (nomad.py:171-183).  Here files
        of arbitrary lengths are packed into ragged batches (``nomad_embed_ragged``: no padding enters the
        arithmetic, results are bit-identical to per-file calls) of at most ``max_batch_samples`` samples, as a
        pipeline (``_staged_batches``): a packer thread has the C ABI's reader decode (and resample) each batch on native
        host threads straight into a pinned staging buffer, and this thread only enqueues GPU work - batch k+1 is
        built and launched while the GPU still runs batch k, and results are fetched one batch late.
        Like the reference, which holds one clip at a time, memory does not grow with the size of the directory:
        never more than ``PIPELINE_BATCHES`` staged batches (plus the decode look-ahead) are alive.
        Inside a ``torch.distributed`` job (one process per GPU) every rank embeds its contiguous slice of the list
        and one all-gather gives every rank the whole table."""
        file_names_arr = np.array(file_names)
        paths = []
        for row in file_names_arr:
            name = row[0] if isinstance(row, np.ndarray) else row
            paths.append(os.path.join(root, name) if root else name)
        # one process per GPU under torch.distributed: every rank embeds a contiguous slice of the file list, one
        # all-gather (RCCL over xGMI; gloo on CPU) hands every rank all embeddings in listing order
        world, rank, collective = _dist_info(getattr(self, "group", None))
        lo, hi = partition(len(paths), world, rank) if collective else (0, len(paths))
        mine = paths[lo:hi]
        embeddings = np.zeros((len(mine), EMB_DIM), dtype=np.float32)
        failure = None
        try:
            self._embed_files_into(mine, embeddings, max_batch_samples)
        except Exception as e:  # noqa: BLE001 - inside a job the other ranks must hear about it before anybody raises
            if not collective:
                raise
            failure = e
        if collective:
            # a rank that failed (an unreadable file in its slice) must not leave the others waiting in the all-gather
            flags = self._all_gather_rows(torch.tensor([0 if failure is None else 1], dtype=torch.int32)).cpu().tolist()
            if any(flags):
                if failure is not None:
                    raise failure
                raise RuntimeError(f"get_embeddings_csv: rank(s) {[r for r, f in enumerate(flags) if f]} failed on their files")
        if collective:
            embeddings = self._all_gather_rows(torch.from_numpy(embeddings)).cpu().numpy()
        emb_df = pd.DataFrame(embeddings)
        df_emb = pd.concat([file_names.reset_index(), emb_df], axis=1).drop("index", axis=1)
        return df_emb

    # files are decoded (and resampled) by the C ABI's reader on plain host threads (0: everything through load_processing);
    # files it does not take (unusual headers / encodings) are decoded on a few Python threads - more than two of those
    # only fight over the interpreter lock
    NATIVE_WAV_THREADS = int(os.environ.get("NOMAD_WAV_THREADS", min(8, os.cpu_count() or 1)))
    DECODE_THREADS = int(os.environ.get("NOMAD_DECODE_THREADS", min(2, os.cpu_count() or 1)))
    PIPELINE_BATCHES = 2      # staged batches alive at any time: one on the GPU, one being built / waiting

    def load_processing(self, filepath, target_sr=16000, trim=False):
        """file -> (1, N) fp32 mono tensor at 16 kHz, like the reference (nomad.py:192-212) but without torchaudio."""
        return torch.from_numpy(wavio.load_processing(filepath, target_sr, trim))
