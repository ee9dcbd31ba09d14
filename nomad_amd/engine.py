"""Host-side handle on the HIP engine: owns a nomad_ctx, hands it device buffers that PyTorch-ROCm
allocates, and launches on torch's current HIP stream.  PyTorch is plumbing here (memory, streams,
torch.distributed); every FLOP of the NOMAD path runs in libnomad_hip.so."""
from __future__ import annotations

import os

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from .weights import check_state_dict, num_frames

P = "ssl_model."


def _weights_struct(sd: Dict[str, torch.Tensor]):
    """Build the nomad_weights struct of HOST pointers; returns (struct, keepalive list)."""
    keep = []

    def ptr(key):
        t = sd[key].detach().to(device="cpu", dtype=torch.float32).contiguous()
        keep.append(t)
        return t.data_ptr()

    w = _lib.Weights()
    for i in range(7):
        w.conv_w[i] = ptr(P + f"feature_extractor.conv_layers.{i}.0.weight")
    w.gn_w = ptr(P + "feature_extractor.conv_layers.0.2.weight")
    w.gn_b = ptr(P + "feature_extractor.conv_layers.0.2.bias")
    w.feat_ln_w = ptr(P + "layer_norm.weight")
    w.feat_ln_b = ptr(P + "layer_norm.bias")
    w.proj_w = ptr(P + "post_extract_proj.weight")
    w.proj_b = ptr(P + "post_extract_proj.bias")
    w.pos_v = ptr(P + "encoder.pos_conv.0.weight_v")
    w.pos_g = ptr(P + "encoder.pos_conv.0.weight_g")
    w.pos_b = ptr(P + "encoder.pos_conv.0.bias")
    w.enc_ln_w = ptr(P + "encoder.layer_norm.weight")
    w.enc_ln_b = ptr(P + "encoder.layer_norm.bias")
    for l in range(_lib.NUM_LAYERS):
        q = P + f"encoder.layers.{l}."
        lw = w.layers[l]
        for short, name in (("q", "self_attn.q_proj"), ("k", "self_attn.k_proj"), ("v", "self_attn.v_proj"),
                            ("o", "self_attn.out_proj"), ("ln1", "self_attn_layer_norm"), ("fc1", "fc1"),
                            ("fc2", "fc2"), ("ln2", "final_layer_norm")):
            setattr(lw, short + "_w", ptr(q + name + ".weight"))
            setattr(lw, short + "_b", ptr(q + name + ".bias"))
    w.emb_w = ptr("embedding_layer.1.weight")
    w.emb_b = ptr("embedding_layer.1.bias")
    return w, keep


class _AsyncFetch:
    """Device tensor -> pinned host copy enqueued behind the work that produces it; ``result()`` waits for THAT copy
    only (an event), not for whatever has been enqueued on the stream since."""

    def __init__(self, dev: torch.Tensor):
        self.host = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True)
        self.host.copy_(dev, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(dev.device))

    def result(self):
        self.event.synchronize()
        return self.host.numpy()


class Engine:
    """One engine per (process, GPU).  Not thread-safe; asynchronous on torch's current stream."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device: int = 0, diag=None):
        """diag=True: run on libnomad_diag.so (same path + the experimental kernel instantiations the measurement tools and
        the kernel tests select by tile id); None: ``NOMAD_DIAG_LIB=1`` decides; the product never passes it."""
        self.lib = _lib.load(diag)
        check_state_dict(state_dict)
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        w, keep = _weights_struct(state_dict)
        handle = C.c_void_p()
        _lib.check(self.lib.nomad_create(C.byref(handle), self.device_index, C.byref(w)), "nomad_create")
        del keep
        self.ctx = handle
        self._state_dict = state_dict     # host copy: nomad_train_enable re-reads it for the master parameters
        self._train_segments = None
        self._ws: Optional[torch.Tensor] = None
        self._ws_side: Dict[int, torch.Tensor] = {}     # further workspaces for concurrent forwards on side streams
        self._side_streams: Dict[int, "torch.cuda.Stream"] = {}
        self._l1_scratch: Optional[torch.Tensor] = None
        # A library built WITH packed-FP32 instructions (NOMAD_PACKED_FP32=1 / the _pk A/B variants) must not run two of its
        # forwards concurrently: v_pk_fma_f32 can lose a product next to the other forward's bf16 128 x 128 GEMM (DESIGN.md
        # "The packed-FP32 hazard").  The shipped build reports 0 here; with the bit set every two-stream batch split is off.
        self.build_flags = int(self.lib.nomad_build_flags())
        self.stochastic = self.stochastic_branches = False   # train_set_stochastic / train_set_branches: dropout or LayerDrop is on
        if self.build_flags & 1:
            import warnings
            warnings.warn("nomad_amd: this library was built with packed-FP32 instructions; the two-stream batch splits are switched "
                          "off (results of concurrent forwards would not be reproducible)", RuntimeWarning, stacklevel=2)
            self.F32_SPLIT_ROWS = self.BF16_SPLIT_ROWS = self.X3_SPLIT_ROWS = 0

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.nomad_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers -----------------------------------------------------------------------------
    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def workspace_bytes(self, B: int, n_samples: int) -> int:
        n = C.c_size_t()
        _lib.check(self.lib.nomad_workspace_bytes(self.ctx, B, n_samples, C.byref(n)), "nomad_workspace_bytes")
        return n.value

    def _workspace(self, nbytes: int, side=False) -> torch.Tensor:
        """side: False / 0 = the main workspace, True / k >= 1 = the workspace of side stream k."""
        k = int(side)
        if k:
            ws = self._ws_side.get(k)
            if ws is None or ws.numel() < nbytes:
                self._ws_side.pop(k, None)
                ws = self._ws_side[k] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        else:
            if self._ws is None or self._ws.numel() < nbytes:
                self._ws = None
                self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            ws = self._ws
        if ws.is_cuda:
            # the caller's stream may differ from the one the block was allocated on: when the block is replaced by a larger one
            # the caching allocator must not hand it out again before this stream's kernels are done with it
            ws.record_stream(torch.cuda.current_stream(self.device))
        return ws

    def side_stream(self, k: int = 1) -> "torch.cuda.Stream":
        if k not in self._side_streams:
            self._side_streams[k] = torch.cuda.Stream(device=self.device)
        return self._side_streams[k]

    def _check_dev(self, t: torch.Tensor, name: str):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError(f"{name} must be a contiguous fp32 tensor on {self.device}")

    # ---- hot path ------------------------------------------------------------------------------
    def embed(self, wav: torch.Tensor, head: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
              want_layers: bool = False, side: bool = False):
        """wav (B,N) or (B,1,N) fp32 on the GPU -> emb (B,256) [, layers (12,B,T,768)].
        side=True uses a second workspace so the call may run concurrently with another forward on a
        different stream (the launch stream is always torch's current stream)."""
        if wav.dim() == 3:
            wav = wav.squeeze(1)
        self._check_dev(wav, "wav")
        B, N = wav.shape
        T = num_frames(N)
        if T < 1:
            raise ValueError(f"clip of {N} samples is shorter than the conv stack's receptive field")
        emb = torch.empty(B, 256, dtype=torch.float32, device=self.device)
        layers = torch.empty(12, B, T, 768, dtype=torch.float32, device=self.device) if want_layers else None
        hw = hb = None
        if head is not None:
            hw, hb = head
            self._check_dev(hw, "head weight")
            self._check_dev(hb, "head bias")

        def run(w, e, use_side):
            b = w.shape[0]
            ws = self._workspace(self.workspace_bytes(b, N), use_side)
            _lib.check(self.lib.nomad_embed(self.ctx, w.data_ptr(), b, N,
                                            hw.data_ptr() if hw is not None else None,
                                            hb.data_ptr() if hb is not None else None,
                                            e.data_ptr(), layers.data_ptr() if layers is not None else None,
                                            ws.data_ptr(), ws.numel(), self._stream()), "nomad_embed")

        ways = min(self.F32_SPLIT_WAYS, B)
        if side or want_layers or ways < 2 or not self.F32_SPLIT_ROWS or not (self.F32_SPLIT_ROWS <= B * T < self.F32_SPLIT_MAX_ROWS):
            if not side:
                self.lib.nomad_set_concurrent_parts(self.ctx, 1)   # tile-shape hint only: results never depend on it (include/nomad_hip.h)
            run(wav, emb, side)
        else:
            self.lib.nomad_set_concurrent_parts(self.ctx, ways)
            # parts of the batch on separate streams: each part's kernels fill the CUs the others' partial last rounds of tiles
            # leave idle (every instantiation contracts k in the same order, so the parts' bits equal the whole batch's)
            cur = torch.cuda.current_stream(self.device)
            cuts = [B * i // ways for i in range(ways + 1)]
            if ways == 2 and self.F32_SPLIT_FRAC != 0.5:   # (experiment: unequal halves, so that the two parts' launches do not end together)
                cuts[1] = max(1, min(B - 1, int(round(B * self.F32_SPLIT_FRAC))))
            for k in range(1, ways):
                st = self.side_stream(k)
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    run(wav[cuts[k]:cuts[k + 1]], emb[cuts[k]:cuts[k + 1]], k)
            run(wav[:cuts[1]], emb[:cuts[1]], False)
            for k in range(1, ways):
                cur.wait_stream(self.side_stream(k))
        return (emb, layers) if want_layers else emb

    # Plain scoring batches of at least F32_SPLIT_ROWS frames run as two halves on two streams: each half's kernels fill the
    # CUs the other's partial last round of tiles leaves idle: +7.6 % at 32 clips of 4 s, +4 % at 64, +5 % at 128
    # (profiles/r01_f32_two_streams.txt), +0.7 % at 256 (the bench workload, 50 944 frames: 2253-2260 vs 2238-2245 clips/s
    # alternating in one run).  bench.py takes its per-kernel timings (roofline) from a second pass with the split off
    # (F32_SPLIT_ROWS = 0), where kernels run alone.  NOMAD_F32_SPLIT_ROWS / _MAX_ROWS override.
    F32_SPLIT_ROWS = int(os.environ.get("NOMAD_F32_SPLIT_ROWS", 4000))
    F32_SPLIT_MAX_ROWS = int(os.environ.get("NOMAD_F32_SPLIT_MAX_ROWS", 1 << 30))
    F32_SPLIT_WAYS = int(os.environ.get("NOMAD_F32_SPLIT_WAYS", 2))
    F32_SPLIT_FRAC = float(os.environ.get("NOMAD_F32_SPLIT_FRAC", 0.5))

    def fetch_async(self, dev: torch.Tensor) -> _AsyncFetch:
        """Start copying a result to the host; ``.result()`` (numpy) later waits for this copy alone."""
        return _AsyncFetch(dev)

    def pack_ragged_host(self, waves):
        """Host-side half of ``embed_ragged`` for host inputs: clips -> ONE pinned (B, stride) fp32 staging tensor
        (rows are only read up to their length: no zero fill) + lengths.  Touches no GPU state, so a worker thread
        can build the next batch's staging buffer while the GPU runs the current one (``Nomad.get_embeddings_csv``)."""
        flat = [torch.as_tensor(w, dtype=torch.float32).reshape(-1) for w in waves]
        lens = [int(w.numel()) for w in flat]
        stride = (max(lens) + 3) // 4 * 4
        host = torch.empty(len(flat), stride, dtype=torch.float32, pin_memory=True)
        for i, w in enumerate(flat):
            host[i, :lens[i]] = w
        return host, lens

    def embed_ragged(self, waves, head: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, bf16: bool = False,
                     precision: Optional[str] = None, packed: Optional[Tuple[torch.Tensor, list]] = None) -> torch.Tensor:
        """Embed clips of different lengths in ONE launch sequence (no padding in the arithmetic).

        waves: list of 1-D (or (1,N)) fp32 tensors / numpy arrays (host or device).  Returns (B,256) fp32 on the
        GPU, bit-identical to embedding every clip on its own.  precision: "fp32" (default), "bf16x3" (fp32-class
        scores from split bf16 operands) or "bf16" (also bf16=True); the last two take no head override.
        packed: instead of ``waves``, the (staging tensor, lengths) pair ``pack_ragged_host`` made."""
        precision = precision or ("bf16" if bf16 else "fp32")
        if precision not in ("fp32", "bf16", "bf16x3"):
            raise ValueError("precision must be 'fp32', 'bf16x3' or 'bf16'")
        if precision != "fp32" and head is not None:
            raise ValueError(f"the {precision} path has no head override")
        if packed is not None:
            host, lens = packed
            B, stride = host.shape
            buf = host.to(self.device, non_blocking=True)
        else:
            flat = [torch.as_tensor(w, dtype=torch.float32).reshape(-1) for w in waves]
            lens = [int(w.numel()) for w in flat]
            B, stride = len(flat), max(lens)
            stride = (stride + 3) // 4 * 4
            # rows are only read up to lens[i], so the buffer needs no zero fill; device inputs are packed on the device,
            # host inputs in one pinned staging buffer and ONE asynchronous copy
            if all(w.is_cuda for w in flat):
                buf = torch.empty(B, stride, dtype=torch.float32, device=self.device)
                for i, w in enumerate(flat):
                    buf[i, :lens[i]] = w
            else:
                host, lens = self.pack_ragged_host(flat)
                buf = host.to(self.device, non_blocking=True)
        emb = torch.empty(B, 256, dtype=torch.float32, device=self.device)
        hw, hb = head if head is not None else (None, None)
        if precision != "fp32":
            enable = self.lib.nomad_enable_bf16 if precision == "bf16" else self.lib.nomad_enable_bf16x3
            _lib.check(enable(self.ctx), f"nomad_enable_{precision}")

        def run(lo, hi, side):
            n = hi - lo
            arr = (C.c_int * n)(*lens[lo:hi])
            nb = C.c_size_t()
            src, dst = buf[lo:hi], emb[lo:hi]
            if precision == "fp32":
                _lib.check(self.lib.nomad_workspace_bytes_ragged(self.ctx, n, arr, C.byref(nb)), "nomad_workspace_bytes_ragged")
                ws = self._workspace(nb.value, side)
                _lib.check(self.lib.nomad_embed_ragged(self.ctx, src.data_ptr(), n, stride, arr,
                                                       hw.data_ptr() if hw is not None else None,
                                                       hb.data_ptr() if hb is not None else None,
                                                       dst.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()),
                           "nomad_embed_ragged")
                return
            size, fwd = ((self.lib.nomad_workspace_bytes_ragged_bf16, self.lib.nomad_embed_ragged_bf16) if precision == "bf16" else
                         (self.lib.nomad_workspace_bytes_ragged_bf16x3, self.lib.nomad_embed_ragged_bf16x3))
            _lib.check(size(self.ctx, n, arr, C.byref(nb)), f"nomad_workspace_bytes_ragged_{precision}")
            ws = self._workspace(nb.value, side)
            _lib.check(fwd(self.ctx, src.data_ptr(), n, stride, arr, dst.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()),
                       f"nomad_embed_ragged_{precision}")

        # two halves (by audio length) on two streams where that pays, as in embed / embed_bf16x3; a clip's result does not
        # depend on the batch it is in, so the split changes no bit
        rows = sum(num_frames(n) for n in lens)
        split = B >= 2 and ((precision == "bf16x3" and self.X3_SPLIT_ROWS and rows >= self.X3_SPLIT_ROWS) or
                            (precision == "bf16" and self.BF16_SPLIT_ROWS and rows >= self.BF16_SPLIT_ROWS) or
                            (precision == "fp32" and self.F32_SPLIT_ROWS and self.F32_SPLIT_ROWS <= rows < self.F32_SPLIT_MAX_ROWS))
        if not split:
            run(0, B, False)
            return emb
        acc, h = 0, 1
        for i, n in enumerate(lens[:-1]):
            acc += n
            h = i + 1
            if 2 * acc >= sum(lens):
                break
        cur, ss = torch.cuda.current_stream(self.device), self.side_stream()
        ss.wait_stream(cur)
        with torch.cuda.stream(ss):
            run(h, B, True)
        run(0, h, False)
        cur.wait_stream(ss)
        return emb

    def pairwise(self, deg: torch.Tensor, ref: torch.Tensor, want_matrix: bool = True):
        """deg (Nd,256), ref (Nr,256) fp32 on GPU -> (dist (Nd,Nr) float64 or None, mean (Nd,) float64)."""
        self._check_dev(deg, "deg")
        self._check_dev(ref, "ref")
        if deg.shape[1] != 256 or ref.shape[1] != 256:
            raise ValueError("embeddings must be 256-dimensional")
        Nd, Nr = deg.shape[0], ref.shape[0]
        dist = torch.empty(Nd, Nr, dtype=torch.float64, device=self.device) if want_matrix else None
        mean = torch.empty(Nd, dtype=torch.float64, device=self.device)
        _lib.check(self.lib.nomad_pairwise(self.ctx, deg.data_ptr(), Nd, ref.data_ptr(), Nr,
                                           dist.data_ptr() if dist is not None else None, mean.data_ptr(),
                                           self._stream()), "nomad_pairwise")
        return dist, mean

    def l1_loss(self, a_layers, b_layers, a_emb, b_emb) -> torch.Tensor:
        for t, n in ((a_layers, "a_layers"), (b_layers, "b_layers"), (a_emb, "a_emb"), (b_emb, "b_emb")):
            self._check_dev(t, n)
        _, B, T, _ = a_layers.shape
        if self._l1_scratch is None:
            self._l1_scratch = torch.empty(self.lib.nomad_l1_scratch_bytes(), dtype=torch.uint8, device=self.device)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.nomad_l1_loss(self.ctx, a_layers.data_ptr(), b_layers.data_ptr(), a_emb.data_ptr(),
                                          b_emb.data_ptr(), B, T, loss.data_ptr(), self._l1_scratch.data_ptr(),
                                          self._stream()), "nomad_l1_loss")
        return loss[0]

    # ---- bf16 path (long-form clips, config C5) -----------------------------------------------------
    # Like the bf16x3 path below, bf16 batches with at least this many frames are embedded as two halves on two streams:
    # every GEMM of the path runs one 256 x 256 workgroup per CU, so the partial last round of one half's tiles (the
    # N = 768 GEMMs of 32 clips x 30 s are 2.2 rounds) and its per-tile prologue / epilogue are filled by the other
    # half's kernels.  A clip's bits do not depend on the batch it is in, so the split changes no result.
    # (Round 2 switched this off because embed_bf16 then differed run to run in ~1 % of the calls.  Round 3 found the cause -
    # not the split: v_pk_fma_f32 in conv0 lost products while the other half's 128 x 128 bf16 GEMM shared its SIMD, DESIGN.md
    # "The packed-FP32 hazard" - and the library is now built without packed-FP32 instructions; tests/test_gpu_race_screen.py
    # holds every precision to bit-identical results with the split on.)  NOMAD_BF16_SPLIT_ROWS overrides; 0 disables.
    BF16_SPLIT_ROWS = int(os.environ.get("NOMAD_BF16_SPLIT_ROWS", 4000))

    def _embed_bf16_into(self, wav: torch.Tensor, emb: torch.Tensor, side: bool):
        B, N = wav.shape
        ws = self._workspace(self._size(self.lib.nomad_workspace_bytes_bf16, B, N, "nomad_workspace_bytes_bf16"), side=side)
        _lib.check(self.lib.nomad_embed_bf16(self.ctx, wav.data_ptr(), B, N, emb.data_ptr(), ws.data_ptr(), ws.numel(),
                                             self._stream()), "nomad_embed_bf16")

    def embed_bf16(self, wav: torch.Tensor) -> torch.Tensor:
        """Scoring forward with bf16 activations/weights (fp32 accumulation and statistics)."""
        if wav.dim() == 3:
            wav = wav.squeeze(1)
        self._check_dev(wav, "wav")
        if not wav.is_contiguous():
            wav = wav.contiguous()
        B, N = wav.shape
        if num_frames(N) < 1:
            raise ValueError(f"clip of {N} samples is shorter than the conv stack's receptive field")
        _lib.check(self.lib.nomad_enable_bf16(self.ctx), "nomad_enable_bf16")
        emb = torch.empty(B, 256, dtype=torch.float32, device=self.device)
        rows = B * int(self.lib.nomad_num_frames(N))
        ways = min(self.BF16_SPLIT_WAYS, B)
        if ways < 2 or not self.BF16_SPLIT_ROWS or rows < self.BF16_SPLIT_ROWS:
            self.lib.nomad_set_concurrent_parts(self.ctx, 1)   # scheduling hint only: results never depend on it (include/nomad_hip.h)
            self._embed_bf16_into(wav, emb, side=False)
            return emb
        self.lib.nomad_set_concurrent_parts(self.ctx, ways)
        cur = torch.cuda.current_stream(self.device)
        cuts = [B * i // ways for i in range(ways + 1)]
        for k in range(1, ways):
            st = self.side_stream(k)
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                self._embed_bf16_into(wav[cuts[k]:cuts[k + 1]], emb[cuts[k]:cuts[k + 1]], side=k)
        self._embed_bf16_into(wav[:cuts[1]], emb[:cuts[1]], side=False)
        for k in range(1, ways):
            cur.wait_stream(self.side_stream(k))
        return emb

    BF16_SPLIT_WAYS = int(os.environ.get("NOMAD_BF16_SPLIT_WAYS", 2))

    # ---- bf16x3 path: fp32-class scores on the bf16 matrix cores ---------------------------------------
    # Batches with at least this many frames (rows of the encoder GEMMs) are embedded as two halves on two streams: the
    # path's kernels run one 256 x 256 workgroup per CU, and e.g. the N = 768 GEMMs of a 256-clip batch are 2.33 rounds of
    # the 256 CUs - the other half's kernels fill the idle third round (+4 % at 256 clips of 4 s, +14 % at 128, break-even at 16; bit-identical results:
    # profiles/r01_bf16x3_two_streams.txt).  NOMAD_X3_SPLIT_ROWS overrides; 0 disables.
    X3_SPLIT_ROWS = int(os.environ.get("NOMAD_X3_SPLIT_ROWS", 4000))

    def _embed_bf16x3_into(self, wav: torch.Tensor, emb: torch.Tensor, side: bool):
        B, N = wav.shape
        ws = self._workspace(self._size(self.lib.nomad_workspace_bytes_bf16x3, B, N, "nomad_workspace_bytes_bf16x3"), side=side)
        _lib.check(self.lib.nomad_embed_bf16x3(self.ctx, wav.data_ptr(), B, N, emb.data_ptr(), ws.data_ptr(), ws.numel(),
                                               self._stream()), "nomad_embed_bf16x3")

    def embed_bf16x3(self, wav: torch.Tensor, head: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                     want_layers: bool = False, side: bool = False):
        """Forward whose GEMMs run as three bf16 MFMA products over hi/lo-split operands (fp32 accumulation,
        fp32 softmax / LayerNorm / head): NOMAD scores agree with the fp32 path to ~1e-6.
        want_layers / head: as in ``embed`` -> (emb, layers (12,B,T,768)), the LossNetLayers outputs (no gradient:
        the branch of ``forward()`` that needs one stays on ``embed_train``)."""
        if wav.dim() == 3:
            wav = wav.squeeze(1)
        self._check_dev(wav, "wav")
        if not wav.is_contiguous():
            wav = wav.contiguous()
        B, N = wav.shape
        T = num_frames(N)
        if T < 1:
            raise ValueError(f"clip of {N} samples is shorter than the conv stack's receptive field")
        _lib.check(self.lib.nomad_enable_bf16x3(self.ctx), "nomad_enable_bf16x3")
        emb = torch.empty(B, 256, dtype=torch.float32, device=self.device)
        if want_layers or head is not None:
            hw, hb = head if head is not None else (None, None)
            for t, name in ((hw, "head weight"), (hb, "head bias")):
                if t is not None:
                    self._check_dev(t, name)
            layers = torch.empty(12, B, T, 768, dtype=torch.float32, device=self.device)
            ws = self._workspace(self._size(self.lib.nomad_workspace_bytes_bf16x3, B, N, "nomad_workspace_bytes_bf16x3"), side=side)
            _lib.check(self.lib.nomad_embed_layers_bf16x3(self.ctx, wav.data_ptr(), B, N,
                                                          hw.data_ptr() if hw is not None else None,
                                                          hb.data_ptr() if hb is not None else None,
                                                          emb.data_ptr(), layers.data_ptr(), ws.data_ptr(), ws.numel(),
                                                          self._stream()), "nomad_embed_layers_bf16x3")
            return (emb, layers) if want_layers else emb
        if side:
            self._embed_bf16x3_into(wav, emb, side=True)
            return emb
        rows = B * int(self.lib.nomad_num_frames(N))
        if B < 2 or not self.X3_SPLIT_ROWS or rows < self.X3_SPLIT_ROWS:
            self._embed_bf16x3_into(wav, emb, side=False)
            return emb
        h = B // 2
        cur, side = torch.cuda.current_stream(self.device), self.side_stream()
        side.wait_stream(cur)                      # the waveform (and anything queued before) is ready
        with torch.cuda.stream(side):
            self._embed_bf16x3_into(wav[h:], emb[h:], side=True)
        self._embed_bf16x3_into(wav[:h], emb[:h], side=False)
        cur.wait_stream(side)
        return emb

    def diag_split_bf16(self, x: torch.Tensor) -> torch.Tensor:
        """fp32 tensor -> split buffer (bf16 tensor of shape (2, *x.shape): plane 0 = hi, plane 1 = lo)."""
        x = x.contiguous()
        out = torch.empty((2,) + tuple(x.shape), dtype=torch.bfloat16, device=self.device)
        _lib.check(self.lib.nomad_diag_split_bf16(self.ctx, x.data_ptr(), out.data_ptr(), x.numel(), x.numel(), 0,
                                                  self._stream()), "nomad_diag_split_bf16")
        return out

    def diag_unsplit_bf16(self, xs: torch.Tensor) -> torch.Tensor:
        out = torch.empty(tuple(xs.shape[1:]), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.nomad_diag_split_bf16(self.ctx, xs.data_ptr(), out.data_ptr(), out.numel(), out.numel(), 1,
                                                  self._stream()), "nomad_diag_split_bf16")
        return out

    def diag_gemm_bf16x3(self, A, W, bias=None, R=None, gelu=False, out_f32=True, out=None, variant=None):
        """A (2,M,K), W (2,N,K), R (2,M,N) split buffers (diag_split_bf16); returns fp32 (M,N) or a split (2,M,N).
        variant (int, optional): kernel selector of nomad_diag_gemm_bf16x3 - 0/1 K-concatenated kernel (split / fp32 out),
        7/8 staged-once kernel (split / fp32 out), 12/13 the same with three A buffers (K % 192 == 0), others are timing
        probes with fp32 output."""
        if variant is None:
            variant = 8 if out_f32 else 7
        out_f32 = variant % 100 not in (0, 7, 12)
        _, M, K = A.shape
        N = W.shape[1]
        if out is None:
            out = (torch.zeros(M, N, dtype=torch.float32, device=self.device) if out_f32
                   else torch.zeros(2, M, N, dtype=torch.bfloat16, device=self.device))
        _lib.check(self.lib.nomad_diag_gemm_bf16x3(self.ctx, A.data_ptr(), W.data_ptr(),
                                                   bias.data_ptr() if bias is not None else None,
                                                   R.data_ptr() if R is not None else None, out.data_ptr(),
                                                   M, N, K, int(gelu), int(variant), self._stream()), "nomad_diag_gemm_bf16x3")
        return out

    def diag_attention_bf16x3(self, qkv_split, B, T, waves: int = -1):
        """qkv_split (2, B*T, 2304) split buffer -> split (2, B*T, 768).  waves: -1 the forward's choice, 0 the tiled
        kernel, 4 / 8 the K/V-resident kernel (T <= 256)."""
        out = torch.empty(2, B * T, 768, dtype=torch.bfloat16, device=self.device)
        _lib.check(self.lib.nomad_diag_attention_bf16x3(self.ctx, qkv_split.data_ptr(), out.data_ptr(), B, T, waves, self._stream()),
                   "nomad_diag_attention_bf16x3")
        return out

    def diag_attention_bf16(self, qkv, B, T, q_has_log2e: bool = False):
        """qkv (B*T, 2304) bf16 -> (B*T, 768) bf16.  q_has_log2e: the q columns already carry log2(e) (what the bf16
        forward's QKV projection produces); otherwise the kernel scales q itself (one more bf16 rounding of q)."""
        out = torch.empty(B * T, 768, dtype=torch.bfloat16, device=self.device)
        _lib.check(self.lib.nomad_diag_attention_bf16(self.ctx, qkv.data_ptr(), out.data_ptr(), B, T, int(q_has_log2e), self._stream()),
                   "nomad_diag_attention_bf16")
        return out

    def diag_gemm_bf16(self, A, W, bias=None, R=None, gelu=False, tile=0, out=None):
        M, K = A.shape
        N = W.shape[0]
        if out is None:
            out = torch.zeros(M, N, dtype=torch.bfloat16, device=self.device)  # zeros: a kernel that writes nothing shows
        _lib.check(self.lib.nomad_diag_gemm_bf16(self.ctx, A.data_ptr(), W.data_ptr(),
                                                 bias.data_ptr() if bias is not None else None,
                                                 R.data_ptr() if R is not None else None, out.data_ptr(),
                                                 M, N, K, int(gelu), tile, self._stream()), "nomad_diag_gemm_bf16")
        return out

    # ---- training (differentiable forward) ---------------------------------------------------------
    @property
    def feature_grad_mult(self) -> float:
        """fairseq ``feature_grad_mult``: scale of the gradient entering the conv feature extractor in
        ``embed_backward`` (0.1 = wav2vec 2.0 BASE / wav2vec_small.pt; 1.0 = plain chain rule)."""
        v = C.c_float()
        _lib.check(self.lib.nomad_get_feature_grad_mult(self.ctx, C.byref(v)), "nomad_get_feature_grad_mult")
        return float(v.value)

    @feature_grad_mult.setter
    def feature_grad_mult(self, mult: float):
        _lib.check(self.lib.nomad_set_feature_grad_mult(self.ctx, float(mult)), "nomad_set_feature_grad_mult")

    @property
    def gemm_precision(self) -> str:
        """"fp32" (exact fp32 MFMA, the default) or "bf16x3" (three bf16 MFMA products over hi / lo halves split in
        registers, fp32 accumulation) for the GEMMs of ``embed`` / ``embed_ragged`` / ``embed_train`` / the backward passes;
        buffers and every other kernel stay fp32 (``nomad_set_gemm_precision``)."""
        v = C.c_int()
        _lib.check(self.lib.nomad_get_gemm_precision(self.ctx, C.byref(v)), "nomad_get_gemm_precision")
        return "bf16x3" if v.value else "fp32"

    @gemm_precision.setter
    def gemm_precision(self, mode: str):
        if mode not in ("fp32", "bf16x3"):
            raise ValueError("gemm_precision must be 'fp32' or 'bf16x3'")
        _lib.check(self.lib.nomad_set_gemm_precision(self.ctx, int(mode == "bf16x3")), "nomad_set_gemm_precision")

    def enable_backward(self):
        _lib.check(self.lib.nomad_enable_backward(self.ctx), "nomad_enable_backward")

    def _size(self, fn, B, n_samples, what):
        n = C.c_size_t()
        _lib.check(fn(self.ctx, B, n_samples, C.byref(n)), what)
        return n.value

    def embed_train(self, wav: torch.Tensor, head: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        """Training-mode forward: -> (emb (B,256), layers (12,B,T,768), saved block for embed_backward)."""
        if wav.dim() == 3:
            wav = wav.squeeze(1)
        self._check_dev(wav, "wav")
        self.enable_backward()   # also allocates the split-K scratch: the first call must run the same kernels as later ones
        B, N = wav.shape
        T = num_frames(N)
        emb = torch.empty(B, 256, dtype=torch.float32, device=self.device)
        layers = torch.empty(12, B, T, 768, dtype=torch.float32, device=self.device)
        saved = torch.empty(self._size(self.lib.nomad_saved_bytes, B, N, "nomad_saved_bytes"), dtype=torch.uint8,
                            device=self.device)
        hw, hb = head if head is not None else (None, None)
        ws = self._workspace(self.workspace_bytes(B, N))
        _lib.check(self.lib.nomad_embed_train(self.ctx, wav.data_ptr(), B, N,
                                              hw.data_ptr() if hw is not None else None,
                                              hb.data_ptr() if hb is not None else None,
                                              emb.data_ptr(), layers.data_ptr(), saved.data_ptr(), saved.numel(),
                                              ws.data_ptr(), ws.numel(), self._stream()), "nomad_embed_train")
        return emb, layers, saved

    def embed_backward(self, wav, layers, saved, dlayers, demb, head=None) -> torch.Tensor:
        """d loss / d wav (B,N) from d loss / d layers (12,B,T,768 or None) and d loss / d emb (B,256)."""
        if wav.dim() == 3:
            wav = wav.squeeze(1)
        B, N = wav.shape
        self.enable_backward()
        nb = self._size(self.lib.nomad_backward_workspace_bytes, B, N, "nomad_backward_workspace_bytes")
        ws = self._workspace(nb)
        dwav = torch.empty(B, N, dtype=torch.float32, device=self.device)
        hw, hb = head if head is not None else (None, None)
        _lib.check(self.lib.nomad_embed_backward(self.ctx, wav.data_ptr(), B, N,
                                                 hw.data_ptr() if hw is not None else None,
                                                 hb.data_ptr() if hb is not None else None,
                                                 layers.data_ptr(), saved.data_ptr(), saved.numel(),
                                                 dlayers.data_ptr() if dlayers is not None else None, demb.data_ptr(),
                                                 dwav.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()),
                   "nomad_embed_backward")
        return dwav

    def l1_loss_backward(self, a_layers, b_layers, a_emb, b_emb, upstream: torch.Tensor):
        """(d loss/d a_layers, d loss/d a_emb) for NomadLoss, scaled by the 0-dim device tensor `upstream`."""
        _, B, T, _ = a_layers.shape
        dl = torch.empty_like(a_layers)
        de = torch.empty_like(a_emb)
        up = upstream.to(self.device, torch.float32).reshape(1).contiguous()
        _lib.check(self.lib.nomad_l1_loss_backward(self.ctx, a_layers.data_ptr(), b_layers.data_ptr(), a_emb.data_ptr(),
                                                   b_emb.data_ptr(), B, T, up.data_ptr(), dl.data_ptr(), de.data_ptr(),
                                                   self._stream()), "nomad_l1_loss_backward")
        return dl, de

    # ---- triplet fine-tuning step (train_triplet.py:112-133) ---------------------------------------
    def train_enable(self):
        """Allocate master parameters / gradients / Adam moments and re-point the engine at them."""
        if self._train_segments is not None:
            return
        w, keep = _weights_struct(self._state_dict)
        _lib.check(self.lib.nomad_train_enable(self.ctx, C.byref(w)), "nomad_train_enable")
        del keep
        segs = []
        name = C.create_string_buffer(128)
        off, cnt = C.c_size_t(), C.c_size_t()
        for i in range(self.lib.nomad_train_num_segments()):
            _lib.check(self.lib.nomad_train_segment(i, name, 128, C.byref(off), C.byref(cnt)), "nomad_train_segment")
            segs.append((name.value.decode(), off.value, cnt.value))
        self._train_segments = segs

    def train_segments(self):
        """[(checkpoint key, offset, count)] of the flat parameter vector."""
        self.train_enable()
        return list(self._train_segments)

    def train_param_count(self) -> Tuple[int, int]:
        total, head = C.c_size_t(), C.c_size_t()
        _lib.check(self.lib.nomad_train_param_count(C.byref(total), C.byref(head)), "nomad_train_param_count")
        return total.value, head.value

    def train_zero_grad(self):
        _lib.check(self.lib.nomad_train_zero_grad(self.ctx, self._stream()), "nomad_train_zero_grad")

    def train_backward(self, wav: torch.Tensor, layers: torch.Tensor, saved: torch.Tensor, demb: torch.Tensor):
        """Accumulate d loss / d parameters for one embed_train call, given d loss / d emb (B,256)."""
        if wav.dim() == 3:
            wav = wav.squeeze(1)
        self._check_dev(wav, "wav")
        self._check_dev(demb, "demb")
        B, N = wav.shape
        ws = self._workspace(self._size(self.lib.nomad_train_workspace_bytes, B, N, "nomad_train_workspace_bytes"))
        _lib.check(self.lib.nomad_train_backward(self.ctx, wav.data_ptr(), B, N, layers.data_ptr(), saved.data_ptr(),
                                                 saved.numel(), demb.data_ptr(), ws.data_ptr(), ws.numel(),
                                                 self._stream()), "nomad_train_backward")

    def triplet_loss(self, a: torch.Tensor, p: torch.Tensor, n: torch.Tensor, margin: float, want_grad: bool = True):
        """nn.TripletMarginLoss(margin) -> (loss (1,), da, dp, dn) (gradients None when want_grad=False)."""
        for t, nm in ((a, "a"), (p, "p"), (n, "n")):
            self._check_dev(t, nm)
        B = a.shape[0]
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        da, dp, dn = (torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)) if want_grad else (None, None, None)
        _lib.check(self.lib.nomad_triplet_loss(self.ctx, a.data_ptr(), p.data_ptr(), n.data_ptr(), B, float(margin),
                                               loss.data_ptr(), da.data_ptr() if want_grad else None,
                                               dp.data_ptr() if want_grad else None,
                                               dn.data_ptr() if want_grad else None, self._stream()),
                   "nomad_triplet_loss")
        return loss, da, dp, dn

    def adam_step(self, lr_body: float, lr_head: float, betas=(0.9, 0.999), eps: float = 1e-8):
        _lib.check(self.lib.nomad_train_adam_step(self.ctx, float(lr_body), float(lr_head), float(betas[0]),
                                                  float(betas[1]), float(eps), self._stream()), "nomad_train_adam_step")

    def train_read(self, what: int = 0) -> torch.Tensor:
        """Flat copy of: 0 parameters, 1 gradients, 2 Adam exp_avg, 3 Adam exp_avg_sq."""
        total, _ = self.train_param_count()
        out = torch.empty(total, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.nomad_train_read(self.ctx, what, out.data_ptr(), self._stream()), "nomad_train_read")
        return out

    def train_write(self, what: int, flat: torch.Tensor):
        self._check_dev(flat, "flat")
        total, _ = self.train_param_count()
        if flat.numel() != total:
            raise ValueError(f"expected {total} floats, got {flat.numel()}")
        _lib.check(self.lib.nomad_train_write(self.ctx, what, flat.data_ptr(), self._stream()), "nomad_train_write")

    def train_set_stochastic(self, dropout: float = 0.0, attention_dropout: float = 0.0, dropout_input: float = 0.0,
                             seed: int = 0, layer_mask: int = 0xFFF):
        """model.train() regularisation for the following embed_train / train_backward calls (defaults = eval)."""
        _lib.check(self.lib.nomad_train_set_stochastic(self.ctx, float(dropout), float(attention_dropout),
                                                       float(dropout_input), int(seed) & (2 ** 64 - 1),
                                                       int(layer_mask) & 0xFFF), "nomad_train_set_stochastic")
        # (what GraphedLoss asks: a captured graph freezes the dropout masks and the step counter of its capture)
        self.stochastic = bool(dropout > 0 or attention_dropout > 0 or dropout_input > 0 or (int(layer_mask) & 0xFFF) != 0xFFF)

    def train_set_branches(self, layer_masks=None):
        """The following training batches are len(layer_masks) equal groups of clips, each with its own LayerDrop
        mask (None: back to one group using train_set_stochastic's layer_mask)."""
        self.stochastic_branches = bool(layer_masks) and any((int(m) & 0xFFF) != 0xFFF for m in layer_masks)
        if not layer_masks:
            _lib.check(self.lib.nomad_train_set_branches(self.ctx, 1, None), "nomad_train_set_branches")
            return
        arr = (C.c_uint * len(layer_masks))(*[int(m) & 0xFFF for m in layer_masks])
        _lib.check(self.lib.nomad_train_set_branches(self.ctx, len(layer_masks), arr), "nomad_train_set_branches")

    def train_set_frozen(self, freeze_encoder: bool):
        """``freeze_all: True`` of the reference's config: no parameter gradients for the encoder (nor the extractor);
        post_extract_proj, the feature LayerNorm and the head keep training."""
        _lib.check(self.lib.nomad_train_set_frozen(self.ctx, int(bool(freeze_encoder))), "nomad_train_set_frozen")

    def train_set_convnet(self, trainable: bool):
        """``freeze_convnet: False`` of the reference's config: the conv feature extractor's weights and GroupNorm get
        gradients too (scaled by ``feature_grad_mult``, like everything that enters the extractor)."""
        _lib.check(self.lib.nomad_train_set_convnet(self.ctx, int(bool(trainable))), "nomad_train_set_convnet")

    def train_set_step(self, step: int):
        _lib.check(self.lib.nomad_train_set_step(self.ctx, int(step)), "nomad_train_set_step")

    def train_unflatten(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Flat vector -> {checkpoint key: CPU tensor in the checkpoint's shape}."""
        from .weights import expected_shapes
        shapes = expected_shapes()
        host = flat.detach().cpu()
        return {k: host[o:o + n].reshape(shapes[k]).clone() for k, o, n in self.train_segments()}

    def train_flatten(self, tensors: Dict[str, torch.Tensor], fill: Optional[float] = None) -> torch.Tensor:
        """{checkpoint key: tensor} -> flat device vector.  A key that is missing raises, unless ``fill`` gives the value
        its slice gets (e.g. 0.0 for gradient dicts that leave out frozen parameters)."""
        total, _ = self.train_param_count()
        host = torch.empty(total, dtype=torch.float32)
        for k, o, n in self.train_segments():
            if k not in tensors and fill is not None:
                host[o:o + n] = fill
            else:
                host[o:o + n] = tensors[k].detach().reshape(-1).to(torch.float32)
        return host.to(self.device)

    def train_state_dict(self) -> Dict[str, torch.Tensor]:
        """Full checkpoint-layout state dict with the current (fine-tuned) parameters (torch.save-able)."""
        sd = {k: v.detach().cpu().clone() for k, v in self._state_dict.items()}
        sd.update(self.train_unflatten(self.train_read(0)))
        return sd

    # ---- measurement -----------------------------------------------------------------------------
    def profile_enable(self, on: bool = True):
        _lib.check(self.lib.nomad_profile_enable(self.ctx, int(on)), "nomad_profile_enable")

    def profile_reset(self):
        _lib.check(self.lib.nomad_profile_reset(self.ctx), "nomad_profile_reset")

    def profile_read(self):
        ms = (C.c_double * _lib.K_COUNT)()
        n = (C.c_longlong * _lib.K_COUNT)()
        fl = (C.c_double * _lib.K_COUNT)()
        _lib.check(self.lib.nomad_profile_read(self.ctx, ms, n, fl), "nomad_profile_read")
        return {name: {"ms": ms[i], "launches": n[i], "flops": fl[i]}
                for i, name in enumerate(_lib.KERNEL_CLASS_NAMES)}

    # ---- diagnostics (tests) ---------------------------------------------------------------------
    def diag_clock_probe(self, ms: float, stream: "torch.cuda.Stream") -> torch.Tensor:
        """Launch the one-wave clock probe on `stream` for ~ms milliseconds; returns the (2,) int64 device tensor
        [shader cycles, 100 MHz ticks] (read it after synchronising)."""
        with torch.cuda.stream(stream):  # the zero fill must not queue behind the load on the main stream
            out = torch.zeros(2, dtype=torch.int64, device=self.device)
        _lib.check(self.lib.nomad_diag_clock_probe(self.ctx, int(ms * 1e5), out.data_ptr(), stream.cuda_stream),
                   "nomad_diag_clock_probe")
        return out

    def diag_gemm(self, A, W, bias=None, R=None, gelu=False, tile=0):
        M, K = A.shape
        N = W.shape[0]
        out = torch.empty(M, N, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.nomad_diag_gemm(self.ctx, A.data_ptr(), W.data_ptr(),
                                            bias.data_ptr() if bias is not None else None,
                                            R.data_ptr() if R is not None else None, out.data_ptr(),
                                            M, N, K, int(gelu), tile, self._stream()), "nomad_diag_gemm")
        return out

    def diag_layernorm(self, x, gamma, beta):
        M, N = x.shape
        out = torch.empty_like(x)
        _lib.check(self.lib.nomad_diag_layernorm(self.ctx, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                 out.data_ptr(), M, N, self._stream()), "nomad_diag_layernorm")
        return out

    def diag_attention(self, qkv, B, T):
        out = torch.empty(B * T, 768, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.nomad_diag_attention(self.ctx, qkv.data_ptr(), out.data_ptr(), B, T, self._stream()),
                   "nomad_diag_attention")
        return out

    def diag_layernorm_bwd(self, x, g, gamma):
        M, N = x.shape
        dx = torch.empty_like(x)
        _lib.check(self.lib.nomad_diag_layernorm_bwd(self.ctx, x.data_ptr(), g.data_ptr(), gamma.data_ptr(),
                                                     dx.data_ptr(), M, N, self._stream()), "nomad_diag_layernorm_bwd")
        return dx

    def diag_attention_bwd(self, qkv, dctx, B, T):
        out = torch.empty(B * T, 768, dtype=torch.float32, device=self.device)
        lse = torch.empty(B * 12, T, dtype=torch.float32, device=self.device)
        dqkv = torch.empty(B * T, 2304, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.nomad_diag_attention_bwd(self.ctx, qkv.data_ptr(), dctx.data_ptr(), out.data_ptr(),
                                                     lse.data_ptr(), dqkv.data_ptr(), B, T, self._stream()),
                   "nomad_diag_attention_bwd")
        return out, lse, dqkv

    def diag_keep_intermediates(self, on: bool):
        _lib.check(self.lib.nomad_diag_keep_intermediates(self.ctx, int(on)), "nomad_diag_keep_intermediates")
        self._ws = None

    def diag_region(self, B: int, n_samples: int, name: str) -> torch.Tensor:
        """View (fp32, flat) of a named intermediate of the LAST embed() call with this (B, n_samples)."""
        off, nb = C.c_size_t(), C.c_size_t()
        _lib.check(self.lib.nomad_diag_workspace_region(self.ctx, B, n_samples, name.encode(), C.byref(off), C.byref(nb)),
                   "nomad_diag_workspace_region")
        return self._ws[off.value:off.value + nb.value].view(torch.float32)
