"""ctypes binding of libnomad_hip.so (include/nomad_hip.h).  No torch types cross this boundary:
only raw device/host addresses, sizes and the hipStream_t handle."""
from __future__ import annotations

import ctypes as C
import os

NUM_LAYERS = 12
K_GEMM, K_ATTN, K_FRONT, K_ROW, K_PAIR, K_GEMM_BIG, K_GEMM_FINE, K_COUNT = 0, 1, 2, 3, 4, 5, 6, 7
KERNEL_CLASS_NAMES = ("gemm_mfma_all", "attention_mfma", "frontend", "rowwise", "pairwise_f64",
                      "gemm_mfma_256x128", "gemm_mfma_128x64")

_fp = C.c_void_p  # every pointer is passed as an address


class LayerWeights(C.Structure):
    _fields_ = [(n, _fp) for n in ("q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "o_w", "o_b", "ln1_w", "ln1_b",
                                   "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ln2_w", "ln2_b")]


class Weights(C.Structure):
    _fields_ = ([("conv_w", _fp * 7)] +
                [(n, _fp) for n in ("gn_w", "gn_b", "feat_ln_w", "feat_ln_b", "proj_w", "proj_b",
                                    "pos_v", "pos_g", "pos_b", "enc_ln_w", "enc_ln_b")] +
                [("layers", LayerWeights * NUM_LAYERS), ("emb_w", _fp), ("emb_b", _fp)])


class WavInfo(C.Structure):
    _fields_ = [("sample_rate", C.c_int), ("channels", C.c_int), ("format_tag", C.c_int), ("bits", C.c_int),
                ("frames", C.c_longlong), ("data_offset", C.c_longlong)]


# name -> (restype, argtypes): the complete export list of include/nomad_hip.h
SIGNATURES = {
    "nomad_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(Weights)]),
    "nomad_destroy": (None, [C.c_void_p]),
    "nomad_last_error": (C.c_char_p, []),
    "nomad_version": (C.c_char_p, []),
    "nomad_abi_version": (C.c_int, []),
    "nomad_build_flags": (C.c_int, []),
    "nomad_set_concurrent_parts": (C.c_int, [C.c_void_p, C.c_int]),
    "nomad_num_frames": (C.c_int, [C.c_int]),
    "nomad_workspace_bytes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "nomad_embed": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, C.c_size_t, _fp]),
    "nomad_workspace_bytes_ragged": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "nomad_embed_ragged": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_int, C.POINTER(C.c_int), _fp, _fp, _fp, _fp,
                                     C.c_size_t, _fp]),
    "nomad_pairwise": (C.c_int, [C.c_void_p, _fp, C.c_int, _fp, C.c_int, _fp, _fp, _fp]),
    "nomad_wav_probe": (C.c_int, [C.POINTER(C.c_char_p), C.c_int, C.POINTER(WavInfo), C.POINTER(C.c_int), C.c_int]),
    "nomad_wav_frames_at": (C.c_int, [C.POINTER(WavInfo), C.c_int, C.POINTER(C.c_longlong)]),
    "nomad_wav_read_rows": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(WavInfo), C.c_int, C.POINTER(C.c_int), _fp, C.c_longlong,
                                      C.c_int, C.POINTER(C.c_int), C.c_int]),
    "nomad_l1_scratch_bytes": (C.c_size_t, []),
    "nomad_l1_loss": (C.c_int, [C.c_void_p, _fp, _fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp, _fp]),
    "nomad_enable_backward": (C.c_int, [C.c_void_p]),
    "nomad_set_feature_grad_mult": (C.c_int, [C.c_void_p, C.c_float]),
    "nomad_get_feature_grad_mult": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "nomad_set_gemm_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "nomad_get_gemm_precision": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "nomad_saved_bytes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "nomad_backward_workspace_bytes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "nomad_embed_train": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, C.c_size_t, _fp,
                                    C.c_size_t, _fp]),
    "nomad_l1_loss_backward": (C.c_int, [C.c_void_p, _fp, _fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp, _fp, _fp]),
    "nomad_embed_backward": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_int, _fp, _fp, _fp, _fp, C.c_size_t, _fp, _fp,
                                       _fp, _fp, C.c_size_t, _fp]),
    "nomad_train_param_count": (C.c_int, [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "nomad_train_num_segments": (C.c_int, []),
    "nomad_train_segment": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "nomad_train_enable": (C.c_int, [C.c_void_p, C.POINTER(Weights)]),
    "nomad_train_workspace_bytes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "nomad_train_zero_grad": (C.c_int, [C.c_void_p, _fp]),
    "nomad_train_backward": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_int, _fp, _fp, C.c_size_t, _fp, _fp, C.c_size_t, _fp]),
    "nomad_triplet_loss": (C.c_int, [C.c_void_p, _fp, _fp, _fp, C.c_int, C.c_float, _fp, _fp, _fp, _fp, _fp]),
    "nomad_train_adam_step": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _fp]),
    "nomad_train_read": (C.c_int, [C.c_void_p, C.c_int, _fp, _fp]),
    "nomad_train_write": (C.c_int, [C.c_void_p, C.c_int, _fp, _fp]),
    "nomad_train_set_step": (C.c_int, [C.c_void_p, C.c_longlong]),
    "nomad_train_set_frozen": (C.c_int, [C.c_void_p, C.c_int]),
    "nomad_train_set_convnet": (C.c_int, [C.c_void_p, C.c_int]),
    "nomad_train_set_branches": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint)]),
    "nomad_train_set_stochastic": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_ulonglong, C.c_uint]),
    "nomad_enable_bf16": (C.c_int, [C.c_void_p]),
    "nomad_workspace_bytes_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "nomad_embed_bf16": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_int, _fp, _fp, C.c_size_t, _fp]),
    "nomad_workspace_bytes_ragged_bf16": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "nomad_embed_ragged_bf16": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_int, C.POINTER(C.c_int), _fp, _fp, C.c_size_t, _fp]),
    "nomad_enable_bf16x3": (C.c_int, [C.c_void_p]),
    "nomad_workspace_bytes_bf16x3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "nomad_embed_bf16x3": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_int, _fp, _fp, C.c_size_t, _fp]),
    "nomad_embed_layers_bf16x3": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, C.c_size_t, _fp]),
    "nomad_workspace_bytes_ragged_bf16x3": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "nomad_embed_ragged_bf16x3": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_int, C.POINTER(C.c_int), _fp, _fp, C.c_size_t, _fp]),
    "nomad_diag_attention_bf16x3": (C.c_int, [C.c_void_p, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp]),
    "nomad_diag_split_bf16": (C.c_int, [C.c_void_p, _fp, _fp, C.c_longlong, C.c_longlong, C.c_int, _fp]),
    "nomad_diag_gemm_bf16x3": (C.c_int, [C.c_void_p, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "nomad_diag_gemm_bf16": (C.c_int, [C.c_void_p, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "nomad_diag_attention_bf16": (C.c_int, [C.c_void_p, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp]),
    "nomad_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "nomad_profile_reset": (C.c_int, [C.c_void_p]),
    "nomad_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double)]),
    "nomad_diag_gemm": (C.c_int, [C.c_void_p, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "nomad_diag_clock_probe": (C.c_int, [C.c_void_p, C.c_ulonglong, _fp, _fp]),
    "nomad_diag_layernorm": (C.c_int, [C.c_void_p, _fp, _fp, _fp, _fp, C.c_int, C.c_int, _fp]),
    "nomad_diag_attention": (C.c_int, [C.c_void_p, _fp, _fp, C.c_int, C.c_int, _fp]),
    "nomad_diag_layernorm_bwd": (C.c_int, [C.c_void_p, _fp, _fp, _fp, _fp, C.c_int, C.c_int, _fp]),
    "nomad_diag_attention_bwd": (C.c_int, [C.c_void_p, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, _fp]),
    "nomad_diag_keep_intermediates": (C.c_int, [C.c_void_p, C.c_int]),
    "nomad_diag_workspace_region": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p,
                                              C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
}

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libnomad_hip.so")
DIAG_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libnomad_diag.so")
_libs = {}


ABI_VERSION = 3   # == NOMAD_ABI_VERSION in include/nomad_hip.h (tests/test_abi.py compares the two)


def load(diag=None):
    """Load an in-tree shared library; raises if it has not been built (no fallback exists).

    diag=False (default): libnomad_hip.so, the product.  diag=True (or ``NOMAD_DIAG_LIB=1`` in the environment when the
    argument is left None): libnomad_diag.so - the same exports plus every experimental kernel instantiation, for the
    measurement tools and the tests of those kernels."""
    if diag is None:
        diag = os.environ.get("NOMAD_DIAG_LIB", "0") == "1"
    diag = bool(diag)
    if diag in _libs:
        return _libs[diag]
    path = DIAG_LIB_PATH if diag else LIB_PATH
    variant = os.environ.get("NOMAD_LIB_VARIANT")     # A/B builds (python -m nomad_amd.build --pk / --variant <name>): measurement tools
    if variant and variant != "main":                 # and the packed-FP32 hazard's reproducer only ("main": the library itself, for alternating runs)
        path = path[:-3] + f"_{variant}.so"
    if not os.path.isfile(path):
        raise RuntimeError(f"{path} is missing: build it with `python -m nomad_amd.build` "
                           "(hipcc --offload-arch=gfx950). nomad_amd has no CPU or PyTorch fallback path.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    # the binding above was written against ONE layout of the entry points (include/nomad_hip.h: NOMAD_ABI_VERSION): refuse a
    # library built from another one before any call that takes arguments
    if lib.nomad_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path} reports ABI {lib.nomad_abi_version()}, this binding is written against {ABI_VERSION}: rebuild it "
                           "(`python -m nomad_amd.build`)")
    _libs[diag] = lib
    return lib


# nomad_status (include/nomad_hip.h)
NOMAD_OK, NOMAD_ERR_INVALID, NOMAD_ERR_NO_DEVICE, NOMAD_ERR_HIP, NOMAD_ERR_WORKSPACE, NOMAD_ERR_IO, NOMAD_ERR_FORMAT = 0, -1, -2, -3, -4, -5, -6


class NomadHipError(RuntimeError):
    pass


def check(rc: int, what: str):
    if rc != 0:
        msg = b"; ".join(m for m in (lib.nomad_last_error() for lib in _libs.values()) if m)
        raise NomadHipError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")
